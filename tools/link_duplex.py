"""The link in both directions at once: pinned -> device and device -> pinned copies of 1 GiB, each alone and together
(two streams).  What bounds a stream front end that sends rows AND decoded qualities back while the next chunks come in."""
import time
import torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_out = torch.ones(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, frac_back=1.0):
    torch.cuda.synchronize()
    m = int(n * frac_back)
    t0 = time.perf_counter()
    if h2d:
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)
    if d2h:
        with torch.cuda.stream(s2):
            h_out[:m].copy_(d_out[:m], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(2):
    run(True, True)
for name, a, b, f in (("host -> device alone", True, False, 1.0), ("device -> host alone", False, True, 1.0),
                      ("both, 1 GiB each way", True, True, 1.0), ("both, 0.67 GiB back per GiB in (the decode stream's ratio)", True, True, 0.67),
                      ("both, 0.15 GiB back per GiB in (rows only)", True, True, 0.15)):
    best = min(run(a, b, f) for _ in range(5))
    moved = (n if a else 0) + (int(n * f) if b else 0)
    print("%-62s %6.2f ms  in %5.1f GB/s  total %5.1f GB/s" % (name, best * 1e3, (n / best / 1e9) if a else 0.0, moved / best / 1e9), flush=True)

# ---- the same bytes as the stream front end moves them: 16 MiB chunks in, each in two halves on TWO streams (as the loader
# does: one engine gives 41-45 GB/s), and per chunk 11 MB back on a third -- by the copy engine, or by a KERNEL that writes
# the pinned memory itself
piece = 16 << 20
back = int(piece * 0.67)
s3 = torch.cuda.Stream()


def chunked(two_in_streams, back_mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n // piece):
        a = k * piece
        if two_in_streams:
            with torch.cuda.stream(s1):
                d_in[a:a + piece // 2].copy_(h_in[a:a + piece // 2], non_blocking=True)
            with torch.cuda.stream(s2):
                d_in[a + piece // 2:a + piece].copy_(h_in[a + piece // 2:a + piece], non_blocking=True)
        else:
            with torch.cuda.stream(s1):
                d_in[a:a + piece].copy_(h_in[a:a + piece], non_blocking=True)
        if back_mode:
            with torch.cuda.stream(s3):
                if back_mode == "engine":
                    h_out[a:a + back].copy_(d_out[a:a + back], non_blocking=True)
                else:
                    h_out_dev[a:a + back].copy_(d_out[a:a + back])          # a device kernel writing host-mapped memory
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# the pinned output buffer as a DEVICE tensor (its pages are mapped into the GPU's address space)
h_out_dev = None
try:
    import ctypes
    class _V:
        pass
    v = _V()
    v.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (h_out.data_ptr(), False), "version": 2}
    h_out_dev = torch.as_tensor(v, device="cuda")
except Exception as e:      # noqa: BLE001
    print("no device view of the pinned buffer:", e)
for name, two, mode in (("chunks in on one stream, nothing back", False, None), ("chunks in on two streams, nothing back", True, None),
                        ("one stream in, copy engine back", False, "engine"), ("two streams in, copy engine back", True, "engine"),
                        ("one stream in, kernel back", False, "kernel"), ("two streams in, kernel back", True, "kernel")):
    if mode == "kernel" and h_out_dev is None:
        continue
    best = min(chunked(two, mode) for _ in range(4))
    print("%-45s %6.2f ms  in %5.1f GB/s%s" % (name, best * 1e3, n / best / 1e9, ("  back %5.1f GB/s" % (back * (n // piece) / best / 1e9)) if mode else ""), flush=True)
