"""Where does a FIRST load into freshly allocated device memory lose its time?  (round 6, VERDICT weak #7)
  cold      fresh hipMalloc -> ffq_load_fd
  warm      the same buffer again
  slept     fresh hipMalloc -> 100 ms of nothing -> ffq_load_fd      (does the driver clear new VRAM on its own?)
  touched   fresh hipMalloc -> one pass of a kernel over it (timed apart) -> ffq_load_fd
One file in /dev/shm (page cache), sizes 0.5 / 2 / 8 GiB.   python tools/load_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip

import ctypes
ctx = hip.Context(0)
L = hip.lib()


def link(n=1 << 30, reps=3):
    """raw pinned -> device copy of n bytes on the context's stream (ffq_copy_h2d): what the link gives THIS box, now"""
    hp = ctypes.c_void_p()
    hip.check(L.ffq_pinned_alloc(n, ctypes.byref(hp)))
    ctypes.memset(hp, 1, n)
    d = ctx.dev_alloc(n)
    out = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        hip.check(L.ffq_copy_h2d(ctx.handle, ctypes.c_void_p(d), hp, n, 0))
        out.append(n / (time.perf_counter() - t0) / 1e9)
    ctx.dev_free(d)
    L.ffq_pinned_free(hp)
    return "/".join("%.1f" % x for x in out[1:])


print("raw link, pinned -> device, 1 GiB x3 (GB/s), before any kernel of this process:", link(), flush=True)
path = "/dev/shm/ffq_load_probe.bin"
blk = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8).tobytes()
for gib in (0.5, 2, 8):
    n = int(gib * (1 << 30))
    with open(path, "wb") as fh:
        for _ in range(n // len(blk)):
            fh.write(blk)
    fd = os.open(path, os.O_RDONLY)

    def load(d):
        t0 = time.perf_counter()
        got = ctx.load_fd(fd, 0, n, d)
        ctx.sync()
        assert got == n
        return time.perf_counter() - t0
    res = {}
    for rep in range(2):
        d = ctx.dev_alloc(n + 64); res.setdefault("cold", []).append(load(d)); res.setdefault("warm", []).append(load(d)); res.setdefault("warm2", []).append(load(d)); ctx.dev_free(d)
        d = ctx.dev_alloc(n + 64); time.sleep(0.1); res.setdefault("slept", []).append(load(d)); ctx.dev_free(d)
        d = ctx.dev_alloc(n + 64); t0 = time.perf_counter(); ctx.arrayadd_b_device(d, n, 0); ctx.sync(); tt = time.perf_counter() - t0
        res.setdefault("touch_ms", []).append(tt); res.setdefault("touched", []).append(load(d)); ctx.dev_free(d)
    os.close(fd)
    print("%.1f GiB:" % gib, "  ".join("%s %s" % (k, "/".join(("%.1f GB/s" % (n / t / 1e9)) if k != "touch_ms" else ("%.1f ms" % (t * 1e3)) for t in v)) for k, v in res.items()), flush=True)
print("raw link again, after the loads and the touch kernels:", link(), flush=True)
os.unlink(path)
