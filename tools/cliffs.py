"""Timing of 1 GiB of S-single with one local irregularity in the middle (which tier ends up
doing the work, and how long it takes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
n = (1 << 30) // 322
base = torch.empty(n * 322 + (1 << 20), dtype=torch.uint8, device='cuda')
ctx.synth_single(base.data_ptr(), 0, n, 42)
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
mid = (n // 2) * 322
cases = {
    "clean": None,
    "one wrapped record": ("wrap", None),
    "100 KB of blank lines": ("blank", 100 << 10),
    "one record cut short (INVALID)": ("cut", None),
    "one 1 MB wrapped record": ("longwrap", 1 << 20),
    "one 1 MB four-line record": ("long4", 1 << 20),
    "100 KB of 10-base reads": ("tiny", 100 << 10),
    "2 MB of blank lines at the end": ("blankend", 2 << 20),
}
for name, what in cases.items():
    buf = base.clone()
    nb = n * 322
    if what:
        kind, arg = what
        if kind == "wrap":
            buf[mid + 18 + 75] = 10                 # a newline inside the sequence line
        elif kind == "blank":
            buf[mid:mid + arg] = 10                 # records replaced by newlines
        elif kind == "cut":
            buf[mid + 171 + 10] = 10                # a newline inside the quality line
        elif kind in ("longwrap", "long4"):
            L = arg // 2
            w = 80 if kind == "longwrap" else L
            seq = np.full(L, 65, dtype=np.uint8); q = np.full(L, 73, dtype=np.uint8)
            fold = lambda a: b"\n".join(a[i:i + w].tobytes() for i in range(0, L, w))
            rec = b"@long\n" + fold(seq) + b"\n+\n" + fold(q) + b"\n"
            pad = (-len(rec)) % 322
            rec = b"@long" + b"x" * pad + rec[5:]
            assert len(rec) % 322 == 0
            buf[mid:mid + len(rec)] = torch.from_numpy(np.frombuffer(rec, dtype=np.uint8).copy()).cuda()
        elif kind == "tiny":
            rec = b"".join(b"@t%06d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(arg // 32))
            rec = rec[:len(rec) // 322 * 322 // 32 * 32]
            m = len(rec) // 322 * 322
            buf[mid:mid + len(rec)] = torch.from_numpy(np.frombuffer(rec, dtype=np.uint8).copy()).cuda()
            buf[mid + len(rec):mid + m + 322] = 10  # (blank up to the next record boundary)
        elif kind == "blankend":
            buf[nb - arg:nb] = 10
    torch.cuda.synchronize()          # (the scan runs on the context's own stream)
    ctx.forget()
    for rep in range(3):
        t0 = time.perf_counter()
        rc, res = ctx.scan_device(buf.data_ptr(), nb, table.data_ptr(), n + 64)
        el = time.perf_counter() - t0
        print('   rep', rep, 'path', res.path, '%.3f ms' % (el * 1e3), 'index %.3f chain %.3f' % (res.ms_index, res.ms_chain))
    print("%-32s: %8.3f ms  n %d path %d retries %d end_state %d status %d" % (name, el * 1e3, res.n_records, res.path, res.retries, res.end_state, res.last_status), flush=True)
    del buf

# a whole buffer of very short reads with short headers (every tile dense)
for bases, nrec in ((12, 4000000), (18, 3000000)):
    rec = b"".join(b"@s%d\n%s\n+\n%s\n" % (i, b"ACGTACGTACGTACGTACGT"[:bases], b"IIIIIIIIIIIIIIIIIIII"[:bases]) for i in range(nrec))
    buf = torch.from_numpy(np.frombuffer(rec, dtype=np.uint8).copy()).cuda()
    tb = torch.empty((nrec + 64, 6), dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    ctx.forget()
    for rep in range(2):
        t0 = time.perf_counter()
        rc, res = ctx.scan_device(buf.data_ptr(), len(rec), tb.data_ptr(), nrec + 64)
        el = time.perf_counter() - t0
    print("%d MB of %d-base reads, short headers: %9.3f ms (%.2f GB/s)  n %d path %d retries %d" % (len(rec) >> 20, bases, el * 1e3, len(rec) / el / 1e9, res.n_records, res.path, res.retries), flush=True)
