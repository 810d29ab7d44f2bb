"""Rates of the native stream front end over a file in /dev/shm: chunk sizes, file sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth
ctx = hip.Context(0)
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
n = nbytes // 322
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
host = buf[:n * 322].cpu().numpy()
path = "/dev/shm/ffq_rate_%d.fq" % os.getpid()
host.tofile(path)
try:
    # raw read rate of the file into ordinary memory, one thread
    t0 = time.perf_counter(); fd = os.open(path, os.O_RDONLY)
    got = 0
    while True:
        b = os.pread(fd, 1 << 24, got)
        if not b: break
        got += len(b)
    os.close(fd); el = time.perf_counter() - t0
    print("os.pread loop, one thread: %.1f GB/s" % (got / el / 1e9))
    for fb in (1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26):
        best = None
        for rep in range(4):
            fd = os.open(path, os.O_RDONLY)
            t0 = time.perf_counter()
            st = hip.FileStream(ctx, fd, fb)
            recs = sum(rows.shape[0] for rows, _f, _o, _e, _x in st)
            st.close(); os.close(fd)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
            assert recs == n
        print("fbufsize %3d MiB: %.1f GB/s  (%.1f M reads/s)" % (fb >> 20, n * 322 / best / 1e9, n / best / 1e6), flush=True)
    best = None
    for rep in range(3):
        fd = os.open(path, os.O_RDONLY)
        t0 = time.perf_counter()
        st = hip.FileStream(ctx, fd, 1 << 24, decode=True)
        recs = sum(rows.shape[0] for rows, _f, _o, _e, _x in st)
        st.close(); os.close(fd)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    print("fbufsize  16 MiB with decode: %.1f GB/s" % (n * 322 / best / 1e9))
finally:
    os.unlink(path)
