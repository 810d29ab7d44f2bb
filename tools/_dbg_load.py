import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip, sharded, synth
import torch
ctx = hip.Context(0)
n = (2 << 30) // 322
path = "/dev/shm/dbg_load.fq"
buf = torch.empty(n * 322, dtype=torch.uint8, device="cuda")
ctx.synth_single(buf.data_ptr(), 0, n, seed=42); ctx.sync()
open(path, "wb").write(buf.cpu().numpy().tobytes()); del buf
for rep in range(3):
    sh = sharded.FileShard(ctx, path, 0, 1)
    t0 = time.perf_counter(); nb = sh.load(); t1 = time.perf_counter(); res = sh.scan(); t2 = time.perf_counter()
    print("threads", os.environ.get("FFQ_POOL_THREADS", "16"), "load %.1f GB/s (%.1f ms) step %.2f ms" % (nb / (t1 - t0) / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    sh.close()
fd = os.open(path, os.O_RDONLY)
for fb in (16 << 20, 64 << 20):
    t0 = time.perf_counter(); st = hip.FileStream(ctx, fd, fb); r = sum(x[0].shape[0] for x in st); st.close()
    print("stream fbuf %d MiB: %.1f GB/s" % (fb >> 20, n * 322 / (time.perf_counter() - t0) / 1e9))
os.close(fd); os.unlink(path)
