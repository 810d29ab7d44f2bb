"""Host time of ffq_scan_submit / ffq_scan_wait per step, two contexts one step ahead (as bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
from fastqandfurious_amd.sharded import SyntheticShard
kind = sys.argv[1] if len(sys.argv) > 1 else "wrapped"
nbytes = int(float(sys.argv[2])) if len(sys.argv) > 2 else (10 << 30)
ctx = hip.Context(0)
sh = SyntheticShard(ctx, kind, nbytes, 0, 1, torch.device("cuda:0"))
ctx2 = hip.Context(share=ctx)
for c in (ctx, ctx2):
    c.reserve(sh.ext.numel())
tabs = [torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device="cuda") for _ in range(2)]
cs = (ctx, ctx2)
def submit(i):
    cs[i & 1].scan_submit(sh.ext.data_ptr(), sh.n_own_bytes, tabs[i & 1].data_ptr(), tabs[i & 1].shape[0])
ts, tw = [], []
submit(0)
for i in range(1, 24):
    t0 = time.perf_counter(); submit(i); t1 = time.perf_counter()
    rc, res = cs[(i - 1) & 1].scan_wait(); t2 = time.perf_counter()
    ts.append((t1 - t0) * 1e3); tw.append((t2 - t1) * 1e3)
cs[23 & 1].scan_wait()
print(kind, "submit ms:", " ".join("%.2f" % x for x in ts[4:]))
print(kind, "wait   ms:", " ".join("%.2f" % x for x in tw[4:]))
