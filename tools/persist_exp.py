"""k_scan_lines_p (persistent workgroups) against k_scan_lines in ONE process: index-kernel time per configuration.
usage: persist_exp.py [bytes] [single|wrapped]   (FFQ_SCAN_PEXP 31 / 33 leave the entry stores out: the index of the
scan before is still there, so the rest of the scan runs as usual)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
from fastqandfurious_amd.sharded import SyntheticShard
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
kind = sys.argv[2] if len(sys.argv) > 2 else "single"
ctx = hip.Context(0)
sh = SyntheticShard(ctx, kind, nbytes, 0, 1, torch.device("cuda:0"))
n = sh.ext_scanned_bytes
cap = sh.max_records
table = torch.empty((cap, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n)
def run(g, x, reps=6):
    if g: os.environ["FFQ_SCAN_PERSIST"] = str(g)
    else: os.environ.pop("FFQ_SCAN_PERSIST", None)
    os.environ["FFQ_SCAN_PEXP"] = str(x)
    t = []
    for i in range(reps):
        rc, res = ctx.scan_device(sh.ext.data_ptr(), n, table.data_ptr(), cap)
        t.append(res.ms_index * 1e3)
    print("G=%-5d exp=%-3d index kernel %s us (path %d, n %d)" % (g, x, " ".join("%.1f" % v for v in t[1:]), res.path, res.n_records), flush=True)
for rnd in range(2):
    for g, x in ((0, 0), (2048, 0), (2048, 31), (2048, 32), (2048, 33), (1024, 0), (1024, 33), (1536, 0), (1792, 0), (4096, 0)):
        run(g, x)
