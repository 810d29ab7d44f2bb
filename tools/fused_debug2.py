import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["FFQ_DEBUG"] = "1"
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, sharded, index
hip.use_probe_build()
dev = torch.device("cuda:0")
ctx = hip.Context(0)
sh = sharded.SyntheticShard(ctx, "single", 10 << 30, 0, 1, dev)
table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
qual = torch.empty(sh.n_own_bytes // 2 + 4096, dtype=torch.int8, device=dev)
qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
def T(msg, f):
    torch.cuda.synchronize(); t0 = time.time(); r = f(); torch.cuda.synchronize(); print("%-30s %.3f s" % (msg, time.time() - t0), flush=True); return r
out = T("two-pass", lambda: sh.scan(table, flags=hip.F_DECODE_QUAL, qual=qual, qoff=qoff))
print("path", out.res.path)
n = int(out.n_rows)
if "--col" in sys.argv:
    seqs, soff = T("select_column", lambda: index.select_column_device(ctx, sh.ext, table[:n], "sequence"))
    del seqs, soff
t2 = torch.empty_like(table); q2 = torch.empty_like(qual); o2 = torch.empty_like(qoff)
if "--general" in sys.argv:
    ctx.forget()
    os.environ["FFQ_NO_FAST4"] = "1"
    out2 = T("general", lambda: sh.scan(t2, flags=hip.F_DECODE_QUAL, qual=q2, qoff=o2))
    del os.environ["FFQ_NO_FAST4"]
    print("path", out2.res.path)
ctx.forget()
if "--zero" in sys.argv:
    t2.zero_(); q2.zero_(); o2.zero_()
if "--nosync" in sys.argv:
    torch.cuda.synchronize()
    t2.zero_(); q2.zero_(); o2.zero_()
    t0 = time.time()
    out3 = sh.scan(t2, flags=hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, qual=q2, qoff=o2)
    print("single pass right behind torch's fills: path", out3.res.path, "%.3f s" % (time.time() - t0), flush=True)
out3 = T("single pass", lambda: sh.scan(t2, flags=hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, qual=q2, qoff=o2))
print("path", out3.res.path, "ms_index", out3.res.ms_index)
out3 = T("single pass again", lambda: sh.scan(t2, flags=hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, qual=q2, qoff=o2))
print("path", out3.res.path, "ms_index", out3.res.ms_index)
