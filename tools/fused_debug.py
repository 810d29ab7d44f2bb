import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["FFQ_DEBUG"] = "1"
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()
ctx = hip.Context(0)
for nbytes in [int(float(x)) for x in sys.argv[1:]]:
    n = nbytes // 322
    buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device="cuda")
    ctx.synth_single(buf.data_ptr(), 0, n, 42)
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    qual = torch.empty(n * 161 + 4096, dtype=torch.int8, device="cuda")
    qoff = torch.empty(n + 65, dtype=torch.int64, device="cuda")
    for rep in range(3):
        ctx.forget()
        rc, res = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64, flags=hip.F_DECODE_QUAL | hip.F_SINGLE_PASS,
                                  d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
        print(nbytes, n, "path", res.path, "ms_index %.3f" % res.ms_index, "tail bytes", (n * 322) % 16384, flush=True)
    del buf, table, qual, qoff
    torch.cuda.empty_cache()
