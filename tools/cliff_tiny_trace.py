"""The '100 KB of 10-base reads' case of tools/cliffs.py alone, ten scans (for a kernel trace: tools/kstats_py.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
n = (1 << 30) // 322
buf = torch.empty(n * 322 + (1 << 20), dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
mid = (n // 2) * 322
arg = 100 << 10
rec = b"".join(b"@t%06d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(arg // 32))
rec = rec[:len(rec) // 322 * 322 // 32 * 32]
m = len(rec) // 322 * 322
buf[mid:mid + len(rec)] = torch.from_numpy(np.frombuffer(rec, dtype=np.uint8).copy()).cuda()
buf[mid + len(rec):mid + m + 322] = 10
torch.cuda.synchronize()
for rep in range(10):
    rc, res = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64)
print(res.path, res.ms_index, res.ms_chain)
