import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth
ctx = hip.Context(0)
n = (1<<30)//322
buf = torch.empty(n*322+64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
table = torch.empty((n+64,6), dtype=torch.int64, device='cuda')
ctx.reserve(n*322)
import numpy as np
for abl in (0,1,2,3,4,0):
    os.environ['FFQ_ABLATE'] = str(abl)
    ms = []
    for i in range(8):
        rc, res = ctx.scan_device(buf.data_ptr(), n*322, table.data_ptr(), n+64, flags=hip.F_FORCE_SERIAL if False else 0)
        ms.append((res.ms_index, res.ms_chain))
    ms = np.array(ms[2:])
    print("ablate", abl, "index %.1f us chain-total %.1f us" % (ms[:,0].mean()*1e3, ms[:,1].mean()*1e3), "path", res.path, flush=True)
