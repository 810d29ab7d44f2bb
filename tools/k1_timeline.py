"""Index-kernel time of every step of a long pipelined run (does it depend on how long the GPU
has been busy?).  k1_timeline.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = (1 << 30) // 322
c0 = hip.Context(0); c1 = hip.Context(share=c0)
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
c0.synth_single(buf.data_ptr(), 0, n, 42)
tabs = [torch.empty((n + 64, 6), dtype=torch.int64, device='cuda') for _ in range(2)]
c0.reserve(n * 322); c1.reserve(n * 322)
cs = (c0, c1)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "0.5")))
out = []
t0 = time.perf_counter()
def sub(i): cs[i & 1].scan_submit(buf.data_ptr(), n * 322, tabs[i & 1].data_ptr(), n + 64, sentinel=True, eof=True)
sub(0)
for i in range(1, nsteps):
    sub(i)
    rc, res = cs[(i - 1) & 1].scan_wait()
    out.append((time.perf_counter() - t0, res.ms_index * 1e3, res.ms_chain * 1e3))
rc, res = cs[(nsteps - 1) & 1].scan_wait()
for k in range(0, len(out), max(1, len(out) // 40)):
    print("step %4d at %7.2f ms: index %.1f us chain %.1f us" % (k, out[k][0] * 1e3, out[k][1], out[k][2]))
