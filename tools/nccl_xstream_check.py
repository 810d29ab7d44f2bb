"""One-GPU check of the plumbing the N>1 bench relies on: RCCL work issued under a
torch ExternalStream that wraps the scan context's HIP stream, followed by a scan on that
stream.  World size 1 (one box, one GPU): the collectives are trivial, the stream ordering
and torch's event handling on a foreign stream are what is exercised."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("fastq-and-furious_amd.hip")
synth = importlib.import_module("fastq-and-furious_amd.synth")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
ctx = hip.Context(0)
xs = torch.cuda.ExternalStream(ctx.stream(), device=dev)
data = synth.single(0, 5000).tobytes()
n = len(data)
ext = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
table = torch.empty((6000, 6), dtype=torch.int64, device=dev)
for it in range(3):
    with torch.cuda.stream(xs):
        ext[:n].copy_(src, non_blocking=True)          # stands in for the edge receive
        t = torch.ones(4, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        flat = torch.empty(4, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(flat, t)
    rc, res = ctx.scan_device(ext.data_ptr(), n, table.data_ptr(), 6000, sentinel=True, eof=True)
    assert rc == 0 and res.n_records == 5000, (rc, res.n_records)
    ext.zero_()
torch.cuda.synchronize()
print("ok: RCCL under ExternalStream + scan on the same stream, 3 rounds;", flat.tolist())
dist.destroy_process_group()
