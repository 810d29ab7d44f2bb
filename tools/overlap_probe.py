"""Steps per second of a workload with the steps queued (a) one ahead on ONE stream (what bench.py does),
(b) on two / three contexts with streams of their own, same input buffer, (c) the same with an input buffer
per context (no cache reuse between concurrent scans)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, sharded
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "single-1g"
wl = bench.WORKLOADS[name]
decode = wl["decode"]
dev = torch.device("cuda", 0)
base = hip.Context(0)
shard = sharded.SyntheticShard(base, wl["kind"], wl["bytes"], 0, 1, dev)
nb = shard.n_own_bytes
nsteps = max(8, int(0.25 / (wl["bytes"] / 4.0e12)))


def trial(nctx, own_streams, own_buffers, poll):
    flags = hip.F_DECODE_QUAL if decode else (hip.F_POLL_RESULT if poll else 0)
    ctxs = [base] + [hip.Context(0) if own_streams else hip.Context(share=base) for _ in range(nctx - 1)]
    bufs = [shard.ext] + [shard.ext.clone() if own_buffers else shard.ext for _ in range(nctx - 1)]
    tabs = [torch.empty((shard.max_records + 64, 6), dtype=torch.int64, device=dev) for _ in range(nctx)]
    quals = [torch.empty(shard.ext.numel(), dtype=torch.int8, device=dev) if decode else None for _ in range(nctx)]
    qoffs = [torch.empty(tabs[0].shape[0] + 1, dtype=torch.int64, device=dev) if decode else None for _ in range(nctx)]
    for c in ctxs:
        c.reserve(shard.ext.numel())

    def submit(i):
        k = i % nctx
        ctxs[k].scan_submit(bufs[k].data_ptr(), nb, tabs[k].data_ptr(), tabs[k].shape[0], sentinel=True, eof=True, flags=flags,
                            d_qual=quals[k].data_ptr() if decode else None, qual_cap=quals[k].numel() if decode else 0,
                            d_qoff=qoffs[k].data_ptr() if decode else None)

    def wait(i):
        rc, res = ctxs[i % nctx].scan_wait()
        assert rc == hip.OK and res.path in (0, 3)
        return res

    def run(n):
        idx = []
        for i in range(min(nctx - 1, n)):
            submit(i)
        for i in range(nctx - 1, n):
            submit(i)
            idx.append(wait(i - (nctx - 1)).ms_index)
        for i in range(max(n - (nctx - 1), 0), n):
            idx.append(wait(i).ms_index)
        return idx

    run(nsteps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = run(nsteps)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / nsteps
    for c in ctxs[1:]:
        c.close()
    return el, sum(idx) / len(idx)


for rnd in range(2):
    for nctx, own_s, own_b in ((2, False, False), (2, True, False), (2, True, True), (3, True, True), (4, True, True)):
        el, ki = trial(nctx, own_s, own_b, True)
        print("%s round %d: %d contexts, %s, %s: %.4f ms per step (%.2f TB/s of file bytes), index kernel %.1f us between its events"
              % (name, rnd, nctx, "own streams" if own_s else "one stream", "own buffers" if own_b else "one buffer",
                 el * 1e3, nb / el / 1e12, ki * 1e3), flush=True)
