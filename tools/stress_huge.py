"""GiB-sized hostile inputs (a 16-32 MiB block of distinct hostile text repeated: tens of thousands of groups, every
tier at size) against the oracle: full table, end state, decoded qualities.  tools/stress_huge.py [seeds] [GiB]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle
import test_gpu_parity as T
ctx = hip.Context(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
bad = 0
for seed in range(nseeds):
    rng = np.random.default_rng(880000 + int(os.environ.get("FFQ_STRESS_SEED0", "0")) + seed)
    kind = seed % 4
    if kind == 0: block = T._mess(rng, 150000, fatal=False)
    elif kind == 1: block = T.mutate(rng, T.random_records(rng, 60000, 100, 300), 8)
    elif kind == 2: block = T.mutate(rng, T.random_records(rng, 6000, 50, 12000, wrap=int(rng.integers(60, 101))), 4)
    else: block = T.random_records(rng, 30000, 30, 400, wrap=70) + T._mess(rng, 30000, fatal=False) + T.random_records(rng, 500, 5000, 40000)
    block = np.frombuffer(bytes(block), dtype=np.uint8)
    if block[-1] != 10:
        block = np.concatenate([block, np.frombuffer(b"\n", dtype=np.uint8)])
    reps = max(1, int(gib * (1 << 30)) // block.size)
    big = np.tile(block, reps)
    t0 = time.perf_counter()
    want, end, status, off = oracle.scan(big)
    wq, wqoff = oracle.decode_quals(big, want)
    t_or = time.perf_counter() - t0
    d = torch.from_numpy(big).cuda()
    n = len(want)
    cap = n + 64
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    qual = torch.empty(((big.size + 16383) >> 14) * 16384, dtype=torch.int8, device="cuda")
    qoff = torch.empty(cap + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.reserve(d.numel())
    wt = torch.from_numpy(want).cuda()
    wqt = torch.from_numpy(wq).cuda()
    wqo = torch.from_numpy(wqoff).cuda()
    for name, flags in (("default", 0), ("general", hip.F_FORCE_GENERAL), ("ranked", hip.F_FORCE_RANKED), ("decode", hip.F_DECODE_QUAL),
                        ("single pass", hip.F_DECODE_QUAL | hip.F_SINGLE_PASS)):
        ctx.forget()
        for it in range(2):
            table.zero_(); torch.cuda.synchronize()
            rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), cap, flags=flags, d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
            ok = rc == 0 and int(res.n_records) == n and int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
            ok = ok and bool((table[:n] == wt).all())
            if ok and (flags & hip.F_DECODE_QUAL) and n:
                lens = wt[:, 5] - wt[:, 4]
                q0 = qoff[:n]
                if res.path != 6:
                    ok = bool((qoff[:n + 1] == wqo).all()) and bool((qual[:wqt.numel()] == wqt).all())
                else:
                    idx = torch.repeat_interleave(q0 - wqo[:n], lens) + torch.arange(wqt.numel(), device="cuda")
                    ok = bool((qual[idx] == wqt).all())
            print("seed %d kind %d %.2f GiB %8d records (oracle %.1f s) %-11s run %d: path %d retries %d %.2f ms %s" %
                  (seed, kind, big.size / (1 << 30), n, t_or, name, it, res.path, res.retries, res.ms_total, "ok" if ok else "MISMATCH"), flush=True)
            bad += 0 if ok else 1
    del d, table, qual, qoff, wt, wqt, wqo
print("seeds", nseeds, "mismatches", bad)
