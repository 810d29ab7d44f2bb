"""k_scan_fused (csrc/ffq_fused.h) with parts of it cut out -- instrumented build only:
python tools/fused_ablate.py [bytes]  (FFQ_FZ_ABLATE bits: 1 no gather, 2 no quality table, 4 no write-out,
8 no prefix, 16 no index stores; FFQ_FZ_GRID: persistent workgroups)"""
import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, R)
    import torch
    import fastqandfurious_amd
    from fastqandfurious_amd import hip
    hip.use_probe_build()
    nbytes = int(float(sys.argv[2]))
    ctx = hip.Context(0)
    n = nbytes // 322
    buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device="cuda")
    ctx.synth_single(buf.data_ptr(), 0, n, 42)
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    qual = torch.empty(n * 161 + 4096, dtype=torch.int8, device="cuda")
    qoff = torch.empty(n + 65, dtype=torch.int64, device="cuda")
    ctx.reserve(n * 322)
    ms = []
    for i in range(8):
        rc, res = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64, flags=hip.F_DECODE_QUAL,
                                  d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
        ms.append((res.ms_index, res.ms_total, res.path))
    print("ablate %s grid %s: k_scan_fused %.3f ms, step %.3f ms, path %d  (%.2f GiB)" % (
        os.environ.get("FFQ_FZ_ABLATE", "0"), os.environ.get("FFQ_FZ_GRID", "-"), min(m[0] for m in ms[2:]),
        min(m[1] for m in ms[2:]), ms[-1][2], n * 322 / 2**30), flush=True)
    sys.exit(0)
nbytes = sys.argv[1] if len(sys.argv) > 1 else "2e9"
for abl in os.environ.get("ABLS", "0 1 3 7 15 31 4 8 16").split():
    for grid in os.environ.get("GRIDS", "1024").split():
        env = dict(os.environ, FFQ_FZ_ABLATE=abl, FFQ_FZ_GRID=grid)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", nbytes], env=env, timeout=300)
