"""Throughput of the length-filter push-down (ffq_table_select_seqlen) on a table from a device scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, index
from fastqandfurious_amd.sharded import SyntheticShard
ctx = hip.Context(0)
sh = SyntheticShard(ctx, "wrapped", 4 << 30, 0, 1, torch.device("cuda:0"))
table = torch.empty((sh.max_records, 6), dtype=torch.int64, device="cuda")
rc, res = ctx.scan_device(sh.ext.data_ptr(), sh.ext_scanned_bytes, table.data_ptr(), sh.max_records)
n = int(res.n_records)
out = torch.empty_like(table)
for lo, hi in ((100, 200), (50, 300), (301, 400)):
    ctx.table_select_seqlen(table.data_ptr(), n, lo, hi, out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        k = ctx.table_select_seqlen(table.data_ptr(), n, lo, hi, out.data_ptr())
    el = (time.perf_counter() - t0) / 10
    traffic = n * 16 + n * 16 + k * 48 * 2          # pos2/pos3 twice, kept rows read + written
    print("select [%d, %d]: %d of %d rows kept, %.3f ms, %.1f G rows/s, %.2f TB/s of algorithmic traffic"
          % (lo, hi, k, n, el * 1e3, n / el / 1e9, traffic / el / 1e12))
