"""Worker of tests/test_config5.py: one process, backend nccl (= RCCL), world size 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip, sharded


def main(mode):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    if mode == "native":
        # the step behind the C ABI on the library's own RCCL transport (world 1: no peers, but the communicators, the
        # gather and the queueing are the product's)
        ctx = hip.Context(0)
        sh = sharded.SyntheticShard(ctx, "single", 256 << 20, 0, 1, dev, transport=dist, native=True)
        assert sh.native and sh.scanner.sh.transport() == "rccl"
        ctx.reserve(sh.ext.numel())
        table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
        out = sh.scan(table)
        sh.verify(table, out)
        assert out.record_base == 0 and out.total_records == out.n_own_records and out.rounds == 0
        assert out.comm is not None and out.comm["allgather_ms"] > 0 and out.comm["handoff_bytes"] == 0
        sh.make_lanes(2)
        tabs = (table, torch.empty_like(table))
        sh.submit(0, tabs[0])
        for i in range(1, 6):
            sh.submit(i & 1, tabs[i & 1])
            sh.verify(tabs[(i - 1) & 1], sh.finish((i - 1) & 1))
        sh.verify(tabs[1], sh.finish(1))
        # wrapped input: the first front of a fresh context is the fast path's, refused -- the gather is repeated once the
        # general kernels are through
        ctx2 = hip.Context(0)
        sw = sharded.SyntheticShard(ctx2, "wrapped", 64 << 20, 0, 1, dev, transport=dist, native=True)
        ctx2.reserve(sw.ext.numel())
        t2 = torch.empty((sw.max_records + 64, 6), dtype=torch.int64, device=dev)
        o2 = sw.scan(t2)
        sw.verify(t2, o2)
        assert o2.comm["regathers"] >= 1
        o2 = sw.scan(t2)
        sw.verify(t2, o2)
        assert o2.comm["regathers"] == 0
        a = (torch.arange(1 << 20, device=dev) % 251).to(torch.uint8)
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        try:
            sh.scanner.sh.self_exchange(a.data_ptr(), b.data_ptr(), a.numel())
        except Exception as e:      # noqa: BLE001
            print("self-p2p unsupported: %r" % (e,), flush=True)
            sys.exit(3)
        assert bool((a == b).all())
        print("native step on rccl ok", flush=True)
    else:
        raise SystemExit("unknown mode %r" % mode)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
