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
    tr = sharded.DistTransport(dist, None, dev)
    assert not tr.gloo
    if mode == "transport":
        # the eight hand-off words through all_gather_into_tensor on device tensors
        words = [5 * (1 << 32) + 7, -1, -2, 0, 1 << 20, 0, 0, (1 << 40) + 3]
        assert tr.allgather(words) == [words]
        tr.exchange([], None, None)                  # no peers: no P2P op, no hang
        # a sharded step of bench.py's shard over this transport (world 1), plain and pipelined
        ctx = hip.Context(0)
        sh = sharded.SyntheticShard(ctx, "single", 256 << 20, 0, 1, dev, transport=tr)
        ctx.reserve(sh.ext.numel())
        table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
        out = sh.scan(table)
        sh.verify(table, out)
        assert out.record_base == 0 and out.total_records == out.n_own_records
        sh.make_lanes(2)
        tabs = (table, torch.empty_like(table))
        sh.submit(0, tabs[0])
        for i in range(1, 4):
            sh.submit(i & 1, tabs[i & 1])
            sh.verify(tabs[(i - 1) & 1], sh.finish((i - 1) & 1))
        sh.verify(tabs[1], sh.finish(1))
        # ordering: an RCCL collective on the hand-off stream writes the first MiB of the buffer, the
        # scan stream waits for its end event (HipBackend.comm_context) -- the rows can only come out
        # right if the scan read what the collective wrote
        be = sharded.HipBackend(ctx)
        good = sh.ext[:1 << 20].clone()
        for rep in range(8):
            sh.ext[:1 << 20].zero_()
            torch.cuda.synchronize()
            with be.comm_context(sh.ext):
                dist.all_gather_into_tensor(sh.ext[:1 << 20], good)
            rc, res = ctx.scan_device(sh.ext.data_ptr(), sh.n_own_bytes, table.data_ptr(), table.shape[0])
            assert rc == hip.OK and int(res.n_records) == out.n_own_records, (rep, int(res.n_records))
            k = torch.arange(0, 4096, dtype=torch.int64, device=dev) * 322
            assert bool((table[:4096, 0] == k).all()), "the scan ran ahead of the collective on the hand-off stream"
        print("nccl transport ok", flush=True)
    elif mode == "native":
        # the step behind the C ABI on the library's own RCCL transport (world 1: no peers, but the communicators, the
        # gather and the queueing are the product's)
        ctx = hip.Context(0)
        sh = sharded.SyntheticShard(ctx, "single", 256 << 20, 0, 1, dev, transport=tr, native=True)
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
        sw = sharded.SyntheticShard(ctx2, "wrapped", 64 << 20, 0, 1, dev, transport=tr, native=True)
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
        a = torch.arange(1 << 20, dtype=torch.uint8, device=dev) if False else (torch.arange(1 << 20, device=dev) % 251).to(torch.uint8)
        b = torch.zeros_like(a)
        try:
            tr.exchange([(0, 0, 0, 1 << 20)], lambda lo, hi: a[lo:hi], lambda lo, hi: b[lo:hi])
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            print("self-p2p unsupported: %r" % (e,), flush=True)
            sys.exit(3)
        assert bool((a == b).all())
        print("nccl self p2p ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
