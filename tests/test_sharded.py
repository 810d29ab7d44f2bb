"""Byte-range sharding host logic (fastq-and-furious_amd/sharded.py) over gloo,
world_size 2 and 3, on CPU tensors.  The scan engine is the CPU oracle (test
infrastructure); what is under test is the edge hand-off, the row ownership
cut, the 8-byte verification and the record-ordinal all_gather."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    def __init__(self):
        from oracle import ffq_oracle
        self.o = ffq_oracle

    def scan(self, ext, n_bytes, sentinel, offset, eof, add, table, flags=0, qual=None, qoff=None,
             table_cap=None):
        data = ext[:n_bytes].numpy()
        t, end, status, off = self.o.scan(data, sentinel=sentinel, offset=offset, eof=eof, add=add)
        n = len(t)
        table[:n] = torch.from_numpy(t)
        res = types.SimpleNamespace(n_records=n, end_state=end, end_offset=off, last_status=status,
                                    last_pos=[-1] * 6, path=0, n_qual_bytes=0)
        if end != 0 and end != 1:
            st, pos = self.o.entrypos(np.concatenate([np.array([10], dtype=np.uint8), data])
                                      if sentinel else data, off, 0)
            res.last_pos = [int(p) + add if p >= 0 else -1 for p in pos]
        return 0, res

    # submit / wait: the engine runs at wait time (what matters is the protocol above it)
    def scan_submit(self, *a, **kw):
        self._queued = (a, kw)

    def scan_wait(self):
        a, kw = self._queued
        return self.scan(*a, **kw)

    def lower_bound(self, table, n_rows, value):
        return int(np.searchsorted(table[:n_rows, 0].numpy(), value, side="left"))

    def row(self, table, idx):
        return [int(x) for x in table[idx]]

    def sync_inputs(self):
        pass


def _worker(rank, world, port, kind, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqandfurious_amd  # noqa: F401
        from fastqandfurious_amd import sharded, synth
        from oracle import ffq_oracle
        if kind == "single":
            stream = synth.single(0, 14000, seed=42)
        elif kind == "wrapped":
            stream, _ = synth.wrapped(0, 12000, seed=43)
        else:   # a stream whose run-in starts on a false '@' candidate that dies INVALID
            stream, _ = synth.wrapped(0, 12000, seed=43)
        total = stream.size
        S = sharded.shard_bounds(total, world)
        lo, hi = S[rank], S[rank + 1]
        tail = min(sharded.TAIL_BYTES, S[rank] - S[rank - 1]) if rank > 0 else 0
        head = min(sharded.HEAD_BYTES, S[rank + 2] - S[rank + 1]) if rank < world - 1 else 0
        ext = torch.zeros(tail + (hi - lo) + head, dtype=torch.uint8)
        ext[tail:tail + hi - lo] = torch.from_numpy(stream[lo:hi].copy())
        # P2P sizes must agree on both sides: neighbours send min(HEAD/TAIL, their n_own)
        sharded.exchange_edges(dist, ext, tail, hi - lo, head, rank, world)
        assert bytes(ext.numpy()) == bytes(stream[lo - tail:hi + head]), "edge bytes differ"
        table = torch.empty((20000, 6), dtype=torch.int64)
        sc = sharded.ShardScanner(OracleBackend(), rank, world, dist, None, torch.device("cpu"))
        out = sc.scan(ext, tail, hi - lo, head, lo, hi, table)
        # the same step through submit / finish, two lanes, two rounds (the queue bench.py keeps)
        lanes = [sc, sharded.ShardScanner(OracleBackend(), rank, world, dist, None, torch.device("cpu"))]
        tabs = [table, torch.empty_like(table)]
        lanes[0].submit(ext, tail, hi - lo, head, lo, hi, tabs[0])
        for i in range(1, 4):
            lanes[i & 1].submit(ext, tail, hi - lo, head, lo, hi, tabs[i & 1])
            o2 = lanes[(i - 1) & 1].finish()
            assert (o2.row_lo, o2.row_hi, o2.exit_pos, o2.first_pos, o2.record_base) == \
                   (out.row_lo, out.row_hi, out.exit_pos, out.first_pos, out.record_base)
            assert (tabs[(i - 1) & 1][:o2.n_rows] == table[:out.n_rows]).all()
        lanes[3 & 1].finish()
        want, end, st, off = ffq_oracle.scan(stream)
        mine = want[(want[:, 0] >= lo) & (want[:, 0] < hi)]
        got = table[out.row_lo:out.row_hi].numpy()
        assert got.shape == mine.shape and (got == mine).all(), "shard rows differ from the single-range table"
        first = int(np.searchsorted(want[:, 0], lo))
        assert out.record_base == first
        assert out.total_records == len(want)
        np.save(os.path.join(tmpdir, "rows_%d.npy" % rank), got)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,kind", ((2, "single"), (2, "wrapped"), (3, "wrapped")))
def test_sharded_scan_gloo(tmp_path, oracle, world, kind):
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    stream = synth.single(0, 14000, seed=42) if kind == "single" else synth.wrapped(0, 12000, seed=43)[0]
    want, *_ = oracle.scan(stream)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "rows_%d.npy" % r)) for r in range(world)])
    assert (got == want).all()


def test_shard_bounds(pkg):
    from fastqandfurious_amd import sharded
    b = sharded.shard_bounds(1000003, 4)
    assert b[0] == 0 and b[-1] == 1000003 and all(x % 16 == 0 for x in b[:-1]) and b == sorted(b)
