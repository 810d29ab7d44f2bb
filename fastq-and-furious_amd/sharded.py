"""Byte-range sharding of one FASTQ stream over the GPUs of a node.

The reference has nothing like this (it is a single-threaded generator); the
unit that shards is the record chain of readfastq_iter
(/root/reference/src/fastqandfurious.py:251-279): once a record start is known,
records are independent.

Rank r owns the bytes [S_r, S_{r+1}) of the stream and every record whose '@'
lies in that range.  One process per GPU; per step:

  1. edge hand-off (torch.distributed P2P = RCCL send/recv over xGMI): each
     rank sends the first HEAD bytes of its range to its left neighbour (so the
     neighbour can finish the record that straddles the edge) and the last TAIL
     bytes to its right neighbour (run-in: a chain started anywhere in it has
     re-synchronised with the true record chain before the range starts);
  2. one ordinary scan of [tail | own | head] on the local GPU;
  3. the rows with S_r <= pos0 < S_{r+1} are the shard's records;
  4. verification, 8 bytes per edge: the first record start at/after S_{r+1}
     as rank r sees it must equal the first row rank r+1 claims.  Together with
     rank 0's exact start this proves every shard's rows by induction;
  5. all_gather of the per-rank record counts -> global record ordinals.

No collective touches the data path; traffic is KiB per edge.

The scan engine is injected (`backend`): the product backend is HipBackend
(libffq_hip.so); tests drive the same host logic on CPU tensors over gloo with
a checker backend.
"""
import ctypes

import numpy as np

from . import hip as _hip

TAIL_BYTES = 1 << 20      # run-in taken from the left neighbour
HEAD_BYTES = 1 << 20      # look-ahead taken from the right neighbour


def shard_bounds(total_bytes, world):
    """S_0..S_world: 16-byte aligned cut points of the stream."""
    b = [(r * total_bytes // world) // 16 * 16 for r in range(world)]
    b.append(total_bytes)
    return b


class ScanOutput:
    def __init__(self, res, n_rows, row_lo, row_hi, exit_pos, first_pos):
        self.res = res
        self.n_rows = n_rows
        self.row_lo = row_lo
        self.row_hi = row_hi
        self.exit_pos = exit_pos
        self.first_pos = first_pos
        self.n_own_records = row_hi - row_lo
        self.record_base = 0          # global ordinal of this shard's first record


class HipBackend:
    """Scan engine on the local MI355X through the C ABI."""

    def __init__(self, ctx, post_ctx=None):
        self.ctx = ctx
        self.post = post_ctx or ctx     # context (stream) the small table queries run on

    def scan(self, ext, n_bytes, sentinel, offset, eof, add, table, flags=0, qual=None, qoff=None,
             table_cap=None):
        cap = table.shape[0] if table_cap is None else table_cap
        rc, res = self.ctx.scan_device(
            ext.data_ptr(), n_bytes, table.data_ptr(), cap, sentinel=sentinel, offset=offset,
            eof=eof, add=add, flags=flags,
            d_qual=qual.data_ptr() if qual is not None else None,
            qual_cap=qual.numel() if qual is not None else 0,
            d_qoff=qoff.data_ptr() if qoff is not None else None)
        return rc, res

    def scan_submit(self, ext, n_bytes, sentinel, offset, eof, add, table, flags=0, qual=None, qoff=None):
        self.ctx.scan_submit(
            ext.data_ptr(), n_bytes, table.data_ptr(), table.shape[0], sentinel=sentinel, offset=offset,
            eof=eof, add=add, flags=flags,
            d_qual=qual.data_ptr() if qual is not None else None,
            qual_cap=qual.numel() if qual is not None else 0,
            d_qoff=qoff.data_ptr() if qoff is not None else None)

    def scan_wait(self):
        return self.ctx.scan_wait()

    def lower_bound(self, table, n_rows, value):
        return self.ctx.table_lower_bound(table.data_ptr(), n_rows, 0, value)

    def cut(self, table, n_rows, lo, hi):
        """(i0, i1, pos0[i0], pos0[i1]) in one launch and one host wait."""
        return self.post.table_cut(table.data_ptr(), n_rows, lo, hi)

    def row(self, table, idx):
        out = np.empty(6, dtype=np.int64)
        self.ctx.d2h(out, table.data_ptr() + idx * 48)
        return [int(x) for x in out]

    def sync_inputs(self):
        import torch
        torch.cuda.synchronize()


def exchange_edges(dist, ext, tail, n_own, head, rank, world, group=None):
    """Fill ext[:tail] from the left neighbour's last bytes and
    ext[tail+n_own : tail+n_own+head] from the right neighbour's first bytes."""
    if world == 1:
        return
    ops = []
    own = ext[tail:tail + n_own]
    # a transport that cannot move device memory (gloo: the CPU tests, and the one-GPU dry run
    # of bench.py) gets the edges through host copies; RCCL moves them GPU to GPU
    staged = ext.is_cuda and dist.get_backend(group) == "gloo"
    recv = []

    def _send(t, peer):
        ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer, group))

    def _recv(t, peer):
        if staged:
            h = t.cpu()
            recv.append((t, h))
            t = h
        ops.append(dist.P2POp(dist.irecv, t, peer, group))

    if rank > 0:
        _send(own[:min(HEAD_BYTES, n_own)], rank - 1)
        _recv(ext[:tail], rank - 1)
    if rank < world - 1:
        _send(own[n_own - min(TAIL_BYTES, n_own):], rank + 1)
        _recv(ext[tail + n_own:tail + n_own + head], rank + 1)
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for t, h in recv:
        t.copy_(h)


class ShardScanner:
    """Steps 2-5 above for one rank."""

    def __init__(self, backend, rank, world, dist=None, group=None, device=None):
        self.backend = backend
        self.rank = rank
        self.world = world
        self.dist = dist
        self.group = group
        self.device = device

    def _start_offset(self, ext, tail, add, table):
        """A search offset inside the run-in whose chain survives into the own
        range.  A chain that starts at a false '@' candidate can stop at an
        INVALID entry right away; step past that candidate and try again."""
        if self.rank == 0 or tail == 0:
            return 0
        probe = min(ext.numel(), tail + (64 << 10))
        offset = 0
        for _ in range(32):
            rc, res = self.backend.scan(ext, probe, False, offset, False, add, table)
            if res.end_state == _hip.END_REFILL or res.end_offset >= tail:
                return offset
            if res.last_pos[0] < 0:
                break
            offset = int(res.last_pos[0]) - add      # the '@' of the failing entry: search after it
        raise RuntimeError("rank %d: no record chain survives the %d-byte run-in" % (self.rank, tail))

    def submit(self, ext, tail, n_own, head, own_lo_file, own_hi_file, table, flags=0, qual=None, qoff=None):
        """Enqueue the (offset 0) scan of a step and return; finish() completes the step.  With a
        second ShardScanner on a context that shares the stream, the next step is queued while
        this one is finished: the GPU does not idle during the host's part of a step."""
        rank, world = self.rank, self.world
        sentinel = rank == 0
        eof = rank == world - 1
        add = (own_lo_file - tail) - (1 if sentinel else 0)
        self.backend.scan_submit(ext, tail + n_own + head, sentinel, 0, eof, add, table, flags, qual, qoff)
        self._pending = (ext, tail, n_own, head, own_lo_file, own_hi_file, table, flags, qual, qoff)

    def finish(self):
        args = self._pending
        self._pending = None
        return self.scan(*args, first=self.backend.scan_wait())

    def scan(self, ext, tail, n_own, head, own_lo_file, own_hi_file, table, flags=0, qual=None, qoff=None,
             first=None):
        """ext = [tail | own | head] bytes (1-D uint8 tensor); the file offset of
        ext[tail] is own_lo_file.  Returns ScanOutput; table rows are absolute
        file offsets."""
        rank, world = self.rank, self.world
        sentinel = rank == 0
        eof = rank == world - 1
        ext_start = own_lo_file - tail
        add = ext_start - (1 if sentinel else 0)
        n_bytes = tail + n_own + head
        # Start the chain at the first "\n@" of the run-in.  If that is a false candidate the
        # chain re-synchronises long before the own range starts (the hand-off check below
        # proves it); only if such a chain stops at an invalid entry inside the run-in is a
        # start that survives searched for (a scan of the run-in alone) and the scan redone.
        offset = 0
        for attempt in range(2):
            if first is not None:
                rc, res = first          # the offset-0 scan was submitted ahead (submit / finish)
                first = None
            else:
                rc, res = self.backend.scan(ext, n_bytes, sentinel, offset, eof, add, table, flags, qual, qoff)
            if rc != _hip.OK:
                raise RuntimeError("rank %d: offset table too small (%d records)" % (rank, res.n_records))
            good = res.end_state == (_hip.END_OK if eof else _hip.END_REFILL)
            if good or rank == 0 or attempt == 1 or res.end_offset >= tail:
                break
            offset = self._start_offset(ext, tail, add, table)
        n = int(res.n_records)
        if eof:
            if res.end_state != _hip.END_OK:
                raise ValueError("stream does not end cleanly (end state %d at byte %d)"
                                 % (res.end_state, add + res.end_offset))
        elif res.end_state != _hip.END_REFILL:
            raise ValueError("rank %d: invalid entry at byte %d" % (rank, add + res.end_offset))
        cut = getattr(self.backend, "cut", None)
        if cut is not None:
            i0, i1, first_pos, exit_pos = cut(table, n, -(1 << 62) if rank == 0 else own_lo_file,
                                              (1 << 62) if eof else own_hi_file)
        else:
            i0 = 0 if rank == 0 else self.backend.lower_bound(table, n, own_lo_file)
            i1 = n if eof else self.backend.lower_bound(table, n, own_hi_file)
            first_pos = self.backend.row(table, i0)[0] if i0 < n else -1
            exit_pos = self.backend.row(table, i1)[0] if i1 < n else -1
        if not eof and exit_pos < 0:
            raise RuntimeError("rank %d: no complete record starts after byte %d within the %d-byte "
                               "look-ahead (record longer than the halo)" % (rank, own_hi_file, head))
        out = ScanOutput(res, n, i0, i1, exit_pos, first_pos)
        if world > 1:
            self._verify_and_count(out)
        return out

    def _verify_and_count(self, out):
        import torch
        dist, rank, world = self.dist, self.rank, self.world
        dev = self.device
        if dist.get_backend(self.group) == "gloo":
            dev = torch.device("cpu")
        mine = torch.tensor([out.exit_pos, out.n_own_records], dtype=torch.int64, device=dev)
        if dev.type == "cuda":
            # one collective into one tensor, one copy back
            flat = torch.empty(2 * world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(flat, mine, group=self.group)
            v = flat.tolist()
            allv = [v[2 * r:2 * r + 2] for r in range(world)]
        else:
            allv = [torch.empty(2, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(allv, mine, group=self.group)
            allv = [[int(x) for x in t.tolist()] for t in allv]
        if rank > 0 and allv[rank - 1][0] != out.first_pos:
            raise RuntimeError("rank %d: edge hand-off mismatch: left neighbour's chain enters this range "
                               "at byte %d, this rank started at %d" % (rank, allv[rank - 1][0], out.first_pos))
        out.record_base = sum(v[1] for v in allv[:rank])
        out.total_records = sum(v[1] for v in allv)


class SyntheticShard:
    """bench.py's input: this rank's byte range of a synthetic stream, built in
    HBM.  The logical stream is world * n_per records of S-single / S-wrapped
    (SURVEY.md 8d); cut points are moved off the record boundaries so that a
    record straddles every edge."""

    def __init__(self, ctx, kind, bytes_per_gpu, rank, world, dev, edge_shift=144):
        import torch
        from . import synth
        self.ctx, self.kind, self.rank, self.world, self.dev = ctx, kind, rank, world, dev
        self.dist = None
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
        if kind == "single":
            n_per = bytes_per_gpu // synth.RECORD_BYTES
            blk_bytes = [n_per * synth.RECORD_BYTES] * world
            starts = None
        else:
            n_per = int(bytes_per_gpu // 379.3)
            sizes = synth.wrapped_sizes(rank * n_per, n_per + 1, seed=43)
            starts = np.zeros(n_per + 2, dtype=np.int64)
            np.cumsum(sizes, out=starts[1:])
            mine = int(starts[n_per])
            if world > 1:
                t = torch.tensor([mine], dtype=torch.int64, device=dev)
                allt = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
                self.dist.all_gather(allt, t)
                blk_bytes = [int(x.item()) for x in allt]
            else:
                blk_bytes = [mine]
        self.n_per = n_per
        B = [0]
        for b in blk_bytes:
            B.append(B[-1] + b)
        total = B[-1]
        S = [0] + [(B[r] + edge_shift) // 16 * 16 for r in range(1, world)] + [total]
        self.own_lo, self.own_hi = S[rank], S[rank + 1]
        self.n_own_bytes = self.own_hi - self.own_lo
        self.tail = TAIL_BYTES if rank > 0 else 0
        self.head = HEAD_BYTES if rank < world - 1 else 0
        self.block_start = B[rank]

        # records [rank*n_per, (rank+1)*n_per (+1)) generated record-aligned, then the range is cut out
        n_gen = n_per + (1 if rank < world - 1 else 0)
        if kind == "single":
            gen_bytes = n_gen * synth.RECORD_BYTES
            tmp = torch.empty(gen_bytes + 64, dtype=torch.uint8, device=dev)
            ctx.synth_single(tmp.data_ptr(), rank * n_per, n_gen, seed=42)
        else:
            gen_bytes = int(starts[n_gen])
            tmp = torch.empty(gen_bytes + 64, dtype=torch.uint8, device=dev)
            dstart = torch.from_numpy(starts[:n_gen + 1].copy()).to(dev)
            torch.cuda.synchronize()
            ctx.synth_wrapped(tmp.data_ptr(), dstart.data_ptr(), rank * n_per, n_gen, seed=43)
            self.starts = starts
        self.ext = torch.zeros(self.tail + self.n_own_bytes + self.head + 64, dtype=torch.uint8, device=dev)
        a = self.own_lo - self.block_start
        self.ext[self.tail:self.tail + self.n_own_bytes] = tmp[a:a + self.n_own_bytes]
        del tmp
        torch.cuda.synchronize()
        self.ext_scanned_bytes = self.tail + self.n_own_bytes + self.head
        min_rec = 322 if kind == "single" else 120
        self.max_records = n_per + (self.tail + self.head) // min_rec + 64
        self.scanner = ShardScanner(HipBackend(ctx), rank, world, self.dist, None, dev)
        self._xstream = None
        self._lanes = None

    def scan(self, table, flags=0, qual=None, qoff=None):
        if self.world > 1:
            # the hand-off runs on the scan's own stream (RCCL orders itself against the current
            # stream): the scan that follows needs no host synchronisation in between
            import torch
            if self._xstream is None:
                self._xstream = torch.cuda.ExternalStream(self.ctx.stream(), device=self.dev)
            with torch.cuda.stream(self._xstream):
                exchange_edges(self.dist, self.ext, self.tail, self.n_own_bytes, self.head, self.rank, self.world)
        return self.scanner.scan(self.ext, self.tail, self.n_own_bytes, self.head, self.own_lo, self.own_hi,
                                 table, flags, qual, qoff)

    # ---- pipelined steps: submit(i + 1) before finish(i) -------------------------------------
    def make_lanes(self, n=2):
        """n scanners on contexts that share the scan stream (own scratch each) + one context
        with its own stream for the small queries of finish(), so that they do not queue
        behind the next step's kernels."""
        from . import hip
        post = hip.Context(self.ctx.device)
        lanes = [ShardScanner(HipBackend(self.ctx, post), self.rank, self.world, self.dist, None, self.dev)]
        for _ in range(n - 1):
            c = hip.Context(share=self.ctx)
            c.reserve(self.ext.numel())
            lanes.append(ShardScanner(HipBackend(c, post), self.rank, self.world, self.dist, None, self.dev))
        self._lanes = lanes
        self._post = post
        return lanes

    def submit(self, lane, table, flags=0, qual=None, qoff=None):
        import torch
        if self.world > 1:
            if self._xstream is None:
                self._xstream = torch.cuda.ExternalStream(self.ctx.stream(), device=self.dev)
            with torch.cuda.stream(self._xstream):
                exchange_edges(self.dist, self.ext, self.tail, self.n_own_bytes, self.head, self.rank, self.world)
        self._lanes[lane].submit(self.ext, self.tail, self.n_own_bytes, self.head, self.own_lo, self.own_hi,
                                 table, flags, qual, qoff)

    def finish(self, lane):
        return self._lanes[lane].finish()

    def host_sample(self, nbytes):
        """First whole records of this rank's range, on the host."""
        import torch
        skip = 0
        if self.rank > 0:
            skip = 322 - (self.own_lo - self.block_start)   # only used on rank 0 in practice
        n = min(nbytes, self.n_own_bytes - skip)
        if self.kind == "single":
            n = n // 322 * 322
        else:
            k = int(np.searchsorted(self.starts, n, side="right")) - 1
            n = int(self.starts[k])
        return self.ext[self.tail + skip:self.tail + skip + n].cpu().numpy()

    def verify(self, table, out):
        """Size-independent parity properties on the full-size output: the rows
        must equal the closed form of the generator (which the parity tests
        prove equal to the reference on the same bytes)."""
        import torch
        rows = table[out.row_lo:out.row_hi]
        n = rows.shape[0]
        # ownership is by '@' position: the first owned record is the first whose start >= own_lo
        if self.kind == "single":
            k0 = -(-self.own_lo // 322)
            k = torch.arange(k0, k0 + n, dtype=torch.int64, device=rows.device) * 322
            want = torch.stack([k, k + 17, k + 18, k + 168, k + 171, k + 321], dim=1)
            assert n == -(-self.own_hi // 322) - k0, "record count differs from the closed form"
            assert bool((rows == want).all()), "offset table differs from the closed form"
        else:
            st = torch.from_numpy(self.starts).to(rows.device) + self.block_start
            k0 = int(np.searchsorted(self.starts + self.block_start, self.own_lo, side="left"))
            k1 = int(np.searchsorted(self.starts + self.block_start, self.own_hi, side="left"))
            if self.rank == self.world - 1:
                k1 = self.n_per
            assert n == k1 - k0, "record count differs from the generator's"
            assert bool((rows[:, 0] == st[k0:k1]).all()), "record starts differ from the generator's"
            assert bool((rows[:, 1] == st[k0:k1] + 17).all())
            assert bool((rows[:, 5] == st[k0 + 1:k1 + 1] - 1).all()), "record ends differ"
            assert bool((rows[:, 5] - rows[:, 4] == rows[:, 3] - rows[:, 2]).all())

    def verify_decode(self, table, out, qual, qoff):
        """The decode's output at full size, with torch ops only: the CSR offsets must be the
        running sum of pos5 - pos4 over ALL rows of the scan, and the decoded bytes of a spread of
        records must be the buffer's bytes [pos4, pos5) minus 33 (int8 arithmetic)."""
        import torch
        n = int(out.n_rows)
        lens = table[:n, 5] - table[:n, 4]
        assert int(qoff[0].item()) == 0, "quality offsets do not start at 0"
        assert bool((qoff[1:n + 1] - qoff[:n] == lens).all()), "quality offsets are not the running sum of pos5 - pos4"
        assert int(qoff[n].item()) == int(out.res.n_qual_bytes), "closing quality offset differs from the reported total"
        if n == 0:
            return
        shift = self.own_lo - self.tail                      # file offset of ext[0]
        idx = torch.unique(torch.cat([torch.arange(0, min(n, 64), device=table.device),
                                      torch.linspace(0, n - 1, 4096, device=table.device).long(),
                                      torch.arange(max(n - 64, 0), n, device=table.device)]))
        p4 = table[idx, 4] - shift
        ln = lens[idx]
        q0 = qoff[idx]
        if self.kind == "single":
            assert bool((ln == 150).all())
            ar = torch.arange(150, device=table.device)
            src = self.ext[(p4[:, None] + ar[None, :]).reshape(-1)].to(torch.int16) - 33
            got = qual[(q0[:, None] + ar[None, :]).reshape(-1)].to(torch.int16)
            assert bool((src == got).all()), "decoded qualities differ from the buffer's bytes - 33"
        else:
            for a, l, q in zip(p4[::8].tolist(), ln[::8].tolist(), q0[::8].tolist()):
                src = (self.ext[a:a + l].to(torch.int16) - 33).to(torch.int8)
                assert bool((src == qual[q:q + l]).all()), "decoded qualities differ from the buffer's bytes - 33"

