"""Byte-range sharding of one FASTQ stream over the GPUs of a node.

The reference has nothing like this (it is a single-threaded generator); the
unit that shards is the record chain of readfastq_iter
(/root/reference/src/fastqandfurious.py:251-279): once a record start is known,
records are independent.

Rank r owns the bytes [S_r, S_{r+1}) of the stream and every record whose '@'
lies in that range.  One process per GPU; per step:

  1. halo hand-off (torch.distributed P2P = RCCL send/recv over xGMI): every
     rank receives the TAIL bytes in front of its range (run-in: a chain started
     anywhere in it has re-synchronised with the true record chain before the
     range starts) and the HEAD bytes behind it (to finish the record that
     straddles the edge), from whichever ranks own them;
  2. one ordinary scan of [tail | own | head] on the local GPU;
  3. the rows with S_r <= pos0 < S_{r+1} are the shard's records;
  4. verification, a few words per rank (one all_gather): the first record start
     at/after S_{r+1} as rank r sees it must equal the first record start rank
     r+1 sees in its range.  Together with rank 0's exact start this proves
     every shard's rows by induction;
  5. the same all_gather carries the per-rank record counts -> global ordinals.

What the reference does with a record that does not fit its buffer -- keep
`buf[offset:]` and read more until it does (:274-279) -- happens here per edge:
a rank whose look-ahead ends inside the record that straddles its right edge asks
for a larger one (doubling, served by however many ranks own those bytes) and
scans again; a rank whose guessed entry the left neighbour's chain contradicts is
scanned again from the neighbour's exit (no run-in speculation), and everything
is verified again.  Each round makes the first unsettled rank exact, so the
rounds terminate.  Errors of the stream (the iterator's three ValueErrors) are
raised by every rank together, and only once the failing rank's entry is proven.

No collective touches the data path; traffic is KiB per edge.

The scan engine and the transport are injected: the product pair is HipBackend
(libffq_hip.so) + DistTransport (torch.distributed: RCCL on GPUs, gloo in the CPU
tests); LocalTransport runs k logical ranks as threads of one process (k ranges
of one resident buffer on one GPU).
"""
import contextlib
import os
import threading

import numpy as np

from . import hip as _hip

TAIL_BYTES = 1 << 20      # run-in taken from the left
HEAD_BYTES = 1 << 20      # look-ahead taken from the right (grown when a record needs more)

NONE_POS = -1             # no record starts at / after the bound: the view reaches the end of the stream
UNKNOWN_POS = -2          # not known yet (more look-ahead needed, or the guessed entry led nowhere)
ERR_TABLE_FULL = 100      # (beside the END_ERR_* codes of the stream) the caller's table cannot hold a rank's rows


def shard_bounds(total_bytes, world):
    """S_0..S_world: 16-byte aligned cut points of the stream."""
    b = [(r * total_bytes // world) // 16 * 16 for r in range(world)]
    b.append(total_bytes)
    return b


def halo_sizes(bounds, rank, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES):
    """(tail, head) of rank's first scan: the same rule on every rank, so that each knows what
    the others need without asking."""
    lo, hi, total = bounds[rank], bounds[rank + 1], bounds[-1]
    return min(tail_bytes, lo - bounds[0]), min(head_bytes, total - hi)


def range_plan(bounds, dst, lo, hi):
    """[(src, dst, a, b)]: the pieces of stream bytes [lo, hi) by owner (dst's own bytes left out)."""
    plan = []
    for p in range(len(bounds) - 1):
        a, b = max(lo, bounds[p]), min(hi, bounds[p + 1])
        if a < b and p != dst:
            plan.append((p, dst, a, b))
    return plan


def halo_plan(bounds, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES):
    plan = []
    for q in range(len(bounds) - 1):
        t, h = halo_sizes(bounds, q, tail_bytes, head_bytes)
        plan += range_plan(bounds, q, bounds[q] - t, bounds[q])
        plan += range_plan(bounds, q, bounds[q + 1], bounds[q + 1] + h)
    return plan


# ---- transports -------------------------------------------------------------------------------
class SoloTransport:
    rank, world = 0, 1

    def allgather(self, vals):
        return [list(vals)]

    def exchange(self, plan, provide, accept):
        assert not plan


class DistTransport:
    """torch.distributed: backend "nccl" (= RCCL, device tensors move GPU to GPU over xGMI) or
    gloo (CPU tensors; device tensors are staged through host copies -- the one-GPU dry run)."""

    def __init__(self, dist, group=None, device=None):
        self.dist, self.group, self.device = dist, group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.gloo = dist.get_backend(group) == "gloo"

    def allgather(self, vals):
        import torch
        dev = torch.device("cpu") if self.gloo or self.device is None else self.device
        mine = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=dev)
        if dev.type == "cuda":
            flat = torch.empty(len(vals) * self.world, dtype=torch.int64, device=dev)   # one collective, one copy back
            self.dist.all_gather_into_tensor(flat, mine, group=self.group)
            v = flat.tolist()
            return [v[len(vals) * r:len(vals) * (r + 1)] for r in range(self.world)]
        allv = [torch.empty(len(vals), dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(allv, mine, group=self.group)
        return [[int(x) for x in t.tolist()] for t in allv]

    def exchange(self, plan, provide, accept):
        dist = self.dist
        ops, staged = [], []
        for src, dst, a, b in plan:
            if src == self.rank:
                t = provide(a, b)
                ops.append(dist.P2POp(dist.isend, t.cpu() if (self.gloo and t.is_cuda) else t, dst, self.group))
            if dst == self.rank:         # (src == dst only in the world-1 transport test: both ops, one group)
                t = accept(a, b)
                if self.gloo and t.is_cuda:
                    h = t.cpu()
                    staged.append((t, h))
                    t = h
                ops.append(dist.P2POp(dist.irecv, t, src, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, h in staged:
            t.copy_(h)


class LocalWorld:
    """k logical ranks as threads of one process (ranges of one resident buffer, one GPU)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.providers = [None] * world
        self._native = None
        self._lock = threading.Lock()

    def transport(self, rank):
        return LocalTransport(self, rank)

    def native_world(self):
        """The same k logical ranks for the library's own step (hip.ShardWorld: ffq_shard_world_*)."""
        with self._lock:
            if self._native is None:
                self._native = _hip.ShardWorld(self.world)
            return self._native

    def abort(self):
        self.barrier.abort()
        if self._native is not None:
            self._native.abort()


class LocalTransport:
    def __init__(self, lw, rank):
        self.lw, self.rank, self.world = lw, rank, lw.world

    def allgather(self, vals):
        lw = self.lw
        lw.slots[self.rank] = [int(v) for v in vals]
        lw.barrier.wait()
        out = [list(v) for v in lw.slots]
        lw.barrier.wait()
        return out

    def exchange(self, plan, provide, accept):
        lw = self.lw
        lw.providers[self.rank] = provide
        lw.barrier.wait()
        last = None
        for src, dst, a, b in plan:
            if dst == self.rank:
                last = accept(a, b)
                last.copy_(lw.providers[src](a, b))
        if last is not None and last.is_cuda:
            import torch
            torch.cuda.current_stream(last.device).synchronize()      # the sources must stay as they are until read
        lw.barrier.wait()


# ---- scan engine ------------------------------------------------------------------------------
class HipBackend:
    """Scan engine on the local MI355X through the C ABI."""

    def __init__(self, ctx, post_ctx=None):
        self.ctx = ctx
        self.post = post_ctx or ctx     # context (stream) the small table queries run on
        self._xstream = None
        self._comm = None               # stream of the overlapped hand-offs (comm_context)

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

    def cut(self, table, n_rows, lo, hi):
        """(i0, i1, pos0[i0], pos0[i1], pos5[i0 - 1], pos5[i1 - 1]) in one launch and one host wait."""
        return self.post.table_cut(table.data_ptr(), n_rows, lo, hi)

    def stream_context(self, ext):
        """torch work on `ext` (hand-off copies, RCCL send/recv) ordered on the scan's own HIP
        stream: the scan that follows needs no host synchronisation in between."""
        if not ext.is_cuda:
            return contextlib.nullcontext()
        import torch
        if self._xstream is None:
            self._xstream = torch.cuda.ExternalStream(self.ctx.stream(), device=ext.device)
        return torch.cuda.stream(self._xstream)

    @contextlib.contextmanager
    def comm_context(self, ext):
        """torch work on `ext` on a stream of its own, the scan stream made to wait for its end: a
        hand-off whose buffers no scan in flight touches (a step's own [tail | own | head] buffer)
        then runs beside the previous step's scan instead of behind it.  The caller vouches for that."""
        if not ext.is_cuda:
            yield
            return
        import torch
        if self._xstream is None:
            self._xstream = torch.cuda.ExternalStream(self.ctx.stream(), device=ext.device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=ext.device)
        with torch.cuda.stream(self._comm):
            yield
            done = torch.cuda.Event()
            done.record(self._comm)
        self._xstream.wait_event(done)


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
        self.total_records = self.n_own_records
        self.rounds = 0               # repair rounds of the step (0: the first scan of every rank stood)
        self.ext = None               # the [tail | own | head] buffer the rows refer to (grown: a new one)
        self.tail = self.head = 0
        self.comm = None              # native step: {handoff_ms, handoff_bytes, allgather_ms, rescan_rounds, regathers}


class _View:
    """A rank's [tail | own | head] buffer and its coordinates."""

    def __init__(self, ext, tail, head, lo, hi, total, origin=0):
        self.ext, self.tail, self.head, self.lo, self.hi, self.total = ext, tail, head, lo, hi, total
        self.origin = origin                        # offset of the stream's first byte
        self.start = lo - tail                      # stream offset of ext[0]
        self.end = hi + head
        self.sentinel = self.start == origin        # the iterator's b'\n' in front of the stream (:245)
        self.eof = self.end == total                # the view reaches the end of the stream
        self.add = self.start - (1 if self.sentinel else 0)
        self.n_bytes = tail + (hi - lo) + head


class _State:
    pass


_ERR_TEXT = {_hip.END_ERR_FINAL_QUAL: "Incomplete final quality string at byte",
             _hip.END_ERR_INCOMPLETE: "Incomplete entry at byte %i",
             _hip.END_ERR_INVALID: "Entry is invalid at byte %i"}


def raise_stream_error(end_state, byte):
    """The reference iterator's ValueErrors (fastqandfurious.py:262, :269, :272)."""
    text = _ERR_TEXT[end_state]
    raise ValueError(text % byte if "%" in text else text)


class ShardScanner:
    """Steps 1-5 above for one rank."""

    def __init__(self, backend, transport, bounds, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES):
        self.backend = backend
        self.tr = transport
        self.rank, self.world = transport.rank, transport.world
        self.bounds = list(bounds)
        assert len(self.bounds) == self.world + 1
        # (the byte in front of the range must be in view: a record that starts exactly at the
        # range's first byte is found through the "\n" before it)
        assert tail_bytes >= 1 and head_bytes >= 1
        self.tail_bytes, self.head_bytes = tail_bytes, head_bytes
        self.lo, self.hi, self.total = self.bounds[self.rank], self.bounds[self.rank + 1], self.bounds[-1]
        self.origin = self.bounds[0]        # offsets count from here on (readfastq_iter's `globaloffset`, :198-242)
        self._pending = None

    def halo(self):
        return halo_sizes(self.bounds, self.rank, self.tail_bytes, self.head_bytes)

    # ---- step 1 -------------------------------------------------------------------------------
    def _serve(self, plan, ext, tail, dst_ext=None, dst_start=None, overlap=False):
        """One collective exchange: this rank provides its own bytes out of `ext` and receives what
        the plan sends it into dst_ext (stream offset dst_start at index 0).  overlap: on the
        backend's hand-off stream instead of the scan stream (see HipBackend.comm_context)."""
        own_lo = self.lo
        if dst_ext is None:
            dst_ext, dst_start = ext, own_lo - tail

        def provide(a, b):
            return ext[tail + a - own_lo:tail + b - own_lo]

        def accept(a, b):
            return dst_ext[a - dst_start:b - dst_start]

        comm = getattr(self.backend, "comm_context", None) if overlap else None
        with (comm(ext) if comm is not None else self.backend.stream_context(ext)):
            self.tr.exchange(plan, provide, accept)

    def exchange_halo(self, ext, tail, head, overlap=False):
        """Fill ext[:tail] and ext[tail + n_own:tail + n_own + head] from the ranks that own those bytes.
        overlap=True: beside whatever the scan stream is doing -- only for a buffer no scan in flight
        reads (bench.py's pipelined steps give every lane its own)."""
        assert (tail, head) == self.halo()
        if self.world > 1:
            self._serve(halo_plan(self.bounds, self.tail_bytes, self.head_bytes), ext, tail, overlap=overlap)

    # ---- steps 2-3: one local scan and what it says about the two edges -------------------------
    def _local(self, v, table, flags, qual, qoff, start=None, first=None):
        """start: stream offset the first "\\n@" search starts at (None: the beginning of the view,
        i.e. a guess unless the view starts the stream)."""
        offset = 0 if start is None else max(start, v.start) - v.add
        if first is not None:
            rc, res = first                  # the offset-0 scan was submitted ahead (submit / finish)
        else:
            rc, res = self.backend.scan(v.ext, v.n_bytes, v.sentinel, offset, v.eof, v.add, table, flags, qual, qoff)
        st = _State()
        st.v, st.res, st.start = v, res, start
        n = st.n = int(res.n_records)
        if rc != _hip.OK:
            # the caller's table is too small: every rank learns it with the next all_gather and
            # raises (a rank that raised on its own would leave the others waiting in a collective)
            st.n = st.row_lo = st.row_hi = 0
            st.first = st.exit = UNKNOWN_POS
            st.exit_search = 0
            st.err, st.err_byte, st.want = ERR_TABLE_FULL, n, 0
            return st
        i0, i1, p_i0, p_i1, _q0, q1 = self.backend.cut(table, n, -(1 << 62) if v.lo == v.origin else v.lo,
                                                       (1 << 62) if v.hi == v.total else v.hi)
        good = _hip.END_OK if v.eof else _hip.END_REFILL
        # the entry the chain stops at (incomplete / invalid): a record start like the rows'
        p_inc = None
        if res.end_state != _hip.END_OK and res.last_status != _hip.POS_HEAD_BEG and res.last_pos[0] >= 0:
            p_inc = int(res.last_pos[0])
        st.row_lo, st.row_hi = i0, i1
        unknown = NONE_POS if (v.eof and res.end_state == _hip.END_OK) else UNKNOWN_POS

        def edge(idx, p_row, bound):
            if idx < n:
                return p_row
            if p_inc is not None and p_inc >= bound:
                return p_inc
            return unknown

        st.first = edge(i0, p_i0, v.lo)
        st.exit = edge(i1, p_i1, v.hi) if v.hi < v.total else NONE_POS
        # where the search that found the exit started (the iterator's `offset`, :254): the right
        # neighbour re-enters there if its own guess does not hold
        if i1 < n:
            st.exit_search = (q1 - 1) if i1 > 0 else offset + v.add
        else:
            st.exit_search = int(res.end_offset) + v.add
        st.err, st.err_byte, st.want = 0, 0, 0
        if res.end_state in (_hip.END_ERR_FINAL_QUAL, _hip.END_ERR_INCOMPLETE, _hip.END_ERR_INVALID):
            # a stream error: mine if the failing entry starts in my range (or nowhere: no entry at all)
            if p_inc is None or (v.lo <= p_inc < v.hi) or (v.hi == v.total and p_inc >= v.lo):
                st.err, st.err_byte = int(res.end_state), int(res.end_offset) + v.add
            elif p_inc < v.lo:
                st.first = st.exit = UNKNOWN_POS          # the guessed entry led nowhere
        elif res.end_state != good:
            raise RuntimeError("rank %d: end state %d of a scan with eof=%d" % (self.rank, res.end_state, v.eof))
        if not v.eof and not st.err and st.exit == UNKNOWN_POS and not (p_inc is not None and p_inc < v.lo):
            if res.end_state == _hip.END_REFILL:
                # the record that straddles my right edge does not end inside the look-ahead
                st.want = min(max(2 * v.head, self.head_bytes, 4096), v.total - v.hi)
        return st

    def _grown_view(self, v, new_head):
        ext = v.ext.new_empty(v.tail + (v.hi - v.lo) + new_head + 64)
        with self.backend.stream_context(v.ext):
            ext[:v.n_bytes] = v.ext[:v.n_bytes]
        return _View(ext, v.tail, new_head, v.lo, v.hi, v.total, v.origin)

    # ---- steps 4-5 ------------------------------------------------------------------------------
    def _settle(self, st, table, flags, qual, qoff):
        W, rank, B = self.world, self.rank, self.bounds
        rounds = 0
        while True:
            allv = self.tr.allgather([st.exit, st.first, st.row_hi - st.row_lo, st.want, st.v.head, st.err,
                                      st.err_byte, st.exit_search])
            ex, fi, cnt, want, head, err, errb, exs = (list(c) for c in zip(*allv))
            for r in range(W):
                if err[r] == ERR_TABLE_FULL:
                    raise RuntimeError("rank %d: offset table too small (%d records in its view)" % (r, errb[r]))
            grow = [r for r in range(W) if want[r] > 0]
            force = [r for r in range(1, W) if B[r] > B[0] and ex[r - 1] != UNKNOWN_POS and fi[r] != ex[r - 1]]
            if not grow and not force:
                for r in range(W):
                    if err[r]:
                        # The byte the iterator names is its `offset` when the failing search started:
                        # pos5 - 1 of the last COMPLETE record in front of the failing entry (:254, :275).
                        # A rank that owns no row in front of that entry does not know it -- its scan
                        # started at a guess, or at rows of the run-in that nothing has proven: the
                        # record in question straddles in from the left, and the nearest rank to the
                        # left that owns a row (or rank 0, whose start is exact) has it as the start of
                        # the search that found its exit.
                        byte = errb[r]
                        if r > 0 and cnt[r] == 0:
                            q = r - 1
                            while q > 0 and cnt[q] == 0:
                                q -= 1
                            byte = exs[q]
                        raise_stream_error(err[r], byte)         # every rank raises the same error
                    if ex[r] == UNKNOWN_POS:
                        raise RuntimeError("sharded scan: rank %d has no exit and nobody can move" % r)
                return st, cnt, rounds
            rounds += 1
            if rounds > 2 * W + 48:
                raise RuntimeError("sharded scan does not settle (%d rounds)" % rounds)
            start = st.start
            v = st.v
            if grow:
                plan = []
                for r in grow:
                    plan += range_plan(B, r, B[r + 1] + head[r], B[r + 1] + want[r])
                nv = self._grown_view(v, want[rank]) if rank in grow else None
                self._serve(plan, v.ext, v.tail, nv.ext if nv else None, nv.start if nv else None)
                if nv is not None:
                    v = nv
            if rank in force:
                prev = ex[rank - 1]
                if prev == NONE_POS or prev >= v.hi and v.hi < v.total:
                    # the chain passes over my whole range (or ends before it): I own nothing
                    res = st.res
                    st = _State()
                    st.v, st.res, st.start, st.n = v, res, exs[rank - 1], 0
                    st.row_lo = st.row_hi = 0
                    st.first = st.exit = prev
                    st.exit_search = exs[rank - 1]
                    st.err = st.err_byte = st.want = 0
                    continue
                start = exs[rank - 1]
            if rank in grow or rank in force:
                st = self._local(v, table, flags, qual, qoff, start=start)

    def submit(self, ext, tail, head, table, flags=0, qual=None, qoff=None):
        """Enqueue the first scan of a step and return; finish() completes the step.  With a second
        ShardScanner on a context that shares the stream, the next step is queued while this one
        is finished: the GPU does not idle during the host's part of a step."""
        v = _View(ext, tail, head, self.lo, self.hi, self.total, self.origin)
        self.backend.scan_submit(ext, v.n_bytes, v.sentinel, 0, v.eof, v.add, table, flags, qual, qoff)
        self._pending = (v, table, flags, qual, qoff)

    def finish(self):
        v, table, flags, qual, qoff = self._pending
        self._pending = None
        return self._complete(v, table, flags, qual, qoff, self.backend.scan_wait())

    def scan(self, ext, tail, head, table, flags=0, qual=None, qoff=None):
        """ext = [tail | own | head] bytes (1-D uint8 tensor), halo filled (exchange_halo).  Returns
        ScanOutput; table rows are absolute stream offsets."""
        return self._complete(_View(ext, tail, head, self.lo, self.hi, self.total, self.origin), table, flags, qual, qoff, None)

    def _complete(self, v, table, flags, qual, qoff, first):
        st = self._local(v, table, flags, qual, qoff, first=first)
        st, cnt, rounds = self._settle(st, table, flags, qual, qoff)
        out = ScanOutput(st.res, st.n, st.row_lo, st.row_hi, st.exit, st.first)
        out.record_base = sum(cnt[:self.rank])
        out.total_records = sum(cnt)
        out.rounds = rounds
        out.ext, out.tail, out.head = st.v.ext, st.v.tail, st.v.head
        return out


class _DevView:
    """A device buffer the library owns (a shard's grown view), for torch.as_tensor."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def native_unique_id(dist, device):
    """The communicator id of the library's own RCCL transport, drawn by rank 0 and sent round with the process group
    that is there (any backend)."""
    import torch
    cpu = dist.get_backend() == "gloo"
    t = torch.zeros(128, dtype=torch.uint8, device="cpu" if cpu else device)
    if dist.get_rank() == 0:
        t.copy_(torch.frombuffer(bytearray(_hip.shard_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, 0)
    return bytes(t.cpu().numpy().tobytes())


def native_output(shard, ext, rc, res):
    """ScanOutput of a step of the library's own (ffq_shard_step_wait): same fields as ShardScanner's, errors raised alike."""
    import torch
    if rc == _hip.E_TABLE_FULL:
        raise RuntimeError("offset table too small (%d records in a rank's view)" % int(res.scan.n_records))
    if res.err_state:
        raise_stream_error(int(res.err_state), int(res.err_byte))
    out = ScanOutput(res.scan, int(res.n_rows), int(res.row_lo), int(res.row_hi), int(res.exit_pos), int(res.first_pos))
    out.record_base, out.total_records, out.rounds = int(res.record_base), int(res.total_records), int(res.rounds)
    out.tail, out.head = int(res.tail), int(res.head)
    out.ext = ext if int(res.d_ext or 0) == ext.data_ptr() else torch.as_tensor(
        _DevView(res.d_ext, res.tail + (shard.bounds[shard.rank + 1] - shard.bounds[shard.rank]) + res.head), device=ext.device)
    out.comm = {"handoff_ms": float(res.handoff_ms), "handoff_bytes": int(res.handoff_bytes),
                "allgather_ms": float(res.allgather_ms), "rescan_rounds": int(res.rounds), "regathers": int(res.regathers)}
    return out


class NativeShardScanner:
    """ShardScanner's interface over the library's own step (ffq_shard_*: hand-off over RCCL -- or the in-process
    transport --, scan, cut and the gather of the hand-off words queued by ONE call, one read-back per step)."""

    def __init__(self, ctx, bounds, rank, world, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES, unique_id=None,
                 local_world=None, parent=None):
        self.ctx, self.bounds, self.rank, self.world = ctx, list(bounds), rank, world
        self.tail_bytes, self.head_bytes = tail_bytes, head_bytes
        if parent is not None:
            self.sh = parent.sh.lane(ctx)
        else:
            self.sh = _hip.Shard(ctx, bounds, rank, world, tail_bytes, head_bytes, unique_id=unique_id, local_world=local_world)
        self._pending = None

    def lane(self, ctx):
        return NativeShardScanner(ctx, self.bounds, self.rank, self.world, self.tail_bytes, self.head_bytes, parent=self)

    def halo(self):
        return self.sh.halo()

    def submit(self, ext, tail, head, table, flags=0, qual=None, qoff=None, overlap=False):
        assert (tail, head) == self.halo()
        self.sh.step_submit(ext.data_ptr(), table.data_ptr(), table.shape[0], flags=flags,
                            d_qual=qual.data_ptr() if qual is not None else None,
                            qual_cap=qual.numel() if qual is not None else 0,
                            d_qoff=qoff.data_ptr() if qoff is not None else None, overlap=overlap)
        self._pending = ext

    def finish(self):
        ext, self._pending = self._pending, None
        rc, res = self.sh.step_wait()
        return native_output(self, ext, rc, res)

    def scan(self, ext, tail, head, table, flags=0, qual=None, qoff=None):
        """The whole step (the halos are handed off by it: no exchange_halo in front)."""
        self.submit(ext, tail, head, table, flags, qual, qoff)
        return self.finish()

    def close(self):
        self.sh.close()


DENSE_TEMPLATE = b"@foo#2\nAATTGCCG\n+\n3425@!#!\n"      # /root/reference/tests.py:8-35, single-line variant: 27 bytes


class SyntheticShard:
    """bench.py's input: this rank's byte range of a synthetic stream, built in
    HBM.  The logical stream is world * n_per records of S-single / S-wrapped
    (SURVEY.md 8d); cut points are moved off the record boundaries so that a
    record straddles every edge."""

    def __init__(self, ctx, kind, bytes_per_gpu, rank, world, dev, edge_shift=144, transport=None, total_records=None,
                 native=None):
        """total_records (S-single only): the whole stream has exactly this many records, dealt out as
        evenly as they go (BASELINE configs[4]: 333 460 193 records = 107 374 182 146 B over 8 ranges);
        bytes_per_gpu is ignored then.
        native: the steps run behind the C ABI (ffq_shard_*: RCCL hand-offs, or the in-process transport for logical
        ranks) instead of through this module's protocol over torch.distributed.  None: yes where that is the product
        path -- world > 1 over the nccl backend."""
        import torch
        from . import synth
        self.ctx, self.kind, self.rank, self.world, self.dev = ctx, kind, rank, world, dev
        if transport is None:
            if world > 1:
                import torch.distributed as dist
                transport = DistTransport(dist, None, dev)
            else:
                transport = SoloTransport()
        self.transport = transport
        first_rec = None
        self.rec_bytes, self.rec_cols = synth.RECORD_BYTES, (0, 17, 18, 168, 171, 321)
        if kind == "dense":
            # the reference's own test template repeated (/root/reference/tests.py:8-35: '@foo#2', 8 bases, '+', 8
            # qualities): 27 bytes per record, 6.75 per line -- every index tile over its slot.  One range only.
            assert world == 1, "the dense workload is a single range"
            self.rec_bytes, self.rec_cols = len(DENSE_TEMPLATE), (0, 6, 7, 15, 18, 26)
            per = [bytes_per_gpu // self.rec_bytes]
            n_per, first_rec, starts = per[0], 0, None
            blk_bytes = [per[0] * self.rec_bytes]
        elif kind == "single":
            if total_records is not None:
                per = [total_records // world + (1 if r < total_records % world else 0) for r in range(world)]
            else:
                per = [bytes_per_gpu // synth.RECORD_BYTES] * world
            n_per = per[rank]
            first_rec = sum(per[:rank])
            blk_bytes = [n * synth.RECORD_BYTES for n in per]
            starts = None
        else:
            n_per = int(bytes_per_gpu // 379.3)
            sizes = synth.wrapped_sizes(rank * n_per, n_per + 1, seed=43)
            self.w_len, self.w_rep = synth.wrapped_fields(rank * n_per, n_per + 1, seed=43)
            starts = np.zeros(n_per + 2, dtype=np.int64)
            np.cumsum(sizes, out=starts[1:])
            blk_bytes = [v[0] for v in transport.allgather([int(starts[n_per])])]
        self.n_per = n_per
        B = [0]
        for b in blk_bytes:
            B.append(B[-1] + b)
        total = B[-1]
        S = [0] + [(B[r] + edge_shift) // 16 * 16 for r in range(1, world)] + [total]
        self.bounds = S
        self.own_lo, self.own_hi = S[rank], S[rank + 1]
        self.n_own_bytes = self.own_hi - self.own_lo
        self.tail, self.head = halo_sizes(S, rank)
        self.block_start = B[rank]

        # records [rank*n_per, (rank+1)*n_per (+1)) are generated record-aligned straight into the
        # [tail | own | head] buffer, placed so that the range's first byte lands at ext[tail] (a 100 GiB
        # range has no room for a second copy); what the generator leaves in the halos is wiped -- they
        # are filled by the hand-off
        n_gen = n_per + (1 if rank < world - 1 else 0)
        a = self.own_lo - self.block_start              # the range starts `a` bytes into its first generated record
        assert 0 <= a <= self.tail or (a == 0 and self.tail == 0)
        if kind in ("single", "dense"):
            gen_bytes = n_gen * self.rec_bytes
        else:
            gen_bytes = int(starts[n_gen])
        room = max(self.tail + self.n_own_bytes + self.head, self.tail - a + gen_bytes) + 64
        self.ext = torch.empty(room, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        if kind == "dense":
            tpl = torch.tensor(list(DENSE_TEMPLATE), dtype=torch.uint8, device=dev)
            self.ext[:gen_bytes] = tpl.repeat(n_gen)
            del tpl
        elif kind == "single":
            ctx.synth_single(self.ext.data_ptr() + self.tail - a, first_rec, n_gen, seed=42)
        else:
            dstart = torch.from_numpy(starts[:n_gen + 1].copy()).to(dev)
            torch.cuda.synchronize()
            ctx.synth_wrapped(self.ext.data_ptr() + self.tail - a, dstart.data_ptr(), rank * n_per, n_gen, seed=43)
            self.starts = starts
        self.ext[:self.tail].zero_()
        self.ext[self.tail + self.n_own_bytes:].zero_()
        torch.cuda.synchronize()
        self.ext_scanned_bytes = self.tail + self.n_own_bytes + self.head
        min_rec = self.rec_bytes if kind in ("single", "dense") else 120
        self.max_records = n_per + (self.tail + self.head) // min_rec + 64
        auto = native is None
        if auto:
            native = isinstance(transport, DistTransport) and not transport.gloo and world > 1 and os.environ.get("FFQ_SHARD_NATIVE", "1") != "0"
        self.native = bool(native)
        if self.native:
            try:
                if isinstance(transport, LocalTransport):
                    self.scanner = NativeShardScanner(ctx, S, rank, world, local_world=transport.lw.native_world())
                else:
                    uid = native_unique_id(transport.dist, dev) if isinstance(transport, DistTransport) else _hip.shard_unique_id()
                    self.scanner = NativeShardScanner(ctx, S, rank, world, unique_id=uid)
                made = 1
            except Exception as e:      # noqa: BLE001
                if not auto:
                    raise
                made = 0
                import warnings
                warnings.warn("rank %d: the library's own sharded step is not available (%s): this package's protocol over "
                              "torch.distributed instead" % (rank, e))
            if auto:
                # every rank takes the same path: the C step only if every rank has it
                flag = torch.tensor([made], dtype=torch.int32, device=dev)
                transport.dist.all_reduce(flag, op=transport.dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if made:
                        self.scanner.close()
                    self.native = False
        if not self.native:
            self.scanner = ShardScanner(HipBackend(ctx), transport, S)
        self._lanes = None

    def scan(self, table, flags=0, qual=None, qoff=None):
        # the hand-off runs on the scan's own stream: the scan that follows needs no host
        # synchronisation in between
        if not self.native:
            self.scanner.exchange_halo(self.ext, self.tail, self.head)
        return self.scanner.scan(self.ext, self.tail, self.head, table, flags, qual, qoff)

    # ---- pipelined steps: submit(i + 1) before finish(i) -------------------------------------
    def make_lanes(self, n=2):
        """n scanners on contexts that share the scan stream (own scratch each) + one context
        with its own stream for the small queries of finish(), so that they do not queue
        behind the next step's kernels."""
        from . import hip
        if self.native:
            lanes = [self.scanner]
            for _ in range(n - 1):
                c = hip.Context(share=self.ctx)
                c.reserve(self.ext.numel())
                lanes.append(self.scanner.lane(c))
            self._lane_ctx = [ln.ctx for ln in lanes]
            self._post = None
        else:
            post = hip.Context(self.ctx.device)
            lanes = [ShardScanner(HipBackend(self.ctx, post), self.transport, self.bounds)]
            for _ in range(n - 1):
                c = hip.Context(share=self.ctx)
                c.reserve(self.ext.numel())
                lanes.append(ShardScanner(HipBackend(c, post), self.transport, self.bounds))
            self._post = post
        self._lanes = lanes
        # With peers every lane gets a [tail | own | head] buffer of its own, as consecutive steps of
        # a real stream have: the hand-off of step i + 1 then writes no byte the scan of step i reads
        # and runs beside it, on the hand-off stream (FFQ_SHARD_OVERLAP=0: behind it, on the scan stream).
        self._overlap = self.world > 1 and os.environ.get("FFQ_SHARD_OVERLAP", "1") != "0"
        self._exts = [self.ext] + [self.ext.clone() if self._overlap else self.ext for _ in range(n - 1)]
        return lanes

    def submit(self, lane, table, flags=0, qual=None, qoff=None):
        ext = self._exts[lane]
        if self.native:
            self._lanes[lane].submit(ext, self.tail, self.head, table, flags, qual, qoff, overlap=self._overlap)
            return
        self._lanes[lane].exchange_halo(ext, self.tail, self.head, overlap=self._overlap)
        self._lanes[lane].submit(ext, self.tail, self.head, table, flags, qual, qoff)

    def finish(self, lane):
        return self._lanes[lane].finish()

    def host_sample(self, nbytes):
        """First whole records of this rank's range, on the host."""
        import torch
        skip = 0
        if self.rank > 0:
            skip = 322 - (self.own_lo - self.block_start)   # only used on rank 0 in practice
        n = min(nbytes, self.n_own_bytes - skip)
        if self.kind in ("single", "dense"):
            n = n // self.rec_bytes * self.rec_bytes
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
        if self.kind in ("single", "dense"):
            rb = self.rec_bytes
            k0 = -(-self.own_lo // rb)
            assert n == -(-self.own_hi // rb) - k0, "record count differs from the closed form"
            col = torch.tensor(list(self.rec_cols), dtype=torch.int64, device=rows.device)
            for c0 in range(0, n, 1 << 24):            # (in pieces: at 100 GiB the table is 16 GB)
                c1 = min(n, c0 + (1 << 24))
                k = torch.arange(k0 + c0, k0 + c1, dtype=torch.int64, device=rows.device) * rb
                assert bool((rows[c0:c1] == k[:, None] + col[None, :]).all()), "offset table differs from the closed form"
        else:
            st = torch.from_numpy(self.starts).to(rows.device) + self.block_start
            k0 = int(np.searchsorted(self.starts + self.block_start, self.own_lo, side="left"))
            k1 = int(np.searchsorted(self.starts + self.block_start, self.own_hi, side="left"))
            if self.rank == self.world - 1:
                k1 = self.n_per
            assert n == k1 - k0, "record count differs from the generator's"
            # every column from the generator's closed form: 17 header bytes, the read wrapped at 80
            # columns, '+' (+ 16 repeated header bytes for one record in four), the quality likewise
            s0 = st[k0:k1]
            ln = torch.from_numpy(self.w_len[k0:k1]).to(rows.device)
            rep = torch.from_numpy(self.w_rep[k0:k1]).to(rows.device)
            p3 = s0 + 18 + ln + (ln + 79) // 80 - 1
            p4 = p3 + 3 + rep
            want = torch.stack([s0, s0 + 17, s0 + 18, p3, p4, p4 + p3 - (s0 + 18)], dim=1)
            assert bool((rows == want).all()), "offset table differs from the generator's closed form"
            assert bool((rows[:, 5] == st[k0 + 1:k1 + 1] - 1).all()), "record ends differ"

    def verify_decode(self, table, out, qual, qoff):
        """The decode's output at full size, with torch ops only: the CSR offsets must be the
        running sum of pos5 - pos4 over ALL rows of the scan (segmented output, res.path 6: every
        record's bytes behind the previous record's, the last offset where the last record ends), and
        the decoded bytes of a spread of records must be the buffer's bytes [pos4, pos5) minus 33
        (int8 arithmetic)."""
        import torch
        n = int(out.n_rows)
        lens = table[:n, 5] - table[:n, 4]
        if out.res.path == 6:
            assert bool((qoff[1:n] >= qoff[:n - 1] + lens[:n - 1]).all()), "records' decoded bytes overlap or are out of order"
            assert n == 0 or int(qoff[n].item()) == int((qoff[n - 1] + lens[n - 1]).item())
        else:
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
        if self.kind in ("single", "dense"):
            ql = self.rec_cols[5] - self.rec_cols[4]
            assert bool((ln == ql).all())
            ar = torch.arange(ql, device=table.device)
            src = self.ext[(p4[:, None] + ar[None, :]).reshape(-1)].to(torch.int16) - 33
            got = qual[(q0[:, None] + ar[None, :]).reshape(-1)].to(torch.int16)
            assert bool((src == got).all()), "decoded qualities differ from the buffer's bytes - 33"
        else:
            for a, l, q in zip(p4[::8].tolist(), ln[::8].tolist(), q0[::8].tolist()):
                src = (self.ext[a:a + l].to(torch.int16) - 33).to(torch.int8)
                assert bool((src == qual[q:q + l]).all()), "decoded qualities differ from the buffer's bytes - 33"



# ---- a rank's byte range of a FILE -------------------------------------------------------------------------------
class FileShard:
    """Rank `rank` of `world`'s byte range of a FASTQ file, resident in HBM: what the reference's single reader does
    for the whole stream (/root/reference/src/fastqandfurious.py:30-36 read(), :241-245 the first fill and its
    sentinel, :274-279 the carry of an unfinished entry) happens once per rank for [S_r - tail, S_r+1 + head) --
    pread by the library's helper threads into pinned slots, over the link on two copy streams
    (ffq_shard_load_fd) -- and ONE native step (ffq_shard_step_*) scans it, cuts this rank's rows out and proves
    them against the neighbours' (one gather of eight words; no hand-off: the halos are the file's own bytes, a
    look-ahead that must grow is read from the file).  No torch: device memory comes from the context.

    comm: how the ranks find each other -- a hip.ShardWorld (k logical ranks as threads of one process), 128 bytes of
    communicator id (ffq_shard_unique_id, handed round by the caller), or None: a world of one needs nothing, a larger
    one takes torch.distributed's default process group (any backend) to hand the id round; the steps themselves
    are RCCL.  start / end: the part of the file that is the stream (offsets in every row are FILE offsets; the
    default cut points are shard_bounds' -- 16-byte aligned, even shares --, bounds= names others)."""

    def __init__(self, ctx, path, rank=0, world=1, comm=None, start=0, end=None, tail_bytes=TAIL_BYTES,
                 head_bytes=HEAD_BYTES, device=None, bounds=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._own_fd = not isinstance(path, int)
        self.fd = os.open(path, os.O_RDONLY) if self._own_fd else path
        self.path = path
        size = os.fstat(self.fd).st_size
        end = size if end is None else min(int(end), size)
        start = min(int(start), end)
        self.bounds = [start + b for b in shard_bounds(end - start, world)]
        if bounds is not None:               # (the caller's cut points: file offsets, world + 1 of them, not decreasing)
            self.bounds = [int(b) for b in bounds]
            assert len(self.bounds) == world + 1 and self.bounds[-1] <= size
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self._world_obj = None
        if isinstance(comm, _hip.ShardWorld):
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, local_world=comm)
        elif isinstance(comm, (bytes, bytearray)):
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, unique_id=bytes(comm))
        elif world == 1:
            self._world_obj = _hip.ShardWorld(1)
            self.sh = _hip.Shard(ctx, self.bounds, 0, 1, tail_bytes, head_bytes, local_world=self._world_obj)
        else:
            import torch.distributed as dist
            if not dist.is_initialized() or dist.get_world_size() != world:
                raise ValueError("FileShard: world %d needs comm= (a hip.ShardWorld, a communicator id) or an initialised "
                                 "torch.distributed group of that size" % world)
            import torch
            dev = device if device is not None else torch.device("cuda", ctx.device)
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, unique_id=native_unique_id(dist, dev))
        self.tail, self.head = self.sh.halo()
        self.n_view = self.tail + (self.hi - self.lo) + self.head
        self.d_ext = ctx.dev_alloc(self.n_view + 64)
        self.view_start = self.lo - self.tail                     # file offset of d_ext[0]
        self.d_table = self.d_qual = self.d_qoff = None
        self.table_cap = self.qual_cap = 0
        self.loaded = False
        self.out = None

    def load(self):
        """This rank's bytes from the file into HBM; returns the bytes loaded."""
        n = self.sh.load_fd(self.fd, self.d_ext)
        self.loaded = True
        return n

    def _alloc(self, rows, decode):
        c = self.ctx
        if rows > self.table_cap:
            for p in (self.d_table, self.d_qoff):
                if p:
                    c.dev_free(p)
            self.d_table = c.dev_alloc(rows * 48)
            self.d_qoff = c.dev_alloc((rows + 1) * 8) if decode else None
            self.table_cap = rows
        elif decode and not self.d_qoff:
            self.d_qoff = c.dev_alloc((self.table_cap + 1) * 8)
        if decode:
            need = max(self.n_view // 2 + 64, -(-(self.n_view + 16) // 16384) * _hip.SEG_STRIDE)
            if need > self.qual_cap:
                if self.d_qual:
                    c.dev_free(self.d_qual)
                self.d_qual = c.dev_alloc(need)
                self.qual_cap = need

    def scan(self, decode=False, flags=0, rows_hint=None):
        """The step (collective: every rank calls it).  Returns the step's ShardResult; this rank's records are rows
        [row_lo, row_hi) of the device table, `record_base` their global ordinal.  A table that turns out too small
        on ANY rank is grown on every rank and the step repeated.  Stream errors are raised on every rank alike."""
        if not self.loaded:
            self.load()
        if decode:
            flags |= _hip.F_DECODE_QUAL | _hip.F_SINGLE_PASS
        rows = int(rows_hint) if rows_hint else self.n_view // 160 + 1024
        while True:
            self._alloc(rows, decode)
            self.ctx.reserve(self.n_view + 64)
            self.sh.step_submit(self.d_ext, self.d_table, self.table_cap, flags=flags, d_qual=self.d_qual if decode else None,
                                qual_cap=self.qual_cap if decode else 0, d_qoff=self.d_qoff if decode else None)
            rc, res = self.sh.step_wait()
            if rc == _hip.E_TABLE_FULL:
                rows = max(2 * self.table_cap, int(res.scan.n_records) + 1024)
                continue
            break
        if res.err_state:
            raise_stream_error(int(res.err_state), int(res.err_byte))
        self.out = res
        self.decoded = bool(decode)
        return res

    # ---- this rank's rows (and decoded qualities) back on the host, a batch at a time -------------------------------
    def rows(self, i0=None, i1=None):
        """Rows [i0, i1) of THIS RANK's records (0 = its first) as int64[n][6], absolute file offsets."""
        res = self.out
        n_own = int(res.row_hi - res.row_lo)
        i0 = 0 if i0 is None else i0
        i1 = n_own if i1 is None else min(i1, n_own)
        out = np.empty((max(i1 - i0, 0), 6), dtype=np.int64)
        if out.size:
            self.ctx.d2h(out, self.d_table + (int(res.row_lo) + i0) * 48)
        return out

    def quals(self, i0, i1, rows):
        """(qual int8[], qoff int64[n + 1]) of this rank's records [i0, i1) (scan(decode=True)): record j's decoded
        bytes are qual[qoff[j] : qoff[j] + pos5 - pos4] (packed or segmented alike, include/ffq.h FFQ_F_SINGLE_PASS)."""
        res = self.out
        base = int(res.row_lo) + i0
        n = i1 - i0
        qoff = np.empty(n + 1, dtype=np.int64)
        self.ctx.d2h(qoff, self.d_qoff + base * 8)
        q0 = int(qoff[0])
        q1 = int(qoff[n - 1] + rows[n - 1, 5] - rows[n - 1, 4]) if n else q0
        qual = np.empty(max(q1 - q0, 0), dtype=np.int8)
        if qual.size:
            self.ctx.d2h(qual, self.d_qual + q0)
        qoff -= q0
        qoff[n] = q1 - q0
        return qual, qoff

    def close(self):
        if getattr(self, "sh", None) is not None:
            self.sh.close()
            self.sh = None
        c = self.ctx
        for name in ("d_ext", "d_table", "d_qual", "d_qoff"):
            p = getattr(self, name, None)
            if p:
                c.dev_free(p)
                setattr(self, name, None)
        if self._world_obj is not None:
            self._world_obj.close()
            self._world_obj = None
        if self._own_fd and self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
