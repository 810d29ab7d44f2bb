"""Byte-range sharding of one FASTQ stream over the GPUs of a node.

The reference has nothing like this (it is a single-threaded generator); the
unit that shards is the record chain of readfastq_iter
(/root/reference/src/fastqandfurious.py:251-279): once a record start is known,
records are independent.

Rank r owns the bytes [S_r, S_{r+1}) of the stream and every record whose '@'
lies in that range.  One process per GPU; per step:

  1. halo hand-off: every rank receives the TAIL bytes in front of its range
     (run-in: a chain started anywhere in it has re-synchronised with the true
     record chain before the range starts) and the HEAD bytes behind it (to
     finish the record that straddles the edge), from whichever ranks own them
     (ncclSend / ncclRecv over xGMI) -- or reads them from the file itself;
  2. one ordinary scan of [tail | own | head] on the local GPU;
  3. the rows with S_r <= pos0 < S_{r+1} are the shard's records;
  4. verification, eight words per rank (one all-gather): the first record start
     at/after S_{r+1} as rank r sees it must equal the first record start rank
     r+1 sees in its range.  Together with rank 0's exact start this proves
     every shard's rows by induction; the counts give global ordinals.

What the reference does with a record that does not fit its buffer -- keep
`buf[offset:]` and read more until it does (:274-279) -- happens here per edge:
a rank whose look-ahead ends inside the record that straddles its right edge asks
for a larger one and scans again; a rank whose guessed entry the left neighbour's
chain contradicts is scanned again from the neighbour's exit, and everything is
verified again.  Errors of the stream (the iterator's three ValueErrors) are
raised by every rank together, and only once the failing rank's entry is proven.

THE PROTOCOL LIVES IN THE LIBRARY (csrc/ffq_shard_proto.h) and nowhere else.  This
module binds its two drivers:
  NativeShardScanner  the device step (ffq_shard_step_*): RCCL between processes, or
                      k logical ranks as threads of one process (hip.ShardWorld)
  HostShardScanner    the host step (ffq_shard_host_step): buffers in host memory, the
                      transport handed in -- torch.distributed over gloo (the CPU
                      test-suite, a dry run of several ranks on one GPU) or threads --,
                      the scan the GPU's (ffq_scan_host) unless a test brings its own
  FileShard           a rank's range of a FILE (ffq_shard_load_fd + the device step)
"""
import ctypes
import os
import threading

import numpy as np

from . import hip as _hip

TAIL_BYTES = 1 << 20      # run-in taken from the left
HEAD_BYTES = 1 << 20      # look-ahead taken from the right (grown when a record needs more)

NONE_POS = -1             # no record starts at / after the bound: the view reaches the end of the stream
UNKNOWN_POS = -2          # not known yet (more look-ahead needed, or the guessed entry led nowhere)


def shard_bounds(total_bytes, world):
    """S_0..S_world: 16-byte aligned cut points of the stream."""
    b = [(r * total_bytes // world) // 16 * 16 for r in range(world)]
    b.append(total_bytes)
    return b


def halo_sizes(bounds, rank, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES):
    """(tail, head) of rank's first scan (ffq_shard_halo's rule: the same on every rank, so that each knows what the
    others need without asking)."""
    lo, hi, total = bounds[rank], bounds[rank + 1], bounds[-1]
    return min(tail_bytes, lo - bounds[0]), min(head_bytes, total - hi)


# ---- transports of the host step: the same list of pieces (src, dst, a, b, ptr) on every rank -------------------------
def _bytes_at(ptr, n):
    import torch
    return torch.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=torch.uint8)


class SoloTransport:
    rank, world = 0, 1

    def allgather(self, vals):
        return [list(vals)]

    def exchange(self, pieces):
        assert not pieces


class DistTransport:
    """torch.distributed on host memory (gloo): the CPU test-suite's processes, bench.py's dry run of N ranks on one GPU."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather(self, vals):
        import torch
        mine = torch.tensor([int(v) for v in vals], dtype=torch.int64)
        allv = [torch.empty(len(vals), dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(allv, mine, group=self.group)
        return [[int(x) for x in t.tolist()] for t in allv]

    def exchange(self, pieces):
        dist, ops = self.dist, []
        for src, dst, a, b, ptr in pieces:
            if src == self.rank:
                ops.append(dist.P2POp(dist.isend, _bytes_at(ptr, b - a), dst, self.group))
            elif dst == self.rank:
                ops.append(dist.P2POp(dist.irecv, _bytes_at(ptr, b - a), src, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()


class LocalWorld:
    """k logical ranks as threads of one process."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self._native = None
        self._lock = threading.Lock()

    def transport(self, rank):
        return LocalTransport(self, rank)

    def native_world(self):
        """The same k logical ranks for the device step (hip.ShardWorld: ffq_shard_world_*)."""
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

    def exchange(self, pieces):
        lw = self.lw
        lw.slots[self.rank] = {(a, b, dst): ptr for src, dst, a, b, ptr in pieces if src == self.rank}
        lw.barrier.wait()
        for src, dst, a, b, ptr in pieces:
            if dst == self.rank:
                ctypes.memmove(ptr, lw.slots[src][(a, b, dst)], b - a)
        lw.barrier.wait()               # the sources must stay as they are until read


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
        self.comm = None              # {handoff_ms, handoff_bytes, allgather_ms, rescan_rounds, regathers}


_ERR_TEXT = {_hip.END_ERR_FINAL_QUAL: "Incomplete final quality string at byte",
             _hip.END_ERR_INCOMPLETE: "Incomplete entry at byte %i",
             _hip.END_ERR_INVALID: "Entry is invalid at byte %i"}


def raise_stream_error(end_state, byte):
    """The reference iterator's ValueErrors (fastqandfurious.py:262, :269, :272)."""
    text = _ERR_TEXT[end_state]
    raise ValueError(text % byte if "%" in text else text)


def _output(res, rc):
    if rc == _hip.E_TABLE_FULL:
        if int(res.scan.n_records) == 0 and int(res.scan.n_qual_bytes) > 0:
            raise RuntimeError("quality buffer too small (%d decoded bytes in a rank's view)" % int(res.scan.n_qual_bytes))
        raise RuntimeError("offset table too small (%d records in a rank's view)" % int(res.scan.n_records))
    if res.err_state:
        raise_stream_error(int(res.err_state), int(res.err_byte))
    out = ScanOutput(res.scan, int(res.n_rows), int(res.row_lo), int(res.row_hi), int(res.exit_pos), int(res.first_pos))
    out.record_base, out.total_records, out.rounds = int(res.record_base), int(res.total_records), int(res.rounds)
    out.tail, out.head = int(res.tail), int(res.head)
    out.comm = {"handoff_ms": float(res.handoff_ms), "handoff_bytes": int(res.handoff_bytes),
                "allgather_ms": float(res.allgather_ms), "rescan_rounds": int(res.rounds), "regathers": int(res.regathers),
                "nranks": int(res.nranks), "mode": "serial" if res.serial else "pipelined"}
    return out


class HostShardScanner:
    """One rank's step over HOST memory (ffq_shard_host_step): the library's protocol, this module's transport.
    scan: None -- the GPU scans (ffq_scan_host on `ctx`) --, or a test's own engine:
    scan(buf uint8[n], sentinel, offset, eof, add, table int64[cap][6]) -> (rc, n_records, end_state, end_offset,
    last_status, last_pos0); rows written to `table`."""

    def __init__(self, transport, bounds, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES, ctx=None, scan=None):
        self.tr, self.ctx, self._scan = transport, ctx, scan
        self.rank, self.world = transport.rank, transport.world
        self.bounds = list(bounds)
        assert len(self.bounds) == self.world + 1
        self.tail_bytes, self.head_bytes = tail_bytes, head_bytes
        self.transport_name = type(transport).__name__

    def halo(self):
        return halo_sizes(self.bounds, self.rank, self.tail_bytes, self.head_bytes)

    def _scan_cb(self, buf, n, sentinel, offset, eof, add, table, cap, res):
        data = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(buf)) if n else np.zeros(0, np.uint8)
        rows = np.ctypeslib.as_array((ctypes.c_int64 * (cap * 6)).from_address(table)).reshape(cap, 6) if cap else np.zeros((0, 6), np.int64)
        rc, nrec, end, off, status, pos0 = self._scan(data, bool(sentinel), offset, bool(eof), add, rows)
        res.n_records, res.end_state, res.end_offset, res.last_status = int(nrec), int(end), int(off), int(status)
        res.last_pos[0] = int(pos0)
        return rc

    def scan(self, ext, tail, head, table):
        """ext = [tail | own | head] bytes (C-contiguous uint8 numpy array, the middle filled), table int64[cap][6] (numpy).
        Returns ScanOutput; out.ext is the view the rows refer to (ext, or a larger copy when a look-ahead grew)."""
        assert (tail, head) == self.halo()
        assert ext.dtype == np.uint8 and ext.flags.c_contiguous and table.dtype == np.int64 and table.flags.c_contiguous
        rc, res = _hip.shard_host_step(self.rank, self.world, self.bounds, self.tail_bytes, self.head_bytes,
                                       ext.ctypes.data, table.ctypes.data, table.shape[0], self.tr.exchange, self.tr.allgather,
                                       scan=self._scan_cb if self._scan is not None else None, ctx=self.ctx)
        view = None
        if int(res.d_ext or 0) and int(res.d_ext) != ext.ctypes.data:
            n = int(res.tail) + self.bounds[self.rank + 1] - self.bounds[self.rank] + int(res.head)
            view = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(int(res.d_ext))).copy()
            _hip.lib().ffq_shard_host_free(ctypes.c_void_p(int(res.d_ext)))
        out = _output(res, rc)
        out.ext = ext if view is None else view
        return out

    # (the interface of the pipelined device step, for callers that treat both alike: the host step runs when it is finished)
    def submit(self, ext, tail, head, table):
        self._pending = (ext, tail, head, table)

    def finish(self):
        args, self._pending = self._pending, None
        return self.scan(*args)

    def close(self):
        pass


class _DevView:
    """A device buffer the library owns (a shard's grown view), for torch.as_tensor."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def native_unique_id(dist, device, group=None):
    """The communicator id of the library's own RCCL transport, drawn by rank 0 and sent round with the process group
    that is there (any backend; group: a side group, e.g. a gloo one that still works when the GPUs' fabric does not).
    One id serves ONE communicator set-up (ncclCommInitRank's bootstrap root stops listening once every rank has joined):
    draw a new one for every Shard."""
    import torch
    cpu = dist.get_backend(group) == "gloo"
    t = torch.zeros(128, dtype=torch.uint8, device="cpu" if cpu else device)
    if dist.get_rank(group) == 0:
        t.copy_(torch.frombuffer(bytearray(_hip.shard_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(t.cpu().numpy().tobytes())


def abort_together(dist, group, shards):
    """After hip.FFQTimeout: the ranks meet over the process group FIRST, then every one aborts its communicators.  A
    collective that does not come back does so on every rank, but not at the same instant -- each rank's deadline runs from
    its own stage -- and a rank whose ncclCommAbort starts after its peers have already torn THEIR ends down waits in RCCL's
    teardown as it does for a peer whose process is gone (seen at world 8, tests/multigpu_worker.py: seven ranks through in
    1.1 s, the one that tripped a second later still in ncclCommAbort after 30).  A peer that never reports (its process is
    gone) fails the meeting instead -- RuntimeError, nothing aborted, leave with _exit.  -> whether every shard drained."""
    try:
        dist.barrier(group=group)
    except Exception as e:          # (gloo: connection reset / timeout)
        raise RuntimeError("the step did not come back and a peer did not report its own tripped step over the process group: %s" % (e,))
    return all([sh.abort() for sh in shards])


def check_bounds(bounds, world, size):
    """A caller's cut points of a file: world + 1 offsets, not decreasing, inside [0, size].  (Any byte will do: a rank's
    view is read from the file into an aligned device buffer whatever its file offset -- tests/test_fileshard.py cuts inside
    headers and on '+' / '@' bytes.)  ValueError on the calling rank, before any collective is entered."""
    b = [int(x) for x in bounds]
    if len(b) != int(world) + 1:
        raise ValueError("bounds: %d cut points for a world of %d (world + 1 are needed)" % (len(b), world))
    if b[0] < 0 or b[-1] > int(size):
        raise ValueError("bounds [%d, %d] reach outside the file (%d bytes)" % (b[0], b[-1], size))
    for i in range(len(b) - 1):
        if b[i] > b[i + 1]:
            raise ValueError("bounds decrease: %d > %d (ranks %d, %d)" % (b[i], b[i + 1], i, i + 1))
    return b


def check_peers(info, expect_world=None):
    """What a host asserts before it trusts a multi-GPU number (Shard.info()): the communicators count the ranks the job
    was started with, and -- when the bus ids are known (RCCL gathers them) -- those ranks sit on DISTINCT GPUs."""
    world = info["world"] if expect_world is None else int(expect_world)
    if info["nranks_handoff"] != world or (not info["serial"] and info["nranks_gather"] not in (0, world)) or info["world"] != world:
        raise RuntimeError("the shard's communicators count %d / %d ranks, the job has %d" % (info["nranks_handoff"], info["nranks_gather"], world))
    known = [b for b in info["bus_ids"] if b is not None]
    if len(known) == world and len(set(known)) != world:
        raise RuntimeError("%d ranks share GPUs: bus ids %s" % (world, info["bus_ids"]))
    return True


def native_output(shard, ext, rc, res):
    """ScanOutput of a device step (ffq_shard_step_wait); errors raised as the host step raises them."""
    import torch
    out = _output(res, rc)
    out.ext = ext if int(res.d_ext or 0) == ext.data_ptr() else torch.as_tensor(
        _DevView(res.d_ext, res.tail + (shard.bounds[shard.rank + 1] - shard.bounds[shard.rank]) + res.head), device=ext.device)
    return out


class NativeShardScanner:
    """ShardScanner's interface over the library's own step (ffq_shard_*: hand-off over RCCL -- or the in-process
    transport --, scan, cut and the gather of the hand-off words queued by ONE call, one read-back per step)."""

    def __init__(self, ctx, bounds, rank, world, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES, unique_id=None,
                 local_world=None, parent=None, hosted=None, serial=None):
        """serial: True -- the serial step (ONE communicator, ONE stream: hand-off, scan, words, gather in order on the scan
        stream), the fallback after a watchdog trip; None: what FFQ_SHARD_SERIAL says (default: the pipelined step)."""
        self.ctx, self.bounds, self.rank, self.world = ctx, list(bounds), rank, world
        self.tail_bytes, self.head_bytes = tail_bytes, head_bytes
        if parent is not None:
            self.sh = parent.sh.lane(ctx)
        else:
            self.sh = _hip.Shard(ctx, bounds, rank, world, tail_bytes, head_bytes, unique_id=unique_id, local_world=local_world,
                                 hosted=hosted, serial=serial)
        self._pending = None

    def info(self):
        return self.sh.info()

    def abort(self):
        """After hip.FFQTimeout: stop what the step left on the GPU (ncclCommAbort, streams drained); close() follows."""
        self._pending = None
        return self.sh.abort()

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
    communicator id (ffq_shard_unique_id, handed round by the caller), a transport object with exchange / allgather
    (DistTransport over gloo: several processes that cannot talk RCCL, e.g. sharing one GPU), or None: a world of one needs nothing, a larger
    one takes torch.distributed's default process group (any backend) to hand the id round; the steps themselves
    are RCCL.  start / end: the part of the file that is the stream (offsets in every row are FILE offsets; the
    default cut points are shard_bounds' -- 16-byte aligned, even shares --, bounds= names others: world + 1 file offsets,
    not decreasing, inside the file -- ValueError otherwise, before anything collective).
    serial: the serial step from the start (sharded.NativeShardScanner); group: the torch.distributed group the communicator
    id travels over (default: the world's).  A step that does not come back raises hip.FFQTimeout on every rank (the
    watchdog, include/ffq.h); where the ranks found each other through torch.distributed, scan() takes the serial step ONCE
    on a new communicator before it gives up (self.recovered says so).

    slab_bytes: a range that does NOT fit the GPU (a 2 TB file over eight of them) goes through ONE device buffer of that many
    bytes, slab after slab (ffq_shard_scan_fd_slabs: the reference's own buf[offset:] + read-more loop inside the rank,
    /root/reference/src/fastqandfurious.py:274-279; the rank's edges are proven as ever) -- given, or FFQ_SHARD_SLAB_BYTES, or
    taken (1 GiB) when the device has no room for the range; every rank decides for itself.  Rows, ordinals and the iterator
    are the same; the decode is not part of a scan over slabs (scan(decode=True) raises): the range iterator decodes
    entryfunc_phred's qualities batch by batch on the device instead (quals_from_file)."""

    def __init__(self, ctx, path, rank=0, world=1, comm=None, start=0, end=None, tail_bytes=TAIL_BYTES,
                 head_bytes=HEAD_BYTES, device=None, bounds=None, qual_room=None, serial=None, group=None, slab_bytes=None):
        self._open(ctx, path, rank, world, qual_room)
        size = os.fstat(self.fd).st_size
        end = size if end is None else min(int(end), size)
        start = min(int(start), end)
        self.bounds = [start + b for b in shard_bounds(end - start, world)]
        if bounds is not None:
            self.bounds = check_bounds(bounds, world, size)
        self._make_shard(comm, tail_bytes, head_bytes, device, serial, group)
        env = os.environ.get("FFQ_SHARD_SLAB_BYTES")
        self._alloc_view(int(slab_bytes) if slab_bytes else (int(env) if env else 0))

    def _open(self, ctx, path, rank, world, qual_room):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError("FileShard: rank %d of %d" % (self.rank, self.world))
        # scan(decode=True): bytes of the quality buffer per 16 KiB tile of the view -- hip.SEG_STRIDE (reads of a few hundred
        # bases in one pass) unless given; hip.INPLACE_STRIDE lets four-line reads of any length decode in one pass as well
        self.qual_room = int(qual_room) if qual_room else _hip.SEG_STRIDE
        self._own_fd = not isinstance(path, int)
        self.fd = os.open(path, os.O_RDONLY) if self._own_fd else path
        self.path = path
        self._world_obj = None
        self._dist = None                       # (set: the ranks found each other through torch.distributed -- a watchdog trip can be recovered from)
        self.sh = None

    def _make_shard(self, comm, tail_bytes, head_bytes, device, serial, group):
        """self.bounds are there: the shard object of the step over them, on whatever the ranks find each other through."""
        ctx, rank, world = self.ctx, self.rank, self.world
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self._halo = (tail_bytes, head_bytes)
        if isinstance(comm, _hip.ShardWorld):
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, local_world=comm, serial=serial)
        elif isinstance(comm, (bytes, bytearray)):
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, unique_id=bytes(comm), serial=serial)
        elif comm is not None and hasattr(comm, "allgather") and hasattr(comm, "exchange"):
            # a transport of the host step's kind (DistTransport over gloo, ...): the device step over it (file-backed
            # shards hand off nothing: 64 bytes of words per rank and step go through it)
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, hosted=comm, serial=serial)
        elif world == 1:
            self._world_obj = _hip.ShardWorld(1)
            self.sh = _hip.Shard(ctx, self.bounds, 0, 1, tail_bytes, head_bytes, local_world=self._world_obj, serial=serial)
        else:
            import torch.distributed as dist
            if not dist.is_initialized() or dist.get_world_size(group) != world:
                raise ValueError("FileShard: world %d needs comm= (a hip.ShardWorld, a communicator id) or an initialised "
                                 "torch.distributed group of that size" % world)
            import torch
            dev = device if device is not None else torch.device("cuda", ctx.device)
            self._dist = (dist, dev, group)
            self.sh = _hip.Shard(ctx, self.bounds, rank, world, tail_bytes, head_bytes, unique_id=native_unique_id(dist, dev, group),
                                 serial=serial)
        self.tail, self.head = self.sh.halo()
        self.n_view = self.tail + (self.hi - self.lo) + self.head

    def _alloc_view(self, slab_bytes):
        ctx = self.ctx
        self.slab_bytes = slab_bytes
        self.d_ext = None
        if not self.slab_bytes:
            try:
                self.d_ext = ctx.dev_alloc(self.n_view + 64)
            except _hip.FFQError as e:
                if e.code != _hip.E_NOMEM:
                    raise
                self.slab_bytes = 1 << 30             # (no room for the range: it goes through a slab)
        self.view_start = self.lo - self.tail                     # file offset of d_ext[0]
        self.d_table = self.d_qual = self.d_qoff = None
        self.table_cap = self.qual_cap = 0
        self.loaded = False
        self.out = None

    def load(self):
        """This rank's bytes from the file into HBM; returns the bytes loaded (over slabs: 0 -- scan() reads them as it goes)."""
        if self.slab_bytes:
            self.loaded = True
            return 0
        n = self.sh.load_fd(self.fd, self.d_ext)
        self.loaded = True
        return n

    def _alloc(self, rows, decode, qneed=0):
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
            need = max(self.n_view // 2 + 64, -(-(self.n_view + 16) // 16384) * self.qual_room, qneed)
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
        if decode and self.slab_bytes:
            raise ValueError("FileShard: no decode over slabs (the qualities of a range that does not fit the GPU would not fit either)")
        if decode:
            flags |= _hip.F_DECODE_QUAL | _hip.F_SINGLE_PASS
        rows = int(rows_hint) if rows_hint else self.n_view // 160 + 1024
        qneed = 0
        while True:
            self._alloc(rows, decode, qneed)
            self.ctx.reserve(min(self.n_view, self.slab_bytes or self.n_view) + 64)
            try:
                if self.slab_bytes:
                    rc, res = self.sh.scan_fd_slabs(self.fd, self.slab_bytes, self.d_table, self.table_cap, flags=flags)
                else:
                    self.sh.step_submit(self.d_ext, self.d_table, self.table_cap, flags=flags, d_qual=self.d_qual if decode else None,
                                        qual_cap=self.qual_cap if decode else 0, d_qoff=self.d_qoff if decode else None)
                    rc, res = self.sh.step_wait()
            except _hip.FFQTimeout as e:
                if self._dist is None or self.recovered:
                    raise
                self._recover_serial(e)
                continue
            if rc == _hip.E_TABLE_FULL:
                if int(res.scan.n_records) == 0 and int(res.scan.n_qual_bytes) > 0:      # (some rank's quality buffer: a view that grew)
                    qneed = max(2 * self.qual_cap, int(res.scan.n_qual_bytes) + 4096)
                else:
                    rows = max(2 * self.table_cap, int(res.scan.n_records) + 1024)
                continue
            break
        if res.err_state:
            raise_stream_error(int(res.err_state), int(res.err_byte))
        self.out = res
        self.decoded = bool(decode)
        return res

    recovered = None          # the watchdog's message, once a step was taken again in serial mode

    def _recover_serial(self, err):
        """Every rank's step ran into the watchdog (a collective never returns on ONE rank only): stop what is left on the
        GPU, a NEW communicator -- one, for the serial step -- over the process group, the range loaded again."""
        dist, dev, group = self._dist
        self.recovered = str(err)
        if not abort_together(dist, group, [self.sh]):
            # (ncclCommAbort itself is still busy -- a peer's process is gone -- and holds the device: nothing to rebuild on)
            raise RuntimeError("the step did not come back (%s) and the communicator's abort is still busy: %s"
                               % (err, _hip.lib().ffq_last_error().decode("utf-8", "replace")))
        self.sh.close()
        tail_bytes, head_bytes = self._halo
        self.sh = _hip.Shard(self.ctx, self.bounds, self.rank, self.world, tail_bytes, head_bytes,
                             unique_id=native_unique_id(dist, dev, group), serial=True)
        self.load()

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

    # ---- the length filter of the reference's user guide, on the device, over THIS RANK's rows --------------------------
    def select(self, min_len=None, max_len=None):
        """Rows of this rank whose sequence length pos3 - pos2 lies in [min_len, max_len] (None: open), selected on the
        DEVICE (ffq_table_select_seqlen_idx): returns (k, index int64[k]) -- index[i] = the ordinal, among this rank's
        records, of kept row i; the kept rows themselves stay on the device until kept_rows() asks for a batch.  What
        /root/reference/doc/user-guide.rst:153-180 evaluates per record in an entryfunc, for the whole range at once: a
        dropped record never reaches the host."""
        res, c = self.out, self.ctx
        n_own = int(res.row_hi - res.row_lo)
        for name in ("d_sel", "d_idx"):
            p = getattr(self, name, None)
            if p:
                c.dev_free(p)
                setattr(self, name, None)
        self.n_kept = 0
        if n_own == 0:
            return 0, np.zeros(0, dtype=np.int64)
        self.d_sel, self.d_idx = c.dev_alloc(n_own * 48), c.dev_alloc(n_own * 8)
        lo = 0 if min_len is None else int(min_len)
        hi = (1 << 62) if max_len is None else int(max_len)
        k = c.table_select_seqlen_idx(self.d_table + int(res.row_lo) * 48, n_own, lo, hi, self.d_sel, self.d_idx)
        self.n_kept = k
        idx = np.empty(k, dtype=np.int64)
        if k:
            c.d2h(idx, self.d_idx)
        return k, idx

    def kept_rows(self, k0, k1):
        """Kept rows [k0, k1) of the last select() as int64[n][6], absolute file offsets."""
        out = np.empty((max(k1 - k0, 0), 6), dtype=np.int64)
        if out.size:
            self.ctx.d2h(out, self.d_sel + k0 * 48)
        return out

    def kept_column(self, k0, k1, column, rows):
        """(bytes uint8[], offsets int64[n + 1]) of component `column` ("header" as entryfunc cuts it, "sequence", "quality") of
        kept rows [k0, k1) -- gathered on the DEVICE from the resident range (ffq_table_gather_column); None when the range is
        not resident (slabs): the caller cuts the kept rows' slices out of the file."""
        if not self.d_ext or k1 <= k0 or (int(self.out.d_ext or 0) and int(self.out.d_ext) != self.d_ext):
            return None                          # (slabs; or a view that grew lives in the shard's own buffer: cut from the file instead)
        c, n = self.ctx, k1 - k0
        ca, sh, cb = c.COLUMNS[column]
        need = int((rows[:, cb] - rows[:, ca] - sh).clip(min=0).sum())
        d_col, d_off = c.dev_alloc(need + 64), c.dev_alloc((n + 1) * 8)
        try:
            sentinel = self.view_start == self.bounds[0]
            rc, nb = c.table_gather_column(self.d_ext, self.n_view, self.d_sel + k0 * 48, n, column, d_col, need + 64, d_off,
                                           sentinel=sentinel, add=self.view_start - (1 if sentinel else 0))
            if rc != _hip.OK:
                return None
            col, off = np.empty(nb, dtype=np.uint8), np.empty(n + 1, dtype=np.int64)
            if nb:
                c.d2h(col, d_col)
            c.d2h(off, d_off)
            return col, off
        finally:
            c.dev_free(d_col)
            c.dev_free(d_off)

    def quals_from_file(self, rows):
        """(qual int8[], qoff int64[n + 1]) of the records of `rows` (absolute file offsets, in order) when the range is NOT
        resident (slabs): the batch's bytes [pos0 of the first, pos5 of the last] are read into a scratch buffer on the device
        again and the Phred decode is the column gather with value_add = -33 there (ffq_table_gather_column; doc/user-guide.rst
        :206-214) -- the decode stays the GPU's, at the price of a second trip of those bytes over the link."""
        c, n = self.ctx, int(rows.shape[0])
        if n == 0:
            return np.zeros(0, np.int8), np.zeros(1, np.int64)
        a, b = int(rows[0, 0]), int(rows[-1, 5]) + 1
        need = int((rows[:, 5] - rows[:, 4]).sum())
        d_buf, d_rows = c.dev_alloc(b - a + 64), c.dev_alloc(n * 48)
        d_col, d_off = c.dev_alloc(need + 64), c.dev_alloc((n + 1) * 8)
        try:
            assert c.load_fd(self.fd, a, b - a, d_buf) == b - a
            c.h2d(d_rows, np.ascontiguousarray(rows))
            rc, nb = c.table_gather_column(d_buf, b - a, d_rows, n, "quality", d_col, need + 64, d_off, sentinel=False, add=a, value_add=-33)
            _hip.check(rc)
            col, off = np.empty(nb, dtype=np.int8), np.empty(n + 1, dtype=np.int64)
            if nb:
                c.d2h(col, d_col)
            c.d2h(off, d_off)
            return col, off
        finally:
            for p in (d_buf, d_rows, d_col, d_off):
                c.dev_free(p)

    def close(self):
        if getattr(self, "sh", None) is not None:
            self.sh.close()
            self.sh = None
        c = self.ctx
        for name in ("d_ext", "d_table", "d_qual", "d_qoff", "d_sel", "d_idx"):
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


def is_bgzf(path_or_fd):
    """Does the file begin with a BGZF member (bgzip's output: a gzip member whose extra field says how long it is)?"""
    fd = path_or_fd if isinstance(path_or_fd, int) else os.open(path_or_fd, os.O_RDONLY)
    try:
        h = os.pread(fd, 18, 0)
    finally:
        if not isinstance(path_or_fd, int):
            os.close(fd)
    return len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[12:14] == b"BC" and h[14:16] == b"\x02\x00"


def is_gzip(path_or_fd):
    """Does the file begin with a gzip member (of any kind)?"""
    fd = path_or_fd if isinstance(path_or_fd, int) else os.open(path_or_fd, os.O_RDONLY)
    try:
        h = os.pread(fd, 3, 0)
    finally:
        if not isinstance(path_or_fd, int):
            os.close(fd)
    return h == b"\x1f\x8b\x08"


class _Shifted:
    """bytes [base, base + len) of a stream held in memory, sliced with STREAM offsets (what a map of the file is for a
    plain one): view[a:b] -> a memoryview, or, as_bytes, a bytes object."""

    def __init__(self, arr, base, as_bytes=False):
        self._mv, self._base, self._as_bytes = memoryview(arr), int(base), as_bytes

    def __getitem__(self, sl):
        v = self._mv[sl.start - self._base:sl.stop - self._base]
        return v.tobytes() if self._as_bytes else v

    def release(self):
        pass

    close = release


class BgzfFileShard(FileShard):
    """Rank `rank` of `world`'s share of a BGZF-compressed FASTQ file (bgzip: gzip members of at most 64 KiB that say how
    long they are -- the one compressed format that can be read by ranges): the members whose first byte lies in the rank's
    share of the COMPRESSED file are its own; their trailers say how many bytes they hold, one exchange of those numbers gives
    every rank the cut points of the UNCOMPRESSED stream, the rank inflates its members side by side on the host
    (ffq_bgzf_range), the bytes go to its GPU, and the ordinary sharded step runs over them -- the halos handed over between
    the ranks (nothing but a rank's own members is ever inflated).  Rows, ordinals and entries are those of the uncompressed
    stream, as the reference's loop over gzip.open(...) yields them (/root/reference/src/fastqandfurious.py:241-279,
    :290-334 automagic_open); the ranks' rows concatenated are the scan of the whole inflated file.

    comm: as FileShard.  exchange: how the ranks tell each other their sizes BEFORE the shard exists -- a callable
    (list of ints) -> list over ranks of those lists; default: comm.allgather where comm is a transport object, else
    torch.distributed (all_gather_object over `group`); a world of one needs none.  threads: inflating threads (0: the
    library's default share of the host's cores).  No slabs: the inflated range must fit the GPU."""

    def __init__(self, ctx, path, rank=0, world=1, comm=None, exchange=None, tail_bytes=TAIL_BYTES, head_bytes=HEAD_BYTES,
                 device=None, qual_room=None, serial=None, group=None, threads=0):
        self._open(ctx, path, rank, world, qual_room)
        try:
            self.threads = int(threads)
            size = os.fstat(self.fd).st_size
            self.c_lo, self.c_hi = size * self.rank // self.world, size * (self.rank + 1) // self.world
            c_first, c_end, n_bytes, n_members = _hip.bgzf_range(self.fd, self.c_lo, self.c_hi)      # (nothing inflated yet)
            mine = [c_first, c_end, n_bytes, n_members]
            if self.world == 1:
                every = [mine]
            elif exchange is not None:
                every = exchange(mine)
            elif comm is not None and hasattr(comm, "allgather") and hasattr(comm, "exchange"):
                every = comm.allgather(mine)
            else:
                import torch.distributed as dist
                if not dist.is_initialized() or dist.get_world_size(group) != self.world:
                    raise ValueError("BgzfFileShard: world %d needs exchange= or an initialised torch.distributed group of that size"
                                     % self.world)
                every = [None] * self.world
                dist.all_gather_object(every, mine, group=group)
            every = [[int(x) for x in e] for e in every]
            for r in range(1, self.world):
                if every[r - 1][1] != every[r][0]:
                    raise _hip.FFQGzipError(_hip.E_ARG, "BGZF: rank %d's members end at byte %d of %r, rank %d found its first one at %d"
                                            % (r - 1, every[r - 1][1], path, r, every[r][0]))
            if every[-1][1] != size:
                raise _hip.FFQGzipError(_hip.E_ARG, "BGZF: the members of %r end at byte %d of %d" % (path, every[-1][1], size))
            self.members = [e[3] for e in every]
            self.bounds = [0]
            for e in every:
                self.bounds.append(self.bounds[-1] + e[2])
            self._make_shard(comm, tail_bytes, head_bytes, device, serial, group)
            self._alloc_view(0)
            if self.slab_bytes:
                raise _hip.FFQError(_hip.E_NOMEM, "BgzfFileShard: the inflated range (%d bytes) does not fit the GPU" % (self.hi - self.lo))
            self.h_own = None
            self._h_view = None
        except BaseException:
            self.close()
            raise

    def load(self):
        """This rank's members inflated (host threads) and its bytes in HBM; returns the bytes loaded.  The halos either side
        come with the step, from the neighbours."""
        own = self.hi - self.lo
        self.h_own = np.empty(max(own, 1), dtype=np.uint8)[:own]
        if own:
            got = _hip.bgzf_range(self.fd, self.c_lo, self.c_hi, out=self.h_own, threads=self.threads)
            assert got[2] == own, (got, own)
            self.ctx.h2d(self.d_ext + self.tail, self.h_own)
        self._h_view = None
        self.loaded = True
        return own

    def scan(self, decode=False, flags=0, rows_hint=None):
        res = super().scan(decode=decode, flags=flags, rows_hint=rows_hint)
        self._h_view = None
        return res

    def host_bytes(self):
        """(base, uint8[]) -- the bytes of this rank's records in host memory: array[x - base] is byte x of the uncompressed
        stream, from the rank's first byte to the end of its last record (which may lie in the next rank's range: those bytes
        came over with the look-ahead and are fetched from the device)."""
        if self._h_view is None:
            res = self.out
            n_own = int(res.row_hi - res.row_lo)
            end = self.hi
            if n_own:
                end = max(end, int(self.rows(n_own - 1, n_own)[0, 5]) + 1)
            extra = end - self.hi
            if extra > 0:
                d_ext = int(res.d_ext or 0) or self.d_ext
                tail = int(res.tail) if int(res.d_ext or 0) else self.tail
                more = np.empty(extra, dtype=np.uint8)
                self.ctx.d2h(more, d_ext + tail + (self.hi - self.lo))
                arr = np.concatenate([self.h_own, more])
            else:
                arr = self.h_own
            self._h_view = (self.lo, arr)
        return self._h_view

    def quals_from_file(self, rows):
        raise ValueError("BgzfFileShard: no slabs")


def __getattr__(name):
    # (bench.py's and the tests' synthetic input lives in its own module; old spellings keep working)
    if name in ("SyntheticShard", "DENSE_TEMPLATE"):
        from . import synthshard
        return getattr(synthshard, name)
    raise AttributeError(name)
