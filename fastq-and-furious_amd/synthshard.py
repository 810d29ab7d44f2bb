"""bench.py's and the tests' synthetic input: one rank's byte range of a synthetic FASTQ stream, built in HBM
(S-single / S-wrapped of SURVEY.md 8d, the reference's test template repeated), with the closed forms its rows must
equal.  Not product code: the product's shards are sharded.NativeShardScanner / FileShard."""
import os

import numpy as np

from . import hip as _hip
from .sharded import (DistTransport, HostShardScanner, LocalTransport, NativeShardScanner, SoloTransport, halo_sizes,
                      abort_together, native_unique_id)

DENSE_TEMPLATE = b"@foo#2\nAATTGCCG\n+\n3425@!#!\n"      # /root/reference/tests.py:8-35, single-line variant: 27 bytes


class SyntheticShard:
    """bench.py's input: this rank's byte range of a synthetic stream, built in
    HBM.  The logical stream is world * n_per records of S-single / S-wrapped
    (SURVEY.md 8d); cut points are moved off the record boundaries so that a
    record straddles every edge."""

    def __init__(self, ctx, kind, bytes_per_gpu, rank, world, dev, edge_shift=144, transport=None, total_records=None,
                 native=None, solo_rccl=False, serial=None, ctl_group=None):
        """total_records (S-single only): the whole stream has exactly this many records, dealt out as
        evenly as they go (BASELINE configs[4]: 333 460 193 records = 107 374 182 146 B over 8 ranges);
        bytes_per_gpu is ignored then.
        native: the DEVICE step (ffq_shard_step_*: RCCL hand-offs between processes, the in-process transport for
        logical ranks and for a world of one, the hosted transport over gloo for a dry run of several processes on ONE
        GPU -- RCCL refuses two ranks per device) -- the product path, and the default; False: the HOST step
        (ffq_shard_host_step) over host copies of the range with this transport, the GPU scanning through ffq_scan_host
        (tests).  solo_rccl: a world of one on the library's RCCL transport (communicators of one rank) instead of the
        in-process one -- what the product's step costs with no peers.  serial: the serial step (ONE communicator, ONE
        stream) from the start; ctl_group: the torch.distributed group communicator ids travel over (a gloo side group still
        works when the GPUs' fabric does not)."""
        import torch
        from . import synth
        self.ctx, self.kind, self.rank, self.world, self.dev = ctx, kind, rank, world, dev
        if transport is None:
            if world > 1:
                import torch.distributed as dist
                transport = DistTransport(dist) if dist.get_backend() == "gloo" else dist
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
            blk_bytes = [v[0] for v in self._allgather1(transport, int(starts[n_per]), dev)]
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
        gloo = isinstance(transport, DistTransport)
        if native is None:
            native = True
        self.native = bool(native)
        self._own_world = None
        self._ctl_group, self._solo_rccl = ctl_group, bool(solo_rccl)
        self.recovered = None             # the watchdog's message, once the steps were taken again in serial mode
        if self.native:
            if isinstance(transport, LocalTransport):
                self.scanner = NativeShardScanner(ctx, S, rank, world, local_world=transport.lw.native_world(), serial=serial)
            elif isinstance(transport, SoloTransport) and solo_rccl:
                self.scanner = NativeShardScanner(ctx, S, 0, 1, unique_id=_hip.shard_unique_id(), serial=serial)
            elif isinstance(transport, SoloTransport):
                self._own_world = _hip.ShardWorld(1)
                self.scanner = NativeShardScanner(ctx, S, 0, 1, local_world=self._own_world, serial=serial)
            elif gloo:
                # several processes that cannot talk RCCL (a dry run on ONE GPU): the device step all the same, its hand-offs
                # staged through host memory around gloo, its words gathered by gloo (ffq_shard_create_hosted)
                self.scanner = NativeShardScanner(ctx, S, rank, world, hosted=transport, serial=serial)
            else:
                self.scanner = NativeShardScanner(ctx, S, rank, world, unique_id=native_unique_id(transport, dev, ctl_group), serial=serial)
        else:
            self.scanner = HostShardScanner(transport, S, ctx=ctx)
        self._lanes = None

    @staticmethod
    def _allgather1(transport, value, dev):
        if hasattr(transport, "allgather"):
            return transport.allgather([value])
        import torch                                       # (torch.distributed itself: the nccl group of a real run)
        vals = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(transport.get_world_size())]
        transport.all_gather(vals, torch.tensor([value], dtype=torch.int64, device=dev))
        return [[int(v.item())] for v in vals]

    def _host_step(self, sc, table):
        """The host step over a host copy of this rank's bytes (dry runs, tests); rows and view back on the device."""
        import torch
        n = self.tail + self.n_own_bytes + self.head
        h_ext = np.zeros(n + 64, dtype=np.uint8)
        h_ext[self.tail:self.tail + self.n_own_bytes] = self.ext[self.tail:self.tail + self.n_own_bytes].cpu().numpy()
        h_table = np.empty((table.shape[0], 6), dtype=np.int64)
        out = sc.scan(h_ext, self.tail, self.head, h_table)
        table[:out.n_rows] = torch.from_numpy(h_table[:out.n_rows]).to(table.device)
        out.ext = torch.from_numpy(out.ext).to(self.ext.device)
        return out

    def scan(self, table, flags=0, qual=None, qoff=None):
        if not self.native:
            assert not (flags & _hip.F_DECODE_QUAL), "the host step has no decode"
            return self._host_step(self.scanner, table)
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
        else:
            lanes = [self.scanner] * n             # (the host step is synchronous: one scanner serves every lane)
        self._lanes = lanes
        # With peers every lane gets a [tail | own | head] buffer of its own, as consecutive steps of
        # a real stream have: the hand-off of step i + 1 then writes no byte the scan of step i reads
        # and runs beside it, on the hand-off stream (FFQ_SHARD_OVERLAP=0: behind it, on the scan stream).
        self._overlap = self.native and self.world > 1 and os.environ.get("FFQ_SHARD_OVERLAP", "1") != "0"
        self._exts = [self.ext] + [self.ext.clone() if self._overlap else self.ext for _ in range(n - 1)]
        self._queued = [None] * n
        return lanes

    def recover_serial(self, err):
        """After hip.FFQTimeout (the step watchdog; a collective that does not come back does so on EVERY rank): stop what
        the abandoned steps left on the GPU (ncclCommAbort, streams drained), a NEW communicator -- ONE, for the serial
        step -- from an id handed round over the control group, the lanes again.  Only for the RCCL transport: the others
        have no communicator to replace."""
        assert self.native and self.scanner.sh.transport() == "rccl", "nothing to rebuild on this transport"
        self.recovered = str(err)
        lanes = self._lanes or [self.scanner]
        n_lanes = len(lanes) if self._lanes else 0
        if self._solo_rccl:
            drained = all([ln.abort() for ln in lanes])
        else:
            drained = abort_together(self.transport, self._ctl_group, lanes)      # (the ranks meet first: sharded.abort_together)
        if not drained:
            # ncclCommAbort itself is still busy (a peer's process is gone): it holds the device, every further HIP call of this
            # process would wait behind it -- nothing to rebuild on; the caller reports and leaves (bench.py: an "error" line, _exit)
            raise RuntimeError("the step did not come back (%s) and the communicators' abort is still busy: %s"
                               % (err, _hip.lib().ffq_last_error().decode("utf-8", "replace")))
        for ln in reversed(lanes):                 # (the lanes before the shard whose communicators they borrow)
            ln.close()
        for c in (getattr(self, "_lane_ctx", None) or [])[1:]:
            c.close()
        self._lanes = None
        if self._solo_rccl:
            uid = _hip.shard_unique_id()
        else:
            uid = native_unique_id(self.transport, self.dev, self._ctl_group)
        self.scanner = NativeShardScanner(self.ctx, self.bounds, self.rank if not self._solo_rccl else 0, self.world, unique_id=uid, serial=True)
        if n_lanes:
            self.make_lanes(n_lanes)

    def submit(self, lane, table, flags=0, qual=None, qoff=None):
        ext = self._exts[lane]
        if self.native:
            self._lanes[lane].submit(ext, self.tail, self.head, table, flags, qual, qoff, overlap=self._overlap)
            return
        self._queued[lane] = table                  # (the host step runs when it is finished)

    def finish(self, lane):
        if self.native:
            return self._lanes[lane].finish()
        table, self._queued[lane] = self._queued[lane], None
        return self._host_step(self._lanes[lane], table)

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
        if out.res.path in (6, 8):
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
