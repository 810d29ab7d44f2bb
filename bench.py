#!/usr/bin/env python
"""bench.py -- FASTQ buffer-scan hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path (line index -> record chain -> int64[n][6]
offset table, optionally Phred decode) over one batch of synthetic FASTQ that is
already resident in HBM.  Prints ONE JSON line (rank 0).

Workloads (BASELINE.json configs):
    single-1g    1 GiB S-single, 150 bp, Phred+33            (configs[1], default)
    decode-10g   10 GiB S-single + quality -> int8 decode     (configs[2]; single pass, segmented output)
    decode-10g-packed  DIAGNOSTIC: the same with the packed CSR stream of rounds 1-2 (two passes over the input; streams and
                 the decoding iterator have used the single pass since round 4)
    wrapped-10g  10 GiB S-wrapped, 50-300 bp, 80-col wrap     (configs[3])
    dense-1g     1 GiB of the reference's test template (27-byte records: every index tile dense)
    single-100g  ONE 100 GiB S-single stream cut into N byte ranges (configs[4]; strong scaling:
                 12.5 GiB per GPU at N = 8, all of it on one GPU at N = 1)
For N > 1 the same per-GPU workload is one byte range of a single logical
stream N times as long (weak scaling); ranks exchange only the bytes around
their range edges (RCCL send/recv) and verify the hand-off.  `--gpus N` without a
launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).  The default
run reports single-100g (and at N = 1 decode-10g, wrapped-10g) under `other_workloads`.
At N = 1 it also carries `shapes`: read shapes that are no BASELINE config (long four-line reads with the decode in one
pass, wrapped reads of kilobases, FASTA), 4 GiB each, every row verified against the generator's own offsets (shape_rates).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = 1 << 30
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
WORKLOADS = {
    "single-1g": dict(kind="single", bytes=1 * GIB, decode=False),
    # the decode through the single pass: the caller accepts the qualities SEGMENTED (FFQ_F_SINGLE_PASS: record i =
    # qual[qoff[i] : qoff[i] + pos5 - pos4], gaps between records) and the index kernel writes them itself
    "decode-10g": dict(kind="single", bytes=10 * GIB, decode=True, single_pass=True),
    # ... and as rounds 1-2 measured it: packed CSR stream, two passes over the input
    "decode-10g-packed": dict(kind="single", bytes=10 * GIB, decode=True, diagnostic=True),
    "wrapped-10g": dict(kind="wrapped", bytes=10 * GIB, decode=False),
    # ... with the decode (no BASELINE config: configs[2] decodes single-line reads, configs[3] wraps without decoding): one
    # pass -- the index kernel writes EVERY byte decoded in place (FFQ_F_SINGLE_PASS with FFQ_INPLACE_STRIDE per tile,
    # res.path 8) -- and, as a diagnostic, the two passes (packed stream)
    "wrapped-10g-decode": dict(kind="wrapped", bytes=10 * GIB, decode=True, single_pass=True, in_place=True),
    "wrapped-10g-decode-packed": dict(kind="wrapped", bytes=10 * GIB, decode=True, diagnostic=True),
    "wrapped-64m-decode": dict(kind="wrapped", bytes=64 << 20, decode=True, single_pass=True, in_place=True),
    # the reference's own test template repeated (/root/reference/tests.py:8-35: 27-byte records, 6.75 bytes per line):
    # every index tile is DENSE (over its slot); 75 B of SURVEY 8(d) traffic per record, 48 of them the row
    "dense-1g": dict(kind="dense", bytes=1 * GIB, decode=False),
    "dense-64m": dict(kind="dense", bytes=64 << 20, decode=False),
    # small variants for quick checks
    "single-64m": dict(kind="single", bytes=64 << 20, decode=False),
    "wrapped-64m": dict(kind="wrapped", bytes=64 << 20, decode=False),
    "decode-64m": dict(kind="single", bytes=64 << 20, decode=True),
    "single-10g": dict(kind="single", bytes=10 * GIB, decode=False),
    # BASELINE configs[4]: ONE 100 GiB stream (107 374 182 146 B, 333 460 193 records) cut into `world` byte
    # ranges -- 12.5 GiB per GPU at N = 8, the whole stream on one GPU at N = 1 (it fits): total work is
    # fixed, so this entry is the strong-scaling figure (`value`, the metric's weak-scaling line, stays configs[1]'s)
    "single-100g": dict(kind="single", bytes=100 * GIB, decode=False, split=True),
    "single-4g-split": dict(kind="single", bytes=4 * GIB, decode=False, split=True),      # (quick check of the same path)
}


def cpu_baseline(sample_u8, budget_s=10.0):
    """The oracle's whole-buffer chain (C restatement of the reference scanner),
    one host core, timed on a bounded sample of the same bytes."""
    from oracle import ffq_oracle
    ffq_oracle.lib()
    n = int(sample_u8.size)
    cap = n // 64 + 16
    table = np.empty((cap, 6), dtype=np.int64)
    out = np.zeros(4, dtype=np.int64)
    L = ffq_oracle.lib()
    reps, t0 = 0, time.perf_counter()
    while True:
        L.ffq_oracle_scan(sample_u8.ctypes.data, n, 1, 0, 1, 0, -1, table.ctypes.data, cap, out.ctypes.data)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 100000:
            break
    recs = int(out[0])
    return {
        "value": round(n * reps / el / 1e9, 4),
        "unit": "GB/s",
        "m_reads_per_s": round(recs * reps / el / 1e6, 4),
        "cores": 1,
        "kind": "port",
        "sample": "%d passes of oracle/ffq_oracle_scan (C restatement of _fastqandfurious.c entrypos + "
                  "the readfastq_iter chain) over the first %d bytes (%d records) of the workload, %.1f s"
                  % (reps, n, recs, el),
    }


def cpu_baseline_all_cores(sample_u8, budget_s=5.0):
    """Same oracle scan, one independent pass loop per host core (threads; ctypes drops the GIL)."""
    import threading
    from oracle import ffq_oracle
    L = ffq_oracle.lib()
    n = int(sample_u8.size)
    cores = os.cpu_count() or 1
    cap = n // 64 + 16
    counts = [0] * cores
    t_end = time.perf_counter() + budget_s

    def work(i):
        table = np.empty((cap, 6), dtype=np.int64)
        out = np.zeros(4, dtype=np.int64)
        while time.perf_counter() < t_end:
            L.ffq_oracle_scan(sample_u8.ctypes.data, n, 1, 0, 1, 0, -1, table.ctypes.data, cap, out.ctypes.data)
            counts[i] += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    return {"value": round(n * sum(counts) / el / 1e9, 4), "unit": "GB/s", "cores": cores,
            "sample": "%d passes over %d bytes on %d threads, %.1f s" % (sum(counts), n, cores, el)}


def cpu_reference_c(sample_bytes, budget_s=3.0):
    """The reference's own C scanner (oracle/_ref/_fastqandfurious.so, compiled from the reference's
    source by oracle/Makefile; the binary travels with the repo), called once per record from a
    Python loop as the reference's iterator does (fastqandfurious.py:252-254), one core.  None if
    the binary is not there."""
    from array import array
    from oracle import refload
    if not refload.have_reference_ext():
        return None
    ext = refload.load_ext()
    buf = b"\n" + sample_bytes
    pos = array("q", [-1] * 6)
    n, offset = 0, 0
    t0 = time.perf_counter()
    while True:
        if ext.entrypos(buf, offset, pos) != 6:
            break
        offset = pos[5] - 1
        n += 1
        if (n & 4095) == 0 and time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": round(offset / el / 1e9, 5), "unit": "GB/s", "m_reads_per_s": round(n / el / 1e6, 4), "cores": 1,
            "kind": "reference",
            "sample": "%d entrypos() calls of the reference C extension (bare scanner calls in a Python loop), "
                      "%.1f s" % (n, el)}


def cpu_reference_iter(sample_bytes, budget_s=3.0, fbufsize=50000):
    """The reference's C scanner driven the way the reference drives it: one entrypos() + one entryfunc()
    per record inside the refill loop (this package's readfastq_iter, the mirror of
    /root/reference/src/fastqandfurious.py:251-279, takes that per-record path for any scanner
    without a batched protocol), fbufsize = the reference benchmark's default 50 000 (benchmark.py:415)."""
    import io
    from oracle import refload
    from fastqandfurious_amd import fastqandfurious as F
    if not refload.have_reference_ext():
        return None
    ext = refload.load_ext()
    n, nbytes = 0, 0
    t0 = time.perf_counter()
    for h, sq, q in F.readfastq_iter(io.BytesIO(sample_bytes), fbufsize, F.entryfunc, ext.entrypos):
        n += 1
        nbytes += len(h) + len(sq) + len(q) + 6
        if (n & 4095) == 0 and time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": round(nbytes / el / 1e9, 5), "unit": "GB/s", "m_reads_per_s": round(n / el / 1e6, 4), "cores": 1,
            "kind": "reference", "fbufsize": fbufsize,
            "sample": "%d (header, sequence, quality) tuples: the reference C extension's entrypos + entryfunc per record "
                      "inside the readfastq_iter refill loop, %.1f s" % (n, el)}


def cpu_iterator_rate(sample_bytes, budget_s=3.0):
    """The reference-shaped number: the package's readfastq_iter with its pure-Python entrypos
    (mirror of src/fastqandfurious.py:39-100, 198-279), one record per Python call, one core."""
    import io
    from fastqandfurious_amd import fastqandfurious as F
    n = 0
    t0 = time.perf_counter()
    last = 0
    for e in F.readfastq_iter(io.BytesIO(sample_bytes), 1 << 20, F.entryfunc_abspos, F.entrypos):
        n += 1
        last = e[5]
        if (n & 1023) == 0 and time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": round(last / el / 1e9, 5), "unit": "GB/s", "m_reads_per_s": round(n / el / 1e6, 4), "cores": 1,
            "sample": "%d records through readfastq_iter(entryfunc_abspos, Python entrypos), %.1f s" % (n, el)}


def host_inclusive(ctx, sample_u8, flags):
    """Host buffer in, host table out (ffq_scan_host): pinned staging + H2D + kernels + D2H.
    PCIe-bound; reported beside `value`, never as `value`."""
    ctx.scan_host(sample_u8, flags=flags)
    reps, t0 = 3, time.perf_counter()
    for _ in range(reps):
        out = ctx.scan_host(sample_u8, flags=flags)
    el = (time.perf_counter() - t0) / reps
    return {"value": round(sample_u8.size / el / 1e9, 3), "unit": "GB/s",
            "m_reads_per_s": round(int(out[1].n_records) / el / 1e6, 3),
            "sample": "ffq_scan_host over the first %d bytes, pageable host memory in and out" % sample_u8.size}


def stream_inclusive(ctx, sample_u8, fbufsize=1 << 24):
    """File in, offset table out through the native stream front end (ffq_stream_*): chunked
    reads into pinned memory overlapped with H2D + scan + D2H.  The file sits in /dev/shm (page
    cache speed, no disk); like `host_inclusive` this is never `value`."""
    import tempfile
    from fastqandfurious_amd import hip
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(d, "ffq_bench_%d.fq" % os.getpid())
    with open(path, "wb") as fh:
        fh.write(sample_u8.tobytes())
    try:
        best, recs = None, 0
        for _ in range(3):
            fd = os.open(path, os.O_RDONLY)
            t0 = time.perf_counter()
            st = hip.FileStream(ctx, fd, fbufsize)
            recs = sum(rows.shape[0] for rows, _f, _o, _e, _x in st)
            st.close()
            os.close(fd)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        # the reference-shaped consumer on top of it: one Python tuple of three bytes objects per
        # record (readfastq_iter + entryfunc with the GPU scanner), bounded to ~3 s
        from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
        t0 = time.perf_counter()
        n_it, last = 0, 0
        with open(path, "rb") as fh:
            for h, sq, q in F.readfastq_iter(fh, 1 << 24, F.entryfunc, C.entrypos):
                n_it += 1
                if (n_it & 0xFFFF) == 0 and time.perf_counter() - t0 > 3.0:
                    break
        el_it = time.perf_counter() - t0
    finally:
        os.unlink(path)
    return {"value": round(sample_u8.size / best / 1e9, 3), "unit": "GB/s",
            "m_reads_per_s": round(recs / best / 1e6, 3),
            "sample": "ffq_stream over a %d-byte file in %s, %d MiB chunks, best of 3" % (sample_u8.size, d, fbufsize >> 20),
            "iterator_m_reads_per_s": round(n_it / el_it / 1e6, 3),
            "iterator_sample": "%d (header, sequence, quality) tuples from readfastq_iter(entryfunc, GPU "
                               "scanner) over the same file, %.1f s" % (n_it, el_it)}


def sharded_file(ctx, shard, rank, world, dev, dist, per_rank_bytes=1 << 30, budget_s=3.0):
    """File in, this rank's rows out through the file-backed byte-range shards (sharded.FileShard: ffq_shard_load_fd +
    one ffq_shard_step): every rank writes the first bytes of its range of the synthetic stream into ONE file in
    /dev/shm, then every rank loads ITS range of that file (pread -> pinned slots -> two copy streams) and the ranks
    prove their rows against each other with one gather of eight words -- no byte passes between GPUs.  Load-bound
    (page cache -> PCIe); like the other host-inclusive figures never `value`.  Collective."""
    import tempfile
    import torch
    from fastqandfurious_amd import fastqandfurious as F, hip, sharded
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    tag = os.environ.get("MASTER_PORT", str(os.getpid())) if world > 1 else str(os.getpid())
    path = os.path.join(d, "ffq_bench_shards_%s.fq" % tag)
    # rank r's piece: the first whole records of its range of the logical stream, up to per_rank_bytes; the pieces go
    # into the file back to back (a well-formed FASTQ file whose even cut points fall inside records, as a real file's do)
    blk = shard.own_lo - shard.block_start
    if shard.kind in ("single", "dense"):
        skip = (-blk) % shard.rec_bytes
        n = (min(per_rank_bytes, shard.n_own_bytes) - skip) // shard.rec_bytes * shard.rec_bytes
    else:
        k0 = int(np.searchsorted(shard.starts, blk, side="left"))
        k1 = int(np.searchsorted(shard.starts, blk + min(per_rank_bytes, shard.n_own_bytes), side="right")) - 1
        skip, n = int(shard.starts[k0]) - blk, int(shard.starts[k1] - shard.starts[k0])
    piece = shard.ext[shard.tail + skip:shard.tail + skip + n].cpu().numpy()
    at, whole = 0, n
    if world > 1:
        lens = [torch.zeros(1, dtype=torch.int64, device=dev if not dry_gloo(dist) else "cpu") for _ in range(world)]
        dist.all_gather(lens, torch.tensor([n], dtype=torch.int64, device=lens[0].device))
        lens = [int(x.item()) for x in lens]
        at, whole = sum(lens[:rank]), sum(lens)
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
    try:
        os.pwrite(fd, piece.tobytes(), at)
        os.close(fd)
        del piece
        if world > 1:
            dist.barrier()
        # RCCL between the ranks' GPUs (comm=None: every FileShard / iterator draws a communicator id of its OWN over the
        # process group -- an id is good for one ncclCommInitRank round); a dry run of several ranks on ONE GPU: the
        # device step over gloo
        comm = sharded.DistTransport(dist) if (world > 1 and dry_gloo(dist)) else None
        # ONE FileShard, loaded three times into the buffer it keeps (round 6: a FileShard built and closed around every
        # load measured the aftermath of the hipFree of the one before -- for ~130 ms after a buffer of this size is freed
        # every load runs at 40 GB/s instead of 52, profiles/r06_probes/loader_vs_link.txt); the first load (staging slots,
        # helper threads, new VRAM) is reported apart
        best = None
        sh = sharded.FileShard(ctx, path, rank, world, comm=comm)
        loads = []
        for _ in range(6):
            t0 = time.perf_counter()
            nb = sh.load()
            t1 = time.perf_counter()
            res = sh.scan()
            t2 = time.perf_counter()
            recs, total = int(res.row_hi - res.row_lo), int(res.total_records)
            src, tr = int(res.halo_source), sh.sh.transport()
            el = [t1 - t0, t2 - t1]
            if world > 1:                                   # (the slowest rank's load and step)
                tt = torch.tensor(el, dtype=torch.float64, device="cpu" if dry_gloo(dist) else dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = [float(x) for x in tt.tolist()]
            loads.append(el[0])
            if best is None or sum(el) < sum(best):
                best = el
        sh.close()
        # what the link gives this box right now: one raw pinned -> device copy of the same size on one stream
        link = None
        try:
            import ctypes
            L = hip.lib()
            hp = ctypes.c_void_p()
            hip.check(L.ffq_pinned_alloc(int(nb), ctypes.byref(hp)))
            ctypes.memset(hp, 1, int(nb))
            dd = ctx.dev_alloc(int(nb))
            for _ in range(2):
                t0 = time.perf_counter()
                hip.check(L.ffq_copy_h2d(ctx.handle, ctypes.c_void_p(dd), hp, int(nb), 0))
                link = nb / (time.perf_counter() - t0) / 1e9
            ctx.dev_free(dd)
            L.ffq_pinned_free(hp)
        except Exception as e:      # noqa: BLE001
            sys.stderr.write("link probe skipped: %s\n" % (e,))
        best_load, best_step = best
        # the per-rank iterator on top: Python tuples of THIS rank's records (readfastq_iter_range), bounded
        it = F.readfastq_iter_range(path, rank, world, F.entryfunc, comm=comm, ctx=ctx)
        t0 = time.perf_counter()
        n_it = 0
        for _e in it:
            n_it += 1
            if (n_it & 0xFFFF) == 0 and time.perf_counter() - t0 > budget_s:
                break
        el_it = time.perf_counter() - t0
        it.close()
        if world > 1:
            dist.barrier()
    finally:
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    return {"value": round(whole / (best_load + best_step) / 1e9, 3), "unit": "GB/s",
            "m_reads_per_s": round(total / (best_load + best_step) / 1e6, 3),
            "load_gb_s_per_rank": round(nb / best_load / 1e9, 3), "load_ms": round(best_load * 1e3, 3), "step_ms": round(best_step * 1e3, 3),
            "first_load_gb_s_per_rank": round(nb / loads[0] / 1e9, 3),
            "link_gb_s": None if link is None else round(link, 2),
            "load_over_link": None if link is None else round(nb / best_load / 1e9 / link, 3),
            "transport": tr, "halo_source": "file" if src else "ranks", "records_rank0": recs, "total_records": total,
            "iterator_m_reads_per_s_per_rank": round(n_it / el_it / 1e6, 3),
            "sample": "one %d-byte file in %s read by %d rank(s), an even share each (+ 1 MiB either side): ffq_shard_load_fd + one "
                      "ffq_shard_step per rank, max over ranks, best of 6 into the buffer the shard keeps (first_load: the first of them; link_gb_s: one raw pinned copy of the same bytes, this box, right behind); iterator: readfastq_iter_range(entryfunc) tuples of "
                      "rank 0's records, %.1f s" % (whole, d, world, el_it)}


def iterator_rates(ctx, sample_u8, n_gz, budget_s=3.0):
    """The drop-in iterator as a user of the reference calls it -- readfastq_iter(fh, fbufsize, entryfunc,
    entrypos) with this package's GPU scanner -- at the reference's own buffer size (50 000 bytes:
    /root/reference/src/demo/benchmark.py:415; reads are coalesced into k * fbufsize per device call)
    and at 16 MiB, over a plain file and over a gzip file (inflated by the library's reader thread),
    with the default entryfunc and with entryfunc_phred (the user guide's decode, from the device's
    bulk decode).  M reads/s of Python tuples on one host core; beside each gzip figure the
    reference's own C scanner through the per-record loop over the same gzip file (oracle/_ref),
    when that binary is there.  n_gz: how many bytes of the sample go into the gzip file (whole records)"""
    import gzip
    import tempfile
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    plain = os.path.join(d, "ffq_iter_%d.fq" % os.getpid())
    gz = plain + ".gz"
    with open(plain, "wb") as fh:
        fh.write(sample_u8.tobytes())
    with gzip.open(gz, "wb", compresslevel=1) as fh:
        fh.write(sample_u8[:n_gz].tobytes())
    # the same bytes as bgzip writes them (BGZF: members that carry their length; the library's reader
    # inflates them side by side, FFQ_GZ_THREADS)
    from fastqandfurious_amd import bgzf, hip
    bg = plain + ".bgz"
    with open(bg, "wb") as fh:
        fh.write(bgzf.compress(sample_u8[:n_gz].tobytes(), level=1))

    def rate(opener, fbufsize, entryfunc, scanner):
        n, t0 = 0, time.perf_counter()
        with opener() as fh:
            for _e in F.readfastq_iter(fh, fbufsize, entryfunc, scanner):
                n += 1
                if (n & 0xFFFF) == 0 and time.perf_counter() - t0 > budget_s:
                    break
        return round(n / (time.perf_counter() - t0) / 1e6, 3)

    out = {"unit": "M reads/s", "cores": 1,
           "what": "readfastq_iter(fh, fbufsize, entryfunc, GPU scanner): Python tuples per second, one host core; "
                   "plain = %d-byte file in %s, gzip = its first %d bytes at level 1 (ONE member: inflated by gz_threads threads, "
                   "csrc/ffq_pgz.h), bgzf = the same bytes in BGZF members (inflated side by side); *_stream_gb_s = decompressed GB/s of the stream front end alone (tables, no "
                   "tuples); bgzf_range_shard_gb_s = the BGZF file as one rank's range shard (sharded.BgzfFileShard: inflate + link + step)" % (sample_u8.size, d, n_gz)}
    def table_rate(path):
        # decompressed GB/s through the stream front end alone: offset tables out, no Python object per record
        best = None
        for _ in range(3):
            fd = os.open(path, os.O_RDONLY)
            t0 = time.perf_counter()
            st = hip.FileStream(ctx, fd, 1 << 24, gzip=True)
            n = sum(rows.shape[0] for rows, _f, _o, _e, _x in st)
            st.close()
            os.close(fd)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        assert n > 0
        return round(n_gz / best / 1e9, 3)

    try:
        out["gzip_stream_gb_s"] = table_rate(gz)
        out["bgzf_stream_gb_s"] = table_rate(bg)
        out["gz_threads"] = int(os.environ.get("FFQ_GZ_THREADS", "0")) or min(os.cpu_count() or 1, 32)
        # the same BGZF file as a RANGE shard (sharded.BgzfFileShard at world 1: members sized, inflated side by side into host
        # memory, over the link, one sharded step): decompressed GB/s of load + scan, best of 3
        from fastqandfurious_amd import sharded
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            bsh = sharded.BgzfFileShard(ctx, bg, 0, 1)
            try:
                bsh.load()
                nrec = int(bsh.scan().total_records)
            finally:
                bsh.close()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        assert nrec > 0
        out["bgzf_range_shard_gb_s"] = round(n_gz / best / 1e9, 3)
        out["gzip_engine"] = hip.gunzip_stats()
        for tag, opener in (("plain", lambda: open(plain, "rb")), ("gzip", lambda: gzip.open(gz, "rb")),
                            ("bgzf", lambda: gzip.open(bg, "rb"))):
            for fb in (50000, 1 << 24):
                out["%s_fbufsize_%d" % (tag, fb)] = rate(opener, fb, F.entryfunc, C.entrypos)
            out["%s_phred_fbufsize_50000" % tag] = rate(opener, 50000, F.entryfunc_phred, C.entrypos)
        try:
            from oracle import refload
            if refload.have_reference_ext():
                ext = refload.load_ext()
                out["reference_c_plain_fbufsize_50000"] = rate(lambda: open(plain, "rb"), 50000, F.entryfunc, ext.entrypos)
                out["reference_c_gzip_fbufsize_50000"] = rate(lambda: gzip.open(gz, "rb"), 50000, F.entryfunc, ext.entrypos)
        except Exception as e:      # noqa: BLE001
            out["reference_c_error"] = repr(e)
    finally:
        for f in (plain, gz, bg):
            try:
                os.unlink(f)
            except OSError:
                pass
    return out


def pushdown_rates(local_rank, dev, budget_s=3.0, nbytes=256 << 20, threshold=76):
    """The user guide's length filter (/root/reference/doc/user-guide.rst:153-180) over S-wrapped reads (50-300 bases: a
    threshold of 76 keeps ~10 %): M INPUT reads/s of the drop-in iterator with (a) the default entryfunc, nothing dropped,
    (b) the guide's function as a plain Python entryfunc behind the GPU scanner (a call per record), (c)
    entryfunc_lengthfilter(threshold): rows filtered and the kept sequences gathered on the device (ffq_stream_set_filter),
    a dropped record = one None in a list.  One host core; never `value`."""
    import tempfile
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, hip, sharded
    ctx = hip.Context(local_rank)
    sh = sharded.SyntheticShard(ctx, "wrapped", nbytes, 0, 1, dev)
    sample = sh.host_sample(nbytes)
    del sh
    ctx.close()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(d, "ffq_pushdown_%d.fq" % os.getpid())
    with open(path, "wb") as fh:
        fh.write(sample.tobytes())

    def guide(buf, posarray, globaloffset=None):
        if posarray[3] - posarray[2] < threshold:
            return buf[posarray[2]:posarray[3]]
        return None

    from oracle import ffq_oracle
    n_total = len(ffq_oracle.scan(sample)[0])            # (the file's record count: what "input reads" is counted in)

    def rate(entryfunc, reps=3):
        # the guide's own loop (`for sequence in it: if sequence is None: # do nothing ... else: # do something`) as the consumer
        best, kept = None, 0
        for _ in range(reps):
            kept = 0
            t0 = time.perf_counter()
            with open(path, "rb") as fh:
                for e in F.readfastq_iter(fh, 1 << 24, entryfunc, C.entrypos):
                    if e is not None:
                        kept += 1
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        return round(n_total / best / 1e6, 3), kept
    try:
        rate(F.entryfunc_lengthfilter(threshold), 1)                  # (warm: pinned buffers, page cache)
        a, b = rate(F.entryfunc), rate(guide, 1)
        c, e = rate(F.entryfunc_lengthfilter(threshold)), rate(F.entryfunc_lengthfilter(threshold, yield_dropped=False))
    finally:
        os.unlink(path)
    assert a[1] == n_total and b[1] == c[1] == e[1]
    return {"unit": "M input reads/s", "cores": 1, "threshold": threshold, "kept_fraction": round(c[1] / max(n_total, 1), 4),
            "unfiltered_entryfunc": a[0], "guide_function_per_record": b[0], "pushed_down": c[0], "pushed_down_kept_only": e[0],
            "speedup_vs_unfiltered": round(c[0] / a[0], 2), "speedup_kept_only_vs_unfiltered": round(e[0] / a[0], 2), "input_records": n_total,
            "what": "whole passes of readfastq_iter over a %d-byte S-wrapped file in %s (best of 3), GPU scanner, consumed by the guide's own "
                    "loop: default entryfunc / the guide's lengthfilter_entryfunc called per record / entryfunc_lengthfilter(%d): filter and "
                    "sequence gather on the device, one item per record (None for a dropped one, as the reference's loop yields) / the same "
                    "with yield_dropped=False: a dropped record never reaches the interpreter" % (sample.size, d, threshold)}


def pmc_traffic(workload, kernel="k_scan_lines<"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes of this
    workload (profiles/*/pmc_fetch_write.json): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
    FETCH_SIZE counts 64 B per 128 B request of a wide streaming read, so it is doubled
    (MI355X_MICROARCH.md, HBM section).  (bytes, file, commit the profile was taken at) or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_fetch_write.json"))):
        if os.path.basename(os.path.dirname(f)).split("_", 2)[-1] != workload:      # (rNN_x_<workload>, exactly)
            continue
        try:
            d = json.load(open(f))
        except Exception:
            continue
        sha = None
        try:
            sha = open(os.path.join(os.path.dirname(f), "COMMIT")).read().strip()
        except OSError:
            pass
        for k, v in d.items():
            if kernel in k and v.get("FETCH_SIZE_KiB_avg_per_launch", 0) > 1024:
                best = (int(2 * v["FETCH_SIZE_KiB_avg_per_launch"] * 1024 +
                            v.get("WRITE_SIZE_KiB_avg_per_launch", 0) * 1024), os.path.relpath(f, ROOT), sha)
    return best


def dry_gloo(dist):
    return dist is not None and dist.get_backend() == "gloo"


def spread(xs):
    xs = sorted(float(x) for x in xs)
    return {"min": round(xs[0], 4), "median": round(xs[len(xs) // 2], 4), "max": round(xs[-1], 4)} if xs else None


def run_workload(name, args, ctx, rank, world, dev, dist, ctl=None):
    """Time `args.steps` passes of the hot path over workload `name`; returns (line dict, closure
    of what the CPU-side extras need) on rank 0, (None, None) elsewhere.  ctl: the side group (gloo) communicator ids
    travel over at N > 1."""
    import torch
    from fastqandfurious_amd import hip, sharded
    wl = WORKLOADS[name]
    decode = wl["decode"]
    # without the decode a step's last kernel publishes the result block and the host polls it:
    # no event record behind the step (each is a few microseconds of idle GPU)
    flags = hip.F_DECODE_QUAL if decode else hip.F_POLL_RESULT
    if decode and (args.single_pass or wl.get("single_pass")) and not args.packed:
        flags |= hip.F_SINGLE_PASS               # (opt-in: csrc/ffq_fused.h -- less traffic, more time on this part)

    # ---- this rank's byte range of the logical stream, generated in HBM ----------
    split = bool(wl.get("split"))                   # the workload's bytes are the whole job's, not one GPU's
    # (N = 1 --native-step: the device step on the library's RCCL transport at world 1 instead of the in-process one)
    shard = sharded.SyntheticShard(ctx, wl["kind"], wl["bytes"] // world if split else wl["bytes"], rank, world, dev,
                                   total_records=wl["bytes"] // 322 if split else None,
                                   solo_rccl=bool(getattr(args, "native_step", False)), ctl_group=ctl,
                                   # (FFQ_BENCH_SOLO_NCCL: a torch.distributed world of ONE rank -- the N > 1 glue with no peers)
                                   transport=dist if (dist is not None and world == 1) else None)
    # Who is there (N > 1, the library's own RCCL transport): the communicators must count --gpus ranks and those ranks
    # must sit on distinct GPUs -- asserted BEFORE anything is timed, printed in the line's `comm`
    peers = None
    if hasattr(shard.scanner, "sh"):
        peers = shard.scanner.info()
        if peers and shard.scanner.sh.transport() == "rccl":
            if os.environ.get("FFQ_BENCH_RANKS_ON_ONE_GPU") == "1":      # (every rank on GPU 0 on purpose: the counts only)
                assert peers["nranks_handoff"] == world and len(set(peers["bus_ids"])) == 1, peers
            else:
                sharded.check_peers(peers, world)
    n_own = shard.n_own_bytes
    ctx.reserve(shard.ext.numel())
    table = torch.empty((shard.max_records + 64, 6), dtype=torch.int64, device=dev)
    qual = qoff = None
    if decode:
        # (room for the segmented layout of --single-pass too: 8704 bytes per 16 KiB tile)
        qual = torch.empty(max(shard.ext.numel(), ((shard.ext.numel() + 16383) >> 14) * (hip.INPLACE_STRIDE if wl.get("in_place") else hip.SEG_STRIDE)),
                           dtype=torch.int8, device=dev)
        qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ms_index, ms_chain, ms_decode, ms_total, t_done = [], [], [], [], []
    # Settling: after the set-up above the GPU has idled and its clocks have dropped; the index
    # kernel of the first ~15 ms of steps runs 5-7 % slower than in steady state
    # (tools/k1_timeline.py).  The untimed phase therefore starts with as many extra steps as make
    # ~settle-ms of GPU work -- a count computed from the workload's size, the same on every rank.
    settle_steps = int(args.settle_ms * 1e-3 / ((wl["bytes"] // world if split else wl["bytes"]) / 4.0e12)) if args.settle_ms > 0 else 0
    n_untimed = settle_steps + args.warmup

    comm_steps = []

    def note(out):
        if getattr(out, "comm", None):
            comm_steps.append(out.comm)
        if out.res.ms_index > 0:                       # (steps submitted with FFQ_F_NO_TIMING carry no marks)
            ms_index.append(out.res.ms_index)
        ms_chain.append(out.res.ms_chain)
        ms_decode.append(out.res.ms_decode)
        ms_total.append(out.res.ms_total)
        t_done.append(time.perf_counter())

    extra_ctx = []
    if world == 1 and not args.sharded_step and not args.lanes_step:
        # Single range: steps are submitted through two contexts that share their HIP streams,
        # one step ahead (ffq_scan_submit / ffq_scan_wait): while the host waits for step i the
        # kernels of step i+1 are already queued behind it.  Every step is a full scan of the
        # resident buffer into its own output table.
        ctx2 = hip.Context(share=ctx)
        extra_ctx.append(ctx2)
        ctx2.reserve(shard.ext.numel())
        ctxs = (ctx, ctx2)
        tables = (table, torch.empty_like(table))
        quals = (qual, torch.empty_like(qual) if decode else None)
        qoffs = (qoff, torch.empty_like(qoff) if decode else None)
        torch.cuda.synchronize()

        # The index kernel is timed (HIP events around it, on the scan stream) on every TIME_EVERY-th step; the
        # other steps carry no stream marker at all (FFQ_F_NO_TIMING: each marker is a barrier packet with a
        # few microseconds of idle GPU around it, and without one the index kernel starts while the previous
        # step's last one-workgroup kernel is still running).  --time-every 1: marks on every step.
        te = max(1, args.time_every) if not decode else 1

        def submit(i):
            k = i & 1
            ctxs[k].scan_submit(shard.ext.data_ptr(), shard.n_own_bytes, tables[k].data_ptr(), tables[k].shape[0],
                                sentinel=True, eof=True, flags=flags | (hip.F_NO_TIMING if (i % te) else 0),      # (te: read at call time)
                                d_qual=quals[k].data_ptr() if decode else None,
                                qual_cap=quals[k].numel() if decode else 0,
                                d_qoff=qoffs[k].data_ptr() if decode else None)

        def wait(i):
            rc, res = ctxs[i & 1].scan_wait()
            assert rc == hip.OK
            return sharded.ScanOutput(res, int(res.n_records), 0, int(res.n_records), -1, 0)

        def run(nsteps, record):
            out = None
            submit(0)
            for i in range(1, nsteps):
                submit(i)
                out = wait(i - 1)
                if record:
                    note(out)
            out = wait(nsteps - 1)
            if record:
                note(out)
            return out

        if n_untimed:
            out = run(n_untimed, False)
        barrier()
        t0 = time.perf_counter()
        out = run(args.steps, True)
        barrier()
        elapsed = time.perf_counter() - t0
        table = tables[(args.steps - 1) & 1]
        qual, qoff = quals[(args.steps - 1) & 1], qoffs[(args.steps - 1) & 1]
    elif world == 1 and not args.lanes_step:
        def step():
            return shard.scan(table, flags=flags, qual=qual, qoff=qoff)

        for _ in range(n_untimed):
            out = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
            note(out)
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        # Byte-range shards: the same one-step-ahead queue per rank.  A step = halo hand-off
        # (RCCL send/recv, ordered on the scan stream) + scan of [tail | own | head] + cut of the
        # own rows + hand-off verification (one all_gather of a few words per rank).  submit(i + 1)
        # is queued before finish(i); finish's small kernel runs on a stream of its own.
        shard.make_lanes(2)
        tables = (table, torch.empty_like(table))
        quals = (qual, torch.empty_like(qual) if decode else None)
        qoffs = (qoff, torch.empty_like(qoff) if decode else None)
        torch.cuda.synchronize()

        # (timing marks around the index kernel on every --time-every-th step only, as in the single-range queue; the figures
        # `roofline` is computed from come from marked steps run after the timed region)
        te_l = max(1, args.time_every) if (not decode and getattr(shard, "native", False)) else 1

        def run(nsteps, record, every=None):
            ev = te_l if every is None else every
            out = None
            fl = lambda i: flags | (hip.F_NO_TIMING if (i % ev) else 0)
            shard.submit(0, tables[0], fl(0), quals[0], qoffs[0])
            for i in range(1, nsteps):
                k = i & 1
                shard.submit(k, tables[k], fl(i), quals[k], qoffs[k])
                out = shard.finish(k ^ 1)
                if record:
                    note(out)
            out = shard.finish((nsteps - 1) & 1)
            if record:
                note(out)
            return out

        # (diagnostics, tests/test_watchdog.py: FFQ_BENCH_INJECT_STALL=hand-off|scan|gather makes the first step of this rank hang
        # there -- a kernel that waits for a host flag --, so that the watchdog and the recovery below run without a broken peer)
        inj = os.environ.get("FFQ_BENCH_INJECT_STALL")
        if inj and getattr(shard, "native", False):
            shard.scanner.sh.inject_stall(hip.STAGE_NAMES.index(inj), 60.0)
        # A step that does not come back is an exception on every rank (the step watchdog: hip.FFQTimeout names the stage
        # and the ranks), not a hang.  On the RCCL transport the steps are then taken ONCE more in serial mode -- a new
        # communicator, ONE, everything on the scan stream (SyntheticShard.recover_serial) -- and `comm` says so.
        for attempt in (0, 1):
            try:
                if n_untimed:
                    out = run(n_untimed, False)
                barrier()
                t0 = time.perf_counter()
                out = run(args.steps, True)
                barrier()
                elapsed = time.perf_counter() - t0
                if te_l > 1:
                    t_keep, c_keep = list(t_done), list(comm_steps)
                    ms_index.clear()
                    run(max(20, args.steps + (args.steps & 1)), True, every=1)      # (an even count: the last step's lane stays the same)
                    t_done[:] = t_keep
                    comm_steps[:] = c_keep
                    barrier()
                break
            except hip.FFQTimeout as e:
                sys.stderr.write("[bench rank %d] %s\n" % (rank, e))
                if attempt or not (getattr(shard, "native", False) and shard.scanner.sh.transport() == "rccl"):
                    raise
                sys.stderr.write("[bench rank %d] -> once more with the serial step (one communicator, one stream)\n" % rank)
                shard.recover_serial(e)
                for lst in (comm_steps, ms_index, ms_chain, ms_decode, ms_total, t_done):
                    lst.clear()
        table = tables[(args.steps - 1) & 1]
        qual, qoff = quals[(args.steps - 1) & 1], qoffs[(args.steps - 1) & 1]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([out.n_own_records, n_own], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        total_records, total_bytes = int(tot[0].item()), int(tot[1].item())
    else:
        total_records, total_bytes = out.n_own_records, n_own
    assert out.res.path in (0, 3, 6, 8), "a parallel chain path must be the one measured (got path %d)" % out.res.path

    # The same step once more OUTSIDE the timed region, one at a time and with an end event instead of
    # the polled completion word: the device span first kernel -> last kernel of ONE step (res.ms_total,
    # HIP events on the scan stream) -- round 1's definition of the whole-path time, kept so that rounds
    # compare like for like.  (Wall-clock steps are shorter: the next step's index kernel starts behind
    # this step's last kernel with no idle gap, and host-side waits are hidden.)
    span_ms = []
    if world == 1 and not args.sharded_step and not args.lanes_step:
        # ... and the figures `roofline` is computed from: MARKED more steps after the timed region, pipelined as the
        # timed ones but with the two HIP-event marks around the dominant kernel on EVERY step (the timed region itself
        # carries them on every --time-every-th step only: a mark is a few microseconds of idle GPU)
        if te > 1:
            ms_index.clear()
            te_keep, te = te, 1
            n_marked = max(20, args.steps)
            submit(0)
            for i in range(1, n_marked):
                submit(i)
                o2 = wait(i - 1)
                ms_index.append(o2.res.ms_index)
            ms_index.append(wait(n_marked - 1).res.ms_index)
            te = te_keep
        for _ in range(5):
            ctxs[0].scan_submit(shard.ext.data_ptr(), shard.n_own_bytes, tables[0].data_ptr(), tables[0].shape[0],
                                sentinel=True, eof=True, flags=flags & ~hip.F_POLL_RESULT,
                                d_qual=quals[0].data_ptr() if decode else None,
                                qual_cap=quals[0].numel() if decode else 0,
                                d_qoff=qoffs[0].data_ptr() if decode else None)
            rc, res = ctxs[0].scan_wait()
            assert rc == hip.OK
            span_ms.append(float(res.ms_total))
        torch.cuda.synchronize()

    # ---- parity spot check on the measured output (size-independent properties) -----
    shard.verify(table, out)
    if decode:
        shard.verify_decode(table, out, qual, qoff)

    # what this box's memory system delivers to a pure streaming read of the same buffer in the scan
    # kernel's launch geometry (16 B per lane, non-temporal), right behind the timed region: boxes of
    # the pool differ, and a throttled one shows here first
    probe_gbs = None
    if rank == 0 and world == 1:
        nprobe = min(shard.n_own_bytes, 2 * GIB) & ~((1 << 14) - 1)
        if nprobe >= (1 << 14):
            try:
                rp = hip.ReadProbe(ctx.device)              # the instrumented build, beside the product library
                probe_gbs = nprobe / (rp.read_ms(shard.ext.data_ptr(), nprobe, 6, 10) * 1e-3) / 1e9
                rp.close()
            except Exception as e:      # noqa: BLE001  (diagnostics only: never fails the bench line)
                sys.stderr.write("hbm_read_probe skipped: %s\n" % (e,))

    line = None
    if rank == 0:
        step_s = elapsed / args.steps
        value = total_bytes / step_s / 1e9
        n_rec = out.n_own_records
        # Dominant kernel of the workload (the longest launch of a step): the Phred decode when
        # decoding, else the line-index kernel.  Algorithmic bytes per launch:
        #   k_scan_lines     every byte of the scanned buffer read once + 2 B of index per newline
        #   k_decode_stream  quality bytes read + written + 16 B of (offset, pos4) per record
        if decode and out.res.path == 8:
            # one pass on the general path (k_scan_lines<.., WIDE>): every byte read once AND written once (decoded in place),
            # 2 B of index per newline -- the kernel's own bytes; what of them is quality shows in path_roofline
            dom, t_dom = "k_scan_lines<WIDE>", float(np.mean(ms_index)) * 1e-3
            algo = 2 * shard.ext_scanned_bytes + 2 * int(out.res.n_lines)
            traffic = pmc_traffic(name, "k_scan_lines<")
        elif decode and out.res.path == 6:
            # single pass (ffq_fused.h): the index kernel also writes the decoded qualities (segmented) -- every byte
            # of the buffer read once, 2 B of index per newline and the decoded bytes written
            dom, t_dom = "k_scan_seg", float(np.mean(ms_index)) * 1e-3
            nq = int((table[:n_rec, 5] - table[:n_rec, 4]).sum().item())
            algo = shard.ext_scanned_bytes + 2 * int(out.res.n_lines) + nq
            traffic = pmc_traffic(name, "k_scan_seg")
        elif decode:
            dom, t_dom = "k_decode_stream", float(np.mean(ms_decode)) * 1e-3
            algo = 2 * int(out.res.n_qual_bytes) + 16 * int(out.n_rows)
            traffic = pmc_traffic(name, "k_decode_stream")
        else:
            dom, t_dom = "k_scan_lines", float(np.mean(ms_index)) * 1e-3
            algo = shard.ext_scanned_bytes + 2 * int(out.res.n_lines)
            traffic = pmc_traffic(name, "k_scan_lines<")
        achieved = algo / t_dom / 1e9
        # whole path priced with SURVEY.md 8(d): record bytes + 48 B row (+ decode bytes), over the
        # wall-clock step (steps are queued one ahead and their small kernels overlap the next
        # step's index kernel, so a step's own first-to-last-kernel span is longer than a step)
        algo_path = n_own + 48 * n_rec
        if decode:
            algo_path += int((table[:n_rec, 5] - table[:n_rec, 4]).sum().item()) + 8 * n_rec
        steps_ms = np.diff(np.array([t0] + t_done)) * 1e3
        line = {
            "metric": "GB/s FASTQ parsed",
            "value": round(value, 3),
            "unit": "GB/s",
            "m_reads_per_s": round(total_records / step_s / 1e6, 3),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(step_s * 1e3, 4),
            "ms_per_step_spread": spread(steps_ms[1:] if len(steps_ms) > 2 else steps_ms),
            "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "build_id": hip.build_id(),
            "config": {
                "workload": name,
                "description": "%s synthetic FASTQ, %d bytes/GPU, %d records/GPU%s"
                               % ("S-single 150 bp" if wl["kind"] == "single" else
                                  "tests.py template x n (27-byte records, every tile dense)" if wl["kind"] == "dense" else "S-wrapped 50-300 bp",
                                  n_own, n_rec, "" if not decode else
                                  (", quality->int8 decode, segmented output (record i = qual[qoff[i] : qoff[i] + pos5 - pos4]), one pass"
                                   if out.res.path == 6 else
                                   ", quality->int8 decode IN PLACE (every byte of the buffer decoded at its own offset by the index pass, qoff[i] = pos4's offset), one pass"
                                   if out.res.path == 8 else ", quality->int8 decode, packed CSR stream, two passes")),
                "diagnostic": bool(wl.get("diagnostic")),      # True: kept for comparison, not what a caller of the product runs
                "bytes_per_gpu": n_own,
                "records_per_gpu": n_rec,
                "total_bytes": total_bytes,
                "total_records": total_records,
                "sharding": ("byte ranges, halo hand-off over %s" % (
                                 "RCCL, the step behind the C ABI (ffq_shard_step_submit / _wait)" if dist.get_backend() == "nccl"
                                 else "%s: a functional dry run, every rank on ONE GPU -- the device step over a hosted transport (ffq_shard_create_hosted)" % dist.get_backend())
                             if world > 1 else "single range"),
            },
            # what a step costs beside its scan (N > 1, the library's own step: ffq_shard_step_*): device time of the halo
            # hand-off (ncclSend / ncclRecv in one group, on the hand-off stream beside the previous step's scan) and of the
            # gather of the eight hand-off words (ncclAllGather behind the scan), bytes handed off per rank and step, and how
            # often a rank had to scan again (a look-ahead grown, a guessed entry contradicted) -- rank 0's figures
            "comm": None if not comm_steps else {
                "transport": ("RCCL (ffq_shard_*, include/ffq.h)" if shard.scanner.sh.transport() == "rccl" else shard.scanner.sh.transport())
                             if hasattr(shard.scanner, "sh") else "host step (ffq_shard_host_step) over %s" % shard.scanner.transport_name,
                # who was there: ranks of the communicator the words were gathered on (ncclCommCount; asserted == --gpus
                # before the timed region), every rank's GPU by PCI bus id (gathered over RCCL at set-up; asserted distinct),
                # which step ran (pipelined: two communicators, three streams; serial: one and one) and, if the watchdog
                # tripped on the way, what it said
                "nranks": comm_steps[-1].get("nranks"),
                "mode": comm_steps[-1].get("mode"),
                "bus_ids": (shard.scanner.info() if hasattr(shard.scanner, "sh") else {}).get("bus_ids"),
                "watchdog_s": (peers or {}).get("timeout_s"),
                "recovered_from": getattr(shard, "recovered", None),
                "steps": len(comm_steps),
                "handoff_ms": round(float(np.mean([c["handoff_ms"] for c in comm_steps])), 4),
                "handoff_bytes": int(np.mean([c["handoff_bytes"] for c in comm_steps])),
                "allgather_ms": round(float(np.mean([c["allgather_ms"] for c in comm_steps])), 4),
                "rescan_rounds": int(sum(c["rescan_rounds"] for c in comm_steps)),
                "regathers": int(sum(c["regathers"] for c in comm_steps)),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic[0] if traffic else None,
                "traffic_source": traffic[1] if traffic else None,
                "traffic_commit": traffic[2] if traffic else None,
                "algorithmic_bytes_per_launch": algo,
                "avg_launch_ms": round(t_dom * 1e3, 4),
                "launches_timed": len(ms_decode if dom == "k_decode_stream" else ms_index),
                "launch_ms_spread": spread(ms_decode if dom == "k_decode_stream" else ms_index),
            },
            "hbm_read_probe": None if probe_gbs is None else {
                "value": round(probe_gbs, 1), "unit": "GB/s", "frac_of_peak": round(probe_gbs / HBM_PEAK_GBS, 4),
                "what": "pure non-temporal 16 B/lane read of the same resident buffer (k_read_probe of libffq_probe.so, "
                        "the instrumented build; not part of the product library), this box, this run"},
            "path_roofline": {
                "what": "all kernels of one step over the wall-clock step, SURVEY.md 8(d) bytes (record bytes + 48 B row%s)"
                        % (" + decoded bytes + 8 B CSR offset" if decode else ""),
                "algorithmic_bytes_per_step": algo_path,
                "achieved": round(algo_path / step_s / 1e9, 2),
                "frac": round(algo_path / step_s / 1e9 / HBM_PEAK_GBS, 4),
                "ms_index": round(float(np.mean(ms_index)), 4),
                "ms_decode": round(float(np.mean(ms_decode)), 4),
                "basis": "throughput: wall-clock time per step with steps queued one ahead (what the driver recomputes "
                         "from ms_per_step); device_span_* below is the round-1 definition",
                "device_span_ms": round(float(np.median(span_ms)), 4) if span_ms else None,
                "device_span_frac": round(algo_path / (float(np.median(span_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if span_ms else None,
                "device_span_what": "median of 5 unpipelined steps after the timed region: first kernel -> last kernel "
                                    "of one step by HIP events on the scan stream (ffq_scan_result.ms_total)",
            },
        }
    for c2 in extra_ctx:
        c2.close()
    return line, (shard, flags)


def shape_rates(local_rank, dev, gib=4):
    """Read shapes that are no BASELINE config (never `value`): long four-line reads with the decode in ONE pass (qualities
    decoded in place, csrc/ffq_fused.h k_scan_ident), wrapped reads of kilobases (the group kernels / list ranking), FASTA.
    A 32 MiB block of distinct records, generated here with their rows (the generator knows every offset: pos0 = the '@',
    pos1 = the header's newline, pos3 = the newline in front of the '+' line, pos4 = pos3 + 3, pos5 = pos4 + pos3 - pos2), is
    repeated on the device to `gib` GiB; EVERY row of every repeat is compared with the block's rows + the repeat's offset,
    the decoded bytes of the first and the last repeat with the block's own bytes - 33."""
    import torch
    from fastqandfurious_amd import hip
    rng = np.random.default_rng(2025)
    out = {"what": shape_rates.__doc__.split("\n\n")[0].replace("\n    ", " "), "bytes": None}
    ctx = hip.Context(local_rank)

    def block_of(L, wrap, fasta=False):
        parts, rows, at, i = [], [], 0, 0
        while at < (32 << 20):
            n = int(rng.integers(L // 2, L * 3 // 2 + 1))
            seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()
            w = wrap or n
            st = b"\n".join(seq[k:k + w] for k in range(0, n, w))
            if fasta:
                h = b">entry%d len=%d" % (i, n)
                r = h + b"\n" + st + b"\n"
                rows.append((at, at + len(h), at + len(h) + 1, at + len(r) - 1, -1, -1))
            else:
                q = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=n).tobytes()
                qt = b"\n".join(q[k:k + w] for k in range(0, n, w))
                h = b"@read%d len=%d" % (i, n)
                r = h + b"\n" + st + b"\n+\n" + qt + b"\n"
                p1 = at + len(h); p3 = p1 + 1 + len(st); p4 = p3 + 3
                rows.append((at, p1, p1 + 1, p3, p4, p4 + len(st)))
            parts.append(r); at += len(r); i += 1
        return np.frombuffer(b"".join(parts), dtype=np.uint8), np.array(rows, dtype=np.int64)

    for name, L, wrap, flags, fasta in (("four-line ~3 kbp, decode in one pass (in place)", 3000, 0, hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, False),
                                        ("four-line ~30 kbp, decode in one pass (in place)", 30000, 0, hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, False),
                                        ("wrapped at 80 columns, ~1 kbp", 1000, 80, 0, False),
                                        ("wrapped at 80 columns, ~3 kbp", 3000, 80, 0, False),
                                        ("wrapped at 80 columns, ~20 kbp", 20000, 80, 0, False),
                                        ("FASTA at 60 columns, ~10 kbp entries", 10000, 60, 0, True)):
        block, rows = block_of(L, wrap, fasta)
        reps = max(1, (gib << 30) // block.size)
        hb = torch.from_numpy(block.copy()).to(dev)
        d = hb.repeat(reps) if not fasta else torch.cat([torch.tensor([10], dtype=torch.uint8, device=dev), hb.repeat(reps)])
        nb, n = rows.shape[0], rows.shape[0] * reps - (1 if fasta else 0)     # (FASTA: the last entry is never COMPLETE)
        table = torch.empty((n + 64, 6), dtype=torch.int64, device=dev)
        decode = bool(flags & hip.F_DECODE_QUAL)
        qual = torch.zeros(((d.numel() + 16383) >> 14) * hip.INPLACE_STRIDE, dtype=torch.int8, device=dev) if decode else None
        qoff = torch.zeros(n + 65, dtype=torch.int64, device=dev) if decode else None
        torch.cuda.synchronize()
        ctx.reserve(d.numel()); ctx.forget()
        ms = []
        for it in range(6):
            if fasta:
                rc, res = ctx.scan_fasta_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
            else:
                rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64, flags=flags,
                                          d_qual=qual.data_ptr() if decode else None, qual_cap=qual.numel() if decode else 0,
                                          d_qoff=qoff.data_ptr() if decode else None)
            assert rc == 0 and int(res.n_records) == n, (name, rc, int(res.n_records), n)
            if it >= 2:
                ms.append(float(res.ms_total))
        want = torch.from_numpy(rows).to(dev)
        shift = (torch.arange(reps, device=dev, dtype=torch.int64) * block.size).view(reps, 1, 1)
        exp = (want.view(1, nb, 6) + shift + (1 if fasta else 0)).view(-1, 6)[:n]
        if fasta:
            exp = exp.clone(); exp[:, 4:] = -1
        ok = bool((table[:n] == exp).all())
        if decode and ok:
            mask = np.zeros(block.size, dtype=bool)
            for a, b in zip(rows[:, 4], rows[:, 5]):
                mask[a:b] = True
            at = torch.from_numpy(np.nonzero(mask)[0]).to(dev)
            src = (hb[at].to(torch.int16) - 33).to(torch.int8)
            ok = bool((qoff[:n] == table[:n, 4]).all())
            for r in (0, reps - 1):
                ok = ok and bool((qual[r * block.size:(r + 1) * block.size][at] == src).all())
        assert ok, "shape_rates: %s does not verify" % name
        best = min(ms)
        out[name] = {"value": round(d.numel() / (best * 1e-3) / 1e9, 1), "unit": "GB/s", "ms": round(best, 4), "path": int(res.path),
                     "retries": int(res.retries), "records": n, "bytes": int(d.numel()), "verified": "every row" + (", decoded bytes of two repeats" if decode else "")}
        del d, hb, table, qual, qoff, want, exp
        torch.cuda.empty_cache()
    out.pop("bytes")
    ctx.close()
    return out


def guarded(fn, *a, **kw):
    """An extra leg of the line (everything that is not the timed region): its result, or what went wrong -- the line goes
    out either way."""
    try:
        return fn(*a, **kw)
    except Exception as e:      # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    t_main = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="single-1g", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true",
                    help="N=1: do not also time decode-10g and wrapped-10g (BASELINE configs[2], [3])")
    ap.add_argument("--settle-ms", type=float, default=50.0,
                    help="untimed steps in front of the warm-up until the GPU has been busy this long "
                         "(its clocks take ~15 ms of load to settle after an idle spell); 0: none")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--packed", action="store_true",
                    help="decode workloads: the packed CSR stream of rounds 1-2 (two passes) even where the workload asks for the single pass")
    ap.add_argument("--time-every", type=int, default=4,
                    help="N=1, no decode: HIP-event marks around the index kernel on every n-th step only (default 4; "
                         "1: every step)")
    ap.add_argument("--single-pass", action="store_true",
                    help="decode workloads: the caller accepts the qualities segmented (FFQ_F_SINGLE_PASS) and the index pass "
                         "writes them itself: the input is read once")
    ap.add_argument("--sharded-step", action="store_true",
                    help="N=1 through the synchronous step the N>1 ranks run (diagnostics)")
    ap.add_argument("--native-step", action="store_true",
                    help="N=1 through the step the N>1 ranks run behind the C ABI (ffq_shard_step_submit / _wait on the library's "
                         "RCCL transport, world 1, two lanes): its queueing, the gather of the hand-off words and the comm "
                         "figures, with no peers (diagnostics)")
    ap.add_argument("--lanes-step", action="store_true",
                    help="N=1 through the pipelined step of the N>1 ranks (two lanes, hand-off with no peers): "
                         "what the host side of a sharded step costs (diagnostics)")
    args = ap.parse_args()
    if args.native_step:
        args.lanes_step = True

    # FFQ_BENCH_RANKS_ON_ONE_GPU=1: the N > 1 line over RCCL with every rank on GPU 0 -- ranks that tell RCCL they sit on
    # different hosts (NCCL_HOSTID) pass its duplicate-GPU check and talk over the socket transport (loopback).  Functional
    # only (the ranks share the GPU, nothing is xGMI): it is how a one-GPU box shows that `--gpus N` runs on librccl with
    # real peers.  (Before torch -- and with it librccl -- is loaded.)
    one_gpu = os.environ.get("FFQ_BENCH_RANKS_ON_ONE_GPU") == "1"
    if one_gpu and "RANK" in os.environ:
        os.environ["NCCL_HOSTID"] = "ffq-rank-as-host-%s" % os.environ["RANK"]
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")

    import torch
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import hip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one rank per GPU ourselves (the driver's own command line does the same)
        import socket
        import subprocess
        import random
        port = None
        for _ in range(200):
            # (below the kernel's ephemeral range: a bind-to-0 port can be taken by anybody's outgoing connection before the
            # launcher listens on it)
            cand = random.randrange(15000, 30000)
            with socket.socket() as so:
                try:
                    so.bind(("127.0.0.1", cand))
                except OSError:
                    continue
            port = cand
            break
        if port is None:
            raise SystemExit("no free rendezvous port between 15000 and 30000")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # FFQ_BENCH_DRY_MULTI=1: every rank on GPU 0 over gloo -- a functional dry run of the N > 1
    # code path on a one-GPU box (its number means nothing; RCCL refuses two ranks per device)
    dry = os.environ.get("FFQ_BENCH_DRY_MULTI") == "1"
    if dry or one_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        # (a line that says so, not a traceback the driver has to dig for)
        if rank == 0:
            print(json.dumps({"metric": "GB/s FASTQ parsed", "value": None, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                              "config": {"workload": args.workload},
                              "error": "rank %d has LOCAL_RANK %d but this process sees %d GPU(s): one process per GPU needs every rank's device visible"
                                       % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0)}), flush=True)
        raise SystemExit(1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    ctl = None
    # FFQ_BENCH_SOLO_NCCL=1 under the launcher with ONE rank: everything the N > 1 line does on RCCL -- the nccl process group,
    # the gloo side group the communicator ids travel over, the library's own RCCL transport with its peers check, the lanes,
    # the reductions of the timings -- with no peer (tests/test_watchdog.py: the glue the first multi-GPU run will execute)
    solo_nccl = os.environ.get("FFQ_BENCH_SOLO_NCCL") == "1" and world == 1 and "RANK" in os.environ
    if solo_nccl:
        args.lanes_step = True
    if world > 1 or solo_nccl:
        import datetime
        import torch.distributed as dist
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            # (torch's own collectives -- barriers, the reduction of the timings -- give up after 5 minutes instead of the
            # default 10; the control group is gloo: the library's communicator ids travel over it, also when a new one is
            # needed because the GPUs' fabric has just swallowed a collective)
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
            ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))

    ctx = hip.Context(local_rank)
    try:
        line, (shard, flags) = run_workload(args.workload, args, ctx, rank, world, dev, dist, ctl)
    except Exception as e:      # noqa: BLE001
        # the measured workload itself failed (N > 1: a step that did not come back even in serial mode, ranks that share a
        # GPU, a communicator that counts the wrong number of ranks): a line that SAYS so instead of a silent death at the
        # driver's timeout
        import traceback
        traceback.print_exc()
        if rank == 0:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            print(json.dumps({"metric": "GB/s FASTQ parsed", "value": None, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                              "config": {"workload": args.workload}, "error": "%s: %s" % (type(e).__name__, e),
                              "wall_s": round(time.perf_counter() - t_main, 1)}), flush=True)
        os._exit(1)              # (no collective tear-down: a peer may be gone)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            sample = shard.host_sample(256 << 20)
            line["cpu_baseline"] = guarded(cpu_baseline, sample, args.cpu_seconds)
            line["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            line["cpu_baseline"]["all_cores"] = guarded(cpu_baseline_all_cores, sample[:64 << 20], args.cpu_seconds / 2)
            line["cpu_baseline"]["python_iterator"] = guarded(cpu_iterator_rate, sample.tobytes(), 3.0)
            line["cpu_baseline"]["what"] = ("`value` is the PORT (oracle/ffq_oracle.c, whole-buffer chain, no interpreter); "
                                            "reference_c_extension = the reference's own compiled scanner, bare calls; "
                                            "reference_c_iterator = the same scanner inside the per-record iterator loop")
            line["cpu_baseline"]["reference_c_extension"] = guarded(cpu_reference_c, sample.tobytes(), 3.0)
            line["cpu_baseline"]["reference_c_iterator"] = guarded(cpu_reference_iter, sample.tobytes(), 3.0)
            line["host_inclusive"] = guarded(host_inclusive, ctx, sample, flags)
            line["host_inclusive"]["stream_fd"] = guarded(lambda: stream_inclusive(ctx, shard.host_sample(1 << 30)))
            line["host_inclusive"]["iterator"] = guarded(iterator_rates, ctx, sample, int(sample.size))
            del sample
            line["host_inclusive"]["pushdown"] = guarded(pushdown_rates, local_rank, dev)
        elif not args.no_cpu_baseline:
            # N > 1: the same CPU legs beside the multi-GPU line, on rank 0's host cores with a 3 s budget each (the other
            # ranks wait at the next collective); north_star: "GB/s and reads/s at 1/2/4/8 GPUs reported next to the
            # reference C path timed on the same box's host cores"
            sample = shard.host_sample(64 << 20)
            line["cpu_baseline"] = guarded(cpu_baseline, sample, min(args.cpu_seconds, 3.0))
            line["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            line["cpu_baseline"]["all_cores"] = guarded(cpu_baseline_all_cores, sample[:16 << 20], min(args.cpu_seconds, 3.0))
            line["cpu_baseline"]["reference_c_extension"] = guarded(cpu_reference_c, sample.tobytes(), 2.0)
            line["cpu_baseline"]["reference_c_iterator"] = guarded(cpu_reference_iter, sample.tobytes(), 2.0)
            line["cpu_baseline"]["what"] = ("rank 0's host, while the other ranks wait: `value` is the PORT (oracle/ffq_oracle.c) on one core, "
                                            "all_cores the same on every core; reference_c_* = the reference's own compiled scanner")
            del sample
        else:
            line["cpu_baseline"] = None
    if not args.no_cpu_baseline and (world == 1 or dry_gloo(dist) or os.environ.get("FFQ_BENCH_SHARDED_FILE") == "1"):
        # file-backed byte-range shards (collective: every rank takes part; the line is rank 0's).  On a real multi-GPU run
        # only on request (FFQ_BENCH_SHARDED_FILE=1): it is a host-inclusive extra, and nothing that is not the metric
        # should be able to hold up the ranks of the measured line
        # (guarded at every N: with the step watchdog a collective that does not come back is an exception on every rank,
        # not a hang, and the measured line goes out either way)
        sf = guarded(sharded_file, ctx, shard, rank, world, dev, dist)
        if rank == 0:
            line.setdefault("host_inclusive", {})
            if line["host_inclusive"] is None:
                line["host_inclusive"] = {}
            line["host_inclusive"]["sharded_file"] = sf
    # N = 1, default workload: BASELINE configs[2] and [3] timed the same way, under their own key
    # (`value` stays configs[1]'s)
    # (at every N also BASELINE configs[4]: the one 100 GiB stream cut into `world` ranges)
    if args.workload == "single-1g" and not args.no_others and not args.sharded_step and not args.lanes_step:
        del shard
        torch.cuda.empty_cache()
        others = {}
        names = ("decode-10g", "decode-10g-packed", "wrapped-10g", "dense-1g", "single-100g") if world == 1 else ("single-100g",)
        if os.environ.get("FFQ_BENCH_DRY_MULTI") == "1" or os.environ.get("FFQ_BENCH_RANKS_ON_ONE_GPU") == "1":
            names = ("single-4g-split",)             # (the dry run shares ONE GPU between the ranks)
        for other in names:
            ctx_o = hip.Context(local_rank)
            if world == 1:
                got = guarded(run_workload, other, args, ctx_o, rank, world, dev, dist)
                if isinstance(got, dict):          # (the extra failed: say so under its name)
                    others[other] = got
                    continue
                ol, keep = got
            else:
                # (guarded at N > 1 too: a step that does not come back is an exception on every rank -- the watchdog --, and
                # the measured line goes out either way)
                got = guarded(run_workload, other, args, ctx_o, rank, world, dev, dist, ctl)
                if isinstance(got, dict):
                    others[other] = got
                    continue
                ol, keep = got
            del keep
            ctx_o.close()
            torch.cuda.empty_cache()
            if rank == 0:
                others[other] = {k: ol[k] for k in ("value", "unit", "m_reads_per_s", "n_gpus", "scaling", "ms_per_step",
                                                     "ms_per_step_spread", "settle_steps", "config", "comm", "roofline",
                                                     "hbm_read_probe", "path_roofline")}
        if rank == 0:
            line["other_workloads"] = others
        if rank == 0 and world == 1 and os.environ.get("FFQ_BENCH_DRY_MULTI") != "1":
            line["shapes"] = guarded(shape_rates, local_rank, dev)
    if rank == 0:
        # (RCCL prints its version banner through C stdio: out with it first, the JSON line is the last line of stdout)
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        line["wall_s"] = round(time.perf_counter() - t_main, 1)          # this process, start to line (imports and set-up included)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
