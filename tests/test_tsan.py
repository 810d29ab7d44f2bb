"""The library's host threads under ThreadSanitizer -- and the same runs under AddressSanitizer + UBSan -- (SURVEY.md
section 5: the reference is single-threaded and needs no race detector; this build's reader pool, gzip pools, chunked
inflate and in-process shard world do).

A -fsanitize=thread (then: address,undefined) build of libffq_hip.so (hipcc instruments the HOST code; no device is touched by what runs here) +
tests/tsan_host_threads.cpp + the oracle (the scan of the shard mode), all in one TSan runtime:
  * ffq_gunzip_fd = the stream front end's gzip reader: BGZF members side by side (GzPool), ONE plain member by several
    threads (csrc/ffq_pgz.h: chunks entered at block headers, hand-overs at odd bits, stitch, CRC pieces combined),
    concatenated members, zlib taking over, corrupt and cut files;
  * the helper threads of a context (csrc/ffq_pool.h): slices of consecutive chunks through one queue;
  * ffq_shard_host_step with k ranks as threads over ffq_shard_world's breakable barrier (csrc/ffq_shard_proto.h): the
    settle rounds of the shard protocol, and a rank that fails while the others wait.
0 reports is the bar; the detector is shown to be awake by a deliberate race.  What stays outside: the feeder thread of
ffq_stream_* and the device step's streams (they need the GPU, and the HIP runtime is not instrumented)."""
import gzip
import os
import random
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module", params=("thread", "address,undefined"))
def tsan(request, tmp_path_factory):
    import fastqandfurious_amd  # noqa: F401
    san = request.param
    rtname = "libclang_rt.tsan-x86_64.so" if san == "thread" else "libclang_rt.asan-x86_64.so"
    from fastqandfurious_amd import build
    d = tmp_path_factory.mktemp("san")
    if not os.path.exists(CLANG):
        pytest.skip("no clang++")
    try:
        hipcc = build._hipcc()
    except RuntimeError:
        pytest.skip("no hipcc")
    rt = None
    for cand in subprocess.run([CLANG, "--print-runtime-dir"], capture_output=True, text=True).stdout.split() + \
            [os.path.join(os.path.dirname(os.path.dirname(CLANG)), "lib", "clang", v, "lib", "linux")
             for v in (os.listdir(os.path.join(os.path.dirname(os.path.dirname(CLANG)), "lib", "clang")) if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(CLANG)), "lib", "clang")) else [])]:
        if os.path.exists(os.path.join(cand, rtname)):
            rt = cand
            break
    if rt is None:
        pytest.skip("no shared sanitizer runtime (%s)" % rtname)
    lib = d / "libffq_tsan.so"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=" + san, "-shared-libsan",
                        "-Wno-unused-function", "-Wno-option-ignored", '-DFFQ_BUILD_ID="tsan"', "-o", str(lib),
                        os.path.join(build.CSRC, "ffq_hip.hip"), "-lz"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("the library does not build with -fsanitize=%s here: %s" % (san, r.stderr[-300:]))
    obj = d / "oracle_san.o"
    subprocess.run([CLANG.replace("clang++", "clang"), "-O1", "-g", "-std=c99", "-D_GNU_SOURCE", "-fPIC", "-fsanitize=" + san, "-c",
                    os.path.join(ROOT, "oracle", "ffq_oracle.c"), "-o", str(obj)], check=True)
    exe = d / "san_drv"
    r = subprocess.run([CLANG, "-fsanitize=" + san, "-shared-libsan", "-O1", "-g", "-std=c++17", "-pthread",
                        os.path.join(ROOT, "tests", "tsan_host_threads.cpp"), str(obj), "-L" + str(d), "-lffq_tsan", "-lz",
                        "-Wl,-rpath," + str(d), "-Wl,-rpath," + rt, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    def run(*args, env=None, ok=(0,)):
        e = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0", ASAN_OPTIONS="detect_leaks=0 exitcode=66", UBSAN_OPTIONS="halt_on_error=1 exitcode=66",
                 LD_LIBRARY_PATH=rt + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        e.update(env or {})
        p = subprocess.run([str(exe)] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=900)
        assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
        assert p.returncode in ok, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
        return p
    run.exe, run.rt, run.san = str(exe), rt, san
    return run


def _fastq(n, seed):
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    return synth.wrapped(seed, n, seed=43)[0].tobytes()


def test_detector_is_awake(tsan):
    if tsan.san != "thread":
        pytest.skip("the deliberate race is ThreadSanitizer's to find")
    e = dict(os.environ, TSAN_OPTIONS="exitcode=66", LD_LIBRARY_PATH=tsan.rt)
    p = subprocess.run([tsan.exe, "race", "x"], env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode == 66 and "ThreadSanitizer: data race" in p.stderr


def test_gunzip_threads_under_tsan(tsan, tmp_path):
    from fastqandfurious_amd import bgzf
    data = _fastq(30000, 5)                                     # 11 MB
    want = "%d %08x" % (len(data), zlib.crc32(data))
    plain = tmp_path / "one.gz"
    plain.write_bytes(gzip.compress(data, 6))
    bg = tmp_path / "blocks.bgz"
    bg.write_bytes(bgzf.compress(data, level=1))
    cat = tmp_path / "cat.gz"
    cat.write_bytes(gzip.compress(data[:3_000_000], 1) + gzip.compress(data[3_000_000:], 9) + b"\0" * 100)
    # one plain member: small chunks (many hand-overs at odd bits), default chunks, the forced hand-over to zlib
    for env, threads in (({"FFQ_PGZ_MIN": "1", "FFQ_PGZ_CHUNK": "32768"}, 6), ({"FFQ_PGZ_MIN": "1", "FFQ_PGZ_CHUNK": "262144"}, 3),
                         ({"FFQ_PGZ_MIN": "1", "FFQ_PGZ_CHUNK": "65536", "FFQ_PGZ_GIVEUP_AFTER": "2"}, 4), ({}, 8)):
        out = tsan("gunzip", plain, threads, 1 << 20, env=env).stdout.split()
        assert " ".join(out[:2]) == want
        if "FFQ_PGZ_CHUNK" in env:
            assert int(out[3]) > 4, "the several-thread engine did not run"
    out = tsan("gunzip", bg, 6, 1 << 20).stdout.split()
    assert " ".join(out[:2]) == want and int(out[2]) > 100          # members inflated side by side
    out = tsan("gunzip", cat, 5, 300000, env={"FFQ_PGZ_MIN": "1", "FFQ_PGZ_CHUNK": "65536"}).stdout.split()
    assert " ".join(out[:2]) == want
    # corrupt and cut files: an error (exit 1) or the good bytes, never a report
    blob = plain.read_bytes()
    random.seed(11)
    bad = tmp_path / "bad.gz"
    for it in range(12):
        b = bytearray(blob)
        if it % 3 == 0:
            b[random.randrange(20, len(b))] ^= 1 << random.randrange(8)
        elif it % 3 == 1:
            b = b[:random.randrange(20, len(b))]
        else:
            at = random.randrange(20, len(b))
            b[at:at + 2000] = bytes(random.randrange(256) for _ in range(2000))
        bad.write_bytes(bytes(b))
        tsan("gunzip", bad, 4, 1 << 20, env={"FFQ_PGZ_MIN": "1", "FFQ_PGZ_CHUNK": "65536"}, ok=(0, 1))


def test_reader_pool_under_tsan(tsan, tmp_path):
    f = tmp_path / "x.fq"
    f.write_bytes(_fastq(40000, 9))
    assert tsan("pool", f).stdout.startswith("equal")


def test_shard_world_and_host_step_under_tsan(tsan, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sharded import make_stream
    for kind, worlds in (("wrapped", (2, 3, 8)), ("tricky", (2, 8)), ("long-wrapped", (3,)), ("invalid", (3,))):
        f = tmp_path / ("%s.fq" % kind)
        f.write_bytes(make_stream(kind).tobytes())
        for w in worlds:
            assert tsan("shards", f, w, 1 << 20, 1 << 20).stdout.startswith("equal")
            assert tsan("shards", f, w, 100, 64).stdout.startswith("equal")
    # a rank that fails breaks the barrier: nobody is left waiting
    f = tmp_path / "wrapped.fq"
    for w in (2, 5):
        assert ("%d of %d ranks" % (w, w)) in tsan("abort", f, w).stdout


def test_a_rank_that_never_arrives_trips_the_deadline_on_every_rank(tsan, tmp_path):
    """The watchdog of the in-process world (csrc/ffq_shard_proto.h: ffq_shard_world::wait_for): rank 1 of k never enters
    its step; the first rank whose wait runs out breaks the barrier for everybody, EVERY other rank's host step comes back
    with FFQ_E_TIMEOUT within the deadline (not at the test's timeout), and the world names rank 1 as the absent one."""
    import sys
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sharded import make_stream
    f = tmp_path / "wrapped.fq"
    f.write_bytes(make_stream("wrapped").tobytes())
    for w in (2, 3, 8):
        t0 = time.perf_counter()
        out = tsan("stall", f, w, 1.5).stdout
        assert ("%d of %d ranks came back with FFQ_E_TIMEOUT; absent: 1" % (w - 1, w - 1)) in out, out
        assert time.perf_counter() - t0 < 60
