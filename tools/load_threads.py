"""The file loader by number of helper threads (FFQ_POOL_THREADS) and with its halves switched off (FFQ_LOAD_ABLATE, the
instrumented build): which side bounds ffq_load_fd on THIS box -- the preads out of the page cache or the link.
tools/load_threads.py            (the sweep: one process per setting)
tools/load_threads.py one        (one setting from the environment: six loads of 1 GiB from /dev/shm)"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
PATH = "/dev/shm/ffq_load_threads.bin"
N = 1 << 30


def one():
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import hip
    if os.environ.get("FFQ_LOAD_ABLATE"):
        hip.use_probe_build()
    ctx = hip.Context(0)
    fd = os.open(PATH, os.O_RDONLY)
    d = ctx.dev_alloc(N + 64)
    out = []
    for _ in range(6):
        t0 = time.perf_counter(); ctx.load_fd(fd, 0, N, d); ctx.sync(); out.append(N / (time.perf_counter() - t0) / 1e9)
    print(" ".join("%5.1f" % x for x in out), flush=True)
    os.close(fd)
    os._exit(0)


if len(sys.argv) > 1:
    one()
import numpy as np
blk = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8).tobytes()
with open(PATH, "wb") as fh:
    for _ in range(N // len(blk)):
        fh.write(blk)
print("host cores", os.cpu_count(), flush=True)
for threads in (4, 8, 12, 16, 24, 32, 48, 64):
    for ab, what in (("", "loader"), ("1", "reads only (no copies)"), ("2", "copies only (no reads)")):
        if ab and threads not in (16, 32):
            continue
        env = dict(os.environ, FFQ_POOL_THREADS=str(threads))
        if ab:
            env["FFQ_LOAD_ABLATE"] = ab
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=300)
        print("threads %2d %-24s GB/s: %s" % (threads, what, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
os.unlink(PATH)
