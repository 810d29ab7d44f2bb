"""Why is ffq_load_fd fast into some device buffers and slow into others?  (round 6)
For several simultaneous allocations of 2 GiB: the loader's rate, the raw pinned copy's rate into the SAME buffer (one
stream), and the loader into the same allocation at other offsets."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip
ctx = hip.Context(0)
L = hip.lib()
n = 2 << 30
path = "/dev/shm/ffq_load_probe.bin"
blk = np.random.default_rng(1).integers(0, 255, 64 << 20, dtype=np.uint8).tobytes()
with open(path, "wb") as fh:
    for _ in range(n // len(blk)):
        fh.write(blk)
fd = os.open(path, os.O_RDONLY)
hp = ctypes.c_void_p()
hip.check(L.ffq_pinned_alloc(n, ctypes.byref(hp)))
ctypes.memset(hp, 1, n)


def load(d, m=n):
    best = 0
    for _ in range(2):
        t0 = time.perf_counter(); got = ctx.load_fd(fd, 0, m, d); ctx.sync(); best = max(best, m / (time.perf_counter() - t0) / 1e9)
    return best


def raw(d, m=n):
    best = 0
    for _ in range(2):
        t0 = time.perf_counter(); hip.check(L.ffq_copy_h2d(ctx.handle, ctypes.c_void_p(d), hp, m, 0)); best = max(best, m / (time.perf_counter() - t0) / 1e9)
    return best


def loads(d, k=5, m=n):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); ctx.load_fd(fd, 0, m, d); ctx.sync(); out.append(m / (time.perf_counter() - t0) / 1e9)
    return " ".join("%.1f" % x for x in out)


d = ctx.dev_alloc(n + 64)
print("first buffer of the process, five loads in a row:", loads(d), flush=True)
print("  after 0.3 s of nothing:", (time.sleep(0.3), loads(d))[1], flush=True)
ctx.dev_free(d)
d = ctx.dev_alloc(n + 64)
print("freed, allocated again (same size), five loads in a row:", loads(d), flush=True)
print("  after 0.3 s of nothing:", (time.sleep(0.3), loads(d))[1], flush=True)
print("  after 2 s of nothing:", (time.sleep(2.0), loads(d))[1], flush=True)
e = ctx.dev_alloc(n + 64)
print("a second buffer beside it (the first still alive), five loads:", loads(e), flush=True)
print("  the first one again:", loads(d), flush=True)
ctx.dev_free(d)
print("  the first one freed; the second, five loads:", loads(e), flush=True)
print("  after 0.3 s of nothing:", (time.sleep(0.3), loads(e))[1], flush=True)
ctx.dev_free(e)
ctx2 = hip.Context(0)
d = ctx2.dev_alloc(n + 64)
t = []
for _ in range(4):
    t0 = time.perf_counter(); ctx2.load_fd(fd, 0, n, d); ctx2.sync(); t.append(n / (time.perf_counter() - t0) / 1e9)
print("a NEW context (its own staging slots and helper threads), four loads:", " ".join("%.1f" % x for x in t), flush=True)
ctx2.dev_free(d)
os.close(fd); os.unlink(path)
