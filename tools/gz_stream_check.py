"""A gzip file of 1 GiB of synthetic FASTQ through the stream front end (ffq_stream_open_gzip: the several-thread inflate
feeding the pinned chunks) against the same bytes as a plain file: record count, SHA-256 of all rows, seconds."""
import os, sys, time, zlib, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from fastqandfurious_amd import hip, sharded
ctx = hip.Context(0)
shard = sharded.SyntheticShard(ctx, "single", 1 << 30, 0, 1, torch.device("cuda:0"))
data = shard.host_sample(1 << 30)
print("bytes", data.size)
plain = "/dev/shm/big.fq"; gz = plain + ".gz"
data.tofile(plain)
t = time.time()
c = zlib.compressobj(1, zlib.DEFLATED, 31)
with open(gz, "wb") as f:
    for i in range(0, data.size, 64 << 20):
        f.write(c.compress(data[i:i + (64 << 20)].tobytes()))
    f.write(c.flush())
print("gz bytes", os.path.getsize(gz), "in %.1f s" % (time.time() - t))
def run(path, gzip):
    fd = os.open(path, os.O_RDONLY)
    t = time.time()
    st = hip.FileStream(ctx, fd, 1 << 26, gzip=gzip)
    h = hashlib.sha256(); n = 0
    for rows, _f, off, _e, _x in st:
        a = np.asarray(rows) + 0
        h.update(a.tobytes()); n += a.shape[0]
    st.close(); os.close(fd)
    return n, h.hexdigest(), time.time() - t
s0 = hip.gunzip_stats()
a = run(plain, False); b = run(gz, True)
print("plain", a[0], "%.2f s" % a[2]); print("gzip ", b[0], "%.2f s" % b[2], "%.2f GB/s" % (data.size / b[2] / 1e9))
print("engine", {k: v - s0[k] for k, v in hip.gunzip_stats().items()})
print("EQUAL" if a[:2] == b[:2] else "DIFFERENT")
os.unlink(plain); os.unlink(gz)
