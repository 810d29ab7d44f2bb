"""Do streams, inflate pools and contexts give back what they take?  Threads, descriptors and resident memory over many
open / close cycles (a trend that flattens is the allocator's and the HIP runtime's caches filling; a line is a leak)."""
import os, sys, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth, bgzf, fastqandfurious as F, _fastqandfurious as C
import psutil
p = psutil.Process()
data = synth.single(0, 20000).tobytes()
open("/tmp/l.bgz", "wb").write(bgzf.compress(data, level=1))
open("/tmp/l.gz", "wb").write(gzip.compress(data, 1))
open("/tmp/l.fq", "wb").write(data)
ctx = hip.default_context()


def once(path, gz):
    fd = os.open(path, os.O_RDONLY)
    st = hip.FileStream(ctx, fd, 1 << 20, gzip=gz)
    n = sum(r.shape[0] for r, *_ in st)
    st.close(); os.close(fd)
    return n


def show(tag):
    print("%-28s threads %3d  fds %3d  rss %5d MiB" % (tag, p.num_threads(), p.num_fds(), p.memory_info().rss >> 20), flush=True)


for i in range(3):
    once("/tmp/l.bgz", True)
show("start")
for rnd in range(5):
    for i in range(200):
        assert once("/tmp/l.bgz", True) == 20000
        assert once("/tmp/l.gz", True) == 20000
        assert once("/tmp/l.fq", False) == 20000
    fd = os.open("/tmp/l.bgz", os.O_RDONLY); st = hip.FileStream(ctx, fd, 1 << 18, gzip=True); it = iter(st); next(it); st.close(); os.close(fd)
    with gzip.open("/tmp/l.bgz", "rb") as fh:
        for k, e in enumerate(F.readfastq_iter(fh, 50000, F.entryfunc, C.entrypos)):
            if k == 100:
                break
    show("%d streams" % (600 * (rnd + 1)))
arr = np.frombuffer(data, np.uint8)
for rnd in range(5):
    for i in range(100):
        c2 = hip.Context(0); c2.scan_host(arr); c2.close()
    show("%d contexts" % (100 * (rnd + 1)))
