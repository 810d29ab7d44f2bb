"""The drop-in iterator's rate on one host core, outside bench.py: readfastq_iter over a plain file in /dev/shm with the
default entryfunc and with entryfunc_phred; WITH_TORCH=1 imports torch first (its objects make every full collection of the
cycle collector slower: DESIGN_LOG.md).  tools/iter_rate.py [records]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if os.environ.get("WITH_TORCH") == "1":
    import torch
import fastqandfurious_amd
from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800000
p = "/dev/shm/ffq_iter_rate_%d.fq" % os.getpid()
open(p, "wb").write(bytes(synth.single(0, n, seed=1)))
def run(ef):
    t = time.perf_counter(); k = 0
    with open(p, "rb") as fh:
        for e in F.readfastq_iter(fh, 50000, ef, C.entrypos):
            k += 1
    return k / (time.perf_counter() - t) / 1e6
try:
    for _ in range(3):
        print("torch=%s records=%d: entryfunc %.2f M reads/s  entryfunc_phred %.2f  entryfunc_namedtuple %.2f" %
              (os.environ.get("WITH_TORCH", "0"), n, run(F.entryfunc), run(F.entryfunc_phred), run(F.entryfunc_namedtuple)), flush=True)
finally:
    os.unlink(p)
