"""Host-side rates around the GPU scan: the native stream front end, build_index, scan_host."""
import io, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, index, synth, hip
n = (1 << 30) // 322
blob = synth.single(0, n, seed=42).tobytes()
ctx = hip.default_context(0)
d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
path = os.path.join(d, "ffq_rates.fq")
open(path, "wb").write(blob)
try:
    for bs in (1 << 22, 1 << 24, 1 << 26, 1 << 28):
        fd = os.open(path, os.O_RDONLY)
        t0 = time.perf_counter(); k = 0
        st = hip.FileStream(ctx, fd, bs)
        for rows, fill, off, end, err in st:
            k += rows.shape[0]
        st.close(); os.close(fd)
        el = time.perf_counter() - t0
        print("ffq_stream  fbufsize %4d MiB: %6.2f GB/s  (%d records, %.3f s)" % (bs >> 20, len(blob) / el / 1e9, k, el), flush=True)
    for bs in (1 << 24, 1 << 26):
        t0 = time.perf_counter()
        with open(path, "rb") as fh, open(os.path.join(d, "ffq_rates.idx"), "wb") as fi:
            k = index.build_index(fh, fi, bs)
        el = time.perf_counter() - t0
        print("build_index(file, native stream) fbufsize %4d MiB: %6.2f GB/s (%d records)" % (bs >> 20, len(blob) / el / 1e9, k), flush=True)
    t0 = time.perf_counter(); fi = io.BytesIO(); k = index.build_index(io.BytesIO(blob[:256 << 20] [:(256 << 20) // 322 * 322]), fi, 1 << 24); el = time.perf_counter() - t0
    print("build_index(BytesIO, Python loop)  fbufsize   16 MiB: %6.2f GB/s" % (((256 << 20) // 322 * 322) / el / 1e9))
    a = np.frombuffer(blob, dtype=np.uint8)
    ctx.scan_host(a)
    t0 = time.perf_counter(); t, res = ctx.scan_host(a); el = time.perf_counter() - t0
    print("scan_host 1 GiB (pageable in, table out): %6.2f GB/s" % (len(blob) / el / 1e9))
finally:
    for f in ("ffq_rates.fq", "ffq_rates.idx"):
        try: os.unlink(os.path.join(d, f))
        except OSError: pass
