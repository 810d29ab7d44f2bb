#!/bin/bash
# Host-only scaling of the several-thread gzip inflate (csrc/ffq_pgz.h) through its harness: tools/pgz_bench.sh [records]
R=$(cd "$(dirname "$0")/.." && pwd)
n=${1:-2000000}
mkdir -p /tmp/pgzb && cd /tmp/pgzb
g++ -O2 -std=c++17 -pthread -o pgz_main $R/tools/pgz_main.cpp -lz || exit 1
python - $n <<'PY'
import sys, numpy as np
n = int(sys.argv[1]); L = 150
rng = np.random.default_rng(1)
with open("a.fq", "wb") as f:
    for lo in range(0, n, 100000):
        m = min(100000, n - lo)
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(m, L))
        q = (np.clip(rng.normal(36, 4, size=(m, L)).astype(np.int64), 2, 41) + 33).astype(np.uint8)
        out = bytearray()
        for i in range(m):
            out += b"@SRR1234567.%d %d/1\n" % (lo + i + 1, lo + i + 1)
            out += seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n"
        f.write(out)
PY
ls -l a.fq
for l in 1 6; do
    gzip -$l -c a.fq > a$l.gz; ls -l a$l.gz
    ./pgz_main a$l.gz 4 check | tail -1
    for t in 1 2 4 8 16 32 64; do echo "level $l threads $t: $(./pgz_main a$l.gz $t | tr '\n' ' ')"; done
done
