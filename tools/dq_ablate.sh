#!/bin/bash
export FFQ_USE_PROBE_BUILD=1   # the ablation switches exist only in libffq_probe.so
# ablation of k_decode_stream: bit 1 no search, 2 no stores, 4 no source loads, 8 no straddle loop
R=$(cd "$(dirname "$0")/.." && pwd)
for a in 0 1 2 4 8 6 7 15; do
  echo "ablate $a"; FFQ_DQ_ABLATE=$a bash $R/tools/kstats.sh 4e9 single decode | grep decode_stream
done
