#!/bin/bash
# tools/gpu.sh [--timeout S] -- '<command>' : gpurun with the commit id shipped beside the tree
# (the box gets no .git; profiles record the commit they were taken at from .commit)
R=$(cd "$(dirname "$0")/.." && pwd)
sha=$(git -C $R rev-parse HEAD)
git -C $R diff --quiet HEAD -- . ':!gpurun_out' || sha="$sha-dirty"
echo $sha > $R/.commit
exec /usr/local/graft/bin/gpurun "$@"
