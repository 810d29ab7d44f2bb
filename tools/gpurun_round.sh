#!/bin/bash
# HERE (not on the box, which has no .git): note the commit the snapshot is taken from, then one box for tools/_run_round6.sh
# tools/gpurun_round.sh <tag> [timeout]
R=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:?tag}
git -C $R rev-parse HEAD > $R/.commit
if [ -n "$(git -C $R status --porcelain --untracked-files=no)" ]; then echo "+ uncommitted changes" >> $R/.commit; fi
exec /usr/local/graft/bin/gpurun --timeout ${2:-3300} -- "tools/_run_round6.sh $tag"
