#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_build.sh <git-ref-for-A>
#   builds csrc as of <ref> into gpurun_ab/libffq_hip_A.so (B = the working tree's in-tree build);
#   then on the GPU box:  tools/ab_run.sh <workload> [reps]
R=$(cd "$(dirname "$0")/.." && pwd)
ref=${1:-HEAD}
rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src $R/gpurun_ab
git -C $R archive $ref fastq-and-furious_amd/csrc include | tar -x -C /tmp/ab_src
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
    -o $R/gpurun_ab/libffq_hip_A.so /tmp/ab_src/fastq-and-furious_amd/csrc/ffq_hip.hip -lz && echo "A = $ref -> gpurun_ab/libffq_hip_A.so"
