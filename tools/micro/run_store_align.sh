#!/bin/bash
# usage (on the box): bash tools/micro/run_store_align.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/store_align tools/micro/store_align.hip 2>&1 | grep -v warning | tail -2
/tmp/store_align 4096 | tee gpurun_out/store_align.txt
