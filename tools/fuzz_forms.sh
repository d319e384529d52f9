#!/bin/bash
# The differential fuzz (general + reader-block seeds: encode == oracle or the same error text, decode == oracle) under EVERY alternative formulation, on the GPU.
# usage (on the box): bash tools/fuzz_forms.sh [general seeds = 800] [block seeds = 200]
cd $GRAFT_REPO_ROOT; NG=${1:-800}; NB=${2:-200}
for sw in "" "RFQ_GATHER=old" "RFQ_QUAL=bytes" "RFQ_QUAL=bytes RFQ_CODER=list" "RFQ_QUAL=bytes RFQ_CODER=mask" "RFQ_INDEX=2pass" "RFQ_IDX_TILES=4" "RFQ_STREAMS=1" "RFQ_MATERIALISE=1" "RFQ_WALK=exact" "RFQ_SLICE_BYTES=4000000 RFQ_SLICE_BASES=1500000"; do
  env $sw timeout 600 python tools/fuzz_more.py $NG $NB 2>&1 | grep -v amdgpu.ids
done
