#!/bin/bash
# tools/chunk_sweep.sh TAG B DTYPE chunk...   -- prospect_bench.py under SPART_CHUNK (the band kernels' samples per workgroup)
TAG=$1; B=$2; DT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/$TAG
for c in "$@"; do
  echo -n "chunk $c: "
  SPART_CHUNK=$c timeout -k 10 120 python3 $ROOT/tools/prospect_bench.py $B $DT 200 | tail -1
done | tee $ROOT/gpurun_out/$TAG/chunk_sweep_${B}_${DT}.txt
