#!/bin/bash
# rocprofv3 kernel statistics of the LUT inversion kernels (tools/lut_profile_run.py):  tools/lut_profile.sh TAG [nb] [dtype]
# writes gpurun_out/TAG/lut_{uniform,correlated}_kernel_stats.csv
set -e
TAG=${1:?tag}; NB=${2:-13}; DT=${3:-float32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for k in uniform correlated; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lut_$k" -o p -- python3 "$ROOT/tools/lut_profile_run.py" $k $NB $DT > "$OUT/prof_lut_$k.log" 2>&1
  f=$(find "$OUT/prof_lut_$k" -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/lut_${k}_nb${NB}_${DT}_kernel_stats.csv"; grep lut "$f" | cut -d, -f1-8; fi
  tail -1 "$OUT/prof_lut_$k.log"
  rm -rf "$OUT/prof_lut_$k"
done
