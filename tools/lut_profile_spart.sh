#!/bin/bash
# rocprofv3 kernel statistics of spart_lut_nearest on bench.py's own workload (tools/mode_run.py lut_invert): tools/lut_profile_spart.sh TAG
set -e
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lut_spart" -o p -- python3 "$ROOT/tools/mode_run.py" lut_invert 8 > "$OUT/prof_lut_spart.log" 2>&1
f=$(find "$OUT/prof_lut_spart" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/lut_spart_kernel_stats.csv"; grep lut "$f" | cut -d'"' -f2,3 | sed 's/(.*)"//' ; fi
grep "lut_invert" "$OUT/prof_lut_spart.log" || true
rm -rf "$OUT/prof_lut_spart"
