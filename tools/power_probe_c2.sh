#!/bin/bash
# Package power / shader clock while BASELINE config 2 AS STATED (spart_prospect_batch, 10 000 leaves x 2001 bands, float64: one
# k_prelude + one k_prospect<double> launch per call) loops back to back, next to the same loop at 1M leaves and to the idle card:
#   tools/power_probe_c2.sh TAG          -> gpurun_out/TAG/power_config2.txt
# (the power reading of DESIGN.md section 4 was taken on the 1M loop.)
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
{
echo "== idle"
for i in 1 2; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 0.5; done
for CASE in "10000 float64 120000" "1000000 float64 1500" "10000 float32 200000"; do
  set -- $CASE
  python3 $ROOT/tools/prospect_bench.py $1 $2 $3 > $O/power_config2_$1_$2.run 2>&1 &
  PID=$!
  sleep 8
  echo "== spart_prospect_batch B=$1 $2, looping"
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 0.5; done
  wait $PID
  cat $O/power_config2_$1_$2.run | grep prospect
done
} 2>&1 | tee $O/power_config2.txt
