#!/bin/bash
# Package power / shader clock while k_prospect runs in a loop:  tools/power_probe.sh TAG
# (normal / arithmetic-only / store-only builds of build_ab/x_*.so; tools/prospect_split.sh builds nothing)
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
for DT in float64 float32; do for v in normal nostore storeonly; do
  SPART_HIP_LIB=$ROOT/build_ab/x_$v.so python3 $ROOT/tools/prospect_bench.py 1000000 $DT 900 > $O/power_prospect_${DT}_$v.run 2>&1 &
  PID=$!
  sleep 7
  echo "== k_prospect<$DT> 1M leaves, $v"
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 0.5; done
  kill $PID 2>/dev/null; wait $PID 2>/dev/null
done; done 2>&1 | tee $O/power_prospect.txt
