#!/bin/bash
# arithmetic-only / store-only / normal variants of k_prospect (build_ab/x_{normal,nostore,storeonly}.so: python -c "import build; build.build(out=..., extra=['-DSPART_EXPERIMENT=1'])" for nostore, =2 for storeonly, no flag for normal)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/$1
for B in 10000 1000000; do for DT in float64 float32; do for v in normal nostore storeonly normal; do
  echo -n "$v: "; SPART_HIP_LIB=$ROOT/build_ab/x_$v.so timeout -k 10 120 python3 $ROOT/tools/prospect_bench.py $B $DT $([ $B = 10000 ] && echo 200 || echo 10) | tail -1
done; done; done | tee $ROOT/gpurun_out/$1/prospect_split.txt
