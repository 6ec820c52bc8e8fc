#!/bin/bash
# The driver-style line AFTER the round's counters have been ingested (tools/ingest_profiles.py TAG):
#   gpurun --timeout 900 -- 'bash tools/bench_after_ingest.sh r4_a'   ->  gpurun_out/TAG/bench.json (copy to profiles/TAG_bench.json)
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 800 python $R/bench.py > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
tail -c 600 $R/gpurun_out/$TAG/bench.json; tail -3 $R/gpurun_out/$TAG/bench.err
# HIP float64 / float32 columns against the oracle on 2 x 262 144 rows (the driver-visible version of this is
# tests/test_gpu_parity.py::test_parity_at_scale_against_the_oracle, 2 x 65 536 rows)
timeout -k 10 900 python $R/tools/big_parity.py 262144 > $R/gpurun_out/$TAG/parity_262144rows.json 2> $R/gpurun_out/$TAG/parity.err
tail -2 $R/gpurun_out/$TAG/parity.err
