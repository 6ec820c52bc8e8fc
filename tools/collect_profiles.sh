#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r1_d'
# Writes under gpurun_out/<tag>/ ; tools/ingest_profiles.py copies the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r1_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 python $R/bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-rows 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 > $O/pmc_sq.log 2>&1
# materialised-spectra mode (the HBM-store-bound variant of the band kernel), B = 200k, dense vs padded row pitch
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mat_stats -- python3 $R/tools/mat_bench.py > $O/mat_stats.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/mat_pmc_write -- python3 $R/tools/mat_bench.py > $O/mat_pmc_write.log 2>&1
cd $R
timeout -k 10 300 python tools/mat_bench.py > $O/mat_bench.txt 2>&1
timeout -k 10 400 python tools/measure_configs.py > $O/configs.json 2> $O/configs.err
echo done > $O/DONE
