#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r2_a'
# Writes under gpurun_out/<tag>/ ; tools/ingest_profiles.py <tag> copies the summaries into profiles/.
# Counter passes are their own runs (--kernel-trace + --pmc only), FETCH_SIZE and WRITE_SIZE apart (TCC slots).
set -e -o pipefail
TAG=${1:-r2_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout -k 10 400 python $R/bench.py > $O/bench.json 2> $O/bench.err
echo bench done
cd /tmp && export TMPDIR=/tmp
for DT in float32 float64; do
  ST=10; [ $DT = float64 ] && ST=5
  A="--dtype $DT --cpu-rows 0 --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$DT -- python3 $R/bench.py --steps $ST --warmup 2 $A > $O/stats_$DT.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$DT -- python3 $R/bench.py --steps 3 --warmup 1 $A > $O/pmc_fetch_$DT.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$DT -- python3 $R/bench.py --steps 3 --warmup 1 $A > $O/pmc_write_$DT.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq_$DT -- python3 $R/bench.py --steps 3 --warmup 1 $A > $O/pmc_sq_$DT.log 2>&1
  echo $DT done
done
# BASELINE config 2: the PROSPECT-only kernel at 10k x 2001, float64
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -- python3 $R/tools/prospect_bench.py > $O/c2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c2_pmc_fetch -- python3 $R/tools/prospect_bench.py 10000 float64 5 > $O/c2_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c2_pmc_write -- python3 $R/tools/prospect_bench.py 10000 float64 5 > $O/c2_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/c2_pmc_sq -- python3 $R/tools/prospect_bench.py 10000 float64 5 > $O/c2_pmc_sq.log 2>&1
echo config2 done
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts in this library's access widths
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/ubench/fetch_calib > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/tools/ubench/fetch_calib > $O/calib_write.log 2>&1
cd $R
timeout -k 10 120 python tools/prospect_bench.py > $O/c2_bench.txt 2>&1
timeout -k 10 120 python tools/prospect_bench.py 1000000 float64 5 >> $O/c2_bench.txt 2>&1
timeout -k 10 120 python tools/prospect_bench.py 1000000 float32 5 >> $O/c2_bench.txt 2>&1
timeout -k 10 300 python tools/mode_cost.py > $O/mode_cost.txt 2>&1
timeout -k 10 300 python tools/lut_rate.py > $O/lut_rate.txt 2>&1
echo done > $O/DONE
