#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r5_final'
# Writes under gpurun_out/<tag>/ ; tools/ingest_profiles.py <tag> copies the summaries into profiles/.
# Counter passes are their own runs (--kernel-trace + --pmc only), FETCH_SIZE and WRITE_SIZE apart (TCC slots).
set -e -o pipefail
TAG=${1:-r2_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
# (the driver-style bench line is NOT taken here: bench.py reads the committed counter files of THIS collection, so it is run
#  after tools/ingest_profiles.py -- tools/bench_after_ingest.sh -- and its roofline.issue.profiled_sources_match is true)
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
# the modes beside the headline (bench.py configs.materialized / lut_invert / pruned): stats + counter passes over tools/mode_run.py
MFMA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for M in materialized lut_invert pruned; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${M}_stats -- python3 $R/tools/mode_run.py $M 10 > $O/${M}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${M}_pmc_fetch -- python3 $R/tools/mode_run.py $M 3 > $O/${M}_pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${M}_pmc_write -- python3 $R/tools/mode_run.py $M 3 > $O/${M}_pmc_write.log 2>&1
  if [ $M = lut_invert ]; then C="$MFMA"; else C="$SQ"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${M}_pmc_sq -- python3 $R/tools/mode_run.py $M 3 > $O/${M}_pmc_sq.log 2>&1
  echo $M done
done
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts in this library's access widths
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/ubench/fetch_calib > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/tools/ubench/fetch_calib > $O/calib_write.log 2>&1
cd $R
timeout -k 10 120 python tools/prospect_bench.py > $O/c2_bench.txt 2>&1
timeout -k 10 120 python tools/prospect_bench.py 1000000 float64 5 >> $O/c2_bench.txt 2>&1
timeout -k 10 120 python tools/prospect_bench.py 1000000 float32 5 >> $O/c2_bench.txt 2>&1
timeout -k 10 300 python tools/mode_cost.py > $O/mode_cost.txt 2>&1
timeout -k 10 300 python tools/lut_rate.py > $O/lut_rate.txt 2>&1
timeout -k 10 300 python tools/lut_invert_rate.py > $O/lut_invert_rate.txt 2>&1
timeout -k 10 300 bash tools/lut_profile.sh $TAG > $O/lut_profile.txt 2>&1
timeout -k 10 300 python tools/mat_bench.py > $O/mat_bench.txt 2>&1
timeout -k 10 300 python tools/fast_prelude_dev.py > $O/fast_prelude_dev.txt 2>&1
# package power / clock while the materialised mode runs (it is power-bound: DESIGN.md section 4)
python3 $R/tools/mode_run.py materialized 7000 > $O/power_materialized.run 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk"; sleep 0.5; done > $O/power_materialized.txt 2>&1
wait $PID
python3 $R/tools/mode_run.py headline 1200 > $O/power_headline.run 2>&1 &
PID=$!
sleep 10
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk"; sleep 0.5; done > $O/power_headline.txt 2>&1
wait $PID
rocm-smi --showmaxpower 2>&1 | grep -i "power" > $O/power_cap.txt
echo done > $O/DONE
