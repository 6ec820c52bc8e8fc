set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/mat_fetch -- python3 $R/tools/mode_run.py materialized 3 > $O/mat_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/mat_write -- python3 $R/tools/mode_run.py materialized 3 > $O/mat_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mat_sq -- python3 $R/tools/mode_run.py materialized 3 > $O/mat_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/mat_sq2 -- python3 $R/tools/mode_run.py materialized 3 > $O/mat_sq2.log 2>&1
echo done
