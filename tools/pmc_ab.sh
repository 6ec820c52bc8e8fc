#!/bin/bash
# SQ counter comparison of two library builds (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash tools/pmc_ab.sh a=build_ab/base.so b=build_ab/new.so'
# k_bands dispatches alternate a, b, a, b (tools/pmc_two.py: fixed order); tools/pmc_ab_read.py prints the table.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_ab
rm -rf $O && mkdir -p $O
ARGS=()
for a in "$@"; do ARGS+=("${a%%=*}=$R/${a#*=}"); done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_IFETCH" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VSKIPPED SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/pmc_two.py "${ARGS[@]}" > $O/p$i.log 2>&1
  cp $(find $O/p$i -name "*counter_collection.csv" | head -1) $O/p$i.csv
  rm -rf $O/p$i
done
echo done
