set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_types
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS --output-format csv -d $O/p1 -- python3 $R/tools/ab_bench.py a=$R/build_ab/sweep4.so b=$R/build_ab/sweep4.so --rounds 1 > $O/p1.log 2>&1
cp $(find $O/p1 -name "*counter_collection.csv" | head -1) $O/p1.csv
rm -rf $O/p1
