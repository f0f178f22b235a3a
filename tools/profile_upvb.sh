#!/bin/bash
# PMC counters of the three forms of conv_f16s_upvb (tools/bench_upvb.py runs them back to back) -> gpurun_out/$1/
set -u
TAG=${1:-upvb_prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/p1 -o p -- python3 tools/bench_upvb.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 -d $OUT/p2 -o p -- python3 tools/bench_upvb.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/p3 -o p -- python3 tools/bench_upvb.py > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p4 -o p -- python3 tools/bench_upvb.py > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p5 -o p -- python3 tools/bench_upvb.py > $OUT/p5.log 2>&1
python3 tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 --csv $OUT/pmc.csv
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
grep -E "upvb|t2v2|blur_act_fform" $OUT/pmc.csv
