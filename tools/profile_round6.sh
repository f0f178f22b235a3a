#!/bin/bash
# Round-6 profiling session on the GPU box (run through gpurun): kernel statistics of the bench command (default: precision f16s-g2, 2 streams,
# launch plans), PMC passes (HBM traffic per kernel; MFMA / LDS / wave-state counters) of a shortened single-stream run, the HBM traffic of ONE
# steady-state W+ step in BOTH stream configurations (1 and 2 streams: bench.py sets the PMC bytes against a wall time of the same configuration),
# and the M2 leg.  Counters are collected in their own passes (--pmc with --kernel-trace only).  Output: gpurun_out/$1/
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LEGS="--no-cpu-baseline --no-modconv --no-single-stream --no-end-to-end --no-forward-only --no-generator-fwd --no-b1 --no-lpips --no-precision-ab"
SHORT="--wsteps 10 --steps 1 --warmup 1 $LEGS --streams 1 --no-plan"
rocprofv3 --kernel-trace -d $OUT/stats_s1 -o k -- python3 bench.py --streams 1 $LEGS > $OUT/bench_s1.json 2> $OUT/bench_s1.err
rocprofv3 --kernel-trace -d $OUT/stats_s2 -o k -- python3 bench.py $LEGS > $OUT/bench_sN.json 2> $OUT/bench_sN.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o p -- python3 bench.py $SHORT > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d $OUT/pmc_sq2 -o p -- python3 bench.py $SHORT > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o p -- python3 bench.py $SHORT > /dev/null 2> $OUT/pmc_grbm.err
for st in 1 2; do
  export OODGAN_STREAMS=$st
  for n in 10 30; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/step_fetch_s${st}_$n -o p -- python3 tools/wplus_only.py $n > $OUT/wplus_only_s${st}_$n.txt 2> $OUT/step_fetch_s${st}_$n.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/step_write_s${st}_$n -o p -- python3 tools/wplus_only.py $n > /dev/null 2> $OUT/step_write_s${st}_$n.err
  done
done
unset OODGAN_STREAMS
rocprofv3 --kernel-trace -d $OUT/m2_stats -o k -- python3 tools/m2_and_copy.py > $OUT/m2.json 2> $OUT/m2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/m2_fetch -o p -- python3 tools/m2_and_copy.py > /dev/null 2>> $OUT/m2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/m2_write -o p -- python3 tools/m2_and_copy.py > /dev/null 2>> $OUT/m2.err
python3 tools/rocpd_stats.py $OUT/stats_s1/k_results.db --csv $OUT/kernel_stats_streams1.csv
python3 tools/rocpd_stats.py $OUT/stats_s1/k_results.db --per-grid --csv $OUT/kernel_stats_streams1_per_grid.csv
python3 tools/rocpd_stats.py $OUT/stats_s2/k_results.db --csv $OUT/kernel_stats_streams2.csv
python3 tools/rocpd_stats.py $OUT/m2_stats/k_results.db --csv $OUT/m2_kernel_stats.csv
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write --csv $OUT/pmc_fetch_write.csv
python3 tools/pmc_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm --csv $OUT/pmc_sq.csv
python3 tools/pmc_summary.py $OUT/m2_fetch $OUT/m2_write --csv $OUT/m2_pmc_fetch_write.csv
for st in 1 2; do
  for n in 10 30; do python3 tools/pmc_summary.py $OUT/step_fetch_s${st}_$n $OUT/step_write_s${st}_$n --csv $OUT/step_pmc_s${st}_$n.csv; done
  python3 tools/step_traffic.py $OUT/step_pmc_s${st}_10.csv 10 $OUT/step_pmc_s${st}_30.csv 30 > $OUT/step_traffic_streams$st.json
done
python3 tools/kernel_table.py $OUT/kernel_stats_streams1.csv $OUT/pmc_fetch_write.csv $OUT/pmc_sq.csv --csv $OUT/kernel_counters.csv --top 30
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
ls -la $OUT
