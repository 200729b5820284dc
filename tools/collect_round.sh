#!/bin/bash
# Round evidence on one MI355X box (run from the repo root through gpurun): bench lines of every workload in both modes,
# the grid sweep, and rocprofv3 kernel stats + PMC passes.  Everything lands under gpurun_out/<tag>/; copy what is worth
# keeping into profiles/<round>/.
#   tools/collect_round.sh r2
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for w in cfg2 cfg1 cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --mode train --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err
done
timeout 600 python bench.py --workload grid > $OUT/bench_grid.json 2> $OUT/bench_grid.err
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
tools/profile.sh cfg2 --workload cfg2 --steps 20 --warmup 3 > /dev/null 2>&1
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES"
tools/profile.sh cfg3 --workload cfg3 --steps 10 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg4 --workload cfg4 --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg5 --workload cfg5 --steps 4 --warmup 1 > /dev/null 2>&1
export TTRNN_PROFILE_COUNTERS=""
tools/profile.sh train_cfg2 --workload cfg2 --mode train --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh train_cfg4 --workload cfg4 --mode train --steps 4 --warmup 1 > /dev/null 2>&1
tools/profile.sh train_cfg5 --workload cfg5 --mode train --steps 3 --warmup 1 > /dev/null 2>&1
for t in cfg2 cfg3 cfg4 cfg5 train_cfg2 train_cfg4 train_cfg5; do
  cp gpurun_out/prof_$t/summary.json $OUT/prof_${t}_summary.json 2>/dev/null
  cp gpurun_out/prof_$t/kernel_stats.csv $OUT/rocprof_kernel_stats_$t.csv 2>/dev/null
done
ls $OUT
