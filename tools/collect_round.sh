#!/bin/bash
# Round evidence on one MI355X box (run from the repo root through gpurun): bench lines of every workload in both modes,
# the grid sweep, the K-in GEMM harness, and rocprofv3 kernel stats + PMC passes.  Everything lands under gpurun_out/<tag>/;
# copy what is worth keeping into profiles/<round>/ (tools/update_traffic.py turns the PMC summaries into profiles/traffic.json).
#   tools/collect_round.sh r3 [quick]
TAG=${1:-r5}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for w in cfg2 cfg1 cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
# the reference's own GRU dtype (fp32) at cfg3's and cfg2's sizes (round 5: k_gru_fwd_f10vh)
for w in cfg3_fp32 gru64; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  timeout 600 python bench.py --workload $w --mode train --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err
done
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --mode train --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err
done
TTRNN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload cfg2 --mode train --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_train_cfg2_rccl1.json 2> $OUT/bench_train_cfg2_rccl1.err
[ -x tools/bin/gemm3_bench ] && tools/bin/gemm3_bench 131072 1024 4096 4 > $OUT/gemm3_cfg5.log 2>&1
if [ -z "$QUICK" ]; then
  timeout 900 python bench.py --workload grid > $OUT/bench_grid.json 2> $OUT/bench_grid.err
fi
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
tools/profile.sh cfg2 --workload cfg2 --steps 20 --warmup 3 > /dev/null 2>&1
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES"
tools/profile.sh cfg1 --workload cfg1 --steps 10 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg3 --workload cfg3 --steps 10 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg3_fp32 --workload cfg3_fp32 --steps 10 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg4 --workload cfg4 --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg5 --workload cfg5 --steps 4 --warmup 1 > /dev/null 2>&1
export TTRNN_PROFILE_COUNTERS=""
tools/profile.sh train_cfg2 --workload cfg2 --mode train --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh train_cfg3 --workload cfg3 --mode train --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh train_cfg4 --workload cfg4 --mode train --steps 4 --warmup 1 > /dev/null 2>&1
tools/profile.sh train_cfg5 --workload cfg5 --mode train --steps 3 --warmup 1 > /dev/null 2>&1
for t in cfg1 cfg2 cfg3 cfg3_fp32 cfg4 cfg5 train_cfg2 train_cfg3 train_cfg4 train_cfg5; do
  cp gpurun_out/prof_$t/summary.json $OUT/prof_${t}_summary.json 2>/dev/null
  cp gpurun_out/prof_$t/kernel_stats.csv $OUT/rocprof_kernel_stats_$t.csv 2>/dev/null
done
ls $OUT
