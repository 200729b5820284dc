mkdir -p gpurun_out/r1b
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r1b/bench_$w.json 2> gpurun_out/r1b/bench_$w.err
done
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $w --mode train --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r1b/bench_train_$w.json 2> gpurun_out/r1b/bench_train_$w.err
done
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT"
tools/profile.sh cfg4 --workload cfg4 --steps 6 --warmup 2 > /dev/null 2>&1
tools/profile.sh cfg5 --workload cfg5 --steps 4 --warmup 1 > /dev/null 2>&1
tools/profile.sh train_cfg4 --workload cfg4 --mode train --steps 4 --warmup 1 > /dev/null 2>&1
tools/profile.sh train_cfg5 --workload cfg5 --mode train --steps 3 --warmup 1 > /dev/null 2>&1
tools/profile.sh train_cfg2 --workload cfg2 --mode train --steps 6 --warmup 2 > /dev/null 2>&1
ls gpurun_out/r1b gpurun_out/prof_*
