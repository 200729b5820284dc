#!/bin/bash
# Round-6 evidence beyond tools/collect_round.sh (run from the repo root through gpurun): the speaker-encoder workload (bench lines,
# rocprof kernel statistics + PMC passes, harness variants), the chain weight-gradient kernel alone and its stamps (ablation build),
# the stress grid, the training step's launch sequence at the encoder's size.
tools/collect_round.sh r6 > /dev/null 2>&1
O=gpurun_out/r6
timeout 600 python bench.py --workload spk --steps 20 --warmup 5 > $O/bench_spk.json 2> $O/bench_spk.err
timeout 600 python bench.py --workload spk --mode train --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_spk.json 2> $O/bench_train_spk.err
timeout 600 python bench.py --workload cfg1 --mode train --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_cfg1.json 2> $O/bench_train_cfg1.err
export TTRNN_PROFILE_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
tools/profile.sh spk --workload spk --steps 10 --warmup 2 > /dev/null 2>&1
tools/profile.sh train_spk --workload spk --mode train --steps 6 --warmup 2 > /dev/null 2>&1
export TTRNN_PROFILE_COUNTERS=""
for t in spk train_spk; do
  cp gpurun_out/prof_$t/summary.json $O/prof_${t}_summary.json 2>/dev/null
  cp gpurun_out/prof_$t/kernel_stats.csv $O/rocprof_kernel_stats_$t.csv 2>/dev/null
done
bash tools/variant_sweep.sh > $O/variants_benchmarking.txt 2>&1
# the chain weight gradient alone: the register hand-off kernel (default for the rank-2 encoder shapes), k_c2w on the same calls
# (dev2 bit 6), rank 4 (k_c2w's compile-time plans), k_c2w with its plan at run time (bits 6 + 2)
( python tools/c2w_bench.py 2 3 20; python tools/c2w_bench.py 2 2 20
  TTRNN_DEV2=64 python tools/c2w_bench.py 2 3 20; TTRNN_DEV2=64 python tools/c2w_bench.py 2 2 20
  python tools/c2w_bench.py 4 3 20; python tools/c2w_bench.py 4 2 20
  TTRNN_DEV2=68 python tools/c2w_bench.py 2 3 20 ) 2>&1 | grep "per call" > $O/c2w_bench.txt
if [ -f tools/bin/libttrnn_abl.so ]; then
  ( export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_abl.so
    for m in 3 2; do for bits in 0 1 2 3 4 8 16; do TTRNN_DEV2=$((65536*bits)) python tools/c2w_bench.py 2 $m 10 2>&1 | grep "per call\|cycles per block"; done; done ) > $O/stamps_c2r.txt 2>&1
  ( export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_abl.so
    for m in 3 2; do for bits in 0 1 2 3 4 8; do TTRNN_DEV2=$((65536*bits+64)) python tools/c2w_bench.py 2 $m 10 2>&1 | grep "per call\|cycles per block"; done; done ) > $O/stamps_c2w.txt 2>&1
fi
python tools/stress_backward.py --grid --reps 8 > $O/stress_backward_grid.txt 2>&1
python tools/stress_backward.py --reps 40 > $O/stress_backward.txt 2>&1
tools/launch_sequence.sh cfg2_train --workload cfg2 --mode train > /dev/null 2>&1; cp gpurun_out/seq_cfg2_train.txt $O/seq_cfg2_train.txt
tools/launch_sequence.sh spk_train --workload spk --mode train > /dev/null 2>&1; cp gpurun_out/seq_spk_train.txt $O/seq_spk_train.txt
DIAG_B=64 python tools/diag_stamps_bwd.py 2>&1 | tail -10 > $O/stamps_cfg2_f10bh.txt
python tools/diag_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps_cfg2_f10q.txt
ls $O | wc -l
