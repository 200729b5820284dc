#!/bin/bash
# Round-4 evidence beyond tools/collect_round.sh (run from the repo root through gpurun): training lines the collection script does not take,
# the grid training step replayed from a hipGraph, the reference harness' variants, per-phase stamps of the reverse-kernel families
# (ablation build: make -C tensorized-rnn_amd/csrc ablation) and rocprof summaries of the reference's default benchmark step.
tools/collect_round.sh r4 > /dev/null 2>&1
O=gpurun_out/r4
timeout 600 python bench.py --workload cfg1 --mode train --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_cfg1.json 2> $O/bench_train_cfg1.err
timeout 900 python bench.py --workload grid --mode train --graph > $O/bench_grid_train_graph.json 2> $O/bench_grid_train_graph.err
bash tools/variant_sweep.sh > $O/variants_benchmarking.txt 2>&1
DIAG_H=512 DIAG_IN=256 DIAG_B=256 DIAG_T=160 python tools/diag_stamps_bwd.py 2>&1 | tail -10 > $O/stamps_h512_f10l.txt
DIAG_B=64 python tools/diag_stamps_bwd.py 2>&1 | tail -10 > $O/stamps_cfg2_f10bh.txt
python tools/diag_stamps_g2bwd.py --gru 2>&1 | grep -v "^wave [4-7] [0-9]\{10\}" | tail -11 > $O/stamps_g2bwd_gru_h512.txt
python tools/diag_stamps_g2bwd.py --in_size 40 --hidden_size 768 --ncores 4 2>&1 | grep -v "^wave [4-7] [0-9]\{10\}" | tail -7 > $O/stamps_g2bwd_h768_d4.txt
python tools/diag_stamps_g2bwd.py --naive_tt 2>&1 | grep -v "^wave [4-7] [0-9]\{10\}" | tail -7 > $O/stamps_g2bwd_naive_h512.txt
python tools/diag_stamps_g2fwd.py --gru --hidden_size 256 --ncores 2 2>&1 | tail -11 > $O/stamps_g2fwd_gru_h256_d2.txt
export TTRNN_PROFILE_COUNTERS="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES FETCH_SIZE"
tools/profile_harness.sh h512train --tt --train > /dev/null 2>&1; cp gpurun_out/profh_h512train/summary.json $O/profh_h512_train_summary.json
tools/profile_harness.sh grutrain --tt --train --gru > /dev/null 2>&1; cp gpurun_out/profh_grutrain/summary.json $O/profh_gru_h512_train_summary.json
SEQ_HARNESS=1 tools/launch_sequence.sh --tt --train > /dev/null 2>&1; cp gpurun_out/seq_--tt.txt $O/seq_h512_train.txt
ls $O | wc -l
