#!/bin/bash
# Round-5 evidence beyond tools/collect_round.sh (run from the repo root through gpurun): forward stamps of the headline kernel and of
# the fp32 GRU kernel, the A/B kernels built and rejected this round, the fused set-up launch against the round-4 library, the GPU
# timeline of the headline loop, the whole stress grid, the reference harness' variants, the training step's launch sequence.
tools/collect_round.sh r5 > /dev/null 2>&1
O=gpurun_out/r5
timeout 600 python bench.py --workload cfg1 --mode train --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_cfg1.json 2> $O/bench_train_cfg1.err
timeout 900 python bench.py --workload grid --mode train --graph > $O/bench_grid_train_graph.json 2> $O/bench_grid_train_graph.err
bash tools/variant_sweep.sh > $O/variants_benchmarking.txt 2>&1
python tools/diag_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps_cfg2_f10q.txt
DIAG_CELL=gru python tools/diag_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps_gru_f10vh.txt
DIAG_CELL=naive python tools/diag_stamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps_naive_f10n.txt
DIAG_B=64 python tools/diag_stamps_bwd.py 2>&1 | tail -10 > $O/stamps_cfg2_f10bh.txt
[ -f tools/bin/libttrnn_r4.so ] && python tools/ab_vs_r4.py 2>&1 | grep -v amdgpu.ids > $O/ab_vs_r4.txt
tools/gap_report.sh 2>&1 | tail -10 > $O/gap_report_cfg2.txt
python tools/diag_stamps_g2fwd.py --naive_tt --in_size 256 --hidden_size 512 --batch_size 512 2>&1 | grep -v amdgpu.ids > $O/stamps_naive_g2fwdp.txt
python tools/diag_stamps_g2bwd.py --naive_tt 2>&1 | grep -v amdgpu.ids > $O/stamps_naive_g2bwd.txt
python tools/diag_stamps_g2bwd.py --ttrank 16 2>&1 | grep -v amdgpu.ids > $O/stamps_r16_g2bwd.txt
tools/pair_ab.sh eval > $O/pair_ab.txt 2>&1
python tools/stress_backward.py --grid --reps 8 > $O/stress_backward_grid.txt 2>&1
python tools/stress_backward.py --reps 40 > $O/stress_backward.txt 2>&1
tools/launch_sequence.sh cfg2_train --workload cfg2 --mode train > /dev/null 2>&1; cp gpurun_out/seq_cfg2_train.txt $O/seq_cfg2_train.txt
tools/launch_sequence.sh gru64_train --workload gru64 --mode train > /dev/null 2>&1; cp gpurun_out/seq_gru64_train.txt $O/seq_gru64_train.txt
ls $O | wc -l
