#!/bin/bash
# Timing sweep over the variants the reference's benchmarking.py exposes (its own defaults: TT-LSTM in=256 H=512 d=3 r=8, batch 512,
# 160 steps) — eval and train — to catch configurations that fall off the fast routes.   tools/variant_sweep.sh > gpurun_out/variants.txt
export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
run() { echo "== $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
for mode in "" "--train"; do
run $mode
run $mode --gru
run $mode --naive_tt
run $mode --ttrank 16
run $mode --naive_tt --gru
run $mode --ttrank 16 --gru
run $mode --hidden_size 256 --gru --ttrank 16
run $mode --in_size 1 --hidden_size 256 --seq_len 784 --batch_size 64 --gru
run $mode --in_size 1 --hidden_size 256 --seq_len 784 --batch_size 64 --naive_tt
run $mode --in_size 1 --hidden_size 256 --seq_len 784 --batch_size 64 --naive_tt --gru
run $mode --in_size 40 --hidden_size 768 --ncores 4
run $mode --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
run $mode --in_size 40 --hidden_size 768 --ncores 2 --ttrank 4
run $mode --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
run $mode --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2
run $mode --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2 --gru
run $mode --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2 --gru
run $mode --n_layers 2 --hidden_size 384
done
