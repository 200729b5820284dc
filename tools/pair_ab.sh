#!/bin/bash
# A/B of the runtime tier's paired forward kernel (dev bit 19 = one sample per workgroup) and timing of the training steps that run the
# tier's reverse kernel, with the reference harness' semantics.   tools/pair_ab.sh [eval|train|all]
export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
run() { echo "== ${TTRNN_DEV:+dev=$TTRNN_DEV }$*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
what=${1:-all}
if [ $what != train ]; then
for f in "--naive_tt" "--naive_tt --gru" "--ttrank 16" "--ttrank 16 --gru"; do run $f; TTRNN_DEV=524288 run $f; done
fi
if [ $what != eval ]; then
for f in "--naive_tt" "--ttrank 16" "--gru" "--hidden_size 256 --gru --ttrank 16" "--in_size 40 --hidden_size 768 --ncores 4" "--in_size 1 --hidden_size 256 --seq_len 784 --batch_size 64 --naive_tt" "--naive_tt --gru"; do run --train $f; done
fi
