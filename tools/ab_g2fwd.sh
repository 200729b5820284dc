#!/bin/bash
# A/B of the runtime-shape tier's forward (inference) on the variants of the reference's benchmarking.py: the current library against
# tools/bin/libttrnn_old.so (run from the repo root through gpurun)
for flags in "--gru" "--naive_tt" "--ttrank 16" "--n_layers 2 --hidden_size 384" "--in_size 40 --hidden_size 768 --ncores 4" "--in_size 40 --hidden_size 768 --ncores 2 --ttrank 4" "--hidden_size 256 --gru --ttrank 16" "--in_size 1 --hidden_size 256 --seq_len 784 --batch_size 64 --naive_tt"; do
  for lib in old new; do
    if [ $lib = old ]; then export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_old.so; else unset TTRNN_LIB_PATH; fi
    echo "== $lib $flags: $(python examples/benchmarking.py --tt -n 5 $flags 2>&1 | grep 'mean time' | tail -1)"
  done
done
unset TTRNN_LIB_PATH
