#!/bin/bash
# A/B of the runtime-shape tier's training step on the variants of the reference's benchmarking.py (run from the repo root through gpurun):
# the current library against tools/bin/libttrnn_old.so (a copy of an earlier build: cp tensorized-rnn_amd/ttrnn_hip/libttrnn.so there first),
# then the per-phase stamps of k_g2_bwd from the ablation build (make -C tensorized-rnn_amd/csrc ablation).
for flags in "--gru" "--naive_tt" "--ttrank 16" "--n_layers 2 --hidden_size 384" "--in_size 40 --hidden_size 768 --ncores 4" "--in_size 40 --hidden_size 768 --ncores 2 --ttrank 4"; do
  for lib in old new; do
    if [ $lib = old ]; then export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_old.so; else unset TTRNN_LIB_PATH; fi
    echo "== $lib $flags: $(python examples/benchmarking.py --tt -n 5 --train $flags 2>&1 | grep 'mean time' | tail -1)"
  done
done
unset TTRNN_LIB_PATH
python tools/diag_stamps_g2bwd.py --gru 2>&1 | tail -11
python tools/diag_stamps_g2bwd.py --naive_tt 2>&1 | tail -11
python tools/diag_stamps_g2bwd.py --in_size 40 --hidden_size 768 --ncores 4 2>&1 | tail -11
python tools/diag_stamps_g2bwd.py --hidden_size 384 2>&1 | tail -11
