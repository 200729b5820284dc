export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "rank_16 or half_piece_gru or gru_fp32 or gru_four or dense_gemm_paths_gru" 2>&1 | tail -5
run() { echo "== $TTRNN_DEV2 $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
for d in 0 128; do export TTRNN_DEV2=$d
run --train --hidden_size 256 --gru --ttrank 16
run --hidden_size 256 --gru --ttrank 16
done
