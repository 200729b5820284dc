export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "h512 or rank_16 or half_piece_gru" 2>&1 | tail -5
run() { echo "== $TTRNN_DEV2 $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
for d in 0 256; do export TTRNN_DEV2=$d
run --train --gru
done
