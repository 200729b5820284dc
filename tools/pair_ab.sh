export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
mkdir -p gpurun_out/nv
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_samples_per_workgroup" 2>&1 | tail -15
run() { echo "== $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
run --naive_tt
TTRNN_DEV=524288 run --naive_tt
run --in_size 40 --hidden_size 768 --ncores 4
TTRNN_DEV=524288 run --in_size 40 --hidden_size 768 --ncores 4
run --naive_tt --gru
TTRNN_DEV=524288 run --naive_tt --gru
run --train --naive_tt
