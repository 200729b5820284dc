export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "naive_sets_reverse" 2>&1 | tail -8
