export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd
python -m pytest tests/test_chain_wgrad.py -q -x 2>&1 | tail -2
for m in 3 2; do python tools/c2w_bench.py 2 $m 10 | tail -1; done
export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_abl.so
for m in 2 3; do
  for b in 0 3 8; do
    TTRNN_DEV2=$((65536*b)) python tools/c2w_bench.py 2 $m 10 2>&1 | tail -4
  done
done
