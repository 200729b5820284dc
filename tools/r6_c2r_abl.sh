export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd
export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_abl.so
for m in 3 2; do
  for b in 0 3 8 11; do
    TTRNN_DEV2=$((256*b)) python tools/c2w_bench.py 2 $m 10 2>&1 | tail -3
  done
done
