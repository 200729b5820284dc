export TTRNN_LIB_PATH=$PWD/tools/bin/libttrnn_abl.so
for m in 2 3; do
  TTRNN_DEV2=0 python tools/c2w_bench.py 2 $m 10 2>&1 | tail -3
done
