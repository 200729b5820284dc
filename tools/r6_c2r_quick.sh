export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests/test_chain_wgrad.py -q -x 2>&1 | tail -2
for r in 2; do for m in 3 2; do
  python tools/c2w_bench.py $r $m 10 | tail -1
  TTRNN_DEV2=64 python tools/c2w_bench.py $r $m 10 | tail -1
done; done
run() { echo "== $TTRNN_DEV2 $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
for d in 0 64; do
  export TTRNN_DEV2=$d
  run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --train --gru --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
done
