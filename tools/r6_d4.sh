# four-core encoder shapes: chain weight gradient tests + harness timings (round 6):  tools/r6_d4.sh <tag>
export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
TAG=${1:-d4}
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests/test_chain_wgrad.py tests/test_w2.py -x -q -m gpu 2>&1 | tail -8
run() { echo "== $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
( run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2
  run --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2
  run --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
) > gpurun_out/r6/${TAG}_times.txt 2>&1
cat gpurun_out/r6/${TAG}_times.txt
