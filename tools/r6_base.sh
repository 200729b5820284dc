export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q -k "spk or gradients_golden" 2>&1 | tail -5 > gpurun_out/r6/base_pytest.txt
run() { echo "== $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
( run --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  run --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 4
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 2
) > gpurun_out/r6/base_spk.txt 2>&1
python bench.py > gpurun_out/r6/base_bench.json 2> gpurun_out/r6/base_bench.err
cat gpurun_out/r6/base_pytest.txt gpurun_out/r6/base_spk.txt; tail -c 600 gpurun_out/r6/base_bench.json
