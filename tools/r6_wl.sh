export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "half_piece" 2>&1 | tail -2
for d in 0 1024 0 1024; do
  TTRNN_DEV2=$d python bench.py --workload cfg2 --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dev2 $d', d['ms_per_step'], d.get('ms_per_step_blocks'))"
done
DIAG_B=64 python tools/diag_stamps_bwd.py 2>&1 | tail -10
