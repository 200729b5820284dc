export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
mkdir -p gpurun_out/r6b
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r6b/pytest_gpu.txt
python tools/stress_backward.py --grid --reps 8 > gpurun_out/r6b/stress_backward_grid.txt 2>&1
python tools/stress_backward.py --reps 40 > gpurun_out/r6b/stress_backward.txt 2>&1
bash tools/variant_sweep.sh > gpurun_out/r6b/variants_benchmarking.txt 2>&1
cat gpurun_out/r6b/pytest_gpu.txt; tail -1 gpurun_out/r6b/stress_backward_grid.txt; tail -3 gpurun_out/r6b/stress_backward.txt
