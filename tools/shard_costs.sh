mkdir -p gpurun_out/shards
for w in cfg4 cfg5; do for n in 1 2 4 8; do for m in forward train; do
python bench.py --workload $w --mode $m --shard-of $n --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > gpurun_out/shards/${w}_${m}_of$n.json
python -c "import json; d=json.load(open('gpurun_out/shards/${w}_${m}_of$n.json')); print('$w $m of $n B', d['config']['per_gpu_batch'], 'ms', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median'],3))"
done; done; done
