python -m pytest tests -m gpu -x -q -k "by_products or gradients_golden or cfg4 or stack or reverse or dense or bwd or speaker or harness or naive or grid or runtime or ttlinear or heads or column_ranges" 2>&1 | tail -4
for i in 1 2; do for d in 0 1024; do for w in cfg2 cfg3 cfg4; do
TTRNN_DEV=$d python bench.py --workload $w --mode train --no-cpu-baseline --steps 16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dev $d', d['config']['workload'][:12], round(d['ms_per_step'],4), round(d.get('ms_per_step_median'),4))"
done; done; done
