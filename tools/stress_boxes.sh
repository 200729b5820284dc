#!/bin/bash
# One determinism-stress sample on whatever MI355X box this call landed on; writes gpurun_out/stress/<tag>.log
#   tools/stress_boxes.sh <tag> [reps] [pytest process runs]
tag=${1:-s}
mkdir -p gpurun_out/stress
{
  echo "== $(hostname) $(date -u +%FT%TZ)"
  rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
  python tools/stress_determinism.py --reps ${2:-500} --cases small,mid,cfg4 --routes default,nogemm
  python tools/stress_determinism.py --poison --reps 50 --cases small,mid,cfg4,cfg2,cfg3 --routes default,nogemm
  # the pytest case that was seen to differ in round 1, as separate processes (fresh allocator state each time)
  fails=0
  for i in $(seq 1 ${3:-3}); do
    python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeat_runs or poisoned" > gpurun_out/stress/$tag.pytest.$i.log 2>&1 || { fails=$((fails+1)); tail -30 gpurun_out/stress/$tag.pytest.$i.log; }
  done
  echo "pytest process runs: ${3:-3}, failed: $fails"
} > gpurun_out/stress/$tag.log 2>&1
grep -E "summary|failed|Unique" gpurun_out/stress/$tag.log
