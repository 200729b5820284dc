#!/bin/bash
# One determinism-stress sample on whatever MI355X box this call landed on; appends to gpurun_out/stress/<tag>.log
tag=${1:-s}
mkdir -p gpurun_out/stress
{
  echo "== $(hostname) $(date -u +%FT%TZ)"
  rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
  python tools/stress_determinism.py --reps ${2:-2000} --cases small,mid,cfg4 --routes default,nogemm
  python tools/stress_determinism.py --poison --reps 50 --cases small,mid,cfg4,cfg2,cfg3 --routes default,nogemm
  echo "rc=$?"
} > gpurun_out/stress/$tag.log 2>&1
tail -9 gpurun_out/stress/$tag.log
