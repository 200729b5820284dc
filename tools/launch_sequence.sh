#!/bin/bash
# The ordered kernel launches of ONE step of a bench.py workload (rocprofv3 --kernel-trace, the last step of the run):
#     tools/launch_sequence.sh <tag> [bench.py args...]     -> gpurun_out/seq_<tag>.txt
# name | start offset within the step (us) | duration (us) | gap to the previous kernel's end (us)
set -u
REPO=$PWD
TAG=$1; shift
OUT=$REPO/gpurun_out/seqraw_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# SEQ_HARNESS=1: the step of examples/benchmarking.py (the reference's own harness flags: --tt --hidden_size ... --train) instead
if [ "${SEQ_HARNESS:-0}" = "1" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- \
    python3 "$REPO/examples/benchmarking.py" "$@" -n 4 > "$OUT/bench.json" 2> "$OUT/err.txt"
else
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- \
    python3 "$REPO/bench.py" --no-cpu-baseline "$@" --steps 4 --warmup 2 > "$OUT/bench.json" 2> "$OUT/err.txt"
fi
python3 - "$OUT" "$REPO/gpurun_out/seq_$TAG.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the step = the kernels between the last two launches of the step's first recurrent forward kernel
# training runs: a step ends with the optimizer's multi-tensor kernels; forward runs: a step = one recurrent forward kernel
# (single-layer workloads) — the kernels between the last two boundaries
ends = [i for i, n in enumerate(names) if 'multi_tensor_apply' in n and (i + 1 == len(names) or 'multi_tensor_apply' not in names[i + 1])]
if len(ends) >= 2:
    lo, hi = ends[-2] + 1, ends[-1] + 1
else:
    marks = [i for i, n in enumerate(names) if 'fwd_f10' in n or 'lstm_fwd' in n or 'gru_fwd' in n or 'g2_fwd' in n]
    lo = marks[-2] if len(marks) >= 2 else 0
    hi = marks[-1]
with open(sys.argv[2], 'w') as out:
    t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.write('%-100s %9.1f %8.1f %7.1f\n' % (r['Kernel_Name'][:100], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
    out.write('kernels %d  span %.1f us\n' % (hi - lo, (prev_end - t0) / 1e3))
PY
rm -rf "$OUT"
cat "$REPO/gpurun_out/seq_$TAG.txt"
