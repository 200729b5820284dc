#!/bin/bash
# Developer tool: GPU timeline of the headline forward loop (per-dispatch start / end stamps from rocprofv3 --kernel-trace):
# kernel durations and the gaps between consecutive dispatches, steady state.   tools/gap_report.sh [bench.py args]
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gapprof
if [ -n "$GAP_LOOP" ]; then
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gapprof -o t -- python3 $REPO/tools/gap_loop.py > /dev/null 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gapprof -o t -- python3 $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > /dev/null 2>&1
fi
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/gapprof/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
prev_end = None
out = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((r['Kernel_Name'][:60], (s - prev_end) / 1e3 if prev_end else 0.0, (e - s) / 1e3))
    prev_end = e
# the first long run of (setup, K-rec) pairs
idx = [i for i, o in enumerate(out) if 'k_lstm_fwd_f10q' in o[0] or 'k_gru_fwd_f10vh' in o[0]]
for i in idx[8:13]:
    for j in (i - 1, i):
        print("%-62s gap before %8.2f us   duration %9.2f us" % out[j])
PY
