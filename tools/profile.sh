#!/bin/bash
# Profile one bench.py workload with rocprofv3 on the GPU box (run from the repo root):
#     tools/profile.sh <tag> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel durations).  Then ONE --pmc pass per counter, never combined with tracing
# (MI355X guide, HBM/rocprofv3 section; FETCH_SIZE + WRITE_SIZE in one pass already exceeds what the hardware collects
# at once and rocprofv3 then aborts and hangs — hence the per-pass timeout).  Raw output and summary.json land in
# gpurun_out/prof_<tag>/; copy what is worth keeping into profiles/.
set -u
REPO=$PWD
TAG=$1; shift
OUT=$REPO/gpurun_out/prof_$TAG
COUNTERS=${TTRNN_PROFILE_COUNTERS:-"FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
  python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
for C in $COUNTERS; do
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- \
    python3 "$REPO/bench.py" --no-cpu-baseline "$@" --steps 3 --warmup 1 > /dev/null 2> "$OUT/pmc_$C.err" \
    || echo "pass $C failed (rc $?)" >> "$OUT/failed_passes.txt"
done
python3 "$REPO/tools/profile_summary.py" "$OUT" > "$OUT/summary.json"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT"/pmc_*/ "$OUT/trace"          # raw per-dispatch dumps: large, summarised above
cat "$OUT/summary.json"
