#!/bin/bash
# rocprofv3 kernel stats + PMC passes for ONE configuration of examples/benchmarking.py (the reference's own harness flags):
#     tools/profile_harness.sh <tag> [benchmarking.py flags...]      -> gpurun_out/profh_<tag>/summary.json
set -u
REPO=$PWD
TAG=$1; shift
OUT=$REPO/gpurun_out/profh_$TAG
COUNTERS=${TTRNN_PROFILE_COUNTERS:-"SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD FETCH_SIZE"}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
  python3 "$REPO/examples/benchmarking.py" "$@" -n 6 > "$OUT/harness.txt" 2> "$OUT/trace.err"
for C in $COUNTERS; do
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- \
    python3 "$REPO/examples/benchmarking.py" "$@" -n 2 > /dev/null 2> "$OUT/pmc_$C.err" \
    || echo "pass $C failed (rc $?)" >> "$OUT/failed_passes.txt"
done
python3 "$REPO/tools/profile_summary.py" "$OUT" > "$OUT/summary.json"
rm -rf "$OUT"/pmc_*/ "$OUT/trace"
