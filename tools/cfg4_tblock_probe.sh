#!/bin/bash
# VERDICT r3 "Next round" 4 (cfg4: get gin out of HBM): does the recurrent kernel care where its hoisted input projection
# gin[B][T][H][4] comes from?  A T-blocked pipeline (K-in GEMM and K-rec alternating over blocks of 16-32 steps through one
# <= 67 MB block buffer that stays in the 256 MB Infinity Cache) can only pay if K-rec's step is shorter on a cache-resident gin.
# Measured here WITHOUT building the pipeline: the same three-layer cfg4 model at B = 512 with T = 16 / 32 / 64 / 160 — at T <= 32
# a layer's gin (34 / 67 MB) has just been written by the GEMM and is what the T-blocked pipeline's block buffer would be; at
# T = 160 (335 MB) it comes from HBM.  Per T: rocprofv3 kernel durations (K-rec ns per step), then FETCH_SIZE / WRITE_SIZE per launch.
#   tools/cfg4_tblock_probe.sh   -> gpurun_out/cfg4_tblock/summary.txt
REPO=$PWD
OUT=$REPO/gpurun_out/cfg4_tblock
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in 16 32 64 160; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$T -o t -- \
    python3 $REPO/bench.py --workload cfg4 --seq-len $T --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_T$T.json 2> $OUT/err_T$T.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p${T}_$C -o p -- \
      python3 $REPO/bench.py --workload cfg4 --seq-len $T --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/err_T$T.txt
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
lines = ["T   kernel                         calls  avg_us   ns/step   FETCH MB(x2)  WRITE MB   (per launch; B = 512, one layer's kernels)"]
for T in (16, 32, 64, 160):
    st = {}
    for f in glob.glob("%s/t%d/**/*kernel_stats.csv" % (out, T), recursive=True):
        for r in csv.DictReader(open(f)):
            st[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    pm = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = {}
        for f in glob.glob("%s/p%d_%s/**/*counter_collection.csv" % (out, T, C), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (tot, n) in acc.items():
            pm.setdefault(k, {})[C] = tot / n
    for name, (calls, us) in sorted(st.items(), key=lambda kv: -kv[1][1] * kv[1][0])[:6]:
        short = name.split("(")[0][-40:]
        key = next((k for k in pm if k.split("(")[0] == name.split("(")[0]), None)
        fe = 2 * pm[key].get("FETCH_SIZE", 0) * 1024 / 1e6 if key else float("nan")     # KB; gfx950 reports half of wide reads
        wr = pm[key].get("WRITE_SIZE", 0) * 1024 / 1e6 if key else float("nan")
        lines.append("%-3d %-40s %5d %8.1f %8.1f %10.1f %10.1f" % (T, short, calls, us, us * 1e3 / T, fe, wr))
    try:
        d = json.loads(open("%s/bench_T%d.json" % (out, T)).read().strip().splitlines()[-1])
        lines.append("%-3d forward %.4f ms per step = %.1f ns per timestep (three layers)" % (T, d["ms_per_step"], d["ms_per_step"] * 1e6 / T))
    except Exception as e:
        lines.append("%-3d bench line missing: %s" % (T, e))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/t* $OUT/p*
