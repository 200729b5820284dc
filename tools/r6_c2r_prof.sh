export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "2 3" "2 2" "4 2"; do
  set -- $cfg
  rm -rf /tmp/prof_c2r
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2r -o t -- python3 $REPO/tools/c2w_bench.py $1 $2 10 > /dev/null 2>&1
  echo "== rank $1 mats $2"
  head -8 $(find /tmp/prof_c2r -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-330
done
