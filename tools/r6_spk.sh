# speaker-encoder-sized harness timings + kernel stats (round 6):  tools/r6_spk.sh <tag>
export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
TAG=${1:-spk}
mkdir -p gpurun_out/r6
run() { echo "== $*"; python examples/benchmarking.py --tt -n 5 "$@" 2>&1 | grep "mean time" | tail -1; }
( run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  TTRNN_DEV2=1 run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  TTRNN_DEV2=2 run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  TTRNN_DEV2=1 run --train --in_size 40 --hidden_size 768 --ncores 4 --ttrank 4
  run --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 4
  run --train --gru --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
  TTRNN_DEV2=1 run --train --gru --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2
) > gpurun_out/r6/${TAG}_times.txt 2>&1
cat gpurun_out/r6/${TAG}_times.txt
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- \
  python3 $REPO/examples/benchmarking.py --tt -n 6 --train --in_size 40 --hidden_size 768 --ncores 2 --ttrank 2 > /dev/null 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r6/${TAG}_kernel_stats.csv
head -12 $REPO/gpurun_out/r6/${TAG}_kernel_stats.csv | cut -c1-200
