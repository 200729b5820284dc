export PYTHONPATH=$PWD:$PWD/tensorized-rnn_amd:$PWD/examples
python -m pytest tests -q -x -m gpu 2>&1 | tail -3
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o t -- python3 $REPO/examples/benchmarking.py --tt -n 6 --train --gru > /dev/null 2>&1
head -12 $(find /tmp/prof_g -name "*kernel_stats.csv" | head -1) | cut -c1-100,180-260
