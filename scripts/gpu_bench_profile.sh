set -u
mkdir -p gpurun_out/final
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/final/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-pass ) > gpurun_out/final/rocprof.log 2>&1
python -c "import json; d=json.load(open('gpurun_out/final/bench.json')); print(d['value'], d['ms_per_step'])"
