#!/bin/bash
# Round 4: the transport changes on the hardware (count round on its own communicator, no stream sync at the end of
# Redistribute), the brick tests, the dry-comm line on one GPU, and the bench line again (Redistribute no longer syncs).
#   gpurun --timeout 900 -- 'bash scripts/sessions/gpu_round4_g.sh'
set -u
OUT=$(pwd)/gpurun_out/r4g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_multibrick_gpu.py -m gpu -q -rf -s -k "rccl or bricks or overlap" 2>&1 | tail -12 | tee $OUT/pytest_transport_and_bricks.txt
timeout 300 python bench.py --dry-comm --ncell 128 2>&1 | tail -2 | tee $OUT/dry_comm_one_gpu.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['value'],{k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; tail -3 $OUT/bench.err
du -sh $OUT
