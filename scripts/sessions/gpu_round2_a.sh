#!/bin/bash
# Round 2, first GPU session: whole parity suite (no -x), the gated overlapped-halo test on its own under a
# short timeout, smoke, the default bench line.
set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > gpurun_out/r2a/pytest_gpu.txt
tail -15 gpurun_out/r2a/pytest_gpu.txt
WXA_UNVERIFIED_GPU_TESTS=1 timeout 300 python -m pytest tests/test_multibrick_gpu.py -m gpu -q -rf -k overlapped 2>&1 | tail -30 > gpurun_out/r2a/pytest_overlap.txt
tail -5 gpurun_out/r2a/pytest_overlap.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -c 2500 gpurun_out/r2a/bench.json
