#!/bin/bash
# Round 2, second GPU session: the new deposition configurations -- parity first, then A/B timing.
set -u
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "tile_variants or zero_disp or lds_tiles or deposit_current" 2>&1 | tail -15 > gpurun_out/r2b/pytest_variants.txt
tail -6 gpurun_out/r2b/pytest_variants.txt
timeout 900 python scripts/deposit_variants.py > gpurun_out/r2b/variants.txt 2> gpurun_out/r2b/variants.err
cat gpurun_out/r2b/variants.txt; tail -3 gpurun_out/r2b/variants.err
