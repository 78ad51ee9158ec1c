#!/bin/bash
# First GPU call of the next round: the tests written after round 1's GPU budget was spent
# (WXA_UNVERIFIED_GPU_TESTS=1 un-skips them), one test at a time under its own timeout so that a
# failure or a hang in one does not hide the others, then the ordinary suite.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_unverified.sh'
set -u
mkdir -p gpurun_out/unverified
export WXA_UNVERIFIED_GPU_TESTS=1
OUT=gpurun_out/unverified/report.txt
: > $OUT
for t in \
    "tests/test_kernels_gpu.py::test_apply_particle_boundaries" \
    "tests/test_kernels_gpu.py::test_apply_pec_rho" \
    "tests/test_kernels_gpu.py::test_shift_field_window" \
    "tests/test_kernels_gpu.py::test_laser_push" \
    "tests/test_step_gpu.py::test_pec_particle_golden_on_gpu" \
    "tests/test_step_gpu.py::test_particle_boundaries_golden_on_gpu" \
    "tests/test_step_gpu.py::test_laser_acceleration_golden_on_gpu" \
    "tests/test_step_gpu.py::test_picmi_langmuir_golden_on_gpu" \
    "tests/test_step_gpu.py::test_decks_reach_the_reference_golden_checksums_on_gpu" \
    "tests/test_kernels_gpu.py::test_evolve_b_guard_layer" \
    "tests/test_kernels_gpu.py::test_gather_push_in_two_parts" \
    "tests/test_kernels_gpu.py::test_add_plasma" \
    "tests/test_multibrick_gpu.py::test_bricks_with_overlapped_halo_exchange"; do
    echo "=== $t" >> $OUT
    timeout 600 python -m pytest "$t" -q 2>&1 | tail -15 >> $OUT
done
unset WXA_UNVERIFIED_GPU_TESTS
echo "=== full suite" >> $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 >> $OUT
grep -E "^===|passed|failed|error" $OUT
