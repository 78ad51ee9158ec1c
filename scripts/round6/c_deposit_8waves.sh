#!/bin/bash
# round 6, session c: the deposition on two waves per SIMD with the whole register file, with and without the next chunk's particles in registers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6c; mkdir -p $O
WXA_PRODUCT_LIB=warpx_amd/libwarpx_amd_dev.so timeout 1200 python scripts/variants.py base WXA_DEPOSIT_VARIANT=120 WXA_DEPOSIT_VARIANT=121 \
   WXA_DEPOSIT_VARIANT=122 WXA_DEPOSIT_VARIANT=123 WXA_DEPOSIT_VARIANT=91 --repeat 2 --no-step-time 2>&1 | grep -v "^\[{" | tee $O/deposit_8_waves.txt
