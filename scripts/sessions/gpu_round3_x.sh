#!/bin/bash
# Round 3, GPU session x: non-temporal stores (WXA_GATHER_PF=4) and loads (5) of the particle arrays in the gather, non-temporal
# (no gain; the variant code was removed again -- this script is the record of how r3x_* was produced)
# particle loads in the deposition (variant 60) -- A/B timing in a dev build.
set -u
OUT=$(pwd)/gpurun_out/r3x
mkdir -p $OUT
export TMPDIR=/tmp
DEV=$(pwd)/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_GATHER_RB=2,WXA_GATHER_PF=3 WXA_GATHER_RB=2,WXA_GATHER_PF=4 WXA_GATHER_RB=2,WXA_GATHER_PF=5 WXA_GATHER_RB=2,WXA_GATHER_PF=3,WXA_DEPOSIT_VARIANT=60 --repeat 3 \
    > $OUT/nontemporal.txt 2> $OUT/nontemporal.err
grep -v "^\[" $OUT/nontemporal.txt | head -12; tail -2 $OUT/nontemporal.err
