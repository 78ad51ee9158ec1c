#!/bin/bash
# round 6, sessions v, x and z (x: with the sparse tiles through the table; z: the tiles in turn over the XCDs): the streaming deposition as committed (frames of a cell brought onto one, second particles deferred, lane pairs
# sharing the adding): the deposition / step / deck tests, BASELINE config 5 on one GPU twice, the headline, the kernel's SQ counters and
# the profile build's population table
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "deposit or streaming or cold_stream or laser or btd or boost or golden or deck" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
timeout 500 python scripts/bench_lwfa_boosted.py > $O/lwfa_boosted.json 2> $O/lwfa_boosted.err; echo "config 5 rc=$?"
python -c "
import json
d=json.load(open('$O/lwfa_boosted.json'))
print('config 5: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
"
done | tee $O/lwfa_boosted.txt
timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity > $O/tmp.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/tmp.json'))
print('headline:', 'ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})" | tee $O/headline.txt
rm -f $O/tmp.json
timeout 900 python scripts/lwfa_sq_counters.py $O/sq > $O/lwfa_sq_counters.txt 2>/dev/null; rm -rf $O/sq; cat $O/lwfa_sq_counters.txt | tail -4
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_prof.so timeout 600 python scripts/bench_lwfa_boosted.py --steps 10 > /dev/null 2> $O/prof.err
grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL\|amdgpu.ids" $O/prof.err | tail -22 | tee $O/lwfa_population_and_bodies.txt; rm -f $O/prof.err
