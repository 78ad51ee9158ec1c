#!/bin/bash
# round 6, session n: the suite and the bench line once more at the last commit, the way the driver runs them (flakiness check)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6n2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest rc=$?"
grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" $O/pytest_full.txt | tail -40 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default bench: ms/step %.3f value %.4e traffic %s frac %.3f cpu %s' % (d['ms_per_step'], d['value'], d['roofline']['traffic'], d['roofline']['frac'], d['cpu_baseline']['value']))"
