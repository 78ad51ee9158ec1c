for S in 2 3 4; do for r in 1 2; do python bench.py --no-cpu-baseline --no-sanity --steps 12 --sort-interval $S 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sort interval $S', 'ms/step %.3f value %.3e' % (j['ms_per_step'], j['value']), {k: round(v['avg_ms'],3) for k,v in j['kernels'].items()})
"; done; done
