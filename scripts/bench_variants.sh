for f in "" "--pusher vay" "--deposition direct" "--deposition direct --pusher vay" "--order 1" "--sync-each-call --steps 20 --warmup 5" "--steps 20 --warmup 5"; do python bench.py --no-cpu-baseline --no-sanity $f 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$f]', 'ms/step %.3f value %.3e' % (j['ms_per_step'], j['value']), {k: round(v['avg_ms'],3) for k,v in j['kernels'].items()})
"; done
