timeout 300 python tools/submit_time.py wrapped 2>&1 | tail -2
timeout 300 python tools/submit_time.py single 2>&1 | tail -2
for i in 1 2 3; do python bench.py --workload single-1g --no-cpu-baseline --no-others 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['path_roofline']
print('value %.1f step %.4f ms spread %s index %.4f path frac %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_spread'], p['ms_index'], p['frac']))"; done
python bench.py --workload wrapped-10g --no-cpu-baseline --no-others 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['path_roofline']
print('wrapped value %.1f step %.4f ms spread %s index %.4f path frac %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_spread'], p['ms_index'], p['frac']))"
