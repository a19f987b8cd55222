"""one line per record of tools/adapt_probe.py's output"""
import json
import sys
for l in open(sys.argv[1]):
    try:
        r = json.loads(l)
    except Exception:
        continue
    c = r['chosen']
    forced = {k: (v['ms'] if isinstance(v, dict) else 'n/a') for k, v in r.items() if k in ('automaton', 'filter', 'flat_automaton')}
    eq = all(v.get('equal', True) for k, v in r.items() if isinstance(v, dict) and k != 'chosen')
    best = min([v for v in forced.values() if v != 'n/a'] or [0])
    print('%-11s %-10s m%-2d eng %d flips %d launches %s | forced %s | last/best %.2f eq %s' % (r['corpus'], r['set'], r['m'], c['engine_now'], c['flips'], c['per_launch_ms'], forced, c['last_ms'] / best if best else 0, eq))
