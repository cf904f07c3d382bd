import sys, json
for l in sys.stdin:
    if l.startswith('{"metric'):
        d = json.loads(l); p = d["phase_ms_per_forward"]
        print("poses/s %.2f" % d["value"], {k: round(v, 3) for k, v in p.items() if v > 0.2})
