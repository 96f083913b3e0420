import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print(round(d["ms_per_step"],4), round(d["value"],1), [(k["kernel"],k["avg_ms"]) for k in d.get("kernels",[])])
