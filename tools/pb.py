import json,sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_ms"],4), {k:round(v,3) for k,v in d["breakdown_ms_per_step"].items()})
