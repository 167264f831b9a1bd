import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"],d["ms_per_step"]); print(json.dumps(d["roofline"])[:700]); print(json.dumps(d["cpu_baseline"])[:1800])
print(list(d.keys()))
