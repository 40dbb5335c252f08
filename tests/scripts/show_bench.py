import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("  ",k, round(v["avg_launch_ms"],4), round(v["ms_per_registration"],3))
