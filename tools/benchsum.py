import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
for k, v in d["roofline"].get("families", {}).items(): print(k, v)
if len(sys.argv) > 2:
    for k, v in d["roofline"]["per_shape"].items(): print(k, v)
