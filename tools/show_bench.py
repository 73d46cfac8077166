import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("algorithmic_bytes_per_launch"), d["roofline"].get("traffic_unit"))
print({k: round(v.get("ms_per_step", v.get("ms_per_batch", 0)), 2) for k, v in d["workloads"].items()})
print(d["workloads"]["cfg5"]["roofline"]["frac"], d["workloads"]["cfg5"]["weight_gradients_beside_the_chains"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"])
