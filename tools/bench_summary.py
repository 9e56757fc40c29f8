import json, sys
d = json.loads(sys.stdin.readlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "pairs/s", d["ms_per_step"], "ms/step frac", r["frac"], "avg_us", r["avg_launch_us"],
      "jobs/launch", r.get("jobs_per_launch"), "sweep_ms", r["kernels_ms"].get("nn_sweep"), "lat_ms", d.get("single_pair_latency_ms"))
