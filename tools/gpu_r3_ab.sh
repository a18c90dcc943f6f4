#!/bin/bash
# round 3, call AB: the default bench line once more (its configs[2] sub-record now finds the traffic file of its workload)
O=gpurun_out/r3ab
mkdir -p $O
( time python bench.py ) > $O/bench.log 2>&1; grep '^{' $O/bench.log | tail -1 > $O/r3_bench_line.json
python - <<PY
import json
d=json.load(open("$O/r3_bench_line.json"))
print("bench", d["ms_per_step"], d["value"], "roofline", {k:d["roofline"].get(k) for k in ("frac","achieved","traffic")})
print("e2e", d["end_to_end_from_files"] and d["end_to_end_from_files"].get("walls_s"))
c=d.get("config3_single_gpu") or {}
print("config3", {k:c.get(k) for k in ("ms_per_step","value","error")}, (c.get("roofline") or {}).get("frac"), (c.get("roofline") or {}).get("traffic"), (c.get("roofline") or {}).get("traffic_note"))
PY
