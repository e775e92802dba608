#!/bin/bash
mkdir -p gpurun_out/final
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/driver.json 2> gpurun_out/final/driver.err ) 2>&1 | grep real
( time timeout 900 python bench.py > gpurun_out/final/default.json 2> gpurun_out/final/default.err ) 2>&1 | grep real
python - <<'P'
import json
for n in ("driver", "default"):
    d = json.loads(open(f"gpurun_out/final/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("mrays_per_s"), "seq", ((d.get("sequence") or {}).get("ladder") or {}).get("mrays_per_s"),
          "dropin", {k: v.get("mrays_per_s") for k, v in ((d.get("dropin") or {}).get("legs") or {}).items()}, "cpu", (d.get("cpu_baseline") or {}).get("value"), "roof", d["roofline"]["frac"], d["roofline"].get("frac_from_profiles", {}).get("frac"))
P
