#!/bin/bash
# usage (on the GPU box, repo root): tools/train_traffic.sh TAG
# HBM traffic of the whole training step (BASELINE configs[2]) from the PMC counters: FETCH_SIZE and WRITE_SIZE in separate
# passes (no tracing flags beside --pmc), summed per kernel over the step, divided by the number of steps profiled.
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
out=gpurun_out/traffic_$tag; mkdir -p $out
STEPS=3; WARM=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload train --steps $STEPS --warmup $WARM --no-cpu-baseline > $out/pmc_$c.log 2>&1
  cp $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) $out/$c.csv
  rm -rf $out/pmc_$c
done
python3 - $out $STEPS $WARM <<'PY'
import csv, sys, json, collections
out, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tot = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}.csv")):
        k = r["Kernel_Name"].split("(")[0][:90]
        tot[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            tot[k]["n"] += 1
calls = steps + warm                      # every step() call of the run is identical work (the first also builds the engine)
rows = sorted(tot.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
per_step = lambda v: v * 1024 / calls      # counters are in KB
res = {"command": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --workload train --steps {steps} --warmup {warm} --no-cpu-baseline",
       "calls_profiled": calls,
       "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM); dword gathers uncalibrated",
       "fetch_bytes_per_step_raw": per_step(sum(v["FETCH_SIZE"] for _, v in rows)),
       "write_bytes_per_step": per_step(sum(v["WRITE_SIZE"] for _, v in rows)),
       "kernels": [{"kernel": k, "launches_per_step": v["n"] / calls, "fetch_MB_per_step_x2": 2 * per_step(v["FETCH_SIZE"]) / 1e6,
                    "write_MB_per_step": per_step(v["WRITE_SIZE"]) / 1e6} for k, v in rows[:24]]}
res["bytes_per_step"] = 2 * res["fetch_bytes_per_step_raw"] + res["write_bytes_per_step"]
json.dump(res, open(f"{out}/train_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
for r in res["kernels"][:12]:
    print(r)
PY
rm -f $out/FETCH_SIZE.csv $out/WRITE_SIZE.csv
