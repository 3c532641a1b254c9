#!/bin/bash
# usage (GPU box, repo root): tools/train_pmc.sh TAG "COUNTERS..."  - SQ counters of the training step's kernels (mean per launch)
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
out=gpurun_out/trainpmc_$tag; mkdir -p $out
rocprofv3 --pmc $@ --output-format csv -d $out/pmc -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc.log 2>&1
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > $out/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', kv[1].get('GRBM_GUI_ACTIVE', [0])))):
    print(k, ' '.join(f"{c}={sum(x)/len(x):.4g}" for c, x in v.items()), f"n={len(next(iter(v.values())))}")
PY
rm -rf $out/pmc
head -12 $out/pmc.txt
