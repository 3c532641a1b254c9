#!/bin/bash
# usage: tools/pmc.sh TAG "CTR1 CTR2 ..." ["CTR..." ...]   - one rocprofv3 --pmc pass per counter set (bench.py, 5 steps)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
tag=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.log 2>&1
  f=$(find gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'mlp_kernel<0' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:36]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items():
        print(f"{k:38s} {c:30s} n={len(vals):2d} mean={sum(vals)/len(vals):.6g}")
PY
done
