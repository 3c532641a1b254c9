#!/bin/bash
# usage (GPU box): tools/pmc_linear.sh "COUNTERS" [plain|aol]  - SQ counters of the forward row GEMM (tools/bench_linear_fwd.py)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
mode=${2:-plain}
rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pl_$mode -- python3 tools/bench_linear_fwd.py $mode > /dev/null 2>&1
f=$(find gpurun_out/pl_$mode -name "*counter_collection.csv" | head -1)
python3 - "$f" $mode <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    if 'gemm_rows' in k:
        print(sys.argv[2], k, ' '.join(f"{c}={sum(x)/len(x):.4g}" for c, x in sorted(v.items())))
PY
rm -rf gpurun_out/pl_$mode
