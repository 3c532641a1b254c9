#!/bin/bash
# usage (on the GPU box, repo root): tools/profile_round.sh TAG
# Collects everything profiles/ needs for one build: bench lines (render + train), rocprofv3 kernel stats of the same
# commands, and the HBM traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, no tracing flags beside --pmc).
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
out=gpurun_out/round_$tag; mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --workload train > $out/train_bench.json 2> $out/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/prof.log 2>&1
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_train.log 2>&1
cp $(find $out/prof_train -name "*kernel_stats.csv" | head -1) $out/train_kernel_stats.csv
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > $out/pmc_$t.log 2>&1
  f=$(find $out/pmc_$t -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    for c, vals in v.items():
        print(f"{k:50s} {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
PY
done
rm -rf $out/prof $out/prof_train $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cat $out/bench.json; cat $out/train_bench.json | cut -c1-400; cat $out/pmc.txt | grep mlp_kernel
