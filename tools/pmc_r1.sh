cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_r1a_$tag -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
  f=$(find gpurun_out/pmc_r1a_$tag -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items():
        print(f"{k:42s} {c:32s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
PY
done
