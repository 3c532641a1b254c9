cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcx2_$t -- python3 bench.py --steps 5 --warmup 2 --headline-only > gpurun_out/pmcx2_$t.log 2>&1
  f=$(find gpurun_out/pmcx2_$t -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_i8" in r['Kernel_Name']:
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    for c, vals in v.items():
        print(f"{k:42s} {c:30s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
PY
  rm -rf gpurun_out/pmcx2_$t
done
