#!/bin/bash
# usage (GPU box): tools/trace_gaps.sh [extra bench.py arguments, e.g. --train-graph]  - kernel trace of the training bench: busy time vs span of the timed steps, gap histogram
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline "$@" > $out/log 2>&1
f=$(find $out/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# last 4 steps: find Adam launches as step delimiters
adam = [i for i, r in enumerate(rows) if 'adam' in r[2].lower()]
print("kernels", len(rows), "adam launches", len(adam))
if len(adam) >= 5:
    a, b = adam[-5], adam[-1]
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"4 steps: span {span/4e6:.3f} ms/step, kernel busy {busy/4e6:.3f} ms/step, launches/step {len(seg)/4:.0f}, idle {sum(pos)/4e6:.3f} ms/step")
    import collections
    h = collections.Counter(min(int(g / 1000), 20) for g in pos)
    print("gap histogram (us: count/step):", {k: round(v / 4, 1) for k, v in sorted(h.items())})
    per = collections.defaultdict(lambda: [0, 0])
    for s_, e_, n_ in seg:
        per[n_][0] += 1; per[n_][1] += e_ - s_
    for n_, (c_, t_) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {n_:52s} {c_/4:6.1f} per step  {t_/c_/1e3:8.1f} us")
    big = sorted(((g, seg[i][2], seg[i + 1][2]) for i, g in enumerate(gaps) if g > 8000), reverse=True)[:12]
    for g, a_, b_ in big: print(f"  {g/1e3:7.1f} us between {a_} -> {b_}")
PY
rm -rf $out/tr
