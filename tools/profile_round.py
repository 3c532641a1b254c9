#!/usr/bin/env python3
"""On the GPU box, repo root:  python3 tools/profile_round.py TAG [--precision i8x3] [--no-train]

Collects under gpurun_out/round_TAG/ what profiles/ needs for one build:
  bench.json            the default `python3 bench.py` line
  kernel_stats.csv      rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-sweep --no-train --no-aux` (only 4096-ray launches of
                        the dominant kernel, so its AverageNs IS roofline.kernel_ms)
  pmc.txt               PMC passes, one counter set per run (--pmc only, no tracing flags): FETCH_SIZE, WRITE_SIZE, the MFMA set
  traffic.json          HBM bytes per launch of the dominant kernel (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
  recompute.txt         frac = algorithmic FLOP / AverageNs / peak, from the stats file alone
  train_*               the same for `bench.py --workload train` (kernel stats; traffic of the whole step)
  sweep_* / w512_* / exact_solar_*   kernel stats, PMC passes and HBM traffic of `bench.py --aux-kernel sweep | w512 | exact_solar` (the seasonal-sweep kernel at
                        512 x 512 x 96 x 12, the fused field kernel at W = 512, the ray-visibility kernel over the 6.3e6 secondary rays of a 256 x 256 x 96 image):
                        what `sweep_roofline` / `w512_roofline` / `exact_solar` of the bench line are measured on
rocprofv3 runs `python3 bench.py ...` directly (no shell / env hop after the `--`)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOP = 2 * (743936 + 71040 / 96.0) * 4096 * 96


def sh(cmd, log):
    with open(log, "w") as f:
        return subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=REPO, env=dict(os.environ, TMPDIR="/tmp"))


def find(d, pat):
    c = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return c[0] if c else None


def pmc(out, name, counters, bench_args):
    d = os.path.join(out, "pmc_" + name)
    sh(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", "bench.py"] + bench_args, os.path.join(out, f"pmc_{name}.log"))
    f = find(d, "*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "x"
    prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "auto"
    out = os.path.join(REPO, "gpurun_out", "round_" + tag)
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench.json"), "w") as f:
        subprocess.call(["python3", "bench.py"] + (["--precision", prec] if "--precision" in sys.argv else []), stdout=f, stderr=open(os.path.join(out, "bench.err"), "w"), cwd=REPO)
    bench = json.load(open(os.path.join(out, "bench.json")))
    kname = bench["roofline"]["kernel"].split(" ")[0]                      # e.g. snerf::mlp_i8_kernel<0,256,0>
    key = kname.split("::")[-1].split("<")[0]                               # mlp_i8_kernel
    targs = kname.split("<")[1].rstrip(">").replace(",", ", ")              # "0, 256, 0" as the trace spells it

    def dominant(name):
        return key + "<" in name and ("<" + targs in name.replace("(snerf::Program)", "").replace("ILi", "<") or targs.replace(" ", "") in name.replace(" ", ""))

    # kernel stats of the same command (only the 4096-ray launches)
    d = os.path.join(out, "prof")
    sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--precision", prec,
        "--no-cpu-baseline", "--no-sweep", "--no-train", "--no-aux"], os.path.join(out, "prof.log"))
    st = find(d, "*kernel_stats.csv")
    rec = []
    if st:
        shutil.copy(st, os.path.join(out, "kernel_stats.csv"))
        for r in csv.DictReader(open(st)):
            if dominant(r["Name"]):
                avg = float(r["AverageNs"])
                rec.append(f"{r['Name'][:80]}: calls {r['Calls']}, AverageNs {avg:.0f}, MinNs {r['MinNs']}, MaxNs {r['MaxNs']}\n"
                           f"  {FLOP / 1e9:.1f} GFLOP / {avg / 1e6:.4f} ms = {FLOP / (avg * 1e-9) / 1e12:.1f} TFLOP/s: frac {FLOP / (avg * 1e-9) / 5.0e15:.4f} of the int8 peak (5 Pop/s), "
                           f"{FLOP / (avg * 1e-9) / 2.5e15:.4f} of the bf16 peak   "
                           f"(bench.json roofline: kernel_ms {bench['roofline']['kernel_ms']:.4f}, frac {bench['roofline']['frac']:.4f} of peak {bench['roofline']['peak']:.0f}, "
                           f"frac_of_bf16_peak {bench['roofline'].get('frac_of_bf16_peak', float('nan')):.4f})")
    shutil.rmtree(d, ignore_errors=True)
    open(os.path.join(out, "recompute.txt"), "w").write("\n".join(rec) + "\n")
    # PMC passes
    bargs = ["--precision", prec, "--steps", "5", "--warmup", "2", "--headline-only"]
    lines, tot = [], {}
    for name, ctrs in [("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                       ("MFMA", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]),
                       ("INSTS", ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA"])]:
        acc = pmc(out, name, ctrs, bargs)
        for k, v in acc.items():
            for c, vals in v.items():
                lines.append(f"{k[:60]:62s} {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")
                if dominant(k):
                    tot[c] = sum(vals) / len(vals)
    open(os.path.join(out, "pmc.txt"), "w").write("\n".join(lines) + "\n")
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        json.dump({"kernel": kname, "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py " + " ".join(bargs) + " (separate passes, tools/profile_round.py)",
                   "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
                   "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM)",
                   "bytes_per_launch": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    if "--no-train" not in sys.argv and "--only" not in sys.argv:
        with open(os.path.join(out, "train_bench.json"), "w") as f:
            subprocess.call(["python3", "bench.py", "--workload", "train"], stdout=f, stderr=open(os.path.join(out, "train_bench.err"), "w"), cwd=REPO)
        d = os.path.join(out, "prof_train")
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--workload", "train", "--steps", "5",
            "--warmup", "2", "--no-cpu-baseline"], os.path.join(out, "prof_train.log"))
        st = find(d, "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(out, "train_kernel_stats.csv"))
        shutil.rmtree(d, ignore_errors=True)
        steps, warm = 3, 1
        per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = pmc(out, "train_" + c, [c], ["--workload", "train", "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline"])
            for k, v in acc.items():
                per[k][c] += sum(v[c])
                if c == "FETCH_SIZE":
                    per[k]["n"] += len(v[c])
        calls = steps + warm + 6            # bench_train also enqueues six steps onto an idle GPU after the timed region (host_enqueue_ms_per_step)
        rows = sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
        ps = lambda v: v * 1024 / calls
        res = {"command": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --workload train --steps {steps} --warmup {warm} --no-cpu-baseline",
               "calls_profiled": calls, "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM); dword gathers uncalibrated",
               "fetch_bytes_per_step_raw": ps(sum(v["FETCH_SIZE"] for _, v in rows)), "write_bytes_per_step": ps(sum(v["WRITE_SIZE"] for _, v in rows)),
               "kernels": [{"kernel": k[:90], "launches_per_step": v["n"] / calls, "fetch_MB_per_step_x2": 2 * ps(v["FETCH_SIZE"]) / 1e6,
                            "write_MB_per_step": ps(v["WRITE_SIZE"]) / 1e6} for k, v in rows[:24]]}
        res["bytes_per_step"] = 2 * res["fetch_bytes_per_step_raw"] + res["write_bytes_per_step"]
        json.dump(res, open(os.path.join(out, "train_traffic.json"), "w"), indent=1)
        with open(os.path.join(out, "train_copy_rate.txt"), "w") as f:      # the step priced against the rate of a read + write stream
            subprocess.call([sys.executable, os.path.join(REPO, "tools", "train_table.py"), os.path.join(out, "train_traffic.json"),
                             os.path.join(out, "train_kernel_stats.csv")], stdout=f, stderr=subprocess.STDOUT)
        # the dominant training kernel on its own (bench.py --workload train --train-kernel-only: what train_roofline is measured on)
        kargs = ["--workload", "train", "--train-kernel-only", "--steps", "8"]
        d = os.path.join(out, "prof_tk")
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py"] + kargs, os.path.join(out, "prof_tk.log"))
        st = find(d, "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(out, "train_kernel_kernel_stats.csv"))
        shutil.rmtree(d, ignore_errors=True)
        tk = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = pmc(out, "tk_" + c, [c], kargs)
            for k, v in acc.items():
                if "gemm_rows16_kernel" in k:
                    tk[c] = sum(v[c]) / len(v[c])
        if len(tk) == 2:
            json.dump({"kernel": "snerf::gemm_rows16_kernel<8,4,1,0>, forward 256->256, M = 393216", "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py " + " ".join(kargs),
                       "FETCH_SIZE_KB": tk["FETCH_SIZE"], "WRITE_SIZE_KB": tk["WRITE_SIZE"], "correction": "gfx950: FETCH_SIZE x2 (16 B/lane streaming reads)",
                       "bytes_per_launch": (2 * tk["FETCH_SIZE"] + tk["WRITE_SIZE"]) * 1024.0}, open(os.path.join(out, "train_kernel_traffic.json"), "w"), indent=1)
    # the two auxiliary kernels the bench line carries a roofline for: the seasonal-sweep kernel at configs[4]'s size and the fused field kernel at W = 512
    for aux, match, counters in (("sweep", "sweep_kernel", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]),
                                 ("w512", "mlp_i8_kernel", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]),
                                 # the exact-solar pass at 256 x 256 x 96 (6.3e6 secondary rays through the ray-visibility variant of the field kernel)
                                 ("exact_solar", "mlp_i8x2_kernel<256, 3>", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]),
                                 # round 6: the bf16x3 kernel of width 512 (K split over wave pairs) on the benchmark's rays, and the reference's DEFAULT render
                                 # configuration - width 512, converged (sharp) weights, exact solar - through its ray-visibility variant
                                 ("w512_bf16x3", "mlp_ks_kernel<512, 0>", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]),
                                 ("exact_solar_w512", "mlp_ks_kernel<512, 3>", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"])):
        if "--only" in sys.argv and aux not in sys.argv[sys.argv.index("--only") + 1].split(","):
            continue
        kargs = ["--aux-kernel", aux, "--steps", "6"]
        with open(os.path.join(out, aux + "_bench.json"), "w") as f:
            subprocess.call(["python3", "bench.py"] + kargs, stdout=f, stderr=subprocess.DEVNULL, cwd=REPO)
        d = os.path.join(out, "prof_" + aux)
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py"] + kargs, os.path.join(out, f"prof_{aux}.log"))
        st = find(d, "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(out, aux + "_kernel_stats.csv"))
        shutil.rmtree(d, ignore_errors=True)
        ak, lines = {}, []
        for name, ctrs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("SQ", counters)):
            acc = pmc(out, aux + "_" + name, ctrs, kargs)
            for k, v in acc.items():
                if match in k:
                    for c, vals in v.items():
                        ak[c] = sum(vals) / len(vals)
                        lines.append(f"{k[:60]:62s} {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")
        open(os.path.join(out, aux + "_pmc.txt"), "w").write("\n".join(lines) + "\n")
        if "FETCH_SIZE" in ak and "WRITE_SIZE" in ak:
            json.dump({"kernel": match, "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py " + " ".join(kargs) + " (separate passes)",
                       "FETCH_SIZE_KB": ak["FETCH_SIZE"], "WRITE_SIZE_KB": ak["WRITE_SIZE"], "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)",
                       "bytes_per_launch": (2 * ak["FETCH_SIZE"] + ak["WRITE_SIZE"]) * 1024.0}, open(os.path.join(out, aux + "_traffic.json"), "w"), indent=1)
    print(open(os.path.join(out, "recompute.txt")).read())
    print({k: v for k, v in tot.items()})


if __name__ == "__main__":
    main()
