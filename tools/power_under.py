"""Shader clock and package power (rocm-smi, every 0.5 s) while a command runs:   python3 tools/power_under.py LABEL -- <command ...>
Prints one line: LABEL, samples taken while the GPU was busy (use > 90 %), median clock, median / max power.  The command is started as a child (never exec'd over a
process that has touched the GPU) and this process itself makes no HIP call."""
import re
import subprocess
import sys
import time

label = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = []
while p.poll() is None:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\): ([\d.]+)", out)
    use = re.search(r"GPU use \(%\): (\d+)", out)
    if sclk and pw and use:
        rows.append((int(use.group(1)), int(sclk.group(1)), float(pw.group(1))))
    time.sleep(0.5)
busy = [r for r in rows if r[0] > 90] or rows
busy = busy[len(busy) // 4:] if len(busy) > 8 else busy      # skip the ramp
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print(f"{label:<44} rc {p.returncode}  busy samples {len(busy):3d} of {len(rows):3d}   sclk median {med([r[1] for r in busy])} MHz   power median {med([r[2] for r in busy])} W  max {max([r[2] for r in busy], default=float('nan'))} W", flush=True)
