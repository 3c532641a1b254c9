#!/usr/bin/env python3
"""The training step priced against the rate of a read + write stream (tools/probes/copy_patterns.hip: 5.3 TB/s on MI355X):
    python3 tools/train_table.py profiles/r3/b_train_traffic.json profiles/r3/b_train_kernel_stats.csv
per kernel: bytes per launch by the counters (FETCH_SIZE x 2 + WRITE_SIZE of train_traffic.json, divided by ITS launch count), the average
duration of train_kernel_stats.csv, the rate, and that rate as a fraction of the copy rate; `floor` = the kernel's bytes per step at the copy rate."""
import csv
import json
import sys

COPY = 5.3e12


def main():
    t = json.load(open(sys.argv[1]))
    rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(sys.argv[2]))}
    print(f"{'kernel':50s} {'launches/step':>13s} {'MB/launch':>10s} {'us/launch':>10s} {'TB/s':>6s} {'of copy rate':>12s} {'ms/step':>8s} {'floor ms':>8s}")
    tot_ms = tot_floor = 0.0
    for k in t["kernels"]:
        cand = [v for n, v in rows.items() if n.startswith(k["kernel"])]
        lps = k["launches_per_step"]
        mb = (k["fetch_MB_per_step_x2"] + k["write_MB_per_step"]) / max(lps, 1e-9)
        if not cand or mb < 20:
            continue
        us = float(cand[0]["AverageNs"]) / 1e3
        ms, floor = us * lps / 1e3, mb * lps * 1e6 / COPY * 1e3
        tot_ms += ms
        tot_floor += floor
        print(f"{k['kernel'].replace('void snerf::', '').replace('snerf::', '')[:50]:50s} {lps:13.2f} {mb:10.0f} {us:10.1f} {mb * 1e6 / (us * 1e-6) / 1e12:6.2f} {100 * floor / ms:11.0f}% {ms:8.2f} {floor:8.2f}")
    print(f"{'sum of the rows above':50s} {'':13s} {'':10s} {'':10s} {'':6s} {100 * tot_floor / tot_ms:11.0f}% {tot_ms:8.2f} {tot_floor:8.2f}")
    print(f"whole step by the counters: {t['bytes_per_step'] / 1e9:.1f} GB = {t['bytes_per_step'] / COPY * 1e3:.2f} ms at {COPY / 1e12:.1f} TB/s")


if __name__ == "__main__":
    main()
