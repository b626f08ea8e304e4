#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into a markdown table.
    python tools/prof_summary.py gpurun_out/prof2 profiles/r01_step.md --steps 7 --title "..." """
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--steps", type=int, required=True, help="training steps the profiled command executed")
    ap.add_argument("--title", default="")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.src, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(a.dst, "w") as o:
        o.write(f"# {a.title}\n\n")
        o.write(f"Source: `rocprofv3 --kernel-trace --stats --output-format csv -- {a.cmd}` on one MI355X "
                f"(gfx950), {a.steps} steps in the run (warm-up + timed + 2 event-profiled).\n\n")
        o.write(f"Sum of kernel time: {tot / 1e6:.2f} ms = **{tot / 1e6 / a.steps:.2f} ms per step**, "
                f"{sum(int(r['Calls']) for r in rows)} dispatches.\n\n")
        o.write("| ms/step | % | calls/step | avg us | min us | max us | kernel |\n|---:|---:|---:|---:|---:|---:|---|\n")
        for r in rows[:40]:
            n = r["Name"].replace("(anonymous namespace)::", "").replace("htrvt::", "")
            n = n.split("(")[0][:90]
            o.write(f"| {float(r['TotalDurationNs']) / 1e6 / a.steps:.3f} | {float(r['Percentage']):.1f} | "
                    f"{int(r['Calls']) / a.steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                    f"{float(r['MaxNs']) / 1e3:.1f} | `{n}` |\n")


if __name__ == "__main__":
    main()
