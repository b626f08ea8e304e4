#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into a markdown table.
    python tools/prof_summary.py gpurun_out/prof2 profiles/r02_step.md --steps 7 --title "..." --cmd "python bench.py ..."

Two tables: per kernel SYMBOL (what `--stats` prints; bench.py's roofline.kernel names one of these rows and its
avg_ms must agree with the row's average), and, for the MFMA kernels, per symbol x launch grid, so that a row can be
matched to one launch shape (grid = tiles_m * tiles_n [* split_k] workgroups)."""
import argparse
import csv
import glob
import os
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("htrvt::", "")
    return n.split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--steps", type=int, required=True, help="training steps the profiled command executed")
    ap.add_argument("--title", default="")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--by-grid", default="gemm,attn_", help="comma-separated symbol prefixes listed per launch grid as well")
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
        for r in rows[:45]:
            o.write(f"| {float(r['TotalDurationNs']) / 1e6 / a.steps:.3f} | {float(r['Percentage']):.1f} | "
                    f"{int(r['Calls']) / a.steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                    f"{float(r['MaxNs']) / 1e3:.1f} | `{short(r['Name'])}` |\n")
        tr = glob.glob(os.path.join(a.src, "*", "*_kernel_trace.csv"))
        prefixes = tuple(p for p in a.by_grid.split(",") if p)
        if tr and prefixes:
            groups = defaultdict(list)
            for r in csv.DictReader(open(tr[0])):
                n = short(r["Kernel_Name"]).replace("void ", "")
                if n.startswith(prefixes):
                    wg = int(r["Workgroup_Size_X"]) or 1
                    grid = (int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
                    groups[(n, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            o.write("\n## MFMA kernels per launch grid (workgroups x, y, z)\n\n")
            o.write("| ms/step | calls/step | avg us | min us | max us | kernel | grid |\n|---:|---:|---:|---:|---:|---|---|\n")
            for (n, grid), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:60]:
                o.write(f"| {sum(ds) / 1e6 / a.steps:.3f} | {len(ds) / a.steps:.1f} | {sum(ds) / len(ds) / 1e3:.1f} | "
                        f"{min(ds) / 1e3:.1f} | {max(ds) / 1e3:.1f} | `{n}` | {grid[0]} x {grid[1]} x {grid[2]} |\n")


if __name__ == "__main__":
    main()
