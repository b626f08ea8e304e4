#!/usr/bin/env python3
"""Summarise the counter passes of tools/collect_pmc.sh per kernel symbol -> profiles/rNN_pmc_*.{md,json}.

    python tools/pmc_summary.py gpurun_out/r2_pmc profiles/r02_pmc --steps 4

HBM bytes follow MI355X_MICROARCH.md (HBM section): read bytes = 2 x FETCH_SIZE (KiB) x 1024 -- on gfx950 FETCH_SIZE
counts 128-byte requests at 64 bytes --, written bytes = WRITE_SIZE (KiB) x 1024.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES
/ (kernel cycles x 1024 SIMDs) with kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("htrvt::", "").replace("void ", "")
    return n.split("(")[0][:90]


def load(d, by_grid=False):
    """{kernel symbol: {counter: [value per dispatch]}} and {kernel: [duration ns per dispatch]}; by_grid: the key is
    (symbol, workgroups of the launch) for the MFMA kernels"""
    vals, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if by_grid:
                k = (k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            per[(r["Dispatch_Id"], k)][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, k), cs in per.items():
            for c, v in cs.items():
                vals[k][c].append(v)
    if by_grid:
        return vals, dur
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return vals, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst", help="output prefix: <dst>_traffic.json and <dst>.md")
    ap.add_argument("--steps", type=int, required=True)
    a = ap.parse_args()
    mf, _ = load(os.path.join(a.src, "mfma"))
    fe, dur = load(os.path.join(a.src, "fetch"))
    wr, _ = load(os.path.join(a.src, "write"))
    ld, _ = load(os.path.join(a.src, "lds"))
    kernels = sorted(dur, key=lambda k: -sum(dur[k]))
    rows, symbols = [], {}
    for k in kernels[:40]:
        n = len(dur[k])
        rd = 2.0 * 1024 * sum(fe[k].get("FETCH_SIZE", [])) / max(1, len(fe[k].get("FETCH_SIZE", [])))
        wb = 1024.0 * sum(wr[k].get("WRITE_SIZE", [])) / max(1, len(wr[k].get("WRITE_SIZE", [])))
        busy = sum(mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        gui = sum(mf[k].get("GRBM_GUI_ACTIVE", []))
        util = busy / (gui / 8.0 * 1024) if gui else None
        conf, idx = sum(ld[k].get("SQ_LDS_BANK_CONFLICT", [])), sum(ld[k].get("SQ_LDS_IDX_ACTIVE", []))
        wait_any, act = sum(ld[k].get("SQ_WAIT_ANY", [])), sum(ld[k].get("SQ_ACTIVE_INST_ANY", []))
        rows.append((k, n / a.steps, sum(dur[k]) / n / 1e3, rd / 1e6, wb / 1e6, util, conf / idx if idx else None,
                     wait_any / (wait_any + act) if wait_any + act else None))
        symbols[k] = {"launches_per_step": n / a.steps, "read_bytes_per_launch_mean": rd, "written_bytes_per_launch_mean": wb,
                      "traffic_bytes_per_launch_mean": rd + wb, "mfma_busy_fraction": util}
    # per launch grid (= per launch shape) for the MFMA kernels
    gfe, _ = load(os.path.join(a.src, "fetch"), by_grid=True)
    gwr, _ = load(os.path.join(a.src, "write"), by_grid=True)
    gmf, _ = load(os.path.join(a.src, "mfma"), by_grid=True)
    grows = []
    for key in gfe:
        if not key[0].startswith(("gemm", "attn_")):
            continue
        fs, ws = gfe[key].get("FETCH_SIZE", []), gwr.get(key, {}).get("WRITE_SIZE", [])
        busy, gui = sum(gmf.get(key, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [])), sum(gmf.get(key, {}).get("GRBM_GUI_ACTIVE", []))
        grows.append((key[0], key[1], len(fs) / a.steps, 2.0 * 1024 * sum(fs) / max(1, len(fs)) / 1e6,
                      1024.0 * sum(ws) / max(1, len(ws)) / 1e6, busy / (gui / 8.0 * 1024) if gui else None, gui / 8.0 / max(1, len(fs))))
    grows.sort(key=lambda r: -(r[6] * r[2]))
    with open(a.dst + "_traffic.json", "w") as f:
        json.dump({"source": "tools/collect_pmc.sh (rocprofv3 --pmc, separate passes) over bench.py --steps 1 --warmup 1 "
                             "--no-overlap-wgrad; read bytes = 2 x FETCH_SIZE KiB x 1024, written = WRITE_SIZE KiB x 1024",
                   "symbols": symbols}, f, indent=1)
    with open(a.dst + ".md", "w") as o:
        o.write("# Hardware counters per kernel symbol, one training step (bf16, B=128, 64x1024)\n\n")
        o.write("`tools/collect_pmc.sh` = four `rocprofv3 --pmc ... --kernel-trace` passes over `python3 bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline --no-overlap-wgrad`; averages per launch over all launches of a symbol.  Durations are from the "
                "FETCH_SIZE pass (counter collection serialises dispatches; see the kernel-trace profile for timings).\n\n")
        o.write("| kernel | launches/step | avg us | HBM read MB | HBM written MB | MFMA busy | LDS conflict cycles / LDS cycles | wave wait share |\n")
        o.write("|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, n, us, rd, wb, util, conf, wt in rows:
            f = lambda v, p=2: "-" if v is None else f"{v:.{p}f}"
            o.write(f"| `{k}` | {n:.1f} | {us:.1f} | {rd:.1f} | {wb:.1f} | {f(util)} | {f(conf, 3)} | {f(wt)} |\n")
        o.write("\n## MFMA kernels per launch grid (one row = one launch shape; grid = workgroups = tiles_m x tiles_n [x split_k])\n\n")
        o.write("| kernel | workgroups | launches/step | HBM read MB / launch | HBM written MB / launch | MFMA busy | kcycles / launch |\n")
        o.write("|---|---:|---:|---:|---:|---:|---:|\n")
        for k, grid, n, rd, wb, util, cyc in grows[:48]:
            o.write(f"| `{k}` | {grid} | {n:.1f} | {rd:.1f} | {wb:.1f} | {'-' if util is None else f'{util:.2f}'} | {cyc / 1e3:.0f} |\n")


if __name__ == "__main__":
    main()
