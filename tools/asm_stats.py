#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -save-temps .s file, per basic-block span (loop bodies show up as the spans
with the MFMAs).   python tools/asm_stats.py file.s [substring-of-kernel-name]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith(("global_load", "buffer_load", "scratch_load")): return "vload"
    if op.startswith(("global_store", "buffer_store", "scratch_store", "global_atomic")): return "vstore"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        print("==", name)
        block, cnt = "entry", collections.Counter()
        def flush():
            if sum(cnt.values()) >= 20:
                print(f"  {block:12s}", " ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith((";", "//")):
                continue
            if re.match(r"\.LBB\d+_\d+:", t):
                flush()
                block, cnt = t.rstrip(":"), collections.Counter()
                continue
            if t.startswith("."):
                continue
            cnt[classify(t.split()[0])] += 1
        flush()
        meta = re.search(r"\.name:\s+" + re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", text, re.S)


if __name__ == "__main__":
    main()
