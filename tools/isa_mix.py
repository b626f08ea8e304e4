#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file, with the waits spelled out (which s_waitcnt counts the
compiler and the inline asm put where) and, optionally, the instruction stream between two line numbers.
    python tools/isa_mix.py file.s kernel-name-substring [--dump]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
    c = collections.Counter()
    for l in lines:
        op = l.split()[0]
        if op.startswith("s_waitcnt"):
            c[l.split(";")[0].strip()] += 1
        elif op.startswith(("s_load", "s_buffer_load")):
            c["s_load"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("buffer_store") or op.startswith("global_store"):
            c["vstore"] += 1
        elif op.startswith("buffer_load") or op.startswith("global_load"):
            c["vload" + (" lds" if " lds" in l else "")] += 1
        elif op.startswith("scratch"):
            c["scratch"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith("ds_read"):
            c["ds_read"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            c["branch"] += 1
    print(name)
    for k, v in sorted(c.items()):
        print(f"  {v:5d} {k}")
    print(f"  {len(lines)} instructions")
    if "--dump" in sys.argv:
        print("\n".join(lines))
