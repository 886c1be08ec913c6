#!/usr/bin/env python3
"""Flag exposed memory round trips in loops: a global/buffer load followed within a few instructions by s_waitcnt vmcnt(0)
(or vmcnt(N) that cannot leave it in flight).  usage: isa_waits.py file.s [min_loop_lines]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
fn = None
start = 0
for i, ln in enumerate(lines):
    m = re.match(r"^(_Z\S+):", ln)
    if m:
        fn, start = m.group(1), i
        body_labels = {}
    if fn is None:
        continue
    if re.match(r"^\.LBB\d+_\d+:", ln):
        body_labels[ln.split(":")[0]] = i
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ln)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in body_labels and i - body_labels[lab] > int(sys.argv[2] if len(sys.argv) > 2 else 60):
            a, b = body_labels[lab], i
            # scan loop for load -> near wait
            for j in range(a, b):
                if re.search(r"\b(global_load|buffer_load|scratch_load)", lines[j]) and " lds" not in lines[j]:
                    n_ins = 0
                    for k in range(j + 1, min(b, j + 40)):
                        t = lines[k].strip()
                        if not t or t.startswith(";") or t.startswith("."):
                            continue
                        n_ins += 1
                        if re.search(r"\b(global_load|buffer_load|scratch_load|global_store|buffer_store)", t):
                            continue
                        mm = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
                        if mm and n_ins <= 8 and int(mm.group(1)) <= 1:
                            print(f"{fn[:90]}: loop {a - start}-{b - start}: load at +{j - start} `{lines[j].strip()[:50]}` waited {n_ins} instr later ({t})")
                            break
                        if n_ins > 8:
                            break
