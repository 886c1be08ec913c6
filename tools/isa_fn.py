#!/usr/bin/env python3
"""Extract one kernel's ISA from a hipcc -save-temps .s file and print an instruction histogram of its hottest loop(s).
usage: isa_fn.py file.s <substring of the mangled name> [--dump out.s]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = end = None
for i, ln in enumerate(lines):
    if start is None and re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", ln):
        start = i
    elif start is not None and ln.startswith(".Lfunc_end"):
        end = i
        break
body = lines[start:end]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
def ops(ls):
    c = collections.Counter()
    for ln in ls:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        c[t.split()[0]] += 1
    return c
tot = ops(body)
print(f"{key}: {len(body)} lines; mfma {sum(v for k, v in tot.items() if k.startswith('v_mfma'))}, accvgpr_read {tot['v_accvgpr_read_b32']}, "
      f"accvgpr_write {tot['v_accvgpr_write_b32']}, scratch {sum(v for k, v in tot.items() if k.startswith('scratch_'))}")
# loops: label ... backward branch to label
labels = {ln.split(":")[0]: i for i, ln in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", ln)}
loops = []
for i, ln in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", ln)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
loops.sort(key=lambda ab: -(ab[1] - ab[0]))
for a, b in loops[:3]:
    c = ops(body[a:b + 1])
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(f"  loop lines {a}-{b}: VALU {valu}, MFMA {sum(v for k, v in c.items() if k.startswith('v_mfma'))}, "
          f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, scratch {sum(v for k, v in c.items() if k.startswith('scratch_'))}, "
          f"accvgpr {c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32']}, s_nop {c['s_nop']}, waitcnt {c['s_waitcnt']}, barrier {c['s_barrier']}")
    print("    " + ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
