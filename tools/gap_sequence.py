#!/usr/bin/env python3
"""The kernel sequence of the last hipGraph-replayed step with the idle gap in front of every kernel, its duration and its dispatch
resources (rocprofv3 --kernel-trace csv).  usage: gap_sequence.py <kernel_trace.csv> > sequence.txt"""
import csv, sys, collections
rd = list(csv.DictReader(open(sys.argv[1])))
rows = sorted(rd, key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
ends = [i for i, r in enumerate(rows) if 'k_adamw_tick' in r['Kernel_Name']]
a, b = ends[-2], ends[-1]
w = rows[a + 1:b + 1]
tot = 0
byn, byp = collections.Counter(), collections.Counter()
cn, cp = collections.Counter(), collections.Counter()
print(f"{'gap us':>8s} {'dur us':>9s}  {'LDS':>7s} {'VGPR':>4s} {'AGPR':>4s} {'grid':>9s} {'wg':>4s}  kernel")
for i, r in enumerate(w):
    gap = (int(r['Start_Timestamp']) - int(w[i - 1]['End_Timestamp'])) / 1e3 if i else 0.0
    tot += max(gap, 0)
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{gap:8.2f} {dur:9.2f}  {r.get('LDS_Block_Size', '?'):>7s} {r.get('VGPR_Count', '?'):>4s} {r.get('Accum_VGPR_Count', '?'):>4s} "
          f"{r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s} {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4s}  {short(r['Kernel_Name'])}")
    if i:
        byn[short(r['Kernel_Name'])] += max(gap, 0); cn[short(r['Kernel_Name'])] += 1
        byp[short(w[i - 1]['Kernel_Name'])] += max(gap, 0); cp[short(w[i - 1]['Kernel_Name'])] += 1
span = (int(w[-1]['End_Timestamp']) - int(w[0]['Start_Timestamp'])) / 1e3
print(f"# launches {len(w)} span {span / 1e3:.3f} ms, idle {tot / 1e3:.3f} ms")
print("# idle in FRONT of a kernel, by kernel (total us, count, mean):")
for k, v in byn.most_common(25): print(f"#   {v:8.1f} {cn[k]:4d} {v / cn[k]:6.2f}  {k}")
print("# idle BEHIND a kernel, by kernel:")
for k, v in byp.most_common(25): print(f"#   {v:8.1f} {cp[k]:4d} {v / cp[k]:6.2f}  {k}")
