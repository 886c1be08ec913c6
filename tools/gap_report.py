#!/usr/bin/env python3
"""Idle gaps between the kernels of one hipGraph-replayed step, from a rocprofv3 --kernel-trace csv.
usage: gap_report.py <kernel_trace.csv> [memory_copy_trace.csv]"""
import csv, sys, collections
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Scratch_Size', 0) or 0)) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '')[:48]
ends = [i for i, r in enumerate(rows) if 'k_adamw' in r[2]]
per = 3                                  # k_adamw launches per step
steps = [ends[i] for i in range(per - 1, len(ends), per)]
a, b = steps[-2], steps[-1]              # the last step
w = rows[a + 1:b + 1]
span, busy = w[-1][1] - w[0][0], sum(e - s for s, e, _, _ in w)
gaps = [max(0, w[i + 1][0] - w[i][1]) for i in range(len(w) - 1)]
print(f'last step: launches {len(w)} span {span/1e6:.3f} ms busy {busy/1e6:.3f} ms gaps {sum(gaps)/1e6:.3f} ms (mean {sum(gaps)/len(gaps)/1e3:.2f} us)')
hist = collections.Counter(min(int(g / 1000), 20) for g in gaps)
print('gap histogram (us: count):', sorted(hist.items()))
print('kernels with scratch:', dict(collections.Counter((short(r[2]), r[3]) for r in w if r[3] > 0)))
cl = collections.Counter()
for i, g in enumerate(gaps):
    if g > 3000: cl[(short(w[i][2]), short(w[i + 1][2]))] += g
for (x, y), g in cl.most_common(25): print(f'  {g/1e3:8.1f} us  {x:48s} -> {y}')
if len(sys.argv) > 2:
    cp = list(csv.DictReader(open(sys.argv[2])))
    t0, t1 = w[0][0], w[-1][1]
    inside = [r for r in cp if t0 <= int(r['Start_Timestamp']) <= t1]
    print(f'memory copies: {len(cp)} total, {len(inside)} inside the last step')
    for r in inside[:20]: print('  ', {k: r[k] for k in r if k in ('Direction', 'Start_Timestamp', 'End_Timestamp', 'Bytes', 'Kind')})
