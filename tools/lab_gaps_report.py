#!/usr/bin/env python3
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '')[:40]
marks = [i for i, r in enumerate(rows) if 'cos' in r['Kernel_Name'].lower() and int(r['Grid_Size_X']) < 100000]
print('kernels', len(rows), 'markers', len(marks))
# the graph is replayed 3 times after capture warm-up: take the LAST full replay = the last 10 markers before the sin marker
sin = [i for i, r in enumerate(rows) if 'sin' in r['Kernel_Name'].lower()]
last = [m for m in marks if not sin or m < sin[0]][-13:]
names = "A rmsnorm,B w13_swiglu,C rmsnorm+w13,D seed_next,E gemm tn_n256,F w13 prealloc,G rmsnorm prealloc,H rmsnorm+w13 prealloc,I torch mul,J rmsnorm->w13 dependent,K w13->w2->rmsnorm dependent,L same in one allocation".split(',')
for k in range(len(last) - 1):
    w = rows[last[k]:last[k + 1] + 1]
    gaps = [(int(w[i + 1]['Start_Timestamp']) - int(w[i]['End_Timestamp'])) / 1e3 for i in range(len(w) - 1)]
    per = collections.defaultdict(list)
    for i, g in enumerate(gaps): per[short(w[i + 1]['Kernel_Name'])].append(g)
    print(f'{names[k]:24s}', {kk: (round(sorted(v)[len(v) // 2], 1), round(max(v), 1), len(v)) for kk, v in per.items()})
