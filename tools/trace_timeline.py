"""usage: trace_timeline.py <rocprofv3 -d dir with *_kernel_trace.csv> <last N dispatches>: start offset, duration, grid, LDS, kernel name"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name']
    n = n[n.find('k_'):][:64] if 'k_' in n else n[:64]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  grid=({r.get('Grid_Size_X')},{r.get('Grid_Size_Y')},{r.get('Grid_Size_Z')}) lds={r.get('LDS_Block_Size')} {n}")
