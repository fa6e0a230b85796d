"""Raw kernel timeline window from a rocprofv3 kernel trace (diagnostic): python tools/timeline_raw.py <dir> <start_frac> <count>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = int(len(rows) * frac)
# align to a step head
while not rows[i0]['Kernel_Name'].startswith('k_tf_ortho_coefs'):
    i0 += 1
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + cnt]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} d={(e-s)/1e3:6.1f} q={r['Queue_Id']:>3s} {r['Kernel_Name'][:50]}")
