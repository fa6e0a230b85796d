"""Print the kernel timeline of one steady-state step from a rocprofv3 kernel trace (diagnostic).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [step_number]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam(')]
seg = rows[idx[n] + 1: idx[n + 1] + 2]
first = next(i for i, r in enumerate(seg) if not r['Kernel_Name'].startswith(('k_adam', '__amd_rocclr')))
t0 = int(seg[first]['Start_Timestamp'])
busy_end = t0
for r in seg:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} d={(e-s)/1e3:6.1f} q={r['Queue_Id']:>3s} {r['Kernel_Name'][:64]}")
print("kernels:", len(seg), " span us:", (int(seg[-1]['End_Timestamp']) - t0) / 1e3)
# steady-state period: start of the step's first kernel to the next step's (median over the replayed steps)
import statistics
head = seg[first]['Kernel_Name'].split('(')[0]
starts = sorted(int(r['Start_Timestamp']) for r in rows if r['Kernel_Name'].startswith(head + '('))
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
ends = sorted(int(r['End_Timestamp']) for r in rows if r['Kernel_Name'].startswith('k_adam('))
import bisect
gaps = [(starts[i] - e) / 1e3 for e in ends for i in [bisect.bisect_left(starts, e)] if i < len(starts)]
if per and gaps:
    print(f"period us: median {statistics.median(per):.1f}  (10th pct {sorted(per)[len(per)//10]:.1f});"
          f"  last kernel of a step -> first kernel of the next: median {statistics.median(gaps):.1f}")
