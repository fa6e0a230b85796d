"""Print the kernel timeline of one steady-state step from a rocprofv3 kernel trace (diagnostic).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [step_number] [anchor kernel prefix]
A step runs from one launch of the anchor kernel (default: the first kernel of the explicit bank step's main chain) to the
next one."""
import csv, glob, os, sys, statistics
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
anchor = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if anchor is None:
    names = {r['Kernel_Name'].split('(')[0] for r in rows}
    anchor = next(a for a in ('k_tf_compose_fwd', 'k_tf_ortho_coefs', 'k_tf8_coefs', 'k_adam') if a in names)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(anchor + '(')]
n = min(n, len(idx) - 2)
seg = rows[idx[n]: idx[n + 1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} d={(e-s)/1e3:6.1f} q={r['Queue_Id']:>3s} {r['Kernel_Name'][:64]}")
print("kernels per step:", len(seg) - 1, " anchor:", anchor)
starts = [int(rows[i]['Start_Timestamp']) for i in idx]
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])][len(idx) // 4:]
if per:
    print(f"period us: median {statistics.median(per):.1f}  (10th pct {sorted(per)[len(per)//10]:.1f})")
