"""Same-box A/B of a form of the explicit bank step: runs bench.py in this process with class attributes of
bankstep.FusedBankStep overridden.   usage: python tools/ab_attr.py name=value [name=value ...] -- [bench.py arguments]
(e.g. python tools/ab_attr.py fused_tail=False -- --no-cpu-baseline --no-extras --steps 400); prints the overrides and the
bench line's ms_per_step."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index('--') if '--' in args else len(args)
overrides, bench_args = args[:cut], args[cut + 1:]
from diffgfdn_amd.bankstep import FusedBankStep      # noqa: E402
for ov in overrides:
    name, val = ov.split('=', 1)
    if not hasattr(FusedBankStep, name):
        raise SystemExit(f"FusedBankStep has no attribute {name}")
    setattr(FusedBankStep, name, {'True': True, 'False': False}.get(val, int(val) if val.lstrip('-').isdigit() else val))
sys.argv = ['bench.py'] + bench_args
import bench                                        # noqa: E402
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
line = [l for l in buf.getvalue().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print(' '.join(overrides) or '(defaults)', '|', ' '.join(bench_args), '| ms_per_step', d['ms_per_step'])
