"""Same-box A/B of the directional bench: python tools/ab_dir.py [attr=value ...] -- <bench.py flags>
flips class attributes of model.DiffDirectionalFDNVarReceiverPos / trainer.DirectionalFDNVarReceiverPosTrainer for one process
(e.g. sub_fdn_by_transforms=False) and prints ms per band-step."""
import io, json, os, runpy, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
sep = args.index('--') if '--' in args else len(args)
sets, flags = args[:sep], args[sep + 1:]
from diffgfdn_amd.model import DiffDirectionalFDNVarReceiverPos as Model
from diffgfdn_amd.trainer import DirectionalFDNVarReceiverPosTrainer as Trainer
for kv in sets:
    k, v = kv.split('=')
    val = {'True': True, 'False': False}.get(v, v)
    tgt = Model if hasattr(Model, k) else Trainer
    assert hasattr(tgt, k), k
    setattr(tgt, k, val)
sys.argv = ['bench.py', '--config', 'directional', '--no-cpu-baseline'] + flags
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(' '.join(sets) or '(defaults)', '| ms_per_band_step', d['config'].get('ms_per_band_step'))
