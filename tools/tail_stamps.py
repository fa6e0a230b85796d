"""Wall-clock stamps of the stages of k_tf_tail inside the replayed 7-band step, from a probe build of the library
(tools/build_probe_lib.sh tft blocktf.hip -DTFT_TIMING).   usage: python tools/tail_stamps.py lib_tft.so"""
import ctypes, io, os, sys
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from diffgfdn_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', '_probe', sys.argv[1])
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extras', '--steps', '50', '--repeats', '1']
import bench
with redirect_stdout(io.StringIO()):
    bench.main()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (64 * 8))()
lib.gfdn_probe_tf_tail_times.restype = ctypes.c_int
rc = lib.gfdn_probe_tf_tail_times(buf, 64 * 8)
t = np.array(buf, dtype=np.int64).reshape(64, 8).astype(np.float64) / 100.0
t = t[t[:, 0] > 0]
print("rc", rc, "workgroups", len(t))
names = ['row sums of the partial records', 'records -> dL/dQQ, dL/db, dL/dc (cofactor map)', 'expm adjoint -> dL/dM', 'Adam', 'expm + Q Q of the updated block',
         'next records (coefficients)']
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-50s median %6.2f  max %6.2f" % (nm, np.median(d), d.max()))
print("lifetime median %.2f max %.2f; start spread %.2f" % (np.median(t[:, 6] - t[:, 0]), (t[:, 6] - t[:, 0]).max(), t[:, 0].max() - t[:, 0].min()))
