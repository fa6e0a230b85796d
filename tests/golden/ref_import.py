"""Import shim for the *reference* DiffGFDN package (container-only test infrastructure).

The reference lives at /root/reference (read-only) and imports many third-party
packages that are absent from this image (loguru, torchaudio, pyfar, slope2noise,
spaudiopy, ...).  This module installs a permissive ``sys.meta_path`` finder that
returns empty stand-in modules for those roots so that the reference's own
``diff_gfdn`` sources can be imported unmodified, purely to (a) generate the golden
vectors committed under ``tests/golden/`` and (b) time the reference CPU trainer.

It is NEVER imported by the product package and nothing here travels to the GPU box
in a form that is used at run time (``/root/reference`` does not exist there).
"""
import enum
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"

_ABSENT_ROOTS = (
    "loguru", "torchaudio", "librosa", "pyfar", "slope2noise", "spaudiopy", "sofar",
    "soundfile", "optuna", "h5py", "onnx", "onnxruntime", "torchcodec", "IPython",
    "DecayFitNet", "pyroomacoustics", "seaborn", "mat73", "tikzplotlib", "tqdm_stub",
)


class _Anything:
    """Callable / attribute sink: every access returns another sink."""

    def __init__(self, name="stub"):
        self.__name__ = name

    def __call__(self, *a, **k):
        return _Anything(self.__name__)

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(f"{self.__name__}.{item}")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(f"{self.__name__}.{item}")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _ABSENT_ROOTS:
            try:
                # real module wins when it is installed
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "diff_gfdn"))


_installed = False


def install():
    """Make ``import diff_gfdn`` resolve to the reference sources."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference sources not present at " + REFERENCE_SRC)
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)

    # spatial_sampling/config.py raises TypeError on python 3.10
    # (Optional[MLPConfig()]); the only names diff_gfdn needs are the enums.
    pkg = types.ModuleType("spatial_sampling")
    pkg.__path__ = [os.path.join(REFERENCE_SRC, "spatial_sampling")]
    sys.modules["spatial_sampling"] = pkg
    cfg = types.ModuleType("spatial_sampling.config")

    class BeamformerType(enum.Enum):
        MAX_DI = "max_di"
        MAX_RE = "max_re"
        BUTTER = "butter"

    class DNNType(enum.Enum):
        MLP = "mlp"
        CNN = "cnn"

    cfg.BeamformerType = BeamformerType
    cfg.DNNType = DNNType
    cfg.SpatialSamplingConfig = _Anything("SpatialSamplingConfig")
    sys.modules["spatial_sampling.config"] = cfg
    pkg.config = cfg
    _installed = True
