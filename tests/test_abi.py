"""CPU checks of the C-ABI boundary: the shared library builds/loads and exports every symbol that
include/diffgfdn_hip.h declares, the ctypes table covers the header one to one, and the product
path fails loudly without a GPU (no compute is launched here)."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "diffgfdn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfdn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from diffgfdn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "diffgfdn_amd", "csrc"), "-j4"])
    return _lib.load()


def test_header_and_ctypes_table_agree():
    from diffgfdn_amd import _lib
    assert _header_functions() == sorted(_lib.SIGNATURES), "ctypes table must mirror the header"


def test_library_exports_every_declared_symbol(lib):
    for name in _header_functions():
        assert hasattr(lib, name), name
    from diffgfdn_amd import _lib as _l
    assert lib.gfdn_abi_version() == _l.ABI_VERSION == 8


def test_host_side_queries(lib):
    assert lib.gfdn_stft_nframes(65537, 4096) == 32          # losses.py:512-535 at K = 65 537
    assert lib.gfdn_stft_nframes(100, 4096) == 0
    # 65 537 is a Fermat prime -> Rader on 2^16 points: chirp spectrum + two int32 index tables
    assert lib.gfdn_bluestein_table_bytes(65537) == 65536 * (8 + 4 + 4)
    assert lib.gfdn_bluestein_work_bytes(65537, 32) == 32 * 65536 * 8 + 32 * 64 * 4
    # 4097 = 17 * 241 -> Bluestein on 2^13 >= 4097 + 2048 points: chirp (n) + chirp spectrum (L)
    assert lib.gfdn_bluestein_table_bytes(4097) == (4097 + 8192) * 8
    assert lib.gfdn_bluestein_table_bytes(65536) == 0        # even length: not this transform
    assert lib.gfdn_solve_bwd_work_bytes(4, 4) == 2048 * 4 * (16 + 8) * 4    # n <= 4: 2048 partial-sum slots
    assert lib.gfdn_solve_bwd_work_bytes(2, 16) == 256 * 2 * (256 + 32) * 4
    assert lib.gfdn_irfft_pow2_work_bytes(131072, 2) == 2 * 131072 * 8
    # argument errors are reported before anything is launched
    assert lib.gfdn_solve_fwd(None, None, 0, 0, 0, None, None, None, None, 0, None, None) == -1
    assert lib.gfdn_irfft_odd_fwd(None, 9, None, 5, 1, None, 9, None, None) == -1


def test_product_path_is_loud_without_gpu():
    from diffgfdn_amd import hip_ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.irfft_odd_fwd(torch.zeros(1, 9, dtype=torch.complex64), 9)
    from diffgfdn_amd.losses import edr_loss
    with pytest.raises(RuntimeError):
        edr_loss(8000.0, win_size=64, hop_size=32)(torch.zeros(2, 257, dtype=torch.complex64),
                                                   torch.zeros(2, 257, dtype=torch.complex64))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "diffgfdn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_philox_restatement_known_answer():
    """tests/helpers.philox_mask (the checker of gfdn_draw_mask) reproduces the published
    Philox4x32-10 known-answer vector for an all-zero counter and key."""
    import numpy as np
    from tests.helpers import philox_mask
    m, count = philox_mask(0, 0, 128, 1.0)
    bits = (m > 0).astype(np.uint64).reshape(4, 32)
    words = [int((b << np.arange(32, dtype=np.uint64)).sum()) for b in bits]
    assert words == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert count == sum(bin(w).count("1") for w in words)
