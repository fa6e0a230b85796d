"""diffgfdn_amd -- MI355X-native hot path of DiffGFDN (frequency-sampled GFDN + EDR/EDC losses).

Host code is Python on PyTorch-ROCm; the arithmetic runs in hand-written HIP kernels for gfx950
behind the C ABI of ``include/diffgfdn_hip.h`` (``diffgfdn_amd/lib/libdiffgfdn_hip.so``).
"""
__version__ = "0.1.0"
