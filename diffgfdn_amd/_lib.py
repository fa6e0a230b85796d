"""ctypes binding of the C-ABI shared library (include/diffgfdn_hip.h).

The library is built in-tree by ``make -C diffgfdn_amd/csrc`` (``__graft_entry__.build``) into
``diffgfdn_amd/lib/libdiffgfdn_hip.so``.  There is NO fallback: if the library is missing or a
call returns an error, a ``RuntimeError`` is raised -- the product path never silently runs
anything else.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdiffgfdn_hip.so")
ABI_VERSION = 8

_P = c_void_p

# name -> (restype, [argtypes])   -- mirrors include/diffgfdn_hip.h line by line
SIGNATURES = {
    "gfdn_abi_version": (c_int, []),
    "gfdn_zprep": (c_int, [_P, c_int, _P, _P, _P]),
    "gfdn_ortho_fwd": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "gfdn_ortho_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gfdn_solve_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "gfdn_solve_bwd_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_solve_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_precise_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "gfdn_solve_precise_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_absorb_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "gfdn_solve_absorb_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_phi_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_phi_bwd_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_solve_phi_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_phi_absorb_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_solve_phi_absorb_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          _P]),
    "gfdn_svf_coefficients": (c_int, [_P, _P, ctypes.c_double, c_int, c_int, _P, _P, _P]),
    "gfdn_sos_response": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_sos_compose_fwd": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P]),
    "gfdn_sos_compose_bwd_chunks": (c_int, [c_int, c_int, _P, _P]),
    "gfdn_sos_compose_bwd": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "gfdn_compose_fwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P]),
    "gfdn_compose_bwd_work_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfdn_compose_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "gfdn_compose_banded_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P]),
    "gfdn_compose_banded_bwd_work_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "gfdn_compose_banded_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "gfdn_compose_sh_fwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    "gfdn_compose_sh_bwd_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfdn_compose_sh_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_spectral_stats_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_spectral_stats": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "gfdn_colorless_terms": (c_int, [_P, c_int, _P, c_int, c_float, c_float, c_float, _P, _P, _P]),
    "gfdn_colorless_terms_banded": (c_int, [_P, c_int, c_int, _P, c_int, c_float, c_float, c_float, _P, _P, _P]),
    "gfdn_weighted_sums_banded": (c_int, [_P, c_int, _P, _P, c_float, _P, c_float, c_int, c_int, _P, _P]),
    "gfdn_mlp_gains_banded_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_mlp_gains_banded_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_mlp_gains_banded_bwd_parts": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "gfdn_mlp_gains_banded_bwd_parts_scaled": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                                       _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "gfdn_tf_gain_chunks": (c_int, [c_int]),
    "gfdn_mlp_bwd_takes_parts": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "gfdn_mlp_bands_sizes": (c_int, [c_int, c_int, c_int, _P, _P, c_int, _P]),
    "gfdn_mlp_gains_bands_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_mlp_gains_bands_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_float, c_float, _P, _P, _P, _P,
                                         c_int, _P, _P, _P, _P]),
    "gfdn_subfdn_normalize_work_bytes": (c_size_t, [c_int]),
    "gfdn_subfdn_normalize": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_subfdn_colorless_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_subfdn_colorless_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "gfdn_spectral_stats_binmajor": (c_int, [_P, c_int, c_int, _P, c_int, c_float, _P, _P, _P, _P]),
    "gfdn_subfdn_colorless_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_tf_coefs_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    "gfdn_tf_coefs_fwd2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "gfdn_tf_ortho_coefs": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gfdn_tf_coefs_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gfdn_tf_param_grads": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_tf_rows_sum": (c_int, [_P, c_int, c_int, _P, _P]),
    "gfdn_tf_parts": (c_int, [c_int, c_int]),
    "gfdn_tf_work_bytes": (c_size_t, [c_int]),
    "gfdn_tf_gpart_bytes": (c_size_t, [c_int]),
    "gfdn_tf_eval": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "gfdn_tf_energy": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_double, _P]),
    "gfdn_tf_energy_gains": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_double, _P, _P, c_int, c_int,
                                     _P]),
    "gfdn_tf_colorless": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_int, c_float, _P, _P, _P, c_double, _P]),
    "gfdn_tf_compose_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P,
                                    _P]),
    "gfdn_tf_compose_parts": (c_int, [c_int]),
    "gfdn_tf_compose_bwd_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfdn_tf_compose_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P,
                                    c_int, _P]),
    "gfdn_tf_tail": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                             _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P]),
    "gfdn_irfft_odd_pairs_bwd_tslots3": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_pairs_fwd_scaled": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, c_int, _P, c_int, _P]),
    "gfdn_edc_lin_one_max_len": (c_int, []),
    "gfdn_edc_lin_one": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P,
                                 c_int, c_float, c_float, _P, _P, c_int, _P, c_int, c_int, _P]),
    "gfdn_lin_merge_slots": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "gfdn_f64_fft_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_irfft_odd_f64_length": (c_int, [c_int]),
    "gfdn_rfft_pow2_f64": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_f64_plan": (c_int, [c_int, _P, _P, _P]),
    "gfdn_irfft_odd_f64": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "gfdn_lin_gamma_win": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, c_int,
                                   _P]),
    "gfdn_tf_gain_grad_work_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfdn_tf_gain_grad": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P, _P]),
    "gfdn_tf8_coefs": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "gfdn_tf8_parts": (c_int, [c_int]),
    "gfdn_tf8_part_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_tf8_energy": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_double, _P]),
    "gfdn_tf8_tsave": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "gfdn_tf8_colorless": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P, _P, _P, c_double, _P]),
    "gfdn_tf8_compose_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P,
                                     _P, _P, _P]),
    "gfdn_tf8_param_grads_work_bytes": (c_size_t, [c_int]),
    "gfdn_tf8_param_grads": (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                     _P]),
    "gfdn_tfp_parts": (c_int, []),
    "gfdn_tfp_forward": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, c_int, _P, _P]),
    "gfdn_tfp_energy": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "gfdn_tfp_colorless": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_float, c_int, _P, _P, c_int, _P, _P, _P,
                                   _P, _P]),
    "gfdn_tfp_compose_bwd": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int,
                                     _P, c_int, _P, c_int, _P, _P, _P]),
    "gfdn_tf8_tail": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                              _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_tf9_coefs": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "gfdn_tf9_rec_grads_work_bytes": (c_size_t, [c_int]),
    "gfdn_tf9_rec_grads": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "gfdn_tfp_ratio_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_tfp_ratio_bwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "gfdn_ortho_bwd_add": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "gfdn_exp_contract_mfma": (c_int, [_P, c_int, _P, _P, c_int, _P, _P]),
    "gfdn_weighted_sums": (c_int, [_P, c_int, _P, _P, c_float, _P, c_float, c_int, _P, _P]),
    "gfdn_normalize_io": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "gfdn_bluestein_table_bytes": (c_size_t, [c_int]),
    "gfdn_bluestein_table_init": (c_int, [c_int, _P]),
    "gfdn_bluestein_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_irfft_odd_fwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_bwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_slot_order": (c_int, [c_int, _P, _P]),
    "gfdn_irfft_odd_slots_fwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_slots_bwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_pairs_fwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_pairs_compose_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P,
                                                 _P, c_int, _P, c_int, _P]),
    "gfdn_irfft_odd_pairs_bwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_odd_pairs_gains_parts": (c_int, [c_int]),
    "gfdn_irfft_odd_pairs_gains_bwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P,
                                               _P, c_int, _P]),
    "gfdn_stft_power_pairs": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_stft_power_pairs_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gfdn_stft_power_pairs_bwd_phase": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P]),
    "gfdn_edc_loss_pairs": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_irfft_odd_stages": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, _P]),
    "gfdn_irfft_pow2_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_irfft_pow2_fwd": (c_int, [c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_irfft_pow2_fwd_band": (c_int, [c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P]),
    "gfdn_irfft_pow2_bwd": (c_int, [c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_rownorm_fwd": (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    "gfdn_rownorm_bwd": (c_int, [_P, c_int, c_int, c_float, _P, _P, _P]),
    "gfdn_irfft_pow2_bwd_window": (c_int, [c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_edc_mixed_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfdn_edc_loss_model_mixed": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P,
                                          c_float, c_float, _P, _P, _P, _P]),
    "gfdn_group_sums_fwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_group_sums_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "gfdn_dirlin_tiles": (c_int, [c_int]),
    "gfdn_dirlin_line_tiles": (c_int, [c_int]),
    "gfdn_dirlin_lines_fwd": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P]),
    "gfdn_dirlin_lines_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "gfdn_dirlin_combine": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "gfdn_dirlin_gamma_dots": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P]),
    "gfdn_edc_loss_model_mixed_stages": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P,
                                                 c_float, c_float, _P, _P, _P, c_int, _P]),
    "gfdn_rfft_pow2": (c_int, [c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_sh_to_directional": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "gfdn_stft_nframes": (c_int, [c_int, c_int]),
    "gfdn_stft_power": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_stft_power_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_edr_work_bytes": (c_size_t, [c_int, c_int]),
    "gfdn_edr_target": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_edr_loss": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P]),
    "gfdn_edc_work_bytes": (c_size_t, [c_int]),
    "gfdn_edc_target": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gfdn_mlp_param_count": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfdn_mlp_bwd_work_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "gfdn_mlp_gains_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_mlp_gains_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "gfdn_pick_rows": (c_int, [_P, _P, _P, c_int, _P]),
    "gfdn_adam_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, _P]),
    "gfdn_adam_step_counted": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, _P, _P]),
    "gfdn_adam_step_mirrored": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, _P, _P]),
    "gfdn_edc_loss": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_edc_loss_model": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_float, c_float, _P, _P, _P, _P]),
    "gfdn_draw_mask": (c_int, [ctypes.c_ulonglong, _P, c_int, c_float, _P, _P]),
    "gfdn_lin_gain_chunks": (c_int, [c_int]),
    "gfdn_lin_combine_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "gfdn_lin_gamma": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, _P]),
    "gfdn_lin_gamma_dots_tiles": (c_int, [c_int]),
    "gfdn_lin_gamma_dots": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_int,
                                    c_int, c_int, _P, _P, _P]),
    "gfdn_stft_pairs_spectrum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "gfdn_stft_pairs_spectrum_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "gfdn_edr_lin_parts": (c_int, [c_int]),
    "gfdn_edr_lin_loss": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, c_int,
                                  c_int, c_int, _P]),
    "gfdn_edr_lin_gsum": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P]),
    "gfdn_edr_lin_loss_gsum": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, _P, c_int, _P,
                                       c_int, c_int, _P, c_int, c_int, _P]),
    "gfdn_edr_lin_band_parts": (c_int, [c_int]),
    "gfdn_edc_loss_pairs_lin": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int,
                                        _P, _P, c_int, c_float, c_float, _P, _P, c_int, _P, _P, _P]),
    "gfdn_irfft_odd_time_slots": (c_int, [c_int, _P]),
    "gfdn_irfft_odd_pairs_bwd_tslots": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, _P]),
    "gfdn_lin_gain_dots": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "gfdn_edc_loss_banded": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_float, c_float,
                                     _P, _P, _P, _P]),
    "gfdn_edc_loss_pairs_banded": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_float,
                                           c_float, _P, _P, _P, _P]),
    "gfdn_draw_mask_banded": (c_int, [ctypes.c_ulonglong, _P, _P, c_int, c_int, c_int, c_float, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once) and attach the signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"diffgfdn_amd: HIP library not built ({LIB_PATH}); run `make -C diffgfdn_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    # PyTorch first: it brings its own copy of the HIP runtime, and the library must bind to THAT copy (the streams and
    # device pointers it is handed come from it).  Loaded the other way round -- __graft_entry__.build() followed by smoke()
    # in one process did -- the library pulls in the system's runtime, torch then loads a second one, and every launch
    # returns hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.gfdn_abi_version()
    if ver != ABI_VERSION:
        raise RuntimeError(f"diffgfdn_amd: ABI version mismatch (library {ver}, python {ABI_VERSION})")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported size"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"diffgfdn_amd: {what} failed: {kind}")
