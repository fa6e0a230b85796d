/*
 * diffgfdn_hip.h -- C ABI of the MI355X (gfx950) hot path of DiffGFDN.
 *
 * The reference (orchidas/DiffGFDN) is pure Python and has no FFI of its own: the path below
 * sits behind torch.nn.Module classes.  This header is therefore the boundary a maintainer
 * would bind with ctypes (INTEGRATION.md shows the stub); each entry point cites the
 * reference code it replaces (paths relative to the reference's src/diff_gfdn/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; PyTorch (or any caller) owns all
 *     buffers, the library never allocates or frees device memory and keeps no global state;
 *   - every launch is asynchronous on the caller's hipStream_t (passed as void*); the only
 *     synchronous call is gfdn_bluestein_table_init (plan creation);
 *   - complex values are interleaved float pairs (re, im) = torch.complex64; "c128" marks
 *     interleaved doubles = torch.complex128;
 *   - return value: 0 on success, a hipError_t (> 0) from the runtime, or a negative
 *     GFDN_E_* code for argument errors.  Nothing is launched when an error is returned.
 *   - matrices are row-major; "bin-major" means index [k][n] with the delay line fastest.
 */
#ifndef DIFFGFDN_HIP_H
#define DIFFGFDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the gfdn_tf_* block-transfer-function entry points, the transforms with the output stage folded in, the device-side
 * receiver schedule; every entry point of version 1 keeps its signature */
#define GFDN_ABI_VERSION 8
#define GFDN_E_BADARG (-1)
#define GFDN_E_UNSUPPORTED (-2)
#define GFDN_MAX_BLOCK 32      /* largest dense block the per-bin solver takes        */
#define GFDN_MAX_GROUPS 8      /* largest number of groups the compose kernels take   */
#define GFDN_PARTIAL_BLOCKS 256 /* per-launch partial-sum slots of the reducing kernels */

int gfdn_abi_version(void);

/* ---- frequency grid --------------------------------------------------------------------
 * z (K complex128, dataloader.py:552-566 `z_values`) -> turns[k] = arg(z_k)/2pi and
 * logr[k] = ln|z_k| in float64, so that z^m is evaluated with an exactly reduced phase
 * (feedback_loop.py:330 does z**delays in complex128).                                   */
int gfdn_zprep(const double* z_c128, int K, double* turns, double* logr, void* stream);

/* ---- orthogonal feedback-matrix parameterisation  (feedback_loop.py:16-36, :270, :393-404) -----
 * M (G, n, n) raw parameters.  X_g = triu(M_g,1) - triu(M_g,1)^T,  Q_g = expm(X_g) -> Q (G,n,n),
 * QQ_g = Q_g Q_g (the diagonal blocks of the feedback matrix under zero coupling) -> QQ (G,n,n).
 * Either output may be NULL.  Backward: gQ / gQQ (either may be NULL) -> gM (G,n,n); Q: the
 * forward's Q when the caller still holds it (saves recomputing the n x n exponential) or NULL. */
int gfdn_ortho_fwd(const float* M, int G, int n, float* Q, float* QQ, void* stream);
int gfdn_ortho_bwd(const float* M, int G, int n, const float* gQ, const float* gQQ, const float* Q,
                   float* gM, void* stream);
/* as gfdn_ortho_bwd, plus gM_add (G,n,n) or NULL added to the result: a gradient that reaches M directly
 * (the sub-FDNs of the colorless loss use the raw M_g, model.py:237-240).                               */
int gfdn_ortho_bwd_add(const float* M, int G, int n, const float* gQ, const float* gQQ, const float* Q,
                       const float* gM_add, float* gM, void* stream);

/* ---- per-bin resolvent solve  (feedback_loop.py:326-391, model.py:237-240, :615-619) ---
 * For every bin k and diagonal block q (nblk blocks of size nper, N = nblk*nper):
 *     T_k = diag(z_k^{m_i} * inv_gamma_i) - A_q            (transpose = 0)
 *     T_k = diag(z_k^{m_i} * inv_gamma_i) - A_q^T          (transpose = 1, model.py:1083)
 *     Y[k][q*nper + i] = (T_k^{-1} b)_i
 * A: (nblk, nper, nper) real; zero inter-group coupling -> nblk = G blocks (Q_g^2 for the
 * damped loop, raw M_g for the sub-FDNs); coupled feedback matrix -> nblk = 1, nper = N.
 * Replaces torch.linalg.inv + the einsums: the inverse is never formed.
 * logr may be NULL (unit circle).                                                         */
int gfdn_solve_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                   const float* A, const float* delays, const float* inv_gamma,
                   const float* b, int transpose, float* Y_c64, void* stream);

/* Backward of gfdn_solve_fwd: given gY (K, N) complex64 (= dL/dRe + i dL/dIm of Y) returns
 *   gA (nblk,nper,nper), gb (N), ginv_gamma (N).
 * Y: the forward solution of gfdn_solve_fwd for the same arguments when the caller still holds it
 * (the kernel then skips re-solving T y = b), or NULL.
 * work: gfdn_solve_bwd_work_bytes(nblk, nper) bytes of scratch.  Sums over bins are formed
 * in a fixed order (per-block partials + a second pass): results are bitwise reproducible. */
size_t gfdn_solve_bwd_work_bytes(int nblk, int nper);
int gfdn_solve_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                   const float* A, const float* delays, const float* inv_gamma,
                   const float* b, int transpose, const float* gY_c64, const float* Y_c64,
                   float* gA, float* gb, float* ginv_gamma, void* work, void* stream);

/* "Precise" variants of gfdn_solve_fwd / _bwd: matrix entries (z^m / gamma in float64 from the exactly reduced phase)
 * and the elimination in float64, results rounded to complex64 / float32 at the end -- what the reference's
 * complex128 torch.linalg.inv (feedback_loop.py:389-391) delivers.  For nearly lossless loops (T60 of tens of
 * seconds: condition numbers of 1e3..1e5, where float32 entries no longer hold 1e-4).  The inverse gains come in
 * FLOAT64 (N): at pole radii of 0.9999 the float32 rounding of gamma alone moves the resonance peaks.  Same
 * other arguments and work size; lane-parallel kernels only (no thread-per-system shortcut).          */
int gfdn_solve_precise_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                           const float* A, const float* delays, const double* inv_gamma_f64, const float* b,
                           int transpose, float* Y_c64, void* stream);
int gfdn_solve_precise_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                           const float* A, const float* delays, const double* inv_gamma_f64, const float* b,
                           int transpose, const float* gY_c64, const float* Y_c64, float* gA, float* gb,
                           float* ginv_gamma, void* work, void* stream);

/* Frequency-dependent absorption (feedback_loop.py:332-344, :376-381: Gamma(z) = diag of per-line filter
 * responses, GEQ / Prony designs of absorption_filters.py): T_k = diag(z_k^{m_i} / Gamma_i(z_k)) - A.
 * inv_gamma_bins (K, N) complex64 = 1 / Gamma_i(z_k); ones (N) = device vector of ones.  The filters are
 * fixed (not learnable in the reference either): no gradient w.r.t. them.                          */
int gfdn_solve_absorb_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                          const float* A, const float* delays, const float* ones,
                          const float* inv_gamma_bins_c64, const float* b, int transpose, float* Y_c64,
                          void* stream);
int gfdn_solve_absorb_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                          const float* A, const float* delays, const float* ones,
                          const float* inv_gamma_bins_c64, const float* b, int transpose,
                          const float* gY_c64, const float* Y_c64, float* gA, float* gb,
                          float* ginv_scratch, void* work, void* stream);

/* FILTER coupling (feedback_loop.py:90-143, :362-373, :441-455): A(z_k)[i][j] = BM[i][j] Phi_k[g(i)][g(j)] with the
 * paraunitary FIR coupling matrix evaluated per bin, Phi (K, G, G) complex64, and the real block mixing matrix
 * BM (N, N), N = G * nper <= 32.  Backward: gBM (N, N), gb (N), ginv_gamma (N), gPhi (K, G, G) complex64
 * (per bin); work: gfdn_solve_phi_bwd_work_bytes(G, nper).                                          */
int gfdn_solve_phi_fwd(const double* turns, const double* logr, int K, int G, int nper, const float* BM,
                       const float* Phi_c64, const float* delays, const float* inv_gamma, const float* b,
                       float* Y_c64, void* stream);
size_t gfdn_solve_phi_bwd_work_bytes(int G, int nper);
int gfdn_solve_phi_bwd(const double* turns, const double* logr, int K, int G, int nper, const float* BM,
                       const float* Phi_c64, const float* delays, const float* inv_gamma, const float* b,
                       const float* gY_c64, const float* Y_c64, float* gBM, float* gb, float* ginv_gamma,
                       float* gPhi_c64, void* work, void* stream);
/* The same systems with absorption FILTERS on the delay lines (feedback_loop.py:332-344, :376-381 together with the
 * FILTER coupling of :362-373 -- the reference's forward handles both in one pass): the diagonal entry of line i at bin
 * k is z_k^{m_i} inv_gamma[i] inv_gamma_bins[k][i], inv_gamma_bins (K, N) complex64 = 1 / Gamma_i(z_k) (NULL: the
 * entry points above).  The filters are fixed designs: ginv_gamma is the gradient w.r.t. the REAL factor inv_gamma.   */
int gfdn_solve_phi_absorb_fwd(const double* turns, const double* logr, int K, int G, int nper, const float* BM,
                              const float* Phi_c64, const float* delays, const float* inv_gamma,
                              const float* inv_gamma_bins_c64, const float* b, float* Y_c64, void* stream);
int gfdn_solve_phi_absorb_bwd(const double* turns, const double* logr, int K, int G, int nper, const float* BM,
                              const float* Phi_c64, const float* delays, const float* inv_gamma,
                              const float* inv_gamma_bins_c64, const float* b, const float* gY_c64,
                              const float* Y_c64, float* gBM, float* gb, float* ginv_gamma, float* gPhi_c64,
                              void* work, void* stream);

/* ---- second-order-section cascades: SVF output filters  (gain_filters.py:221-241 SOSFilter.forward, :262-402
 * SVF_from_MLP, model.py:588-619) -------------------------------------------------------------------------
 * coef (R, S, 6) float32 = b0 b1 b2 a0 a1 a2 per section (S <= 12), z (K) complex128.  Sections are evaluated in
 * float64 and rounded to complex64, the running product is complex64.
 *   gfdn_sos_response:    out[r][k] = prod_s (b0 + b1 z^-1 + b2 z^-2) / (a0 + a1 z^-1 + a2 z^-2)   (R, K) c64
 *   gfdn_sos_compose_fwd: H[b][k] = sum_g response_{b G + g}(z_k) T[k][g] + direct[b][k]   (rows receiver-major,
 *                         G <= 8; T (K, G) c64 = group transfer functions; direct (B, ldd) c64 or NULL)
 *   gfdn_sos_compose_bwd: gT_partial (t_chunks, K, G) c64 and gcoef_partial (B G, c_chunks, S, 6) f32, to be
 *                         summed over their chunk axis (chunk counts from gfdn_sos_compose_bwd_chunks)          */
/* SVF parameters -> biquad coefficients (gain_filters.py:327-330, :36-103, :117-151): raw (R, S, 2) unconstrained
 * [resonance, gain] per section, cutoff (S) float64 normalised cut-offs; gcoef == NULL: out = coef (R, S, 6);
 * else out = d<gcoef, coef>/draw (R, S, 2).                                                                  */
int gfdn_svf_coefficients(const float* raw, const double* cutoff, double compress_pole_factor, int R, int S,
                          const float* gcoef, float* out, void* stream);
int gfdn_sos_response(const float* coef, int R, int S, const double* z_c128, int K, float* out_c64, void* stream);
int gfdn_sos_compose_fwd(const float* coef, int B, int G, int S, const double* z_c128, int K, const float* T_c64,
                         const float* direct_c64, int ldd, float* H_c64, void* stream);
int gfdn_sos_compose_bwd_chunks(int B, int K, int* t_chunks, int* c_chunks);
int gfdn_sos_compose_bwd(const float* coef, int B, int G, int S, const double* z_c128, int K, const float* T_c64,
                         const float* gH_c64, float* gT_partial_c64, float* gcoef_partial, void* stream);

/* ---- output stage  (model.py:583-619, gain_filters.py:526-534, trainer.py:459) ----------
 *   S[g][k]  = sum_{n in group g} c_n Y[k][n]
 *   H[b][k]  = (sum_g rgain[b][g] S[g][k] + direct[b][k]) * filt[k]
 * rgain: (B, G) receiver gains (MLP output, or eye(G) to obtain the sub-FDN responses
 * model.py:243-250); direct (B, ldd) complex64 or NULL; filt (K) complex64 or NULL;
 * S_out (G, K) complex64 or NULL.
 * Row indirection (here and in gfdn_edr_loss / gfdn_edc_loss / gfdn_mlp_gains_*): when the int64
 * device array `*_rows` (B entries) is not NULL, item b reads row rows[b] of a dataset-level
 * store (all receivers) instead of row b of a gathered batch -- the gather of custom_collate
 * (dataloader.py:674-704) never materialises.  The caller guarantees 0 <= rows[b] < store rows. */
int gfdn_compose_fwd(const float* Y_c64, int K, int G, int nper, const float* c,
                     const float* rgain, int B, const float* direct_c64, int ldd,
                     const long long* direct_rows, const float* filt_c64, float* H_c64, int ldh,
                     float* S_out_c64, void* stream);

/* Backward: gH (B, ldh) complex64 (gradient w.r.t. the FILTERED H when filt != NULL) ->
 *   gY (K, N) complex64, gc (N), grgain (B, G).   work: gfdn_compose_bwd_work_bytes(). */
size_t gfdn_compose_bwd_work_bytes(int K, int G, int nper, int B);
int gfdn_compose_bwd(const float* Y_c64, int K, int G, int nper, const float* c,
                     const float* rgain, int B, const float* filt_c64, const float* gH_c64,
                     int ldh, float* gY_c64, float* gc, float* grgain, void* work, void* stream);

/* ---- band-stacked variants  (run_subband_training_treble.py:175-204: one independent model per
 * octave band, trained one after another by the reference).  Here `nbands` models of identical
 * shape are stepped by ONE launch each: their delay lines sit side by side in one solve
 * (gfdn_solve_* with nblk = nbands * G blocks, Y (K, nbands * N)), items are band-major
 * (item i = band * B + b, B items per band), per-band arrays are stacked along the first axis:
 * c (nbands, N), rgain (nbands * B, G), filt (nbands, ldf) complex64, S_out (nbands * G, K),
 * gc (nbands, N), grgain (nbands * B, G).  nbands = 1 is the plain entry point.
 * direct / direct_rows: item i reads row direct_rows[i] of ONE store (all bands' receivers). */
int gfdn_compose_banded_fwd(const float* Y_c64, int K, int nbands, int G, int nper, const float* c,
                            const float* rgain, int B, const float* direct_c64, int ldd,
                            const long long* direct_rows, const float* filt_c64, int ldf,
                            float* H_c64, int ldh, float* S_out_c64, void* stream);
size_t gfdn_compose_banded_bwd_work_bytes(int K, int nbands, int G, int nper, int B);
int gfdn_compose_banded_bwd(const float* Y_c64, int K, int nbands, int G, int nper, const float* c,
                            const float* rgain, int B, const float* filt_c64, int ldf,
                            const float* gH_c64, int ldh, float* gY_c64, float* gc, float* grgain,
                            void* work, void* stream);

/* Directional output stage (model.py:1056-1088): H_sh[b][l][k] = sum_g w[b][g][l] c_{g,l} Y[k][g*nper+l]
 * and its backward.                                                                       */
int gfdn_compose_sh_fwd(const float* Y_c64, int K, int G, int nper, const float* c,
                        const float* w, int B, const float* filt_c64, float* H_c64,
                        void* stream);
size_t gfdn_compose_sh_bwd_work_bytes(int G, int nper, int B);
int gfdn_compose_sh_bwd(const float* Y_c64, int K, int G, int nper, const float* c,
                        const float* w, int B, const float* filt_c64, const float* gH_c64,
                        float* gY_c64, float* gc, float* gw, void* work, void* stream);

/* SH-domain -> directional responses (trainer.py:853-865, einsum 'jl,blk->bjk'):
 * adjoint = 0: in = H_sh (B, C, K) -> out = H_dir (B, J, K);  adjoint = 1: in = gH_dir -> out = gH_sh.
 * A: (J, C) real analysis matrix.                                                              */
int gfdn_sh_to_directional(const float* A, int J, int C, int K, int B, const float* in_c64,
                           float* out_c64, int adjoint, void* stream);

/* ---- colorless statistics of the sub-FDN responses  (colorless_fdn/losses.py:20-73,
 * trainer.py:323-324).  S (G, K) complex64.  Per group g:
 *   energy[g] = mean_k |S|^2 ;  loss[g] = mean_k (|S|-1)^p, p = 2, or (asym) 4 where |S|-1 > 1.
 * gS (G, K) complex64 = scale * dloss[g]/dS, or NULL.                                     */
size_t gfdn_spectral_stats_work_bytes(int G, int K);
int gfdn_spectral_stats(const float* S_c64, int G, int K, int asym, float scale,
                        float* energy, float* loss, float* gS_c64, void* work, void* stream);

/* Scalar loss bookkeeping of one step (trainer.py:298-313): loss_g (G) per-group spectral losses,
 * Q (G,n,n) rotations.  out3 = { (w_spec sum_g loss_g + sparsity) * inv_world, w_spec sum_g loss_g,
 * sparsity } with sparsity = w_sparse * sparsity_loss(Q[G-1]) (colorless_fdn/losses.py:7-17; only the
 * last group counts, trainer.py:305-308).  gQ (G,n,n), optional: d out3[0] / dQ.              */
int gfdn_colorless_terms(const float* loss_g, int G, const float* Q, int n, float w_spec,
                         float w_sparse, float inv_world, float* out3, float* gQ, void* stream);
/* band-stacked: loss_g (nbands * G), Q (nbands * G, n, n) -> out3 (nbands, 3), gQ (nbands * G, n, n);
 * the sparsity term of band q is taken on ITS last group.                                      */
int gfdn_colorless_terms_banded(const float* loss_g, int nbands, int G, const float* Q, int n,
                                float w_spec, float w_sparse, float inv_world, float* out3, float* gQ,
                                void* stream);
/* out3 = { wa sum(a) + wb sum(b), wa sum(a), wb sum(b) } over n items (a or b may be NULL).
 * Item i of a is the sum of its a_cols partials (a: (n, a_cols), a_cols = 1 for plain items),
 * divided by a_div[a_rows ? a_rows[i] : i] when a_div != NULL -- the deferred form of
 * gfdn_edr_loss (loss_item = NULL) hands its per-tile partials and normalisers straight here. */
int gfdn_weighted_sums(const float* a, int a_cols, const float* a_div, const long long* a_rows,
                       float wa, const float* b, float wb, int n, float* out3, void* stream);

/* band-stacked: n items per band, nbands * n items in all -> out3 (nbands, 3). */
int gfdn_weighted_sums_banded(const float* a, int a_cols, const float* a_div, const long long* a_rows,
                              float wa, const float* b, float wb, int n, int nbands, float* out3,
                              void* stream);

/* Energy normalisation of the input / output gains (trainer.py:317-332): for n in group g,
 * b[n] /= energy[g]^(1/4), c[n] /= energy[g]^(1/4), in place (float32, N = G * nper).        */
int gfdn_normalize_io(const float* energy, float* b, float* c, int G, int nper, void* stream);

/* The whole of Trainer.normalize (trainer.py:317-332) in two launches, without writing the sub-FDN
 * responses: E_g = mean_k |sum_{n in g} c_n y_n[k]|^2 with y = (diag(z_k^m) - M_g)^{-1} b_g
 * (model.py:237-250: raw M_g (G, nper, nper), no absorption), then b_n, c_n /= E_g^(1/4) in place.
 * energy (G), optional: E_g as used.  work: gfdn_subfdn_normalize_work_bytes(G).                 */
size_t gfdn_subfdn_normalize_work_bytes(int G);
int gfdn_subfdn_normalize(const double* turns, const double* logr, int K, int G, int nper,
                          const float* M, const float* delays, float* b, float* c, float* energy,
                          void* work, void* stream);

/* ---- colorless side branch, fused (nper <= 4, G <= 64 blocks; bands stack as more blocks)  --------
 * Trainer.normalize and the spectral loss of the step that follows it read the SAME un-damped sub-FDN
 * responses (model.py:209-252) up to a per-group constant: y = (D - M_g)^-1 b_g is linear in b, so after
 * b, c /= d_g (d_g = E_g^(1/4), trainer.py:323-332) the responses are y / d_g and Hout[:, g] / d_g^2.
 *   fwd  : Y (K, G*nper) complex64 raw responses, S (K, G) complex64 raw group sums sum_{n in g} c_n y_n
 *          (BIN-MAJOR), energy (G) = mean_k |S[k][g]|^2; normalize != 0: b, c /= energy^(1/4) in place.
 *   stats: loss (G) = mean_k (|S'| - 1)^p (p as gfdn_spectral_stats) on S' = S / sqrt(energy_g)
 *          (energy == NULL: S' = S); gS (K, G) complex64 = scale * dloss_g / dS', or NULL.
 *   bwd  : with the CURRENT (rescaled) b, c and the energy the forward returned (NULL if it did not
 *          normalize): gM (G, nper, nper), gb (N), gc (N) of sum_g <gS[:, g], S'[:, g]>.
 * work: gfdn_subfdn_colorless_work_bytes(G, nper) for each call.                                     */
size_t gfdn_subfdn_colorless_work_bytes(int G, int nper);
int gfdn_subfdn_colorless_fwd(const double* turns, const double* logr, int K, int G, int nper,
                              const float* M, const float* delays, float* b, float* c, int normalize,
                              float* Y_c64, float* S_c64, float* energy, void* work, void* stream);
int gfdn_spectral_stats_binmajor(const float* S_c64, int G, int K, const float* energy, int asym,
                                 float scale, float* loss, float* gS_c64, void* work, void* stream);
int gfdn_subfdn_colorless_bwd(const double* turns, const double* logr, int K, int G, int nper,
                              const float* M, const float* delays, const float* b, const float* c,
                              const float* energy, const float* Y_c64, const float* gS_c64, float* gM,
                              float* gb, float* gc, void* work, void* stream);

/* ---- block transfer functions in polynomial form (nper <= 4, zero coupling)  ---------------------------
 * With zero inter-group coupling (feedback_loop.py:298-303, :439-443) a receiver sees the loop only through
 *     T_g(z) = c_g^T (D_g(z) Gamma_g^-1 - A_g)^-1 b_g      (model.py:583-619; sub-FDNs: model.py:237-250),
 * which for a block of n <= 4 lines is a ratio of multilinear polynomials in the phasors z^{m_i}:
 *     T = (sum_S P_S e_S) / (sum_S Q_S e_S),   e_S = prod_{i in S} z^{m_i},  S subset of the block's lines,
 *     Q_S = (-1)^{|S^c|} det A[S^c, S^c] prod_{i in S} 1/gamma_i,
 *     P_S = ((-1)^{|S^c|} det (A - b c^T)[S^c, S^c]) prod_{i in S} 1/gamma_i - Q_S   (matrix determinant lemma).
 * coef (nblk, 32) float32 = [P_S (16) | Q_S (16)], S a bit mask (bit i = line i of the block).  The delay-line
 * responses (K, N) of gfdn_solve_* are never formed; dL/d(A, b, c) goes through dL/dcoef (30 sums over the bins
 * per block) and a bin-independent float64 map per block.
 *   coefs_fwd : A (nblk, nper, nper), b, c (nblk*nper), inv_gamma (nblk*nper) or NULL (ones) -> coef;
 *               coefs_fwd2: two record sets sharing b, c in one launch.
 *   eval      : T (K, nblk) complex64 BIN-MAJOR = scale_blk * T_blk(z_k)  (scale (nblk) or NULL).
 *   energy    : Trainer.normalize (trainer.py:317-332) from the records of the sub-FDNs: energy (nblk) =
 *               mean_k |T|^2 (optional), scale (nblk) = energy^(-1/2) (optional: what T -- and the numerator
 *               records -- scale by once b, c are divided by energy^(1/4)), b, c (both or neither) rescaled in
 *               place.  phase: 1 = the pass over the bins only (partial sums into work), 2 = the finish only
 *               (energy, scale, rescale -- the only part that writes b, c), 3 = both.  work: gfdn_tf_work_bytes(nblk).
 *               dturn (energy, colorless): != 0 asserts a uniform grid on the unit circle, turns[k] = turns[0] +
 *               k dturn (logr NULL) -- the passes then step the phasors through runs of 8 bins by constant
 *               rotations instead of one exact evaluation per bin; 0: any grid.
 *   colorless : spectral loss of the sub-FDNs (colorless_fdn/losses.py:20-73, trainer.py:298-304) on
 *               S' = scale T: loss (nblk, optional) = mean_k (|S'| - 1)^p per block and the gradient record
 *               grec (nblk, 32) of L = gscale * sum_blk loss_blk: entry S < 15 dL/dP'_S (P' = scale P), 16 + S
 *               dL/dQ_S (entry 15 holds loss_blk).  work: gfdn_tf_gpart_bytes(nblk).
 *   compose   : H[b][k] = (sum_g rgain[b][g] scale_g T_g(z_k) + direct[rows[b]][k]) filt[k], band-stacked as
 *               gfdn_compose_banded_* (blocks band*G + g, items band*B + b); Tsave (nbands*G, K) complex64,
 *               optional: the scaled, unfiltered scale_g T_g(z_k), which the backward takes back.  The backward
 *               is two streaming passes over dL/dH: gain_grad -> grgain (nbands*B, G), compose_bwd -> the gradient
 *               records grec (nbands*G, 32) of the SCALED records.  G <= 4.  work: the *_work_bytes queries.
 *   coefs_bwd : maps gradient records to dL/dA, dL/db, dL/dc for up to two record sets sharing b, c (set 0:
 *               damped loop, A0 = Q Q; set 1: sub-FDNs, A1 = raw M, or A1 = NULL): gA0, gA1 (nblk, nper, nper),
 *               gb, gc (nblk*nper) = the sum over the sets.  b, c: the values the records' gradients refer to.  */
int gfdn_tf_coefs_fwd(const float* A, const float* b, const float* c, const float* inv_gamma, int nblk,
                      int nper, float* coef, void* stream);
int gfdn_tf_coefs_fwd2(const float* A0, const float* inv_gamma0, float* coef0, const float* A1,
                       const float* inv_gamma1, float* coef1, const float* b, const float* c, int nblk, int nper,
                       void* stream);
/* Head of the step in one launch (one workgroup per block): Q = expm(skew(M)), QQ = Q Q as gfdn_ortho_fwd, then the
 * records of (QQ, inv_gamma) -> coef and, with coef_sub != NULL, of (M, gamma = 1) -> coef_sub as gfdn_tf_coefs_fwd2.
 * M, Q, QQ (nblk, nper, nper).  Bit-identical to the separate launches.  */
int gfdn_tf_ortho_coefs(const float* M, const float* inv_gamma, const float* b, const float* c, int nblk, int nper,
                        float* Q, float* QQ, float* coef, float* coef_sub, void* stream);
int gfdn_tf_coefs_bwd(const float* A0, const float* inv_gamma0, const float* grec0, const float* A1,
                      const float* inv_gamma1, const float* grec1, const float* b, const float* c, int nblk,
                      int nper, float* gA0, float* gA1, float* gb, float* gc, void* stream);
/* The tail of the backward in one launch (one workgroup per block): [sum of the partial record rows the output-stage
 * adjoint left in its work buffer: grec0 = that buffer, nparts0 = gfdn_tf_compose_parts(K); nparts0 = 1: grec0 (nblk,
 * 32) summed records] -> records to (dL/dQQ, dL/dM_raw, dL/db, dL/dc) as gfdn_tf_coefs_bwd -> adjoint of
 * Q = expm(skew(M)), QQ = Q Q as gfdn_ortho_bwd_add (gQ: a gradient that reaches Q directly, or NULL; Q: the forward's
 * Q or NULL): gM (nblk, nper, nper), gb, gc (nblk*nper).  Bit-identical to the separate launches.  */
int gfdn_tf_param_grads(const float* A0, const float* inv_gamma0, const float* grec0, int nparts0, const float* A1,
                        const float* inv_gamma1, const float* grec1, const float* b, const float* c, int nblk,
                        int nper, const float* M, const float* gQ, const float* Q, float* gb, float* gc, float* gM,
                        void* stream);
/* gfdn_tf_param_grads + Adam on the blocks' own parameters + gfdn_tf_ortho_coefs of the NEXT step in one launch (single-
 * process training; reference src/diff_gfdn/trainer.py:452-477 backward -> optimizer.step, then :373-379 the next batch's
 * normalize / forward): after the gradients (written to gb, gc, gM as above) workgroup blk updates elements
 * offM + blk nper^2 + e, offb + blk nper + i, offc + blk nper + j of the flat Adam buffers (maths of gfdn_adam_step at
 * t = step_count + 1; the last workgroup writes t back and re-arms block_counter) and leaves Q, QQ and both record sets of
 * the UPDATED block in Q_next / QQ_next / coef_next / coef_sub_next (which may be the arrays the step read: a block's
 * workgroup is their only reader at that point).  b, c, M are the flat buffer's views of those leaves.  A1 = the raw
 * blocks (no 1 / gamma), grec1 their summed records.                                                                    */
int gfdn_tf_tail(const float* A0, const float* inv_gamma0, const float* grec0, int nparts0, const float* A1,
                 const float* grec1, const float* b, const float* c, int nblk, int nper, const float* M, const float* gQ,
                 const float* Q, float* gb, float* gc, float* gM, float* flat_p, float* flat_m, float* flat_v,
                 const unsigned char* seg, const float* lr_seg, float* step_count, unsigned int* block_counter, int offM,
                 int offb, int offc, float beta1, float beta2, float eps, float* Q_next, float* QQ_next, float* coef_next,
                 float* coef_sub_next, void* stream);
/* out[r] = sum_p part[r * cols + p]: one wavefront per row, fixed order */
int gfdn_tf_rows_sum(const float* part, int cols, int rows, float* out, void* stream);
int gfdn_tf_parts(int K, int nblk);
size_t gfdn_tf_work_bytes(int nblk);
size_t gfdn_tf_gpart_bytes(int nblk);
int gfdn_tf_eval(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                 const float* delays, const float* scale, float* T_c64, void* stream);
int gfdn_tf_energy(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                   const float* delays, float* b, float* c, float* energy, float* scale, void* work,
                   int phase, double dturn, void* stream);
/* ... whose finish also stores gains_scaled[band Bper + r][g] = gains[band Bper + r][g] scale[band G + g] (nblk = bands x G):
 * the normalisation scale folded into the receiver gains (trainer.py:317-332: T' = scale T, and H = sum_g gain[b][g] T'_g +
 * direct is linear in both), so that the launches of the linear step run on group signals of the UNSCALED functions with
 * gains_scaled as their receiver gains and nothing on the transform's stream waits for the scale; the gain network's backward
 * then takes dL/d(gains_scaled) (gfdn_mlp_gains_banded_bwd_parts_scaled), the records pass divides dL/dT by the scale
 * (gfdn_tf_compose_bwd gain_fold).  phase 3 (both) or 2 (the finish alone, behind gfdn_tf_energy(phase = 1)).             */
int gfdn_tf_energy_gains(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                         const float* delays, float* b, float* c, float* energy, float* scale, void* work, int phase,
                         double dturn, const float* gains, float* gains_scaled, int Bper, int G, void* stream);
int gfdn_tf_colorless(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                      const float* delays, const float* scale, int asym, float gscale, float* grec,
                      float* loss, void* work, double dturn, void* stream);
int gfdn_tf_compose_fwd(const double* turns, const double* logr, int K, int nbands, int G, int nper,
                        const float* coef, const float* delays, const float* scale, const float* rgain, int B,
                        const float* direct_c64, int ldd, const long long* direct_rows, const float* filt_c64,
                        int ldf, float* H_c64, int ldh, float* Tsave_c64, float* Tquad_c64, void* stream);
int gfdn_tf_compose_parts(int K);   /* partial rows per record entry that gfdn_tf_compose_bwd(grec = NULL) leaves in work */
size_t gfdn_tf_compose_bwd_work_bytes(int K, int nbands, int G);
int gfdn_tf_compose_bwd(const double* turns, const double* logr, int K, int nbands, int G, int nper,
                        const float* coef, const float* delays, const float* Tsave_c64, const float* tscale,
                        const float* rgain, int B, const float* filt_c64, int ldf, const float* gH_c64, int ldh,
                        float* grec, void* work, int gain_fold, void* stream);
/* gain_fold (with tscale): the normalisation scale sits in the receiver gains (gfdn_tf_energy_gains), the group
 * signals are those of the UNSCALED functions and gH is dL/d(T filt): dL/dT' = dL/dT / tscale.                            */
/* tscale (nbands * G floats or NULL): Tsave holds the UNSCALED group transfer functions (gfdn_tf_compose_fwd with scale =
 * NULL) and T' = tscale T is formed where the records pass reads them -- for a step whose normalisation scale joins the
 * group signals behind the transform (gfdn_irfft_odd_pairs_fwd_scaled).                                                  */
size_t gfdn_tf_gain_grad_work_bytes(int K, int nbands, int G, int B);
int gfdn_tf_gain_grad(int K, int nbands, int G, int B, const float* Tsave_c64, const float* filt_c64, int ldf,
                      const float* gH_c64, int ldh, float* grgain, void* work, void* stream);
/* Columns of the partial rows gfdn_tf_gain_grad leaves in ``work`` when ``grgain`` is NULL ((nbands B G, chunks) floats, row
 * (band B + b) G + g): the sums over the chunks then belong to the consumer (gfdn_mlp_gains_banded_bwd_parts).          */
int gfdn_tf_gain_chunks(int K);

/* ---- block transfer functions for blocks of 5..8 lines (BASELINE.json configs[4]: N = 32 = 4 groups x 8 lines) -----
 * csrc/blocktf8.hip.  The same reference maths as the gfdn_tf_* family (feedback_loop.py:326-391 resolvent of one block,
 * model.py:583-619 output stage, model.py:209-252 sub-FDN responses, trainer.py:317-332 normalize,
 * colorless_fdn/losses.py:20-73) with the polynomial evaluation AND the gradient-record accumulation on the matrix
 * cores (v_mfma_f32_16x16x4_f32): T = P / Q, 256 real coefficients each, subset index S = S1 + 16 S2 (S1: lines 0..3,
 * S2: 4..7).  coef (nblk, 9, 256) float32: the determinant polynomial Q and the numerators Y_i of y_i = (X^-1 b)_i
 * (P = sum_i c_i Y_i is formed from the CURRENT output gains when a pass starts).  Unit-circle grids only (turns).
 * Scaling convention: coefficients are built from the gains BEFORE normalize's rescale; the passes take the CURRENT c
 * and scale (NULL before the rescale) and work on the records of the current gains.  dturn != 0 (energy, colorless):
 * the grid is uniform on the unit circle with that step in turns -- the phasors are then stepped by constant rotations
 * between exact evaluations, as the gfdn_tf_* passes do.
 *   gfdn_tf8_coefs      : records of (A0, 1/gamma0) -> coef0 and optionally (A1, 1/gamma1) -> coef1, sharing b.
 *   gfdn_tf8_energy     : E_blk = mean_k |T|^2 -> energy, scale = E^(-1/2), b, c /= E^(1/4) in place; work: nblk *
 *                         gfdn_tf8_parts(K) floats.
 *   gfdn_tf8_tsave      : T' (nbands * G, K) complex64 scaled group transfer functions (+ Tquad (nbands, K, 4), the
 *                         layout gfdn_irfft_odd_pairs_compose_fwd consumes), G <= 4.
 *   gfdn_tf8_colorless  : loss[blk] = mean_k (|T'| - 1)^p and the gradient records of gscale * sum loss.
 *   gfdn_tf8_compose_bwd: gradient records of the output stage, dL/dT'_g = conj(filt) sum_b rgain[b][g] dL/dH[b].
 *     part[(blk * 512 + e) * gfdn_tf8_parts(K) + p]: e < 256 dL/dP_S, 256 + S dL/dQ_S (gfdn_tf8_part_bytes).
 *   gfdn_tf8_param_grads: records -> dL/dA (cofactors of the 256 masked 8 x 8 / bordered 9 x 9 matrices per block and
 *     set, float64) for set 0 (damped loop, A0 = Q Q, 1/gamma0) and set 1 (sub-FDN, A1 = raw M), dL/db, dL/dc w.r.t. the
 *     CURRENT gains b, c, then dL/dM through the adjoint of Q = expm(skew(M)), Q Q (as gfdn_ortho_bwd with dL/dM_raw
 *     added; gQ: a gradient that reaches Q directly, e.g. the sparsity term's).  work: gfdn_tf8_param_grads_work_bytes. */
int gfdn_tf8_coefs(const float* A0, const float* inv_gamma0, float* coef0, const float* A1, const float* inv_gamma1,
                   float* coef1, const float* b, const float* c, int nblk, int nper, void* stream);
int gfdn_tf8_parts(int K);
size_t gfdn_tf8_part_bytes(int nblk, int K);
int gfdn_tf8_energy(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays, float* b,
                    float* c, float* energy, float* scale, void* work, double dturn, void* stream);
int gfdn_tf8_tsave(const double* turns, int K, int nbands, int G, int nper, const float* coef, const float* delays,
                   const float* c, const float* scale, float* Tsave_c64, float* Tquad_c64, const float* filt_c64, int ldf,
                   float* Hout_c64, float* Dinv_c64, const int* hslot, void* stream);
/* Hout (nbands G, K; NULL: none) = T' filt with band row blk / G of filt (nbands, ldf; NULL: Hout = T'): the group responses
 * through the band's filter, what the time-domain output stage transforms (no tensor operation between the two).
 * hslot (K; NULL: Hout in grid order): column of grid point k in Hout and filt, bit 31 set where that column holds the
 * conjugate -- the grid in BIN order (Tsave, Dinv: what gfdn_tfp_compose_bwd reads), the group responses in the slot
 * order of the odd-length transform (gfdn_irfft_odd_slot_order).
 * Dinv (nbands G, K; NULL: none) = 1 / Q per bin: with Tsave what gfdn_tf8_compose_bwd on the SAME grid takes back
 * (Tsave_c64, Dinv_c64: both or neither) instead of evaluating the two polynomials again.                               */
int gfdn_tf8_colorless(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays,
                       const float* c, const float* scale, int asym, float gscale, float* part, float* lossp,
                       float* loss, double dturn, void* stream);
int gfdn_tf8_compose_bwd(const double* turns, int K, int nbands, int G, int nper, const float* coef, const float* delays,
                         const float* c, const float* scale, const float* rgain, int B, const float* filt_c64, int ldf,
                         const float* gH_c64, int ldh, const float* Tsave_c64, const float* Dinv_c64, float* part,
                         void* stream);
size_t gfdn_tf8_param_grads_work_bytes(int nblk);
int gfdn_tf8_param_grads(const float* A0, const float* inv_gamma0, const float* part0, int nparts0, const float* A1,
                         const float* inv_gamma1, const float* part1, int nparts1, const float* b, const float* c,
                         int nblk, int nper, const float* M, const float* gQ, const float* Q, float* gb, float* gc,
                         float* gM, void* work, void* stream);
/* gfdn_tf8_param_grads with the optimiser update of the blocks' own M, b, c (the flat buffers of gfdn_adam_step at the element
 * offsets offM / offb / offc; torch.optim.Adam of trainer.py:475 with the same roundings) and the NEXT step's Q = expm(skew(M)),
 * Q Q and a snapshot c_next of the updated output gains in the same launch: the step's tail and most of its next head
 * (gfdn_tf_tail for blocks of 5..8 lines; gfdn_tf8_coefs stays a launch of its own).  Both record sets are required.   */
int gfdn_tf8_tail(const float* A0, const float* inv_gamma0, const float* part0, int nparts0, const float* A1, const float* part1,
                  int nparts1, const float* b, const float* c, int nblk, int nper, const float* M, const float* gQ,
                  const float* Q, float* gb, float* gc, float* gM, void* work, float* flat_p, float* flat_m, float* flat_v,
                  const unsigned char* seg, const float* lr_seg, float* step_count, unsigned int* block_counter, int offM,
                  int offb, int offc, float beta1, float beta2, float eps, float* Q_next, float* QQ_next, float* c_next,
                  void* stream);
/* ---- polynomial passes of the same blocks on the reference's own grid by fast transforms (csrc/polyfft.hip).
 * Preconditions: INTEGER delay lengths (config.py:131-140) and the grid z_k = e^{2 pi i k / nfft}, k = 0 .. nfft / 2
 * (dataloader.py:552-566), nfft = 2^p: then Q(z_k) = conj(rfft(q, nfft)[k]) for the real sequence q[m] = sum of the
 * coefficients of degree m mod nfft (<= 256 non-zero samples), and the gradient records are 256 samples of two inverse real
 * transforms.  Records / scaling convention of gfdn_tf8_* (sequences from the gains BEFORE normalize's rescale; T' = scale T).
 * Used for normalize, the colorless pass and BOTH adjoints; the FORWARD group responses of the damped loop stay on
 * gfdn_tf8_tsave: a float32 transform carries an absolute error of ~1e-6 |q|, which next to the loop's poles (small |Q|)
 * is 1e-5 of T -- 100 x the rounding of the direct evaluation, and it reaches dL/dM through the dB stages of the decay
 * losses (measured: 1.15e-3 against 7.6e-4 of its largest entry; the adjoints are linear in 1 / Q and do not care).
 *   gfdn_tfp_forward    : X (2 nblk, ldx >= nfft / 2 + 1) complex64 = rfft of the sequences [Q | P] of ONE record set (nblk
 *                         rows each; nsub = 256: records (nblk, 9, 256) of gfdn_tf8_coefs, nsub = 512: (nblk, 10, 512) of gfdn_tf9_coefs).  T: samples per sequence kept in seq (2 nblk, T): max degree + 1 <= T <= nfft.
 *                         work: gfdn_irfft_pow2_work_bytes(nfft, 2 nblk).
 *   gfdn_tfp_energy     : normalize on rows (Xq, Xp) of the raw sub-FDN blocks, K = nfft / 2 + 1 bins: energy (or NULL),
 *                         scale = E^(-1/2), b, c /= E^(1/4) in place.  work: nblk * gfdn_tfp_parts() floats.  gains /
 *                         gains_scaled ((nblk / G) Bper, G; or NULL): as gfdn_tf_energy_gains.
 *   gfdn_tfp_colorless  : gfdn_tf8_colorless on the transformed sequences: loss[blk], gradient records part (nblk, 512)
 *                         (ONE partial row: nparts = 1 for gfdn_tf8_param_grads).  UV (2 nblk, ldx) complex64, x (2 nblk, ldt >=
 *                         nfft) float, work: gfdn_irfft_pow2_work_bytes(nfft, 2 nblk), lossp: nblk * gfdn_tfp_parts() floats.
 *   gfdn_tfp_compose_bwd: gH (nblk, ldh) = dL/d(T'_g filt) on the slot order (slot_of_bin: gfdn_tf8_tsave's hslot) -> gradient
 *                         records part (nblk, 512) of the damped blocks (the linear step's adjoint: one gradient row per
 *                         group).  Tnat, Dnat (nblk, Ku): gfdn_tf8_tsave's Tsave / Dinv on bins 0 .. Ku - 1 in bin order;
 *                         tscale (nblk; or NULL): Tnat holds the UNSCALED functions, T' = tscale Tnat; gain_fold: as
 *                         gfdn_tf_compose_bwd's.                                                                          */
/* ---- blocks of up to NINE lines (the directional model, BASELINE.json configs[3]: N = 27 = 3 x 9; csrc/blocktf9.hip): the
 * coefficient records and the records -> parameters map of the gfdn_tf8_* family with 512 subsets, for ONE record set.
 *   gfdn_tf9_coefs    : coef (nblk, 10, 512) float32 of (A, 1/gamma or NULL): [0] the determinant polynomial, [1 + i] the
 *                       numerator of y_i = (X^-1 b)_i (float64 determinants of the masked 9 x 9 matrices).
 *   gfdn_tf9_rec_grads: gradient records grec (nblk, 1024) = dL/dP_S | dL/dQ_S -> dL/dA (nblk, n, n), dL/db, dL/dc (nblk n)
 *                       w.r.t. the gains b, c given (cofactors of the masked n x n and bordered (n + 1) x (n + 1) matrices,
 *                       float64).  work: gfdn_tf9_rec_grads_work_bytes(nblk).
 * The evaluation on the grid and its adjoint: gfdn_tfp_forward + gfdn_tfp_ratio_fwd / _bwd below (nper <= 9).              */
int gfdn_tf9_coefs(const float* A, const float* inv_gamma, const float* b, int nblk, int nper, float* coef, void* stream);
size_t gfdn_tf9_rec_grads_work_bytes(int nblk);
int gfdn_tf9_rec_grads(const float* A, const float* inv_gamma, const float* grec, const float* b, const float* c, int nblk,
                       int nper, float* gA, float* gb, float* gc, void* work, void* stream);
/*   gfdn_tfp_ratio_fwd  : T (nblk, K) = P / Q and Dinv (nblk, K) = 1 / Q on the grid from the transformed sequences (rows Xq, Xp
 *                         of gfdn_tfp_forward) -- the un-damped group responses S_g(z) of the colorless branch, model.py:209-252.
 *   gfdn_tfp_ratio_bwd  : gT (nblk, ldg) = dL/dT with dL = sum_k Re(conj(gT_k) dT_k) -> gradient records part (nblk, 2 nsub)
 *                         (nsub as gfdn_tfp_forward's; 512: what gfdn_tf9_rec_grads takes).  UV (2 nblk, ldx), x (2 nblk,
 *                         ldt >= nfft), work: gfdn_irfft_pow2_work_bytes(nfft, 2 nblk).                                     */
int gfdn_tfp_ratio_fwd(const float* Xq_c64, const float* Xp_c64, int ldx, int K, int nblk, float* T_c64, float* Dinv_c64,
                       void* stream);
int gfdn_tfp_ratio_bwd(int nfft, int nblk, int nper, int nsub, const float* delays, const float* gT_c64, int ldg, const float* T_c64,
                       const float* Dinv_c64, int T, float* UV_c64, int ldx, float* x, int ldt, void* work, float* part,
                       void* stream);
/* T (gfdn_tfp_colorless, _compose_bwd, _ratio_bwd): the samples per sequence of gfdn_tfp_forward -- the gathers read the inverse
 * transforms' first T samples only, and at nfft = 131 072 only those are stored (gfdn_irfft_pow2_fwd_band).                 */
int gfdn_tfp_parts(void);
int gfdn_tfp_forward(int nfft, int nblk, int nper, int nsub, const float* coef, const float* delays, const float* c, int T,
                     float* seq, float* X_c64, int ldx, void* work, void* stream);
int gfdn_tfp_energy(const float* Xq_c64, const float* Xp_c64, int ldx, int K, int nblk, int nper, float* b, float* c,
                    float* energy, float* scale, void* work, const float* gains, float* gains_scaled, int Bper, int G,
                    void* stream);
int gfdn_tfp_colorless(const float* Xq_c64, const float* Xp_c64, int ldx, int nfft, int nblk, int nper, const float* delays,
                       const float* scale, int asym, float gscale, int T, float* UV_c64, float* x, int ldt, void* work,
                       float* part, float* lossp, float* loss, void* stream);
int gfdn_tfp_compose_bwd(int nfft, int nbands, int G, int nper, const float* delays, int Ku, const int* slot_of_bin,
                         const float* gH_c64, int ldh, const float* filt_c64, int ldf, const float* Tnat_c64,
                         const float* Dnat_c64, const float* tscale, int gain_fold, int T, float* UV_c64, int ldx, float* x,
                         int ldt, void* work, float* part, void* stream);

/* ---- measurement kernel for BASELINE.json configs[4] ("fp32 vs bf16 feedback-matmul on MFMA") ------------------
 * The reference's dense formulation (feedback_loop.py:389-391 explicit resolvent P (K, N, N); model.py:615-619
 * einsum contraction with the output gains) for N = B = 32: H[b][k] = sum_m (sum_n C[b][n] P_k[n][m]) bvec[m], one
 * wavefront per bin on the matrix cores -- use_bf16 != 0: v_mfma_f32_32x32x16_bf16 on operands rounded to bfloat16;
 * 0: v_mfma_f32_32x32x2_f32.  P (K, 32, 32) complex64, C (32, 32), bvec (32) float32 -> H (32, K) complex64.  Not on
 * the product path (which never forms P); tools/mfma_experiment.py times it against that path.                     */
int gfdn_exp_contract_mfma(const float* P_c64, int K, const float* C, const float* bvec, int use_bf16,
                           float* H_c64, void* stream);

/* ---- odd-length inverse real FFT  (losses.py:207-213, :442-445: irfft(X, n = K)) ---------
 * x[t] = irfft(X[0..(n-1)/2], n), n odd (65 537 = 2^16+1 at nfft = 131 072), by Bluestein's
 * algorithm on power-of-two FFTs of length L >= n + (n-1)/2.
 * table: gfdn_bluestein_table_bytes(n) bytes, filled once by gfdn_bluestein_table_init
 * (host computes chirp / chirp spectrum / twiddles in float64; synchronous).
 * work: gfdn_bluestein_work_bytes(n, batch).  X: (batch, ldx) complex64, only the first
 * (n+1)/2 bins are read.  x: (batch, ldo) float.                                          */
size_t gfdn_bluestein_table_bytes(int n);
int gfdn_bluestein_table_init(int n, void* table);
size_t gfdn_bluestein_work_bytes(int n, int batch);
int gfdn_irfft_odd_fwd(const void* table, int n, const float* X_c64, int ldx, int batch,
                       float* x, int ldo, void* work, void* stream);
/* adjoint: gx (batch, ldo) -> gX (batch, ldx) complex64; bins above (n-1)/2 are set to 0.
 * gx2: optional second gradient of the same shape, summed in on load (gx + gx2, in that
 * order) -- the EDC and EDR gradients of one response meet here without an add kernel.   */
int gfdn_irfft_odd_bwd(const void* table, int n, const float* gx, const float* gx2, int ldo,
                       int batch, float* gX_c64, int ldx, void* work, void* stream);

/* Slot order (n = 65 537, Rader): the transform's natural input order is u[s] = Y[3^s mod n] (Y the Hermitian
 * extension of X).  gfdn_irfft_odd_slot_order fills HOST arrays bins[s], conj[s], s < (n-1)/2, such that
 * u[s] = conj[s] ? conj(X[bins[s]]) : X[bins[s]] (and u[s + (n-1)/2] = conj(u[s])).  A model that is evaluated
 * pointwise on the frequency grid (the solve and the output stage are) can be evaluated directly on the grid
 * points { conj[s] ? conj(z_bins[s]) : z_bins[s] } and hand Xs[0] = X[0], Xs[1 + s] = u[s] to the *_slots entry
 * points: no gather in the forward transform, no scatter in the adjoint (gXs comes back in the same order).
 * GFDN_E_UNSUPPORTED for other lengths.                                                              */
int gfdn_irfft_odd_slot_order(int n, int* bins_host, int* conj_host);
int gfdn_irfft_odd_slots_fwd(const void* table, int n, const float* Xs_c64, int ldx, int batch,
                             float* x, int ldo, void* work, void* stream);
int gfdn_irfft_odd_slots_bwd(const void* table, int n, const float* gx, const float* gx2, int ldo,
                             int batch, float* gXs_c64, int ldx, void* work, void* stream);

/* Pairs: two items per complex transform.  Spectra in slot order as above (batch rows); the TIME signals are
 * pair-interleaved: x2 (ceil(batch / 2), ldo) float2 with item 2p in .x and item 2p + 1 in .y (.y = 0 / ignored
 * for the missing partner of an odd batch) -- one 8-byte scatter / gather per slot serves two items, and the
 * three passes move half the work blocks.  gfdn_stft_power_pairs(_bwd) and gfdn_edc_loss_pairs consume and
 * produce that layout (win = 4096 only); P, T_db, loss_item stay per item; gfdn_edc_loss_pairs takes
 * gfdn_edc_work_bytes(items + 1) bytes of work.  gfdn_stft_power_pairs_bwd STORES gx2 = base2 + d<gP, P>/dx2
 * (base2: another gradient of the same layout or NULL; may alias gx2): even frames first, odd frames in a second
 * launch -- frames of one parity do not overlap, so there are no atomics and gx2 needs no clearing.
 * gfdn_stft_power_pairs_bwd_phase runs ONE of the two launches with the base added by the SECOND: phase 0 = even
 * frames, gx2 = their contribution alone (base2 ignored: the launch does not wait for whoever produces it);
 * phase 1 = odd frames, gx2 += contribution + base2 everywhere (base2 must not alias gx2).  0 then 1 = the above. */
int gfdn_irfft_odd_pairs_fwd(const void* table, int n, const float* Xs_c64, int ldx, int batch,
                             float* x2, int ldo, void* work, void* stream);
/* gfdn_irfft_odd_pairs_fwd with the output stage of the block-transfer-function step (gfdn_tf_compose_fwd) folded into
 * its first pass: x2 = irfft_n(H), H[b][k] = (sum_g rgain[b][g] T[band * G + g][k] + direct[rows[b]][k]) * filt[band][k]
 * (model.py:583-619, trainer.py:459, losses.py:207-213), formed while the pass loads it; H is never stored (same operations as
 * gfdn_tf_compose_fwd in the same order: x2 agrees to float32 rounding).  T (nbands, ldt, 4) = gfdn_tf_compose_fwd's
 * Tquad output (the band's group transfer functions of a column side by side, zeros beyond G; its H_c64 may be NULL);
 * items band-major, batch = nbands * Bper with Bper even; slot-ordered columns, column 0 = bin 0; h0: batch floats of
 * scratch.  stages = 7: the whole transform (bits 0 / 1 / 2 = its three passes, as gfdn_irfft_odd_stages).
 * n = 65537 (the 128 x 512 Rader geometry) only: GFDN_E_UNSUPPORTED otherwise.  */
int gfdn_irfft_odd_pairs_compose_fwd(const void* table, int n, const float* direct_c64, int ldd,
                                     const long long* direct_rows, const float* T_c64, int ldt, const float* rgain,
                                     const float* filt_c64, int ldf, int nbands, int G, int batch, float* h0,
                                     float* x2, int ldo, void* work, int stages, void* stream);
int gfdn_irfft_odd_pairs_bwd(const void* table, int n, const float* gx2, const float* gx2b, int ldo,
                             int batch, float* gXs_c64, int ldx, void* work, void* stream);
/* gfdn_irfft_odd_pairs_bwd with the GAINS pass of the output stage's adjoint (gfdn_tf_gain_grad) folded into its last
 * pass: every value of dL/dH is used while it is in a register.  gpart[(b * G + g) * parts + p], parts =
 * gfdn_irfft_odd_pairs_gains_parts(n): partial sums of dL/drgain[b][g] = sum_k Re(dL/dH[b][k] conj(filt[band][k]
 * T'[band][k][g])) -- sum each row with gfdn_tf_rows_sum.  T (nbands, ldt, 4): gfdn_tf_compose_fwd's Tquad (scaled).
 * stages as gfdn_irfft_odd_stages; n = 65537 only.  */
int gfdn_irfft_odd_pairs_gains_parts(int n);
int gfdn_irfft_odd_pairs_gains_bwd(const void* table, int n, const float* gx2, const float* gx2b, int ldo, int batch,
                                   float* gXs_c64, int ldx, const float* T_c64, int ldt, const float* filt_c64, int ldf,
                                   int nbands, int G, float* gpart, void* work, int stages, void* stream);
int gfdn_stft_power_pairs(const float* x2, int ld, int T, int items, int win, float* P, float* zero_buf2,
                          void* stream);
int gfdn_stft_power_pairs_bwd(const float* x2, int ld, int T, int items, int win, const float* gP,
                              const float* base2, float* gx2, void* stream);
int gfdn_stft_power_pairs_bwd_phase(const float* x2, int ld, int T, int items, int win, const float* gP,
                                    const float* base2, float* gx2, int phase, void* stream);
int gfdn_edc_loss_pairs(const float* x2, int ld, int items, int start, int len, const float* T_db,
                        const long long* target_rows, const float* maskw, float inv_count, float gscale,
                        float* loss_item, float* gx2, void* work, void* stream);

/* ---- The output stage in the TIME domain (csrc/linear.hip).  The reference forms H[b] = (sum_g gain[b][g] T_g + d[b]) filt
 * per receiver (src/diff_gfdn/model.py:583-619, trainer.py:459) and transforms every H[b] inside the decay losses
 * (losses.py:207-213, :442-445: irfft(H, n = K)).  The transform is linear and T_g, filt do not depend on the receiver:
 *     x[b] = irfft(d[b] filt) + sum_g gain[b][g] irfft(T_g filt) = xd[row_b] + sum_g gain[b][g] tau_g,
 * xd a constant of the dataset (transformed once per receiver, like the decay targets) and tau the G transformed group
 * responses of the band -- a step runs G forward and G adjoint transforms per band instead of one per receiver.
 *   gfdn_lin_combine_fwd : x (items = nbands B) from xd (rows, ld_xd >= n) [row indirection: item b reads row rows[b]],
 *                          tau (nbands G signals) and rgain (items, G);
 *   gfdn_lin_gamma       : gamma[band G + g] = sum_{b in band} rgain[b][g] (gx[b] [+ gxb[b]]) -- the G signals per band
 *                          whose adjoint transform is dL/d(T_g filt);
 *   gfdn_lin_gain_dots   : part[((band B + b) G + g) chunks + chunk] = partial sums over the samples of
 *                          dL/dgain[b][g] = <gx[b] [+ gxb[b]], tau[band G + g]>, chunks = gfdn_lin_gain_chunks(n)
 *                          (the rows gfdn_mlp_gains_banded_bwd_parts sums itself).
 * Layout flags (*_pairs): 1 = two signals interleaved sample by sample, (ceil(S / 2), ld, 2) float, as the pair
 * transforms and the pair STFT / EDC kernels use (items 2p, 2p + 1 in one pair; with in_pairs B must be even and
 * gxb NULL); 0 = (S, ld) float.  G <= 4; B G <= 256 for gfdn_lin_gamma.  All sums in fixed order.                       */
int gfdn_lin_gain_chunks(int n);
int gfdn_lin_combine_fwd(const float* xd, int ld_xd, const long long* rows, const float* tau, int ld_tau, int tau_pairs,
                         const float* rgain, int nbands, int B, int G, int n, float* x, int ld_x, int out_pairs,
                         void* stream);
int gfdn_lin_gamma(const float* gx, const float* gxb, int ld_g, int in_pairs, const float* rgain, int nbands, int B, int G,
                   int n, float* gamma, int ld_o, int out_pairs, const int* slot_of_time, const float* base2, int ld_b,
                   void* stream);
/* base2 (out_pairs only; NULL: none): signals in gamma's pair-interleaved layout, TIME order, pitch ld_b, added to the
 * sums -- the part of the same gradient signals that is already summed per group (gfdn_stft_pairs_spectrum_bwd).         */
/* slot_of_time (n ints, device; out_pairs only; NULL: time order): gamma is written in the adjoint pair transform's own
 * order -- sample 0 first, the sample of time t >= 1 at 1 + slot_of_time[t], slot_of_time the inverse of
 * gfdn_irfft_odd_time_slots -- for gfdn_irfft_odd_pairs_bwd_tslots, whose first pass then loads coalesced rows.           */
int gfdn_irfft_odd_time_slots(int n, int* times);
int gfdn_irfft_odd_pairs_bwd_tslots(const void* table, int n, const float* gx2s, int ldo, int batch, float* gXs_c64,
                                    int ldx, void* work, void* stream);
/* ... of the SUM of up to three slot-ordered inputs (b, c optional; c only with b), added where the first pass loads them:
 * parts of one gradient signal left by different launches need no merge pass in front of the transform                   */
int gfdn_irfft_odd_pairs_bwd_tslots3(const void* table, int n, const float* gx2s_a, const float* gx2s_b,
                                     const float* gx2s_c, int ldo, int batch, float* gXs_c64, int ldx, void* work,
                                     void* stream);
/* gfdn_irfft_odd_pairs_fwd with per-item factors on the time signals (oscale: batch floats, device; x2 = oscale[item] *
 * irfft(Xs[item])), applied where the last pass stores.  stages: 7 = the whole transform; 3 = its first two passes, 4 = the
 * last one, the only reader of oscale (a caller whose factors come from another stream waits between the two calls)       */
int gfdn_irfft_odd_pairs_fwd_scaled(const void* table, int n, const float* Xs_c64, int ldx, int batch, const float* oscale,
                                    float* x2, int ldo, void* work, int stages, void* stream);
int gfdn_lin_gain_dots(const float* gx, const float* gxb, int ld_g, int in_pairs, const float* tau, int ld_tau,
                       int tau_pairs, int nbands, int B, int G, int n, float* part, int ld_part, void* stream);
/* (part rows have pitch ld_part >= gfdn_lin_gain_chunks(n): further columns may hold other partial sums of the same
 * gradients, e.g. gfdn_edr_lin_loss's)                                                                                 */

/* gfdn_lin_gamma (pair-interleaved in and out, base2, slot_of_time) and gfdn_lin_gain_dots as ONE sweep over gradient
 * signals that are nonzero on the window [win_start, win_start + win_len) only (band_win_len: optional per-band lengths,
 * device int32): samples outside the window are not read.  part rows of pitch ld_part hold gfdn_lin_gamma_dots_tiles(n)
 * partial sums per (item, group).                                                                                        */
int gfdn_lin_gamma_dots_tiles(int n);
int gfdn_lin_gamma_dots(const float* gx2, int ld_g, const float* rgain, int nbands, int B, int G, int n, const float* tau2,
                        int ld_tau, const float* base2, int ld_b, const int* slot_of_time, float* gamma, int ld_o,
                        float* part, int ld_part, int win_start, int win_len, const int* band_win_len, const float* base2b,
                        void* stream);

/* ---- The EDR loss on linearly composed short-time spectra (csrc/edrlin.hip).  With the output stage in the time domain
 * the STFT of a receiver's signal is Sd[row_b] + sum_g gain[b][g] Stau_g (the STFT is linear): Sd = the STFT of the
 * transformed direct paths, a constant of the dataset; Stau = the STFT of the band's G group signals.  The per-receiver
 * STFTs of src/diff_gfdn/losses.py:501-553 and their adjoints do not run; losses.py:556-575, :478-492 (EDR, dB, L1 ratio)
 * are evaluated per (receiver, frequency) column on the composed spectra:
 *   gfdn_stft_pairs_spectrum      : S (items, nframes, 2049) complex from pair-interleaved signals (win = 4096);
 *   gfdn_edr_lin_loss             : loss partials part (items, gfdn_edr_lin_parts(nfreq)) (as gfdn_edr_loss without
 *                                   loss_item), gP (items, nframes, nfreq) = gscale / sum_abs dloss/d|S|^2, and the EDR part
 *                                   of dL/dgain[b][g] as partial rows dots[(b G + g) ld_dots + col0 + j];
 *   gfdn_edr_lin_gsum             : Gsum (nbands G, nframes, nfreq) complex = sum_{b in band} gain[b][g] 2 gP[b] S[b];
 *   gfdn_stft_pairs_spectrum_bwd  : gx2 = [base2 +] adjoint STFT of gradient spectra G (items, nframes, 2049) -- with
 *                                   G = Gsum: the EDR part of dL/dtau.
 * rows: item -> row of Sd / T_db / sum_abs (NULL: identity).  G <= 4, nframes <= 32.                                    */
int gfdn_stft_pairs_spectrum(const float* x2, int ld, int T, int items, int win, float* S_c64, int tiled, void* stream);
int gfdn_stft_pairs_spectrum_bwd(const float* G_c64, int T, int items, int win, const float* base2, float* gx2, int ld,
                                 int tiled, int nsplit, float* gx2b, void* stream);
/* gx2b (NULL: two launches, the odd frames added into gx2 behind the even ones): ONE launch, the odd frames' contributions
 * STORED to gx2b (layout of gx2; zeros where no odd frame reaches) -- the adjoint is gx2 + gx2b, which gfdn_lin_gamma_dots
 * adds where it reads its base (base2b).                                                                                 */
/* nsplit >= 1: G holds nsplit partial sets (nsplit, items, nframes, 2049) that are added, in order, where they are loaded
 * (the partial planes of gfdn_edr_lin_loss_gsum).                                                                       */
/* tiled = 1: the (nframes, nfreq) planes of S / G / Sd / Stau / T_db / gP / Gsum are stored with the frequencies cut into
 * blocks of 256 and a block's frames contiguous -- cell(m, f) = (f / 256) nframes 256 + m w + f % 256, w = the block's
 * width (the last block holds the rest) -- so that a (receiver, frequency block) workgroup of gfdn_edr_lin_loss streams
 * contiguous runs; 0: cell = m nfreq + f.  gfdn_edr_lin_gsum works cell by cell and takes either.                        */
int gfdn_edr_lin_parts(int nfreq);            /* partial-sum columns of gfdn_edr_lin_loss */
int gfdn_edr_lin_band_parts(int nfreq);       /* ... of gfdn_edr_lin_loss_gsum (one per wave of 8 frequencies) */
int gfdn_edr_lin_loss(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain, int nbands,
                      int B, int G, const float* T_db, const float* sum_abs, int nframes, int nfreq, float gscale,
                      int want_grad, float* gP, float* part, float* dots, int ld_dots, int col0, int tiled, void* stream);
int gfdn_edr_lin_gsum(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain, int nbands,
                      int B, int G, const float* gP, int nframes, int nfreq, float* Gsum_c64, void* stream);
/* gfdn_edr_lin_loss(want_grad = 1) + gfdn_edr_lin_gsum in ONE launch (k_edr_lin_wave): a thread owns cells of the band's
 * (frame, frequency) plane and walks the band's receivers (fixed order); a wave = 8 frequencies x all frames, the scans
 * along the frames inside the wave on the VALU (DPP / permlane swaps), no LDS, no barriers; dL/d|S|^2 is never written and
 * Sd is read once.  part (items, ld_part >= gfdn_edr_lin_band_parts(nfreq)); dots columns [col0, col0 +
 * gfdn_edr_lin_band_parts(nfreq)) (260 at nfreq = 2049); Gsum (nsplit, nbands G, nframes, nfreq): the band's receivers cut
 * into nsplit runs with one partial plane set each (gfdn_stft_pairs_spectrum_bwd adds them); tiled as above.            */
int gfdn_edr_lin_loss_gsum(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain, int nbands,
                           int B, int G, const float* T_db, const float* sum_abs, int nframes, int nfreq, float gscale,
                           float* part, int ld_part, float* dots, int ld_dots, int col0, float* Gsum_c64, int nsplit,
                           int tiled, void* stream);
/* gfdn_edc_loss_pairs[_banded] on signals x[b] = xd[xrows[b]] + sum_g rgain[b][g] tau[band G + g] formed by the first of its
 * three launches (segment energies), which stores them on the EDC window only into the scratch xwin2 (ceil(items / 2), ld, 2)
 * for the two scans (tau2 pair-interleaved).  ld = the signals' length = pitch of gx2 and xwin2; item_len NULL: one window
 * max_len for all items and T_db rows of pitch max_len; else as gfdn_edc_loss_pairs_banded with items_per_band = B.       */
int gfdn_edc_loss_pairs_lin(const float* xd, int ld_xd, const long long* xrows, const float* tau2, int ld_tau,
                            const float* rgain, int nbands, int B, int G, int ld, int start, int max_len,
                            const int* item_len, const float* T_db, int ld_T, const long long* target_rows,
                            const float* maskw, int ld_mask, float inv_count, float gscale, float* loss_item, float* gx2,
                            int fill_outside, float* xwin2, void* work, void* stream);
/* (fill_outside = 0: gx2 is written on the item's window only -- for a consumer that reads nothing else,
 * gfdn_lin_gamma_dots; 1: zeros outside the window, as gfdn_edc_loss_pairs)                                              */

/* The EDC term of the linear step in ONE launch (csrc/edcone.hip; src/diff_gfdn/losses.py:187-238): one workgroup per
 * item keeps the item's window x[b] = xd[xrows[b]] + sum_g rgain[b][g] tau[band G + g] in registers and runs both scans,
 * the dB stage and the gradient without staging anything through memory.  Inputs as gfdn_edc_loss_pairs_lin (T_db rows of
 * pitch ld_T, item_len NULL: max_len for all; maskw row band * ld_mask).  Outputs: loss_item (items); gx (items, ld_gx
 * >= max_len; NULL: none) = dL/dx on the item's window, sample start + j at column j (columns >= the item's length are
 * not written); dots (NULL: none): dots[(item G + g) ld_dots + col] = <dL/dx, tau_g>, the EDC part of dL/drgain.
 * max_len <= gfdn_edc_lin_one_max_len(), G <= 4.                                                                          */
int gfdn_edc_lin_one_max_len(void);
int gfdn_edc_lin_one(const float* xd, int ld_xd, const long long* xrows, const float* tau2, int ld_tau, const float* rgain,
                     int nbands, int B, int G, int start, int max_len, const int* item_len, const float* T_db, int ld_T,
                     const long long* target_rows, const float* maskw, int ld_mask, float inv_count, float gscale,
                     float* loss_item, float* gx, int ld_gx, float* dots, int ld_dots, int col, void* stream);
/* gamma2[band G + g][pos(t)] = base2a + base2b + sum_{b in band} rgain[b][g] gx[b][t - win_start] (the sum on the band's
 * window only): gfdn_lin_gamma_dots without the dot products, on the window-only rows gfdn_edc_lin_one leaves.  gamma2 and
 * the bases pair-interleaved (ceil(nbands G / 2), ld, 2); pos = the adjoint pair transform's slot order when slot_of_time
 * is given (as gfdn_lin_gamma), else t.                                                                                   */
int gfdn_lin_gamma_win(const float* gx, int ld_g, const float* rgain, int nbands, int B, int G, int n, int win_start,
                       int win_len, const int* band_win_len, const float* base2a, const float* base2b, int ld_b,
                       const int* slot_of_time, float* gamma2, int ld_o, void* stream);

/* out2[r][pos(t)] = a2[r][t] + b2[r][t] + c2[r][t] over `rows` pair-interleaved signal rows (ld / ld_o float2 per row;
 * b2, c2 optional, c2 only with b2), pos as gfdn_lin_gamma's slot_of_time (NULL: t): parts of one gradient signal left by
 * different launches, merged into the adjoint pair transform's slot order                                                  */
int gfdn_lin_merge_slots(const float* a2, const float* b2, const float* c2, int rows, int n, int ld, const int* slot_of_time,
                         float* out2, int ld_o, void* stream);

/* Float64 transforms for DATASET CONSTANTS (csrc/fft64.hip, round 6): the reference's rfft runs in float64
 * (src/diff_gfdn/dataloader.py:250, :300-325) and its irfft(H, n = K) on complex128 spectra (losses.py:207-213, :442-445);
 * the linear step's direct-path store xd = irfft(early filt, n) is built once per dataset, so accuracy is free there.
 *   gfdn_rfft_pow2_f64:      X (rows, kout) complex128 = the first kout bins of rfft(x[:, :len] zero-padded, nfft = 2^p)
 *   gfdn_irfft_odd_f64_plan: bhat (M) complex128 = the chirp spectrum of Bluestein's algorithm for the odd length n,
 *                            M = gfdn_irfft_odd_f64_length(n); work: gfdn_f64_fft_work_bytes(1, M)
 *   gfdn_irfft_odd_f64:      out (rows, ldo) = irfft(X [filt], n) on the bins 0 .. (n - 1) / 2 of the complex128 rows X
 *                            (pitch ldx; filt (>= (n + 1) / 2) complex128 multiplies every row; NULL: none), written as
 *                            float32 (out32) and / or float64 (out64); work: gfdn_f64_fft_work_bytes(rows, M).
 * Radix-2 passes through memory on double2, twiddles and chirps from exactly reduced integer phases.                        */
size_t gfdn_f64_fft_work_bytes(int rows, int M);
int gfdn_irfft_odd_f64_length(int n);
int gfdn_rfft_pow2_f64(const double* x, int ldx, int len, int rows, int nfft, double* X_c128, int kout, void* work,
                       void* stream);
int gfdn_irfft_odd_f64_plan(int n, double* bhat_c128, void* work, void* stream);
int gfdn_irfft_odd_f64(const double* X_c128, int ldx, const double* filt_c128, int rows, int n, const double* bhat_c128,
                       float* out32, double* out64, int ldo, void* work, void* stream);

/* Measurement hook: the same transform launched stage by stage (stages: bit 0 column pass +
 * chirp, bit 1 row pass with the chirp-spectrum product, bit 2 inverse column pass + epilogue) so
 * that one kernel can be bracketed by HIP events on the launch stream (bench.py's roofline leg).
 * adjoint = 0: in = X (complex), out = x (real); adjoint = 1: in = gx (real) [+ in2], out = gX;
 * slots = 1: spectrum side in slot order; slots = 2: slot order + pair-interleaved time signals (see above). */
int gfdn_irfft_odd_stages(const void* table, int n, const void* in, const float* in2, int ld_in, int batch,
                          void* out, int ld_out, void* work, int adjoint, int stages, int slots,
                          void* stream);

/* ---- row normalisation of the SH-domain receiver weights (spatial_sampling/model.py:117-190 normalise_weights:
 * weights / (norm(weights, dim=-1, keepdim=True) + 1e-6)): y (rows, len) = w / (||w||_2 + eps) per row, and
 * gw = d<gy, y>/dw.                                                                                           */
int gfdn_rownorm_fwd(const float* w, int rows, int len, float eps, float* y, void* stream);
int gfdn_rownorm_bwd(const float* w, int rows, int len, float eps, const float* gy, float* gw, void* stream);

/* ---- power-of-two inverse real FFT (utils.py:169 get_response, losses.py:344: default
 * n = 2(K-1)): x = irfft(X[0..n/2], n), n = 2^p >= 16 (imaginary parts of the DC and Nyquist
 * bins are ignored, as torch does), and its adjoint gX = d<gx, x>/dX (bins 0..n/2).
 * work: gfdn_irfft_pow2_work_bytes(n, batch).                                            */
size_t gfdn_irfft_pow2_work_bytes(int n, int batch);
int gfdn_irfft_pow2_fwd(int n, const float* X_c64, int ldx, int batch, float* x, int ldo,
                        void* work, void* stream);
/* ... of spectra that vanish from bin kvalid on (those bins are NOT read: they may be uninitialised), storing only the samples
 * t < tout (rounded up to a multiple of 512; the rest of x keeps what it held).  n = 131 072 skips the loads / stores; other
 * lengths run the whole transform and READ every bin (the caller zero-fills the upper band there).                           */
int gfdn_irfft_pow2_fwd_band(int n, const float* X_c64, int ldx, int kvalid, int batch, float* x, int ldo, int tout, void* work,
                             void* stream);
int gfdn_irfft_pow2_bwd(int n, const float* gx, int ldo, int batch, float* gX_c64, int ldx,
                        void* work, void* stream);
/* The adjoint for a gradient that vanishes outside the samples [t_lo, t_hi) (the EDC window of losses.py:344-346): nothing
 * outside the window is read -- it may be uninitialised.  n = 131 072 (GFDN_E_UNSUPPORTED otherwise).               */
int gfdn_irfft_pow2_bwd_window(int n, const float* gx, int ldo, int batch, int t_lo, int t_hi, float* gX_c64, int ldx,
                               void* work, void* stream);

/* Dataset front end (dataloader.py:250, :320-325: scipy.fft.rfft(rirs, n = nfft)):
 * X (batch, ldx >= n/2+1) complex64 = rfft(x[b][0:T] zero-padded to n), n = 2^p, T <= n.
 * work: gfdn_irfft_pow2_work_bytes(n, batch).                                                  */
int gfdn_rfft_pow2(int n, const float* x, int ld, int T, int batch, float* X_c64, int ldx,
                   void* work, void* stream);

/* ---- EDR  (losses.py:501-575: STFT Hann(win) hop win/2 center=False, tail energy, dB) ----
 * x: (batch, ld) float, T valid samples, implicitly zero-padded to a multiple of hop.
 * nframes = gfdn_stft_nframes(T, win).  P: (batch, nframes, win/2+1) float = |STFT|^2.
 * zero_buf: optional (batch, ld) float buffer cleared by the same launch (the accumulation
 * buffer gfdn_stft_power_bwd will add into), or NULL.                                     */
int gfdn_stft_nframes(int T, int win);
int gfdn_stft_power(const float* x, int ld, int T, int batch, int win, float* P, float* zero_buf,
                    void* stream);
/* in place: P -> EDR in dB (10 log10(sum_{tau>=m} P + eps), clipped at -200); sum_abs[b] =
 * sum |EDR| (the per-item normaliser of losses.py:487-490).                               */
size_t gfdn_edr_work_bytes(int batch, int nfreq);
int gfdn_edr_target(float* P_inout, int batch, int nframes, int nfreq, float* sum_abs,
                    void* work, void* stream);
/* achieved side: loss_item[b] = sum_{f,m} wf[f] |T_db - EDR| / sum_abs[b]  (losses.py:478-492)
 * and, when want_grad, P is overwritten with gscale * dloss/dP.  target_rows: row indirection
 * into T_db / sum_abs (see gfdn_compose_fwd) or NULL.  loss_item = NULL defers the last reduction:
 * work then holds (batch, ceil(nfreq/256)) partial sums of wf |T_db - EDR|, NOT yet divided by
 * sum_abs, for gfdn_weighted_sums (one launch less between this call and the STFT adjoint). */
int gfdn_edr_loss(float* P_inout, const float* T_db, const float* sum_abs,
                  const long long* target_rows, const float* wf, int batch, int nframes, int nfreq, float gscale, int want_grad,
                  float* loss_item, void* work, void* stream);
/* adjoint of gfdn_stft_power: gx[b][t] += dL/dx from gP (atomic adds of exactly two frames
 * per sample, hence order-independent).                                                   */
int gfdn_stft_power_bwd(const float* x, int ld, int T, int batch, int win, const float* gP,
                        float* gx, void* stream);

/* ---- EDC  (losses.py:187-238: Schroeder integral of x[start:start+len]^2, dB, mean |diff|)
 * T_db (batch, len) float: target EDC in dB from gfdn_edc_target.
 * maskw (len) float weights (1 = kept index, 0 = dropped; losses.py:221-227) or NULL;
 * inv_count = 1 / (global batch * number of kept indices).
 * loss_item[b] = inv_count * sum_i maskw_i |T_db - EDC_db|; gx (batch, ld), when not NULL,
 * is fully overwritten with gscale * dloss/dx (zeros outside the window).                 */
size_t gfdn_edc_work_bytes(int batch);
int gfdn_edc_target(const float* x, int ld, int batch, int start, int len, float* T_db,
                    void* work, void* stream);
int gfdn_edc_loss(const float* x, int ld, int batch, int start, int len, const float* T_db,
                  const long long* target_rows, const float* maskw, float inv_count, float gscale, float* loss_item,
                  float* gx, void* work, void* stream);
/* Band bank whose bands have different longest decay times (src/diff_gfdn/trainer.py:56-59: every band's trainer derives
 * max_ir_len_ms -- its EDC window -- from ITS OWN T60max; src/run_subband_training_treble.py:286 hands every band its own
 * dataset's decay times).  Item b's window is [start, start + item_len[b]) with item_len[b] <= max_len (device int32, one
 * entry per item); T_db rows have pitch ld_T >= max_len; band b / items_per_band reads the mask row
 * maskw + band * ld_mask (ld_mask = 0: one row shared by all bands).  _pairs_: the pair-interleaved layout of
 * gfdn_edc_loss_pairs, both items of a pair in one band (items_per_band even).  Everything else as gfdn_edc_loss.  */
int gfdn_edc_loss_banded(const float* x, int ld, int batch, int start, int max_len, const int* item_len,
                         const float* T_db, int ld_T, const long long* target_rows, const float* maskw, int ld_mask,
                         int items_per_band, float inv_count, float gscale, float* loss_item, float* gx, void* work,
                         void* stream);
int gfdn_edc_loss_pairs_banded(const float* x2, int ld, int items, int start, int max_len, const int* item_len,
                               const float* T_db, int ld_T, const long long* target_rows, const float* maskw,
                               int ld_mask, int items_per_band, float inv_count, float gscale, float* loss_item,
                               float* gx2, void* work, void* stream);
/* gfdn_edc_loss against the common-slope MODEL of the directional loss (losses.py:354-359) instead of a stored target:
 * target EDC of item b = sum_k amps[b][k] env[k][t] (amps (batch, S), env (S, ld_env >= len)), |.| + eps in dB clipped
 * at -200, evaluated inside the scan -- the (batch, len) target and the passes that build it never exist.  */
int gfdn_edc_loss_model(const float* x, int ld, int batch, int start, int len, const float* amps, int S,
                        const float* env, int ld_env, const float* maskw, float inv_count, float gscale,
                        float* loss_item, float* gx, void* work, void* stream);
/* The same loss on the directional signals x_dir[b][j] = sum_c A[j][c] x_sh[b][c] (trainer.py:853-865: einsum
 * 'jl,blk->bjk' with the real analysis matrix A (J, C), applied here behind the inverse transform -- the maps commute)
 * WITHOUT forming them: x_sh (B C, ld) holds the C SH-domain time signals of every receiver, the J directional samples
 * live in registers.  amps (B J, S); loss_item (B J); gx_sh (B C, ld), when not NULL, receives gscale * dloss/dx_sh on the
 * window samples [start, start + len) ONLY (pair it with gfdn_irfft_pow2_bwd_window).  loss_item carries gscale as
 * well (the weighted term).  C in {1, 4, 9, 16}, J <= 16, S <= 8.
 * work: gfdn_edc_mixed_work_bytes(B, J, len).                                                                          */
size_t gfdn_edc_mixed_work_bytes(int B, int J, int len);
int gfdn_edc_loss_model_mixed(const float* x_sh, int ld, int B, int C, const float* A, int J, int start, int len,
                              const float* amps, int S, const float* env, int ld_env, const float* maskw,
                              float inv_count, float gscale, float* loss_item, float* gx_sh, void* work, void* stream);
/* The same in two separately launched stages (bit mask; 3 = gfdn_edc_loss_model_mixed): 1 = the forward chain (segment
 * energies, carries, loss, carries; work keeps the carries), 2 = the backward kernel k_em_bwd alone (after a stage-1 call on
 * the same arguments and work) -- so that a caller can put events around the one kernel (bench.py's roofline leg).     */
int gfdn_edc_loss_model_mixed_stages(const float* x_sh, int ld, int B, int C, const float* A, int J, int start, int len,
                                     const float* amps, int S, const float* env, int ld_env, const float* maskw,
                                     float inv_count, float gscale, float* loss_item, float* gx_sh, void* work, int stages,
                                     void* stream);

/* ---- The directional model's output stage in the time domain (csrc/dirlin.hip; reference model.py:1056-1088,
 * trainer.py:853-865, losses.py:333-371).  H_sh[b][l] = filt sum_g w[b][g nper + l] c[g nper + l] Y[:, g nper + l] is linear
 * in the receiver's SH weights, so irfft(H_sh[b][l]) = sum_g w[b][g nper + l] tau[g nper + l] with the N = G nper LINE
 * signals tau[n] = irfft(c_n filt Y[:, n]): N transforms per step instead of B nper, no (B, nper, K) responses.
 *   gfdn_dirlin_lines_fwd  : Z (N, ldz >= K) complex = c_n filt_k Y[k][n]  (Y (K, N) complex bin-major, filt NULL: none);
 *   gfdn_dirlin_combine    : x (B nper, ld_x) float, WINDOW samples only: x[b nper + l][t] = sum_g w[b][g nper + l]
 *                            tau[g nper + l][start + t], t < len (ld_x a multiple of 4, >= len rounded up to 4; x 16-byte
 *                            aligned) -- the signals gfdn_edc_loss_model_mixed takes with start = 0;
 *   gfdn_dirlin_gamma_dots : from gx (B nper, ld_g) on the window: gtau (N, ld_o)[n][start + t] = sum_b w[b][n]
 *                            gx[b nper + l(n)][t] (written on the window only: gfdn_irfft_pow2_bwd_window) and
 *                            gw (B, N) = <gx[b nper + l(n)], tau[n]>; part: B N gfdn_dirlin_tiles(len) floats;
 *   gfdn_dirlin_lines_bwd  : gY (K, N) = c_n conj(filt_k) gZ[n][k], gc (N) = sum_k Re(gZ conj(filt Y));
 *                            gc_part: N gfdn_dirlin_line_tiles(K) floats.
 * G <= 4, N <= 64.  All sums in fixed order.                                                                           */
int gfdn_dirlin_tiles(int len);
int gfdn_dirlin_line_tiles(int K);
int gfdn_dirlin_lines_fwd(const float* Y_c64, int K, int N, const float* c, const float* filt_c64, float* Z_c64, int ldz,
                          void* stream);
int gfdn_dirlin_lines_bwd(const float* Y_c64, int K, int N, const float* c, const float* filt_c64, const float* gZ_c64,
                          int ldz, float* gY_c64, float* gc, float* gc_part, void* stream);
int gfdn_dirlin_combine(const float* tau, int ld_tau, int start, int len, const float* w, int B, int G, int nper,
                        float* x, int ld_x, void* stream);
int gfdn_dirlin_gamma_dots(const float* gx, int ld_g, int len, const float* tau, int ld_tau, int start, const float* w,
                           int B, int G, int nper, float* gtau, int ld_o, float* gw, float* part, void* stream);

/* ---- group sums of the sub-FDN responses (model.py:243-250: Hout[k][g] = sum_{n in g} c_n y_n[k]): S (G, K) complex from
 * Y (K, N = G nper) complex bin-major; backward: gY (K, N) = c_n gS[g(n)][k], gc (N) = sum_k Re(conj(gS) Y)
 * (gc_part: N gfdn_dirlin_line_tiles(K) floats).  The G-receivers-with-identity-gains form of gfdn_compose_fwd / _bwd
 * without the receiver machinery.  N <= 64.                                                                            */
int gfdn_group_sums_fwd(const float* Y_c64, int K, int G, int nper, const float* c, float* S_c64, void* stream);
int gfdn_group_sums_bwd(const float* Y_c64, int K, int G, int nper, const float* c, const float* gS_c64, float* gY_c64,
                        float* gc, float* gc_part, void* stream);

/* ---- EDC time mask on the device  (losses.py:221-227: mask = argwhere(bernoulli(U(0,1))) over the
 * window -- marginally every index is kept with probability 1/2, independently).
 * Counter-based draw: bit t of the mask is bit (t mod 128) of Philox4x32-10(counter = (t / 128, 0,
 * step_lo, step_hi), key = (seed_lo, seed_hi)), step = state[0] (uint64, device), which the kernel
 * then increments -- so a HIP-graph replay draws a fresh mask without any host work, and every rank
 * of a data-parallel job that starts from the same (seed, state) draws the SAME mask with no broadcast.
 * maskw[t] = bit_t * scale / count, count = number of kept indices (all zeros if count == 0);
 * scale = 1 / global batch makes maskw the "pre-normalised" weights gfdn_edc_loss takes with
 * inv_count = 1.  len <= 131072.                                                                  */
int gfdn_draw_mask(unsigned long long seed, unsigned long long* state, int len, float scale,
                   float* maskw, void* stream);
/* One row of weights per band (the windows of gfdn_edc_loss_banded) from ONE draw of max_len bits -- the bits
 * gfdn_draw_mask draws at the same (seed, step): band q keeps the first band_len[q] (device int32, <= max_len) of them,
 * maskw[q][t] = bit_t * scale / (kept among the first band_len[q]), zero from band_len[q] to the row pitch
 * ld_mask >= max_len.  (The reference's band trainers each draw their own mask, losses.py:221-223; sharing the bits
 * between bands keeps every band's marginal distribution and the equal-length case bit-identical to gfdn_draw_mask.) */
int gfdn_draw_mask_banded(unsigned long long seed, unsigned long long* state, const int* band_len, int nbands,
                          int max_len, int ld_mask, float scale, float* maskw, void* stream);

/* ---- receiver-position -> group-gain network  (gain_filters.py:497-534; dnn.py:89-126, :331-400,
 * :21-36).  pos (B,3) float64 normalised coordinates; freq_pi (F) float32 = f32(freq_k * pi);
 * w: all parameters packed in named_parameters() order
 *   [W0 (H x 6F) | b0 | gamma0 | beta0 | W1 (H x H) | b1 | gamma1 | beta1 | ... | Wout (G x H) | bout];
 * pos_rows: row indirection into pos (see gfdn_compose_fwd) or NULL.
 * gains (B,G) = lo + (hi-lo) sigmoid(MLP(encoding(pos))).  xhat (B, 1+n_hidden, H) and rstd
 * (B, 1+n_hidden) are saved for the backward, which returns gw (same packing as w).        */
size_t gfdn_mlp_param_count(int F, int H, int n_hidden, int G);
size_t gfdn_mlp_bwd_work_bytes(int B, int F, int H, int n_hidden, int G);
int gfdn_mlp_gains_fwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                       const float* w, int B, int F, int H, int n_hidden, int G, float lo, float hi, float* gains, float* xhat,
                       float* rstd, void* stream);
int gfdn_mlp_gains_bwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                       const float* w, int B, int F, int H, int n_hidden, int G, float lo, float hi, const float* gains,
                       const float* xhat, const float* rstd, const float* ggains, float* gw,
                       void* work, void* stream);

/* band-stacked: w (nbands, P) one parameter set per band, item i = band * Bper + b uses set i / Bper;
 * gains / xhat / rstd / ggains have nbands * Bper rows; gw (nbands, P);
 * work: gfdn_mlp_bwd_work_bytes(nbands * Bper, ...).                                           */
int gfdn_mlp_gains_banded_fwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                              const float* w, int nbands, int Bper, int F, int H, int n_hidden, int G,
                              float lo, float hi, float* gains, float* xhat, float* rstd, void* stream);
int gfdn_mlp_gains_banded_bwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                              const float* w, int nbands, int Bper, int F, int H, int n_hidden, int G,
                              float lo, float hi, const float* gains, const float* xhat,
                              const float* rstd, const float* ggains, float* gw, void* work,
                              void* stream);
/* The same with dL/dgains handed over as (nbands Bper G, gparts) partial rows (gfdn_tf_gain_grad with grgain = NULL): every
 * receiver's wave sums its rows itself, same terms in the same order as the separate row-sum launch it replaces.
 * Wave-per-receiver form only (H, G <= 64, Bper a multiple of 4, parameters staged in LDS): GFDN_E_UNSUPPORTED otherwise. */
int gfdn_mlp_bwd_takes_parts(int F, int H, int n_hidden, int G, int Bper);      /* 1: the call below takes this network */

int gfdn_mlp_gains_banded_bwd_parts(const double* pos, const long long* pos_rows, const float* freq_pi,
                                    const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                                    int G, float lo, float hi, const float* gains, const float* xhat,
                                    const float* rstd, const float* ggains_parts, int gparts, float* gw,
                                    void* work, void* stream);
/* normalize's scale folded into the receiver gains (gfdn_tf_energy_gains): the partial rows hold dL/d(gains colscale), every
 * row sum is multiplied by colscale[band G + g] first.  Wave-per-receiver form only (gfdn_mlp_bwd_takes_parts).             */
int gfdn_mlp_gains_banded_bwd_parts_scaled(const double* pos, const long long* pos_rows, const float* freq_pi, const float* w,
                                           int nbands, int Bper, int F, int H, int n_hidden, int G, float lo, float hi,
                                           const float* gains, const float* xhat, const float* rstd,
                                           const float* ggains_parts, int gparts, const float* colscale, float* gw, void* work,
                                           void* stream);

/* Bands with DIFFERENT layer sizes in ONE launch (round 6): the reference's sub-band driver gives every band its own gain
 * network (src/run_subband_training_treble.py:61-73: 1 x 8, 1 x 16, 5 x 16, 3 x 128 hidden layers x neurons).  H[q],
 * n_hidden[q] (host arrays, nbands <= 16): band q's sizes; F, G, lo, hi shared.  w / gw: the bands' packed parameter sets one
 * after the other; xhat / rstd: the bands' (Bper, nl_q, H_q) / (Bper, nl_q) blocks one after the other.
 * gfdn_mlp_bands_sizes: sizes[0..3] = floats of w, xhat, rstd and of the backward's work buffer.  Backward: ggains (items, G)
 * summed gradients (gparts = 0) or (items G, gparts) partial rows summed in the launch; colscale (optional, (nbands G))
 * multiplies every row sum (normalize's scale folded into the gains, gfdn_tf_energy_gains).                                  */
int gfdn_mlp_bands_sizes(int nbands, int Bper, int F, const int* H, const int* n_hidden, int G, size_t* sizes);
int gfdn_mlp_gains_bands_fwd(const double* pos, const long long* pos_rows, const float* freq_pi, const float* w, int nbands,
                             int Bper, int F, const int* H, const int* n_hidden, int G, float lo, float hi, float* gains,
                             float* xhat, float* rstd, void* stream);
int gfdn_mlp_gains_bands_bwd(const double* pos, const long long* pos_rows, const float* freq_pi, const float* w, int nbands,
                             int Bper, int F, const int* H, const int* n_hidden, int G, float lo, float hi,
                             const float* gains, const float* xhat, const float* rstd, const float* ggains, int gparts,
                             const float* colscale, float* gw, void* work, void* stream);

/* Receiver schedule of a replayed epoch (reference trainer.py:373-379: the DataLoader fixes an epoch's batches when the
 * epoch starts): table (len, B) int64 dataset rows uploaded once; each call copies row state[0] mod state[1] into idx
 * (B) and advances state[0], on the device.  state: two int64 {position, len}.  */
int gfdn_pick_rows(const long long* table, long long* state, long long* idx, int B, void* stream);
/* ---- optimiser step  (trainer.py:152-228, :475: torch.optim.Adam with per-name lr groups) ----------
 * All parameters are views into one flat fp32 buffer p (n floats), gradients into g, Adam moments
 * into m, v.  seg[i] (uint8) = learning-rate group of element i, lr_seg[group] the group's lr (device,
 * so that StepLR edits it without re-recording a HIP graph), step_count (device float) the number of
 * updates done so far -- advanced by one by this call.                                          */
int gfdn_adam_step(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                   const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                   float eps, void* stream);
/* the same in ONE launch for any n: block_counter = a zero-initialised uint32 of the caller's (the last workgroup to
 * finish advances step_count and re-arms it)                                                              */
int gfdn_adam_step_counted(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                           const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                           float eps, unsigned int* block_counter, void* stream);
/* the same, also writing the advanced count to `mirror` (a second counter of the caller's that must stay equal to
 * step_count: an optimiser that steps two ranges of its buffer from two streams keeps one counter per range)  */
int gfdn_adam_step_mirrored(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                            const float* lr_seg, float* step_count, float* mirror, int n, float beta1, float beta2,
                            float eps, unsigned int* block_counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFGFDN_HIP_H */
