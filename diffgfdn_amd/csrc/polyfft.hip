// Block transfer functions of 5..8-line blocks on the reference's OWN frequency grid by fast transforms, for gfx950.
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn): the group transfer function T = c^T X^-1 b of a block,
// X(z) = D(z) Gamma^-1 - A (feedback_loop.py:326-391, model.py:583-619; sub-FDN responses of the colorless loss and of
// normalize: model.py:209-252, trainer.py:317-332), evaluated on z_k = e^{2 pi i k / nfft}, k = 0 .. nfft/2
// (dataloader.py:552-566: rfftfreq(nfft)).  csrc/blocktf8.hip writes T = P / Q with the two multilinear polynomials
//     Q(z) = sum_S Q_S z^{m_S} ,  P(z) = sum_S P_S z^{m_S} ,  m_S = sum_{i in S} m_i  (256 subsets S of the 8 lines)
// and evaluates them bin by bin on the matrix cores: 512 real x complex products per bin and polynomial.  The delay
// lengths m_i of the reference are INTEGERS (config.py:131-140: primes), and on the rfftfreq grid z_k^{m} is a root of
// unity of order nfft: Q(z_k) is bin k of the inverse-sign DFT of the real sequence q[m] = sum_{S: m_S = m mod nfft} Q_S,
//     Q(z_k) = conj( rfft(q, nfft)[k] ) ,
// a sequence of 256 non-zero samples -- one real transform of length nfft per polynomial instead of 256 nfft/2 products.
// The gradient is the transposed transform: with dL = sum_k Re(conj(g_k) dT_k), u = conj(g) / Q,
//     dL/dP_S = sum_k Re(u_k z_k^{m_S}) = G_u[m_S] ,   dL/dQ_S = G_v[m_S] , v = -u T ,
//     G_u = irfft(w u, nfft) ,  w_k = nfft / 2 (nfft at k = 0 and k = nfft / 2: the two bins an inverse real transform
// counts once) -- one inverse real transform per record set and a gather of 256 samples.
// The transforms are csrc/pow2.hip's (gfdn_rfft_pow2 / gfdn_irfft_pow2_fwd: two register-resident passes at nfft = 131 072);
// this file holds what is around them: the sparse sequences, the pointwise stages on the bins and the gathers.
// Layouts: X (2 nblk, ldx) complex64 = rfft of [Q | P] of one record set, nblk rows each; record layout of
// csrc/blocktf8.hip (coef (nblk, 9, 256): determinant polynomial and the eight numerators Y_i; gradient records (nblk, 512):
// dL/dP_S | dL/dQ_S).  Scaling convention as there: the sequences are built from the gains BEFORE normalize's rescale,
// T' = scale T.
// What is NOT here: the forward group responses of the damped loop (gfdn_tf8_tsave keeps them).  A float32 transform
// carries an absolute error of ~1e-6 |q|; next to the loop's poles, where |Q| is small, that is 1e-5 of T -- a hundred times
// the rounding of the direct evaluation -- and the dB stages of the decay losses carry it into dL/dM (measured at the
// bench's size: 1.15e-3 against 7.6e-4 of its largest entry).  normalize and the colorless loss sum |T| over all bins, the
// adjoints are linear in 1 / Q: none of them notices.
#include "common.h"

#define PF_PARTS 64
// blocks of <= 8 lines: 256 subsets, records (9, 256) (csrc/blocktf8.hip); 9 lines: 512 subsets, records (10, 512)
// (csrc/blocktf9.hip).  L = number of subset bits
static inline int pf_bits(int nper) { return nper <= 8 ? 8 : 9; }
static inline bool pf_nsub_ok(int nsub, int nper) { return (nsub == 256 && nper <= 8) || (nsub == 512 && nper <= 9); }

extern "C" int gfdn_tfp_parts(void) { return PF_PARTS; }

__device__ __forceinline__ int pf_degree(const float* __restrict__ delays, int blk, int n, int S, int nfft) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (i < n && ((S >> i) & 1)) m += (int)rintf(delays[blk * n + i]);
  return m & (nfft - 1);                 // z_k^nfft = 1
}

// grid (nblk, 2 polynomials): the sequence of one polynomial, T samples (zero but for <= 256 of them).  Subsets of equal
// degree are added in ascending subset order by the first of them (fixed order, no atomics): the keys (degree, subset)
// are sorted in LDS (bitonic, 36 compare-exchange stages at 256 subsets), equal degrees then sit side by side.
__global__ __launch_bounds__(512) void k_pf_sparse(const float* __restrict__ coefs, const float* __restrict__ delays,
                                                   const float* __restrict__ c, int nblk, int n, int nfft, int T,
                                                   float* __restrict__ seq) {
  __shared__ unsigned key[512];
  __shared__ float sv[512];
  const int nsub = blockDim.x, L = nsub == 512 ? 9 : 8;             // (one thread per subset)
  const int blk = blockIdx.x, poly = blockIdx.y, S = threadIdx.x;
  const float* coef = coefs + (size_t)blk * (L + 1) * nsub;
  float* row = seq + ((size_t)blockIdx.y * nblk + blk) * T;
  for (int t = S; t < T; t += nsub) row[t] = 0.f;
  const int m = pf_degree(delays, blk, n, S, nfft);
  float v;
  if (poly == 0) {
    v = coef[S];
  } else {
    v = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) v += (i < n ? c[blk * n + i] * coef[(1 + i) * nsub + S] : 0.f);
  }
  key[S] = ((unsigned)m << 9) | (unsigned)S;
  sv[S] = v;
  __syncthreads();
  for (int k = 2; k <= nsub; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = S ^ j;
      if (p > S) {
        const unsigned a = key[S], b = key[p];
        const bool up = (S & k) == 0;
        if ((a > b) == up) { key[S] = b; key[p] = a; }
      }
      __syncthreads();
    }
  const unsigned mine = key[S];
  if (S == 0 || (key[S - 1] >> 9) != (mine >> 9)) {            // first of its degree: sums the run (ascending subsets)
    float s = sv[mine & 511u];
    for (int q = S + 1; q < nsub && (key[q] >> 9) == (mine >> 9); ++q) s += sv[key[q] & 511u];
    row[mine >> 9] = s;
  }
}

extern "C" int gfdn_tfp_forward(int nfft, int nblk, int nper, int nsub, const float* coef, const float* delays, const float* c,
                                int T, float* seq, float* X_c64, int ldx, void* work, void* stream) {
  if (!pf_nsub_ok(nsub, nper)) return nper > 9 ? GFDN_E_UNSUPPORTED : GFDN_E_BADARG;
  if (!coef || !delays || !c || !seq || !X_c64 || !work || nblk <= 0 || nper <= 0 || nfft < 16 || (nfft & (nfft - 1)) ||
      T <= 0 || T > nfft || ldx < nfft / 2 + 1)
    return GFDN_E_BADARG;
  if (nper > 9) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pf_sparse, dim3(nblk, 2), dim3(nsub), 0, s, coef, delays, c, nblk, nper, nfft, T, seq);
  GFDN_LAUNCH_CHECK();
  return gfdn_rfft_pow2(nfft, seq, T, T, 2 * nblk, X_c64, ldx, work, stream);
}

// ------------------------------------------------------------------------------------------
// normalize (trainer.py:317-332): E = mean_k |P / Q|^2 of the raw sub-FDN blocks -> scale = E^(-1/2), b, c / E^(1/4)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_energy(const float2* __restrict__ Xq, const float2* __restrict__ Xp, int ldx,
                                                   int K, float* __restrict__ part) {
  __shared__ float s_r[16];
  const int blk = blockIdx.y;
  const float2* q = Xq + (size_t)blk * ldx;
  const float2* p = Xp + (size_t)blk * ldx;
  float acc = 0.f;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    const float2 a = p[k], d = q[k];
    acc += (a.x * a.x + a.y * a.y) / (d.x * d.x + d.y * d.y);
  }
  acc = block_sum(acc, s_r);
  if (threadIdx.x == 0) part[(size_t)blk * gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_pf_energy_finish(const float* __restrict__ partial, int nparts, int K, int nper,
                                                          float* __restrict__ b, float* __restrict__ c,
                                                          float* __restrict__ energy, float* __restrict__ scale,
                                                          const float* __restrict__ gains, float* __restrict__ gains_scaled,
                                                          int Bper, int G) {
  __shared__ float s_r[16];
  const int g = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)g * nparts + p];
  s = block_sum(s, s_r);
  const float E = s / (float)K;
  if (threadIdx.x == 0) {
    if (energy) energy[g] = E;
    if (scale) scale[g] = 1.0f / sqrtf(E);
  }
  if (gains_scaled) {                 // (the scale folded into the receiver gains: gfdn_tf_energy_gains)
    const float sc = 1.0f / sqrtf(E);
    const int band = g / G, col = g - band * G;
    for (int r = threadIdx.x; r < Bper; r += 256) {
      const size_t i = ((size_t)band * Bper + r) * G + col;
      gains_scaled[i] = gains[i] * sc;
    }
  }
  const float d = powf(E, 0.25f);
  for (int i = threadIdx.x; i < nper; i += 256) {
    b[g * nper + i] /= d;
    c[g * nper + i] /= d;
  }
}

extern "C" int gfdn_tfp_energy(const float* Xq_c64, const float* Xp_c64, int ldx, int K, int nblk, int nper, float* b,
                               float* c, float* energy, float* scale, void* work, const float* gains, float* gains_scaled,
                               int Bper, int G, void* stream) {
  if (!Xq_c64 || !Xp_c64 || !b || !c || !scale || !work || K <= 0 || ldx < K || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (gains_scaled && (!gains || Bper <= 0 || G <= 0 || nblk % G)) return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pf_energy, dim3(PF_PARTS, nblk), dim3(256), 0, s, (const float2*)Xq_c64, (const float2*)Xp_c64, ldx, K,
                     (float*)work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pf_energy_finish, dim3(nblk), dim3(256), 0, s, (const float*)work, PF_PARTS, K, nper, b, c, energy, scale,
                     gains, gains_scaled, Bper, G);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// gradient records from the two inverse transforms: part[blk * 2 nsub + S] = G_u[blk][m_S], part[.. + nsub + S] = G_v[blk][m_S]
// (x: rows [0, nblk) = G_u, [nblk, 2 nblk) = G_v)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_pf_gather(const float* __restrict__ x, int ldt, const float* __restrict__ delays,
                                                   int nblk, int n, int nfft, float* __restrict__ part) {
  const int blk = blockIdx.x, S = threadIdx.x, nsub = blockDim.x;
  const bool absent = (S >> n) != 0;
  const int m = pf_degree(delays, blk, n, S, nfft);
  part[(size_t)blk * 2 * nsub + S] = absent ? 0.f : x[(size_t)blk * ldt + m];
  part[(size_t)blk * 2 * nsub + nsub + S] = absent ? 0.f : x[(size_t)(nblk + blk) * ldt + m];
}

// ------------------------------------------------------------------------------------------
// colorless pass (colorless_fdn/losses.py:20-73 on the scaled sub-FDN responses, as k_tf8_pass<T8_COLORLESS>): loss
// partials and the two weighted spectra w u, w v of the inverse transforms
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_colorless(const float2* __restrict__ Xq, const float2* __restrict__ Xp, int ldx,
                                                      int K, int nblk, const float* __restrict__ scale, int asym,
                                                      float gscale, float2* __restrict__ UV, float* __restrict__ lossp) {
  __shared__ float s_r[16];
  const int blk = blockIdx.y;
  const float sc = scale ? scale[blk] : 1.0f;
  const float invK = 1.0f / (float)K, half = (float)(K - 1);          // nfft / 2
  float acc = 0.f;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    const float2 q = cconj(Xq[(size_t)blk * ldx + k]), p = cconj(Xp[(size_t)blk * ldx + k]);
    const float2 dinv = cinv(q);
    const float2 t = cscale(cmul(p, dinv), sc);
    const float mag = sqrtf(t.x * t.x + t.y * t.y);
    const float d = mag - 1.0f, d2 = d * d;
    const bool four = asym && (d > 1.0f);
    const float dl = four ? 4.0f * d2 * d : 2.0f * d;
    const float f = mag > 0.f ? gscale * invK * dl / mag : 0.f;
    const float2 gs = make_float2(f * t.x, f * t.y);
    acc += (four ? d2 * d2 : d2) * invK;
    const float w = (k == 0 || k == K - 1) ? 2.0f * half : half;
    const float2 u = cscale(make_float2(gs.x * dinv.x + gs.y * dinv.y, gs.x * dinv.y - gs.y * dinv.x), w);   // w conj(gs) / Q
    const float2 v = cmul(u, t);
    UV[(size_t)blk * ldx + k] = u;
    UV[(size_t)(nblk + blk) * ldx + k] = make_float2(-v.x, -v.y);
  }
  acc = block_sum(acc, s_r);
  if (threadIdx.x == 0) lossp[(size_t)blk * gridDim.x + blockIdx.x] = acc;
}

extern "C" int gfdn_tfp_colorless(const float* Xq_c64, const float* Xp_c64, int ldx, int nfft, int nblk, int nper,
                                  const float* delays, const float* scale, int asym, float gscale, int T, float* UV_c64,
                                  float* x, int ldt, void* work, float* part, float* lossp, float* loss, void* stream) {
  if (!Xq_c64 || !Xp_c64 || !delays || !UV_c64 || !x || !work || !part || !lossp || !loss || nblk <= 0 || nper <= 0 ||
      nfft < 16 || (nfft & (nfft - 1)) || ldx < nfft / 2 + 1 || ldt < nfft)
    return GFDN_E_BADARG;
  if (nper > 8) return GFDN_E_UNSUPPORTED;          // (the records of csrc/blocktf8.hip: 256 subsets)
  hipStream_t s = (hipStream_t)stream;
  const int K = nfft / 2 + 1;
  hipLaunchKernelGGL(k_pf_colorless, dim3(PF_PARTS, nblk), dim3(256), 0, s, (const float2*)Xq_c64, (const float2*)Xp_c64, ldx, K,
                     nblk, scale, asym, gscale, (float2*)UV_c64, lossp);
  GFDN_LAUNCH_CHECK();
  if (T <= 0 || T > nfft) return GFDN_E_BADARG;
  int rc = gfdn_irfft_pow2_fwd_band(nfft, UV_c64, ldx, K, 2 * nblk, x, ldt, T, work, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pf_gather, dim3(nblk), dim3(1 << pf_bits(nper)), 0, s, (const float*)x, ldt, delays, nblk, nper, nfft, part);
  GFDN_LAUNCH_CHECK();
  return gfdn_tf_rows_sum(lossp, PF_PARTS, nblk, loss, stream);
}

// ------------------------------------------------------------------------------------------
// adjoint of the output stage of the linear step: gH (nblk, ldh) = dL/d(T'_g filt) on the slot order (the adjoint pair
// transform's output) -> gradient records of the damped blocks.  Thread = bin: the slot's gradient is gathered, the
// spectra of the inverse transforms are written where the transform reads them (zero above bin Ku - 1).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_bwd_spectra(const float2* __restrict__ gH, int ldh, const float2* __restrict__ filt,
                                                        int ldf, const float2* __restrict__ Tnat,
                                                        const float2* __restrict__ Dnat, int G, int Ku, int K,
                                                        const int* __restrict__ slot_of_bin, int nblk,
                                                        const float* __restrict__ tscale, int fold, int kfill,
                                                        float2* __restrict__ UV, int ldx) {
  const int blk = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kfill) return;               // (kfill = Ku where the inverse transform skips the bins above, else K: zeros there)
  float2 u = make_float2(0.f, 0.f), v = u;
  if (k < Ku) {
    const int band = blk / G;
    const int so = slot_of_bin[k], col = so & 0x7fffffff;
    float2 g = gH[(size_t)blk * ldh + col];
    if (filt) g = cmulc(g, filt[(size_t)band * ldf + col]);            // dL/dT' of the slot
    if (so < 0) g = cconj(g);                                          // ... of the bin: Re(conj(g_s) dT_s), T_s = conj(T_k)
    const float2 dinv = Dnat[(size_t)blk * Ku + k];
    const float ts = tscale ? tscale[blk] : 1.0f;
    const float w = (k == 0 || k == K - 1) ? 2.0f * (float)(K - 1) : (float)(K - 1);
    u = cscale(make_float2(g.x * dinv.x + g.y * dinv.y, g.x * dinv.y - g.y * dinv.x), w);       // w conj(g) / Q
    if (fold) {                        // g = scale dL/dT' (the scale sits in the receiver gains): v = -(u / s)(s T), u' = u / s
      v = cmul(u, Tnat[(size_t)blk * Ku + k]);
      u = cscale(u, 1.0f / ts);
    } else {
      v = cmul(u, cscale(Tnat[(size_t)blk * Ku + k], ts));
    }
    v = make_float2(-v.x, -v.y);
  }
  UV[(size_t)blk * ldx + k] = u;
  UV[(size_t)(nblk + blk) * ldx + k] = v;
}

extern "C" int gfdn_tfp_compose_bwd(int nfft, int nbands, int G, int nper, const float* delays, int Ku, const int* slot_of_bin,
                                    const float* gH_c64, int ldh, const float* filt_c64, int ldf, const float* Tnat_c64,
                                    const float* Dnat_c64, const float* tscale, int gain_fold, int T, float* UV_c64,
                                    int ldx, float* x, int ldt, void* work, float* part, void* stream) {
  if (!delays || !slot_of_bin || !gH_c64 || !Tnat_c64 || !Dnat_c64 || !UV_c64 || !x || !work || !part || nbands <= 0 || G <= 0 ||
      nper <= 0 || nfft < 16 || (nfft & (nfft - 1)) || Ku <= 0 || Ku > nfft / 2 + 1 || ldh < Ku || (filt_c64 && ldf < Ku) ||
      ldx < nfft / 2 + 1 || ldt < nfft || (gain_fold && !tscale))
    return GFDN_E_BADARG;
  if (nper > 8) return GFDN_E_UNSUPPORTED;          // (the records of csrc/blocktf8.hip: 256 subsets)
  hipStream_t s = (hipStream_t)stream;
  const int K = nfft / 2 + 1, nblk = nbands * G;
  if (T <= 0 || T > nfft) return GFDN_E_BADARG;
  // (nfft = 131 072: the inverse transform's first pass does not read the bins from Ku on and its last pass stores the T
  // samples the gather reads; other lengths run the whole transform on a zero-filled upper band)
  const bool band = nfft == 131072;
  const int kfill = band ? Ku : K;
  hipLaunchKernelGGL(k_pf_bwd_spectra, dim3((kfill + 255) / 256, nblk), dim3(256), 0, s, (const float2*)gH_c64, ldh,
                     (const float2*)filt_c64, ldf, (const float2*)Tnat_c64, (const float2*)Dnat_c64, G, Ku, K, slot_of_bin, nblk,
                     tscale, gain_fold ? 1 : 0, kfill, (float2*)UV_c64, ldx);
  GFDN_LAUNCH_CHECK();
  int rc = gfdn_irfft_pow2_fwd_band(nfft, UV_c64, ldx, band ? Ku : K, 2 * nblk, x, ldt, T, work, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pf_gather, dim3(nblk), dim3(1 << pf_bits(nper)), 0, s, (const float*)x, ldt, delays, nblk, nper, nfft, part);
  GFDN_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------
// The transfer functions themselves and the adjoint of T = P / Q, for callers that keep their own loss on T (the directional
// model's colorless branch under autograd: model.py:209-252 through sub_fdn_group_sums): T (nblk, K) and 1 / Q (nblk, K) from
// the transformed sequences; gT = dL/dT in the convention dL = sum_k Re(conj(gT_k) dT_k) -> gradient records.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_ratio_fwd(const float2* __restrict__ Xq, const float2* __restrict__ Xp, int ldx,
                                                      int K, float2* __restrict__ T, float2* __restrict__ D) {
  const int blk = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float2 q = cconj(Xq[(size_t)blk * ldx + k]), p = cconj(Xp[(size_t)blk * ldx + k]);
  const float2 dinv = cinv(q);
  T[(size_t)blk * K + k] = cmul(p, dinv);
  D[(size_t)blk * K + k] = dinv;
}
__global__ __launch_bounds__(256) void k_pf_ratio_bwd(const float2* __restrict__ gT, int ldg, const float2* __restrict__ T,
                                                      const float2* __restrict__ D, int K, int nblk,
                                                      float2* __restrict__ UV, int ldx) {
  const int blk = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float2 g = gT[(size_t)blk * ldg + k], dinv = D[(size_t)blk * K + k], t = T[(size_t)blk * K + k];
  const float w = (k == 0 || k == K - 1) ? 2.0f * (float)(K - 1) : (float)(K - 1);
  const float2 u = cscale(make_float2(g.x * dinv.x + g.y * dinv.y, g.x * dinv.y - g.y * dinv.x), w);       // w conj(g) / Q
  const float2 v = cmul(u, t);
  UV[(size_t)blk * ldx + k] = u;
  UV[(size_t)(nblk + blk) * ldx + k] = make_float2(-v.x, -v.y);
}

extern "C" int gfdn_tfp_ratio_fwd(const float* Xq_c64, const float* Xp_c64, int ldx, int K, int nblk, float* T_c64,
                                  float* Dinv_c64, void* stream) {
  if (!Xq_c64 || !Xp_c64 || !T_c64 || !Dinv_c64 || K <= 0 || ldx < K || nblk <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_pf_ratio_fwd, dim3((K + 255) / 256, nblk), dim3(256), 0, (hipStream_t)stream, (const float2*)Xq_c64,
                     (const float2*)Xp_c64, ldx, K, (float2*)T_c64, (float2*)Dinv_c64);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_tfp_ratio_bwd(int nfft, int nblk, int nper, int nsub, const float* delays, const float* gT_c64, int ldg,
                                  const float* T_c64, const float* Dinv_c64, int T, float* UV_c64, int ldx, float* x, int ldt,
                                  void* work, float* part, void* stream) {
  if (!delays || !gT_c64 || !T_c64 || !Dinv_c64 || !UV_c64 || !x || !work || !part || nblk <= 0 || nper <= 0 || nfft < 16 ||
      (nfft & (nfft - 1)) || ldg < nfft / 2 + 1 || ldx < nfft / 2 + 1 || ldt < nfft)
    return GFDN_E_BADARG;
  if (!pf_nsub_ok(nsub, nper)) return nper > 9 ? GFDN_E_UNSUPPORTED : GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int K = nfft / 2 + 1;
  hipLaunchKernelGGL(k_pf_ratio_bwd, dim3((K + 255) / 256, nblk), dim3(256), 0, s, (const float2*)gT_c64, ldg,
                     (const float2*)T_c64, (const float2*)Dinv_c64, K, nblk, (float2*)UV_c64, ldx);
  GFDN_LAUNCH_CHECK();
  if (T <= 0 || T > nfft) return GFDN_E_BADARG;
  int rc = gfdn_irfft_pow2_fwd_band(nfft, UV_c64, ldx, K, 2 * nblk, x, ldt, T, work, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pf_gather, dim3(nblk), dim3(nsub), 0, s, (const float*)x, ldt, delays, nblk, nper, nfft, part);
  GFDN_LAUNCH_CHECK();
  return 0;
}
