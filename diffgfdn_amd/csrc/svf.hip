// Frequency responses of second-order-section cascades (the SVF output / input filters of the full-band models)
// and the output stage that contracts them with the group transfer functions.
//
// Reference: src/diff_gfdn/gain_filters.py:221-241 (SOSFilter.forward: prod_s (b0 + b1 z^-1 + b2 z^-2) /
// (a0 + a1 z^-1 + a2 z^-2) over the bins), :262-402 (SVF_from_MLP: one 11-section cascade per (receiver, group)),
// model.py:588-619 (H = sum_g Co[b][g][k] T[k][g] + d[b][k]).  The reference materialises (B, N, K) complex
// tensors section by section; here a thread evaluates the whole cascade of its (row, bin) in registers and the
// (B, G, K) responses never exist in memory:
//   * sections in float64 from the float32 coefficients (b0 + b1 z^-1 + b2 z^-2 cancels to O(f^2) at low
//     frequencies: in complex64 the shelves lose three digits there), each rounded to complex64, the running product
//     in complex64 -- the same numbers as diffgfdn_amd.gain_filters.svf_cascade_response;
//   * backward: dL/dT (a sum over the receivers) and dL/dcoef (a sum over the bins) in two launches that both
//     re-evaluate the cascades (8.4 M cascades of 11 sections: ~50 us each) instead of saving 11 x (B, G, K).
#include "common.h"

#define SOS_T 256
#define SOS_MAX_S 24          // sections per cascade (the SVF equaliser has 11; input x output filter of one group: 22)
#define SOS_MAX_G 8           // cascades (groups) per receiver
#define SOS_BCH 8             // receivers per workgroup in the dL/dT launch
#define SOS_KPT 4             // bins per thread in the dL/dcoef launch

__device__ __forceinline__ double2 dmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// 1 / x from the hardware estimate and two Newton steps (relative error ~1e-15 for normal x): the IEEE division
// sequence costs ~4 x as much and these kernels do one per section and bin
__device__ __forceinline__ double drcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double2 ddiv(double2 a, double2 b) {
  const double d = drcp(b.x * b.x + b.y * b.y);
  return make_double2((a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d);
}
__device__ __forceinline__ double2 dinv(double2 z) {
  const double d = drcp(z.x * z.x + z.y * z.y);
  return make_double2(z.x * d, -z.y * d);
}

// numerator / denominator of one section at z^-1 = zi, z^-2 = zi2; c = {b0, b1, b2, a0, a1, a2}
__device__ __forceinline__ void sos_section(const float* c, double2 zi, double2 zi2, double2& num, double2& den) {
  num = make_double2((double)c[0] + (double)c[1] * zi.x + (double)c[2] * zi2.x,
                     (double)c[1] * zi.y + (double)c[2] * zi2.y);
  den = make_double2((double)c[3] + (double)c[4] * zi.x + (double)c[5] * zi2.x,
                     (double)c[4] * zi.y + (double)c[5] * zi2.y);
}

// prod_s round_c64(num_s / den_s), accumulated in complex64
__device__ __forceinline__ float2 sos_cascade(const float* c, int S, double2 zi, double2 zi2) {
  float2 h = make_float2(1.f, 0.f);
  for (int s = 0; s < S; ++s) {
    double2 num, den;
    sos_section(c + 6 * s, zi, zi2, num, den);
    const double2 q = ddiv(num, den);
    const float2 sec = make_float2((float)q.x, (float)q.y);
    h = s == 0 ? sec : cmul(h, sec);
  }
  return h;
}

__device__ __forceinline__ void sos_stage(const float* __restrict__ src, int n, float* lds) {
  for (int e = threadIdx.x; e < n; e += blockDim.x) lds[e] = src[e];
  __syncthreads();
}

// out[r][k] = cascade_r(z_k)
__global__ __launch_bounds__(SOS_T) void k_sos_response(const float* __restrict__ coef, int S,
                                                        const double2* __restrict__ z, int K,
                                                        float2* __restrict__ out) {
  __shared__ float s_c[SOS_MAX_S * 6];
  const int r = blockIdx.y;
  sos_stage(coef + (size_t)r * S * 6, S * 6, s_c);
  const int k = blockIdx.x * SOS_T + threadIdx.x;
  if (k >= K) return;
  const double2 zi = dinv(z[k]), zi2 = dmul(zi, zi);
  out[(size_t)r * K + k] = sos_cascade(s_c, S, zi, zi2);
}

// H[b][k] = sum_g cascade_{b,g}(z_k) T[k][g] + direct[b][k]
__global__ __launch_bounds__(SOS_T) void k_sos_compose_fwd(const float* __restrict__ coef, int G, int S,
                                                           const double2* __restrict__ z, int K,
                                                           const float2* __restrict__ T,
                                                           const float2* __restrict__ direct, int ldd,
                                                           float2* __restrict__ H) {
  __shared__ float s_c[SOS_MAX_G * SOS_MAX_S * 6];
  const int b = blockIdx.y;
  sos_stage(coef + (size_t)b * G * S * 6, G * S * 6, s_c);
  const int k = blockIdx.x * SOS_T + threadIdx.x;
  if (k >= K) return;
  const double2 zi = dinv(z[k]), zi2 = dmul(zi, zi);
  float2 h = direct ? direct[(size_t)b * ldd + k] : make_float2(0.f, 0.f);
  for (int g = 0; g < G; ++g) {
    const float2 co = sos_cascade(s_c + g * S * 6, S, zi, zi2);
    const float2 t = T[(size_t)k * G + g];
    h.x += co.x * t.x - co.y * t.y;
    h.y += co.x * t.y + co.y * t.x;
  }
  H[(size_t)b * K + k] = h;
}

// partial[chunk][k][g] = sum_{b in chunk} conj(cascade_{b,g}(z_k)) gH[b][k]
__global__ __launch_bounds__(SOS_T) void k_sos_compose_bwd_t(const float* __restrict__ coef, int B, int G, int S,
                                                             const double2* __restrict__ z, int K,
                                                             const float2* __restrict__ gH,
                                                             float2* __restrict__ partial) {
  __shared__ float s_c[SOS_BCH * SOS_MAX_G * SOS_MAX_S * 6];
  const int b0 = blockIdx.y * SOS_BCH;
  const int nb = B - b0 < SOS_BCH ? B - b0 : SOS_BCH;
  sos_stage(coef + (size_t)b0 * G * S * 6, nb * G * S * 6, s_c);
  const int k = blockIdx.x * SOS_T + threadIdx.x;
  if (k >= K) return;
  const double2 zi = dinv(z[k]), zi2 = dmul(zi, zi);
  float2 acc[SOS_MAX_G];
#pragma unroll
  for (int g = 0; g < SOS_MAX_G; ++g) acc[g] = make_float2(0.f, 0.f);
  for (int bb = 0; bb < nb; ++bb) {
    const float2 gh = gH[(size_t)(b0 + bb) * K + k];
#pragma unroll
    for (int g = 0; g < SOS_MAX_G; ++g) {
      if (g < G) {
        const float2 co = sos_cascade(s_c + (bb * G + g) * S * 6, S, zi, zi2);
        acc[g].x += co.x * gh.x + co.y * gh.y;          // conj(co) gh
        acc[g].y += co.x * gh.y - co.y * gh.x;
      }
    }
  }
  float2* out = partial + ((size_t)blockIdx.y * K + k) * G;
#pragma unroll
  for (int g = 0; g < SOS_MAX_G; ++g)
    if (g < G) out[g] = acc[g];
}

// partial[r][chunk][s][6]: with t = conj(gH[b][k]) T[k][g] Co,  d/db_j = Re(t zi^j / num_s),  d/da_j = -Re(t zi^j / den_s)
__global__ __launch_bounds__(SOS_T) void k_sos_compose_bwd_c(const float* __restrict__ coef, int G, int S,
                                                             const double2* __restrict__ z, int K,
                                                             const float2* __restrict__ T,
                                                             const float2* __restrict__ gH,
                                                             float* __restrict__ partial) {
  __shared__ float s_c[SOS_MAX_S * 6];
  __shared__ float s_red[4][SOS_MAX_S * 6];
  const int r = blockIdx.y, b = r / G, g = r - b * G;
  sos_stage(coef + (size_t)r * S * 6, S * 6, s_c);
  // pass 1: t = conj(gH) T Co and z^-1 for the thread's SOS_KPT bins (registers); pass 2: one section at a time
  // over those bins -- six running sums instead of 6 S, and the section loop stays rolled
  const int k0 = blockIdx.x * SOS_T * SOS_KPT;
  float2 tt[SOS_KPT];
  double2 zz[SOS_KPT];
#pragma unroll
  for (int it = 0; it < SOS_KPT; ++it) {
    const int k = k0 + it * SOS_T + threadIdx.x;
    tt[it] = make_float2(0.f, 0.f);
    zz[it] = make_double2(1.0, 0.0);
    if (k < K) {
      const double2 zi = dinv(z[k]);
      const float2 h = sos_cascade(s_c, S, zi, dmul(zi, zi));
      const float2 gh = gH[(size_t)b * K + k], tg = T[(size_t)k * G + g];
      tt[it] = cmul(make_float2(gh.x * tg.x + gh.y * tg.y, gh.x * tg.y - gh.y * tg.x), h);     // conj(gh) tg Co
      zz[it] = zi;
    }
  }
  const int wv = threadIdx.x >> 6;
#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f;
#pragma unroll
    for (int it = 0; it < SOS_KPT; ++it) {
      const double2 zi = zz[it], zi2 = dmul(zi, zi);
      double2 num, den;              // sections in float64 (they cancel at low frequencies), quotients in float32
      sos_section(s_c + 6 * s, zi, zi2, num, den);
      const float2 nf = make_float2((float)num.x, (float)num.y), df = make_float2((float)den.x, (float)den.y);
      const float2 zf = make_float2((float)zi.x, (float)zi.y), zf2 = make_float2((float)zi2.x, (float)zi2.y);
      const float2 u = cmul(tt[it], cinv(nf)), v = cmul(tt[it], cinv(df));
      a0 += u.x;
      a1 += u.x * zf.x - u.y * zf.y;
      a2 += u.x * zf2.x - u.y * zf2.y;
      a3 -= v.x;
      a4 -= v.x * zf.x - v.y * zf.y;
      a5 -= v.x * zf2.x - v.y * zf2.y;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    a3 = wave_sum(a3); a4 = wave_sum(a4); a5 = wave_sum(a5);
    if ((threadIdx.x & 63) == 0) {
      float* d = &s_red[wv][6 * s];
      d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3; d[4] = a4; d[5] = a5;
    }
  }
  __syncthreads();
  float* out = partial + ((size_t)r * gridDim.x + blockIdx.x) * S * 6;
  for (int e = threadIdx.x; e < S * 6; e += SOS_T) out[e] = s_red[0][e] + s_red[1][e] + s_red[2][e] + s_red[3][e];
}

static int sos_args_ok(const float* coef, int R, int S, const double* z, int K) {
  if (!coef || !z || R <= 0 || S <= 0 || K <= 0) return GFDN_E_BADARG;
  if (S > SOS_MAX_S) return GFDN_E_UNSUPPORTED;
  return 0;
}

extern "C" int gfdn_sos_response(const float* coef, int R, int S, const double* z_c128, int K, float* out_c64,
                                 void* stream) {
  int rc = sos_args_ok(coef, R, S, z_c128, K);
  if (rc) return rc;
  if (!out_c64) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_sos_response, dim3((K + SOS_T - 1) / SOS_T, R), dim3(SOS_T), 0, (hipStream_t)stream, coef, S,
                     (const double2*)z_c128, K, (float2*)out_c64);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_sos_compose_fwd(const float* coef, int B, int G, int S, const double* z_c128, int K,
                                    const float* T_c64, const float* direct_c64, int ldd, float* H_c64,
                                    void* stream) {
  int rc = sos_args_ok(coef, B, S, z_c128, K);
  if (rc) return rc;
  if (!T_c64 || !H_c64 || G <= 0 || (direct_c64 && ldd < K)) return GFDN_E_BADARG;
  if (G > SOS_MAX_G) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_sos_compose_fwd, dim3((K + SOS_T - 1) / SOS_T, B), dim3(SOS_T), 0, (hipStream_t)stream, coef,
                     G, S, (const double2*)z_c128, K, (const float2*)T_c64, (const float2*)direct_c64, ldd,
                     (float2*)H_c64);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_sos_compose_bwd_chunks(int B, int K, int* t_chunks, int* c_chunks) {
  if (B <= 0 || K <= 0 || !t_chunks || !c_chunks) return GFDN_E_BADARG;
  *t_chunks = (B + SOS_BCH - 1) / SOS_BCH;
  *c_chunks = (K + SOS_T * SOS_KPT - 1) / (SOS_T * SOS_KPT);
  return 0;
}

extern "C" int gfdn_sos_compose_bwd(const float* coef, int B, int G, int S, const double* z_c128, int K,
                                    const float* T_c64, const float* gH_c64, float* gT_partial_c64,
                                    float* gcoef_partial, void* stream) {
  int rc = sos_args_ok(coef, B, S, z_c128, K);
  if (rc) return rc;
  if (!T_c64 || !gH_c64 || !gT_partial_c64 || !gcoef_partial || G <= 0) return GFDN_E_BADARG;
  if (G > SOS_MAX_G) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sos_compose_bwd_t, dim3((K + SOS_T - 1) / SOS_T, (B + SOS_BCH - 1) / SOS_BCH), dim3(SOS_T), 0, s,
                     coef, B, G, S, (const double2*)z_c128, K, (const float2*)gH_c64, (float2*)gT_partial_c64);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sos_compose_bwd_c, dim3((K + SOS_T * SOS_KPT - 1) / (SOS_T * SOS_KPT), B * G), dim3(SOS_T), 0, s,
                     coef, G, S, (const double2*)z_c128, K, (const float2*)T_c64, (const float2*)gH_c64,
                     gcoef_partial);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// SVF parameters -> biquad coefficients (gain_filters.py:327-330 scaled sigmoids, :36-103 SVF mixing coefficients,
// :117-151 BiquadCascade.from_svf_coeffs), one thread per (cascade, section); the torch expression of the same map
// is ~40 elementwise launches forward and ~80 backward on (B, G, 11) tensors.
//   R = 1e-6 + (1 - 1e-6) sigmoid(r0),  Gn = 10^((-6 + 12 sigmoid(r1)) / 20)
//   section 0: low shelf (m_lp = Gn, m_bp = 2 R sqrt Gn), last: high shelf (m_hp = Gn, m_bp = 2 R sqrt Gn),
//   others: peaking (m_bp = 2 R Gn);  b = [f^2 m_lp + f m_bp + m_hp, (2 f^2 m_lp - 2 m_hp) p, (f^2 m_lp - f m_bp + m_hp) p^2],
//   a = [f^2 + 2 R f + 1, (2 f^2 - 2) p, (f^2 - 2 R f + 1) p^2]   (f: normalised cut-off, p: pole compression factor)
// evaluated in float64 like the reference (float64 cut-offs times float32 parameters) and stored as float32.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void svf_terms(const float* raw, double f, double p, int s, int S, double (&c)[6],
                                          double (&dR)[6], double (&dG)[6], double& R_r0, double& G_r1) {
  const double e0 = 1.0 / (1.0 + exp(-(double)raw[0])), e1 = 1.0 / (1.0 + exp(-(double)raw[1]));
  // the reference forms R and G in float32 (sigmoid outputs of the float32 network), then promotes
  const double R = (double)(float)(1e-6 + (1.0 - 1e-6) * e0);
  const double Gn = (double)(float)pow(10.0, (-6.0 + 12.0 * e1) * 0.05);
  R_r0 = (1.0 - 1e-6) * e0 * (1.0 - e0);
  G_r1 = Gn * 2.302585092994046 * 0.05 * 12.0 * e1 * (1.0 - e1);
  const bool low = s == 0, high = s == S - 1;
  const double sq = sqrt(Gn);
  const double m_lp = low ? Gn : 1.0, m_hp = high ? Gn : 1.0;
  const double m_bp = (low || high) ? 2.0 * R * sq : 2.0 * R * Gn;
  const double dlp_dG = low ? 1.0 : 0.0, dhp_dG = high ? 1.0 : 0.0;
  const double dbp_dR = (low || high) ? 2.0 * sq : 2.0 * Gn;
  const double dbp_dG = (low || high) ? R / sq : 2.0 * R;
  const double f2 = f * f, p2 = p * p;
  c[0] = f2 * m_lp + f * m_bp + m_hp;
  c[1] = (2.0 * f2 * m_lp - 2.0 * m_hp) * p;
  c[2] = (f2 * m_lp - f * m_bp + m_hp) * p2;
  c[3] = f2 + 2.0 * R * f + 1.0;
  c[4] = (2.0 * f2 - 2.0) * p;
  c[5] = (f2 - 2.0 * R * f + 1.0) * p2;
  dR[0] = f * dbp_dR;  dR[1] = 0.0;  dR[2] = -f * dbp_dR * p2;  dR[3] = 2.0 * f;  dR[4] = 0.0;  dR[5] = -2.0 * f * p2;
  dG[0] = f2 * dlp_dG + f * dbp_dG + dhp_dG;
  dG[1] = (2.0 * f2 * dlp_dG - 2.0 * dhp_dG) * p;
  dG[2] = (f2 * dlp_dG - f * dbp_dG + dhp_dG) * p2;
  dG[3] = dG[4] = dG[5] = 0.0;
}

// raw (R, S, 2) -> coef (R, S, 6);  with gcoef: graw (R, S, 2) instead
__global__ __launch_bounds__(256) void k_svf_coef(const float* __restrict__ raw, const double* __restrict__ cutoff,
                                                  double cpf, int R, int S, const float* __restrict__ gcoef,
                                                  float* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= R * S) return;
  const int s = e % S;
  double c[6], dR[6], dG[6], R_r0, G_r1;
  svf_terms(raw + 2 * (size_t)e, cutoff[s], cpf, s, S, c, dR, dG, R_r0, G_r1);
  if (!gcoef) {
#pragma unroll
    for (int j = 0; j < 6; ++j) out[6 * (size_t)e + j] = (float)c[j];
  } else {
    double gR = 0.0, gG = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double g = (double)gcoef[6 * (size_t)e + j];
      gR += g * dR[j];
      gG += g * dG[j];
    }
    out[2 * (size_t)e] = (float)(gR * R_r0);
    out[2 * (size_t)e + 1] = (float)(gG * G_r1);
  }
}

extern "C" int gfdn_svf_coefficients(const float* raw, const double* cutoff, double compress_pole_factor, int R, int S,
                                     const float* gcoef, float* out, void* stream) {
  if (!raw || !cutoff || !out || R <= 0 || S <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_svf_coef, dim3((R * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, cutoff,
                     compress_pole_factor, R, S, gcoef, out);
  GFDN_LAUNCH_CHECK();
  return 0;
}
