// The output stage in the TIME domain, for gfx950.
//
// Reference maths (orchidas/DiffGFDN): H[b] = (sum_g gain[b][g] T_g + d[b]) * filt  (src/diff_gfdn/model.py:583-619 with
// the sub-band filter of trainer.py:459), then x[b] = irfft(H[b], n = K) inside both decay losses (losses.py:207-213,
// :442-445).  The inverse transform is LINEAR and neither the group transfer functions T_g nor the filter depend on the
// receiver, so
//        x[b] = irfft(d[b] filt) + sum_g gain[b][g] irfft(T_g filt) = xd[row_b] + sum_g gain[b][g] tau_g :
// xd is a constant of the dataset (the early response through the band's filter, transformed ONCE per receiver like the
// decay targets), and a step transforms G signals per band instead of one per receiver (4 instead of 32 at the
// north-star size).  The adjoint follows the same way:
//        dL/dgain[b][g] = <dL/dx[b], tau_g>               (a dot product over the time samples)
//        dL/d(T_g filt) = irfft^T( sum_b gain[b][g] dL/dx[b] )   (G adjoint transforms per band)
// so the (items, K) spectra H and dL/dH never exist and both transforms of the step shrink by B / G.
//
// Layouts.  "pairs": two signals interleaved sample by sample, (ceil(S / 2), ld, 2) float -- what the pair transforms
// (gfdn_irfft_odd_pairs_fwd / _bwd) produce and take, and what the pair STFT / EDC kernels read; "plain": (S, ld) float.
#include "common.h"
#include "scan_dev.h"

#define LIN_MAXG 4
#define LIN_V 4              // consecutive time samples per thread
#define LIN_R 8              // receivers per workgroup of the dot-product pass

// tau_s[t .. t + 3] for the four signals s0 .. s0 + G - 1 of a band (zeros for g >= G)
template <bool TAU_PAIRS>
__device__ __forceinline__ void lin_load_tau4(const float* __restrict__ tau, int ld_tau, int s0, int G, int t,
                                              float (&tv)[LIN_MAXG][LIN_V]) {
#pragma unroll
  for (int g = 0; g < LIN_MAXG; ++g) {
    if (g < G) {
      const int s = s0 + g;
      if (TAU_PAIRS) {
        float2 q[4];
        ld4_f2((const float2*)tau + (size_t)(s >> 1) * ld_tau + t, q);
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) tv[g][u] = (s & 1) ? q[u].y : q[u].x;
      } else {
        ld4_f(tau + (size_t)s * ld_tau + t, tv[g]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) tv[g][u] = 0.f;
    }
  }
}
template <bool TAU_PAIRS>
__device__ __forceinline__ float lin_tau1(const float* __restrict__ tau, int ld_tau, int s, int t) {
  return TAU_PAIRS ? tau[((size_t)(s >> 1) * ld_tau + t) * 2 + (s & 1)] : tau[(size_t)s * ld_tau + t];
}

// x[b][t] = xd[rows[b]][t] + sum_g rgain[b][g] tau[band(b) G + g][t]; one workgroup row = one PAIR of items (2p, 2p + 1)
template <bool OUT_PAIRS, bool TAU_PAIRS>
__global__ __launch_bounds__(256) void k_lin_combine_fwd(const float* __restrict__ xd, int ld_xd,
                                                         const long long* __restrict__ rows,
                                                         const float* __restrict__ tau, int ld_tau,
                                                         const float* __restrict__ rgain, int B, int G, int items, int n,
                                                         float* __restrict__ x, int ld_x) {
  const int p = blockIdx.y, b1 = 2 * p, b2 = b1 + 1;
  const bool two = b2 < items;
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * LIN_V;
  if (t0 >= n) return;
  const int band1 = b1 / B, band2 = two ? b2 / B : band1;
  const float* d1 = xd + (size_t)(rows ? rows[b1] : b1) * ld_xd;
  const float* d2 = two ? xd + (size_t)(rows ? rows[b2] : b2) * ld_xd : d1;
  float rg1[LIN_MAXG], rg2[LIN_MAXG];
#pragma unroll
  for (int g = 0; g < LIN_MAXG; ++g) {
    rg1[g] = g < G ? rgain[(size_t)b1 * G + g] : 0.f;
    rg2[g] = (g < G && two) ? rgain[(size_t)b2 * G + g] : 0.f;
  }
  float o1[LIN_V], o2[LIN_V];
  if (t0 + LIN_V <= n) {
    float tv[LIN_MAXG][LIN_V];
    ld4_f(d1 + t0, o1);
    if (two) ld4_f(d2 + t0, o2);
    lin_load_tau4<TAU_PAIRS>(tau, ld_tau, band1 * G, G, t0, tv);
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) o1[u] += rg1[g] * tv[g][u];
    if (two) {
      if (band2 != band1) lin_load_tau4<TAU_PAIRS>(tau, ld_tau, band2 * G, G, t0, tv);
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) o2[u] += rg2[g] * tv[g][u];
    } else {
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) o2[u] = 0.f;
    }
    if (OUT_PAIRS) {
      float2 q[4];
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) q[u] = make_float2(o1[u], o2[u]);
      st4_f2((float2*)x + (size_t)p * ld_x + t0, q);
    } else {
      st4_f(x + (size_t)b1 * ld_x + t0, o1);
      if (two) st4_f(x + (size_t)b2 * ld_x + t0, o2);
    }
    return;
  }
  for (int t = t0; t < n; ++t) {                 // the last, partial group of a row
    float a1 = d1[t], a2 = two ? d2[t] : 0.f;
    for (int g = 0; g < G; ++g) {
      a1 += rg1[g] * lin_tau1<TAU_PAIRS>(tau, ld_tau, band1 * G + g, t);
      if (two) a2 += rg2[g] * lin_tau1<TAU_PAIRS>(tau, ld_tau, band2 * G + g, t);
    }
    if (OUT_PAIRS) {
      ((float2*)x)[(size_t)p * ld_x + t] = make_float2(a1, a2);
    } else {
      x[(size_t)b1 * ld_x + t] = a1;
      if (two) x[(size_t)b2 * ld_x + t] = a2;
    }
  }
}

// gamma[band G + g][t] = sum_{b in band} rgain[b][g] (gx[b][t] [+ gxb[b][t]]): one workgroup = one band x 1024 samples,
// the receivers of the band summed in index order (fixed order: bitwise reproducible)
template <bool IN_PAIRS, bool OUT_PAIRS>
__global__ __launch_bounds__(256) void k_lin_gamma(const float* __restrict__ gx, const float* __restrict__ gxb, int ld_g,
                                                   const float* __restrict__ rgain, int B, int G, int n,
                                                   float* __restrict__ gamma, int ld_o,
                                                   const int* __restrict__ slot_of_time,
                                                   const float* __restrict__ base, int ld_b) {
  __shared__ float s_rg[256];                     // the band's gains (B G <= 256)
  const int band = blockIdx.y;
  for (int i = threadIdx.x; i < B * G; i += 256) s_rg[i] = rgain[(size_t)band * B * G + i];
  __syncthreads();
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * LIN_V;
  if (t0 >= n) return;
  const bool full = t0 + LIN_V <= n;
  float acc[LIN_MAXG][LIN_V];
#pragma unroll
  for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
    for (int u = 0; u < LIN_V; ++u) acc[g][u] = 0.f;
  const int i0 = band * B;
  if (IN_PAIRS) {
    // (B even: the band's items are the pairs i0 / 2 .. i0 / 2 + B / 2 - 1)
    const float2* g2 = (const float2*)gx + (size_t)(i0 >> 1) * ld_g;
    for (int pp = 0; pp < B / 2; ++pp) {
      float2 q[4];
      if (full) {
        ld4_f2(g2 + (size_t)pp * ld_g + t0, q);
      } else {
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) q[u] = t0 + u < n ? g2[(size_t)pp * ld_g + t0 + u] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g) {
        if (g < G) {
          const float ra = s_rg[(2 * pp) * G + g], rb = s_rg[(2 * pp + 1) * G + g];
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) {
            acc[g][u] += ra * q[u].x;
            acc[g][u] += rb * q[u].y;
          }
        }
      }
    }
  } else {
    for (int b = 0; b < B; ++b) {
      float v[LIN_V];
      const float* r1 = gx + (size_t)(i0 + b) * ld_g + t0;
      const float* r2 = gxb ? gxb + (size_t)(i0 + b) * ld_g + t0 : nullptr;
      if (full) {
        ld4_f(r1, v);
        if (r2) {
          float w[LIN_V];
          ld4_f(r2, w);
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) v[u] += w[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) v[u] = t0 + u < n ? r1[u] + (r2 ? r2[u] : 0.f) : 0.f;
      }
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g) {
        if (g < G) {
          const float ra = s_rg[b * G + g];
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) acc[g][u] += ra * v[u];
        }
      }
    }
  }
  if (OUT_PAIRS && base) {
    // another part of the same gradient signals, already summed per group (pair-interleaved, time order): the EDR part,
    // which the adjoint STFT of the G gradient spectra per band left (edrlin.hip)
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g) {
      if (g < G) {
        const int s = band * G + g;
#pragma unroll
        for (int u = 0; u < LIN_V; ++u)
          if (t0 + u < n) acc[g][u] += base[((size_t)(s >> 1) * ld_b + t0 + u) * 2 + (s & 1)];
      }
    }
  }
  if (OUT_PAIRS && slot_of_time) {
    // the adjoint pair transform's own order (gfdn_irfft_odd_pairs_bwd_tslots): sample 0 first, then the sample of time t
    // at 1 + slot_of_time[t] -- the scatter happens here, on G signals per band, instead of a gather in the transform
#pragma unroll
    for (int u = 0; u < LIN_V; ++u) {
      const int t = t0 + u;
      if (t < n) {
        const size_t pos = t == 0 ? 0 : 1 + (size_t)slot_of_time[t];
        if (!((band * G) & 1) && !(G & 1)) {          // whole output pairs: 8-byte stores
#pragma unroll
          for (int g = 0; g < LIN_MAXG; g += 2)
            if (g < G)
              ((float2*)gamma)[(size_t)((band * G + g) >> 1) * ld_o + pos] = make_float2(acc[g][u], acc[g + 1][u]);
        } else {
#pragma unroll
          for (int g = 0; g < LIN_MAXG; ++g) {
            if (g < G) {
              const int s = band * G + g;
              gamma[((size_t)(s >> 1) * ld_o + pos) * 2 + (s & 1)] = acc[g][u];
            }
          }
        }
      }
    }
    return;
  }
  if (OUT_PAIRS && full && !((band * G) & 1) && !(G & 1)) {
    // the band's signals fill whole output pairs: 32-byte stores of (gamma_g, gamma_g+1) for four samples
#pragma unroll
    for (int g = 0; g < LIN_MAXG; g += 2) {
      if (g < G) {
        float2 q[4];
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) q[u] = make_float2(acc[g][u], acc[g + 1][u]);
        st4_f2((float2*)gamma + (size_t)((band * G + g) >> 1) * ld_o + t0, q);
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < LIN_MAXG; ++g) {
    if (g < G) {
      const int s = band * G + g;
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) {
        if (t0 + u < n) {
          if (OUT_PAIRS) gamma[((size_t)(s >> 1) * ld_o + t0 + u) * 2 + (s & 1)] = acc[g][u];
          else gamma[(size_t)s * ld_o + t0 + u] = acc[g][u];
        }
      }
    }
  }
}

// part[((band B + b) G + g) nchunk + chunk] = sum over the chunk's samples of (gx[b][t] [+ gxb[b][t]]) tau[band G + g][t]:
// LIN_R receivers x a stripe of samples per workgroup, LIN_R x G per-thread sums, one block reduction at the end
template <bool IN_PAIRS, bool TAU_PAIRS>
__global__ __launch_bounds__(256) void k_lin_gain_dots(const float* __restrict__ gx, const float* __restrict__ gxb,
                                                       int ld_g, const float* __restrict__ tau, int ld_tau, int B, int G,
                                                       int n, float* __restrict__ part, int ngrp, int ld_part) {
  __shared__ float s_red[4][LIN_R * LIN_MAXG];
  const int band = blockIdx.y / ngrp, b0 = (blockIdx.y - band * ngrp) * LIN_R;
  const int nr = B - b0 < LIN_R ? B - b0 : LIN_R;
  const int i0 = band * B + b0;
  float acc[LIN_R][LIN_MAXG];
#pragma unroll
  for (int r = 0; r < LIN_R; ++r)
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g) acc[r][g] = 0.f;
  const int ngroups = (n + LIN_V - 1) / LIN_V;
  for (int grp = blockIdx.x * 256 + threadIdx.x; grp < ngroups; grp += gridDim.x * 256) {
    const int t0 = grp * LIN_V;
    const bool full = t0 + LIN_V <= n;
    float tv[LIN_MAXG][LIN_V];
    if (full) {
      lin_load_tau4<TAU_PAIRS>(tau, ld_tau, band * G, G, t0, tv);
    } else {
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
        for (int u = 0; u < LIN_V; ++u)
          tv[g][u] = (g < G && t0 + u < n) ? lin_tau1<TAU_PAIRS>(tau, ld_tau, band * G + g, t0 + u) : 0.f;
    }
    if (IN_PAIRS) {
      // (B and LIN_R even: the workgroup's receivers are the pairs i0 / 2 .. )
      const float2* g2 = (const float2*)gx + (size_t)(i0 >> 1) * ld_g;
#pragma unroll
      for (int pp = 0; pp < LIN_R / 2; ++pp) {
        float2 q[4];
        if (2 * pp < nr) {
          if (full) {
            ld4_f2(g2 + (size_t)pp * ld_g + t0, q);
          } else {
#pragma unroll
            for (int u = 0; u < LIN_V; ++u) q[u] = t0 + u < n ? g2[(size_t)pp * ld_g + t0 + u] : make_float2(0.f, 0.f);
          }
        } else {
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) q[u] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) {
            acc[2 * pp][g] += q[u].x * tv[g][u];
            acc[2 * pp + 1][g] += q[u].y * tv[g][u];
          }
      }
    } else {
#pragma unroll
      for (int r = 0; r < LIN_R; ++r) {
        float v[LIN_V];
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) v[u] = 0.f;
        if (r < nr) {
          const float* r1 = gx + (size_t)(i0 + r) * ld_g + t0;
          const float* r2 = gxb ? gxb + (size_t)(i0 + r) * ld_g + t0 : nullptr;
          if (full) {
            ld4_f(r1, v);
            if (r2) {
              float w[LIN_V];
              ld4_f(r2, w);
#pragma unroll
              for (int u = 0; u < LIN_V; ++u) v[u] += w[u];
            }
          } else {
#pragma unroll
            for (int u = 0; u < LIN_V; ++u) v[u] = t0 + u < n ? r1[u] + (r2 ? r2[u] : 0.f) : 0.f;
          }
        }
#pragma unroll
        for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
          for (int u = 0; u < LIN_V; ++u) acc[r][g] += v[u] * tv[g][u];
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < LIN_R; ++r)
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g) {
      const float v = wave_sum_full(acc[r][g]);          // (every lane of the four waves is active here)
      if (lane == 0) s_red[wv][r * LIN_MAXG + g] = v;
    }
  __syncthreads();
  if (threadIdx.x < LIN_R * LIN_MAXG) {
    const int r = threadIdx.x / LIN_MAXG, g = threadIdx.x % LIN_MAXG;
    if (r < nr && g < G) {
      const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
      part[((size_t)(band * B + b0 + r) * G + g) * ld_part + blockIdx.x] = v;
    }
  }
}

// gfdn_lin_gamma and gfdn_lin_gain_dots as ONE sweep over the pair-interleaved gradient signals of the EDC term, which are
// nonzero on the EDC window [w0, w0 + wlen) only (samples outside it are neither read nor required to hold zeros):
//   gamma[band G + g][t] = base[band G + g][t] + sum_{b in band} rgain[b][g] gx[b][t]     (transform order: slot_of_time)
//   part[((band B + b) G + g) ld_part + tile] = sum_{t in tile} gx[b][t] tau[band G + g][t]
// one workgroup = one band x 1024 samples; the 8 dot products of a receiver pair are folded over the workgroup by VALU wave
// sums and one LDS hand-over at the end (fixed order).  wlen_band (optional, device): per-band window lengths.
__global__ __launch_bounds__(256) void k_lin_gamma_dots(const float2* __restrict__ g2, int ld_g,
                                                        const float* __restrict__ rgain, int B, int G, int n,
                                                        const float2* __restrict__ tau2, int ld_tau,
                                                        const float* __restrict__ base, int ld_b,
                                                        const int* __restrict__ slot_of_time, float* __restrict__ gamma,
                                                        int ld_o, float* __restrict__ part, int ld_part, int w0, int wlen,
                                                        const int* __restrict__ wlen_band,
                                                        const float* __restrict__ base_b) {
  __shared__ float s_rg[256];
  __shared__ float s_dot[4][256];                 // [wave][pair * 8 + item * 4 + g]  (B / 2 <= 32 pairs)
  const int band = blockIdx.y, tile = blockIdx.x;
  for (int i = threadIdx.x; i < B * G; i += 256) s_rg[i] = rgain[(size_t)band * B * G + i];
  __syncthreads();
  if (wlen_band) wlen = wlen_band[band];
  const int t0 = (tile * 256 + threadIdx.x) * LIN_V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_lo = tile * 256 * LIN_V, tile_hi = tile_lo + 256 * LIN_V;
  const bool tile_in = tile_hi > w0 && tile_lo < w0 + wlen;          // (workgroup-uniform)
  float acc[LIN_MAXG][LIN_V];
#pragma unroll
  for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
    for (int u = 0; u < LIN_V; ++u) acc[g][u] = 0.f;
  const int npairs = B / 2;
  if (tile_in) {
    bool in[LIN_V];
#pragma unroll
    for (int u = 0; u < LIN_V; ++u) in[u] = t0 + u >= w0 && t0 + u < w0 + wlen && t0 + u < n;
    const bool full = in[0] && in[LIN_V - 1];
    float tv[LIN_MAXG][LIN_V];
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g)
#pragma unroll
      for (int u = 0; u < LIN_V; ++u) {
        const int s = band * G + g;
        tv[g][u] = (g < G && in[u]) ? ((const float*)tau2)[((size_t)(s >> 1) * ld_tau + t0 + u) * 2 + (s & 1)] : 0.f;
      }
    const float2* gb = g2 + (size_t)(band * B / 2) * ld_g;
    for (int pp = 0; pp < npairs; ++pp) {
      float2 q[4];
      if (full) {
        ld4_f2(gb + (size_t)pp * ld_g + t0, q);
      } else {
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) q[u] = in[u] ? gb[(size_t)pp * ld_g + t0 + u] : make_float2(0.f, 0.f);
      }
      float d1[LIN_MAXG], d2[LIN_MAXG];
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g) {
        const float ra = g < G ? s_rg[(2 * pp) * G + g] : 0.f, rb = g < G ? s_rg[(2 * pp + 1) * G + g] : 0.f;
        d1[g] = d2[g] = 0.f;
#pragma unroll
        for (int u = 0; u < LIN_V; ++u) {
          acc[g][u] += ra * q[u].x;
          acc[g][u] += rb * q[u].y;
          d1[g] += q[u].x * tv[g][u];
          d2[g] += q[u].y * tv[g][u];
        }
      }
#pragma unroll
      for (int g = 0; g < LIN_MAXG; ++g) {
        const float a1 = wave_sum_full(d1[g]), a2 = wave_sum_full(d2[g]);
        if (lane == 0) {
          s_dot[wave][pp * 8 + g] = a1;
          s_dot[wave][pp * 8 + 4 + g] = a2;
        }
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < npairs * 8) {
    const int pp = threadIdx.x >> 3, it = (threadIdx.x >> 2) & 1, g = threadIdx.x & 3;
    if (g < G) {
      const float v = tile_in ? (s_dot[0][threadIdx.x] + s_dot[1][threadIdx.x]) + (s_dot[2][threadIdx.x] + s_dot[3][threadIdx.x])
                              : 0.f;
      part[((size_t)(band * B + 2 * pp + it) * G + g) * ld_part + tile] = v;
    }
  }
  if (t0 >= n) return;
  if (base) {
#pragma unroll
    for (int g = 0; g < LIN_MAXG; ++g) {
      if (g < G) {
        const int s = band * G + g;
#pragma unroll
        for (int u = 0; u < LIN_V; ++u)
          if (t0 + u < n) {
            const size_t o = ((size_t)(s >> 1) * ld_b + t0 + u) * 2 + (s & 1);
            acc[g][u] += base_b ? base[o] + base_b[o] : base[o];
          }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < LIN_V; ++u) {
    const int t = t0 + u;
    if (t < n) {
      const size_t pos = slot_of_time ? (t == 0 ? 0 : 1 + (size_t)slot_of_time[t]) : (size_t)t;
      if (!((band * G) & 1) && !(G & 1)) {
#pragma unroll
        for (int g = 0; g < LIN_MAXG; g += 2)
          if (g < G)
            ((float2*)gamma)[(size_t)((band * G + g) >> 1) * ld_o + pos] = make_float2(acc[g][u], acc[g + 1][u]);
      } else {
#pragma unroll
        for (int g = 0; g < LIN_MAXG; ++g) {
          if (g < G) {
            const int s = band * G + g;
            gamma[((size_t)(s >> 1) * ld_o + pos) * 2 + (s & 1)] = acc[g][u];
          }
        }
      }
    }
  }
}

extern "C" int gfdn_lin_gamma_dots_tiles(int n) { return n > 0 ? (n + 256 * LIN_V - 1) / (256 * LIN_V) : 0; }

// all signals pair-interleaved: gx2 (nbands B / 2, ld_g), tau2 / base2 / gamma (ceil(nbands G / 2), .); B even, B <= 64;
// part rows of pitch ld_part >= gfdn_lin_gamma_dots_tiles(n).
extern "C" int gfdn_lin_gamma_dots(const float* gx2, int ld_g, const float* rgain, int nbands, int B, int G, int n,
                                   const float* tau2, int ld_tau, const float* base2, int ld_b, const int* slot_of_time,
                                   float* gamma, int ld_o, float* part, int ld_part, int win_start, int win_len,
                                   const int* band_win_len, const float* base2b, void* stream) {
  if (!gx2 || !rgain || !tau2 || !gamma || !part || nbands <= 0 || B <= 0 || G <= 0 || n <= 0 || ld_g < n || ld_tau < n ||
      ld_o < n || (base2 && (ld_b < n || base2 == gamma)) || win_start < 0 || win_len <= 0 || win_start + win_len > n ||
      (base2b && (!base2 || base2b == gamma)))
    return GFDN_E_BADARG;
  const int tiles = (n + 256 * LIN_V - 1) / (256 * LIN_V);
  if (ld_part < tiles) return GFDN_E_BADARG;
  if (G > LIN_MAXG || (B & 1) || B > 64 || B * G > 256 || nbands > 65535) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_lin_gamma_dots, dim3(tiles, nbands), dim3(256), 0, (hipStream_t)stream, (const float2*)gx2, ld_g, rgain,
                     B, G, n, (const float2*)tau2, ld_tau, base2, ld_b, slot_of_time, gamma, ld_o, part, ld_part, win_start,
                     win_len, band_win_len, base2b);
  GFDN_LAUNCH_CHECK();
  return 0;
}

static int lin_chunks_host(int n) {
  int c = (n + 256 * LIN_V * 2 - 1) / (256 * LIN_V * 2);          // ~2 groups of LIN_V samples per thread
  if (c > 64) c = 64;
  return c < 1 ? 1 : c;
}
extern "C" int gfdn_lin_gain_chunks(int n) { return n > 0 ? lin_chunks_host(n) : 0; }

extern "C" int gfdn_lin_combine_fwd(const float* xd, int ld_xd, const long long* rows, const float* tau, int ld_tau,
                                    int tau_pairs, const float* rgain, int nbands, int B, int G, int n, float* x, int ld_x,
                                    int out_pairs, void* stream) {
  if (!xd || !tau || !rgain || !x || nbands <= 0 || B <= 0 || G <= 0 || n <= 0 || ld_xd < n || ld_tau < n || ld_x < n)
    return GFDN_E_BADARG;
  if (G > LIN_MAXG) return GFDN_E_UNSUPPORTED;
  const int items = nbands * B;
  dim3 grid((n + 256 * LIN_V - 1) / (256 * LIN_V), (items + 1) / 2), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LIN_FWD(OP, TP) \
  hipLaunchKernelGGL((k_lin_combine_fwd<OP, TP>), grid, block, 0, s, xd, ld_xd, rows, tau, ld_tau, rgain, B, G, items, n, x, ld_x)
  if (out_pairs) { if (tau_pairs) LIN_FWD(true, true); else LIN_FWD(true, false); }
  else { if (tau_pairs) LIN_FWD(false, true); else LIN_FWD(false, false); }
#undef LIN_FWD
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_lin_gamma(const float* gx, const float* gxb, int ld_g, int in_pairs, const float* rgain, int nbands,
                              int B, int G, int n, float* gamma, int ld_o, int out_pairs, const int* slot_of_time,
                              const float* base2, int ld_b, void* stream) {
  if (!gx || !rgain || !gamma || nbands <= 0 || B <= 0 || G <= 0 || n <= 0 || ld_g < n || ld_o < n) return GFDN_E_BADARG;
  if ((slot_of_time || base2) && !out_pairs) return GFDN_E_BADARG;
  if (base2 && (ld_b < n || base2 == gamma)) return GFDN_E_BADARG;
  if (G > LIN_MAXG || B * G > 256 || nbands > 65535) return GFDN_E_UNSUPPORTED;
  if (in_pairs && ((B & 1) || gxb)) return GFDN_E_BADARG;        // (pairs never straddle bands; one merged gradient)
  dim3 grid((n + 256 * LIN_V - 1) / (256 * LIN_V), nbands), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LIN_GAM(IP, OP) \
  hipLaunchKernelGGL((k_lin_gamma<IP, OP>), grid, block, 0, s, gx, gxb, ld_g, rgain, B, G, n, gamma, ld_o, slot_of_time, \
                     base2, ld_b)
  if (in_pairs) { if (out_pairs) LIN_GAM(true, true); else LIN_GAM(true, false); }
  else { if (out_pairs) LIN_GAM(false, true); else LIN_GAM(false, false); }
#undef LIN_GAM
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_lin_gain_dots(const float* gx, const float* gxb, int ld_g, int in_pairs, const float* tau, int ld_tau,
                                  int tau_pairs, int nbands, int B, int G, int n, float* part, int ld_part, void* stream) {
  if (!gx || !tau || !part || nbands <= 0 || B <= 0 || G <= 0 || n <= 0 || ld_g < n || ld_tau < n ||
      ld_part < lin_chunks_host(n))
    return GFDN_E_BADARG;
  const int ngrp = (B + LIN_R - 1) / LIN_R;
  if (G > LIN_MAXG || nbands * ngrp > 65535) return GFDN_E_UNSUPPORTED;
  if (in_pairs && ((B & 1) || gxb)) return GFDN_E_BADARG;
  dim3 grid(lin_chunks_host(n), nbands * ngrp), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LIN_DOT(IP, TP) \
  hipLaunchKernelGGL((k_lin_gain_dots<IP, TP>), grid, block, 0, s, gx, gxb, ld_g, tau, ld_tau, B, G, n, part, ngrp, ld_part)
  if (in_pairs) { if (tau_pairs) LIN_DOT(true, true); else LIN_DOT(true, false); }
  else { if (tau_pairs) LIN_DOT(false, true); else LIN_DOT(false, false); }
#undef LIN_DOT
  GFDN_LAUNCH_CHECK();
  return 0;
}
