// The receivers' time signals formed where they are read (the time-domain output stage, linear.hip):
//     x[b][t] = xd[rows[b]][t] + sum_g rgain[b][g] tau[band(b) G + g][t]
// for the kernels that walk pair-interleaved signals (items 2p, 2p + 1 as the .x / .y of a float2): a consumer handed an
// XLin (xd != NULL) reads the dataset's transformed direct paths (two 4-byte streams) and the band's group signals
// (pair-interleaved, shared by all receivers of the band: cache hits) instead of a stored x2 -- which then need not exist.
#pragma once
#include "common.h"
#include "scan_dev.h"

struct XLin {
  const float* xd;             // (R, ld_xd) float; NULL: the consumer reads its stored x2
  int ld_xd;
  const long long* rows;       // item -> row of xd (NULL: identity)
  const float2* tau2;          // (ceil(S / 2), ld_tau) float2: group signals, pair-interleaved
  int ld_tau;
  const float* rgain;          // (items, G)
  int B, G;                    // receivers per band, groups per band (G <= 4)
};

struct XPair {
  const float* d1;
  const float* d2;
  const float2* tq;            // quad path: the band's first signal pair
  const float* tf;
  int ld_tau, s1, s2, G;
  float rg1[4], rg2[4];
  bool two, quad;
};

__device__ __forceinline__ XPair xpair_init(const XLin& L, int p, int items) {
  XPair P;
  const int b1 = 2 * p, b2 = b1 + 1;
  P.two = b2 < items;
  P.G = L.G;
  const int band1 = b1 / L.B, band2 = P.two ? b2 / L.B : band1;
  P.d1 = L.xd + (size_t)(L.rows ? L.rows[b1] : b1) * L.ld_xd;
  P.d2 = P.two ? L.xd + (size_t)(L.rows ? L.rows[b2] : b2) * L.ld_xd : P.d1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    P.rg1[g] = g < L.G ? L.rgain[(size_t)b1 * L.G + g] : 0.f;
    P.rg2[g] = (g < L.G && P.two) ? L.rgain[(size_t)b2 * L.G + g] : 0.f;
  }
  P.s1 = band1 * L.G;
  P.s2 = band2 * L.G;
  // both items in one band whose signals fill whole pairs (the band bank: B even, G even): 8-byte loads of two signals
  P.quad = band1 == band2 && !(P.s1 & 1) && !(L.G & 1);
  P.ld_tau = L.ld_tau;
  P.tq = L.tau2 + (size_t)(P.s1 >> 1) * L.ld_tau;
  P.tf = (const float*)L.tau2;
  return P;
}

__device__ __forceinline__ float2 xpair_compose(const XPair& P, float2 v, float2 t01, float2 t23) {
  v.x += P.rg1[0] * t01.x;
  v.x += P.rg1[1] * t01.y;
  v.x += P.rg1[2] * t23.x;
  v.x += P.rg1[3] * t23.y;
  v.y += P.rg2[0] * t01.x;
  v.y += P.rg2[1] * t01.y;
  v.y += P.rg2[2] * t23.x;
  v.y += P.rg2[3] * t23.y;
  return v;
}

__device__ __forceinline__ float2 xpair_load1(const XPair& P, int t) {
  float2 v = make_float2(P.d1[t], P.two ? P.d2[t] : 0.f);
  if (P.quad) {
    const float2 t01 = P.tq[t], t23 = P.G > 2 ? P.tq[P.ld_tau + t] : make_float2(0.f, 0.f);
    return xpair_compose(P, v, t01, t23);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < P.G) {
      const int s1 = P.s1 + g, s2 = P.s2 + g;
      v.x += P.rg1[g] * P.tf[((size_t)(s1 >> 1) * P.ld_tau + t) * 2 + (s1 & 1)];
      if (P.two) v.y += P.rg2[g] * P.tf[((size_t)(s2 >> 1) * P.ld_tau + t) * 2 + (s2 & 1)];
    }
  }
  return v;
}

// four consecutive samples t .. t + 3
__device__ __forceinline__ void xpair_load4(const XPair& P, int t, float2 (&o)[4]) {
  if (P.quad) {
    float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
    float2 t01[4], t23[4];
    ld4_f(P.d1 + t, a);
    if (P.two) ld4_f(P.d2 + t, b);
    ld4_f2(P.tq + t, t01);
    if (P.G > 2) {
      ld4_f2(P.tq + P.ld_tau + t, t23);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) t23[u] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = xpair_compose(P, make_float2(a[u], b[u]), t01[u], t23[u]);
    return;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = xpair_load1(P, t + u);
}
