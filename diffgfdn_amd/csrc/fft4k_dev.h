// Register-resident 4096-point FFT shared by the STFT kernels (fft.hip) and the fused decay-loss kernel (decay.hip).
#pragma once
#include "common.h"

// 16-point DFT, natural order in and out; sgn = +1 forward (e^-), -1 inverse
__device__ __forceinline__ void bfly16(float2 (&a)[16], float sgn) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
  float2 b[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {       // 4-point DFTs over n2 (elements j, j+4, j+8, j+12) -> index q
    const float2 x0 = a[j], x1 = a[j + 4], x2 = a[j + 8], x3 = a[j + 12];
    const float2 p = cadd(x0, x2), m = csub(x0, x2), q = cadd(x1, x3), t = csub(x1, x3);
    const float2 jt = make_float2(sgn * t.y, -sgn * t.x);       // -j t (forward)
    b[j][0] = cadd(p, q);
    b[j][1] = cadd(m, jt);
    b[j][2] = csub(p, q);
    b[j][3] = csub(m, jt);
  }
  // twiddles W16^(j q)
  auto tw = [&](float2 v, float cr, float ci) {   // v * (cr - i sgn ci)
    return make_float2(v.x * cr + sgn * v.y * ci, v.y * cr - sgn * v.x * ci);
  };
  b[1][1] = tw(b[1][1], c1, s1);   b[1][2] = tw(b[1][2], r, r);     b[1][3] = tw(b[1][3], s1, c1);
  b[2][1] = tw(b[2][1], r, r);     b[2][2] = tw(b[2][2], 0.f, 1.f); b[2][3] = tw(b[2][3], -r, r);
  b[3][1] = tw(b[3][1], s1, c1);   b[3][2] = tw(b[3][2], -r, r);    b[3][3] = tw(b[3][3], -c1, -s1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {       // 4-point DFTs over j -> X[q + 4 s]
    const float2 x0 = b[0][q], x1 = b[1][q], x2 = b[2][q], x3 = b[3][q];
    const float2 p = cadd(x0, x2), m = csub(x0, x2), qq = cadd(x1, x3), t = csub(x1, x3);
    const float2 jt = make_float2(sgn * t.y, -sgn * t.x);
    a[q] = cadd(p, qq);
    a[q + 4] = cadd(m, jt);
    a[q + 8] = csub(p, qq);
    a[q + 12] = csub(m, jt);
  }
}
// a[u] *= w^u, u = 1..15 (powers by squaring: every factor is at most 3 products away from w)
__device__ __forceinline__ void twiddle16(float2 (&a)[16], float2 w1) {
  const float2 w2 = cmul(w1, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
  const float2 w3 = cmul(w2, w1), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
  a[1] = cmul(a[1], w1);  a[2] = cmul(a[2], w2);  a[3] = cmul(a[3], w3);  a[4] = cmul(a[4], w4);
  a[5] = cmul(a[5], w5);  a[6] = cmul(a[6], w6);  a[7] = cmul(a[7], w7);  a[8] = cmul(a[8], w8);
  a[9] = cmul(a[9], cmul(w8, w1));   a[10] = cmul(a[10], cmul(w8, w2)); a[11] = cmul(a[11], cmul(w8, w3));
  a[12] = cmul(a[12], cmul(w8, w4)); a[13] = cmul(a[13], cmul(w8, w5)); a[14] = cmul(a[14], cmul(w8, w6));
  a[15] = cmul(a[15], cmul(w8, w7));
}

#define S4K_T 256
#define S4K_PAD(i) ((i) + ((i) >> 4))
#define S4K_LDS (4096 + 256)

// in: a[k] = x[i + 256 k]; out: a[u] = X[i + 256 u].  w1 = (cos, -sin)(2 pi i / 4096).  Every thread of the
// 256-thread block calls it; buf is free on entry (callers that used it synchronise first) and holds
// nothing of value on exit.
__device__ __forceinline__ void fft4096(float2 (&a)[16], float2* buf, int i, float2 w1, float sgn) {
  bfly16(a, sgn);
  w1.y *= sgn;
  twiddle16(a, w1);
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(16 * i + u)] = a[u];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = buf[S4K_PAD(i + 256 * k)];
  bfly16(a, sgn);
  {
    float sn, cs;
    sincospif(2.0f * (float)(i >> 4) / 256.0f, &sn, &cs);     // W_256^p, p = i >> 4
    twiddle16(a, make_float2(cs, -sgn * sn));
  }
  __syncthreads();
  const int base = (i & 15) + 256 * (i >> 4);
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(base + 16 * u)] = a[u];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = buf[S4K_PAD(i + 256 * k)];
  bfly16(a, sgn);
}

