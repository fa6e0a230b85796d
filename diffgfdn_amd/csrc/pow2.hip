// Power-of-two inverse real FFT x = irfft(X, n), n = 2^p, and its adjoint, for gfx950.
// Used where the reference calls torch.fft.irfft with the default length n = 2(K-1):
// utils.py:169 (get_response / IR export) and losses.py:344 (directional EDC loss).
//
// Four-step factorisation n = L1 x L2 with both passes tiled through LDS so that every global
// access is contiguous along the tile:
//   inverse  pass A: tile of adjacent k1, inverse FFT over k2 (stride L1), conj twiddle,
//                    store transposed  work[n2][k1]
//            pass B: tile of adjacent n2 rows, inverse FFT over k1, x[n1 L2 + n2]
//   adjoint  pass A: tile of adjacent n2, FFT over n1 (stride L2), twiddle, work[k1][n2]
//            pass B: tile of adjacent k1 rows, FFT over n2, gX[k1 + L1 k2] (k <= n/2)
// The Hermitian half of the spectrum is expanded on the fly; the complex transform is full
// length (2x redundant for real data) -- these calls are off the training hot path.
#include "common.h"

extern __shared__ float2 dyn_lds[];

// local copy of the Stockham pass (kept in this translation unit so it can be inlined)
__device__ __forceinline__ float2* p2_fft(float2* x, float2* y, int n, int nseq, int ss,
                                          bool inverse, const float2* tw4, int tn) {
  const int nthr = blockDim.x;
  int nn = n, s = 1, ls = 0;
  const float sgn = inverse ? -1.0f : 1.0f;
  while (nn >= 4) {
    const int m = nn >> 2, tws = tn / nn, per = n >> 2;
    for (int idx = threadIdx.x; idx < nseq * per; idx += nthr) {
      const int seq = idx / per, i = idx - seq * per;
      const int p = i >> ls, q = i & (s - 1);
      float2 w1 = tw4[p * tws];
      w1.y *= sgn;
      const float2 w2 = cmul(w1, w1), w3 = cmul(w1, w2);
      const float2* xb = x + seq * ss;
      float2* yb = y + seq * ss;
      const float2 a = xb[q + s * p], b = xb[q + s * (p + m)];
      const float2 c = xb[q + s * (p + 2 * m)], d = xb[q + s * (p + 3 * m)];
      const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
      const float2 jbmd = make_float2(sgn * bmd.y, -sgn * bmd.x);
      yb[q + s * (4 * p + 0)] = cadd(apc, bpd);
      yb[q + s * (4 * p + 1)] = cmul(w1, cadd(amc, jbmd));
      yb[q + s * (4 * p + 2)] = cmul(w2, csub(apc, bpd));
      yb[q + s * (4 * p + 3)] = cmul(w3, csub(amc, jbmd));
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    nn = m; s <<= 2; ls += 2;
  }
  if (nn == 2) {
    const int per = n >> 1;
    for (int idx = threadIdx.x; idx < nseq * per; idx += nthr) {
      const int seq = idx / per, q = idx - seq * per;
      const float2* xb = x + seq * ss;
      float2* yb = y + seq * ss;
      const float2 a = xb[q], b = xb[q + s];
      yb[q] = cadd(a, b);
      yb[q + s] = csub(a, b);
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
  }
  return x;
}

__device__ __forceinline__ void p2_tw4(float2* tw4, int tn) {
  const int cnt = tn >= 4 ? tn / 4 : 1;
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)j / (float)tn, &s, &c);
    tw4[j] = make_float2(c, -s);
  }
}
// exp(-2 pi i e / L), e < L, direct evaluation (exact argument for power-of-two L)
__device__ __forceinline__ float2 p2_tw(int e, int L) {
  float s, c;
  sincospif(2.0f * (float)e / (float)L, &s, &c);
  return make_float2(c, -s);
}

#define P2_TC 8

struct P2Geom { int n, L1, L2; };
static P2Geom p2_geom(int n) {
  P2Geom g; g.n = n;
  int p = ilog2(n);
  g.L1 = 1 << (p / 2);
  g.L2 = n / g.L1;
  return g;
}

// ---- inverse pass A: k = k1 + L1 k2 ; tile over k1
__global__ __launch_bounds__(256) void k_p2_inv_a(P2Geom g, const float2* __restrict__ X, int ldx,
                                                  float2* __restrict__ work) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, tc = L1 < P2_TC ? L1 : P2_TC, ss = L2 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, c0 = blockIdx.x * tc;
  p2_tw4(tw4, L2);
  const float2* Xb = X + (size_t)b * ldx;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int k2 = idx / tc, cc = idx - k2 * tc;
    const int k = c0 + cc + L1 * k2;
    float2 v;
    if (k <= n / 2) { v = Xb[k]; if (k == 0 || k == n / 2) v.y = 0.f; }
    else { v = Xb[n - k]; v.y = -v.y; }
    bufA[cc * ss + k2] = v;
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L2, tc, ss, true, tw4, L2);
  float2* wk = work + (size_t)b * n;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int n2 = idx / tc, cc = idx - n2 * tc;
    const int k1 = c0 + cc;
    const float2 w = p2_tw((int)(((long long)n2 * k1) & (n - 1)), n);
    wk[(size_t)n2 * L1 + k1] = cmulc(r[cc * ss + n2], w);
  }
}
// ---- inverse pass B: rows n2 (tile), inverse FFT over k1, x[n1 L2 + n2]
__global__ __launch_bounds__(256) void k_p2_inv_b(P2Geom g, const float2* __restrict__ work,
                                                  float* __restrict__ x, int ldo) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, tc = L2 < P2_TC ? L2 : P2_TC, ss = L1 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, r0 = blockIdx.x * tc;
  p2_tw4(tw4, L1);
  const float2* wk = work + (size_t)b * n + (size_t)r0 * L1;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int rr = idx / L1, k1 = idx - rr * L1;
    bufA[rr * ss + k1] = wk[idx];
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L1, tc, ss, true, tw4, L1);
  const float sc = 1.0f / (float)n;
  float* xb = x + (size_t)b * ldo;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int n1 = idx / tc, rr = idx - n1 * tc;
    xb[(size_t)n1 * L2 + r0 + rr] = sc * r[rr * ss + n1].x;
  }
}
// ---- adjoint pass A: t = n1 L2 + n2 ; tile over n2 ; FFT over n1 ; work[k1][n2]
__global__ __launch_bounds__(256) void k_p2_adj_a(P2Geom g, const float* __restrict__ gx, int ldo,
                                                  int T, float2* __restrict__ work) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, tc = L2 < P2_TC ? L2 : P2_TC, ss = L1 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, c0 = blockIdx.x * tc;
  p2_tw4(tw4, L1);
  const float* gb = gx + (size_t)b * ldo;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int n1 = idx / tc, cc = idx - n1 * tc;
    const int t = n1 * L2 + c0 + cc;
    bufA[cc * ss + n1] = make_float2(t < T ? gb[t] : 0.f, 0.f);     // zero padding beyond T
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L1, tc, ss, false, tw4, L1);
  float2* wk = work + (size_t)b * n;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int k1 = idx / tc, cc = idx - k1 * tc;
    const int n2 = c0 + cc;
    const float2 w = p2_tw((int)(((long long)n2 * k1) & (n - 1)), n);
    wk[(size_t)k1 * L2 + n2] = cmul(r[cc * ss + k1], w);
  }
}
// ---- adjoint pass B: rows k1 (tile), FFT over n2, gX[k1 + L1 k2] for k <= n/2
__global__ __launch_bounds__(256) void k_p2_adj_b(P2Geom g, const float2* __restrict__ work,
                                                  float2* __restrict__ gX, int ldx, int plain) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, tc = L1 < P2_TC ? L1 : P2_TC, ss = L2 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, r0 = blockIdx.x * tc;
  p2_tw4(tw4, L2);
  const float2* wk = work + (size_t)b * n + (size_t)r0 * L2;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int rr = idx / L2, n2 = idx - rr * L2;
    bufA[rr * ss + n2] = wk[idx];
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L2, tc, ss, false, tw4, L2);
  float2* o = gX + (size_t)b * ldx;
  const float sc = 1.0f / (float)n;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int k2 = idx / tc, rr = idx - k2 * tc;
    const int k = r0 + rr + L1 * k2;
    if (k <= n / 2) {
      float2 v = r[rr * ss + k2];
      if (!plain) {                                  // adjoint-of-irfft scaling
        if (k == 0 || k == n / 2) v = make_float2(sc * v.x, 0.f);
        else v = cscale(v, 2.0f * sc);
      }
      o[k] = v;
    }
  }
}

static size_t p2_lds(int len, int tc) { return ((size_t)2 * tc * (len + 1) + (len >= 4 ? len / 4 : 1)) * sizeof(float2); }

extern "C" size_t gfdn_irfft_pow2_work_bytes(int n, int batch) {
  if (n < 16 || (n & (n - 1)) || batch <= 0) return 0;
  return (size_t)batch * n * sizeof(float2);
}

extern "C" int gfdn_irfft_pow2_fwd(int n, const float* X, int ldx, int batch, float* x, int ldo,
                                   void* work, void* stream) {
  if (!X || !x || !work || n < 16 || (n & (n - 1)) || batch <= 0) return GFDN_E_BADARG;
  if (ldx < n / 2 + 1 || ldo < n) return GFDN_E_BADARG;
  P2Geom g = p2_geom(n);
  if (g.L2 > 2048) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int tca = g.L1 < P2_TC ? g.L1 : P2_TC, tcb = g.L2 < P2_TC ? g.L2 : P2_TC;
  int rc;
  if ((rc = ensure_dyn_lds(k_p2_inv_a, p2_lds(g.L2, tca)))) return rc;
  if ((rc = ensure_dyn_lds(k_p2_inv_b, p2_lds(g.L1, tcb)))) return rc;
  hipLaunchKernelGGL(k_p2_inv_a, dim3(g.L1 / tca, batch), dim3(256), p2_lds(g.L2, tca), s, g,
                     (const float2*)X, ldx, (float2*)work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_p2_inv_b, dim3(g.L2 / tcb, batch), dim3(256), p2_lds(g.L1, tcb), s, g,
                     (const float2*)work, x, ldo);
  GFDN_LAUNCH_CHECK();
  return 0;
}

static int p2_forward_real(int n, const float* gx, int ldo, int T, int batch, float* gX, int ldx,
                           void* work, void* stream, int plain);

extern "C" int gfdn_irfft_pow2_bwd(int n, const float* gx, int ldo, int batch, float* gX, int ldx,
                                   void* work, void* stream) {
  if (ldo < n) return GFDN_E_BADARG;
  return p2_forward_real(n, gx, ldo, n, batch, gX, ldx, work, stream, 0);
}

// X = rfft(x[0:T], n): the dataset front end (dataloader.py:250, :320-325)
extern "C" int gfdn_rfft_pow2(int n, const float* x, int ld, int T, int batch, float* X, int ldx,
                              void* work, void* stream) {
  if (T <= 0 || T > n || ld < T) return GFDN_E_BADARG;
  return p2_forward_real(n, x, ld, T, batch, X, ldx, work, stream, 1);
}

static int p2_forward_real(int n, const float* gx, int ldo, int T, int batch, float* gX, int ldx,
                           void* work, void* stream, int plain) {
  if (!gx || !gX || !work || n < 16 || (n & (n - 1)) || batch <= 0) return GFDN_E_BADARG;
  if (ldx < n / 2 + 1) return GFDN_E_BADARG;
  P2Geom g = p2_geom(n);
  if (g.L2 > 2048) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int tca = g.L2 < P2_TC ? g.L2 : P2_TC, tcb = g.L1 < P2_TC ? g.L1 : P2_TC;
  int rc;
  if ((rc = ensure_dyn_lds(k_p2_adj_a, p2_lds(g.L1, tca)))) return rc;
  if ((rc = ensure_dyn_lds(k_p2_adj_b, p2_lds(g.L2, tcb)))) return rc;
  hipLaunchKernelGGL(k_p2_adj_a, dim3(g.L2 / tca, batch), dim3(256), p2_lds(g.L1, tca), s, g, gx,
                     ldo, T, (float2*)work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_p2_adj_b, dim3(g.L1 / tcb, batch), dim3(256), p2_lds(g.L2, tcb), s, g,
                     (const float2*)work, (float2*)gX, ldx, plain);
  GFDN_LAUNCH_CHECK();
  return 0;
}
