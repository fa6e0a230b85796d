// Power-of-two inverse real FFT x = irfft(X, n), n = 2^p, and its adjoint, for gfx950.
// Used where the reference calls torch.fft.irfft with the default length n = 2(K-1):
// utils.py:169 (get_response / IR export) and losses.py:344 (directional EDC loss).
//
// Real data through ONE complex transform of half the length, m = n / 2 (z[j] = x[2j] + i x[2j+1]):
//   inverse:  Z[k] = (X[k] + conj X[m-k]) + i e^{+2 pi i k / n} (X[k] - conj X[m-k]),  z = IFFT_m(Z),  x = z / n
//   forward:  Zf = FFT_m(z),  F[k] = 1/2 [(Zf[k] + conj Zf[m-k]) - i e^{-2 pi i k / n} (Zf[k] - conj Zf[m-k])]
// (the full-length complex transform this replaces moved twice the work block and did twice the butterflies:
// 2.15 ms for the four passes of 512 responses of 131 072 samples).
// Four-step factorisation m = L1 x L2 with both passes tiled through LDS so that every global
// access is contiguous along the tile:
//   inverse  pass A: tile of adjacent k1, Z formed on load, inverse FFT over k2 (stride L1), conj twiddle,
//                    store transposed  work[n2][k1]
//            pass B: tile of adjacent n2 rows, inverse FFT over k1, (x[2j], x[2j+1]) = z[j] / n, j = n1 L2 + n2
//   forward  pass A: tile of adjacent n2, z formed on load, FFT over n1 (stride L2), twiddle, work[k1][n2]
//            pass B: tile of rows k1 that holds the mirror rows L1 - k1 as well (F[k] needs Zf[m-k]), FFT over n2,
//                    F[k1 + L1 k2] for k <= m
#include "common.h"
#include "fft4k_dev.h"

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

struct P2Geom { int n, m, L1, L2; };          // n real samples, m = n / 2 = L1 L2 complex points
static P2Geom p2_geom(int n) {
  P2Geom g; g.n = n; g.m = n / 2;
  int p = ilog2(g.m);
  g.L1 = 1 << (p / 2);
  g.L2 = g.m / g.L1;
  return g;
}

// ---- inverse pass A: k = k1 + L1 k2 ; tile over k1
__global__ __launch_bounds__(256) void k_p2_inv_a(P2Geom g, const float2* __restrict__ X, int ldx,
                                                  float2* __restrict__ work) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, m = g.m, tc = L1 < P2_TC ? L1 : P2_TC, ss = L2 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, c0 = blockIdx.x * tc;
  p2_tw4(tw4, L2);
  const float2* Xb = X + (size_t)b * ldx;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int k2 = idx / tc, cc = idx - k2 * tc;
    const int k = c0 + cc + L1 * k2;
    float2 a = Xb[k], bc = Xb[m - k];
    bc.y = -bc.y;
    if (k == 0) { a.y = 0.f; bc.y = 0.f; }            // irfft ignores Im X[0], Im X[n/2]
    const float2 e = cadd(a, bc);
    const float2 o = cmulc(csub(a, bc), p2_tw(k, n));    // (a - bc) e^{+2 pi i k / n}
    bufA[cc * ss + k2] = make_float2(e.x - o.y, e.y + o.x);
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L2, tc, ss, true, tw4, L2);
  float2* wk = work + (size_t)b * m;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int n2 = idx / tc, cc = idx - n2 * tc;
    const int k1 = c0 + cc;
    const float2 w = p2_tw((int)(((long long)n2 * k1) & (m - 1)), m);
    wk[(size_t)n2 * L1 + k1] = cmulc(r[cc * ss + n2], w);
  }
}
// ---- inverse pass B: rows n2 (tile), inverse FFT over k1, (x[2j], x[2j+1]) = z[j] / n with j = n1 L2 + n2
__global__ __launch_bounds__(256) void k_p2_inv_b(P2Geom g, const float2* __restrict__ work,
                                                  float* __restrict__ x, int ldo) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, m = g.m, tc = L2 < P2_TC ? L2 : P2_TC, ss = L1 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, r0 = blockIdx.x * tc;
  p2_tw4(tw4, L1);
  const float2* wk = work + (size_t)b * m + (size_t)r0 * L1;
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
    const size_t j = (size_t)n1 * L2 + r0 + rr;
    const float2 v = r[rr * ss + n1];
    xb[2 * j] = sc * v.x;
    xb[2 * j + 1] = sc * v.y;
  }
}

// ---- forward pass A: j = n1 L2 + n2 ; z[j] = x[2j] + i x[2j+1] ; tile over n2 ; FFT over n1 ; work[k1][n2]
__global__ __launch_bounds__(256) void k_p2_adj_a(P2Geom g, const float* __restrict__ gx, int ldo,
                                                  int T, float2* __restrict__ work) {
  const int L1 = g.L1, L2 = g.L2, m = g.m, tc = L2 < P2_TC ? L2 : P2_TC, ss = L1 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, c0 = blockIdx.x * tc;
  p2_tw4(tw4, L1);
  const float* gb = gx + (size_t)b * ldo;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int n1 = idx / tc, cc = idx - n1 * tc;
    const int t = 2 * (n1 * L2 + c0 + cc);
    bufA[cc * ss + n1] = make_float2(t < T ? gb[t] : 0.f, t + 1 < T ? gb[t + 1] : 0.f);    // zero padding beyond T
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L1, tc, ss, false, tw4, L1);
  float2* wk = work + (size_t)b * m;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int k1 = idx / tc, cc = idx - k1 * tc;
    const int n2 = c0 + cc;
    const float2 w = p2_tw((int)(((long long)n2 * k1) & (m - 1)), m);
    wk[(size_t)k1 * L2 + n2] = cmul(r[cc * ss + k1], w);
  }
}

// rows of tile q: with L1 <= P2_TC all rows; else the P2_TC / 2 rows h q .. h q + h - 1 and their mirrors L1 - k1
// (tile 0 holds row 0, which mirrors onto itself, and takes row L1 / 2 -- the other self-mirrored row -- in the
// free slot)
#define P2_H (P2_TC / 2)
__device__ __forceinline__ int p2_tile_row(int L1, int q, int i) {
  if (L1 <= P2_TC) return i;
  if (i < P2_H) return P2_H * q + i;
  if (q == 0 && i == P2_TC - 1) return L1 / 2;
  return L1 - P2_H * q - (P2_H - 1) + (i - P2_H);
}
__device__ __forceinline__ int p2_tile_mirror(int L1, int q, int i, int k1) {
  if (k1 == 0 || 2 * k1 == L1) return i;
  if (L1 <= P2_TC) return L1 - k1;
  return P2_TC - 1 - i;                     // rows[h + j] = L1 - h q - (h - 1) + j mirrors rows[h - 1 - j]
}

// ---- forward pass B: rows k1 (+ mirrors), FFT over n2, F[k1 + L1 k2] for k <= m
__global__ __launch_bounds__(256) void k_p2_adj_b(P2Geom g, const float2* __restrict__ work,
                                                  float2* __restrict__ gX, int ldx, int plain) {
  const int L1 = g.L1, L2 = g.L2, n = g.n, m = g.m, tc = L1 < P2_TC ? L1 : P2_TC, ss = L2 + 1;
  float2* bufA = dyn_lds; float2* bufB = bufA + tc * ss; float2* tw4 = bufB + tc * ss;
  const int b = blockIdx.y, q = blockIdx.x;
  p2_tw4(tw4, L2);
  const float2* wk = work + (size_t)b * m;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int rr = idx / L2, n2 = idx - rr * L2;
    bufA[rr * ss + n2] = wk[(size_t)p2_tile_row(L1, q, rr) * L2 + n2];
  }
  __syncthreads();
  float2* r = p2_fft(bufA, bufB, L2, tc, ss, false, tw4, L2);
  float2* o = gX + (size_t)b * ldx;
  const float sc = 1.0f / (float)n;
  for (int idx = threadIdx.x; idx < tc * L2; idx += blockDim.x) {
    const int k2 = idx / tc, rr = idx - k2 * tc;
    const int k1 = p2_tile_row(L1, q, rr);
    const int k = k1 + L1 * k2;
    const int rp = p2_tile_mirror(L1, q, rr, k1);
    const int k2p = k1 == 0 ? (L2 - k2) & (L2 - 1) : L2 - 1 - k2;
    const float2 zk = r[rr * ss + k2], zp = cconj(r[rp * ss + k2p]);
    const float2 e = cadd(zk, zp), d = cmul(csub(zk, zp), p2_tw(k, n));      // (zk - zp) e^{-2 pi i k / n}
    float2 v = make_float2(0.5f * (e.x + d.y), 0.5f * (e.y - d.x));           // 1/2 (e - i d)
    if (!plain) {                                  // adjoint-of-irfft scaling
      if (k == 0) v = make_float2(sc * v.x, 0.f);
      else v = cscale(v, 2.0f * sc);
    }
    o[k] = v;
    if (k == 0) {                                  // k = m (Nyquist): Re Zf[0] - Im Zf[0]
      const float ny = zk.x - zk.y;
      o[m] = make_float2(plain ? ny : sc * ny, 0.f);
    }
  }
}
// ------------------------------------------------------------------------------------------
// n = 131 072 (m = 65 536 = 256 x 256): the same four passes with the 256-point transforms held in registers.
// Sixteen threads share one transform: thread r holds x[r + 16 j], j = 0..15 -> 16-point butterfly over j, twiddle
// W_256^(r p), ONE exchange through LDS (thread p collects index p of the sixteen threads), 16-point butterfly over r
// -> X[p + 16 s].  A workgroup of 256 threads carries 16 transforms.  In the passes that walk columns of the (256, 256)
// block (inverse A, forward A) the 16 adjacent columns are the fast thread index: every global access is a 128-byte
// segment per half-wave straight from / to registers; in the passes that walk rows (inverse B, forward B) the position
// in the row is the fast index (128-byte segments again), the exchange stays inside a wave (no block barrier) and the
// results reach their transposed places through one LDS tile.  The generic Stockham passes above (radix 4, eight barriers
// per pass, 8-column tiles = 64-byte segments) took 154-208 us per pass of 384 transforms; see DESIGN §4.
// ------------------------------------------------------------------------------------------
#define PW_CP 292      // LDS slots per column of the column passes (16 x 17 + 20: a wave's 64 accesses fall 2 per bank pair)
#define PW_RP 272      // ... per row of the row passes (16 x 17)
#define PW_TP 18       // pitch of the (256, 16) transposed tile of inverse pass B
#define PW_OP 257      // pitch of the (16, 256) tile of forward pass B
#define PW_LDS 4672    // float2 slots: max(16 * PW_CP, 16 * PW_RP, 256 * PW_TP, 16 * PW_OP)

// (cos, sin)(2 pi j / 32), j = 0..15
__constant__ float2 c_pw32[16] = {
    {1.0f, 0.0f}, {0.98078528040323043f, 0.19509032201612825f}, {0.92387953251128674f, 0.38268343236508977f},
    {0.83146961230254524f, 0.55557023301960218f}, {0.70710678118654752f, 0.70710678118654752f},
    {0.55557023301960218f, 0.83146961230254524f}, {0.38268343236508977f, 0.92387953251128674f},
    {0.19509032201612825f, 0.98078528040323043f}, {0.0f, 1.0f}, {-0.19509032201612825f, 0.98078528040323043f},
    {-0.38268343236508977f, 0.92387953251128674f}, {-0.55557023301960218f, 0.83146961230254524f},
    {-0.70710678118654752f, 0.70710678118654752f}, {-0.83146961230254524f, 0.55557023301960218f},
    {-0.92387953251128674f, 0.38268343236508977f}, {-0.98078528040323043f, 0.19509032201612825f}};

// in: a[j] = x[r + 16 j]; out: a[s] = X[r + 16 s], X[q] = sum x[k] e^(-sgn 2 pi i k q / 256).  `seq`: the transform's 16 x 17
// LDS slots.  BLOCK: the sixteen threads sit in different waves (block barrier) or in one (LDS operations of a wave
// execute in order: none).
template <bool BLOCK>
__device__ __forceinline__ void pw_fft256(float2 (&a)[16], float2* seq, int r, float sgn) {
  bfly16(a, sgn);
  {
    float sn, cs;
    sincospif(2.0f * (float)r / 256.0f, &sn, &cs);
    twiddle16(a, make_float2(cs, -sgn * sn));
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) seq[17 * r + p] = a[p];
  if (BLOCK) __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = seq[17 * q + r];
  bfly16(a, sgn);
}
// a[s] *= t0 step^s
__device__ __forceinline__ void pw_twiddle(float2 (&a)[16], float2 t0, float2 step) {
  twiddle16(a, step);
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) a[s_] = cmul(a[s_], t0);
}
__device__ __forceinline__ float2 pw_cis(float turns) {        // e^(2 pi i turns)
  float sn, cs;
  sincospif(2.0f * turns, &sn, &cs);
  return make_float2(cs, sn);
}

// block -> (tile of 16, item): consecutive blocks go round-robin to the 8 XCDs, each with its own L2 -- the 16 tiles of an
// item are given to ONE XCD (blocks id, id + 8, ...), so that what its tiles share meets in one L2: the cache lines the
// 128-byte segments of neighbouring tiles straddle (rows of 65 537 complex numbers are 8 bytes off the line grid), the
// mirrored bins of inverse pass A, the partial lines of forward pass B (PMC: 692 MB of traffic for inverse pass A of 288
// items against 302 MB of algorithmic bytes with the plain map).  Blocks past the batch (batch not a multiple of 8) exit.
__device__ __forceinline__ bool pw_item_map(int batch, int& tile, int& b) {
  const int id = blockIdx.x, slot = id >> 3;
  tile = slot & 15;
  b = (slot >> 4) * 8 + (id & 7);
  return b < batch;
}
static int pw_grid(int batch) { return ((batch + 7) / 8) * 8 * 16; }

// inverse pass A: columns k1 = c0 + cc, Z formed on load, inverse transform over k2, twiddle e^(+2 pi i n2 k1 / m), work[n2][k1]
// kvalid: bins k >= kvalid count as zero and are not read (a spectrum that ends below the Nyquist bin: csrc/polyfft.hip)
__global__ __launch_bounds__(256) void k_pw_inv_a(const float2* __restrict__ X, int ldx, float2* __restrict__ work, int batch,
                                                  int kvalid) {
  constexpr int m = 65536, n = 131072;
  float2* buf = dyn_lds;
  int tile, b;
  if (!pw_item_map(batch, tile, b)) return;
  const int c0 = tile * 16;
  const int cc = threadIdx.x & 15, r = threadIdx.x >> 4, k1 = c0 + cc;
  const float2* Xb = X + (size_t)b * ldx;
  const float2 e0 = pw_cis((float)(k1 + 256 * r) / (float)n);          // e^(+2 pi i k / n) at j = 0
  float2 a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k1 + 256 * (r + 16 * j);
    float2 x = k < kvalid ? Xb[k] : make_float2(0.f, 0.f), bc = m - k < kvalid ? Xb[m - k] : make_float2(0.f, 0.f);
    bc.y = -bc.y;
    if (k == 0) { x.y = 0.f; bc.y = 0.f; }                // irfft ignores Im X[0], Im X[n/2]
    const float2 e = cadd(x, bc);
    const float2 o = cmul(csub(x, bc), cmul(e0, c_pw32[j]));
    a[j] = make_float2(e.x - o.y, e.y + o.x);
  }
  pw_fft256<true>(a, buf + cc * PW_CP, r, -1.0f);
  pw_twiddle(a, pw_cis((float)(r * k1) / (float)m), pw_cis((float)k1 / 4096.0f));
  float2* wk = work + (size_t)b * m + k1;
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) wk[(size_t)(r + 16 * s_) * 256] = a[s_];
}
// inverse pass B: rows n2 = r0 + rr, inverse transform over k1, (x[2j], x[2j+1]) = z[j] / n with j = n1 256 + n2
// tout: only the samples t < tout are stored (a caller that gathers from the head of the signal)
__global__ __launch_bounds__(256) void k_pw_inv_b(const float2* __restrict__ work, float* __restrict__ x, int ldo, int batch,
                                                  int tout) {
  constexpr int m = 65536, n = 131072;
  float2* buf = dyn_lds;
  int tile, b;
  if (!pw_item_map(batch, tile, b)) return;
  const int r0 = tile * 16;
  const int r = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float2* wk = work + (size_t)b * m + (size_t)(r0 + rr) * 256 + r;
  float2 a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = wk[16 * j];
  pw_fft256<false>(a, buf + rr * PW_RP, r, -1.0f);
  __syncthreads();
  const float sc = 1.0f / (float)n;
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) buf[(r + 16 * s_) * PW_TP + rr] = cscale(a[s_], sc);
  __syncthreads();
  float2* xb = (float2*)(x + (size_t)b * ldo) + r0 + (threadIdx.x & 15);
  const int nlo = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n1 = nlo + 16 * i;
    if (n1 * 512 < tout) xb[(size_t)n1 * 256] = buf[n1 * PW_TP + (threadIdx.x & 15)];      // (samples 2 (n1 256 + n2), + 1)
  }
}
// forward pass A: columns n2 = c0 + cc, z[j] = (x[2j], x[2j+1]) (zero beyond T), transform over n1, twiddle, work[k1][n2]
// (samples outside [t_lo, T) count as zero and are not read)
__global__ __launch_bounds__(256) void k_pw_fwd_a(const float* __restrict__ gx, int ldo, int t_lo, int T,
                                                  float2* __restrict__ work, int batch) {
  constexpr int m = 65536;
  float2* buf = dyn_lds;
  int tile, b;
  if (!pw_item_map(batch, tile, b)) return;
  const int c0 = tile * 16;
  const int cc = threadIdx.x & 15, r = threadIdx.x >> 4, n2 = c0 + cc;
  const float* gb = gx + (size_t)b * ldo;
  const bool vec = (ldo & 1) == 0;
  float2 a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int t = 2 * ((r + 16 * j) * 256 + n2);
    if (vec && t >= t_lo && t + 1 < T) a[j] = *(const float2*)(gb + t);
    else a[j] = make_float2((t >= t_lo && t < T) ? gb[t] : 0.f, (t + 1 >= t_lo && t + 1 < T) ? gb[t + 1] : 0.f);
  }
  pw_fft256<true>(a, buf + cc * PW_CP, r, 1.0f);
  pw_twiddle(a, pw_cis(-(float)(r * n2) / (float)m), pw_cis(-(float)n2 / 4096.0f));
  float2* wk = work + (size_t)b * m + n2;
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) wk[(size_t)(r + 16 * s_) * 256] = a[s_];
}
// rows of tile q of forward pass B: k1 = 8 q + i (i < 8) and their mirrors 256 - k1 (slot 15 - i); tile 0 holds row 0, which
// mirrors onto itself, and row 128 -- the other self-mirrored row -- in the free slot 15
__device__ __forceinline__ int pw_tile_row(int q, int i) {
  if (i < 8) return 8 * q + i;
  if (q == 0 && i == 15) return 128;
  return 256 - 8 * q - 7 + (i - 8);
}
// forward pass B: rows k1 (+ mirrors), transform over n2, F[k1 + 256 k2] for k <= m
__global__ __launch_bounds__(256) void k_pw_fwd_b(const float2* __restrict__ work, float2* __restrict__ gX, int ldx, int plain,
                                                  int batch) {
  constexpr int m = 65536, n = 131072;
  float2* buf = dyn_lds;
  int q, b;
  if (!pw_item_map(batch, q, b)) return;
  const int r = threadIdx.x & 15, i = threadIdx.x >> 4;
  const float2* wk = work + (size_t)b * m + (size_t)pw_tile_row(q, i) * 256 + r;
  float2 a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = wk[16 * j];
  pw_fft256<false>(a, buf + i * PW_RP, r, 1.0f);
  __syncthreads();
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) buf[i * PW_OP + r + 16 * s_] = a[s_];
  __syncthreads();
  const int ii = threadIdx.x & 15, klo = threadIdx.x >> 4;
  const int k1 = pw_tile_row(q, ii);
  const bool self = k1 == 0 || k1 == 128;
  const int rp = self ? ii : 15 - ii;
  float2* o = gX + (size_t)b * ldx;
  const float sc = 1.0f / (float)n;
  const float2 e0 = pw_cis(-(float)(k1 + 256 * klo) / (float)n);        // e^(-2 pi i k / n) at u = 0
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int k2 = klo + 16 * u, k = k1 + 256 * k2;
    const int k2p = k1 == 0 ? (256 - k2) & 255 : 255 - k2;
    const float2 zk = buf[ii * PW_OP + k2], zp = cconj(buf[rp * PW_OP + k2p]);
    const float2 e = cadd(zk, zp), d = cmul(csub(zk, zp), cmulc(e0, c_pw32[u]));   // (zk - zp) e^{-2 pi i k / n}
    float2 v = make_float2(0.5f * (e.x + d.y), 0.5f * (e.y - d.x));                   // 1/2 (e - i d)
    if (!plain) {                                  // adjoint-of-irfft scaling
      if (k == 0) v = make_float2(sc * v.x, 0.f);
      else v = cscale(v, 2.0f * sc);
    }
    o[k] = v;
    if (k == 0) {                                  // k = m (Nyquist): Re Zf[0] - Im Zf[0]
      const float ny = zk.x - zk.y;
      o[m] = make_float2(plain ? ny : sc * ny, 0.f);
    }
  }
}
static bool pw_ok(const P2Geom& g) { return g.L1 == 256 && g.L2 == 256; }

static size_t p2_lds(int len, int tc) { return ((size_t)2 * tc * (len + 1) + (len >= 4 ? len / 4 : 1)) * sizeof(float2); }

extern "C" size_t gfdn_irfft_pow2_work_bytes(int n, int batch) {
  if (n < 16 || (n & (n - 1)) || batch <= 0) return 0;
  return (size_t)batch * n * sizeof(float2);       // (the half-length transform uses the first half)
}

static int p2_inverse_real(int n, const float* X, int ldx, int batch, float* x, int ldo, void* work, void* stream, int kvalid,
                           int tout);
extern "C" int gfdn_irfft_pow2_fwd(int n, const float* X, int ldx, int batch, float* x, int ldo,
                                   void* work, void* stream) {
  return p2_inverse_real(n, X, ldx, batch, x, ldo, work, stream, n / 2 + 1, n);
}
// ... of spectra that vanish from bin kvalid on (those bins are not read: they may be uninitialised), storing the samples
// t < tout only (rounded up to 512; the rest of x is left as it is).  n = 131 072: the register-resident passes skip the
// loads / stores; other lengths run the full transform (the limits are then hints).
extern "C" int gfdn_irfft_pow2_fwd_band(int n, const float* X, int ldx, int kvalid, int batch, float* x, int ldo, int tout,
                                        void* work, void* stream) {
  if (kvalid <= 0 || kvalid > n / 2 + 1 || tout <= 0 || tout > n) return GFDN_E_BADARG;
  return p2_inverse_real(n, X, ldx, batch, x, ldo, work, stream, kvalid, tout);
}
static int p2_inverse_real(int n, const float* X, int ldx, int batch, float* x, int ldo, void* work, void* stream, int kvalid,
                           int tout) {
  if (!X || !x || !work || n < 16 || (n & (n - 1)) || batch <= 0) return GFDN_E_BADARG;
  if (ldx < n / 2 + 1 || ldo < n) return GFDN_E_BADARG;
  P2Geom g = p2_geom(n);
  if (g.L2 > 2048) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (pw_ok(g) && (ldo & 1) == 0) {
    hipLaunchKernelGGL(k_pw_inv_a, dim3(pw_grid(batch)), dim3(256), PW_LDS * sizeof(float2), s, (const float2*)X, ldx, (float2*)work,
                       batch, kvalid);
    GFDN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pw_inv_b, dim3(pw_grid(batch)), dim3(256), PW_LDS * sizeof(float2), s, (const float2*)work, x, ldo, batch,
                       tout);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
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
                           void* work, void* stream, int plain, int t_lo = 0);

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

// adjoint of irfft for a gradient that vanishes outside the samples [t_lo, t_hi): nothing outside the window is read
// (it may be uninitialised).  n = 131 072 only (the register-resident passes).
extern "C" int gfdn_irfft_pow2_bwd_window(int n, const float* gx, int ldo, int batch, int t_lo, int t_hi, float* gX, int ldx,
                                          void* work, void* stream) {
  if (ldo < n || t_lo < 0 || t_hi > n || t_lo >= t_hi) return GFDN_E_BADARG;
  if (n != 131072) return GFDN_E_UNSUPPORTED;
  return p2_forward_real(n, gx, ldo, t_hi, batch, gX, ldx, work, stream, 0, t_lo);
}

static int p2_forward_real(int n, const float* gx, int ldo, int T, int batch, float* gX, int ldx,
                           void* work, void* stream, int plain, int t_lo) {
  if (!gx || !gX || !work || n < 16 || (n & (n - 1)) || batch <= 0) return GFDN_E_BADARG;
  if (ldx < n / 2 + 1) return GFDN_E_BADARG;
  P2Geom g = p2_geom(n);
  if (g.L2 > 2048) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (pw_ok(g)) {
    hipLaunchKernelGGL(k_pw_fwd_a, dim3(pw_grid(batch)), dim3(256), PW_LDS * sizeof(float2), s, gx, ldo, t_lo, T, (float2*)work,
                       batch);
    GFDN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pw_fwd_b, dim3(pw_grid(batch)), dim3(256), PW_LDS * sizeof(float2), s, (const float2*)work, (float2*)gX, ldx,
                       plain, batch);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
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
