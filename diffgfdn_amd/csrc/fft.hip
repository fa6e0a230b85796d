// FFT kernels of the loss front end, for gfx950.
//
// (1) irfft(X, n) with n ODD (losses.py:207-213, :442-445 call torch.fft.irfft(X, n = K) with
//     K = nfft/2 + 1 = 65 537, a Fermat prime; only bins 0..(n-1)/2 are used).  Implemented as a
//     one-sided Bluestein chirp-z transform: the (n+1)/2 input bins are chirp-modulated,
//     circularly convolved with the conjugate chirp by power-of-two FFTs of length
//     L >= n + (n-1)/2 (2^17 at n = 65 537), de-chirped, and the real part kept.  The length-L
//     FFT is a four-step L = L1 x L2 factorisation; because forward FFT, spectrum product and
//     inverse FFT act on the same L2-rows, they are fused into ONE row kernel and the
//     spectrum is never transposed:
//        k_blu_col_fwd : chirp * input, FFT over n1 (column tiles through LDS), twiddle
//        k_blu_row     : FFT over n2, * chirp spectrum, inverse FFT over k2, conj twiddle
//        k_blu_col_inv : inverse FFT over k1, de-chirp, real part / adjoint epilogue
//     The adjoint (backward) runs the same three kernels with conjugated chirps.
//     When n is a Fermat prime (n - 1 = 2^m: 17, 257, 65 537 -- the north-star K) Rader's algorithm
//     is used instead: with g = 3 a primitive root, x[g^-b] = (X'_0 + sum_a Y[g^a] w^(g^(a-b)))/n is a
//     cyclic correlation of length n - 1 = 2^16, HALF the one-sided Bluestein length, run by the very
//     same kernels with a gather (k = g^a) in front and a scatter (t = g^-b) behind; the adjoint
//     multiplies by conj(chat) (-1)^f and swaps the roles of the two index tables.
// (2) |STFT|^2 with a periodic Hann window, hop = win/2, center = False (losses.py:501-535):
//     two real frames are packed into one complex FFT held in LDS; the adjoint recomputes
//     the frame spectra and scatters window * gradient with atomic adds (two frames per
//     sample, so the sum is order-independent).
//
// All FFTs are radix-4 (+ one radix-2) Stockham autosort passes in LDS with ping-pong
// buffers.  Twiddles come from a quarter-period table built in LDS with sincospif (exact
// argument reduction for power-of-two lengths), w2 = w1^2, w3 = w1 w2.
#include "common.h"
#include "fft4k_dev.h"

#include <cmath>
#include <complex>
#include <vector>

// ------------------------------------------------------------------------------------------
// LDS Stockham FFT.  x, y: ping-pong buffers with `nseq` sequences of length n, sequence
// stride ss.  tw4: quarter table, tw4[j] = exp(-2 pi i j / tn), j < tn/4, tn >= n (pow2).
// inverse = true uses conjugated twiddles (un-normalised).  Returns the buffer that holds the
// result.  Every thread of the block must call it.
// ------------------------------------------------------------------------------------------
// in-place 8-point DFT, natural-order output; sgn = +1 forward (e^-), -1 inverse
__device__ __forceinline__ void bfly8(float2 (&a)[8], float sgn) {
  const float r = 0.70710678118654752f;
  float2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
  float2 b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
  float2 b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]);
  float2 b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
  b5 = make_float2(r * (b5.x + sgn * b5.y), r * (b5.y - sgn * b5.x));      // * W8^1
  b6 = make_float2(sgn * b6.y, -sgn * b6.x);                               // * W8^2 = -j
  b7 = make_float2(r * (-b7.x + sgn * b7.y), r * (-b7.y - sgn * b7.x));    // * W8^3
  float2 c0 = cadd(b0, b2), c2 = csub(b0, b2), c1 = cadd(b1, b3), t = csub(b1, b3);
  float2 c3 = make_float2(sgn * t.y, -sgn * t.x);
  float2 d0 = cadd(b4, b6), d2 = csub(b4, b6), d1 = cadd(b5, b7);
  t = csub(b5, b7);
  float2 d3 = make_float2(sgn * t.y, -sgn * t.x);
  a[0] = cadd(c0, c1); a[4] = csub(c0, c1); a[2] = cadd(c2, c3); a[6] = csub(c2, c3);
  a[1] = cadd(d0, d1); a[5] = csub(d0, d1); a[3] = cadd(d2, d3); a[7] = csub(d2, d3);
}
// a[u] *= w^u, u = 1..7
__device__ __forceinline__ void twiddle8(float2 (&a)[8], float2 w1) {
  const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3); a[4] = cmul(a[4], w4);
  a[5] = cmul(a[5], cmul(w4, w1)); a[6] = cmul(a[6], cmul(w3, w3)); a[7] = cmul(a[7], cmul(w4, w3));
}

__device__ __forceinline__ float2* lds_fft(float2* x, float2* y, int n, int nseq, int ss,
                                           bool inverse, const float2* tw4, int tn) {
  const int nthr = blockDim.x;
  int nn = n, s = 1, ls = 0;  // ls = log2(s)
  const float sgn = inverse ? -1.0f : 1.0f;
  const int ln = 31 - __clz(n);          // n, tn are powers of two: shifts instead of divisions
  const int ltn = 31 - __clz(tn);
  int lnn = ln;
  while (nn >= 8) {           // radix-8 Stockham passes
    const int m = nn >> 3;
    const int tws = 1 << (ltn - lnn);
    const int per = n >> 3, lper = ln - 3;
    for (int idx = threadIdx.x; idx < nseq * per; idx += nthr) {
      const int seq = idx >> lper, i = idx & (per - 1);
      const int p = i >> ls, q = i & (s - 1);
      float2 w1 = tw4[p * tws];
      w1.y *= sgn;
      const float2* xb = x + seq * ss;
      float2* yb = y + seq * ss;
      float2 a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = xb[q + s * (p + k * m)];
      bfly8(a, sgn);
      twiddle8(a, w1);
#pragma unroll
      for (int u = 0; u < 8; ++u) yb[q + s * (8 * p + u)] = a[u];
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    nn = m;
    lnn -= 3;
    s <<= 3;
    ls += 3;
  }
  while (nn >= 4) {
    const int m = nn >> 2;
    const int tws = 1 << (ltn - lnn);
    const int per = n >> 2, lper = ln - 2;  // butterflies per sequence
    for (int idx = threadIdx.x; idx < nseq * per; idx += nthr) {
      const int seq = idx >> lper, i = idx & (per - 1);
      const int p = i >> ls, q = i & (s - 1);
      float2 w1 = tw4[p * tws];
      w1.y *= sgn;
      const float2 w2 = cmul(w1, w1);
      const float2 w3 = cmul(w1, w2);
      const float2* xb = x + seq * ss;
      float2* yb = y + seq * ss;
      const float2 a = xb[q + s * p];
      const float2 b = xb[q + s * (p + m)];
      const float2 c = xb[q + s * (p + 2 * m)];
      const float2 d = xb[q + s * (p + 3 * m)];
      const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d);
      float2 bmd = csub(b, d);
      // forward: -j (b - d);  inverse: +j (b - d)
      float2 jbmd = make_float2(sgn * bmd.y, -sgn * bmd.x);  // = -j*bmd (fwd)
      yb[q + s * (4 * p + 0)] = cadd(apc, bpd);
      yb[q + s * (4 * p + 1)] = cmul(w1, cadd(amc, jbmd));
      yb[q + s * (4 * p + 2)] = cmul(w2, csub(apc, bpd));
      yb[q + s * (4 * p + 3)] = cmul(w3, csub(amc, jbmd));
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    nn = m;
    lnn -= 2;
    s <<= 2;
    ls += 2;
  }
  if (nn == 2) {
    const int per = n >> 1, lper = ln - 1;  // here s == n/2, p == 0
    for (int idx = threadIdx.x; idx < nseq * per; idx += nthr) {
      const int seq = idx >> lper, q = idx & (per - 1);
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

// quarter twiddle table for length tn into LDS (tn/4 entries; at least 1)
__device__ __forceinline__ void build_tw4(float2* tw4, int tn) {
  const int cnt = tn >= 4 ? tn / 4 : 1;
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)j / (float)tn, &s, &c);
    tw4[j] = make_float2(c, -s);
  }
}

// W_L^e = exp(-2 pi i e / L) from a two-level table: hi[e >> LOBITS] * lo[e & (2^LOBITS - 1)]
#define TW_LOBITS 8
__device__ __forceinline__ void build_tw2(float2* hi, float2* lo, int L) {
  const int nlo = L < (1 << TW_LOBITS) ? L : (1 << TW_LOBITS);
  const int nhi = L / nlo;
  for (int j = threadIdx.x; j < nlo; j += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)j / (float)L, &s, &c);
    lo[j] = make_float2(c, -s);
  }
  for (int j = threadIdx.x; j < nhi; j += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)j / (float)nhi, &s, &c);
    hi[j] = make_float2(c, -s);
  }
}
__device__ __forceinline__ float2 tw2(const float2* hi, const float2* lo, int e, int L) {
  if (L >= (1 << TW_LOBITS)) return cmul(hi[e >> TW_LOBITS], lo[e & ((1 << TW_LOBITS) - 1)]);
  return lo[e];             // L < 256: hi has the single entry 1
}

// ------------------------------------------------------------------------------------------
// Bluestein geometry + host-side table construction
// ------------------------------------------------------------------------------------------
#define BLU_TC 8      // columns per tile of the column passes
#define BLU_TR 4      // rows per block of the generic row pass
#define COL_IT 8      // load iterations issued together in the column passes

struct BluGeom {
  int n, nin, L, L1, L2;
};
static BluGeom blu_geom(int n) {
  BluGeom g;
  g.n = n;
  g.nin = (n - 1) / 2 + 1;
  int L = 16;
  while (L < n + g.nin - 1) L <<= 1;
  g.L = L;
  int p = ilog2(L);
  g.L1 = 1 << (p / 2);
  g.L2 = L / g.L1;
  return g;
}

// Rader applies when n is prime with n - 1 a power of two (Fermat primes) and 3 is a primitive root
static unsigned long long mulmod(unsigned long long a, unsigned long long b, unsigned long long m) {
  return (a * b) % m;   // operands < 2^17: no overflow
}
static unsigned long long powmod(unsigned long long b, unsigned long long e, unsigned long long m) {
  unsigned long long r = 1 % m;
  b %= m;
  while (e) {
    if (e & 1) r = mulmod(r, b, m);
    b = mulmod(b, b, m);
    e >>= 1;
  }
  return r;
}
static bool rader_ok(int n) {
  if (n < 17 || n > 65537) return false;
  const int N = n - 1;
  if (N & (N - 1)) return false;
  for (int d = 3; d * d <= n; d += 2)
    if (n % d == 0) return false;
  return powmod(3, (unsigned long long)N / 2, (unsigned long long)n) == (unsigned long long)(n - 1);
}
static BluGeom rader_geom(int n) {
  BluGeom g;
  g.n = n;
  g.nin = (n - 1) / 2 + 1;
  g.L = n - 1;
  if (g.L >= 32768) {
    g.L2 = 512;                       // the wave-per-row kernel
    g.L1 = g.L / 512;
  } else {
    const int p = ilog2(g.L);
    g.L1 = 1 << (p / 2);
    g.L2 = g.L / g.L1;
  }
  return g;
}

static void host_fft(std::vector<std::complex<double>>& a) {  // in-place radix-2, forward
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t i = 0; i < n; i += len) {
      for (size_t k = 0; k < len / 2; ++k) {
        double ang = -2.0 * M_PI * (double)k / (double)len;
        std::complex<double> w(cos(ang), sin(ang));
        std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
    }
  }
}

extern "C" size_t gfdn_bluestein_table_bytes(int n) {
  if (n < 3 || (n & 1) == 0) return 0;
  if (rader_ok(n)) return (size_t)(n - 1) * (sizeof(float2) + 2 * sizeof(int));
  BluGeom g = blu_geom(n);
  return ((size_t)g.n + (size_t)g.L) * sizeof(float2);
}

extern "C" size_t gfdn_bluestein_work_bytes(int n, int batch) {
  if (n < 3 || (n & 1) == 0 || batch <= 0) return 0;
  BluGeom g = rader_ok(n) ? rader_geom(n) : blu_geom(n);
  const int tc = g.L2 < BLU_TC ? g.L2 : BLU_TC;
  return (size_t)batch * g.L * sizeof(float2) + (size_t)batch * (g.L2 / tc) * sizeof(float);
}

// Rader table = [ chat[k1*L2 + k2] = FFT_N(v_rev)[k1 + L1 k2], v[c] = exp(2 pi i g^c / n) | perm | iperm ]
static int rader_table_init(int n, void* table) {
  BluGeom g = rader_geom(n);
  const int N = g.L;
  std::vector<int> perm(N), iperm(N);
  unsigned long long v = 1;
  for (int a = 0; a < N; ++a) {
    perm[a] = (int)v;
    v = mulmod(v, 3, (unsigned long long)n);
  }
  for (int b = 0; b < N; ++b) iperm[b] = perm[(N - b) % N];      // g^-b = g^(N-b)
  std::vector<std::complex<double>> ker(N);
  for (int c = 0; c < N; ++c) {
    const double ang = 2.0 * M_PI * (double)perm[(N - c) % N] / (double)n;   // v_rev[c] = v[-c]
    ker[c] = std::complex<double>(cos(ang), sin(ang));
  }
  host_fft(ker);
  std::vector<float2> chat(N);
  for (int k1 = 0; k1 < g.L1; ++k1)
    for (int k2 = 0; k2 < g.L2; ++k2) {
      const std::complex<double> q = ker[(size_t)k1 + (size_t)g.L1 * k2];
      chat[(size_t)k1 * g.L2 + k2] = make_float2((float)q.real(), (float)q.imag());
    }
  char* t = (char*)table;
  hipError_t e = hipMemcpy(t, chat.data(), (size_t)N * sizeof(float2), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(t + (size_t)N * sizeof(float2), perm.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(t + (size_t)N * (sizeof(float2) + sizeof(int)), iperm.data(), (size_t)N * sizeof(int),
                hipMemcpyHostToDevice);
  return (int)e;
}

// table = [ chirp w_t = exp(+i pi t^2 / n), t < n | chat[k1*L2 + k2] = FFT_L(conj chirp kernel)[k1 + L1 k2] ]
extern "C" int gfdn_bluestein_table_init(int n, void* table) {
  if (n < 3 || (n & 1) == 0 || !table) return GFDN_E_BADARG;
  if (rader_ok(n)) return rader_table_init(n, table);
  BluGeom g = blu_geom(n);
  std::vector<float2> host((size_t)g.n + g.L);
  std::vector<std::complex<double>> ker(g.L, std::complex<double>(0.0, 0.0));
  auto chirp = [&](long long m) {
    long long r = (long long)(((unsigned long long)(m * m)) % (unsigned long long)(2LL * n));
    double ang = M_PI * (double)r / (double)n;
    return std::complex<double>(cos(ang), sin(ang));
  };
  for (int t = 0; t < n; ++t) {
    std::complex<double> w = chirp(t);
    host[t] = make_float2((float)w.real(), (float)w.imag());
  }
  // kernel c_m = conj(w_m), m in [-(nin-1), n-1], placed circularly
  for (int m = -(g.nin - 1); m <= n - 1; ++m) {
    long long am = m < 0 ? -m : m;
    ker[(size_t)((m + g.L) % g.L)] = std::conj(chirp(am));
  }
  host_fft(ker);
  for (int k1 = 0; k1 < g.L1; ++k1)
    for (int k2 = 0; k2 < g.L2; ++k2) {
      std::complex<double> v = ker[(size_t)k1 + (size_t)g.L1 * k2];
      host[(size_t)g.n + (size_t)k1 * g.L2 + k2] = make_float2((float)v.real(), (float)v.imag());
    }
  hipError_t e = hipMemcpy(table, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice);
  return (int)e;
}

// ------------------------------------------------------------------------------------------
// Bluestein kernels.  Column tile: TC adjacent columns n2, LDS layout [col][row] (stride L1+1).
// ------------------------------------------------------------------------------------------

// Output stage of the block-transfer-function step folded into the first pass of the paired forward transform
// (k_blu_col128_fwd): the spectrum H[b][k] = (sum_g rgain[b][g] T_g[k] + direct[rows[b]][k]) * filt[band][k] is formed
// while the pass loads it, from the saved group transfer functions T (nbands * G, ldt) -- 7 MB that stay in the
// last-level cache -- instead of being written by one kernel and read back by the next (2 x 58.7 MB at 224 items).
struct BluCompose {
  const float2* T;        // (nbands, ldt, 4): the band's (<= 4) group transfer functions of a column side by side --
                          // one 32-byte access per slot.  NULL: plain transform (the spectrum is `in`)
  int ldt;
  const float* rgain;     // (items, G)
  const float2* filt;     // (nbands, ldf) or NULL
  int ldf, G, Bper;       // Bper: items per band
  const long long* rows;  // item b reads direct row rows[b] (NULL: row b)
  float* h0;              // (items): Re H[b][0], written by the first pass, read by the last
  // adjoint: the gains pass of the output stage's adjoint folded into the LAST pass -- partial sums of
  // dL/drgain[b][g] = sum_k Re(dL/dH[b][k] conj(filt[k] T'_g[k])), gpart[(b * G + g) * ntile + tile], ntile = L2 / 8
  float* gpart;
};

// A launch of at most this many transforms (the linear step's group signals: 14 pairs at 7 bands) is a link of the step's
// critical chain that runs beside long VALU-bound side-stream passes (colorless pass, energy pass): its waves take issue
// priority over theirs (measured at N = 32, where the colorless pass of the 8-line blocks stretched the adjoint transform's
// three passes from 30 to 92 us).  Launches of hundreds of transforms fill the chip themselves and stay at the default.
#define BLU_PRIO_BATCH 32
struct BluArgs {
  BluGeom g;
  BluCompose cmp;
  int rader;            // 0: Bluestein (chirp-z)   1: Rader (n prime, n - 1 = L)
  const int* perm;      // rader: perm[a]  = g^a  mod n   (spectrum index of convolution slot a)
  const int* iperm;     // rader: iperm[b] = g^-b mod n   (time index of convolution slot b)
  float* edge;          // rader: per-(item, column block) partial sums for the t = 0 / k = 0 terms
  int nedge;            // rader: partials per item
  int batch;
  int pair;             // slot mode only: items 2b, 2b+1 share one transform; time signals pair-interleaved (float2)
  int items;            // number of items (rows of the spectrum side)
  int slot;             // Rader + col128: the SPECTRUM side is in slot order (gfdn_irfft_odd_slot_order): no gather / scatter
  int tslots;           // pair adjoint: the TIME side arrives in slot order too -- in[0] = sample 0, in[1 + s] = sample
                        // iperm[s] (gfdn_lin_gamma writes it so): coalesced loads instead of the gather
  const float2* chirp;  // n
  const float2* chat;   // L, [k1][k2]
  float2* work;         // batch * L
  int adjoint;          // 0: X -> x ; 1: gx -> gX
  const float* in2;     // adjoint only: optional second real input, summed in on load (in + in2)
  const float* in3;     // pair adjoint in slot order only: a third input, summed in on load
  const float* oscale;  // pair forward only: per-item factors on the time signals (NULL: 1)
  // forward: in = X (complex, ld_in), out = x (real, ld_out); adjoint: in = gx (real), out = gX (complex)
  const void* in;
  int ld_in;
  void* out;
  int ld_out;
};

extern __shared__ float2 dyn_lds[];

// XCD-aware block -> (tile, item) map.  Workgroups are dealt round-robin over the 8 XCDs by linear id
// (blocks id and id + 8 share an XCD); each XCD has its own 4 MB L2.  All tiles of one item are given
// ids of one residue class mod 8, so an item's spectrum (0.26 MB), work block (0.5 MB) and output stay
// in ONE L2 across the three kernels: the Rader gather / scatter and the transposed work-block
// accesses then hit L2 instead of over-fetching from HBM (PMC: 62 MB fetched vs 17 MB compulsory
// before).  Placement only affects speed, never correctness; any batch not a multiple of 8 uses the
// plain map.
__device__ __forceinline__ void xcd_item_map(int ntiles, int batch, int& tile, int& b) {
  const int id = blockIdx.x;
  if ((batch & 7) == 0) {
    const int j = id >> 3, q = j / ntiles;
    b = (id & 7) + 8 * q;
    tile = j - q * ntiles;
  } else {
    b = id / ntiles;
    tile = id - b * ntiles;
  }
}


// input element t of the length-L work sequence of item b (chirp product / Rader gather), all four modes;
// `edge` accumulates the Rader t = 0 / k = 0 partial sums
__device__ __forceinline__ float2 blu_load_elem(const BluArgs& a, int b, int t, float& edge) {
  const BluGeom& g = a.g;
  float2 v = make_float2(0.f, 0.f);
  if (a.rader) {
    if (!a.adjoint) {              // u[a] = Y[g^a], Y the Hermitian extension of X
      const int k = a.perm[t];
      const float2* Xb = (const float2*)a.in + (size_t)b * a.ld_in;
      if (k < g.nin) { v = Xb[k]; edge += v.x; }
      else v = cconj(Xb[g.n - k]);
    } else {                       // G[b] = gx[g^-b]
      const size_t src = (size_t)b * a.ld_in + a.iperm[t];
      float r = ((const float*)a.in)[src];
      if (a.in2) r += a.in2[src];
      v = make_float2(r, 0.f);
      edge += r;
    }
  } else if (!a.adjoint) {
    if (t < g.nin) {
      float2 X = ((const float2*)a.in)[(size_t)b * a.ld_in + t];
      if (t == 0) X = make_float2(0.5f * X.x, 0.f);   // X'_0 = Re X_0 (factor 2 applied at the end)
      v = cmul(X, a.chirp[t]);
    }
  } else {
    if (t < g.n) {
      float gx = ((const float*)a.in)[(size_t)b * a.ld_in + t];
      if (a.in2) gx += a.in2[(size_t)b * a.ld_in + t];
      const float2 w = a.chirp[t];
      v = make_float2(gx * w.x, -gx * w.y);     // gx * conj(w_t)
    }
  }
  return v;
}

// output element t (value v of the inverse column pass) of item b, all four modes
__device__ __forceinline__ void blu_store_elem(const BluArgs& a, int b, int t, float2 v) {
  const BluGeom& g = a.g;
  const int L = g.L;
  if (a.rader) {
    const float invL = 1.0f / (float)L, invn = 1.0f / (float)g.n;
    if (!a.adjoint) {              // x[g^-b] = (X'_0 + S[b]) / n
      const float x0 = ((const float2*)a.in)[(size_t)b * a.ld_in].x;
      ((float*)a.out)[(size_t)b * a.ld_out + a.iperm[t]] = (x0 + v.x * invL) * invn;
    } else {                       // gX[k] = 2 conj(W[a]) + (2/n) gx[0],  k = g^a <= (n-1)/2
      const int k = a.perm[t];
      if (k < g.nin) {
        const float g0 = ((const float*)a.in)[(size_t)b * a.ld_in] + (a.in2 ? a.in2[(size_t)b * a.ld_in] : 0.f);
        const float sc = 2.0f * invL * invn;
        ((float2*)a.out)[(size_t)b * a.ld_out + k] = make_float2(sc * v.x + 2.0f * invn * g0, -sc * v.y);
      }
    }
  } else {
    const float base = 1.0f / ((float)g.n * (float)L);
    if (!a.adjoint) {
      if (t < g.n) {
        const float2 w = a.chirp[t];
        ((float*)a.out)[(size_t)b * a.ld_out + t] = 2.0f * base * (v.x * w.x - v.y * w.y);
      }
    } else {
      if (t < a.ld_out) {
        float2 o = make_float2(0.f, 0.f);
        if (t < g.nin) {
          const float2 w = a.chirp[t];
          o = cmulc(v, w);                           // conj(w_k) * conv
          if (t == 0) o = make_float2(o.x * base, 0.f);
          else o = cscale(o, 2.0f * base);
        }
        ((float2*)a.out)[(size_t)b * a.ld_out + t] = o;
      }
    }
  }
}

// Rader: t = 0 / k = 0 terms from the column blocks' partial sums (fixed order); one thread per item
__device__ __forceinline__ void blu_edge_fold(const BluArgs& a, int b) {
  const BluGeom& g = a.g;
  float sum = 0.f;
  for (int e = 0; e < a.nedge; ++e) sum += a.edge[(size_t)b * a.nedge + e];
  if (!a.adjoint) {   // x[0] = (X'_0 + 2 sum_{k>=1} Re X_k) / n
    const float x0 = ((const float2*)a.in)[(size_t)b * a.ld_in].x;
    ((float*)a.out)[(size_t)b * a.ld_out] = (x0 + 2.0f * sum) / (float)g.n;
  } else {            // gX[0] = sum_t gx[t] / n
    const float g0 = ((const float*)a.in)[(size_t)b * a.ld_in] + (a.in2 ? a.in2[(size_t)b * a.ld_in] : 0.f);
    ((float2*)a.out)[(size_t)b * a.ld_out] = make_float2((sum + g0) / (float)g.n, 0.f);
  }
}

__global__ __launch_bounds__(256) void k_blu_col_fwd(BluArgs a) {
  __shared__ float s_edge[16];
  const BluGeom g = a.g;
  const int L1 = g.L1, L2 = g.L2, L = g.L;
  const int tc = L2 < BLU_TC ? L2 : BLU_TC;
  const int ss = L1 + 1;
  float2* bufA = dyn_lds;
  float2* bufB = bufA + tc * ss;
  float2* tw4 = bufB + tc * ss;          // L1/4 (>=1)
  float2* thi = tw4 + (L1 >= 4 ? L1 / 4 : 1);
  float2* tlo = thi + (L >> TW_LOBITS > 0 ? (L >> TW_LOBITS) : 1);
  int tile, b;
  xcd_item_map(L2 / tc, a.batch, tile, b);
  const int c0 = tile * tc;
  const int ltc = 31 - __clz(tc);
  build_tw4(tw4, L1);
  build_tw2(thi, tlo, L);
  // load (+ chirp / Rader gather).  The loads of COL_IT iterations are issued together: with a
  // runtime trip count the compiler would serialise one (dependent) global load chain per
  // iteration and the kernel would sit at ~1 load latency per 256 elements.
  float edge = 0.f;                    // Rader: partial sum for the t = 0 / k = 0 terms
  for (int base = 0; base < tc * L1; base += COL_IT * 256) {
    float2 vals[COL_IT];
    int slot[COL_IT];
#pragma unroll
    for (int it = 0; it < COL_IT; ++it) {
      const int idx = base + it * 256 + threadIdx.x;
      slot[it] = -1;
      vals[it] = make_float2(0.f, 0.f);
      if (idx < tc * L1) {
        const int n1 = idx >> ltc, cc = idx & (tc - 1);
        const int t = n1 * L2 + c0 + cc;
        slot[it] = cc * ss + n1;
        vals[it] = blu_load_elem(a, b, t, edge);
      }
    }
#pragma unroll
    for (int it = 0; it < COL_IT; ++it)
      if (slot[it] >= 0) bufA[slot[it]] = vals[it];
  }
  if (a.rader) {                       // one partial per block, folded by k_blu_col_inv
    edge = block_sum(edge, s_edge);
    if (threadIdx.x == 0) a.edge[(size_t)b * a.nedge + tile] = edge;
  }
  __syncthreads();
  float2* r = lds_fft(bufA, bufB, L1, tc, ss, false, tw4, L1);
  // twiddle W_L^{n2 k1}, store [k1][n2]
  float2* wk = a.work + (size_t)b * L;
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int k1 = idx >> ltc, cc = idx & (tc - 1);
    const int n2 = c0 + cc;
    float2 v = cmul(r[cc * ss + k1], tw2(thi, tlo, (n2 * k1) & (L - 1), L));   // n2 k1 < L1 L2 = L
    wk[(size_t)k1 * L2 + n2] = v;
  }
}

__global__ __launch_bounds__(256) void k_blu_row(BluArgs a) {
  const BluGeom g = a.g;
  const int L1 = g.L1, L2 = g.L2, L = g.L;
  const int tr = L1 < BLU_TR ? L1 : BLU_TR;
  const int ss = L2;
  float2* bufA = dyn_lds;
  float2* bufB = bufA + tr * ss;
  float2* tw4 = bufB + tr * ss;
  float2* thi = tw4 + (L2 >= 4 ? L2 / 4 : 1);
  float2* tlo = thi + (L >> TW_LOBITS > 0 ? (L >> TW_LOBITS) : 1);
  int tile, b;
  xcd_item_map(L1 / tr, a.batch, tile, b);
  const int r0 = tile * tr;
  build_tw4(tw4, L2);
  build_tw2(thi, tlo, L);
  float2* wk = a.work + (size_t)b * L + (size_t)r0 * L2;
  for (int idx = threadIdx.x; idx < tr * L2; idx += blockDim.x) bufA[idx] = wk[idx];
  __syncthreads();
  float2* r = lds_fft(bufA, bufB, L2, tr, ss, false, tw4, L2);
  float2* o = (r == bufA) ? bufB : bufA;
  const float2* chat = a.chat + (size_t)r0 * L2;
  for (int idx = threadIdx.x; idx < tr * L2; idx += blockDim.x) {
    float2 ch = chat[idx];
    if (a.adjoint) {
      ch.y = -ch.y;
      if (a.rader && ((r0 + idx / L2) & 1)) ch = make_float2(-ch.x, -ch.y);   // (-1)^f, f = k1 + L1 k2
    }
    r[idx] = cmul(r[idx], ch);
  }
  __syncthreads();
  float2* r2 = lds_fft(r, o, L2, tr, ss, true, tw4, L2);
  for (int idx = threadIdx.x; idx < tr * L2; idx += blockDim.x) {
    const int rr = idx / L2, n2 = idx - rr * L2;
    const int k1 = r0 + rr;
    float2 w = tw2(thi, tlo, (int)(((long long)n2 * k1) & (L - 1)), L);
    wk[idx] = cmulc(r2[idx], w);   // * conj twiddle
  }
}

// ------------------------------------------------------------------------------------------
// Row pass for L2 = 512: one WAVEFRONT per row, 8 points per lane, radix-8 x 3.
// forward FFT -> chirp-spectrum product -> inverse FFT -> conj twiddle are chained through registers;
// between the passes the wave exchanges data through its own padded 512-point LDS region
// (index i -> i + (i >> 3): conflict-free for the stride-8 / stride-64 patterns below).  A wave's LDS
// operations execute in program order, so no barrier is needed between its own write and read.
// ------------------------------------------------------------------------------------------
#define ROW512_PAD(i) ((i) + ((i) >> 3))
#define ROW512_LDS 576   // 512 + 512/8

__device__ __forceinline__ void row512_fft(float2 (&a)[8], float2* buf, const float2* tw4, int lane,
                                           float sgn) {
  // pass 1: nn = 512, m = 64, s = 1: inputs a[k] = x[lane + 64 k]
  bfly8(a, sgn);
  {
    float2 w1 = tw4[lane];            // W_512^lane
    w1.y *= sgn;
    twiddle8(a, w1);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) buf[ROW512_PAD(8 * lane + u)] = a[u];
  // pass 2: nn = 64, m = 8, s = 8: p = lane >> 3, q = lane & 7; inputs x[q + 8 (p + 8 k)] = x[lane + 64 k]
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = buf[ROW512_PAD(lane + 64 * k)];
  bfly8(a, sgn);
  {
    float2 w1 = tw4[(lane >> 3) * 8];  // W_64^p = W_512^(8 p)
    w1.y *= sgn;
    twiddle8(a, w1);
  }
  const int base = (lane & 7) + 64 * (lane >> 3);
#pragma unroll
  for (int u = 0; u < 8; ++u) buf[ROW512_PAD(base + 8 * u)] = a[u];
  // pass 3: nn = 8, m = 1, s = 64: inputs x[lane + 64 k]; outputs X[lane + 64 u] stay in registers
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = buf[ROW512_PAD(lane + 64 * k)];
  bfly8(a, sgn);
}

__global__ __launch_bounds__(256) void k_blu_row512(BluArgs a) {
  const BluGeom g = a.g;
  if (a.batch <= BLU_PRIO_BATCH) __builtin_amdgcn_s_setprio(3);      // (see BLU_PRIO_BATCH)
  const int L2 = 512, L = g.L;
  float2* tw4 = dyn_lds;                    // 128 entries: W_512^j, j < 128
  float2* thi = tw4 + 128;
  float2* tlo = thi + ((L >> TW_LOBITS) > 0 ? (L >> TW_LOBITS) : 1);
  float2* bufs = tlo + (1 << TW_LOBITS);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  build_tw4(tw4, 512);
  build_tw2(thi, tlo, L);
  __syncthreads();
  int tile, b;
  xcd_item_map(g.L1 / 4, a.batch, tile, b);
  const int k1 = tile * 4 + wave;
  float2* buf = bufs + wave * ROW512_LDS;
  float2* wk = a.work + (size_t)b * L + (size_t)k1 * L2;
  float2 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = wk[lane + 64 * k];
  row512_fft(v, buf, tw4, lane, 1.0f);
  const float2* chat = a.chat + (size_t)k1 * L2;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float2 ch = chat[lane + 64 * u];
    if (a.adjoint) {
      ch.y = -ch.y;
      if (a.rader && (k1 & 1)) ch = make_float2(-ch.x, -ch.y);               // (-1)^f, f = k1 + L1 k2
    }
    v[u] = cmul(v[u], ch);
  }
  row512_fft(v, buf, tw4, lane, -1.0f);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int n2 = lane + 64 * u;
    const float2 w = tw2(thi, tlo, (n2 * k1) & (L - 1), L);
    wk[n2] = cmulc(v[u], w);
  }
}

__global__ __launch_bounds__(256) void k_blu_col_inv(BluArgs a) {
  const BluGeom g = a.g;
  const int L1 = g.L1, L2 = g.L2, L = g.L;
  const int tc = L2 < BLU_TC ? L2 : BLU_TC;
  const int ss = L1 + 1;
  float2* bufA = dyn_lds;
  float2* bufB = bufA + tc * ss;
  float2* tw4 = bufB + tc * ss;
  int tile, b;
  xcd_item_map(L2 / tc, a.batch, tile, b);
  const int c0 = tile * tc;
  const int ltc = 31 - __clz(tc);
  build_tw4(tw4, L1);
  const float2* wk = a.work + (size_t)b * L;
  for (int base = 0; base < tc * L1; base += COL_IT * 256) {
    float2 vals[COL_IT];
#pragma unroll
    for (int it = 0; it < COL_IT; ++it) {
      const int idx = base + it * 256 + threadIdx.x;
      if (idx < tc * L1) {
        const int k1 = idx >> ltc, cc = idx & (tc - 1);
        vals[it] = wk[(size_t)k1 * L2 + c0 + cc];
      }
    }
#pragma unroll
    for (int it = 0; it < COL_IT; ++it) {
      const int idx = base + it * 256 + threadIdx.x;
      if (idx < tc * L1) {
        const int k1 = idx >> ltc, cc = idx & (tc - 1);
        bufA[cc * ss + k1] = vals[it];
      }
    }
  }
  if (a.rader && tile == 0 && threadIdx.x == 0) blu_edge_fold(a, b);
  if (a.rader && a.adjoint) {          // bins above (n-1)/2 carry no gradient
    float2* o = (float2*)a.out + (size_t)b * a.ld_out;
    for (int k = g.nin + tile * 256 + threadIdx.x; k < a.ld_out; k += (L2 / tc) * 256)
      o[k] = make_float2(0.f, 0.f);
  }
  __syncthreads();
  float2* r = lds_fft(bufA, bufB, L1, tc, ss, true, tw4, L1);
  for (int idx = threadIdx.x; idx < tc * L1; idx += blockDim.x) {
    const int n1 = idx >> ltc, cc = idx & (tc - 1);
    blu_store_elem(a, b, n1 * L2 + c0 + cc, r[cc * ss + n1]);
  }
}


// ------------------------------------------------------------------------------------------
// Column passes for L1 = 128 (the north-star geometry 128 x 512): ONE WAVEFRONT per tile of 8 columns,
// 16 points per lane, radix-16 x radix-8 chained through registers and a wave-private LDS block -- no
// block barrier after the twiddle tables are built (the block-wide Stockham version above spends its
// time in barriers and half-idle radix-8 passes: 1024 points per 256 threads).
//   lane = 8 l + c:  c = column of the tile (fastest: global accesses are 64-byte runs, as before),
//                    l = row phase.  Pass 1 holds rows n1 = l + 8 k (k < 16) and transforms over k;
//   pass 2 holds u in {l, l + 8} x j < 8 and transforms over j:  k1 = u + 16 v.
// LDS block per wave: [c][j][u] with strides (164, 17, 1): both the pass-1 writes and the pass-2 reads
// touch 32 distinct 8-byte bank slots per half-wave.
// ------------------------------------------------------------------------------------------
#define CW_S 164
#define CW_LDS (8 * CW_S)

// v[k] = x[l + 8 k] -> out0[v'] = X[l + 16 v'], out1[v'] = X[l + 8 + 16 v']; sgn as in bfly8
__device__ __forceinline__ void col128_fft(float2 (&v)[16], float2* buf, int l, int c, float sgn) {
  bfly16(v, sgn);
  {
    float sn, cs;
    sincospif(2.0f * (float)l / 128.0f, &sn, &cs);            // W_128^l
    twiddle16(v, make_float2(cs, -sgn * sn));
  }
  float2* bc = buf + c * CW_S;
#pragma unroll
  for (int u = 0; u < 16; ++u) bc[17 * l + u] = v[u];
  // (a wave's LDS operations execute in order: no barrier between its own write and read)
  float2 y0[8], y1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { y0[j] = bc[17 * j + l]; y1[j] = bc[17 * j + l + 8]; }
  bfly8(y0, sgn);
  bfly8(y1, sgn);
#pragma unroll
  for (int q = 0; q < 8; ++q) { v[q] = y0[q]; v[8 + q] = y1[q]; }
}

__global__ __launch_bounds__(256) void k_blu_col128_fwd(BluArgs a) {
  const BluGeom g = a.g;
  if (a.batch <= BLU_PRIO_BATCH) __builtin_amdgcn_s_setprio(3);      // (see BLU_PRIO_BATCH)
  const int L2 = g.L2, L = g.L;
  float2* thi = dyn_lds;
  float2* tlo = thi + ((L >> TW_LOBITS) > 0 ? (L >> TW_LOBITS) : 1);
  float2* bufs = tlo + (1 << TW_LOBITS);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane & 7, l = lane >> 3;
  build_tw2(thi, tlo, L);
  int tb, b;
  xcd_item_map(L2 / 32, a.batch, tb, b);
  const int tile = tb * 4 + wave, c0 = tile * 8;
  float edge = 0.f, edge2 = 0.f;
  float2 v[16];
  const int b1 = a.pair ? 2 * b : b;
  const bool two = a.pair && b1 + 1 < a.items;
  if (a.pair && !a.adjoint && a.cmp.T) {
    // as the branch below, with the two spectra formed on the fly (BluCompose): same operations in the same order as
    // k_tf_compose_fwd
    const BluCompose& cm = a.cmp;
    const int band = b1 / cm.Bper, G = cm.G;
    const float2* D1 = (const float2*)a.in + (size_t)(cm.rows ? cm.rows[b1] : b1) * a.ld_in;
    const float2* D2 = two ? (const float2*)a.in + (size_t)(cm.rows ? cm.rows[b1 + 1] : b1 + 1) * a.ld_in : D1;
    const float4* T0 = (const float4*)(cm.T + (size_t)band * cm.ldt * 4);
    const float2* F = cm.filt ? cm.filt + (size_t)band * cm.ldf : nullptr;
    float rg1[4], rg2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      rg1[g] = g < G ? cm.rgain[(size_t)b1 * G + g] : 0.f;
      rg2[g] = (g < G && two) ? cm.rgain[(size_t)(b1 + 1) * G + g] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t col = (size_t)(l + 8 * k) * L2 + c0 + c + 1;
      float2 p = D1[col], q = two ? D2[col] : make_float2(0.f, 0.f);
      const float4 t01 = T0[2 * col], t23 = T0[2 * col + 1];
      const float2 tg[4] = {make_float2(t01.x, t01.y), make_float2(t01.z, t01.w), make_float2(t23.x, t23.y),
                            make_float2(t23.z, t23.w)};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < G) {
          p.x += rg1[g] * tg[g].x;
          p.y += rg1[g] * tg[g].y;
          q.x += rg2[g] * tg[g].x;
          q.y += rg2[g] * tg[g].y;
        }
      }
      if (F) {
        const float2 f = F[col];
        p = cmul(p, f);
        q = cmul(q, f);
      }
      if (!two) q = make_float2(0.f, 0.f);
      v[k] = make_float2(p.x - q.y, p.y + q.x);
      v[k + 8] = make_float2(p.x + q.y, q.x - p.y);
      edge += p.x;
      edge2 += q.x;
    }
    if (tile == 0 && lane == 0) {            // bin 0 of both items (the last pass adds it to every sample)
#pragma unroll 1
      for (int it = 0; it < (two ? 2 : 1); ++it) {
        const float2* D = it ? D2 : D1;
        float2 h = D[0];
        const float2* tq = (const float2*)T0;
        for (int g = 0; g < G; ++g) {
          const float rg = cm.rgain[(size_t)(b1 + it) * G + g];
          const float2 t = tq[g];
          h.x += rg * t.x;
          h.y += rg * t.y;
        }
        if (F) h = cmul(h, F[0]);
        cm.h0[b1 + it] = h.x;
      }
    }
  } else if (a.pair && !a.adjoint) {
    // two slot-ordered spectra ride one transform as u1 + i u2 (the results are real: they come back as the
    // real / imaginary part -- see k_blu_col128_inv); rows n1 and n1 + 64 hold conj(u1) + i conj(u2)
    const float2* X1 = (const float2*)a.in + (size_t)b1 * a.ld_in + 1;
    const float2* X2 = X1 + a.ld_in;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t idx = (size_t)(l + 8 * k) * L2 + c0 + c;
      const float2 p = X1[idx], q = two ? X2[idx] : make_float2(0.f, 0.f);
      v[k] = make_float2(p.x - q.y, p.y + q.x);
      v[k + 8] = make_float2(p.x + q.y, q.x - p.y);
      edge += p.x;
      edge2 += q.x;
    }
  } else if (a.pair) {
    // adjoint: the pair-interleaved real gradients ARE the complex input G1 + i G2
    const float2* G = (const float2*)a.in + (size_t)b * a.ld_in;
    const float2* G2 = a.in2 ? (const float2*)a.in2 + (size_t)b * a.ld_in : nullptr;
    const float2* G3 = a.in3 ? (const float2*)a.in3 + (size_t)b * a.ld_in : nullptr;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int sidx = (l + 8 * k) * L2 + c0 + c;
      const int src = a.tslots ? 1 + sidx : a.iperm[sidx];
      float2 gv = G[src];
      if (G2) { const float2 g2 = G2[src]; gv.x += g2.x; gv.y += g2.y; }
      if (G3) { const float2 g3 = G3[src]; gv.x += g3.x; gv.y += g3.y; }
      v[k] = gv;
      edge += gv.x;
      edge2 += gv.y;
    }
  } else if (a.slot && !a.adjoint) {
    // spectrum in slot order: X[0] = bin 0, X[1 + s] = u[s] for s < L/2, and u[s + L/2] = conj(u[s]) -- rows
    // n1 and n1 + 64 of a column are conjugates, both held by this lane: 8 coalesced loads, no gather
    const float2* Xs = (const float2*)a.in + (size_t)b * a.ld_in + 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = Xs[(size_t)(l + 8 * k) * L2 + c0 + c];
      v[k + 8] = cconj(v[k]);
      edge += v[k].x;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = blu_load_elem(a, b, (l + 8 * k) * L2 + c0 + c, edge);
  }
  if (a.rader) {                       // one partial per (item, tile), folded by the inverse column pass
    edge = wave_sum(edge);
    if (lane == 0) a.edge[(size_t)b1 * a.nedge + tile] = edge;
    if (two) {
      edge2 = wave_sum(edge2);
      if (lane == 0) a.edge[(size_t)(b1 + 1) * a.nedge + tile] = edge2;
    }
  }
  __syncthreads();                     // twiddle tables
  col128_fft(v, bufs + wave * CW_LDS, l, c, 1.0f);
  // twiddle W_L^{n2 k1}, store [k1][n2]
  float2* wk = a.work + (size_t)b * L;
  const int n2 = c0 + c;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k1 = l + 8 * (q >> 3) + 16 * (q & 7);
    wk[(size_t)k1 * L2 + n2] = cmul(v[q], tw2(thi, tlo, (n2 * k1) & (L - 1), L));
  }
}

__global__ __launch_bounds__(256) void k_blu_col128_inv(BluArgs a) {
  const BluGeom g = a.g;
  if (a.batch <= BLU_PRIO_BATCH) __builtin_amdgcn_s_setprio(3);      // (see BLU_PRIO_BATCH)
  const int L2 = g.L2, L = g.L;
  float2* bufs = dyn_lds;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane & 7, l = lane >> 3;
  int tb, b;
  xcd_item_map(L2 / 32, a.batch, tb, b);
  const int tile = tb * 4 + wave, c0 = tile * 8;
  const float2* wk = a.work + (size_t)b * L;
  float2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = wk[(size_t)(l + 8 * k) * L2 + c0 + c];
  if (a.pair) {
    const int b1 = 2 * b;
    const bool two = b1 + 1 < a.items;
    const float invL = 1.0f / (float)L, invn = 1.0f / (float)g.n;
    float sum1 = 0.f, sum2 = 0.f;
    const bool folder = tb == 0 && threadIdx.x == 0;
    if (tb == 0 && wave == 0) {
      // t = 0 / k = 0 terms: the tiles' partial sums folded by the whole wave in a fixed order (one thread walking the
      // 64 partials was a serial chain of dependent loads -- invisible beside 112 pairs, the critical path of a launch
      // that transforms the 14 pairs of group signals)
      float e1 = 0.f, e2 = 0.f;
      for (int e = lane; e < a.nedge; e += 64) {
        e1 += a.edge[(size_t)b1 * a.nedge + e];
        if (two) e2 += a.edge[(size_t)(b1 + 1) * a.nedge + e];
      }
      sum1 = wave_sum(e1);
      sum2 = wave_sum(e2);
    }
    col128_fft(v, bufs + wave * CW_LDS, l, c, -1.0f);
    if (!a.adjoint) {
      // the two real results are the real / imaginary part: ONE 8-byte scatter per slot serves both items
      const float2* X = (const float2*)a.in;
      const float x01 = a.cmp.T ? a.cmp.h0[b1] : X[(size_t)b1 * a.ld_in].x;
      const float x02 = two ? (a.cmp.T ? a.cmp.h0[b1 + 1] : X[(size_t)(b1 + 1) * a.ld_in].x) : 0.f;
      float2* xo = (float2*)a.out + (size_t)b * a.ld_out;
      // (per-item factors: the band bank's normalisation scale joins its group signals here, off the chain that produced
      // the spectra -- bankstep.py)
      const float n1s = a.oscale ? invn * a.oscale[b1] : invn, n2s = (a.oscale && two) ? invn * a.oscale[b1 + 1] : invn;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n1 = l + 8 * (q >> 3) + 16 * (q & 7);
        xo[a.iperm[n1 * L2 + c0 + c]] = make_float2((x01 + v[q].x * invL) * n1s, (x02 + v[q].y * invL) * n2s);
      }
      if (folder) xo[0] = make_float2((x01 + 2.0f * sum1) * n1s, (x02 + 2.0f * sum2) * n2s);
    } else {
      // W = W1 + i W2 with W_j[s + L/2] = conj(W_j[s]) (real inputs, kernel symmetry): the two gradients
      // separate from the slot pair (s, s + L/2) = rows (n1, n1 + 64), held by this lane (q, q ^ 4)
      const float2* G = (const float2*)a.in + (size_t)b * a.ld_in;
      float2 g0 = G[0];
      if (a.in2) { const float2 t2 = ((const float2*)a.in2 + (size_t)b * a.ld_in)[0]; g0.x += t2.x; g0.y += t2.y; }
      if (a.in3) { const float2 t3 = ((const float2*)a.in3 + (size_t)b * a.ld_in)[0]; g0.x += t3.x; g0.y += t3.y; }
      const float sc = 2.0f * invL * invn;
      float2* o1 = (float2*)a.out + (size_t)b1 * a.ld_out;
      float2* o2 = o1 + a.ld_out;
      // gains pass of the output stage's adjoint, folded in (BluCompose): every gradient value is used while it is in
      // a register -- dL/drgain[b][g] += Re(dL/dH[b][k] conj(filt[k] T'_g[k])) -- instead of being read back by a
      // launch of its own beside the records pass
      const BluCompose& cm = a.cmp;
      const bool gains = cm.T != nullptr;
      const int band = gains ? b1 / cm.Bper : 0;
      const float4* TQ = gains ? (const float4*)(cm.T + (size_t)band * cm.ldt * 4) : nullptr;
      const float2* F = (gains && cm.filt) ? cm.filt + (size_t)band * cm.ldf : nullptr;
      float ga1[4] = {0.f, 0.f, 0.f, 0.f}, ga2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n1 = l + 8 * (q >> 3) + 16 * (q & 7);
        if (n1 < 64) {
          const float2 p = v[q ^ 4];
          const float2 W1 = make_float2(0.5f * (v[q].x + p.x), 0.5f * (v[q].y - p.y));
          const float2 W2 = make_float2(0.5f * (v[q].y + p.y), -0.5f * (v[q].x - p.x));
          const size_t idx = 1 + (size_t)n1 * L2 + c0 + c;
          const float2 h1 = make_float2(sc * W1.x + 2.0f * invn * g0.x, -sc * W1.y);
          const float2 h2 = make_float2(sc * W2.x + 2.0f * invn * g0.y, -sc * W2.y);
          o1[idx] = h1;
          if (two) o2[idx] = h2;
          if (gains) {
            const float4 t01 = TQ[2 * idx], t23 = TQ[2 * idx + 1];
            const float2 fk = F ? F[idx] : make_float2(1.f, 0.f);
            const float2 w0 = cmul(fk, make_float2(t01.x, t01.y)), w1 = cmul(fk, make_float2(t01.z, t01.w));
            const float2 w2 = cmul(fk, make_float2(t23.x, t23.y)), w3 = cmul(fk, make_float2(t23.z, t23.w));
            ga1[0] += h1.x * w0.x + h1.y * w0.y;
            ga1[1] += h1.x * w1.x + h1.y * w1.y;
            ga1[2] += h1.x * w2.x + h1.y * w2.y;
            ga1[3] += h1.x * w3.x + h1.y * w3.y;
            ga2[0] += h2.x * w0.x + h2.y * w0.y;
            ga2[1] += h2.x * w1.x + h2.y * w1.y;
            ga2[2] += h2.x * w2.x + h2.y * w2.y;
            ga2[3] += h2.x * w3.x + h2.y * w3.y;
          }
        }
      }
      if (folder) {
        const float2 h1 = make_float2((sum1 + g0.x) * invn, 0.f), h2 = make_float2((sum2 + g0.y) * invn, 0.f);
        o1[0] = h1;
        if (two) o2[0] = h2;
        if (gains) {                       // bin 0 (column 0): h is real
          const float2* tq = (const float2*)TQ;
          const float2 fk = F ? F[0] : make_float2(1.f, 0.f);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float2 w = cmul(fk, tq[g]);
            ga1[g] += h1.x * w.x;
            ga2[g] += h2.x * w.x;
          }
        }
      }
      if (gains) {
        const int ntile = L2 / 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float s1 = wave_sum(ga1[g]), s2 = wave_sum(ga2[g]);
          if (lane == 0 && g < cm.G) {
            cm.gpart[((size_t)b1 * cm.G + g) * ntile + tile] = s1;
            if (two) cm.gpart[((size_t)(b1 + 1) * cm.G + g) * ntile + tile] = s2;
          }
        }
      }
    }
    return;
  }
  if (a.rader && tb == 0 && threadIdx.x == 0) blu_edge_fold(a, b);
  if (a.rader && a.adjoint) {          // bins above (n-1)/2 carry no gradient
    float2* o = (float2*)a.out + (size_t)b * a.ld_out;
    for (int k = g.nin + tb * 256 + threadIdx.x; k < a.ld_out; k += (L2 / 32) * 256)
      o[k] = make_float2(0.f, 0.f);
  }
  col128_fft(v, bufs + wave * CW_LDS, l, c, -1.0f);
  if (a.slot && a.adjoint) {
    // gradient w.r.t. the slot-ordered spectrum: dL/du[s] for s < L/2 (rows n1 < 64) takes the same form
    // 2 conj(W[s]) / (n L) + (2 / n) gx[0] whether the slot holds X[k] or conj(X[n - k]) (the partner slot's
    // term is the conjugate, W[s + L/2] = conj(W[s])): coalesced stores, no permutation table
    const float invL = 1.0f / (float)L, invn = 1.0f / (float)g.n;
    const float g0 = ((const float*)a.in)[(size_t)b * a.ld_in] + (a.in2 ? a.in2[(size_t)b * a.ld_in] : 0.f);
    const float sc = 2.0f * invL * invn;
    float2* o = (float2*)a.out + (size_t)b * a.ld_out + 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n1 = l + 8 * (q >> 3) + 16 * (q & 7);
      if (n1 < 64) o[(size_t)n1 * L2 + c0 + c] = make_float2(sc * v[q].x + 2.0f * invn * g0, -sc * v[q].y);
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n1 = l + 8 * (q >> 3) + 16 * (q & 7);
    blu_store_elem(a, b, n1 * L2 + c0 + c, v[q]);
  }
}

static size_t blu_col_lds(const BluGeom& g) {
  int tc = g.L2 < BLU_TC ? g.L2 : BLU_TC;
  size_t e = (size_t)2 * tc * (g.L1 + 1) + (g.L1 >= 4 ? g.L1 / 4 : 1) +
             ((g.L >> TW_LOBITS) > 0 ? (g.L >> TW_LOBITS) : 1) + (1 << TW_LOBITS);
  return e * sizeof(float2);
}
static size_t blu_row_lds(const BluGeom& g) {
  int tr = g.L1 < BLU_TR ? g.L1 : BLU_TR;
  size_t e = (size_t)2 * tr * g.L2 + (g.L2 >= 4 ? g.L2 / 4 : 1) +
             ((g.L >> TW_LOBITS) > 0 ? (g.L >> TW_LOBITS) : 1) + (1 << TW_LOBITS);
  return e * sizeof(float2);
}

static bool slot_order_ok(int n) {
  if (!rader_ok(n)) return false;
  const BluGeom g = rader_geom(n);
  return g.L1 == 128 && g.L2 == 512;
}

static int blu_run(const void* table, int n, const void* in, int ld_in, int batch, void* out,
                   int ld_out, void* work, int adjoint, hipStream_t s, int stages = 7,
                   const float* in2 = nullptr, int slot = 0, const BluCompose* cmp = nullptr, int tslots = 0,
                   const float* in3 = nullptr, const float* oscale = nullptr) {
  if (!table || !in || !out || !work) return GFDN_E_BADARG;
  if (n < 3 || (n & 1) == 0 || batch <= 0) return GFDN_E_BADARG;
  const bool rader = rader_ok(n);
  BluGeom g = rader ? rader_geom(n) : blu_geom(n);
  if (g.L1 > 1024 || g.L2 > 2048) return GFDN_E_UNSUPPORTED;
  if (!adjoint && (ld_in < g.nin || ld_out < n)) return GFDN_E_BADARG;
  if (adjoint && (ld_in < n || ld_out < g.nin || (!rader && ld_out > g.L))) return GFDN_E_BADARG;
  if (slot && adjoint && ld_out != g.nin) return GFDN_E_BADARG;
  BluArgs a;
  a.g = g;
  a.cmp = cmp ? *cmp : BluCompose{nullptr, 0, nullptr, nullptr, 0, 0, 1, nullptr, nullptr, nullptr};
  a.rader = rader ? 1 : 0;
  if (rader) {
    a.chirp = nullptr;
    a.chat = (const float2*)table;
    a.perm = (const int*)((const char*)table + (size_t)g.L * sizeof(float2));
    a.iperm = a.perm + g.L;
  } else {
    a.perm = a.iperm = nullptr;
    a.chirp = (const float2*)table;
    a.chat = a.chirp + g.n;
  }
  a.work = (float2*)work;
  const int tc0 = g.L2 < BLU_TC ? g.L2 : BLU_TC;
  a.nedge = g.L2 / tc0;
  a.edge = (float*)((char*)work + (size_t)batch * g.L * sizeof(float2));   // after the work blocks
  a.batch = batch;
  a.items = batch;
  a.slot = slot ? 1 : 0;
  a.pair = slot == 2 ? 1 : 0;
  a.tslots = tslots ? 1 : 0;
  if (tslots && !(slot == 2 && adjoint)) return GFDN_E_BADARG;
  if ((in3 && !(tslots && in2)) || (oscale && !(slot == 2 && !adjoint))) return GFDN_E_BADARG;
  if (slot && !slot_order_ok(n)) return GFDN_E_UNSUPPORTED;
  if (a.pair) a.batch = (batch + 1) / 2;          // work blocks = item pairs
  a.adjoint = adjoint;
  a.in = in;
  a.in2 = adjoint ? in2 : nullptr;
  a.in3 = adjoint ? in3 : nullptr;
  a.oscale = oscale;
  a.ld_in = ld_in;
  a.out = out;
  a.ld_out = ld_out;
  const int tc = g.L2 < BLU_TC ? g.L2 : BLU_TC;
  const int tr = g.L1 < BLU_TR ? g.L1 : BLU_TR;
  size_t lc = blu_col_lds(g), lr = blu_row_lds(g);
  if (lc > 160 * 1024 || lr > 160 * 1024) return GFDN_E_UNSUPPORTED;
  int rc;
  if ((rc = ensure_dyn_lds(k_blu_col_fwd, lc))) return rc;
  if ((rc = ensure_dyn_lds(k_blu_row, lr))) return rc;
  if ((rc = ensure_dyn_lds(k_blu_col_inv, lc))) return rc;
  const bool col128 = g.L1 == 128 && g.L2 % 32 == 0 && tc == 8;       // wave-per-tile column kernels
  if (cmp && !(col128 && a.pair && rader)) return GFDN_E_UNSUPPORTED;
  const size_t tw2_elems = (size_t)((g.L >> TW_LOBITS) > 0 ? (g.L >> TW_LOBITS) : 1) + (1 << TW_LOBITS);
  if (stages & 1) {
    if (col128)
      hipLaunchKernelGGL(k_blu_col128_fwd, dim3((g.L2 / 32) * a.batch), dim3(256),
                         (tw2_elems + 4 * CW_LDS) * sizeof(float2), s, a);
    else
      hipLaunchKernelGGL(k_blu_col_fwd, dim3((g.L2 / tc) * batch), dim3(256), lc, s, a);
    GFDN_LAUNCH_CHECK();
  }
  if (stages & 2) {
    if (g.L2 == 512 && g.L1 % 4 == 0) {
      const size_t lr5 = (128 + ((g.L >> TW_LOBITS) > 0 ? (g.L >> TW_LOBITS) : 1) + (1 << TW_LOBITS) +
                          4 * ROW512_LDS) * sizeof(float2);
      hipLaunchKernelGGL(k_blu_row512, dim3((g.L1 / 4) * a.batch), dim3(256), lr5, s, a);
    } else {
      hipLaunchKernelGGL(k_blu_row, dim3((g.L1 / tr) * batch), dim3(256), lr, s, a);
    }
    GFDN_LAUNCH_CHECK();
  }
  if (stages & 4) {
    if (col128) {
      // The pair forward pass SCATTERS 8-byte slots over the pair's whole output (512 KB): lines leave L2 partly
      // written when too many pairs are in flight per XCD (PMC: 4.8 x the algorithmic write bytes).  Twice the
      // LDS per block caps a CU at two blocks = four pairs per XCD: 70 -> 60 us (one block per CU: 62 us;
      // non-temporal loads of the work block: no gain).  The adjoint's stores are coalesced and keep 3 blocks.
      const size_t linv = (size_t)((a.pair && !a.adjoint) ? 2 : 1) * 4 * CW_LDS * sizeof(float2);
      if ((rc = ensure_dyn_lds(k_blu_col128_inv, linv))) return rc;
      hipLaunchKernelGGL(k_blu_col128_inv, dim3((g.L2 / 32) * a.batch), dim3(256), linv, s, a);
    }
    else
      hipLaunchKernelGGL(k_blu_col_inv, dim3((g.L2 / tc) * batch), dim3(256), lc, s, a);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gfdn_irfft_odd_stages(const void* table, int n, const void* in, const float* in2,
                                     int ld_in, int batch, void* out, int ld_out, void* work,
                                     int adjoint, int stages, int slots, void* stream) {
  return blu_run(table, n, in, ld_in, batch, out, ld_out, work, adjoint, (hipStream_t)stream, stages, in2, slots);
}

// Slot order of the spectrum side (Rader, n = 65 537): slot s < (n-1)/2 holds u[s] = X[k] (conj = 0) or
// conj(X[k]) (conj = 1) with k = bins[s] in 1..(n-1)/2, where 3^s mod n is k or n - k.  A caller that evaluates
// its spectrum directly on the grid points { z_k or conj(z_k) } (pointwise models do) hands X[0] = bin 0,
// X[1 + s] = u[s] to the *_slots entry points and the transform needs no gather; the adjoint returns the
// gradient in the same order.
extern "C" int gfdn_irfft_odd_slot_order(int n, int* bins, int* conj) {
  if (!bins || !conj) return GFDN_E_BADARG;
  if (n < 3 || !slot_order_ok(n)) return GFDN_E_UNSUPPORTED;
  const int half = (n - 1) / 2;
  unsigned long long v = 1;
  for (int s = 0; s < half; ++s) {
    const int k = (int)v;
    bins[s] = k <= half ? k : n - k;
    conj[s] = k <= half ? 0 : 1;
    v = mulmod(v, 3, (unsigned long long)n);
  }
  return 0;
}
extern "C" int gfdn_irfft_odd_slots_fwd(const void* table, int n, const float* Xs, int ldx, int batch,
                                        float* x, int ldo, void* work, void* stream) {
  return blu_run(table, n, Xs, ldx, batch, x, ldo, work, 0, (hipStream_t)stream, 7, nullptr, 1);
}
extern "C" int gfdn_irfft_odd_slots_bwd(const void* table, int n, const float* gx, const float* gx2, int ldo,
                                        int batch, float* gXs, int ldx, void* work, void* stream) {
  return blu_run(table, n, gx, ldo, batch, gXs, ldx, work, 1, (hipStream_t)stream, 7, gx2, 1);
}

// Pairs: slot-ordered spectra in, pair-interleaved time signals out (x2: (ceil(batch / 2), ldo) float2, item 2p
// in .x, item 2p + 1 in .y) and the adjoint of that -- two items per complex transform.
extern "C" int gfdn_irfft_odd_pairs_fwd(const void* table, int n, const float* Xs, int ldx, int batch,
                                        float* x2, int ldo, void* work, void* stream) {
  return blu_run(table, n, Xs, ldx, batch, x2, ldo, work, 0, (hipStream_t)stream, 7, nullptr, 2);
}
// ... with per-item factors on the time signals (oscale: batch floats; the factor of a signal multiplies all its samples)
// stages: 7 = the whole transform; 3 = the first two passes, 4 = the last one (the only one that reads oscale: a caller whose
// factors come from another stream waits for them between the two calls)
extern "C" int gfdn_irfft_odd_pairs_fwd_scaled(const void* table, int n, const float* Xs, int ldx, int batch,
                                               const float* oscale, float* x2, int ldo, void* work, int stages,
                                               void* stream) {
  if (!(stages & 7)) return GFDN_E_BADARG;
  return blu_run(table, n, Xs, ldx, batch, x2, ldo, work, 0, (hipStream_t)stream, stages & 7, nullptr, 2, nullptr, 0, nullptr,
                 oscale);
}
// The paired forward transform with the output stage of the block-transfer-function step folded into its first pass:
// x2 = irfft_n(H) for H[b][k] = (sum_g rgain[b][g] T[band * G + g][k] + direct[rows[b]][k]) * filt[band][k], items
// band-major (batch = nbands * Bper), every array slot-ordered with column 0 = bin 0; H itself is never stored.
// T (nbands, ldt, 4): gfdn_tf_compose_fwd's Tquad.  stages: 7 = the whole transform (bit 0 / 1 / 2: first / middle /
// last pass alone, for timing a pass between events).  h0: `batch` floats of scratch that must stay untouched until the
// call's kernels have run.
extern "C" int gfdn_irfft_odd_pairs_compose_fwd(const void* table, int n, const float* direct_c64, int ldd,
                                                const long long* direct_rows, const float* T_c64, int ldt,
                                                const float* rgain, const float* filt_c64, int ldf, int nbands, int G,
                                                int batch, float* h0, float* x2, int ldo, void* work, int stages,
                                                void* stream) {
  if (!direct_c64 || !T_c64 || !rgain || !h0 || nbands <= 0 || G <= 0 || batch <= 0 || batch % nbands) return GFDN_E_BADARG;
  if (G > 4) return GFDN_E_UNSUPPORTED;
  const int half = (n - 1) / 2 + 1;
  if (ldd < half || ldt < half || (filt_c64 && ldf < half)) return GFDN_E_BADARG;
  if ((batch / nbands) % 2) return GFDN_E_UNSUPPORTED;          // (a pair must not straddle two bands)
  BluCompose cm{(const float2*)T_c64, ldt, rgain, (const float2*)filt_c64, ldf, G, batch / nbands, direct_rows, h0, nullptr};
  if (!(stages & 7)) return GFDN_E_BADARG;
  return blu_run(table, n, direct_c64, ldd, batch, x2, ldo, work, 0, (hipStream_t)stream, stages & 7, nullptr, 2, &cm);
}
extern "C" int gfdn_irfft_odd_pairs_bwd(const void* table, int n, const float* gx2, const float* gx2b, int ldo,
                                        int batch, float* gXs, int ldx, void* work, void* stream) {
  return blu_run(table, n, gx2, ldo, batch, gXs, ldx, work, 1, (hipStream_t)stream, 7, gx2b, 2);
}

// ... with the time side in slot order as well: gx2s[p][0] = the pair's sample 0, gx2s[p][1 + s] = its sample at time
// gfdn_irfft_odd_time_slots()[s] -- what gfdn_lin_gamma writes when it is handed the inverse of that table; the first pass
// then loads coalesced rows instead of gathering 8-byte samples
extern "C" int gfdn_irfft_odd_pairs_bwd_tslots(const void* table, int n, const float* gx2s, int ldo, int batch, float* gXs,
                                               int ldx, void* work, void* stream) {
  return blu_run(table, n, gx2s, ldo, batch, gXs, ldx, work, 1, (hipStream_t)stream, 7, nullptr, 2, nullptr, 1);
}
// ... as the SUM of up to three slot-ordered inputs (b, c optional; c only with b), added where the first pass loads them:
// parts of one gradient signal that were produced by different launches need no merge pass in front of the transform
extern "C" int gfdn_irfft_odd_pairs_bwd_tslots3(const void* table, int n, const float* gx2s_a, const float* gx2s_b,
                                                const float* gx2s_c, int ldo, int batch, float* gXs, int ldx, void* work,
                                                void* stream) {
  return blu_run(table, n, gx2s_a, ldo, batch, gXs, ldx, work, 1, (hipStream_t)stream, 7, gx2s_b, 2, nullptr, 1, gx2s_c);
}
// time index of convolution slot s (s < n - 1): times[s] = g^-s mod n, the table the pair transforms scatter / gather by
extern "C" int gfdn_irfft_odd_time_slots(int n, int* times) {
  if (!times) return GFDN_E_BADARG;
  if (n < 3 || !slot_order_ok(n)) return GFDN_E_UNSUPPORTED;
  const int N = n - 1;
  unsigned long long v = 1;
  for (int a = 0; a < N; ++a) {            // g^a at slot (N - a) % N
    times[(N - a) % N] = (int)v;
    v = mulmod(v, 3, (unsigned long long)n);
  }
  return 0;
}

// The paired adjoint transform with the GAINS pass of the output stage's adjoint folded into its last pass: besides
// gXs = dL/dH (as gfdn_irfft_odd_pairs_bwd) it leaves the partial sums of dL/drgain[b][g] = sum_k Re(dL/dH[b][k]
// conj(filt[band][k] T'[band][k][g])) in gpart[(b * G + g) * parts + p], parts = gfdn_irfft_odd_pairs_gains_parts(n)
// (sum the rows in a fixed order: gfdn_tf_rows_sum).  T (nbands, ldt, 4): gfdn_tf_compose_fwd's Tquad.
extern "C" int gfdn_irfft_odd_pairs_gains_parts(int n) {
  if (n < 3 || !rader_ok(n)) return 0;
  return rader_geom(n).L2 / 8;
}
extern "C" int gfdn_irfft_odd_pairs_gains_bwd(const void* table, int n, const float* gx2, const float* gx2b, int ldo,
                                              int batch, float* gXs, int ldx, const float* T_c64, int ldt,
                                              const float* filt_c64, int ldf, int nbands, int G, float* gpart,
                                              void* work, int stages, void* stream) {
  if (!T_c64 || !gpart || nbands <= 0 || G <= 0 || batch <= 0 || batch % nbands || !(stages & 7)) return GFDN_E_BADARG;
  if (G > 4 || (batch / nbands) % 2) return GFDN_E_UNSUPPORTED;
  const int half = (n - 1) / 2 + 1;
  if (ldt < half || (filt_c64 && ldf < half)) return GFDN_E_BADARG;
  BluCompose cm{(const float2*)T_c64, ldt, nullptr, (const float2*)filt_c64, ldf, G, batch / nbands, nullptr, nullptr, gpart};
  return blu_run(table, n, gx2, ldo, batch, gXs, ldx, work, 1, (hipStream_t)stream, stages & 7, gx2b, 2, &cm);
}

extern "C" int gfdn_irfft_odd_fwd(const void* table, int n, const float* X, int ldx, int batch,
                                  float* x, int ldo, void* work, void* stream) {
  return blu_run(table, n, X, ldx, batch, x, ldo, work, 0, (hipStream_t)stream);
}
extern "C" int gfdn_irfft_odd_bwd(const void* table, int n, const float* gx, const float* gx2, int ldo,
                                  int batch, float* gX, int ldx, void* work, void* stream) {
  return blu_run(table, n, gx, ldo, batch, gX, ldx, work, 1, (hipStream_t)stream, 7, gx2);
}

// ------------------------------------------------------------------------------------------
// |STFT|^2, periodic Hann, hop = win/2, center = False; two frames per complex FFT.
// ------------------------------------------------------------------------------------------
extern "C" int gfdn_stft_nframes(int T, int win) {
  if (T <= 0 || win < 4 || (win & (win - 1))) return GFDN_E_BADARG;
  const int hop = win / 2;
  const int Tp = ((T + hop - 1) / hop) * hop;
  if (Tp < win) return 0;
  return (Tp - win) / hop + 1;
}

__device__ __forceinline__ float hann_periodic(int j, int W) {
  return 0.5f - 0.5f * cospif(2.0f * (float)j / (float)W);
}

#ifndef STFT_T
#define STFT_T 512   // threads per frame pair (win 4096: one radix-8 butterfly per thread per pass)
#endif
// loads frames (m, m+1) of signal b into z = fr_m + i fr_{m+1}
__device__ __forceinline__ void stft_load_pair(const float* __restrict__ x, int T, int W, int m,
                                               int nframes, float2* buf) {
  const int hop = W >> 1;
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    const float h = hann_periodic(j, W);
    const int ta = m * hop + j, tb = ta + hop;
    const float va = ta < T ? x[ta] : 0.f;
    const float vb = (m + 1 < nframes && tb < T) ? x[tb] : 0.f;
    buf[j] = make_float2(h * va, h * vb);
  }
}

__global__ __launch_bounds__(STFT_T) void k_stft_power(const float* __restrict__ x, int ld, int T,
                                                    int W, int nframes, float* __restrict__ P,
                                                    float* __restrict__ zero_buf) {
  float2* bufA = dyn_lds;
  float2* bufB = bufA + W;
  float2* tw4 = bufB + W;
  const int b = blockIdx.y, m = blockIdx.x * 2, nf = W / 2 + 1;
  if (zero_buf) {      // clear the adjoint's accumulation buffer: this block's W samples, the last block the tail
    float* zb = zero_buf + (size_t)b * ld;
    const int t0 = m * (W >> 1);
    const int t1 = (blockIdx.x == gridDim.x - 1) ? ld : t0 + W;
    for (int t = t0 + threadIdx.x; t < t1 && t < ld; t += blockDim.x) zb[t] = 0.f;
  }
  build_tw4(tw4, W);
  stft_load_pair(x + (size_t)b * ld, T, W, m, nframes, bufA);
  __syncthreads();
  const float2* Z = lds_fft(bufA, bufB, W, 1, W, false, tw4, W);
  float* Pa = P + ((size_t)b * nframes + m) * nf;
  const bool has_b = m + 1 < nframes;
  for (int f = threadIdx.x; f < nf; f += blockDim.x) {
    const float2 zf = Z[f], zc = Z[(W - f) & (W - 1)];
    // S_a = (Z_f + conj Z_{W-f})/2 ; S_b = (Z_f - conj Z_{W-f})/(2i)
    const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
    const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
    Pa[f] = sa.x * sa.x + sa.y * sa.y;
    if (has_b) Pa[nf + f] = sb.x * sb.x + sb.y * sb.y;
  }
}

__global__ __launch_bounds__(STFT_T) void k_stft_power_bwd(const float* __restrict__ x, int ld, int T,
                                                        int W, int nframes,
                                                        const float* __restrict__ gP,
                                                        float* __restrict__ gx) {
  float2* bufA = dyn_lds;
  float2* bufB = bufA + W;
  float2* tw4 = bufB + W;
  const int b = blockIdx.y, m = blockIdx.x * 2, nf = W / 2 + 1, hop = W >> 1;
  build_tw4(tw4, W);
  stft_load_pair(x + (size_t)b * ld, T, W, m, nframes, bufA);
  __syncthreads();
  float2* Z = lds_fft(bufA, bufB, W, 1, W, false, tw4, W);
  float2* O = (Z == bufA) ? bufB : bufA;
  const float* ga = gP + ((size_t)b * nframes + m) * nf;
  const bool has_b = m + 1 < nframes;
  // U = Ga_sym + i Gb_sym, built pairwise (f, W-f) in place
  for (int f = threadIdx.x; f <= W / 2; f += blockDim.x) {
    const int fc = (W - f) & (W - 1);
    const float2 zf = Z[f], zc = Z[fc];
    const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
    const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
    const float pa = ga[f], pb = has_b ? ga[nf + f] : 0.f;
    float2 Ga = cscale(sa, 2.0f * pa);  // G = 2 gP S
    float2 Gb = cscale(sb, 2.0f * pb);
    if (f == 0 || f == W / 2) {
      // real-only bins: U_f = Re Ga + i Re Gb
      Z[f] = make_float2(Ga.x, Gb.x);
    } else {
      // U_f = Ga/2 + i Gb/2 ; U_{W-f} = conj(Ga)/2 + i conj(Gb)/2
      Z[f] = make_float2(0.5f * (Ga.x - Gb.y), 0.5f * (Ga.y + Gb.x));
      Z[fc] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (-Ga.y + Gb.x));
    }
  }
  __syncthreads();
  const float2* u = lds_fft(Z, O, W, 1, W, true, tw4, W);
  float* g = gx + (size_t)b * ld;
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    const float h = hann_periodic(j, W);
    const int ta = m * hop + j, tb = ta + hop;
    if (ta < T) atomicAdd(&g[ta], h * u[j].x);
    if (has_b && tb < T) atomicAdd(&g[tb], h * u[j].y);
  }
}


// ------------------------------------------------------------------------------------------
// win = 4096 (the reference's STFT, losses.py:512-535): register-resident radix-16 x 3.
// 256 threads per frame pair, 16 points per thread; thread i starts with x[i + 256 k] (loaded straight
// from memory, windowed), the three passes exchange through ONE padded LDS buffer (34 KB -> 4 blocks
// per CU; the generic ping-pong Stockham needs 74 KB, 2 blocks, and its stride-8 writes collide in the
// banks), and thread i ends with the bins X[i + 256 u].  Twiddle bases come from sincospif per thread
// (the pass-1 angle 2 pi i / 4096 also yields the Hann window by angle addition): no tables.
// ------------------------------------------------------------------------------------------
// windowed frame pair z[j] = hann(j) (x[m hop + j] + i x[(m+1) hop + j]) at j = i + 256 k; also returns the
// pass-1 twiddle base and keeps the window values for the adjoint's epilogue
__device__ __forceinline__ void s4k_load(const float* __restrict__ x, int T, int m, int nframes, int i,
                                         float2 (&a)[16], float (&h)[16], float2& w1) {
  float sn, cs;
  sincospif(2.0f * (float)i / 4096.0f, &sn, &cs);
  w1 = make_float2(cs, -sn);
  const bool has_b = m + 1 < nframes;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // cos(theta_i + k pi / 8) by angle addition (theta_i = 2 pi i / 4096)
    float sk, ck;
    sincospif((float)k * 0.125f, &sk, &ck);      // compile-time constants after unrolling
    h[k] = 0.5f - 0.5f * (cs * ck - sn * sk);
    const int ta = m * 2048 + i + 256 * k, tb = ta + 2048;
    const float va = ta < T ? x[ta] : 0.f;
    const float vb = (has_b && tb < T) ? x[tb] : 0.f;
    a[k] = make_float2(h[k] * va, h[k] * vb);
  }
}

__global__ __launch_bounds__(S4K_T) void k_stft4k_power(const float* __restrict__ x, int ld, int T,
                                                        int nframes, float* __restrict__ P,
                                                        float* __restrict__ zero_buf) {
  float2* buf = dyn_lds;
  const int b = blockIdx.y, m = blockIdx.x * 2, nf = 2049, i = threadIdx.x;
  if (zero_buf) {
    float* zb = zero_buf + (size_t)b * ld;
    const int t0 = m * 2048;
    const int t1 = (blockIdx.x == gridDim.x - 1) ? ld : t0 + 4096;
    for (int t = t0 + i; t < t1 && t < ld; t += S4K_T) zb[t] = 0.f;
  }
  float2 a[16], w1;
  float h[16];
  s4k_load(x + (size_t)b * ld, T, m, nframes, i, a, h, w1);
  fft4096(a, buf, i, w1, 1.0f);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
  __syncthreads();
  float* Pa = P + ((size_t)b * nframes + m) * nf;
  const bool has_b = m + 1 < nframes;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const float2 zf = a[u], zc = buf[S4K_PAD((4096 - f) & 4095)];
      // S_a = (Z_f + conj Z_{W-f})/2 ; S_b = (Z_f - conj Z_{W-f})/(2i)
      const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
      const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
      Pa[f] = sa.x * sa.x + sa.y * sa.y;
      if (has_b) Pa[nf + f] = sb.x * sb.x + sb.y * sb.y;
    }
  }
}

// Adjoint: recompute the pair's spectrum, build the Hermitian-symmetrised gradient spectrum in place,
// one inverse FFT for both frames, window, scatter (two frame terms per sample, added atomically into a
// zeroed buffer: order-free).  (Measured: walking two pairs per block to halve the atomics is SLOWER --
// 194 vs 138 us at 224 items -- the kernel is bound by registers / occupancy, not by the L2 atomics.)
__global__ __launch_bounds__(S4K_T) void k_stft4k_power_bwd(const float* __restrict__ x, int ld, int T,
                                                            int nframes, const float* __restrict__ gP,
                                                            float* __restrict__ gx) {
  float2* buf = dyn_lds;
  const int b = blockIdx.y, m = blockIdx.x * 2, nf = 2049, i = threadIdx.x;
  float2 a[16], w1;
  {
    float h[16];
    s4k_load(x + (size_t)b * ld, T, m, nframes, i, a, h, w1);
  }
  fft4096(a, buf, i, w1, 1.0f);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
  __syncthreads();
  const float* ga = gP + ((size_t)b * nframes + m) * nf;
  const bool has_b = m + 1 < nframes;
  // U = Ga_sym + i Gb_sym for the pair (f, W - f), from Z_f and Z_{W-f} (both re-read from LDS: the
  // register copy is dead here).  The pair is handled by exactly one thread (the owner of f <= W/2) and
  // nobody else reads either slot, so U_f and U_{W-f} overwrite the two slots in place with no barrier
#pragma unroll 3
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const int fc = (4096 - f) & 4095;
      const float2 zf = buf[S4K_PAD(f)], zc = buf[S4K_PAD(fc)];
      const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
      const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
      const float pa = ga[f], pb = has_b ? ga[nf + f] : 0.f;
      const float2 Ga = cscale(sa, 2.0f * pa), Gb = cscale(sb, 2.0f * pb);   // G = 2 gP S
      if (f == 0 || f == 2048) {
        buf[S4K_PAD(f)] = make_float2(Ga.x, Gb.x);                           // real-only bins
      } else {
        buf[S4K_PAD(f)] = make_float2(0.5f * (Ga.x - Gb.y), 0.5f * (Ga.y + Gb.x));      // U_f
        buf[S4K_PAD(fc)] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (-Ga.y + Gb.x));    // U_{W-f}
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = buf[S4K_PAD(i + 256 * k)];
  __syncthreads();
  fft4096(a, buf, i, w1, -1.0f);
  float* g = gx + (size_t)b * ld + m * 2048 + i;
  const int lim_a = T - m * 2048 - i, lim_b = has_b ? lim_a - 2048 : 0;     // sample j valid iff j < lim
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    float sk, ck;
    sincospif((float)u * 0.125f, &sk, &ck);
    const float hw = 0.5f - 0.5f * (w1.x * ck + w1.y * sk);     // w1 = (cos, -sin) theta_i
    const int j = 256 * u;
    if (j < lim_a) atomicAdd(g + j, hw * a[u].x);
    if (j < lim_b) atomicAdd(g + j + 2048, hw * a[u].y);
    if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the address arithmetic of later chunks out of this one
  }
}


// ------------------------------------------------------------------------------------------
// Pair-interleaved signals (x2 (pairs, ld) float2: item 2p in .x, item 2p + 1 in .y -- the layout the paired
// irfft produces and consumes): the float2 sample IS the complex FFT input, so one complex FFT carries ONE
// frame of TWO items (instead of two frames of one item); the Hermitian split is the same.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void s4k_load_pair(const float2* __restrict__ x2, int T, int m, int i,
                                              float2 (&a)[16], float2& w1) {
  float sn, cs;
  sincospif(2.0f * (float)i / 4096.0f, &sn, &cs);
  w1 = make_float2(cs, -sn);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float sk, ck;
    sincospif((float)k * 0.125f, &sk, &ck);
    const float h = 0.5f - 0.5f * (cs * ck - sn * sk);
    const int t = m * 2048 + i + 256 * k;
    const float2 v = t < T ? x2[t] : make_float2(0.f, 0.f);
    a[k] = make_float2(h * v.x, h * v.y);
  }
}

// (cos, sin)(u pi / 8), u = 0..15: the Hann window of the adjoint's rolled epilogue
__constant__ float2 c_hann_cs[16] = {
    {1.0f, 0.0f}, {0.92387953251128674f, 0.38268343236508977f}, {0.70710678118654752f, 0.70710678118654752f},
    {0.38268343236508977f, 0.92387953251128674f}, {0.0f, 1.0f}, {-0.38268343236508977f, 0.92387953251128674f},
    {-0.70710678118654752f, 0.70710678118654752f}, {-0.92387953251128674f, 0.38268343236508977f}, {-1.0f, 0.0f},
    {-0.92387953251128674f, -0.38268343236508977f}, {-0.70710678118654752f, -0.70710678118654752f},
    {-0.38268343236508977f, -0.92387953251128674f}, {0.0f, -1.0f}, {0.38268343236508977f, -0.92387953251128674f},
    {0.70710678118654752f, -0.70710678118654752f}, {0.92387953251128674f, -0.38268343236508977f}};

// wave priority of the pair STFT kernels (they are the main chain; the colorless pass and the EDC scans that run beside
// them have slack)
#define STFT_PAIR_PRIO 0
__global__ __launch_bounds__(S4K_T, 3) void k_stft4k_pair_power(const float2* __restrict__ x2, int ld, int T,
                                                             int nframes, int items, float* __restrict__ P,
                                                             float2* __restrict__ zero_buf) {
  if (STFT_PAIR_PRIO) __builtin_amdgcn_s_setprio(STFT_PAIR_PRIO);
  float2* buf = dyn_lds;
  const int p = blockIdx.y, m = blockIdx.x, nf = 2049, i = threadIdx.x;
  const int b1 = 2 * p;
  const bool two = b1 + 1 < items;
  if (zero_buf) {      // clear the adjoint's accumulation buffer: this frame's first hop, the last frame the tail
    float2* zb = zero_buf + (size_t)p * ld;
    const int t0 = m * 2048;
    const int t1 = (m == (int)gridDim.x - 1) ? ld : t0 + 2048;
    for (int t = t0 + i; t < t1 && t < ld; t += S4K_T) zb[t] = make_float2(0.f, 0.f);
  }
  float2 a[16], w1;
  s4k_load_pair(x2 + (size_t)p * ld, T, m, i, a, w1);
  fft4096(a, buf, i, w1, 1.0f);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
  __syncthreads();
  float* P1 = P + ((size_t)b1 * nframes + m) * nf;
  float* P2 = P1 + (size_t)nframes * nf;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const float2 zf = a[u], zc = buf[S4K_PAD((4096 - f) & 4095)];
      const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
      const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
      P1[f] = sa.x * sa.x + sa.y * sa.y;
      if (two) P2[f] = sb.x * sb.x + sb.y * sb.y;
    }
  }
}

__global__ __launch_bounds__(S4K_T, 3) void k_stft4k_pair_power_bwd(const float2* __restrict__ x2, int ld, int T,
                                                                 int nframes, int items,
                                                                 const float* __restrict__ gP,
                                                                 const float2* base2, float2* gx2, int parity,
                                                                 int late_base, const float* __restrict__ pbase,
                                                                 int pstart, int plen) {
  if (STFT_PAIR_PRIO) __builtin_amdgcn_s_setprio(STFT_PAIR_PRIO);
  float2* buf = dyn_lds;
  const int p = blockIdx.y, m = 2 * blockIdx.x + parity, nf = 2049, i = threadIdx.x;
  const int b1 = 2 * p;
  const bool two = b1 + 1 < items;
  float2 a[16], w1;
  s4k_load_pair(x2 + (size_t)p * ld, T, m, i, a, w1);
  fft4096(a, buf, i, w1, 1.0f);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
  __syncthreads();
  const float* ga = gP + ((size_t)b1 * nframes + m) * nf;
  const float* gb = ga + (size_t)nframes * nf;
#pragma unroll 3
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const int fc = (4096 - f) & 4095;
      const float2 zf = buf[S4K_PAD(f)], zc = buf[S4K_PAD(fc)];
      const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
      const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
      const float pa = ga[f], pb = two ? gb[f] : 0.f;
      const float2 Ga = cscale(sa, 2.0f * pa), Gb = cscale(sb, 2.0f * pb);   // G = 2 gP S
      if (f == 0 || f == 2048) {
        buf[S4K_PAD(f)] = make_float2(Ga.x, Gb.x);
      } else {
        buf[S4K_PAD(f)] = make_float2(0.5f * (Ga.x - Gb.y), 0.5f * (Ga.y + Gb.x));
        buf[S4K_PAD(fc)] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (-Ga.y + Gb.x));
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = buf[S4K_PAD(i + 256 * k)];
  __syncthreads();
  fft4096(a, buf, i, w1, -1.0f);
  // Frames of one parity tile the time axis without overlap (hop = win / 2): the even launch STORES
  // base + contribution (and base alone past the last even frame), the odd launch that follows adds its
  // contribution with a plain read-modify-write.  No atomics (they cost 50 of 139 us here), no cleared buffer.
  // late_base: the ODD launch adds base2 (everywhere, also where no odd frame reaches), the even launch stores its
  // contribution alone -- it then does not depend on the producer of base2 (the EDC scans on another stream).
  float2* g = gx2 + (size_t)p * ld;
  const float2* bs = base2 ? base2 + (size_t)p * ld : nullptr;
  const float2* src = parity ? g : (late_base ? nullptr : bs);
  const float2* add = (parity && late_base) ? bs : nullptr;
  // planar base (odd launch only): per-item rows of plen floats covering the samples [pstart, pstart + plen) -- the EDC
  // gradient of the fused decay kernel (decay.hip), zero elsewhere
  const float* pb1 = (parity && pbase) ? pbase + (size_t)b1 * plen : nullptr;
  const float* pb2 = (pb1 && two) ? pb1 + plen : nullptr;
  auto padd = [&](int t) {
    const int tt = t - pstart;
    if (tt < 0 || tt >= plen) return make_float2(0.f, 0.f);
    return make_float2(pb1[tt], pb2 ? pb2[tt] : 0.f);
  };
  const int tlim = parity ? T : ld;
  // the transform's outputs wait in the thread's own LDS slots while a rolled loop adds them to the gradient
  // signal: the epilogue then needs a handful of registers instead of the sixteen outputs plus sixteen loads
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
#pragma unroll 4
  for (int u = 0; u < 16; ++u) {
    const int j = i + 256 * u, t = m * 2048 + j;
    if (t < tlim) {
      float2 o = src ? src[t] : make_float2(0.f, 0.f);
      if (add) o = cadd(o, add[t]);
      if (pb1) o = cadd(o, padd(t));
      // periodic Hann at j = i + 256 u by angle addition from the pass-1 twiddle base w1 = (cos, -sin) theta_i and the
      // sixteen constants (cos, sin)(u pi / 8) -- two multiply-adds instead of a cospif per sample
      const float hw = t < T ? 0.5f - 0.5f * (w1.x * c_hann_cs[u].x + w1.y * c_hann_cs[u].y) : 0.f;
      const float2 v = buf[S4K_PAD(j)];
      g[t] = make_float2(o.x + hw * v.x, o.y + (two ? hw * v.y : 0.f));
    }
  }
  if (!parity && m + 2 >= nframes)
    for (int t = (m + 2) * 2048 + i; t < ld; t += S4K_T) g[t] = src ? src[t] : make_float2(0.f, 0.f);
  if (add) {      // samples no odd frame covers: the first hop, and whatever lies beyond the last odd frame
    if (m == 1)
      for (int t = i; t < 2048 && t < ld; t += S4K_T) g[t] = cadd(g[t], add[t]);
    if (m + 2 >= nframes)
      for (int t = (m * 2048 + 4096 < T ? m * 2048 + 4096 : T) + i; t < ld; t += S4K_T) g[t] = cadd(g[t], add[t]);
  }
  if (pb1) {
    if (m == 1)
      for (int t = i; t < 2048 && t < ld; t += S4K_T) g[t] = cadd(g[t], padd(t));
    if (m + 2 >= nframes)
      for (int t = (m * 2048 + 4096 < T ? m * 2048 + 4096 : T) + i; t < ld; t += S4K_T) g[t] = cadd(g[t], padd(t));
  }
}

extern "C" int gfdn_stft_power_pairs(const float* x2, int ld, int T, int items, int win, float* P,
                                     float* zero_buf2, void* stream) {
  if (!x2 || !P || items <= 0 || ld < T) return GFDN_E_BADARG;
  if (win != 4096) return GFDN_E_UNSUPPORTED;
  const int nframes = gfdn_stft_nframes(T, win);
  if (nframes <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_stft4k_pair_power, dim3(nframes, (items + 1) / 2), dim3(S4K_T), S4K_LDS * sizeof(float2),
                     (hipStream_t)stream, (const float2*)x2, ld, T, nframes, items, P, (float2*)zero_buf2);
  GFDN_LAUNCH_CHECK();
  return 0;
}

static int stft_pairs_bwd_run(const float* x2, int ld, int T, int items, int win, const float* gP,
                              const float* base2, float* gx2, int phases, int late_base, void* stream,
                              const float* pbase = nullptr, int pstart = 0, int plen = 0) {
  if (!x2 || !gP || !gx2 || items <= 0 || ld < T) return GFDN_E_BADARG;
  if (win != 4096) return GFDN_E_UNSUPPORTED;
  const int nframes = gfdn_stft_nframes(T, win);
  if (nframes <= 0) return GFDN_E_BADARG;
  if (late_base && nframes < 2) return GFDN_E_UNSUPPORTED;
  for (int parity = 0; parity < 2; ++parity) {
    const int nb = (nframes + 1 - parity) / 2;
    if (nb == 0 || !((phases >> parity) & 1)) continue;
    hipLaunchKernelGGL(k_stft4k_pair_power_bwd, dim3(nb, (items + 1) / 2), dim3(S4K_T),
                       S4K_LDS * sizeof(float2), (hipStream_t)stream, (const float2*)x2, ld, T, nframes, items, gP,
                       (const float2*)base2, (float2*)gx2, parity, late_base, pbase, pstart, plen);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gfdn_stft_power_pairs_bwd(const float* x2, int ld, int T, int items, int win, const float* gP,
                                         const float* base2, float* gx2, void* stream) {
  return stft_pairs_bwd_run(x2, ld, T, items, win, gP, base2, gx2, 3, 0, stream);
}

extern "C" int gfdn_stft_power_pairs_bwd_phase(const float* x2, int ld, int T, int items, int win, const float* gP,
                                               const float* base2, float* gx2, int phase, void* stream) {
  if (phase != 0 && phase != 1) return GFDN_E_BADARG;
  if (phase == 1 && gx2 == base2) return GFDN_E_BADARG;
  return stft_pairs_bwd_run(x2, ld, T, items, win, gP, phase ? base2 : nullptr, gx2, 1 << phase, 1, stream);
}

static size_t stft_lds(int W) { return ((size_t)2 * W + W / 4) * sizeof(float2); }

extern "C" int gfdn_stft_power(const float* x, int ld, int T, int batch, int win, float* P,
                               float* zero_buf, void* stream) {
  if (!x || !P || batch <= 0 || ld < T) return GFDN_E_BADARG;
  int nframes = gfdn_stft_nframes(T, win);
  if (nframes <= 0) return GFDN_E_BADARG;
  if (win == 4096) {
    hipLaunchKernelGGL(k_stft4k_power, dim3((nframes + 1) / 2, batch), dim3(S4K_T), S4K_LDS * sizeof(float2),
                       (hipStream_t)stream, x, ld, T, nframes, P, zero_buf);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (stft_lds(win) > 160 * 1024) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_stft_power, stft_lds(win));
  if (rc) return rc;
  hipLaunchKernelGGL(k_stft_power, dim3((nframes + 1) / 2, batch), dim3(STFT_T), stft_lds(win),
                     (hipStream_t)stream, x, ld, T, win, nframes, P, zero_buf);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_stft_power_bwd(const float* x, int ld, int T, int batch, int win,
                                   const float* gP, float* gx, void* stream) {
  if (!x || !gP || !gx || batch <= 0 || ld < T) return GFDN_E_BADARG;
  int nframes = gfdn_stft_nframes(T, win);
  if (nframes <= 0) return GFDN_E_BADARG;
  if (win == 4096) {
    hipLaunchKernelGGL(k_stft4k_power_bwd, dim3((nframes + 1) / 2, batch), dim3(S4K_T), S4K_LDS * sizeof(float2),
                       (hipStream_t)stream, x, ld, T, nframes, gP, gx);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (stft_lds(win) > 160 * 1024) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_stft_power_bwd, stft_lds(win));
  if (rc) return rc;
  hipLaunchKernelGGL(k_stft_power_bwd, dim3((nframes + 1) / 2, batch), dim3(STFT_T), stft_lds(win),
                     (hipStream_t)stream, x, ld, T, win, nframes, gP, gx);
  GFDN_LAUNCH_CHECK();
  return 0;
}
