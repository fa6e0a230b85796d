// Per-bin resolvent solve of the frequency-sampled GFDN and its output stage, for gfx950.
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn): feedback_loop.py:326-391 forms
// D = diag(z^m), Gamma, A for every bin as dense (K,N,N) complex128 tensors, inverts them with
// torch.linalg.inv, and model.py:615-619 contracts the inverse with c and b.  Here every bin is
// one small linear SOLVE that lives entirely in registers of a group of NP lanes (one matrix
// row per lane, Gauss-Jordan with virtual partial pivoting, pivot rows broadcast with
// ds_bpermute), 64/NP bins per wavefront; nothing of size K*N*N ever exists.
//
// Phase accuracy: z^m is evaluated as exp(m ln|z|) * exp(2 pi i frac(m * turns)) with the
// product and its reduction to [-1/2, 1/2] in float64 and only the final sincospi in float32.
// A naive float32 z**m is off by 1e-3 at m ~ 1600; this is at 1e-7.
#include "common.h"

// ------------------------------------------------------------------------------------------
// z -> (turns, log radius)
// ------------------------------------------------------------------------------------------
__global__ void k_zprep(const double* __restrict__ z, int K, double* __restrict__ turns,
                        double* __restrict__ logr) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double re = z[2 * k], im = z[2 * k + 1];
  turns[k] = atan2(im, re) * 0.15915494309189535;  // 1/(2 pi)
  logr[k] = 0.5 * log(re * re + im * im);
}

extern "C" int gfdn_zprep(const double* z, int K, double* turns, double* logr, void* stream) {
  if (!z || !turns || !logr || K <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_zprep, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, z, K,
                     turns, logr);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// z_k^{m} * inv_gamma for one delay line
__device__ __forceinline__ float2 zeta_pow(const double* __restrict__ turns,
                                           const double* __restrict__ logr, int k, float m,
                                           float inv_gamma) {
  double t = (double)m * turns[k];
  t -= rint(t);
  float s, c;
  sincospif(2.0f * (float)t, &s, &c);
  float mag = inv_gamma;
  if (logr) mag *= (float)exp((double)m * logr[k]);
  return make_float2(c * mag, s * mag);
}

// ------------------------------------------------------------------------------------------
// Complex arithmetic traits of the lane-parallel solve kernels: float2 (default) or double2 (the
// "precise" entry points: matrix entries and elimination in float64, for systems whose condition number
// eats float32 -- nearly lossless loops, T60 of tens of seconds: 1e3..1e5 -- where the reference's complex128
// inverse is 1e-4-exact and float32 is not).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cinv(double2 a) {
  const double d = 1.0 / (a.x * a.x + a.y * a.y);
  return make_double2(a.x * d, -a.y * d);
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 cscale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ double2 zeta_pow_d(const double* __restrict__ turns, const double* __restrict__ logr,
                                              int k, float m, double inv_gamma) {
  double t = (double)m * turns[k];
  t -= rint(t);
  double s, c;
  sincospi(2.0 * t, &s, &c);
  double mag = inv_gamma;
  if (logr) mag *= exp((double)m * logr[k]);
  return make_double2(c * mag, s * mag);
}
template <typename C> struct Cx;
template <> struct Cx<float2> {
  typedef float R;
  static __device__ __forceinline__ float2 mk(float x, float y) { return make_float2(x, y); }
  static __device__ __forceinline__ float2 zeta(const double* t, const double* l, int k, float m, float ig) {
    return zeta_pow(t, l, k, m, ig);
  }
  template <typename A> static __device__ __forceinline__ float ig(const A& a, int i) { return a.inv_gamma[i]; }
};
template <> struct Cx<double2> {
  typedef double R;
  static __device__ __forceinline__ double2 mk(double x, double y) { return make_double2(x, y); }
  static __device__ __forceinline__ double2 zeta(const double* t, const double* l, int k, float m, double ig) {
    return zeta_pow_d(t, l, k, m, ig);
  }
  template <typename A> static __device__ __forceinline__ double ig(const A& a, int i) { return a.ig64[i]; }
};

// ------------------------------------------------------------------------------------------
// Gauss-Jordan on an n x n complex system spread over NP lanes (lane r holds row r).
// On return the lane whose pivot column is `pivcol` holds x[pivcol] = rhs * pivinv.
// ------------------------------------------------------------------------------------------
template <int NP, typename C = float2>
__device__ __forceinline__ C gauss_jordan(C (&row)[NP], C rhs, int n, int r, int& pivcol) {
  typedef typename Cx<C>::R R;
  const bool active = r < n;
  bool used = !active;
  C pivinv = Cx<C>::mk(0, 0);
  pivcol = -1;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (j < n) {  // wave-uniform
      R mag = used ? (R)-1 : (row[j].x * row[j].x + row[j].y * row[j].y);
      int best = r;
#pragma unroll
      for (int off = NP / 2; off >= 1; off >>= 1) {
        R om = __shfl_xor(mag, off, NP);
        int ol = __shfl_xor(best, off, NP);
        bool take = (om > mag) || (om == mag && ol < best);
        mag = take ? om : mag;
        best = take ? ol : best;
      }
      C pr[NP];
#pragma unroll
      for (int c = j; c < NP; ++c) {
        if (c < n) {
          pr[c].x = __shfl(row[c].x, best, NP);
          pr[c].y = __shfl(row[c].y, best, NP);
        }
      }
      C prhs;
      prhs.x = __shfl(rhs.x, best, NP);
      prhs.y = __shfl(rhs.y, best, NP);
      C inv = cinv(pr[j]);
      if (r == best) {
        used = true;
        pivcol = j;
        pivinv = inv;
      } else if (active) {
        C f = cmul(row[j], inv);
#pragma unroll
        for (int c = j + 1; c < NP; ++c) {
          if (c < n) {
            row[c].x -= f.x * pr[c].x - f.y * pr[c].y;
            row[c].y -= f.x * pr[c].y + f.y * pr[c].x;
          }
        }
        rhs.x -= f.x * prhs.x - f.y * prhs.y;
        rhs.y -= f.x * prhs.y + f.y * prhs.x;
        row[j] = Cx<C>::mk(0, 0);
      }
    }
  }
  return cmul(rhs, pivinv);
}

// row r of  diag(zeta) - A   (swap=false: A[r][c];  swap=true: A[c][r])
template <int NP, typename C = float2>
__device__ __forceinline__ void build_row(C (&row)[NP], const float* __restrict__ Ablk,
                                          int n, int r, bool swap, C diag) {
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    float a = 0.f;
    if (c < n && r < n) a = swap ? Ablk[c * n + r] : Ablk[r * n + c];
    row[c] = Cx<C>::mk(-a, 0);
    if (c == r) row[c] = Cx<C>::mk(diag.x - a, diag.y);
  }
}

struct SolveArgs {
  const double* turns;
  const double* logr;
  int K, nblk, nper;
  const float* A;
  const float* delays;
  const float* inv_gamma;
  const float* b;
  int transpose;
  // frequency-dependent absorption (feedback_loop.py:332-344, :376-381): igz[k][i] = 1 / Gamma_i(z_k), complex,
  // (K, N); NULL: the scalar inv_gamma above.  With igz the scalar table must hold ones.
  const float2* igz;
  const double* ig64;     // precise entry points: 1 / gamma in float64 (inv_gamma is then unused)
};

// zeta_i(z_k) = z_k^{m_i} / gamma_i  (scalar gain)  or  z_k^{m_i} / Gamma_i(z_k)  (absorption filter)
__device__ __forceinline__ float2 zeta_abs(const SolveArgs& a, int k, int i, float2 zeta) {
  return a.igz ? cmul(zeta, a.igz[(size_t)k * (a.nblk * a.nper) + i]) : zeta;
}
__device__ __forceinline__ double2 zeta_abs(const SolveArgs& a, int k, int i, double2 zeta) {
  if (!a.igz) return zeta;
  const float2 g = a.igz[(size_t)k * (a.nblk * a.nper) + i];
  return cmul(zeta, make_double2(g.x, g.y));
}

template <int NP, typename C = float2>
__global__ __launch_bounds__(256) void k_solve_fwd(SolveArgs a, float2* __restrict__ Y) {
  constexpr int SPB = 256 / NP;
  const int r = threadIdx.x % NP, grp = threadIdx.x / NP;
  const int blk = blockIdx.y, n = a.nper, N = a.nblk * a.nper;
  const int k = blockIdx.x * SPB + grp;
  const bool valid = k < a.K;
  const int kk = valid ? k : a.K - 1;
  const int i = blk * n + (r < n ? r : 0);
  C zeta = zeta_abs(a, kk, i, Cx<C>::zeta(a.turns, a.logr, kk, a.delays[i], Cx<C>::ig(a, i)));
  C row[NP];
  build_row<NP, C>(row, a.A + (size_t)blk * n * n, n, r, a.transpose != 0, zeta);
  C rhs = Cx<C>::mk(r < n ? a.b[i] : 0.f, 0);
  int pivcol;
  C y = gauss_jordan<NP, C>(row, rhs, n, r, pivcol);
  if (valid && pivcol >= 0) Y[(size_t)k * N + blk * n + pivcol] = make_float2((float)y.x, (float)y.y);
}

// Ysaved: the forward solution (K, N) when the caller still holds it (the re-solve is skipped),
// or NULL.
template <int NP, typename C = float2>
__global__ __launch_bounds__(256) void k_solve_bwd(SolveArgs a, const float2* __restrict__ gY,
                                                   const float2* __restrict__ Ysaved,
                                                   float* __restrict__ partial) {
  constexpr int SPB = 256 / NP;
  __shared__ C s_perm[256];
  __shared__ float s_acc[256 * (NP + 2)];
  const int r = threadIdx.x % NP, grp = threadIdx.x / NP;
  const int blk = blockIdx.y, n = a.nper, N = a.nblk * a.nper;
  const bool active = r < n;
  const int i = blk * n + (active ? r : 0);
  const float* Ablk = a.A + (size_t)blk * n * n;
  const float m_i = a.delays[i], b_i = active ? a.b[i] : 0.f;
  const typename Cx<C>::R ig_i = Cx<C>::ig(a, i);
  const bool tr = a.transpose != 0;

  float acc[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) acc[c] = 0.f;
  float accb = 0.f, accg = 0.f;

  for (int k0 = blockIdx.x * SPB; k0 < a.K; k0 += gridDim.x * SPB) {
    const int k = k0 + grp;
    const bool valid = k < a.K;
    const int kk = valid ? k : a.K - 1;
    // unit-magnitude phase separately: d T_ii / d inv_gamma = z^m
    typedef typename Cx<C>::R R;
    C zpow = Cx<C>::zeta(a.turns, a.logr, kk, m_i, 1.0f);
    C zeta = zeta_abs(a, kk, i, cscale(zpow, ig_i));
    C row[NP];
    int pivcol;
    // forward system  T y = b
    C ynat;
    if (Ysaved) {
      const float2 ys = active ? Ysaved[(size_t)kk * N + i] : make_float2(0.f, 0.f);
      ynat = Cx<C>::mk(ys.x, ys.y);
    } else {
      build_row<NP, C>(row, Ablk, n, r, tr, zeta);
      C y = gauss_jordan<NP, C>(row, Cx<C>::mk(b_i, 0), n, r, pivcol);
      __syncthreads();
      if (pivcol >= 0) s_perm[grp * NP + pivcol] = y;
      __syncthreads();
      ynat = active ? s_perm[grp * NP + r] : Cx<C>::mk(0, 0);
    }
    // adjoint system  T^H w = gY   (row r of T^H = conj of column r of T)
    build_row<NP, C>(row, Ablk, n, r, !tr, cconj(zeta));
    const float2 gs = active ? gY[(size_t)kk * N + i] : make_float2(0.f, 0.f);
    C w = gauss_jordan<NP, C>(row, Cx<C>::mk(gs.x, gs.y), n, r, pivcol);
    __syncthreads();
    if (pivcol >= 0) s_perm[grp * NP + pivcol] = w;
    __syncthreads();
    C wnat = active ? s_perm[grp * NP + r] : Cx<C>::mk(0, 0);
    if (!valid) { ynat = Cx<C>::mk(0, 0); wnat = Cx<C>::mk(0, 0); }
    // gT_ij = -w_i conj(y_j);  T = D - A  (or D - A^T)
    //   transpose=0: gA[i][j] = Re(w_i conj(y_j));  transpose=1: gA[i][j] = Re(w_j conj(y_i))
    const C mine = tr ? ynat : wnat;
    const C other = tr ? wnat : ynat;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      if (c < n) {
        R ox = __shfl(other.x, c, NP), oy = __shfl(other.y, c, NP);
        acc[c] += (float)(mine.x * ox + mine.y * oy);
      }
    }
    accb += (float)wnat.x;
    // g inv_gamma_i = Re(conj(gT_ii) z^m) = -Re(conj(w_i) y_i z^m)
    C yz = cmul(ynat, zpow);
    accg -= (float)(wnat.x * yz.x + wnat.y * yz.y);
  }
  // deterministic reduction over the SPB lane groups of this block
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NP; ++c) s_acc[(grp * NP + r) * (NP + 2) + c] = acc[c];
  s_acc[(grp * NP + r) * (NP + 2) + NP] = accb;
  s_acc[(grp * NP + r) * (NP + 2) + NP + 1] = accg;
  __syncthreads();
  const int per = n * n + 2 * n;
  float* out = partial + ((size_t)blockIdx.x * a.nblk + blk) * per;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    int rr, cc;
    if (e < n * n) { rr = e / n; cc = e % n; }
    else if (e < n * n + n) { rr = e - n * n; cc = NP; }
    else { rr = e - n * n - n; cc = NP + 1; }
    float s = 0.f;
    for (int g2 = 0; g2 < SPB; ++g2) s += s_acc[(g2 * NP + rr) * (NP + 2) + cc];
    out[e] = s;
  }
}

// ------------------------------------------------------------------------------------------
// Blocks of 10 ... 12 lines: NP = n lanes per system, PACKED -- 64 / NP
// systems per wavefront (7 for n = 9) instead of the 4 that the power-of-two kernel above fits with 16 lanes each, seven
// of them idle.  Lane groups of a non-power-of-two size have no xor butterfly: the pivot search reads the NP candidates
// with NP shuffles (every lane of the group reaches the same choice, ties to the lowest row as above) and rows are
// broadcast from an explicit source lane.  Same arithmetic, same pivots: the results equal the 16-lane kernel's.
// ------------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ float2 gauss_jordan_pk(float2 (&row)[NP], float2 rhs, int n, int r, int base, int& pivcol) {
  const bool active = r < n;
  bool used = !active;
  float2 pivinv = make_float2(0.f, 0.f);
  pivcol = -1;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (j < n) {  // wave-uniform
      const float mag = used ? -1.0f : (row[j].x * row[j].x + row[j].y * row[j].y);
      float bm = -2.0f;
      int best = 0;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const float om = __shfl(mag, base + i, 64);
        if (om > bm) { bm = om; best = i; }
      }
      float2 pr[NP];
#pragma unroll
      for (int c = j; c < NP; ++c) {
        if (c < n) {
          pr[c].x = __shfl(row[c].x, base + best, 64);
          pr[c].y = __shfl(row[c].y, base + best, 64);
        }
      }
      float2 prhs;
      prhs.x = __shfl(rhs.x, base + best, 64);
      prhs.y = __shfl(rhs.y, base + best, 64);
      const float2 inv = cinv(pr[j]);
      if (r == best) {
        used = true;
        pivcol = j;
        pivinv = inv;
      } else if (active) {
        const float2 f = cmul(row[j], inv);
#pragma unroll
        for (int c = j + 1; c < NP; ++c) {
          if (c < n) {
            row[c].x -= f.x * pr[c].x - f.y * pr[c].y;
            row[c].y -= f.x * pr[c].y + f.y * pr[c].x;
          }
        }
        rhs.x -= f.x * prhs.x - f.y * prhs.y;
        rhs.y -= f.x * prhs.y + f.y * prhs.x;
        row[j] = make_float2(0.f, 0.f);
      }
    }
  }
  return cmul(rhs, pivinv);
}

// lane -> (system of the wave, row): the last 64 - NP (64 / NP) lanes of a wave idle (row index NP: inactive)
template <int NP>
struct PkLane {
  int r, base, grp;
  bool ok;
  __device__ __forceinline__ PkLane() {
    constexpr int GPW = 64 / NP;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g0 = lane / NP;
    ok = g0 < GPW;
    const int gl = ok ? g0 : GPW - 1;
    r = ok ? lane - gl * NP : NP;
    base = gl * NP;
    grp = wv * GPW + gl;
  }
};

template <int NP>
__global__ __launch_bounds__(256) void k_solve_fwd_pk(SolveArgs a, float2* __restrict__ Y) {
  constexpr int SPB = 4 * (64 / NP);
  const PkLane<NP> L;
  const int r = L.r, blk = blockIdx.y, n = a.nper, N = a.nblk * a.nper;
  const int k = blockIdx.x * SPB + L.grp;
  const bool valid = L.ok && k < a.K;
  const int kk = k < a.K ? k : a.K - 1;
  const int i = blk * n + (r < n ? r : 0);
  const float2 zeta = zeta_abs(a, kk, i, Cx<float2>::zeta(a.turns, a.logr, kk, a.delays[i], Cx<float2>::ig(a, i)));
  float2 row[NP];
  build_row<NP, float2>(row, a.A + (size_t)blk * n * n, n, r, a.transpose != 0, zeta);
  const float2 rhs = make_float2(r < n ? a.b[i] : 0.f, 0.f);
  int pivcol;
  const float2 y = gauss_jordan_pk<NP>(row, rhs, n, r, L.base, pivcol);
  if (valid && pivcol >= 0) Y[(size_t)k * N + blk * n + pivcol] = y;
}

template <int NP>
__global__ __launch_bounds__(256) void k_solve_bwd_pk(SolveArgs a, const float2* __restrict__ gY,
                                                      const float2* __restrict__ Ysaved,
                                                      float* __restrict__ partial) {
  constexpr int SPB = 4 * (64 / NP);
  __shared__ float2 s_perm[256];
  __shared__ float s_acc[SPB * NP * (NP + 2)];
  const PkLane<NP> L;
  const int r = L.r, grp = L.grp, base = L.base;
  const int blk = blockIdx.y, n = a.nper, N = a.nblk * a.nper;
  const bool active = r < n;
  const int i = blk * n + (active ? r : 0);
  const float* Ablk = a.A + (size_t)blk * n * n;
  const float m_i = a.delays[i], b_i = active ? a.b[i] : 0.f;
  const float ig_i = Cx<float2>::ig(a, i);
  const bool tr = a.transpose != 0;

  float acc[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) acc[c] = 0.f;
  float accb = 0.f, accg = 0.f;

  for (int k0 = blockIdx.x * SPB; k0 < a.K; k0 += gridDim.x * SPB) {
    const int k = k0 + grp;
    const bool valid = L.ok && k < a.K;
    const int kk = k < a.K ? k : a.K - 1;
    const float2 zpow = Cx<float2>::zeta(a.turns, a.logr, kk, m_i, 1.0f);
    const float2 zeta = zeta_abs(a, kk, i, cscale(zpow, ig_i));
    float2 row[NP];
    int pivcol;
    float2 ynat;
    if (Ysaved) {
      ynat = active ? Ysaved[(size_t)kk * N + i] : make_float2(0.f, 0.f);
    } else {
      build_row<NP, float2>(row, Ablk, n, r, tr, zeta);
      const float2 y = gauss_jordan_pk<NP>(row, make_float2(b_i, 0.f), n, r, base, pivcol);
      __syncthreads();
      if (pivcol >= 0) s_perm[grp * NP + pivcol] = y;
      __syncthreads();
      ynat = active ? s_perm[grp * NP + r] : make_float2(0.f, 0.f);
    }
    build_row<NP, float2>(row, Ablk, n, r, !tr, cconj(zeta));
    const float2 gs = active ? gY[(size_t)kk * N + i] : make_float2(0.f, 0.f);
    const float2 w = gauss_jordan_pk<NP>(row, gs, n, r, base, pivcol);
    __syncthreads();
    if (pivcol >= 0) s_perm[grp * NP + pivcol] = w;
    __syncthreads();
    float2 wnat = active ? s_perm[grp * NP + r] : make_float2(0.f, 0.f);
    if (!valid) { ynat = make_float2(0.f, 0.f); wnat = make_float2(0.f, 0.f); }
    const float2 mine = tr ? ynat : wnat;
    const float2 other = tr ? wnat : ynat;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      if (c < n) {
        const float ox = __shfl(other.x, base + c, 64), oy = __shfl(other.y, base + c, 64);
        acc[c] += mine.x * ox + mine.y * oy;
      }
    }
    accb += wnat.x;
    const float2 yz = cmul(ynat, zpow);
    accg -= wnat.x * yz.x + wnat.y * yz.y;
  }
  __syncthreads();
  if (L.ok) {
#pragma unroll
    for (int c = 0; c < NP; ++c) s_acc[(grp * NP + r) * (NP + 2) + c] = acc[c];
    s_acc[(grp * NP + r) * (NP + 2) + NP] = accb;
    s_acc[(grp * NP + r) * (NP + 2) + NP + 1] = accg;
  }
  __syncthreads();
  const int per = n * n + 2 * n;
  float* out = partial + ((size_t)blockIdx.x * a.nblk + blk) * per;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    int rr, cc;
    if (e < n * n) { rr = e / n; cc = e % n; }
    else if (e < n * n + n) { rr = e - n * n; cc = NP; }
    else { rr = e - n * n - n; cc = NP + 1; }
    float sum = 0.f;
    for (int g2 = 0; g2 < SPB; ++g2) sum += s_acc[(g2 * NP + rr) * (NP + 2) + cc];
    out[e] = sum;
  }
}

// ------------------------------------------------------------------------------------------
// Blocks of 9 lines (the directional model: 9 SH channels per group): THREE ROWS PER LANE.  A system of NP rows
// sits on LPS = ceil(NP / 3) lanes (lane t holds rows t, t + LPS, t + 2 LPS), 64 / LPS systems per wavefront: 21 for
// n = 9.  What limits the lane-parallel elimination is the LDS crossbar: every value that crosses lanes is a
// ds_bpermute_b32 for the whole wave, and with one row per lane a 9 x 9 solve costs 189 of them per 7 systems (pivot
// search 9 per column, pivot row 2 (9 - j) + 2).  With three rows per lane the pivot search is a local compare plus LPS
// exchanges of ONE key (|a|^2 with the row number in its four low mantissa bits), the pivot row still costs
// 2 (9 - j) + 2 exchanges -- but now per 21 systems: 6.4 instead of 27 crossbar operations per system, and the
// elimination arithmetic per system is unchanged.  Partial pivoting by magnitude (ties and magnitudes equal to 19 bits:
// the lowest row), virtual row exchange as in the kernels above; results agree with them to rounding.
// ------------------------------------------------------------------------------------------
#define RL_R 3
template <int NP>
struct RlLane {
  static constexpr int LPS = (NP + RL_R - 1) / RL_R, GPW = 64 / LPS;
  static_assert(LPS * RL_R == NP, "every lane of a system holds RL_R rows");
  int t, base, grp;
  bool ok;
  __device__ __forceinline__ RlLane() {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g0 = lane / LPS;
    ok = g0 < GPW;                        // (the lanes left over at the end of the wave shadow the last system: same
    const int gl = ok ? g0 : GPW - 1;     //  control flow, nothing stored)
    base = gl * LPS;
    t = lane - g0 * LPS;
    grp = wv * GPW + gl;
  }
};
// row r of  diag(zeta) - A  (swap: of diag(zeta) - A^T), A (NP, NP) in LDS
template <int NP>
__device__ __forceinline__ void rl_build_row(float2 (&row)[NP], const float* sA, int r, bool swap, float2 diag) {
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    const float av = swap ? sA[c * NP + r] : sA[r * NP + c];
    row[c] = make_float2((c == r ? diag.x : 0.f) - av, c == r ? diag.y : 0.f);
  }
}

// (by value: a conditional expression over array ELEMENTS is an lvalue -- the compiler selects the address, the index becomes
// dynamic and the whole row array moves to scratch)
__device__ __forceinline__ float2 rl_pick(int q, float2 a0, float2 a1, float2 a2) {
  return make_float2(q == 0 ? a0.x : (q == 1 ? a1.x : a2.x), q == 0 ? a0.y : (q == 1 ? a1.y : a2.y));
}
// On return row q of the lane (global row t + LPS q) is the pivot row of column pivcol[q] and x[pivcol[q]] = rhs[q] pivinv[q].
// n = NP exactly (the dispatch guarantees it): no guards on the order in the body.
template <int NP>
__device__ __forceinline__ void gauss_jordan_rl(float2 (&row)[RL_R][NP], float2 (&rhs)[RL_R], int t, int base,
                                                int (&pivcol)[RL_R], float2 (&pivinv)[RL_R]) {
  constexpr int LPS = RlLane<NP>::LPS;
  bool used[RL_R];
#pragma unroll
  for (int q = 0; q < RL_R; ++q) {
    used[q] = false;
    pivcol[q] = 0;
    pivinv[q] = make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    // the lane's candidate: largest |a_qj|^2 among its unused rows, row number in the low bits (15 - row: ties go to the
    // lowest row; a candidate's key is never 0)
    unsigned key = 0u;
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      const float mag = row[q][j].x * row[q][j].x + row[q][j].y * row[q][j].y;
      const unsigned kq = (__float_as_uint(mag) & ~15u) | (unsigned)(15 - (t + LPS * q));
      key = (!used[q] && kq > key) ? kq : key;
    }
    unsigned best = 0u;
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const unsigned o = (unsigned)__shfl((int)key, base + i, 64);
      best = o > best ? o : best;
    }
    const int brow = 15 - (int)(best & 15u), bq = brow / LPS, bl = brow - bq * LPS;
    float2 pr[NP];
#pragma unroll
    for (int c = j; c < NP; ++c) {
      const float2 sel = rl_pick(bq, row[0][c], row[1][c], row[2][c]);
      pr[c].x = __shfl(sel.x, base + bl, 64);
      pr[c].y = __shfl(sel.y, base + bl, 64);
    }
    const float2 rsel = rl_pick(bq, rhs[0], rhs[1], rhs[2]);
    float2 prhs;
    prhs.x = __shfl(rsel.x, base + bl, 64);
    prhs.y = __shfl(rsel.y, base + bl, 64);
    const float2 inv = cinv(pr[j]);
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      const bool piv = t + LPS * q == brow;
      used[q] = used[q] || piv;
      pivcol[q] = piv ? j : pivcol[q];
      pivinv[q] = make_float2(piv ? inv.x : pivinv[q].x, piv ? inv.y : pivinv[q].y);
      // (the pivot row itself takes f = 0: no branch)
      float2 f = cmul(row[q][j], inv);
      f = make_float2(piv ? 0.f : f.x, piv ? 0.f : f.y);
#pragma unroll
      for (int c = j + 1; c < NP; ++c) {
        row[q][c].x -= f.x * pr[c].x - f.y * pr[c].y;
        row[q][c].y -= f.x * pr[c].y + f.y * pr[c].x;
      }
      rhs[q].x -= f.x * prhs.x - f.y * prhs.y;
      rhs[q].y -= f.x * prhs.y + f.y * prhs.x;
      row[q][j] = make_float2(piv ? row[q][j].x : 0.f, piv ? row[q][j].y : 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);      // (column by column: scheduled across columns the exchanges pile up in registers)
  }
}

template <int NP>
__global__ __launch_bounds__(256) void k_solve_fwd_rl(SolveArgs a, float2* __restrict__ Y) {
  constexpr int LPS = RlLane<NP>::LPS, SPB = 4 * RlLane<NP>::GPW;
  __shared__ float s_A[NP * NP];
  const RlLane<NP> L;
  const int blk = blockIdx.y, N = a.nblk * NP;
  for (int e = threadIdx.x; e < NP * NP; e += blockDim.x) s_A[e] = a.A[(size_t)blk * NP * NP + e];
  __syncthreads();
  const int k = blockIdx.x * SPB + L.grp;
  const bool valid = L.ok && k < a.K;
  const int kk = k < a.K ? k : a.K - 1;
  float2 row[RL_R][NP], rhs[RL_R], pivinv[RL_R];
  int pivcol[RL_R];
#pragma unroll
  for (int q = 0; q < RL_R; ++q) {
    const int r = L.t + LPS * q, i = blk * NP + r;
    const float2 zeta = zeta_abs(a, kk, i, Cx<float2>::zeta(a.turns, a.logr, kk, a.delays[i], Cx<float2>::ig(a, i)));
    rl_build_row<NP>(row[q], s_A, r, a.transpose != 0, zeta);
    rhs[q] = make_float2(a.b[i], 0.f);
  }
  gauss_jordan_rl<NP>(row, rhs, L.t, L.base, pivcol, pivinv);
  if (valid) {
#pragma unroll
    for (int q = 0; q < RL_R; ++q) Y[(size_t)k * N + blk * NP + pivcol[q]] = cmul(rhs[q], pivinv[q]);
  }
}

template <int NP, bool SAVED>
__global__ __launch_bounds__(256) void k_solve_bwd_rl(SolveArgs a, const float2* __restrict__ gY,
                                                      const float2* __restrict__ Ysaved,
                                                      float* __restrict__ partial) {
  constexpr int LPS = RlLane<NP>::LPS, GPW = RlLane<NP>::GPW, SPB = 4 * GPW;
  // natural-order solutions of the wave's systems (every wave reads only what it wrote: LDS operations of a wave execute
  // in order, no block barrier inside the loop); one more slot set for the lanes that shadow the last system
  __shared__ float2 s_y[(SPB + 4) * NP], s_w[(SPB + 4) * NP];
  __shared__ float s_acc[SPB * NP * (NP + 2)];
  __shared__ float s_A[NP * NP];
  const RlLane<NP> L;
  const int t = L.t, grp = L.grp, base = L.base;
  const int blk = blockIdx.y, N = a.nblk * NP;
  const bool tr = a.transpose != 0;
  for (int e = threadIdx.x; e < NP * NP; e += blockDim.x) s_A[e] = a.A[(size_t)blk * NP * NP + e];
  __syncthreads();
  int li[RL_R];
  float m_i[RL_R], b_i[RL_R], ig_i[RL_R];
#pragma unroll
  for (int q = 0; q < RL_R; ++q) {
    li[q] = blk * NP + t + LPS * q;
    m_i[q] = a.delays[li[q]];
    b_i[q] = a.b[li[q]];
    ig_i[q] = Cx<float2>::ig(a, li[q]);
  }
  float acc[RL_R][NP], accb[RL_R], accg[RL_R];
#pragma unroll
  for (int q = 0; q < RL_R; ++q) {
    accb[q] = 0.f;
    accg[q] = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c) acc[q][c] = 0.f;
  }
  // (shadow lanes write their copies into the spare slot set of their wave)
  const int slot = L.ok ? grp : SPB + (threadIdx.x >> 6);
  float2* sy = s_y + slot * NP;
  float2* sw = s_w + slot * NP;

#pragma unroll 1
  for (int k0 = blockIdx.x * SPB; k0 < a.K; k0 += gridDim.x * SPB) {
    const int k = k0 + grp;
    const bool valid = L.ok && k < a.K;
    const int kk = k < a.K ? k : a.K - 1;
    float2 zpow[RL_R], zeta[RL_R], row[RL_R][NP], rhs[RL_R], pivinv[RL_R];
    int pivcol[RL_R];
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      zpow[q] = Cx<float2>::zeta(a.turns, a.logr, kk, m_i[q], 1.0f);
      zeta[q] = zeta_abs(a, kk, li[q], cscale(zpow[q], ig_i[q]));
    }
    if (SAVED) {
#pragma unroll
      for (int q = 0; q < RL_R; ++q) sy[t + LPS * q] = Ysaved[(size_t)kk * N + li[q]];
    } else {
#pragma unroll
      for (int q = 0; q < RL_R; ++q) {
        rl_build_row<NP>(row[q], s_A, t + LPS * q, tr, zeta[q]);
        rhs[q] = make_float2(b_i[q], 0.f);
      }
      gauss_jordan_rl<NP>(row, rhs, t, base, pivcol, pivinv);
#pragma unroll
      for (int q = 0; q < RL_R; ++q) sy[pivcol[q]] = cmul(rhs[q], pivinv[q]);
    }
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      rl_build_row<NP>(row[q], s_A, t + LPS * q, !tr, cconj(zeta[q]));
      rhs[q] = gY[(size_t)kk * N + li[q]];
    }
    gauss_jordan_rl<NP>(row, rhs, t, base, pivcol, pivinv);
#pragma unroll
    for (int q = 0; q < RL_R; ++q) sw[pivcol[q]] = cmul(rhs[q], pivinv[q]);
    // gT_ij = -w_i conj(y_j);  T = D - A  (or D - A^T)
    //   transpose=0: gA[i][j] = Re(w_i conj(y_j));  transpose=1: gA[i][j] = Re(w_j conj(y_i))
    const float2* so = tr ? sw : sy;
    float2 other[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) other[c] = so[c];
    const float live = valid ? 1.0f : 0.f;
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      const float2 ynat = cscale(sy[t + LPS * q], live), wnat = cscale(sw[t + LPS * q], live);
      const float2 mine = tr ? ynat : wnat;
#pragma unroll
      for (int c = 0; c < NP; ++c) acc[q][c] += mine.x * other[c].x + mine.y * other[c].y;
      accb[q] += wnat.x;
      // g inv_gamma_i = Re(conj(gT_ii) z^m) = -Re(conj(w_i) y_i z^m)
      const float2 yz = cmul(ynat, zpow[q]);
      accg[q] -= wnat.x * yz.x + wnat.y * yz.y;
    }
  }
  // deterministic reduction over the SPB lane groups of this block
  if (L.ok) {
#pragma unroll
    for (int q = 0; q < RL_R; ++q) {
      const int r = t + LPS * q;
#pragma unroll
      for (int c = 0; c < NP; ++c) s_acc[(grp * NP + r) * (NP + 2) + c] = acc[q][c];
      s_acc[(grp * NP + r) * (NP + 2) + NP] = accb[q];
      s_acc[(grp * NP + r) * (NP + 2) + NP + 1] = accg[q];
    }
  }
  __syncthreads();
  constexpr int per = NP * NP + 2 * NP;
  float* out = partial + ((size_t)blockIdx.x * a.nblk + blk) * per;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    int rr, cc;
    if (e < NP * NP) { rr = e / NP; cc = e % NP; }
    else if (e < NP * NP + NP) { rr = e - NP * NP; cc = NP; }
    else { rr = e - NP * NP - NP; cc = NP + 1; }
    float sum = 0.f;
    for (int g2 = 0; g2 < SPB; ++g2) sum += s_acc[(g2 * NP + rr) * (NP + 2) + cc];
    out[e] = sum;
  }
}

// out[e] = sum_p partial[p][e]; one 256-thread block per output element, fixed-order tree
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partial,
                                                         int nparts, int per,
                                                         float* __restrict__ out) {
  __shared__ float s_red[16];
  const int e = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)p * per + e];
  s = block_sum(s, s_red);
  if (threadIdx.x == 0) out[e] = s;
}

// scatter the reduced [nblk][n*n + 2n] record into gA, gb, ginv_gamma
__global__ __launch_bounds__(256) void k_solve_bwd_finish(const float* __restrict__ partial, int nparts, int nblk, int n,
                                   float* __restrict__ gA, float* __restrict__ gb,
                                   float* __restrict__ gig) {
  __shared__ float s_red[16];
  const int per = n * n + 2 * n;
  const int e = blockIdx.x;                       // one block per output element
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)p * nblk * per + e];
  s = block_sum(s, s_red);
  if (threadIdx.x != 0) return;
  int blk = e / per, o = e % per;
  if (o < n * n) gA[(size_t)blk * n * n + o] = s;
  else if (o < n * n + n) gb[blk * n + (o - n * n)] = s;
  else gig[blk * n + (o - n * n - n)] = s;
}



// ------------------------------------------------------------------------------------------
// n <= 4 (the north-star layout: N = 16 = 4 groups x 4): ONE THREAD per system.  The 4 x 4 complex
// matrix, the right-hand side and the pivot bookkeeping live in registers with compile-time indices
// (fully unrolled; the pivot row is picked with selects), so a wavefront solves 64 systems with no
// cross-lane traffic at all -- the NP-lane kernels above solve 16 per wave and spend most of their
// issue slots in ds_bpermute.  Same elimination order and the same pivot rule (largest |.|^2 among
// the unused rows, lowest row on ties) as gauss_jordan<NP>.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 sel4(int i, float2 a0, float2 a1, float2 a2, float2 a3) {
  float2 lo = (i == 1) ? a1 : a0, hi = (i == 3) ? a3 : a2;
  return (i >= 2) ? hi : lo;
}

// solves m x = r in place (m, r destroyed); rows / columns >= n must be the identity with zero rhs
__device__ __forceinline__ void gj4(float2 (&m)[4][4], float2 (&r)[4], float2 (&x)[4]) {
  bool used[4] = {false, false, false, false};
  int piv[4];
  float2 pinv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float bm = -1.0f;
    int best = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mag = used[i] ? -1.0f : (m[i][j].x * m[i][j].x + m[i][j].y * m[i][j].y);
      const bool take = mag > bm;
      bm = take ? mag : bm;
      best = take ? i : best;
    }
    float2 pr[4];
#pragma unroll
    for (int c = j; c < 4; ++c) pr[c] = sel4(best, m[0][c], m[1][c], m[2][c], m[3][c]);
    const float2 prhs = sel4(best, r[0], r[1], r[2], r[3]);
    const float2 inv = cinv(pr[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool isp = (i == best);
      used[i] = used[i] || isp;
      float2 f = cmul(m[i][j], inv);
      if (isp) f = make_float2(0.f, 0.f);
#pragma unroll
      for (int c = j + 1; c < 4; ++c) {
        m[i][c].x -= f.x * pr[c].x - f.y * pr[c].y;
        m[i][c].y -= f.x * pr[c].y + f.y * pr[c].x;
      }
      r[i].x -= f.x * prhs.x - f.y * prhs.y;
      r[i].y -= f.x * prhs.y + f.y * prhs.x;
    }
    piv[j] = best;
    pinv[j] = inv;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = cmul(sel4(piv[j], r[0], r[1], r[2], r[3]), pinv[j]);
}

// m = diag(zeta) - A  (swap: - A^T; conj_t: the conjugate transpose of that, for the adjoint system)
__device__ __forceinline__ void build4(float2 (&m)[4][4], const float* __restrict__ Ablk, int n, bool swap,
                                       const float2 (&zeta)[4], bool adj) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float av = 0.f;
      if (r < n && c < n) av = (swap != adj) ? Ablk[c * n + r] : Ablk[r * n + c];
      m[r][c] = make_float2(-av, 0.f);
      if (r == c) {
        if (r < n) m[r][c] = make_float2(zeta[r].x - av, adj ? -zeta[r].y : zeta[r].y);
        else m[r][c] = make_float2(1.f, 0.f);
      }
    }
}

// Work item w = k * nblk + blk (block index fastest): consecutive threads then touch consecutive
// n * 8-byte chunks of the bin-major (K, nblk * n) arrays -- Y, gY and the saved solution are read and
// written as one linear stream.  (With one block index per workgroup every lane reads 32 bytes out of a
// different 128-byte line: 4x the traffic; measured 103 us for the 28-block backward at K = 65 537.)
// The per-block constants (A, delays, gains) come from an LDS table instead of scalar registers.
#define S4_MAXBLK 64           // blocks whose constants fit the LDS table (28 floats each)
struct S4Const { float A[16], m[4], ig[4], b[4]; };

__device__ __forceinline__ void s4_stage(const SolveArgs& a, S4Const* tab) {
  const int n = a.nper;
  for (int e = threadIdx.x; e < a.nblk * 28; e += blockDim.x) {
    const int blk = e / 28, f = e - blk * 28;
    float v = 0.f;
    if (f < 16) {
      const int r = f >> 2, c = f & 3;
      if (r < n && c < n) v = a.A[(size_t)blk * n * n + r * n + c];
    } else {
      const int r = (f - 16) & 3, which = (f - 16) >> 2;
      const int i = blk * n + (r < n ? r : 0);
      if (which == 0) v = a.delays[i];
      else if (which == 1) v = a.inv_gamma ? a.inv_gamma[i] : 1.0f;
      else v = r < n ? a.b[i] : 0.f;
    }
    ((float*)&tab[blk])[f] = v;
  }
}

// m = diag(zeta) - A from the staged 4 x 4 block (zero-padded); swap / adj as build4
__device__ __forceinline__ void build4c(float2 (&m)[4][4], const S4Const& cst, int n, bool swap,
                                        const float2 (&zeta)[4], bool adj) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = (swap != adj) ? cst.A[c * 4 + r] : cst.A[r * 4 + c];
      m[r][c] = make_float2(-av, 0.f);
      if (r == c) {
        if (r < n) m[r][c] = make_float2(zeta[r].x - av, adj ? -zeta[r].y : zeta[r].y);
        else m[r][c] = make_float2(1.f, 0.f);
      }
    }
}

__global__ __launch_bounds__(256) void k_solve4_fwd(SolveArgs a, float2* __restrict__ Y) {
  __shared__ S4Const tab[S4_MAXBLK];
  s4_stage(a, tab);
  __syncthreads();
  const int n = a.nper;
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= (long long)a.K * a.nblk) return;
  const int k = (int)(w / a.nblk), blk = (int)(w - (long long)k * a.nblk);
  const S4Const cst = tab[blk];
  float2 zeta[4], rhs[4], m[4][4], y[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
    if (r < n) zeta[r] = zeta_abs(a, k, blk * n + r, zeta[r]);
    rhs[r] = make_float2(cst.b[r], 0.f);
  }
  build4c(m, cst, n, a.transpose != 0, zeta, false);
  gj4(m, rhs, y);
  float2* out = Y + (size_t)w * n;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < n) out[r] = y[r];
}

// partial[(part * nblk + blk) * per + e], per = n n + 2 n, laid out as k_solve_bwd does.
// Work items stride by gridDim.x * S4_ITEMS, a multiple of nblk: a thread keeps its block index.
__global__ __launch_bounds__(256) void k_solve4_bwd(SolveArgs a, const float2* __restrict__ gY,
                                                    const float2* __restrict__ Ysaved,
                                                    float* __restrict__ partial, int items) {
  __shared__ S4Const tab[S4_MAXBLK];
  extern __shared__ float s4_acc[];          // [items][25]
  s4_stage(a, tab);
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool tr = a.transpose != 0;
  const bool live = (int)threadIdx.x < items;            // items = the largest multiple of nblk <= 256
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S4Const cst = tab[blk];
  float acc[24];
#pragma unroll
  for (int e = 0; e < 24; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    float2 zpow[4], zeta[4], m[4][4], y[4], w[4], rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zpow[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], 1.0f);   // d T_ii / d inv_gamma = z^m
      zeta[r] = cscale(zpow[r], cst.ig[r]);
      if (r < n) zeta[r] = zeta_abs(a, k, blk * n + r, zeta[r]);
    }
    const size_t row = ((size_t)k * nblk + blk) * n;
    if (Ysaved) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = r < n ? Ysaved[row + r] : make_float2(0.f, 0.f);
    } else {
      build4c(m, cst, n, tr, zeta, false);
#pragma unroll
      for (int r = 0; r < 4; ++r) rhs[r] = make_float2(cst.b[r], 0.f);
      gj4(m, rhs, y);
    }
    // adjoint system T^H w = gY
    build4c(m, cst, n, tr, zeta, true);
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = r < n ? gY[row + r] : make_float2(0.f, 0.f);
    gj4(m, rhs, w);
    // gT_ij = -w_i conj(y_j), T = D - A (or D - A^T):
    //   transpose = 0: gA[i][j] = Re(w_i conj(y_j));  transpose = 1: gA[i][j] = Re(w_j conj(y_i))
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2 p = tr ? y[i] : w[i], q = tr ? w[j] : y[j];
        acc[i * 4 + j] += p.x * q.x + p.y * q.y;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[16 + i] += w[i].x;
      const float2 yz = cmul(y[i], zpow[i]);      // g inv_gamma_i = -Re(conj(w_i) y_i z^m)
      acc[20 + i] -= w[i].x * yz.x + w[i].y * yz.y;
    }
  }
  // fixed-order reduction over the threads of each block index (stride 25: conflict-free rows)
  if (live) {
#pragma unroll
    for (int e = 0; e < 24; ++e) s4_acc[threadIdx.x * 25 + e] = acc[e];
  }
  __syncthreads();
  const int per = n * n + 2 * n;
  for (int o = threadIdx.x; o < nblk * per; o += 256) {
    const int bq = o / per, e = o - bq * per;
    int src;
    if (e < n * n) src = (e / n) * 4 + (e % n);
    else if (e < n * n + n) src = 16 + (e - n * n);
    else src = 20 + (e - n * n - n);
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s4_acc[t * 25 + src];
    partial[((size_t)blockIdx.x * nblk + bq) * per + e] = sum;
  }
}

// partial[blk * gridDim.x + part]: energy of the group sums over this block's bins
__global__ __launch_bounds__(256) void k_subfdn4_energy(SolveArgs a, const float* __restrict__ c,
                                                        float* __restrict__ partial, int items) {
  __shared__ S4Const tab[S4_MAXBLK];
  __shared__ float s_e[256];
  s4_stage(a, tab);
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S4Const cst = tab[blk];
  float c_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c_i[r] = r < n ? c[blk * n + r] : 0.f;
  float acc = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    float2 zeta[4], m[4][4], y[4], rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
      rhs[r] = make_float2(cst.b[r], 0.f);
    }
    build4c(m, cst, n, a.transpose != 0, zeta, false);
    gj4(m, rhs, y);
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s.x += c_i[r] * y[r].x; s.y += c_i[r] * y[r].y; }
    acc += s.x * s.x + s.y * s.y;
  }
  s_e[threadIdx.x] = live ? acc : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < nblk; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}

// ------------------------------------------------------------------------------------------
// Thread-per-system kernels for 4 < n <= 8 (the N = 32 layout: 4 groups x 8 lines).  Same structure as the
// 4 x 4 kernels above: one thread eliminates one zero-padded 8 x 8 complex system held in registers with
// compile-time indices (LU with partial pivoting by conditional row swaps, back substitution) -- no
// cross-lane traffic, where the 8-lane kernels spend their time in ~200 dependent ds_bpermutes per system
// (131 k systems: 60-77 us -> see DESIGN.md).  The backward keeps its 80 per-thread accumulators in LDS
// columns ([e][thread]: conflict-free), which leaves the registers to the matrix.
// ------------------------------------------------------------------------------------------
#define S8_MAXBLK 32
struct S8Const { float A[64], m[8], ig[8], b[8]; };          // 88 floats

__device__ __forceinline__ void s8_stage(const SolveArgs& a, S8Const* tab) {
  const int n = a.nper;
  for (int e = threadIdx.x; e < a.nblk * 88; e += blockDim.x) {
    const int blk = e / 88, f = e - blk * 88;
    float v = 0.f;
    if (f < 64) {
      const int r = f >> 3, c = f & 7;
      if (r < n && c < n) v = a.A[(size_t)blk * n * n + r * n + c];
    } else {
      const int r = (f - 64) & 7, which = (f - 64) >> 3;
      const int i = blk * n + (r < n ? r : 0);
      if (which == 0) v = a.delays[i];
      else if (which == 1) v = a.inv_gamma ? a.inv_gamma[i] : 1.0f;
      else v = r < n ? a.b[i] : 0.f;
    }
    ((float*)&tab[blk])[f] = v;
  }
}

__device__ __forceinline__ void build8c(float2 (&m)[8][8], const S8Const& cst, int n, bool swap,
                                        const float2 (&zeta)[8], bool adj) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float av = (swap != adj) ? cst.A[c * 8 + r] : cst.A[r * 8 + c];
      m[r][c] = make_float2(-av, 0.f);
      if (r == c) {
        if (r < n) m[r][c] = make_float2(zeta[r].x - av, adj ? -zeta[r].y : zeta[r].y);
        else m[r][c] = make_float2(1.f, 0.f);
      }
    }
}

// x = m^-1 r  (m, r destroyed)
__device__ __forceinline__ void lu8(float2 (&m)[8][8], float2 (&r)[8], float2 (&x)[8]) {
  float2 inv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float best = m[j][j].x * m[j][j].x + m[j][j].y * m[j][j].y;
    int bi = j;
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      const float mg = m[i][j].x * m[i][j].x + m[i][j].y * m[i][j].y;
      if (mg > best) { best = mg; bi = i; }
    }
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      const bool sw = bi == i;
#pragma unroll
      for (int c = j; c < 8; ++c) {
        const float2 t = m[j][c];
        m[j][c] = sw ? m[i][c] : t;
        m[i][c] = sw ? t : m[i][c];
      }
      const float2 t = r[j];
      r[j] = sw ? r[i] : t;
      r[i] = sw ? t : r[i];
    }
    inv[j] = cinv(m[j][j]);
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      const float2 f = cmul(m[i][j], inv[j]);
#pragma unroll
      for (int c = j + 1; c < 8; ++c) {
        m[i][c].x -= f.x * m[j][c].x - f.y * m[j][c].y;
        m[i][c].y -= f.x * m[j][c].y + f.y * m[j][c].x;
      }
      r[i].x -= f.x * r[j].x - f.y * r[j].y;
      r[i].y -= f.x * r[j].y + f.y * r[j].x;
    }
  }
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    float2 sacc = r[j];
#pragma unroll
    for (int c = j + 1; c < 8; ++c) {
      sacc.x -= m[j][c].x * x[c].x - m[j][c].y * x[c].y;
      sacc.y -= m[j][c].x * x[c].y + m[j][c].y * x[c].x;
    }
    x[j] = cmul(sacc, inv[j]);
  }
}

__global__ __launch_bounds__(256) void k_solve8_fwd(SolveArgs a, float2* __restrict__ Y) {
  __shared__ S8Const tab[S8_MAXBLK];
  s8_stage(a, tab);
  __syncthreads();
  const int n = a.nper;
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= (long long)a.K * a.nblk) return;
  const int k = (int)(w / a.nblk), blk = (int)(w - (long long)k * a.nblk);
  const S8Const& cst = tab[blk];
  float2 zeta[8], rhs[8], m[8][8], y[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
    if (r < n) zeta[r] = zeta_abs(a, k, blk * n + r, zeta[r]);
    rhs[r] = make_float2(cst.b[r], 0.f);
  }
  build8c(m, cst, n, a.transpose != 0, zeta, false);
  lu8(m, rhs, y);
  float2* out = Y + (size_t)w * n;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (r < n) out[r] = y[r];
}

// partial layout as k_solve4_bwd; accumulators acc[e][thread] in LDS, e < 80 (8 x 8 | 8 | 8)
#ifndef S8_BWD_T
#define S8_BWD_T 128
#endif
#ifndef S8_BWD_MINW
#define S8_BWD_MINW 2
#endif
__global__ __launch_bounds__(S8_BWD_T, S8_BWD_MINW) void k_solve8_bwd(SolveArgs a, const float2* __restrict__ gY,
                                                    const float2* __restrict__ Ysaved,
                                                    float* __restrict__ partial, int items) {
  __shared__ S8Const tab[S8_MAXBLK];
  extern __shared__ float s8_acc[];          // [80][S8_BWD_T]
  s8_stage(a, tab);
  for (int e = 0; e < 80; ++e) s8_acc[e * S8_BWD_T + threadIdx.x] = 0.f;
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool tr = a.transpose != 0;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S8Const& cst = tab[blk];
  float* acc = s8_acc + threadIdx.x;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    // the forward solution comes from the caller (Ysaved is mandatory here) and is loaded only AFTER the adjoint
    // solve; z^m is recovered from the diagonal term (zeta / inv_gamma): the elimination has the registers.
    // (The fence keeps the block's 88 LDS constants from being hoisted out of the loop into registers.)
    asm volatile("" ::: "memory");
    float2 zeta[8], w[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
      if (r < n) zeta[r] = zeta_abs(a, k, blk * n + r, zeta[r]);
    }
    const size_t row = ((size_t)k * nblk + blk) * n;
    {
      float2 m[8][8], rhs[8];
      build8c(m, cst, n, tr, zeta, true);
#pragma unroll
      for (int r = 0; r < 8; ++r) rhs[r] = r < n ? gY[row + r] : make_float2(0.f, 0.f);
      lu8(m, rhs, w);
    }
    float2 y[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) y[r] = r < n ? Ysaved[row + r] : make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float2 p = tr ? y[i] : w[i], q = tr ? w[j] : y[j];
        acc[(i * 8 + j) * S8_BWD_T] += p.x * q.x + p.y * q.y;
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[(64 + i) * S8_BWD_T] += w[i].x;
      const float2 yz = cmul(y[i], cscale(zeta[i], 1.0f / cst.ig[i]));     // y_i z^m  (g inv_gamma_i = -Re(conj(w_i) y_i z^m))
      acc[(72 + i) * S8_BWD_T] -= w[i].x * yz.x + w[i].y * yz.y;
    }
  }
  __syncthreads();
  const int per = n * n + 2 * n;
  for (int o = threadIdx.x; o < nblk * per; o += S8_BWD_T) {
    const int bq = o / per, e = o - bq * per;
    int src;
    if (e < n * n) src = (e / n) * 8 + (e % n);
    else if (e < n * n + n) src = 64 + (e - n * n);
    else src = 72 + (e - n * n - n);
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s8_acc[src * S8_BWD_T + t];
    partial[((size_t)blockIdx.x * nblk + bq) * per + e] = sum;
  }
}

__global__ __launch_bounds__(256, 2) void k_subfdn8_energy(SolveArgs a, const float* __restrict__ c,
                                                        float* __restrict__ partial, int items) {
  __shared__ S8Const tab[S8_MAXBLK];
  __shared__ float s_e[256];
  s8_stage(a, tab);
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S8Const& cst = tab[blk];
  float c_i[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) c_i[r] = r < n ? c[blk * n + r] : 0.f;
  float acc = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    asm volatile("" ::: "memory");                     // (LDS constants re-read per system, not kept in registers)
    float2 zeta[8], m[8][8], y[8], rhs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
      rhs[r] = make_float2(cst.b[r], 0.f);
    }
    build8c(m, cst, n, a.transpose != 0, zeta, false);
    lu8(m, rhs, y);
    float2 sg = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) { sg.x += c_i[r] * y[r].x; sg.y += c_i[r] * y[r].y; }
    acc += sg.x * sg.x + sg.y * sg.y;
  }
  s_e[threadIdx.x] = live ? acc : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < nblk; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}

// number of 256-bin slices a reducing thread-per-system launch cuts K into: enough blocks to fill the
// chip (>= ~1024 with the nblk blocks of the y dimension), at most GFDN_PARTIAL_BLOCKS
#define S4_MAX_PARTS 2048       // partial-sum slots of the reducing thread-per-system launches
static int solve4_items(int nblk) { return (256 / nblk) * nblk; }
static int solve4_parts(int K, int nblk) {
  const int rows = 256 / nblk;                       // bins one workgroup covers per sweep
  const int full = (K + rows - 1) / rows;
  // ~8 bins per thread, at least 1024 workgroups when there is that much work (measured on the 7-band step, same
  // box: 2 bins per thread 0.885 ms, 4: 0.867, 6: 0.865, 8: 0.864 -- fewer partial records for the finish kernels)
  int parts = (K + 8 * rows - 1) / (8 * rows);
  if (parts < 1024) parts = full < 1024 ? full : 1024;
  if (parts > S4_MAX_PARTS) parts = S4_MAX_PARTS;
  return parts;
}

static int pick_np(int nper) {
  if (nper <= 4) return 4;
  if (nper <= 8) return 8;
  if (nper <= 16) return 16;
  return 32;
}

static int check_solve_args(const double* turns, int K, int nblk, int nper, const float* A,
                            const float* delays, const float* ig, const float* b) {
  if (!turns || !A || !delays || !ig || !b) return GFDN_E_BADARG;
  if (K <= 0 || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > GFDN_MAX_BLOCK) return GFDN_E_UNSUPPORTED;
  return 0;
}

static int solve_fwd_run(const double* turns, const double* logr, int K, int nblk, int nper,
                         const float* A, const float* delays, const float* inv_gamma,
                         const float* b, int transpose, float* Y, void* stream, const float2* g_igz,
                         const double* ig64 = nullptr) {
  const bool precise = ig64 != nullptr;
  int rc = check_solve_args(turns, K, nblk, nper, A, delays, precise ? delays : inv_gamma, b);
  if (rc) return rc;
  if (!Y) return GFDN_E_BADARG;
  SolveArgs a{turns, logr, K, nblk, nper, A, delays, inv_gamma, b, transpose, g_igz, ig64};
  const int np = pick_np(nper);
  const int pk = (!precise && nper >= 9 && nper <= 12) ? nper : 0;       // three rows per lane (k_solve_fwd_rl)
  const int spb = pk == 9 ? 4 * (64 / ((pk + RL_R - 1) / RL_R)) : (pk ? 4 * (64 / pk) : 256 / np);
  dim3 grid((K + spb - 1) / spb, nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (pk) {
    switch (pk) {
      case 9: hipLaunchKernelGGL(k_solve_fwd_rl<9>, grid, block, 0, s, a, (float2*)Y); break;
      case 10: hipLaunchKernelGGL(k_solve_fwd_pk<10>, grid, block, 0, s, a, (float2*)Y); break;
      case 11: hipLaunchKernelGGL(k_solve_fwd_pk<11>, grid, block, 0, s, a, (float2*)Y); break;
      default: hipLaunchKernelGGL(k_solve_fwd_pk<12>, grid, block, 0, s, a, (float2*)Y); break;
    }
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (precise) {
    switch (np) {
      case 4: hipLaunchKernelGGL((k_solve_fwd<4, double2>), grid, block, 0, s, a, (float2*)Y); break;
      case 8: hipLaunchKernelGGL((k_solve_fwd<8, double2>), grid, block, 0, s, a, (float2*)Y); break;
      case 16: hipLaunchKernelGGL((k_solve_fwd<16, double2>), grid, block, 0, s, a, (float2*)Y); break;
      default: hipLaunchKernelGGL((k_solve_fwd<32, double2>), grid, block, 0, s, a, (float2*)Y); break;
    }
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (np == 4 && nblk <= S4_MAXBLK) {
    const long long items = (long long)K * nblk;
    hipLaunchKernelGGL(k_solve4_fwd, dim3((unsigned)((items + 255) / 256)), block, 0, s, a, (float2*)Y);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (np == 8 && nblk <= S8_MAXBLK) {
    const long long items = (long long)K * nblk;
    hipLaunchKernelGGL(k_solve8_fwd, dim3((unsigned)((items + 255) / 256)), block, 0, s, a, (float2*)Y);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  switch (np) {
    case 4: hipLaunchKernelGGL(k_solve_fwd<4>, grid, block, 0, s, a, (float2*)Y); break;
    case 8: hipLaunchKernelGGL(k_solve_fwd<8>, grid, block, 0, s, a, (float2*)Y); break;
    case 16: hipLaunchKernelGGL(k_solve_fwd<16>, grid, block, 0, s, a, (float2*)Y); break;
    default: hipLaunchKernelGGL(k_solve_fwd<32>, grid, block, 0, s, a, (float2*)Y); break;
  }
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_solve_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                              const float* A, const float* delays, const float* inv_gamma,
                              const float* b, int transpose, float* Y, void* stream) {
  return solve_fwd_run(turns, logr, K, nblk, nper, A, delays, inv_gamma, b, transpose, Y, stream, nullptr);
}
// float64 matrix entries and elimination (ill-conditioned systems: the lossless prototype); same arguments
extern "C" int gfdn_solve_precise_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                                      const float* A, const float* delays, const double* inv_gamma_f64,
                                      const float* b, int transpose, float* Y, void* stream) {
  if (!inv_gamma_f64) return GFDN_E_BADARG;
  return solve_fwd_run(turns, logr, K, nblk, nper, A, delays, nullptr, b, transpose, Y, stream, nullptr,
                       inv_gamma_f64);
}
// frequency-dependent absorption: ones (N) = a device vector of ones (the scalar gains are folded into igz)
extern "C" int gfdn_solve_absorb_fwd(const double* turns, const double* logr, int K, int nblk, int nper,
                                     const float* A, const float* delays, const float* ones,
                                     const float* inv_gamma_bins_c64, const float* b, int transpose, float* Y,
                                     void* stream) {
  if (!inv_gamma_bins_c64) return GFDN_E_BADARG;
  return solve_fwd_run(turns, logr, K, nblk, nper, A, delays, ones, b, transpose, Y, stream,
                       (const float2*)inv_gamma_bins_c64);
}

extern "C" size_t gfdn_solve_bwd_work_bytes(int nblk, int nper) {
  const int parts = nper <= 8 ? S4_MAX_PARTS : GFDN_PARTIAL_BLOCKS;
  return (size_t)parts * nblk * (nper * nper + 2 * nper) * sizeof(float);
}

static int solve_bwd_run(const double* turns, const double* logr, int K, int nblk, int nper,
                         const float* A, const float* delays, const float* inv_gamma,
                         const float* b, int transpose, const float* gY, const float* Y,
                         float* gA, float* gb, float* ginv_gamma, void* work, void* stream,
                         const float2* g_igz, const double* ig64 = nullptr);

extern "C" int gfdn_solve_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                              const float* A, const float* delays, const float* inv_gamma,
                              const float* b, int transpose, const float* gY, const float* Y,
                              float* gA, float* gb, float* ginv_gamma, void* work, void* stream) {
  return solve_bwd_run(turns, logr, K, nblk, nper, A, delays, inv_gamma, b, transpose, gY, Y, gA, gb, ginv_gamma,
                       work, stream, nullptr);
}
extern "C" int gfdn_solve_precise_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                                      const float* A, const float* delays, const double* inv_gamma_f64,
                                      const float* b, int transpose, const float* gY, const float* Y,
                                      float* gA, float* gb, float* ginv_gamma, void* work, void* stream) {
  if (!inv_gamma_f64) return GFDN_E_BADARG;
  return solve_bwd_run(turns, logr, K, nblk, nper, A, delays, nullptr, b, transpose, gY, Y, gA, gb, ginv_gamma,
                       work, stream, nullptr, inv_gamma_f64);
}
// (the absorption filters are fixed: ginv_scratch (N) receives a by-product without meaning)
extern "C" int gfdn_solve_absorb_bwd(const double* turns, const double* logr, int K, int nblk, int nper,
                                     const float* A, const float* delays, const float* ones,
                                     const float* inv_gamma_bins_c64, const float* b, int transpose,
                                     const float* gY, const float* Y, float* gA, float* gb,
                                     float* ginv_scratch, void* work, void* stream) {
  if (!inv_gamma_bins_c64) return GFDN_E_BADARG;
  return solve_bwd_run(turns, logr, K, nblk, nper, A, delays, ones, b, transpose, gY, Y, gA, gb, ginv_scratch,
                       work, stream, (const float2*)inv_gamma_bins_c64);
}

static int solve_bwd_run(const double* turns, const double* logr, int K, int nblk, int nper,
                         const float* A, const float* delays, const float* inv_gamma,
                         const float* b, int transpose, const float* gY, const float* Y,
                         float* gA, float* gb, float* ginv_gamma, void* work, void* stream,
                         const float2* g_igz, const double* ig64) {
  const bool precise = ig64 != nullptr;
  int rc = check_solve_args(turns, K, nblk, nper, A, delays, precise ? delays : inv_gamma, b);
  if (rc) return rc;
  if (!gY || !gA || !gb || !ginv_gamma || !work) return GFDN_E_BADARG;
  SolveArgs a{turns, logr, K, nblk, nper, A, delays, inv_gamma, b, transpose, g_igz, ig64};
  const int np = pick_np(nper);
  const int pk = (!precise && nper >= 9 && nper <= 12) ? nper : 0;       // three rows per lane (k_solve_bwd_rl)
  const int spb = pk == 9 ? 4 * (64 / ((pk + RL_R - 1) / RL_R)) : (pk ? 4 * (64 / pk) : 256 / np);
  int nparts = (K + spb - 1) / spb;
  if (nparts > GFDN_PARTIAL_BLOCKS) nparts = GFDN_PARTIAL_BLOCKS;
  const bool lin = !precise && np == 4 && nblk <= S4_MAXBLK;
  const bool lin8 = !precise && np == 8 && nblk <= S8_MAXBLK && Y != nullptr;   // (needs the saved solution)
  if (lin || lin8) nparts = solve4_parts(K, nblk);
  dim3 grid(nparts, nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)work;
  if (lin) {
    const int items = solve4_items(nblk);
    hipLaunchKernelGGL(k_solve4_bwd, dim3(nparts), block, (size_t)items * 25 * sizeof(float), s, a,
                       (const float2*)gY, (const float2*)Y, partial, items);
    GFDN_LAUNCH_CHECK();
    const int tot4 = nblk * (nper * nper + 2 * nper);
    hipLaunchKernelGGL(k_solve_bwd_finish, dim3(tot4), dim3(256), 0, s, partial, nparts, nblk, nper, gA, gb, ginv_gamma);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (lin8) {
    const size_t lds8 = (size_t)80 * S8_BWD_T * sizeof(float);
    if ((rc = ensure_dyn_lds(k_solve8_bwd, lds8))) return rc;
    hipLaunchKernelGGL(k_solve8_bwd, dim3(nparts), dim3(S8_BWD_T), lds8, s, a, (const float2*)gY,
                       (const float2*)Y, partial, (S8_BWD_T / nblk) * nblk);
    GFDN_LAUNCH_CHECK();
    const int tot8 = nblk * (nper * nper + 2 * nper);
    hipLaunchKernelGGL(k_solve_bwd_finish, dim3(tot8), dim3(256), 0, s, partial, nparts, nblk, nper, gA, gb, ginv_gamma);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (pk) {
    switch (pk) {
      case 9:
        if (Y) hipLaunchKernelGGL((k_solve_bwd_rl<9, true>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial);
        else hipLaunchKernelGGL((k_solve_bwd_rl<9, false>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial);
        break;
      case 10: hipLaunchKernelGGL(k_solve_bwd_pk<10>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
      case 11: hipLaunchKernelGGL(k_solve_bwd_pk<11>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
      default: hipLaunchKernelGGL(k_solve_bwd_pk<12>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
    }
  } else if (precise) {
    switch (np) {
      case 4: hipLaunchKernelGGL((k_solve_bwd<4, double2>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
      case 8: hipLaunchKernelGGL((k_solve_bwd<8, double2>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
      case 16: hipLaunchKernelGGL((k_solve_bwd<16, double2>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
      default: hipLaunchKernelGGL((k_solve_bwd<32, double2>), grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
    }
  } else
  switch (np) {
    case 4: hipLaunchKernelGGL(k_solve_bwd<4>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
    case 8: hipLaunchKernelGGL(k_solve_bwd<8>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
    case 16: hipLaunchKernelGGL(k_solve_bwd<16>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
    default: hipLaunchKernelGGL(k_solve_bwd<32>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial); break;
  }
  GFDN_LAUNCH_CHECK();
  const int tot = nblk * (nper * nper + 2 * nper);
  hipLaunchKernelGGL(k_solve_bwd_finish, dim3(tot), dim3(256), 0, s, partial, nparts,
                     nblk, nper, gA, gb, ginv_gamma);
  GFDN_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------
// FILTER coupling (feedback_loop.py:362-373, :441-455): the inter-group coupling is a paraunitary FIR matrix
// Phi(z) (G x G, order P), so the feedback matrix is frequency dependent,
//     A(z_k)[i][j] = BM[i][j] * Phi_k[g(i)][g(j)],    BM = block mixing matrix (real, N x N),
// with Phi_k = sum_p Phi_p z_k^-p (K, G, G) complex64 evaluated by the caller.  One dense N x N system per
// bin on NP lanes as k_solve_fwd / k_solve_bwd; the backward also returns dL/dPhi_k per bin.
// ------------------------------------------------------------------------------------------
struct PhiArgs {
  const double* turns;
  const double* logr;
  int K, N, nper;              // N = G * nper
  const float* BM;             // (N, N)
  const float2* Phi;           // (K, G, G)
  const float* delays;
  const float* inv_gamma;
  const float* b;
  const float2* ig_bins;       // (K, N) complex 1 / Gamma_i(z_k) of absorption FILTERS (feedback_loop.py:332-344,
                               // :376-381), multiplying inv_gamma; NULL: scalar gains only
};

// diagonal entry z_k^{m_i} inv_gamma_i [ / Gamma_i(z_k) ]
__device__ __forceinline__ float2 phi_zeta(const PhiArgs& a, int k, int i, float m, float ig) {
  const float2 z = zeta_pow(a.turns, a.logr, k, m, ig);
  return a.ig_bins ? cmul(z, a.ig_bins[(size_t)k * a.N + i]) : z;
}

template <int NP>
__device__ __forceinline__ void build_row_phi(float2 (&row)[NP], const PhiArgs& a, int k, int r, bool adj,
                                              float2 diag) {
  const int N = a.N, G = N / a.nper;
  const float2* Ph = a.Phi + (size_t)k * G * G;
  const int gr = (r < N ? r : 0) / a.nper;
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    float2 v = make_float2(0.f, 0.f);
    if (c < N && r < N) {
      const int gc = c / a.nper;
      // T = D - A;  adjoint: row r of T^H = conj of column r of T
      const float2 ph = adj ? cconj(Ph[gc * G + gr]) : Ph[gr * G + gc];
      const float bm = adj ? a.BM[c * N + r] : a.BM[r * N + c];
      v = make_float2(-bm * ph.x, -bm * ph.y);
    }
    if (c == r) v = make_float2(v.x + diag.x, v.y + diag.y);
    row[c] = v;
  }
}

template <int NP>
__global__ __launch_bounds__(256) void k_solve_phi_fwd(PhiArgs a, float2* __restrict__ Y) {
  constexpr int SPB = 256 / NP;
  const int r = threadIdx.x % NP, grp = threadIdx.x / NP;
  const int N = a.N;
  const int k = blockIdx.x * SPB + grp;
  const bool valid = k < a.K;
  const int kk = valid ? k : a.K - 1;
  const int i = r < N ? r : 0;
  const float2 zeta = phi_zeta(a, kk, i, a.delays[i], a.inv_gamma[i]);
  float2 row[NP];
  build_row_phi<NP>(row, a, kk, r, false, zeta);
  int pivcol;
  const float2 y = gauss_jordan<NP>(row, make_float2(r < N ? a.b[i] : 0.f, 0.f), N, r, pivcol);
  if (valid && pivcol >= 0) Y[(size_t)k * N + pivcol] = y;
}

// partial[part][N N + 2 N] (gBM | gb | unused), gPhi (K, G, G) written per bin
template <int NP>
__global__ __launch_bounds__(256) void k_solve_phi_bwd(PhiArgs a, const float2* __restrict__ gY,
                                                       const float2* __restrict__ Ysaved,
                                                       float* __restrict__ partial,
                                                       float2* __restrict__ gPhi) {
  constexpr int SPB = 256 / NP;
  __shared__ float2 s_perm[256];
  __shared__ float s_acc[256 * (NP + 2)];
  __shared__ float2 s_phi[256 * GFDN_MAX_GROUPS];
  const int r = threadIdx.x % NP, grp = threadIdx.x / NP;
  const int N = a.N, G = N / a.nper;
  const bool active = r < N;
  const int i = active ? r : 0;
  const int gr = i / a.nper;
  const float m_i = a.delays[i], ig_i = a.inv_gamma[i];
  float acc[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) acc[c] = 0.f;
  float accb = 0.f, accg = 0.f;
  for (int k0 = blockIdx.x * SPB; k0 < a.K; k0 += gridDim.x * SPB) {
    const int k = k0 + grp;
    const bool valid = k < a.K;
    const int kk = valid ? k : a.K - 1;
    const float2 zeta = phi_zeta(a, kk, i, m_i, ig_i);
    const float2 ynat = active ? Ysaved[(size_t)kk * N + i] : make_float2(0.f, 0.f);
    float2 row[NP];
    int pivcol;
    build_row_phi<NP>(row, a, kk, r, true, cconj(zeta));
    const float2 g = active ? gY[(size_t)kk * N + i] : make_float2(0.f, 0.f);
    const float2 w = gauss_jordan<NP>(row, g, N, r, pivcol);
    __syncthreads();
    if (pivcol >= 0) s_perm[grp * NP + pivcol] = w;
    __syncthreads();
    float2 wnat = active ? s_perm[grp * NP + r] : make_float2(0.f, 0.f);
    if (!valid) wnat = make_float2(0.f, 0.f);
    // gA[r][c] = w_r conj(y_c)  (complex);  gBM[r][c] += Re(conj(Phi[g(r)][g(c)]) gA[r][c]);
    // gPhi[g(r)][g'] += sum_{c in g'} BM[r][c] gA[r][c]   (summed over the rows of group g(r) below)
    const float2* Ph = a.Phi + (size_t)kk * G * G;
    float2 sphi[GFDN_MAX_GROUPS];
#pragma unroll
    for (int q = 0; q < GFDN_MAX_GROUPS; ++q) sphi[q] = make_float2(0.f, 0.f);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      if (c < N) {
        const float yx = __shfl(ynat.x, c, NP), yy = __shfl(ynat.y, c, NP);
        const float2 ga = make_float2(wnat.x * yx + wnat.y * yy, wnat.y * yx - wnat.x * yy);   // w conj(y)
        const int gc = c / a.nper;
        const float2 ph = Ph[gr * G + gc];
        acc[c] += ph.x * ga.x + ph.y * ga.y;
        const float bm = active ? a.BM[i * N + c] : 0.f;
#pragma unroll
        for (int q = 0; q < GFDN_MAX_GROUPS; ++q)
          if (q == gc) { sphi[q].x += bm * ga.x; sphi[q].y += bm * ga.y; }
      }
    }
    accb += wnat.x;
    {  // g inv_gamma_i = -Re(conj(w_i) y_i z^m [/ Gamma_i(z_k)]), as k_solve_bwd
      const float2 yz = cmul(ynat, phi_zeta(a, kk, i, m_i, 1.0f));
      accg -= wnat.x * yz.x + wnat.y * yz.y;
    }
#pragma unroll
    for (int q = 0; q < GFDN_MAX_GROUPS; ++q) s_phi[(grp * NP + r) * GFDN_MAX_GROUPS + q] = sphi[q];
    __syncthreads();
    if (valid) {
      for (int e = r; e < G * G; e += NP) {
        const int gq = e / G, gp = e - gq * G;
        float2 t = make_float2(0.f, 0.f);
        for (int rr = gq * a.nper; rr < (gq + 1) * a.nper; ++rr) t = cadd(t, s_phi[(grp * NP + rr) * GFDN_MAX_GROUPS + gp]);
        gPhi[(size_t)k * G * G + e] = t;
      }
    }
    __syncthreads();
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NP; ++c) s_acc[(grp * NP + r) * (NP + 2) + c] = acc[c];
  s_acc[(grp * NP + r) * (NP + 2) + NP] = accb;
  s_acc[(grp * NP + r) * (NP + 2) + NP + 1] = accg;
  __syncthreads();
  const int per = N * N + 2 * N;
  float* out = partial + (size_t)blockIdx.x * per;
  for (int e = threadIdx.x; e < per; e += blockDim.x) {
    int rr, cc;
    if (e < N * N) { rr = e / N; cc = e % N; }
    else if (e < N * N + N) { rr = e - N * N; cc = NP; }
    else { rr = e - N * N - N; cc = NP + 1; }
    float sum = 0.f;
    for (int g2 = 0; g2 < SPB; ++g2) sum += s_acc[(g2 * NP + rr) * (NP + 2) + cc];
    out[e] = sum;
  }
}

static int phi_args_ok(const double* turns, int K, int G, int nper, const float* BM, const float* Phi,
                       const float* delays, const float* ig, const float* b) {
  if (!turns || !BM || !Phi || !delays || !ig || !b || K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (G > GFDN_MAX_GROUPS || G * nper > GFDN_MAX_BLOCK) return GFDN_E_UNSUPPORTED;
  return 0;
}

extern "C" int gfdn_solve_phi_absorb_fwd(const double* turns, const double* logr, int K, int G, int nper,
                                         const float* BM, const float* Phi_c64, const float* delays,
                                         const float* inv_gamma, const float* inv_gamma_bins_c64, const float* b,
                                         float* Y, void* stream) {
  int rc = phi_args_ok(turns, K, G, nper, BM, Phi_c64, delays, inv_gamma, b);
  if (rc) return rc;
  if (!Y) return GFDN_E_BADARG;
  PhiArgs a{turns, logr, K, G * nper, nper, BM, (const float2*)Phi_c64, delays, inv_gamma, b,
            (const float2*)inv_gamma_bins_c64};
  const int np = pick_np(G * nper), spb = 256 / np;
  dim3 grid((K + spb - 1) / spb), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (np) {
    case 4: hipLaunchKernelGGL(k_solve_phi_fwd<4>, grid, block, 0, s, a, (float2*)Y); break;
    case 8: hipLaunchKernelGGL(k_solve_phi_fwd<8>, grid, block, 0, s, a, (float2*)Y); break;
    case 16: hipLaunchKernelGGL(k_solve_phi_fwd<16>, grid, block, 0, s, a, (float2*)Y); break;
    default: hipLaunchKernelGGL(k_solve_phi_fwd<32>, grid, block, 0, s, a, (float2*)Y); break;
  }
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_solve_phi_fwd(const double* turns, const double* logr, int K, int G, int nper,
                                  const float* BM, const float* Phi_c64, const float* delays,
                                  const float* inv_gamma, const float* b, float* Y, void* stream) {
  return gfdn_solve_phi_absorb_fwd(turns, logr, K, G, nper, BM, Phi_c64, delays, inv_gamma, nullptr, b, Y, stream);
}

extern "C" size_t gfdn_solve_phi_bwd_work_bytes(int G, int nper) {
  const int N = G * nper;
  return (size_t)GFDN_PARTIAL_BLOCKS * (N * N + 2 * N) * sizeof(float);
}

extern "C" int gfdn_solve_phi_absorb_bwd(const double* turns, const double* logr, int K, int G, int nper,
                                         const float* BM, const float* Phi_c64, const float* delays,
                                         const float* inv_gamma, const float* inv_gamma_bins_c64, const float* b,
                                         const float* gY, const float* Y, float* gBM, float* gb, float* ginv_gamma,
                                         float* gPhi_c64, void* work, void* stream) {
  int rc = phi_args_ok(turns, K, G, nper, BM, Phi_c64, delays, inv_gamma, b);
  if (rc) return rc;
  if (!gY || !Y || !gBM || !gb || !ginv_gamma || !gPhi_c64 || !work) return GFDN_E_BADARG;
  const int N = G * nper;
  PhiArgs a{turns, logr, K, N, nper, BM, (const float2*)Phi_c64, delays, inv_gamma, b,
            (const float2*)inv_gamma_bins_c64};
  const int np = pick_np(N), spb = 256 / np;
  int nparts = (K + spb - 1) / spb;
  if (nparts > GFDN_PARTIAL_BLOCKS) nparts = GFDN_PARTIAL_BLOCKS;
  dim3 grid(nparts), block(256);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)work;
  switch (np) {
    case 4: hipLaunchKernelGGL(k_solve_phi_bwd<4>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial, (float2*)gPhi_c64); break;
    case 8: hipLaunchKernelGGL(k_solve_phi_bwd<8>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial, (float2*)gPhi_c64); break;
    case 16: hipLaunchKernelGGL(k_solve_phi_bwd<16>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial, (float2*)gPhi_c64); break;
    default: hipLaunchKernelGGL(k_solve_phi_bwd<32>, grid, block, 0, s, a, (const float2*)gY, (const float2*)Y, partial, (float2*)gPhi_c64); break;
  }
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_solve_bwd_finish, dim3(N * N + 2 * N), dim3(256), 0, s, partial, nparts, 1, N, gBM, gb, ginv_gamma);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_solve_phi_bwd(const double* turns, const double* logr, int K, int G, int nper,
                                  const float* BM, const float* Phi_c64, const float* delays,
                                  const float* inv_gamma, const float* b, const float* gY, const float* Y,
                                  float* gBM, float* gb, float* ginv_gamma, float* gPhi_c64, void* work,
                                  void* stream) {
  return gfdn_solve_phi_absorb_bwd(turns, logr, K, G, nper, BM, Phi_c64, delays, inv_gamma, nullptr, b, gY, Y, gBM, gb,
                                   ginv_gamma, gPhi_c64, work, stream);
}

// ------------------------------------------------------------------------------------------
// normalize (trainer.py:317-332) in two launches: energy of the sub-FDN responses, then the rescale.
//   Hout[k][g] = sum_{n in g} c_n y_n,  y = (diag(z_k^m) - M_g)^{-1} b_g   (model.py:237-250)
//   E_g = mean_k |Hout[k][g]|^2;   b_n, c_n /= E_g^(1/4)  for n in g.
// The solve kernel's lanes already hold y_n: the group sum is a reduction over the NP lanes of a
// system, |.|^2 is accumulated over the block's bins, per-block partials are folded in a fixed order.
// Neither Y (K x N) nor Hout is written.
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void k_subfdn_energy(SolveArgs a, const float* __restrict__ c,
                                                       float* __restrict__ partial) {
  constexpr int SPB = 256 / NP;
  __shared__ float s_red[16];
  const int r = threadIdx.x % NP, grp = threadIdx.x / NP;
  const int blk = blockIdx.y, n = a.nper;
  const int i = blk * n + (r < n ? r : 0);
  const float m_i = a.delays[i], ig_i = a.inv_gamma ? a.inv_gamma[i] : 1.0f;
  const float b_i = r < n ? a.b[i] : 0.f;
  float acc = 0.f;
  for (int k0 = blockIdx.x * SPB; k0 < a.K; k0 += gridDim.x * SPB) {
    const int k = k0 + grp;
    const bool valid = k < a.K;
    const int kk = valid ? k : a.K - 1;
    const float2 zeta = zeta_pow(a.turns, a.logr, kk, m_i, ig_i);
    float2 row[NP];
    build_row<NP>(row, a.A + (size_t)blk * n * n, n, r, a.transpose != 0, zeta);
    int pivcol;
    const float2 y = gauss_jordan<NP>(row, make_float2(b_i, 0.f), n, r, pivcol);
    float2 s = make_float2(0.f, 0.f);
    if (pivcol >= 0) s = cscale(y, c[blk * n + pivcol]);
#pragma unroll
    for (int o = NP >> 1; o > 0; o >>= 1) {       // sum over the lanes of this system (any order of
      s.x += __shfl_xor(s.x, o, NP);              // columns: the butterfly is symmetric)
      s.y += __shfl_xor(s.y, o, NP);
    }
    if (valid && r == 0) acc += s.x * s.x + s.y * s.y;
  }
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) partial[(size_t)blk * gridDim.x + blockIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_subfdn_rescale(const float* __restrict__ partial, int nparts,
                                                       int K, int nper, float* __restrict__ b,
                                                       float* __restrict__ c, float* __restrict__ energy) {
  const int g = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)g * nparts + p];
  s = wave_sum(s);
  const float E = s / (float)K;
  if (energy && threadIdx.x == 0) energy[g] = E;
  const float d = powf(E, 0.25f);
  for (int i = threadIdx.x; i < nper; i += 64) {
    b[g * nper + i] /= d;
    c[g * nper + i] /= d;
  }
}


// ------------------------------------------------------------------------------------------
// Colorless side branch in three launches (n <= 4): the un-damped sub-FDN responses feed BOTH
// Trainer.normalize (trainer.py:317-332: E_g = mean_k |Hout[k, g]|^2, b, c /= E_g^(1/4)) and the spectral
// loss of the step that follows (trainer.py:298-304) -- and since y = (D - M_g)^-1 b is linear in b, the
// responses after the rescale are the ones before it times a per-group constant:
//      y'_n = y_n / d_g,   Hout'[k, g] = Hout[k, g] / d_g^2,   d_g = E_g^(1/4).
// So ONE solve per step serves both (the step used to solve the sub-FDNs twice, write Y, read it back in
// the output stage, and run the generic output-stage backward on G "receivers"):
//   forward : y (saved, raw), group sums S[k][g] (bin-major, raw), energy partials        -> rescale
//   stats   : loss_g and dL/dS' on S' = S / d_g^2
//   backward: adjoint solve with rhs c'_n dL/dS', on y' = y / d_g: dL/dM_g, dL/db', dL/dc'
// `energy` == NULL everywhere means "no rescale happened" (d_g = 1): the plain colorless forward.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_subfdn4_fwd(SolveArgs a, const float* __restrict__ c,
                                                     float2* __restrict__ Y, float2* __restrict__ S,
                                                     float* __restrict__ partial, int items) {
  __shared__ S4Const tab[S4_MAXBLK];
  __shared__ float s_e[256];
  s4_stage(a, tab);
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S4Const cst = tab[blk];
  float c_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c_i[r] = r < n ? c[blk * n + r] : 0.f;
  float acc = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    float2 zeta[4], m[4][4], y[4], rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
      rhs[r] = make_float2(cst.b[r], 0.f);
    }
    build4c(m, cst, n, false, zeta, false);
    gj4(m, rhs, y);
    const size_t w = (size_t)k * nblk + blk;
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r < n) Y[w * n + r] = y[r];
      s.x += c_i[r] * y[r].x;
      s.y += c_i[r] * y[r].y;
    }
    S[w] = s;
    acc += s.x * s.x + s.y * s.y;
  }
  s_e[threadIdx.x] = live ? acc : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < nblk; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}

// energy[g] = sum_p partial[g][p] / K;  with b, c given: b_n, c_n /= energy_g^(1/4) in place
__global__ __launch_bounds__(256) void k_subfdn_energy_finish(const float* __restrict__ partial, int nparts,
                                                              int K, int nper, float* __restrict__ b,
                                                              float* __restrict__ c,
                                                              float* __restrict__ energy) {
  __shared__ float s_red[16];
  const int g = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)g * nparts + p];
  s = block_sum(s, s_red);
  const float E = s / (float)K;
  if (energy && threadIdx.x == 0) energy[g] = E;
  if (b && c) {
    const float d = powf(E, 0.25f);
    for (int i = threadIdx.x; i < nper; i += 256) {
      b[g * nper + i] /= d;
      c[g * nper + i] /= d;
    }
  }
}

// S (K, G) bin-major, raw; S' = S / sqrt(energy_g) (energy NULL: S' = S).
//   loss partial[g][part] = sum over this block's bins of (|S'| - 1)^p / K;  gS[k][g] = scale dloss_g/dS'
__global__ __launch_bounds__(256) void k_spectral_stats_bm(const float2* __restrict__ S, int G, int K,
                                                           const float* __restrict__ energy, int asym,
                                                           float scale, float* __restrict__ partial,
                                                           float2* __restrict__ gS, int items) {
  __shared__ float s_e[256];
  const bool live = (int)threadIdx.x < items;
  const int g = live ? threadIdx.x % G : 0;
  const int krow = threadIdx.x / G, rows = items / G;
  const float sc = energy ? 1.0f / sqrtf(energy[g]) : 1.0f;     // 1 / d_g^2
  const float invK = 1.0f / (float)K;
  float l = 0.f;
  for (int k = blockIdx.x * rows + krow; live && k < K; k += gridDim.x * rows) {
    const size_t w = (size_t)k * G + g;
    float2 s = S[w];
    s.x *= sc; s.y *= sc;
    const float p = s.x * s.x + s.y * s.y;
    const float mag = sqrtf(p);
    const float d = mag - 1.0f;
    const float d2 = d * d;
    const bool four = asym && (d > 1.0f);
    l += four ? d2 * d2 : d2;
    if (gS) {
      const float dl = four ? 4.0f * d2 * d : 2.0f * d;   // d loss / d |S'|
      const float f = (mag > 0.f) ? scale * invK * dl / mag : 0.f;
      gS[w] = make_float2(f * s.x, f * s.y);
    }
  }
  s_e[threadIdx.x] = live ? l * invK : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < G; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += G) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}
__global__ __launch_bounds__(256) void k_rows_sum256(const float* __restrict__ partial, int nparts,
                                                     float* __restrict__ out) {
  __shared__ float s_red[16];
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)blockIdx.x * nparts + p];
  s = block_sum(s, s_red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// partial[(part * nblk + blk) * per + e], per = n n + 2 n: [gM | gb | gc]
__global__ __launch_bounds__(256) void k_subfdn4_bwd(SolveArgs a, const float* __restrict__ c,
                                                     const float* __restrict__ energy,
                                                     const float2* __restrict__ Yraw,
                                                     const float2* __restrict__ gS,
                                                     float* __restrict__ partial, int items) {
  __shared__ S4Const tab[S4_MAXBLK];
  extern __shared__ float s4_acc[];          // [items][25]
  s4_stage(a, tab);
  __syncthreads();
  const int n = a.nper, nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  const S4Const cst = tab[blk];
  const float ys = energy ? 1.0f / powf(energy[blk], 0.25f) : 1.0f;      // y' = y / d_g
  float c_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c_i[r] = r < n ? c[blk * n + r] : 0.f;
  float acc[24];
#pragma unroll
  for (int e = 0; e < 24; ++e) acc[e] = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    float2 zeta[4], m[4][4], y[4], w[4], rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zeta[r] = zeta_pow(a.turns, a.logr, k, cst.m[r], cst.ig[r]);
    const size_t wi = (size_t)k * nblk + blk;
    const float2 gs = gS[wi];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      y[r] = r < n ? cscale(Yraw[wi * n + r], ys) : make_float2(0.f, 0.f);
      rhs[r] = cscale(gs, c_i[r]);                     // dL/dy'_r = c'_r dL/dS'
    }
    build4c(m, cst, n, false, zeta, true);             // T^H w = dL/dy'
    gj4(m, rhs, w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] += w[i].x * y[j].x + w[i].y * y[j].y;   // gM_ij = Re(w_i conj(y'_j))
      acc[16 + i] += w[i].x;                                                            // gb'_i = Re(w_i)
      acc[20 + i] += gs.x * y[i].x + gs.y * y[i].y;                                     // gc'_i = Re(conj(gS') y'_i)
    }
  }
  if (live) {
#pragma unroll
    for (int e = 0; e < 24; ++e) s4_acc[threadIdx.x * 25 + e] = acc[e];
  }
  __syncthreads();
  const int per = n * n + 2 * n;
  for (int o = threadIdx.x; o < nblk * per; o += 256) {
    const int bq = o / per, e = o - bq * per;
    int src;
    if (e < n * n) src = (e / n) * 4 + (e % n);
    else if (e < n * n + n) src = 16 + (e - n * n);
    else src = 20 + (e - n * n - n);
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s4_acc[t * 25 + src];
    partial[((size_t)blockIdx.x * nblk + bq) * per + e] = sum;
  }
}

extern "C" size_t gfdn_subfdn_colorless_work_bytes(int G, int nper) {
  return (size_t)S4_MAX_PARTS * G * (nper * nper + 2 * nper) * sizeof(float);
}

extern "C" int gfdn_subfdn_colorless_fwd(const double* turns, const double* logr, int K, int G, int nper,
                                         const float* M, const float* delays, float* b, float* c,
                                         int normalize, float* Y, float* S, float* energy, void* work,
                                         void* stream) {
  if (!turns || !M || !delays || !b || !c || !Y || !S || !energy || !work) return GFDN_E_BADARG;
  if (K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4 || G > S4_MAXBLK) return GFDN_E_UNSUPPORTED;
  SolveArgs a{turns, logr, K, G, nper, M, delays, nullptr, b, 0, nullptr, nullptr};
  const int nparts = solve4_parts(K, G);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_subfdn4_fwd, dim3(nparts), dim3(256), 0, s, a, (const float*)c, (float2*)Y, (float2*)S,
                     (float*)work, solve4_items(G));
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_subfdn_energy_finish, dim3(G), dim3(256), 0, s, (const float*)work, nparts, K, nper,
                     normalize ? b : nullptr, normalize ? c : nullptr, energy);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_spectral_stats_binmajor(const float* S, int G, int K, const float* energy, int asym,
                                            float scale, float* loss, float* gS, void* work, void* stream) {
  if (!S || !loss || !work || G <= 0 || K <= 0) return GFDN_E_BADARG;
  if (G > 256) return GFDN_E_UNSUPPORTED;
  const int nparts = solve4_parts(K, G);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_spectral_stats_bm, dim3(nparts), dim3(256), 0, s, (const float2*)S, G, K, energy, asym,
                     scale, (float*)work, (float2*)gS, solve4_items(G));
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_rows_sum256, dim3(G), dim3(256), 0, s, (const float*)work, nparts, loss);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_subfdn_colorless_bwd(const double* turns, const double* logr, int K, int G, int nper,
                                         const float* M, const float* delays, const float* b,
                                         const float* c, const float* energy, const float* Y,
                                         const float* gS, float* gM, float* gb, float* gc, void* work,
                                         void* stream) {
  if (!turns || !M || !delays || !b || !c || !Y || !gS || !gM || !gb || !gc || !work) return GFDN_E_BADARG;
  if (K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4 || G > S4_MAXBLK) return GFDN_E_UNSUPPORTED;
  SolveArgs a{turns, logr, K, G, nper, M, delays, nullptr, b, 0, nullptr, nullptr};
  const int nparts = solve4_parts(K, G), items = solve4_items(G);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_subfdn4_bwd, dim3(nparts), dim3(256), (size_t)items * 25 * sizeof(float), s, a, c, energy,
                     (const float2*)Y, (const float2*)gS, (float*)work, items);
  GFDN_LAUNCH_CHECK();
  // the record layout [n n | n | n] of k_solve_bwd_finish, with gc in the third slot
  hipLaunchKernelGGL(k_solve_bwd_finish, dim3(G * (nper * nper + 2 * nper)), dim3(256), 0, s, (const float*)work,
                     nparts, G, nper, gM, gb, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gfdn_subfdn_normalize_work_bytes(int G) {
  return (size_t)G * S4_MAX_PARTS * sizeof(float);
}

extern "C" int gfdn_subfdn_normalize(const double* turns, const double* logr, int K, int G, int nper,
                                     const float* M, const float* delays, float* b, float* c,
                                     float* energy, void* work, void* stream) {
  if (!turns || !M || !delays || !b || !c || !work) return GFDN_E_BADARG;
  if (K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > GFDN_MAX_BLOCK) return GFDN_E_UNSUPPORTED;
  SolveArgs a{turns, logr, K, G, nper, M, delays, nullptr, b, 0, nullptr, nullptr};
  const int np = pick_np(nper);
  const int spb = 256 / np;
  int nparts = (K + spb - 1) / spb;
  if (nparts > GFDN_PARTIAL_BLOCKS) nparts = GFDN_PARTIAL_BLOCKS;
  const bool lin = np == 4 && G <= S4_MAXBLK;
  const bool lin8 = np == 8 && G <= S8_MAXBLK;
  if (lin || lin8) nparts = solve4_parts(K, G);
  dim3 grid(nparts, G), block(256);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)work;
  if (lin || lin8) {
    if (lin) hipLaunchKernelGGL(k_subfdn4_energy, dim3(nparts), block, 0, s, a, (const float*)c, partial, solve4_items(G));
    else hipLaunchKernelGGL(k_subfdn8_energy, dim3(nparts), block, 0, s, a, (const float*)c, partial, solve4_items(G));
    GFDN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_subfdn_rescale, dim3(G), dim3(64), 0, s, (const float*)partial, nparts, K, nper, b, c, energy);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  switch (np) {
    case 4: hipLaunchKernelGGL(k_subfdn_energy<4>, grid, block, 0, s, a, (const float*)c, partial); break;
    case 8: hipLaunchKernelGGL(k_subfdn_energy<8>, grid, block, 0, s, a, (const float*)c, partial); break;
    case 16: hipLaunchKernelGGL(k_subfdn_energy<16>, grid, block, 0, s, a, (const float*)c, partial); break;
    default: hipLaunchKernelGGL(k_subfdn_energy<32>, grid, block, 0, s, a, (const float*)c, partial); break;
  }
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_subfdn_rescale, dim3(G), dim3(64), 0, s, (const float*)partial, nparts, K, nper, b, c,
                     energy);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// output stage
// ------------------------------------------------------------------------------------------
#define COMPOSE_BCH 8
extern __shared__ float2 compose_lds[];

// Y tile of 256 bins x N delay lines -> LDS rows of N+1 (global side strictly linear: the per-thread
// 128-byte rows of the bin-major layout would otherwise be read 8 bytes at a time across 64 lines)
// ldy: row stride of Y in elements (N for one band; nbands * N when the bands' delay lines sit side by
// side in one (K, nbands * N) solve output -- each bin then contributes one contiguous run of N values)
#ifndef CF_T
#define CF_T 256          // bins per workgroup of the output stage (128: no faster)
#endif
__device__ __forceinline__ void load_bin_tile(const float2* __restrict__ Y, int k0, int K, int N, int ldy,
                                              float2* tile) {
  const size_t base = (size_t)k0 * ldy;
  const int lim = (K - k0 < CF_T ? K - k0 : CF_T) * N;
  for (int e = threadIdx.x; e < CF_T * N; e += CF_T) {
    const int kk = e / N, n = e - kk * N;
    tile[kk * (N + 1) + n] = e < lim ? Y[base + (size_t)kk * ldy + n] : make_float2(0.f, 0.f);
  }
}

__global__ __launch_bounds__(CF_T) void k_compose_fwd(const float2* __restrict__ Y, int K, int G,
                                                     int nper, const float* __restrict__ c,
                                                     const float* __restrict__ rgain, int B,
                                                     const float2* __restrict__ direct, int ldd,
                                                     const long long* __restrict__ drows,
                                                     const float2* __restrict__ filt,
                                                     float2* __restrict__ H, int ldh,
                                                     float2* __restrict__ S_out, int ldy, int ldf, int rpb) {
  const int N = G * nper;
  const int k0 = blockIdx.x * CF_T;
  {   // band blockIdx.z: its delay lines, gains, items (B per band) and filter row
    const int band = blockIdx.z;
    Y += (size_t)band * N;
    c += (size_t)band * N;
    rgain += (size_t)band * B * G;
    if (drows) drows += (size_t)band * B;
    else if (direct) direct += (size_t)band * B * ldd;
    if (filt) filt += (size_t)band * ldf;
    H += (size_t)band * B * ldh;
    if (S_out) S_out += (size_t)band * G * K;
  }
  load_bin_tile(Y, k0, K, N, ldy, compose_lds);
  __syncthreads();
  const int k = k0 + threadIdx.x;
  if (k >= K) return;
  const float2* yrow = compose_lds + threadIdx.x * (N + 1);
  float2 S[GFDN_MAX_GROUPS];
#pragma unroll
  for (int g = 0; g < GFDN_MAX_GROUPS; ++g) {
    S[g] = make_float2(0.f, 0.f);
    if (g < G) {
      for (int i = 0; i < nper; ++i) {
        const float2 y = yrow[g * nper + i];
        const float cc = c[g * nper + i];
        S[g].x += cc * y.x;
        S[g].y += cc * y.y;
      }
      if (S_out && blockIdx.y == 0) S_out[(size_t)g * K + k] = S[g];
    }
  }
  const float2 f = filt ? filt[k] : make_float2(1.f, 0.f);
  // receivers [blockIdx.y * rpb, + rpb) in chunks of COMPOSE_BCH: with many (bin tile, band) units in the
  // launch one workgroup serves ALL receivers of its band, so the Y tile is staged once, not B / 8 times
  const int bend = (int)(blockIdx.y + 1) * rpb < B ? (int)(blockIdx.y + 1) * rpb : B;
  for (int b0 = blockIdx.y * rpb; b0 < bend; b0 += COMPOSE_BCH) {
    float2 d[COMPOSE_BCH];
#pragma unroll
    for (int bb = 0; bb < COMPOSE_BCH; ++bb) {       // all loads of the chunk in flight together
      const int b = b0 + bb;
      d[bb] = (direct && b < bend) ? direct[(size_t)(drows ? drows[b] : b) * ldd + k] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int bb = 0; bb < COMPOSE_BCH; ++bb) {
      const int b = b0 + bb;
      if (b < bend) {
        float2 h = d[bb];
#pragma unroll
        for (int g = 0; g < GFDN_MAX_GROUPS; ++g) {
          if (g < G) {
            const float rg = rgain[b * G + g];
            h.x += rg * S[g].x;
            h.y += rg * S[g].y;
          }
        }
        if (filt) h = cmul(h, f);
        H[(size_t)b * ldh + k] = h;
      }
    }
  }
}

static size_t compose_tile_bytes(int N) { return (size_t)CF_T * (N + 1) * sizeof(float2); }

extern "C" int gfdn_compose_banded_fwd(const float* Y, int K, int nbands, int G, int nper, const float* c,
                                       const float* rgain, int B, const float* direct, int ldd,
                                       const long long* direct_rows, const float* filt, int ldf,
                                       float* H, int ldh, float* S_out, void* stream) {
  if (!Y || !c || !rgain || !H || K <= 0 || G <= 0 || nper <= 0 || B <= 0 || nbands <= 0) return GFDN_E_BADARG;
  if (G > GFDN_MAX_GROUPS || nbands > 65535) return GFDN_E_UNSUPPORTED;
  if (ldh < K || (direct && ldd < K) || (filt && nbands > 1 && ldf < K)) return GFDN_E_BADARG;
  // receivers per workgroup: all of the band's when (bin tiles x bands) alone fills the chip
  const int ktiles = (K + CF_T - 1) / CF_T;
  int rpb = COMPOSE_BCH;
  if ((long long)ktiles * nbands >= 768) rpb = ((B + COMPOSE_BCH - 1) / COMPOSE_BCH) * COMPOSE_BCH;
  dim3 grid(ktiles, (B + rpb - 1) / rpb, nbands);
  if (G * nper > 128) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_compose_fwd, compose_tile_bytes(G * nper));
  if (rc) return rc;
  hipLaunchKernelGGL(k_compose_fwd, grid, dim3(CF_T), compose_tile_bytes(G * nper), (hipStream_t)stream, (const float2*)Y, K,
                     G, nper, c, rgain, B, (const float2*)direct, ldd, direct ? direct_rows : nullptr,
                     (const float2*)filt, (float2*)H, ldh, (float2*)S_out, nbands * G * nper, ldf, rpb);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_compose_fwd(const float* Y, int K, int G, int nper, const float* c,
                                const float* rgain, int B, const float* direct, int ldd,
                                const long long* direct_rows, const float* filt, float* H, int ldh,
                                float* S_out, void* stream) {
  return gfdn_compose_banded_fwd(Y, K, 1, G, nper, c, rgain, B, direct, ldd, direct_rows, filt, K, H, ldh,
                                 S_out, stream);
}

// gS[g][k] = sum_b rgain[b][g] conj(filt_k) gH[b][k];  gY[k][n] = c_n gS[g(n)][k];
// gc partial[block][n] = sum_{k in block} Re(conj(gS) Y);
// grgain partial[(b,g)][block] = sum_{k in block} Re(conj(gH'[b][k]) S[g][k]),  S[g][k] = sum_{n in g} c_n Y[k][n].
// One block = 64 bins.  The Y tile is staged once in LDS (row stride N+1: the per-lane row reads of
// the S prologue then spread over the banks); the receiver loop is split over the 4 wavefronts: each
// gH value is read once and serves both gS (fixed-order fold in LDS) and its own receiver-gain partial
// (wave reduction over the 64 bins) -- the second pass over gH of the first version is gone.  Then all
// 256 threads sweep the 64 x N tile for gY and the gc products in linear (coalesced) order.
#define CBT 64
#define CB_RB 16
__global__ __launch_bounds__(256) void k_compose_bwd_a(const float2* __restrict__ Y, int K, int G,
                                                       int nper, const float* __restrict__ c,
                                                       const float* __restrict__ rgain, int B,
                                                       const float2* __restrict__ filt,
                                                       const float2* __restrict__ gH, int ldh,
                                                       float2* __restrict__ gY,
                                                       float* __restrict__ rg_partial,
                                                       float* __restrict__ gc_partial, int ldy, int ldf) {
  const int N = G * nper, NS = N + 1;
  {   // band blockIdx.y (see k_compose_fwd)
    const int band = blockIdx.y;
    Y += (size_t)band * N;
    gY += (size_t)band * N;
    c += (size_t)band * N;
    rgain += (size_t)band * B * G;
    if (filt) filt += (size_t)band * ldf;
    gH += (size_t)band * B * ldh;
    rg_partial += (size_t)band * B * G * gridDim.x;
    gc_partial += (size_t)band * gridDim.x * N;
  }
  float2* yt = compose_lds;                 // [CBT][N+1] : Y tile
  float* vt = (float*)(yt + CBT * NS);      // [CBT][N]   : Re(conj(gS) Y)
  float* pp = vt + CBT * N;                 // [CB_RB * G][65] : per-bin receiver-gain products, then reused as
  float2 (*s_gS)[GFDN_MAX_GROUPS][CBT] = (float2 (*)[GFDN_MAX_GROUPS][CBT])pp;   // [4][MAX_GROUPS][CBT] gS fold
  const int kx = threadIdx.x & 63, bg = threadIdx.x >> 6;
  const int k0 = blockIdx.x * CBT, nparts = gridDim.x;
  const int k = k0 + kx;
  const bool live = k < K;
  const int kk = live ? k : K - 1;
  const size_t base = (size_t)k0 * ldy;
  const int nb = K - k0 < CBT ? K - k0 : CBT;
  const int lim = nb * N;
  for (int e = threadIdx.x; e < CBT * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = e < lim ? Y[base + (size_t)kq * ldy + n] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  {
    float2 Sl[GFDN_MAX_GROUPS], acc[GFDN_MAX_GROUPS];
#pragma unroll
    for (int g = 0; g < GFDN_MAX_GROUPS; ++g) {
      acc[g] = make_float2(0.f, 0.f);
      Sl[g] = make_float2(0.f, 0.f);
      if (g < G)
        for (int i = 0; i < nper; ++i) {
          const float2 y = yt[kx * NS + g * nper + i];
          const float cc = c[g * nper + i];
          Sl[g].x += cc * y.x;
          Sl[g].y += cc * y.y;
        }
    }
    const float2 fc = filt ? cconj(filt[kk]) : make_float2(1.f, 0.f);
    // receivers in chunks of CB_RB: per-lane products go to LDS rows (stride 65: the column sums
    // below then walk distinct banks) and 2 threads per (b, g) add them up in a fixed order --
    // an order of magnitude fewer instructions than one wave reduction per (b, g)
    for (int b0 = 0; b0 < B; b0 += CB_RB) {
      float2 ghv[CB_RB / 4];
#pragma unroll
      for (int i = 0; i < CB_RB / 4; ++i) {        // this wave's receivers of the chunk: loads in flight together
        const int b = b0 + bg + 4 * i;
        ghv[i] = (live && b < B) ? gH[(size_t)b * ldh + kk] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < CB_RB / 4; ++i) {
        const int b = b0 + bg + 4 * i;
        if (b < B) {
          float2 gh = ghv[i];
          if (filt) gh = cmul(gh, fc);
#pragma unroll
          for (int g = 0; g < GFDN_MAX_GROUPS; ++g) {
            if (g < G) {
              const float rg = rgain[b * G + g];
              acc[g].x += rg * gh.x;
              acc[g].y += rg * gh.y;
              pp[((b - b0) * G + g) * 65 + kx] = gh.x * Sl[g].x + gh.y * Sl[g].y;
            }
          }
        }
      }
      __syncthreads();
      const int nout = (B - b0 < CB_RB ? B - b0 : CB_RB) * G;
      for (int o2 = threadIdx.x; o2 < nout * 2; o2 += 256) {
        const int o = o2 >> 1, h = o2 & 1;
        float sacc = 0.f;
        for (int j = 0; j < 32; ++j) sacc += pp[o * 65 + h * 32 + j];
        sacc += __shfl_xor(sacc, 1);
        if (h == 0) rg_partial[(size_t)(b0 * G + o) * nparts + blockIdx.x] = sacc;
      }
      __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < GFDN_MAX_GROUPS; ++g)
      if (g < G) s_gS[bg][g][kx] = acc[g];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * CBT; idx += 256) {
    const int g = idx / CBT, x = idx - g * CBT;
    float2 t = s_gS[0][g][x];
    t = cadd(t, s_gS[1][g][x]);
    t = cadd(t, s_gS[2][g][x]);
    t = cadd(t, s_gS[3][g][x]);
    s_gS[0][g][x] = t;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < CBT * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    float v = 0.f;
    if (e < lim) {
      const float2 y = yt[kq * NS + n];
      const float2 gs = s_gS[0][n / nper][kq];
      gY[base + (size_t)kq * ldy + n] = cscale(gs, c[n]);
      v = gs.x * y.x + gs.y * y.y;
    }
    vt[e] = v;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {                 // fixed-order column sums
    float sacc = 0.f;
    for (int x = 0; x < CBT; ++x) sacc += vt[x * N + n];
    gc_partial[(size_t)blockIdx.x * N + n] = sacc;
  }
}

// gc[n] = sum_p gc_partial[p][n]  (blocks 0..N-1);  grgain[r] = sum_p rg_partial[r][p]  (blocks N..):
// one wavefront per output, lane-strided partial sums then a wave reduction -- a fixed order.
__global__ __launch_bounds__(64) void k_compose_finish(const float* __restrict__ gc_partial,
                                                       const float* __restrict__ rg_partial,
                                                       int nparts, int N, int BG, float* __restrict__ gc,
                                                       float* __restrict__ grgain) {
  float s = 0.f;
  {   // band blockIdx.y
    const int band = blockIdx.y;
    gc_partial += (size_t)band * nparts * N;
    rg_partial += (size_t)band * BG * nparts;
    gc += (size_t)band * N;
    grgain += (size_t)band * BG;
  }
  if ((int)blockIdx.x < N) {
    for (int p = threadIdx.x; p < nparts; p += 64) s += gc_partial[(size_t)p * N + blockIdx.x];
    s = wave_sum(s);
    if (threadIdx.x == 0) gc[blockIdx.x] = s;
  } else {
    const int r = blockIdx.x - N;
    for (int p = threadIdx.x; p < nparts; p += 64) s += rg_partial[(size_t)r * nparts + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) grgain[r] = s;
  }
}

// out[r] = sum_c part[r][c], one wavefront per row, fixed order
__global__ __launch_bounds__(64) void k_sum_rows(const float* __restrict__ part, int cols,
                                                 float* __restrict__ out) {
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 64) s += part[(size_t)r * cols + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[r] = s;
}

static size_t compose_partial_bytes(int K, int G, int nper) {
  const size_t b = (size_t)((K + CBT - 1) / CBT) * G * nper * sizeof(float);
  return (b + 15) & ~(size_t)15;
}
// gc partial slots [tiles][N] followed by the receiver-gain partial slots [B*G][tiles]
extern "C" size_t gfdn_compose_bwd_work_bytes(int K, int G, int nper, int B) {
  const size_t tiles = (size_t)(K + CBT - 1) / CBT;
  return compose_partial_bytes(K, G, nper) + (size_t)B * G * tiles * sizeof(float);
}

extern "C" size_t gfdn_compose_banded_bwd_work_bytes(int K, int nbands, int G, int nper, int B) {
  const size_t tiles = (size_t)(K + CBT - 1) / CBT;
  return (size_t)nbands * compose_partial_bytes(K, G, nper) + (size_t)nbands * B * G * tiles * sizeof(float);
}

extern "C" int gfdn_compose_bwd(const float* Y, int K, int G, int nper, const float* c,
                                const float* rgain, int B, const float* filt, const float* gH,
                                int ldh, float* gY, float* gc, float* grgain, void* work,
                                void* stream) {
  return gfdn_compose_banded_bwd(Y, K, 1, G, nper, c, rgain, B, filt, K, gH, ldh, gY, gc, grgain, work, stream);
}

extern "C" int gfdn_compose_banded_bwd(const float* Y, int K, int nbands, int G, int nper, const float* c,
                                       const float* rgain, int B, const float* filt, int ldf,
                                       const float* gH, int ldh, float* gY, float* gc, float* grgain,
                                       void* work, void* stream) {
  if (!Y || !c || !rgain || !gH || !gY || !gc || !grgain || !work) return GFDN_E_BADARG;
  if (K <= 0 || G <= 0 || nper <= 0 || B <= 0 || ldh < K || nbands <= 0) return GFDN_E_BADARG;
  if (filt && nbands > 1 && ldf < K) return GFDN_E_BADARG;
  if (G > GFDN_MAX_GROUPS || G * nper > 64 || nbands > 65535) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int N = G * nper;
  const int nparts = (K + CBT - 1) / CBT;
  float* gc_partial = (float*)work;
  float* rg_partial = (float*)((char*)work + (size_t)nbands * compose_partial_bytes(K, G, nper));
  size_t uni = (size_t)CB_RB * G * 65 * sizeof(float);
  if (uni < (size_t)4 * GFDN_MAX_GROUPS * CBT * sizeof(float2)) uni = (size_t)4 * GFDN_MAX_GROUPS * CBT * sizeof(float2);
  const size_t lds = (size_t)CBT * (N + 1) * sizeof(float2) + (size_t)CBT * N * sizeof(float) + uni;
  int rc = ensure_dyn_lds(k_compose_bwd_a, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compose_bwd_a, dim3(nparts, nbands), dim3(256), lds, s, (const float2*)Y,
                     K, G, nper, c, rgain, B, (const float2*)filt, (const float2*)gH, ldh, (float2*)gY,
                     rg_partial, gc_partial, nbands * N, ldf);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_compose_finish, dim3(N + B * G, nbands), dim3(64), 0, s, gc_partial, rg_partial, nparts,
                     N, B * G, gc, grgain);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// directional (SH-domain) output stage, model.py:1056-1088
//   H[b][l][k] = filt_k * sum_g w[b][g][l] c[g*nper+l] Y[k][g*nper+l]
// ------------------------------------------------------------------------------------------
// Y tile of SH_TB bins x N lines staged through LDS once per workgroup (linear global reads; the per-thread rows of
// the bin-major layout are 8 N bytes apart) and reused by the SH_BCH receivers of the workgroup; w[b][n] c[n] from an
// LDS table.  (One thread per (bin, receiver) reading Y straight from memory took 1.22 ms at N = 36, B = 32.)
#define SH_TB 128
#define SH_BCH 8
#define GFDN_MAX_SH_LINES 64      // delay lines the SH output stage's backward keeps in registers
#define GFDN_SH_MAX_TILES 4096    // bin tiles (of SH_TB) its partial sums are sized for
__global__ __launch_bounds__(SH_TB) void k_compose_sh_fwd(const float2* __restrict__ Y, int K, int G,
                                                          int nper, const float* __restrict__ c,
                                                          const float* __restrict__ w, int B,
                                                          const float2* __restrict__ filt,
                                                          float2* __restrict__ H) {
  const int N = G * nper, NS = N + 1 + (N & 1);          // odd row stride: conflict-free rows
  float2* yt = compose_lds;                              // [SH_TB][NS]
  float* sw = (float*)(yt + SH_TB * NS);                 // [SH_BCH][N]
  const int k0 = blockIdx.x * SH_TB, b0 = blockIdx.y * SH_BCH;
  const int nbin = K - k0 < SH_TB ? K - k0 : SH_TB;
  const int nb = B - b0 < SH_BCH ? B - b0 : SH_BCH;
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < nbin * N; e += SH_TB) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = Y[base + e];
  }
  for (int e = threadIdx.x; e < nb * N; e += SH_TB) {
    const int bb = e / N, n = e - bb * N;
    sw[e] = w[(size_t)(b0 + bb) * N + n] * c[n];
  }
  __syncthreads();
  const int k = k0 + threadIdx.x;
  if (k >= K) return;
  const float2 f = filt ? filt[k] : make_float2(1.f, 0.f);
  const float2* yrow = yt + threadIdx.x * NS;
  for (int bb = 0; bb < nb; ++bb) {
    const float* sb = sw + bb * N;
    for (int l = 0; l < nper; ++l) {
      float2 h = make_float2(0.f, 0.f);
      for (int g = 0; g < G; ++g) {
        const int n = g * nper + l;
        const float2 y = yrow[n];
        h.x += sb[n] * y.x;
        h.y += sb[n] * y.y;
      }
      if (filt) h = cmul(h, f);
      H[((size_t)(b0 + bb) * nper + l) * K + k] = h;
    }
  }
}

extern "C" int gfdn_compose_sh_fwd(const float* Y, int K, int G, int nper, const float* c,
                                   const float* w, int B, const float* filt, float* H,
                                   void* stream) {
  if (!Y || !c || !w || !H || K <= 0 || G <= 0 || nper <= 0 || B <= 0) return GFDN_E_BADARG;
  const int N = G * nper, NS = N + 1 + (N & 1);
  const size_t lds = (size_t)SH_TB * NS * sizeof(float2) + (size_t)SH_BCH * N * sizeof(float);
  int rc = ensure_dyn_lds(k_compose_sh_fwd, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compose_sh_fwd, dim3((K + SH_TB - 1) / SH_TB, (B + SH_BCH - 1) / SH_BCH), dim3(SH_TB), lds,
                     (hipStream_t)stream, (const float2*)Y, K, G, nper, c, w, B, (const float2*)filt, (float2*)H);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// Backward of the SH output stage (gH' = gH conj(filt)) in two launches, each reading gH once:
//   k_compose_sh_bwd_y: thread per bin over a staged Y tile: gY[k][n] = c_n sum_b w[b][n] gH'[b][l(n)][k] (written
//                       through the tile: linear stores) and per-tile partials of gc[n] = sum_k Re(acc conj Y);
//   k_compose_sh_bwd_w: gw[b][n] = c_n sum_k Re(conj(gH'[b][l][k]) Y[k][n]) as a split-K product per SH channel l:
//                       a workgroup stages (receivers x 64 bins) of gH' and (64 bins x G) of Y in LDS, thread (b, g)
//                       accumulates its dot product over SH_KC bins; partials [chunk][b][n].
// (The first version looped over all receivers per (bin, line) and over all bins per (receiver, line), both with
// 8 N-byte strided Y accesses: 531 + 426 us; one fused pass with a wave reduction per (receiver, line): 524 us.)
__global__ __launch_bounds__(SH_TB) void k_compose_sh_bwd_y(const float2* __restrict__ Y, int K, int G,
                                                            int nper, const float* __restrict__ c,
                                                            const float* __restrict__ w, int B,
                                                            const float2* __restrict__ filt,
                                                            const float2* __restrict__ gH,
                                                            float2* __restrict__ gY,
                                                            float* __restrict__ gc_partial) {
  const int N = G * nper, NS = N + 1 + (N & 1);
  constexpr int NW = SH_TB / 64;
  float2* yt = compose_lds;                              // [SH_TB][NS]
  float2* sgh = yt + SH_TB * NS;                         // [nper][SH_TB]
  float* sw = (float*)(sgh + nper * SH_TB);              // [B][N]
  float* sg = sw + B * N;                                // [NW][N]
  int* lch = (int*)(sg + NW * N);                        // [N]: SH channel of line n
  const int k0 = blockIdx.x * SH_TB;
  const int nbin = K - k0 < SH_TB ? K - k0 : SH_TB;
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < SH_TB * N; e += SH_TB) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = e < nbin * N ? Y[base + e] : make_float2(0.f, 0.f);
  }
  for (int e = threadIdx.x; e < B * N; e += SH_TB) sw[e] = w[e];
  for (int e = threadIdx.x; e < N; e += SH_TB) lch[e] = e % nper;
  __syncthreads();
  const int k = k0 + threadIdx.x;
  const bool valid = k < K;
  const int kk = valid ? k : K - 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float2 fc = filt ? cconj(filt[kk]) : make_float2(1.f, 0.f);
  float2* yrow = yt + threadIdx.x * NS;
  float2 acc[GFDN_MAX_SH_LINES];
#pragma unroll
  for (int n = 0; n < GFDN_MAX_SH_LINES; ++n) acc[n] = make_float2(0.f, 0.f);
  // per receiver: the thread's nper gradient values go to its own LDS column, then every line n (static register
  // index) picks the value of its channel l(n) from there  (a register array indexed by l(n) would go to scratch;
  // testing n % nper == l per (channel, line) cost 642 us)
  for (int b = 0; b < B; ++b) {
    const float* swb = sw + b * N;
    for (int l = 0; l < nper; ++l) {
      float2 gh = valid ? gH[((size_t)b * nper + l) * K + kk] : make_float2(0.f, 0.f);
      if (filt) gh = cmul(gh, fc);
      sgh[l * SH_TB + threadIdx.x] = gh;
    }
#pragma unroll
    for (int n = 0; n < GFDN_MAX_SH_LINES; ++n) {
      if (n < N) {
        const float2 gh = sgh[lch[n] * SH_TB + threadIdx.x];
        acc[n].x += swb[n] * gh.x;
        acc[n].y += swb[n] * gh.y;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < GFDN_MAX_SH_LINES; ++n) {
    if (n < N) {
      const float2 y = yrow[n];
      const float v = wave_sum(valid ? (acc[n].x * y.x + acc[n].y * y.y) : 0.f);
      if (lane == 0) sg[wv * N + n] = v;
      yrow[n] = cscale(acc[n], c[n]);                    // gY through the tile: linear global writes below
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N; e += SH_TB) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += sg[q * N + e];
    gc_partial[(size_t)blockIdx.x * N + e] = sum;
  }
  for (int e = threadIdx.x; e < nbin * N; e += SH_TB) {
    const int kq = e / N, n = e - kq * N;
    gY[base + e] = yt[kq * NS + n];
  }
}

#define SH_KT 64               // bins per staged sub-tile
#define SH_KC 512              // bins per workgroup (partials: K / SH_KC per (b, n))
__global__ __launch_bounds__(SH_TB) void k_compose_sh_bwd_w(const float2* __restrict__ Y, int K, int G,
                                                            int nper, const float* __restrict__ c, int B,
                                                            const float2* __restrict__ filt,
                                                            const float2* __restrict__ gH,
                                                            float* __restrict__ partial) {
  const int N = G * nper;
  const int rb = SH_TB / G;                              // receivers per workgroup
  float2* ta = compose_lds;                              // [rb][SH_KT + 1]
  float2* ty = ta + rb * (SH_KT + 1);                    // [SH_KT][G]
  const int l = blockIdx.y, b0 = blockIdx.z * rb;
  const int nb = B - b0 < rb ? B - b0 : rb;
  const int bb = threadIdx.x / G, g = threadIdx.x - bb * G;
  const bool mine = bb < nb;
  const int kbeg = blockIdx.x * SH_KC;
  float acc = 0.f;
  for (int ks = kbeg; ks < kbeg + SH_KC && ks < K; ks += SH_KT) {
    for (int e = threadIdx.x; e < rb * SH_KT; e += SH_TB) {
      const int r = e / SH_KT, kq = e - r * SH_KT;
      const int k = ks + kq;
      float2 v = make_float2(0.f, 0.f);
      if (r < nb && k < K) {
        v = gH[((size_t)(b0 + r) * nper + l) * K + k];
        if (filt) v = cmul(v, cconj(filt[k]));
      }
      ta[r * (SH_KT + 1) + kq] = v;
    }
    for (int e = threadIdx.x; e < SH_KT * G; e += SH_TB) {
      const int kq = e / G, gg = e - kq * G;
      const int k = ks + kq;
      ty[e] = k < K ? Y[(size_t)k * N + gg * nper + l] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    if (mine) {
      const float2* ar = ta + bb * (SH_KT + 1);
#pragma unroll 8
      for (int kq = 0; kq < SH_KT; ++kq) {
        const float2 a = ar[kq], y = ty[kq * G + g];
        acc += a.x * y.x + a.y * y.y;
      }
    }
    __syncthreads();
  }
  if (mine) {
    const int n = g * nper + l;
    partial[((size_t)blockIdx.x * B + b0 + bb) * N + n] = acc * c[n];
  }
}

// sums over the 64 lanes of up to 32 values per lane (v[0 .. min(NV, 32) - 1]) by a halving exchange on the VALU: at
// distance 32 the lower half-wave keeps values 0..15 and the upper 16..31 -- v_permlane32_swap hands each half the other's
// sixteen in one instruction per pair --, v_permlane16_swap does the same between the rows of 16 lanes for 8 values, and
// inside the rows DPP (row_ror:8, row_half_mirror, quad_perm) pairs the lanes for 4, 2, 1 values and the last pair:
// ~70 VALU instructions and no LDS traffic, against six ds_bpermute + six adds per value.  Lane l returns sum l >> 1.
template <int NV>
__device__ __forceinline__ float wave_sums_transposed32(const float (&v)[NV]) {
  const int lane = threadIdx.x & 63;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // first operand: what the LOWER half keeps (value i), second: what the UPPER half keeps (value i + 16); the swap
    // trades the first's upper 32 lanes for the second's lower 32
    const unsigned lo_v = __float_as_uint(i < NV ? v[i < NV ? i : 0] : 0.f);
    const unsigned hi_v = __float_as_uint(i + 16 < NV ? v[i + 16 < NV ? i + 16 : 0] : 0.f);
    const auto r = __builtin_amdgcn_permlane32_swap(lo_v, hi_v, false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {          // even rows keep a[i], odd rows a[i + 8]: the first's odd rows for the second's even rows
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 8]), false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const bool hi = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s0 = dpp_pair_sum<0x128>(a[i]), s1 = dpp_pair_sum<0x128>(a[i + 4]);      // row_ror:8
      a[i] = hi ? s1 : s0;
    }
  }
  {
    const bool hi = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float s0 = dpp_pair_sum<0x141>(a[i]), s1 = dpp_pair_sum<0x141>(a[i + 2]);      // row_half_mirror
      a[i] = hi ? s1 : s0;
    }
  }
  {
    const float s0 = dpp_pair_sum<0x4e>(a[0]), s1 = dpp_pair_sum<0x4e>(a[1]);              // quad_perm:[2,3,0,1]
    a[0] = (lane & 2) ? s1 : s0;
  }
  return dpp_pair_sum<0xb1>(a[0]);                                                         // quad_perm:[1,0,3,2]
}

// Both gradients of the SH output stage from ONE pass over gH (nper = 1, 4, 9, 16: ambisonics orders 0..3).  A workgroup
// owns SHF_TB bins; its four waves split the receivers, lane = bin.  Per receiver a thread loads its nper gradient
// values (coalesced 512-byte rows), accumulates acc[n] += w[b][n] gH'[b][l(n)] in registers (static indices: g and l are
// unrolled) and forms Re(conj(gH') Y[k][n]), summed over the wave's 64 bins -> one partial row of gw per (tile, b).
// The waves' acc are then summed through LDS (wave 3 -> 2 -> 1 -> 0), wave 0 writes gY through the staged Y tile and the tile's
// partial of gc.  1025 workgroups x 4 waves instead of 513 x 2 that looped over every receiver, and the second launch
// (k_compose_sh_bwd_w, another 151 MB read of gH) is gone.
#define SHF_TB 64
template <int NPER>
__global__ __launch_bounds__(256, NPER <= 9 ? 4 : 2) void k_compose_sh_bwd_fused(const float2* __restrict__ Y, int K, int G,
                                                              const float* __restrict__ c,
                                                              const float* __restrict__ w, int B,
                                                              const float2* __restrict__ filt,
                                                              const float2* __restrict__ gH,
                                                              float2* __restrict__ gY,
                                                              float* __restrict__ partial) {
  constexpr int MAXG = NPER <= 4 ? 8 : 32 / NPER;      // groups the accumulators are sized for (at most 32 lines)
  constexpr int NL = MAXG * NPER;
  const int N = G * NPER, NS = N + 1 + (N & 1);
  float2* yt = compose_lds;                              // [SHF_TB][NS]
  float2* red = yt + SHF_TB * NS;                        // [N][SHF_TB]
  float* sw = (float*)(red + N * SHF_TB);                // [B][N]
  const int tile = blockIdx.x, k0 = tile * SHF_TB;
  const int nbin = K - k0 < SHF_TB ? K - k0 : SHF_TB;
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < SHF_TB * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = e < nbin * N ? Y[base + e] : make_float2(0.f, 0.f);
  }
  for (int e = threadIdx.x; e < B * N; e += 256) sw[e] = w[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = k0 + lane;
  const bool valid = k < K;
  const int kk = valid ? k : K - 1;
  const float2 fc = filt ? cconj(filt[kk]) : make_float2(1.f, 0.f);
  const float2* yrow = yt + lane * NS;
  float2 acc[NL];
#pragma unroll
  for (int n = 0; n < NL; ++n) acc[n] = make_float2(0.f, 0.f);
  // partial rows: [tile][B * N + N] = gw of the tile's bins per (b, n), then gc per n
  float* prow = partial + (size_t)tile * ((size_t)B * N + N);
  for (int b = wv; b < B; b += 4) {
    float2 gh[NPER];
#pragma unroll
    for (int l = 0; l < NPER; ++l) {
      float2 v = valid ? gH[((size_t)b * NPER + l) * K + kk] : make_float2(0.f, 0.f);
      if (filt) v = cmul(v, fc);
      gh[l] = v;
    }
    const float* swb = sw + b * N;
    float dot[NL];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
#pragma unroll
      for (int l = 0; l < NPER; ++l) {
        const int n = g * NPER + l;
        dot[n] = 0.f;
        if (g < G) {
          const float s = swb[n];
          acc[n].x += s * gh[l].x;
          acc[n].y += s * gh[l].y;
          const float2 y = yrow[n];
          dot[n] = gh[l].x * y.x + gh[l].y * y.y;
        }
      }
    }
    // the sums over the wave's 64 bins of all lines at once
    static_assert(NL <= 32, "the halving exchange takes 32 values");
    const float mine = wave_sums_transposed32<NL>(dot);
    if (!(lane & 1) && (lane >> 1) < N) prow[(size_t)b * N + (lane >> 1)] = mine * c[lane >> 1];
  }
  // acc over the four waves through ONE (N, 64) buffer: 3 -> 2 -> 1 -> 0 (fixed order ((a3 + a2) + a1) + a0)
#pragma unroll 1
  for (int src = 3; src >= 1; --src) {
    if (wv == src) {
#pragma unroll
      for (int n = 0; n < NL; ++n) if (n < N) red[n * SHF_TB + lane] = acc[n];
    }
    __syncthreads();
    if (wv == src - 1) {
#pragma unroll
      for (int n = 0; n < NL; ++n) if (n < N) { const float2 o = red[n * SHF_TB + lane]; acc[n].x += o.x; acc[n].y += o.y; }
    }
    __syncthreads();
  }
  if (wv == 0) {
    float mine = 0.f;
#pragma unroll
    for (int n = 0; n < NL; ++n) {
      if (n < N) {
        const float2 a = acc[n];
        const float2 y = yrow[n];
        const float v = wave_sum(valid ? (a.x * y.x + a.y * y.y) : 0.f);
        if (lane == n) mine = v;
        yt[lane * NS + n] = cscale(a, c[n]);               // gY through the tile: linear global writes below
      }
    }
    if (lane < N) prow[(size_t)B * N + lane] = mine;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nbin * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    gY[base + e] = yt[kq * NS + n];
  }
}

// rows [nparts][per1 + per2] -> out1 (per1), out2 (per2)
__global__ __launch_bounds__(256) void k_reduce_partials2(const float* __restrict__ partial, int nparts, int per1, int per2,
                                                          float* __restrict__ out1, float* __restrict__ out2) {
  __shared__ float s_red[16];
  const int e = blockIdx.x, per = per1 + per2;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)p * per + e];
  s = block_sum(s, s_red);
  if (threadIdx.x == 0) {
    if (e < per1) out1[e] = s;
    else out2[e - per1] = s;
  }
}

template <int NPER>
static int compose_sh_bwd_fused(const float2* Y, int K, int G, const float* c, const float* w, int B, const float2* filt,
                                const float2* gH, float2* gY, float* gc, float* gw, float* work, hipStream_t s) {
  const int N = G * NPER, NS = N + 1 + (N & 1), ntiles = (K + SHF_TB - 1) / SHF_TB;
  const size_t lds = ((size_t)SHF_TB * NS + (size_t)N * SHF_TB) * sizeof(float2) + (size_t)B * N * sizeof(float);
  int rc = ensure_dyn_lds(k_compose_sh_bwd_fused<NPER>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compose_sh_bwd_fused<NPER>, dim3(ntiles), dim3(256), lds, s, Y, K, G, c, w, B, filt, gH, gY, work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials2, dim3(B * N + N), dim3(256), 0, s, work, ntiles, B * N, N, gw, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}
// (the fused pass takes nper in {1, 4, 9, 16}, lane-indexed rows of N <= 64 lines and what fits the LDS)
static bool compose_sh_fused_ok(int K, int G, int nper, int B) {
  if (nper != 1 && nper != 4 && nper != 9 && nper != 16) return false;
  const int maxg = nper <= 4 ? 8 : 32 / nper;
  const int N = G * nper, NS = N + 1 + (N & 1);
  if (G > maxg || N > 64) return false;
  const size_t lds = ((size_t)SHF_TB * NS + (size_t)N * SHF_TB) * sizeof(float2) + (size_t)B * N * sizeof(float);
  return lds <= 64 * 1024 && (K + SHF_TB - 1) / SHF_TB <= GFDN_SH_MAX_TILES * (SH_TB / SHF_TB);
}

extern "C" size_t gfdn_compose_sh_bwd_work_bytes(int G, int nper, int B) {
  // gw partials [K / SH_KC][B][N], then gc partials [K / SH_TB][N]
  const size_t two_pass = ((size_t)(GFDN_SH_MAX_TILES * SH_TB / SH_KC) * B + GFDN_SH_MAX_TILES) * G * nper * sizeof(float);
  // fused pass: [tiles of SHF_TB bins][B N + N]
  const size_t fused = (size_t)GFDN_SH_MAX_TILES * (SH_TB / SHF_TB) * ((size_t)B + 1) * G * nper * sizeof(float);
  return two_pass > fused ? two_pass : fused;
}

extern "C" int gfdn_compose_sh_bwd(const float* Y, int K, int G, int nper, const float* c,
                                   const float* w, int B, const float* filt, const float* gH,
                                   float* gY, float* gc, float* gw, void* work, void* stream) {
  if (!Y || !c || !w || !gH || !gY || !gc || !gw || !work) return GFDN_E_BADARG;
  if (K <= 0 || G <= 0 || nper <= 0 || B <= 0) return GFDN_E_BADARG;
  const int N = G * nper, NS = N + 1 + (N & 1);
  const int ntiles = (K + SH_TB - 1) / SH_TB, nchunks = (K + SH_KC - 1) / SH_KC;
  if (N > GFDN_MAX_SH_LINES || G > SH_TB || ntiles > GFDN_SH_MAX_TILES) return GFDN_E_UNSUPPORTED;
  const size_t lds = ((size_t)SH_TB * NS + (size_t)nper * SH_TB) * sizeof(float2) +
                     ((size_t)B * N + (size_t)(SH_TB / 64) * N + N) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (compose_sh_fused_ok(K, G, nper, B)) {
    const float2 *Yc = (const float2*)Y, *fl = (const float2*)filt, *gh = (const float2*)gH;
    switch (nper) {
      case 1: return compose_sh_bwd_fused<1>(Yc, K, G, c, w, B, fl, gh, (float2*)gY, gc, gw, (float*)work, s);
      case 4: return compose_sh_bwd_fused<4>(Yc, K, G, c, w, B, fl, gh, (float2*)gY, gc, gw, (float*)work, s);
      case 9: return compose_sh_bwd_fused<9>(Yc, K, G, c, w, B, fl, gh, (float2*)gY, gc, gw, (float*)work, s);
      default: return compose_sh_bwd_fused<16>(Yc, K, G, c, w, B, fl, gh, (float2*)gY, gc, gw, (float*)work, s);
    }
  }
  if (lds > 160 * 1024) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_compose_sh_bwd_y, lds);
  if (rc) return rc;
  float* gw_partial = (float*)work;
  float* gc_partial = gw_partial + (size_t)nchunks * B * N;
  hipLaunchKernelGGL(k_compose_sh_bwd_y, dim3(ntiles), dim3(SH_TB), lds, s, (const float2*)Y, K, G, nper, c, w, B,
                     (const float2*)filt, (const float2*)gH, (float2*)gY, gc_partial);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials, dim3(N), dim3(256), 0, s, gc_partial, ntiles, N, gc);
  GFDN_LAUNCH_CHECK();
  const int rb = SH_TB / G;
  const size_t ldsw = ((size_t)rb * (SH_KT + 1) + (size_t)SH_KT * G) * sizeof(float2);
  hipLaunchKernelGGL(k_compose_sh_bwd_w, dim3(nchunks, nper, (B + rb - 1) / rb), dim3(SH_TB), ldsw, s,
                     (const float2*)Y, K, G, nper, c, B, (const float2*)filt, (const float2*)gH, gw_partial);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials, dim3(B * N), dim3(256), 0, s, gw_partial, nchunks, B * N, gw);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// SH domain -> directional responses, trainer.py:853-865: H_dir[b][j][k] = sum_l A[j][l] H_sh[b][l][k]
// and its adjoint gH_sh[b][l][k] = sum_j A[j][l] gH_dir[b][j][k]   (A real, J x C)
// ------------------------------------------------------------------------------------------
// out[b][o][k] = sum_i A'[o][i] in[b][i][k]: the inputs of a bin are loaded ONCE into registers (the version that
// re-read all of them for every output issued nin x nout loads per thread: 383 us for 151 -> 268 MB)
#define SHMIX_MAX 32
__global__ __launch_bounds__(256) void k_sh_mix(const float* __restrict__ A, int J, int C, int K,
                                                const float2* __restrict__ in, float2* __restrict__ out,
                                                int adjoint) {
  __shared__ float s_a[SHMIX_MAX * SHMIX_MAX];
  const int nin = adjoint ? J : C, nout = adjoint ? C : J;
  for (int e = threadIdx.x; e < nout * nin; e += 256) {         // s_a[o][i]
    const int o = e / nin, i = e - o * nin;
    s_a[e] = adjoint ? A[i * C + o] : A[o * C + i];
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (k >= K) return;
  const float2* ib = in + (size_t)b * nin * K;
  float2* ob = out + (size_t)b * nout * K;
  float2 v[SHMIX_MAX];
#pragma unroll
  for (int i = 0; i < SHMIX_MAX; ++i) v[i] = i < nin ? ib[(size_t)i * K + k] : make_float2(0.f, 0.f);
  for (int o = 0; o < nout; ++o) {
    float2 acc = make_float2(0.f, 0.f);
    const float* ar = s_a + o * nin;
#pragma unroll
    for (int i = 0; i < SHMIX_MAX; ++i) {
      if (i < nin) {
        acc.x += ar[i] * v[i].x;
        acc.y += ar[i] * v[i].y;
      }
    }
    ob[(size_t)o * K + k] = acc;
  }
}

extern "C" int gfdn_sh_to_directional(const float* A, int J, int C, int K, int B, const float* in,
                                      float* out, int adjoint, void* stream) {
  if (!A || !in || !out || J <= 0 || C <= 0 || K <= 0 || B <= 0) return GFDN_E_BADARG;
  if (J > SHMIX_MAX || C > SHMIX_MAX) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_sh_mix, dim3((K + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, A, J, C, K,
                     (const float2*)in, (float2*)out, adjoint);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// colorless statistics of the sub-FDN responses
// ------------------------------------------------------------------------------------------
#define SPEC_CHUNK 2048
// grid (chunks, G): partial[(g*2 + {0: energy, 1: loss}) * nchunk + chunk]
__global__ __launch_bounds__(256) void k_spectral_stats(const float2* __restrict__ S, int G, int K,
                                                        int asym, float scale,
                                                        float* __restrict__ partial,
                                                        float2* __restrict__ gS) {
  __shared__ float s_red[16];
  const int g = blockIdx.y, k0 = blockIdx.x * SPEC_CHUNK, nchunk = gridDim.x;
  const float invK = 1.0f / (float)K;
  float e = 0.f, l = 0.f;
  for (int k = k0 + threadIdx.x; k < k0 + SPEC_CHUNK && k < K; k += 256) {
    float2 s = S[(size_t)g * K + k];
    float p = s.x * s.x + s.y * s.y;
    float mag = sqrtf(p);
    float d = mag - 1.0f;
    e += p;
    float d2 = d * d;
    bool four = asym && (d > 1.0f);
    l += four ? d2 * d2 : d2;
    if (gS) {
      float dl = four ? 4.0f * d2 * d : 2.0f * d;   // d loss / d |S|
      float f = (mag > 0.f) ? scale * invK * dl / mag : 0.f;
      gS[(size_t)g * K + k] = make_float2(f * s.x, f * s.y);
    }
  }
  e = block_sum(e, s_red);
  l = block_sum(l, s_red);
  if (threadIdx.x == 0) {
    partial[(size_t)(g * 2 + 0) * nchunk + blockIdx.x] = e * invK;
    partial[(size_t)(g * 2 + 1) * nchunk + blockIdx.x] = l * invK;
  }
}
// rows (g, {energy, loss}) -> energy[g], loss[g]
__global__ __launch_bounds__(64) void k_spectral_finish(const float* __restrict__ partial, int nchunk,
                                                        float* __restrict__ energy,
                                                        float* __restrict__ loss) {
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 64) s += partial[(size_t)r * nchunk + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    if (r & 1) loss[r >> 1] = s;
    else energy[r >> 1] = s;
  }
}

// Scalar loss bookkeeping of one step in ONE launch (trainer.py:298-313, :473):
//   spectral = w_spec * sum_g loss_g ;  sparsity = w_sparse * -(sum|Q_last| - n sqrt n)/(n (sqrt n - 1))
//   out = { (spectral + sparsity) * inv_world, spectral, sparsity }
//   gQ (G,n,n) = d out[0] / dQ : zero except the last group, -w_sparse inv_world sign(Q)/(n (sqrt n - 1))
__global__ __launch_bounds__(256) void k_colorless_terms(const float* __restrict__ loss_g, int G,
                                                         const float* __restrict__ Q, int n,
                                                         float w_spec, float w_sparse, float inv_world,
                                                         float* __restrict__ out, float* __restrict__ gQ) {
  __shared__ float s_red[16];
  {   // one block per band: its G groups, its three outputs
    const int band = blockIdx.x;
    loss_g += (size_t)band * G;
    Q += (size_t)band * G * n * n;
    if (gQ) gQ += (size_t)band * G * n * n;
    out += (size_t)band * 3;
  }
  const float* Ql = Q + (size_t)(G - 1) * n * n;
  const float denom = (float)n * (sqrtf((float)n) - 1.0f);
  float acc = 0.f;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) acc += fabsf(Ql[e]);
  acc = block_sum(acc, s_red);
  if (gQ) {
    for (int e = threadIdx.x; e < G * n * n; e += blockDim.x) {
      float g = 0.f;
      if (e >= (G - 1) * n * n) {
        const float q = Q[e];
        g = -w_sparse * inv_world / denom * (q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f));
      }
      gQ[e] = g;
    }
  }
  if (threadIdx.x == 0) {
    float sp = 0.f;
    for (int g = 0; g < G; ++g) sp += loss_g[g];
    sp *= w_spec;
    const float sparsity = w_sparse * (-(acc - (float)n * sqrtf((float)n)) / denom);
    out[0] = (sp + sparsity) * inv_world;
    out[1] = sp;
    out[2] = sparsity;
  }
}

extern "C" int gfdn_colorless_terms_banded(const float* loss_g, int nbands, int G, const float* Q, int n,
                                           float w_spec, float w_sparse, float inv_world, float* out3,
                                           float* gQ, void* stream) {
  if (!loss_g || !Q || !out3 || G <= 0 || n <= 1 || nbands <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_colorless_terms, dim3(nbands), dim3(256), 0, (hipStream_t)stream, loss_g, G, Q, n,
                     w_spec, w_sparse, inv_world, out3, gQ);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_colorless_terms(const float* loss_g, int G, const float* Q, int n, float w_spec,
                                    float w_sparse, float inv_world, float* out3, float* gQ,
                                    void* stream) {
  return gfdn_colorless_terms_banded(loss_g, 1, G, Q, n, w_spec, w_sparse, inv_world, out3, gQ, stream);
}

// out = { wa * sum(a) + wb * sum(b), wa * sum(a), wb * sum(b) }   (either input may be NULL);
// item i of a = (sum of its a_cols partials) / a_div[a_rows ? a_rows[i] : i]  (a_div optional)
// WS_T / 64 waves per band: a wave sums the partial columns of its items (lanes stride the columns, VALU wave sum: fixed
// order), the waves' sums are added in wave order.  (One lane walking a row of 264 columns per item took 51 us on the side
// stream and held the gain network's backward up; with <= 33 columns the old thread-per-item form took 6 us.)
#define WS_T 512
__global__ __launch_bounds__(WS_T) void k_weighted_sums(const float* __restrict__ a, int a_cols,
                                                        const float* __restrict__ a_div,
                                                        const long long* __restrict__ a_rows, float wa,
                                                        const float* __restrict__ b, float wb, int n,
                                                        float* __restrict__ out) {
  __shared__ float s_a[WS_T / 64], s_b[WS_T / 64];
  {   // one block per band: its n items, its three outputs
    const size_t i0 = (size_t)blockIdx.x * n;
    if (a) a += i0 * a_cols;
    if (a_rows) a_rows += i0;
    else if (a_div) a_div += i0;
    if (b) b += i0;
    out += (size_t)blockIdx.x * 3;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float sa = 0.f, sb = 0.f;
  for (int i = w; i < n; i += WS_T / 64) {
    if (a) {
      float v = 0.f;
      for (int c = lane; c < a_cols; c += 64) v += a[(size_t)i * a_cols + c];
      v = wave_sum_full(v);
      if (a_div) v /= a_div[a_rows ? a_rows[i] : i];
      sa += v;                                            // (the same value in every lane)
    }
    if (b) sb += b[i];
  }
  if (lane == 0) { s_a[w] = sa; s_b[w] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int q = 0; q < WS_T / 64; ++q) { ta += s_a[q]; tb += s_b[q]; }
    ta *= wa; tb *= wb;
    out[0] = ta + tb; out[1] = ta; out[2] = tb;
  }
}

extern "C" int gfdn_weighted_sums_banded(const float* a, int a_cols, const float* a_div, const long long* a_rows,
                                         float wa, const float* b, float wb, int n, int nbands, float* out3,
                                         void* stream) {
  if ((!a && !b) || !out3 || n <= 0 || nbands <= 0 || (a && a_cols <= 0)) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_weighted_sums, dim3(nbands), dim3(WS_T), 0, (hipStream_t)stream, a, a_cols, a_div, a_rows,
                     wa, b, wb, n, out3);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_weighted_sums(const float* a, int a_cols, const float* a_div, const long long* a_rows,
                                  float wa, const float* b, float wb, int n, float* out3, void* stream) {
  return gfdn_weighted_sums_banded(a, a_cols, a_div, a_rows, wa, b, wb, n, 1, out3, stream);
}

// trainer.py:323-332: b_n, c_n /= energy_g^(1/4) for n in group g, in place
__global__ void k_normalize_io(const float* __restrict__ energy, float* __restrict__ b,
                               float* __restrict__ c, int N, int nper) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float s = powf(energy[n / nper], 0.25f);
  b[n] /= s;
  c[n] /= s;
}

extern "C" int gfdn_normalize_io(const float* energy, float* b, float* c, int G, int nper,
                                 void* stream) {
  if (!energy || !b || !c || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  const int N = G * nper;
  hipLaunchKernelGGL(k_normalize_io, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, energy, b, c,
                     N, nper);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gfdn_spectral_stats_work_bytes(int G, int K) {
  return (size_t)2 * G * ((K + SPEC_CHUNK - 1) / SPEC_CHUNK) * sizeof(float);
}

extern "C" int gfdn_spectral_stats(const float* S, int G, int K, int asym, float scale,
                                   float* energy, float* loss, float* gS, void* work,
                                   void* stream) {
  if (!S || !energy || !loss || !work || G <= 0 || K <= 0) return GFDN_E_BADARG;
  const int nchunk = (K + SPEC_CHUNK - 1) / SPEC_CHUNK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_spectral_stats, dim3(nchunk, G), dim3(256), 0, s, (const float2*)S, G, K, asym,
                     scale, (float*)work, (float2*)gS);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_spectral_finish, dim3(2 * G), dim3(64), 0, s, (const float*)work, nchunk, energy,
                     loss);
  GFDN_LAUNCH_CHECK();
  return 0;
}
