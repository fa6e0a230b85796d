// Block transfer functions in polynomial form for blocks of up to NINE delay lines (the directional model of BASELINE.json
// configs[3]: N = 27 = 3 x 9), zero coupling, for gfx950 -- the coefficient records and the records -> parameters map; the
// evaluation on the reference's grid and its adjoint are real transforms of the coefficient sequences (csrc/polyfft.hip).
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn): the un-damped group responses of the colorless branch,
// y^(g) = (D(z) - M_g)^-1 b_g contracted with the output gains, S_g(z) = c_g^T y^(g)(z) (model.py:209-252), which
// csrc/solve.hip evaluates by a per-bin 9 x 9 complex elimination (k_solve_fwd_rl<9> / k_solve_bwd_rl<9>: two launches that
// hold the whole chip for ~45 + ~50 us per band-step).  As for the blocks of <= 8 lines (csrc/blocktf8.hip), with
// X(z) = D(z) Gamma^-1 - A, D = diag(z^{m_i}),
//     det X        = sum_S Q_S e_S ,  Q_S = det((-A)[S^c, S^c]) prod_{i in S} 1/gamma_i ,  e_S = prod_{i in S} z^{m_i}
//     c^T adj(X) b = sum_S P_S e_S ,  P_S = sum_i c_i Y_{i,S} ,  Y_{i,S} = det(((-A) with column i := b)[S^c, S^c]) prod 1/gamma
// over the 2^n subsets S of the lines -- 512 coefficients per polynomial at n = 9, built in float64 -- and the gradient map
//     Q_S = igp_S det(M_S) ,   P_S = -igp_S det(B_S) ,   B_S = [[M_S, b on S^c], [c^T on S^c, 0]]   ((n + 1) x (n + 1))
//     dL/dA_ij = sum_S igp_S (gP_S Cof(B_S)_ij - gQ_S Cof(M_S)_ij) ,  dL/db_i = -sum_S igp_S gP_S Cof(B_S)_{i,n} ,
//     dL/dc_j  = -sum_S igp_S gP_S Cof(B_S)_{n,j}
// with the cofactor matrices as det x inverse^T from an in-register Gauss-Jordan inversion (partial pivoting by selects).
// Record layout: coef (nblk, 10, 512): [0] the determinant polynomial, [1 + i] the numerator of y_i; gradient records
// (nblk, 1024): dL/dP_S | dL/dQ_S; subset S = bit i <-> line i.
#include "common.h"

#define T9_L 9
#define T9_SUB 512
#define T9_REC ((T9_L + 1) * T9_SUB)
#define T9_ACC (T9_L * T9_L + 2 * T9_L)          // 81 dL/dA (row-major 9 x 9) | 9 dL/db | 9 dL/dc

// determinant of an N x N matrix in registers, partial pivoting by row selects (compile-time indices only)
template <int N>
__device__ __forceinline__ double t9_det(double (&m)[N][N]) {
  double det = 1.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double best = fabs(m[j][j]);
    int bi = j;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      const double v = fabs(m[i][j]);
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      const bool sw = bi == i;
#pragma unroll
      for (int c = j; c < N; ++c) {
        const double t = m[j][c];
        m[j][c] = sw ? m[i][c] : t;
        m[i][c] = sw ? t : m[i][c];
      }
    }
    if (bi != j) det = -det;
    const double p = m[j][j];
    det *= p;
    const double inv = p != 0.0 ? 1.0 / p : 0.0;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      const double f = m[i][j] * inv;
#pragma unroll
      for (int c = j + 1; c < N; ++c) m[i][c] -= f * m[j][c];
    }
  }
  return det;
}

// in-place inverse, returns the determinant; a vanishing pivot is replaced by 1e-150 (det x inverse keeps the finite limit)
template <int N>
__device__ __forceinline__ double t9_inverse(double (&m)[N][N]) {
  double det = 1.0;
  int perm[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double best = fabs(m[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double v = fabs(m[i][k]);
      if (v > best) { best = v; p = i; }
    }
    perm[k] = p;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const bool sw = p == i;
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const double t = m[k][c];
        m[k][c] = sw ? m[i][c] : t;
        m[i][c] = sw ? t : m[i][c];
      }
    }
    if (p != k) det = -det;
    double piv = m[k][k];
    if (fabs(piv) < 1e-150) piv = piv < 0.0 ? -1e-150 : 1e-150;
    det *= piv;
    const double pinv = 1.0 / piv;
    m[k][k] = 1.0;
#pragma unroll
    for (int c = 0; c < N; ++c) m[k][c] *= pinv;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i == k) continue;
      const double f = m[i][k];
      m[i][k] = 0.0;
#pragma unroll
      for (int c = 0; c < N; ++c) m[i][c] -= f * m[k][c];
    }
  }
  // (P A)^-1 = A^-1 P^T: undo the row swaps as column swaps, last first
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int c = k + 1; c < N; ++c) {
      const bool sw = perm[k] == c;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const double t = m[r][k];
        m[r][k] = sw ? m[r][c] : t;
        m[r][c] = sw ? t : m[r][c];
      }
    }
  }
  return det;
}

// grid (nblk, 10): y = 0 the determinant polynomial, y = 1 + i the numerator polynomial of y_i = (X^-1 b)_i; thread S = subset
__global__ __launch_bounds__(T9_SUB) void k_tf9_coefs(const float* __restrict__ A, const float* __restrict__ ig,
                                                      const float* __restrict__ b, int n, float* __restrict__ coef) {
  __shared__ double sA[T9_L * T9_L], sb[T9_L], sig[T9_L];
  const int blk = blockIdx.x, task = blockIdx.y, S = threadIdx.x;
  const float* Ab = A + (size_t)blk * n * n;
  float* cf = coef + (size_t)blk * T9_REC;
  if (S < T9_L * T9_L) {
    const int i = S / T9_L, j = S - i * T9_L;
    sA[S] = (i < n && j < n) ? -(double)Ab[i * n + j] : 0.0;
  }
  if (S < T9_L) {
    sb[S] = S < n ? (double)b[blk * n + S] : 0.0;
    sig[S] = (S < n && ig) ? (double)ig[blk * n + S] : 1.0;
  }
  __syncthreads();
  const bool absent = (S >> n) != 0;              // the subset names a line the block does not have: coefficient 0
  double igp = 1.0;
#pragma unroll
  for (int i = 0; i < T9_L; ++i)
    if ((S >> i) & 1) igp *= sig[i];
  const int col = task - 1;                       // -1: (-A) untouched
  double m[T9_L][T9_L];
#pragma unroll
  for (int r = 0; r < T9_L; ++r)
#pragma unroll
    for (int cc = 0; cc < T9_L; ++cc) {
      const bool in = !((S >> r) & 1) && !((S >> cc) & 1) && r < n && cc < n;
      double v = sA[r * T9_L + cc];
      if (cc == col) v = sb[r];
      m[r][cc] = in ? v : (r == cc ? 1.0 : 0.0);
    }
  const double d = t9_det<T9_L>(m);
  const bool none = absent || (task > 0 && (col >= n || ((S >> col) & 1)));
  cf[task * T9_SUB + S] = none ? 0.f : (float)(d * igp);
}

extern "C" int gfdn_tf9_coefs(const float* A, const float* inv_gamma, const float* b, int nblk, int nper, float* coef,
                              void* stream) {
  if (!A || !b || !coef || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > T9_L) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_tf9_coefs, dim3(nblk, T9_L + 1), dim3(T9_SUB), 0, (hipStream_t)stream, A, inv_gamma, b, nper, coef);
  GFDN_LAUNCH_CHECK();
  return 0;
}

__device__ __forceinline__ double t9_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Gradient records -> (dL/dA, dL/db, dL/dc), float64.  grid (nblk, 2, 2): y = 0 the masked n x n matrices (dL/dQ_S -> dL/dA),
// y = 1 the bordered (n + 1) x (n + 1) ones (dL/dP_S -> dL/dA, dL/db, dL/dc); z = which 256 of the 512 subsets; 256 threads,
// one subset each (one wavefront per SIMD: the 10 x 10 inversion lives in 200 registers; with both halves of the subsets in
// one workgroup the launch took 81 us on the step's chain); out[((2 half + z) nblk + blk) 99 + e], added by k_tf9_finish.
__global__ __launch_bounds__(256) void k_tf9_rec_grads(const float* __restrict__ A, const float* __restrict__ ig,
                                                       const float* __restrict__ grec, const float* __restrict__ b,
                                                       const float* __restrict__ c, int n, float* __restrict__ out) {
  __shared__ double sA[T9_L * T9_L], sb[T9_L], sc[T9_L], sig[T9_L], sred[4][T9_ACC];
  const int blk = blockIdx.x, half = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* Ab = A + (size_t)blk * n * n;
  if (tid < T9_L * T9_L) {
    const int i = tid / T9_L, j = tid - i * T9_L;
    sA[tid] = (i < n && j < n) ? -(double)Ab[i * n + j] : 0.0;
  }
  if (tid < T9_L) {
    sb[tid] = tid < n ? (double)b[blk * n + tid] : 0.0;
    sc[tid] = tid < n ? (double)c[blk * n + tid] : 0.0;
    sig[tid] = (tid < n && ig) ? (double)ig[blk * n + tid] : 1.0;
  }
  for (int e = tid; e < 4 * T9_ACC; e += 256) sred[e / T9_ACC][e % T9_ACC] = 0.0;
  __syncthreads();
  {
    const int S = tid + 256 * (int)blockIdx.z;
    const bool absent = (S >> n) != 0, full = S == (1 << n) - 1;      // (the full set's coefficients are constants)
    double igp = 1.0;
#pragma unroll
    for (int i = 0; i < T9_L; ++i)
      if ((S >> i) & 1) igp *= sig[i];
    const double gP = (double)grec[(size_t)blk * 2 * T9_SUB + S], gQ = (double)grec[(size_t)blk * 2 * T9_SUB + T9_SUB + S];
    // (sums over the wave's 64 subsets per entry, the wave's partial added to its own row of sred by lane 0: fixed order)
    if (half == 0) {
      const double wq = (absent || full) ? 0.0 : -gQ * igp;
      double m[T9_L][T9_L];
#pragma unroll
      for (int r = 0; r < T9_L; ++r)
#pragma unroll
        for (int cc = 0; cc < T9_L; ++cc) {
          const bool in = !((S >> r) & 1) && !((S >> cc) & 1) && r < n && cc < n;
          m[r][cc] = in ? sA[r * T9_L + cc] : (r == cc ? 1.0 : 0.0);
        }
      const double wd = wq * t9_inverse<T9_L>(m);
#pragma unroll
      for (int i = 0; i < T9_L; ++i)
#pragma unroll
        for (int j = 0; j < T9_L; ++j) {
          const bool in = !((S >> i) & 1) && !((S >> j) & 1) && i < n && j < n;
          const double v = t9_wave_sum_d(in ? wd * m[j][i] : 0.0);        // Cof = det inverse^T
          if (lane == 0) sred[wv][i * T9_L + j] += v;
        }
    } else {
      const double wp = (absent || full) ? 0.0 : gP * igp;
      double m[T9_L + 1][T9_L + 1];
#pragma unroll
      for (int r = 0; r <= T9_L; ++r)
#pragma unroll
        for (int cc = 0; cc <= T9_L; ++cc) {
          const bool rin = r == T9_L || (!((S >> r) & 1) && r < n), cin = cc == T9_L || (!((S >> cc) & 1) && cc < n);
          double v;
          if (r < T9_L && cc < T9_L) v = (rin && cin) ? sA[r * T9_L + cc] : (r == cc ? 1.0 : 0.0);
          else if (r < T9_L) v = rin ? sb[r] : 0.0;                       // border column: b on S^c
          else if (cc < T9_L) v = cin ? sc[cc] : 0.0;                     // border row: c on S^c
          else v = 0.0;
          m[r][cc] = v;
        }
      const double wd = wp * t9_inverse<T9_L + 1>(m);
#pragma unroll
      for (int i = 0; i < T9_L; ++i) {
        const bool iin = !((S >> i) & 1) && i < n;
#pragma unroll
        for (int j = 0; j < T9_L; ++j) {
          const bool in = iin && !((S >> j) & 1) && j < n;
          const double v = t9_wave_sum_d(in ? wd * m[j][i] : 0.0);
          if (lane == 0) sred[wv][i * T9_L + j] += v;
        }
        const double vb = t9_wave_sum_d(iin ? -wd * m[T9_L][i] : 0.0);    // Cof(B)_{i,n} = det inverse[n][i]
        const double vc = t9_wave_sum_d(iin ? -wd * m[i][T9_L] : 0.0);    // Cof(B)_{n,i} = det inverse[i][n]
        if (lane == 0) { sred[wv][T9_L * T9_L + i] += vb; sred[wv][T9_L * T9_L + T9_L + i] += vc; }
      }
    }
  }
  __syncthreads();
  if (tid < T9_ACC)
    out[((size_t)(2 * half + blockIdx.z) * gridDim.x + blk) * T9_ACC + tid] =
        (float)((sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]));
}

// the four partial sets added -> dL/dA (nblk, n, n), dL/db, dL/dc (nblk n)
__global__ __launch_bounds__(128) void k_tf9_finish(const float* __restrict__ part, int nblk, int n, float* __restrict__ gA,
                                                    float* __restrict__ gb, float* __restrict__ gc) {
  const int blk = blockIdx.x, e = threadIdx.x;
  if (e >= T9_ACC) return;
  const float v = (part[(size_t)blk * T9_ACC + e] + part[((size_t)nblk + blk) * T9_ACC + e]) +
                  (part[((size_t)2 * nblk + blk) * T9_ACC + e] + part[((size_t)3 * nblk + blk) * T9_ACC + e]);
  if (e < T9_L * T9_L) {
    const int i = e / T9_L, j = e - i * T9_L;
    if (i < n && j < n) gA[(size_t)blk * n * n + i * n + j] = v;
  } else if (e < T9_L * T9_L + T9_L) {
    const int i = e - T9_L * T9_L;
    if (i < n) gb[blk * n + i] = v;
  } else {
    const int j = e - T9_L * T9_L - T9_L;
    if (j < n) gc[blk * n + j] = v;
  }
}

extern "C" size_t gfdn_tf9_rec_grads_work_bytes(int nblk) { return (size_t)4 * (nblk > 0 ? nblk : 1) * T9_ACC * sizeof(float); }

extern "C" int gfdn_tf9_rec_grads(const float* A, const float* inv_gamma, const float* grec, const float* b, const float* c,
                                  int nblk, int nper, float* gA, float* gb, float* gc, void* work, void* stream) {
  if (!A || !grec || !b || !c || !gA || !gb || !gc || !work || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > T9_L) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tf9_rec_grads, dim3(nblk, 2, 2), dim3(256), 0, s, A, inv_gamma, grec, b, c, nper, (float*)work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_tf9_finish, dim3(nblk), dim3(128), 0, s, (const float*)work, nblk, nper, gA, gb, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}
