// Device-side pieces of the orthogonal parameterisation shared by csrc/ortho.hip and the fused parameter-gradient
// kernel of csrc/blocktf.hip (see ortho.hip for the maths and the reference lines).
#pragma once
#include "common.h"

#define EXPM_DEG 12
#define EXPM_THETA 0.5

// expm(A) by scaling and squaring around a degree-12 Taylor polynomial, in LDS, by a whole workgroup.
// Every product is ONE pass + ONE barrier (element e is owned by one thread).
// MT > 0: the order is known at compile time -- the inner products are unrolled, so their 2 MT LDS reads are issued
// together instead of one dependent read-read-FMA round trip per term (a step of the 18 x 18 adjoint took ~2 us that
// way, 15 steps per call); same terms in the same order: the same bits.  MT = 0: any order.
template <int MT>
__device__ __forceinline__ double expm_dot(const double* row, const double* col, int m) {
  double acc = 0.0;
  if (MT > 0) {
#pragma unroll
    for (int q = 0; q < MT; ++q) acc += row[q] * col[q * MT];
  } else {
    for (int q = 0; q < m; ++q) acc += row[q] * col[q * m];
  }
  return acc;
}
// Degree-12 Taylor polynomial by Paterson-Stockmeyer (round 6): with A2, A3, A4 in hand
//     p(A) = B0 + A4 (B1 + A4 (B2 + A4 / 12!)),   Bi = c_{4i} I + c_{4i+1} A + c_{4i+2} A2 + c_{4i+3} A3,  c_k = 1 / k!
// is FOUR dependent matrix products (A2; A3 and A4 together; two Horner steps) instead of the eleven of the term-by-term sum:
// every product is one pass + one workgroup barrier of a launch whose whole length is such passes (the band bank's tail runs
// two exponentials back to back on 28 workgroups: 34 us).  Six m x m arrays: A (scaled in place), A2, A3, A4 and two for the
// Horner steps / the squarings.  Returns the array that holds the result.
template <int MT>
__device__ __forceinline__ double* expm_lds_t(double* A, double* A2, double* A3, double* A4, double* U, double* V,
                                              double* s_tmp, int m) {
  // 1-norm = max column sum
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    double c = 0.0;
    for (int i = 0; i < m; ++i) c += fabs(A[i * m + j]);
    s_tmp[j] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double nrm = 0.0;
    for (int j = 0; j < m; ++j) nrm = fmax(nrm, s_tmp[j]);
    int s = 0;
    while (nrm > EXPM_THETA && s < 60) { nrm *= 0.5; ++s; }
    s_tmp[m] = (double)s;
  }
  __syncthreads();
  const int s = (int)s_tmp[m];
  const double sc = ldexp(1.0, -s);
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) A[e] *= sc;
  __syncthreads();
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {            // A2 = A A
    const int i = e / m, j = e - i * m;
    A2[e] = expm_dot<MT>(A + i * m, A + j, m);
  }
  __syncthreads();
  // 1 / k!
  constexpr double c2 = 1.0 / 2, c3 = 1.0 / 6, c4 = 1.0 / 24, c5 = 1.0 / 120, c6 = 1.0 / 720, c7 = 1.0 / 5040,
                   c8 = 1.0 / 40320, c9 = 1.0 / 362880, c10 = 1.0 / 3628800, c11 = 1.0 / 39916800, c12 = 1.0 / 479001600;
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {            // A3 = A2 A, A4 = A2 A2, U = B2 + c12 A4
    const int i = e / m, j = e - i * m;
    const double a3 = expm_dot<MT>(A2 + i * m, A + j, m), a4 = expm_dot<MT>(A2 + i * m, A2 + j, m);
    A3[e] = a3;
    A4[e] = a4;
    U[e] = (i == j ? c8 : 0.0) + c9 * A[e] + c10 * A2[e] + c11 * a3 + c12 * a4;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {            // V = B1 + A4 U
    const int i = e / m, j = e - i * m;
    V[e] = (i == j ? c4 : 0.0) + c5 * A[e] + c6 * A2[e] + c7 * A3[e] + expm_dot<MT>(A4 + i * m, U + j, m);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {            // U = B0 + A4 V
    const int i = e / m, j = e - i * m;
    U[e] = (i == j ? 1.0 : 0.0) + A[e] + c2 * A2[e] + c3 * A3[e] + expm_dot<MT>(A4 + i * m, V + j, m);
  }
  __syncthreads();
  double* Rc = U;
  double* Sc = V;
  for (int i = 0; i < s; ++i) {
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
      const int r = e / m, c = e - r * m;
      Sc[e] = expm_dot<MT>(Rc + r * m, Rc + c, m);
    }
    __syncthreads();
    double* t = Rc; Rc = Sc; Sc = t;
  }
  return Rc;
}
// The term-by-term form on FOUR arrays, for orders whose six arrays do not fit in LDS (m > EXPM_PS_MAX: the adjoint of a
// dense block of more than 24 lines, the lossless prototype's 32 x 32): one product per Taylor term.
#define EXPM_PS_MAX 48
__device__ __forceinline__ double* expm_lds_taylor(double* A, double* P, double* R, double* T, double* s_tmp, int m) {
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    double c = 0.0;
    for (int i = 0; i < m; ++i) c += fabs(A[i * m + j]);
    s_tmp[j] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double nrm = 0.0;
    for (int j = 0; j < m; ++j) nrm = fmax(nrm, s_tmp[j]);
    int s = 0;
    while (nrm > EXPM_THETA && s < 60) { nrm *= 0.5; ++s; }
    s_tmp[m] = (double)s;
  }
  __syncthreads();
  const int s = (int)s_tmp[m];
  const double sc = ldexp(1.0, -s);
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
    const int i = e / m, j = e - i * m;
    const double a = A[e] * sc;
    A[e] = a;
    P[e] = a;
    R[e] = a + (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  double* Pc = P;
  double* Tc = T;
  for (int k = 2; k <= EXPM_DEG; ++k) {          // Tc = Pc A / k ; R += Tc
    const double inv = 1.0 / (double)k;
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
      const int i = e / m, j = e - i * m;
      const double acc = expm_dot<0>(Pc + i * m, A + j, m) * inv;
      Tc[e] = acc;
      R[e] += acc;
    }
    __syncthreads();
    double* t = Pc; Pc = Tc; Tc = t;
  }
  double* Rc = R;
  double* Sc = Tc;                               // free scratch
  for (int i = 0; i < s; ++i) {
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
      const int r = e / m, c = e - r * m;
      Sc[e] = expm_dot<0>(Rc + r * m, Rc + c, m);
    }
    __syncthreads();
    double* t = Rc; Rc = Sc; Sc = t;
  }
  return Rc;
}
__host__ __device__ static inline size_t expm_lds_doubles(int m) {
  return (size_t)(m > EXPM_PS_MAX ? 4 : 6) * m * m + m + 2;
}
// W: expm_lds_doubles(m) doubles, the matrix in W[0 .. m^2) (scaled in place)
__device__ __forceinline__ double* expm_lds(double* W, int m) {
  if (m > EXPM_PS_MAX) return expm_lds_taylor(W, W + m * m, W + 2 * m * m, W + 3 * m * m, W + 4 * m * m, m);
  double* A = W;
  double* A2 = A + m * m;
  double* A3 = A2 + m * m;
  double* A4 = A3 + m * m;
  double* U = A4 + m * m;
  double* V = U + m * m;
  double* tmp = V + m * m;
  switch (m) {                                   // (wave-uniform: the orders the models use, 2 n for the adjoint)
    case 4: return expm_lds_t<4>(A, A2, A3, A4, U, V, tmp, m);
    case 8: return expm_lds_t<8>(A, A2, A3, A4, U, V, tmp, m);
    case 9: return expm_lds_t<9>(A, A2, A3, A4, U, V, tmp, m);
    case 16: return expm_lds_t<16>(A, A2, A3, A4, U, V, tmp, m);
    case 18: return expm_lds_t<18>(A, A2, A3, A4, U, V, tmp, m);
    default: return expm_lds_t<0>(A, A2, A3, A4, U, V, tmp, m);
  }
}
__device__ __forceinline__ double skew_elem(const float* M, int n, int i, int j) {
  if (i < j) return (double)M[i * n + j];
  if (i > j) return -(double)M[j * n + i];
  return 0.0;
}

// Backward of one group, run by a whole workgroup.  lds: 4 m^2 + (m + 2) + n^2 doubles (m = 2 n).  All pointers are
// already offset to the group; gQ / gQQ / Qsaved / gM_add may be NULL, and may point into LDS (the fused kernel hands
// over gQQ and gM_add that way).  Qsaved: Q_g from the forward (float32) when the caller still holds it -- the n x n
// exponential that G needs is then not recomputed.
__device__ __forceinline__ void ortho_bwd_group(double* lds, const float* Mg, int n, const float* gQ,
                                                const float* gQQ, const float* Qsaved, const float* gM_add,
                                                float* gM) {
  const int m = 2 * n;
  double* A = lds;                // the exponential's work space: expm_lds_doubles(m), the matrix first
  double* R = A + m * m;          // (free until the exponential runs: Q as float64)
  double* tmp = A + expm_lds_doubles(m) - (m + 2);     // m + 2 (the exponential's own scratch, free outside it)
  double* Gt = A + expm_lds_doubles(m);      // n*n : total gradient w.r.t. Q
  // G = gQ (+ gQQ Q^T + Q^T gQQ : needs Q = expm(X))
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) Gt[e] = gQ ? (double)gQ[e] : 0.0;
  __syncthreads();
  if (gQQ) {
    const double* Qd;
    if (Qsaved) {
      for (int e = threadIdx.x; e < n * n; e += blockDim.x) R[e] = (double)Qsaved[e];
      __syncthreads();
      Qd = R;
    } else {
      for (int e = threadIdx.x; e < n * n; e += blockDim.x) A[e] = skew_elem(Mg, n, e / n, e % n);
      __syncthreads();
      Qd = expm_lds(A, n);                        // (n x n) = Q  (6 n^2 + n + 2 doubles of the 2 n work space)
    }
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e - i * n;
      double acc = 0.0;
      for (int k = 0; k < n; ++k) {
        // (gQQ Q^T)_ij = sum_k gQQ_ik Q_jk ;  (Q^T gQQ)_ij = sum_k Q_ki gQQ_kj
        acc += (double)gQQ[i * n + k] * Qd[j * n + k] + Qd[k * n + i] * (double)gQQ[k * n + j];
      }
      Gt[e] += acc;
    }
    __syncthreads();
  }
  // normalise G (the derivative is linear in G) so that it does not drive the scaling
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    double c = 0.0;
    for (int i = 0; i < n; ++i) c = fmax(c, fabs(Gt[i * n + j]));
    tmp[j] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double g = 0.0;
    for (int j = 0; j < n; ++j) g = fmax(g, tmp[j]);
    tmp[m + 1] = g;
  }
  __syncthreads();
  const double gmax = tmp[m + 1];
  const double ginv = gmax > 0.0 ? 1.0 / gmax : 0.0;
  // block matrix [[X^T, G/gmax], [0, X^T]],  X^T = -X for a skew matrix
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
    const int i = e / m, j = e - i * m;
    double v = 0.0;
    if (i < n && j < n) v = -skew_elem(Mg, n, i, j);
    else if (i >= n && j >= n) v = -skew_elem(Mg, n, i - n, j - n);
    else if (i < n && j >= n) v = Gt[i * n + (j - n)] * ginv;
    A[e] = v;
  }
  __syncthreads();
  const double* E = expm_lds(A, m);
  // gX = gmax * E[0:n, n:2n];  gM = triu(gX - gX^T, 1)
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e - i * n;
    double v = 0.0;
    if (i < j) v = gmax * (E[i * m + n + j] - E[j * m + n + i]);
    if (gM_add) v += (double)gM_add[e];      // a gradient that reaches M directly (the raw blocks of the sub-FDNs)
    gM[e] = (float)v;
  }
}

__host__ __device__ static inline size_t ortho_bwd_lds_doubles(int n) {
  return expm_lds_doubles(2 * n) + (size_t)n * n;
}
