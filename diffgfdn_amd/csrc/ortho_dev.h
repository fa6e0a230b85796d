// Device-side pieces of the orthogonal parameterisation shared by csrc/ortho.hip and the fused parameter-gradient
// kernel of csrc/blocktf.hip (see ortho.hip for the maths and the reference lines).
#pragma once
#include "common.h"

#define EXPM_DEG 12
#define EXPM_THETA 0.5

// expm(A): A (scaled in place), P, R, T are m*m LDS arrays; s_tmp >= m+1 doubles.  Returns the array
// that holds the result (R, or one of the scratch arrays after the ping-pong of the squarings).
// Every Taylor / squaring step is ONE pass + ONE barrier: the product, its accumulation into R
// (element e is owned by one thread) and the role swap of the buffers need no second pass.
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
template <int MT>
__device__ __forceinline__ double* expm_lds_t(double* A, double* P, double* R, double* T, double* s_tmp, int m) {
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
      const double acc = expm_dot<MT>(Pc + i * m, A + j, m) * inv;
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
      Sc[e] = expm_dot<MT>(Rc + r * m, Rc + c, m);
    }
    __syncthreads();
    double* t = Rc; Rc = Sc; Sc = t;
  }
  return Rc;
}
__device__ __forceinline__ double* expm_lds(double* A, double* P, double* R, double* T, double* s_tmp, int m) {
  switch (m) {                                   // (wave-uniform: the orders the models use, 2 n for the adjoint)
    case 4: return expm_lds_t<4>(A, P, R, T, s_tmp, m);
    case 8: return expm_lds_t<8>(A, P, R, T, s_tmp, m);
    case 9: return expm_lds_t<9>(A, P, R, T, s_tmp, m);
    case 16: return expm_lds_t<16>(A, P, R, T, s_tmp, m);
    case 18: return expm_lds_t<18>(A, P, R, T, s_tmp, m);
    default: return expm_lds_t<0>(A, P, R, T, s_tmp, m);
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
  double* A = lds;                // m*m
  double* P = A + m * m;
  double* R = P + m * m;
  double* T = R + m * m;
  double* tmp = T + m * m;        // m + 1
  double* Gt = tmp + (m + 2);     // n*n : total gradient w.r.t. Q
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
      Qd = expm_lds(A, P, R, T, tmp, n);          // (n x n) = Q
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
  const double* E = expm_lds(A, P, R, T, tmp, m);
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
  const int m = 2 * n;
  return (size_t)4 * m * m + (m + 2) + (size_t)n * n;
}
