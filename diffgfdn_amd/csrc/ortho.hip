// Orthogonal feedback-matrix parameterisation on the device, for gfx950.
//
// reference: feedback_loop.py:16-36 (Skew, MatrixExponential), :270 (ortho_param = expm o skew),
// :393-404 (block (i,j) of the mixing matrix = Q_i Q_j; with zero coupling only Q_g Q_g is used).
//   forward : X_g = triu(M_g,1) - triu(M_g,1)^T,  Q_g = expm(X_g),  QQ_g = Q_g Q_g
//   backward: G = gQ + gQQ Q^T + Q^T gQQ;  gX = L_exp(X^T, G) (adjoint of the Frechet derivative,
//             read off the top-right block of expm([[X^T, G],[0, X^T]]));  gM = triu(gX - gX^T, 1)
// One 256-thread workgroup per group; every matrix lives in LDS in float64 (they are 4x4 ... 32x32,
// the block-triangular one 2n x 2n); expm = scaling and squaring around a degree-12 Taylor
// polynomial (||X/2^s||_1 <= 1/2  ->  truncation error < 1e-13).  torch.matrix_exp would issue
// dozens of launches and a host synchronisation for these tiny matrices.
#include "common.h"

extern __shared__ double ortho_lds[];

#define EXPM_DEG 12
#define EXPM_THETA 0.5

// expm(A): A (scaled in place), P, R, T are m*m LDS arrays; s_tmp >= m+1 doubles.  Returns the array
// that holds the result (R, or one of the scratch arrays after the ping-pong of the squarings).
// Every Taylor / squaring step is ONE pass + ONE barrier: the product, its accumulation into R
// (element e is owned by one thread) and the role swap of the buffers need no second pass.
__device__ __forceinline__ double* expm_lds(double* A, double* P, double* R, double* T, double* s_tmp,
                                            int m) {
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
      double acc = 0.0;
      for (int q = 0; q < m; ++q) acc += Pc[i * m + q] * A[q * m + j];
      acc *= inv;
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
      double acc = 0.0;
      for (int q = 0; q < m; ++q) acc += Rc[r * m + q] * Rc[q * m + c];
      Sc[e] = acc;
    }
    __syncthreads();
    double* t = Rc; Rc = Sc; Sc = t;
  }
  return Rc;
}

__device__ __forceinline__ double skew_elem(const float* M, int n, int i, int j) {
  if (i < j) return (double)M[i * n + j];
  if (i > j) return -(double)M[j * n + i];
  return 0.0;
}

__global__ __launch_bounds__(256) void k_ortho_fwd(const float* __restrict__ M, int n,
                                                   float* __restrict__ Q, float* __restrict__ QQ) {
  double* A = ortho_lds;
  double* P = A + n * n;
  double* R = P + n * n;
  double* T = R + n * n;
  double* tmp = T + n * n;
  const float* Mg = M + (size_t)blockIdx.x * n * n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) A[e] = skew_elem(Mg, n, e / n, e % n);
  __syncthreads();
  const double* E = expm_lds(A, P, R, T, tmp, n);
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    if (Q) Q[(size_t)blockIdx.x * n * n + e] = (float)E[e];
    if (QQ) {
      const int i = e / n, j = e - i * n;
      double acc = 0.0;
      for (int q = 0; q < n; ++q) acc += E[i * n + q] * E[q * n + j];
      QQ[(size_t)blockIdx.x * n * n + e] = (float)acc;
    }
  }
}

// Qsaved: Q_g from the forward (float32) when the caller still holds it -- the n x n exponential
// that G needs is then not recomputed -- or NULL.
__global__ __launch_bounds__(256) void k_ortho_bwd(const float* __restrict__ M, int n,
                                                   const float* __restrict__ gQ,
                                                   const float* __restrict__ gQQ,
                                                   const float* __restrict__ Qsaved,
                                                   const float* __restrict__ gM_add,
                                                   float* __restrict__ gM) {
  const int m = 2 * n;
  double* A = ortho_lds;          // m*m
  double* P = A + m * m;
  double* R = P + m * m;
  double* T = R + m * m;
  double* tmp = T + m * m;        // m + 1
  double* Gt = tmp + (m + 2);     // n*n : total gradient w.r.t. Q
  const size_t off = (size_t)blockIdx.x * n * n;
  const float* Mg = M + off;
  // G = gQ (+ gQQ Q^T + Q^T gQQ : needs Q = expm(X))
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) Gt[e] = gQ ? (double)gQ[off + e] : 0.0;
  __syncthreads();
  if (gQQ) {
    const double* Qd;
    if (Qsaved) {
      for (int e = threadIdx.x; e < n * n; e += blockDim.x) R[e] = (double)Qsaved[off + e];
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
        acc += (double)gQQ[off + i * n + k] * Qd[j * n + k] + Qd[k * n + i] * (double)gQQ[off + k * n + j];
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
    if (gM_add) v += (double)gM_add[off + e];      // a gradient that reaches M directly (the raw blocks of the sub-FDNs)
    gM[off + e] = (float)v;
  }
}

static size_t ortho_fwd_lds(int n) { return ((size_t)4 * n * n + n + 2) * sizeof(double); }
static size_t ortho_bwd_lds(int n) {
  const int m = 2 * n;
  return ((size_t)4 * m * m + (m + 2) + (size_t)n * n) * sizeof(double);
}

extern "C" int gfdn_ortho_fwd(const float* M, int G, int n, float* Q, float* QQ, void* stream) {
  if (!M || (!Q && !QQ) || G <= 0 || n <= 0) return GFDN_E_BADARG;
  if (n > GFDN_MAX_BLOCK) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_ortho_fwd, ortho_fwd_lds(n));
  if (rc) return rc;
  hipLaunchKernelGGL(k_ortho_fwd, dim3(G), dim3(256), ortho_fwd_lds(n), (hipStream_t)stream, M, n, Q, QQ);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_ortho_bwd_add(const float* M, int G, int n, const float* gQ, const float* gQQ,
                                  const float* Q, const float* gM_add, float* gM, void* stream) {
  if (!M || !gM || (!gQ && !gQQ) || G <= 0 || n <= 0) return GFDN_E_BADARG;
  if (n > GFDN_MAX_BLOCK || ortho_bwd_lds(n) > 160 * 1024) return GFDN_E_UNSUPPORTED;
  int rc = ensure_dyn_lds(k_ortho_bwd, ortho_bwd_lds(n));
  if (rc) return rc;
  hipLaunchKernelGGL(k_ortho_bwd, dim3(G), dim3(256), ortho_bwd_lds(n), (hipStream_t)stream, M, n, gQ, gQQ, Q, gM_add,
                     gM);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_ortho_bwd(const float* M, int G, int n, const float* gQ, const float* gQQ,
                              const float* Q, float* gM, void* stream) {
  return gfdn_ortho_bwd_add(M, G, n, gQ, gQQ, Q, nullptr, gM, stream);
}
