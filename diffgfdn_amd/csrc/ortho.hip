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
#include "ortho_dev.h"

extern __shared__ double ortho_lds[];

__global__ __launch_bounds__(256) void k_ortho_fwd(const float* __restrict__ M, int n,
                                                   float* __restrict__ Q, float* __restrict__ QQ) {
  double* A = ortho_lds;
  const float* Mg = M + (size_t)blockIdx.x * n * n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) A[e] = skew_elem(Mg, n, e / n, e % n);
  __syncthreads();
  const double* E = expm_lds(A, n);
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

__global__ __launch_bounds__(512) void k_ortho_bwd(const float* __restrict__ M, int n,
                                                   const float* __restrict__ gQ,
                                                   const float* __restrict__ gQQ,
                                                   const float* __restrict__ Qsaved,
                                                   const float* __restrict__ gM_add,
                                                   float* __restrict__ gM) {
  const size_t off = (size_t)blockIdx.x * n * n;
  ortho_bwd_group(ortho_lds, M + off, n, gQ ? gQ + off : nullptr, gQQ ? gQQ + off : nullptr,
                  Qsaved ? Qsaved + off : nullptr, gM_add ? gM_add + off : nullptr, gM + off);
}

static size_t ortho_fwd_lds(int n) { return expm_lds_doubles(n) * sizeof(double); }
static size_t ortho_bwd_lds(int n) { return ortho_bwd_lds_doubles(n) * sizeof(double); }

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
  // (the adjoint works on 2n x 2n matrices: 324 elements at n = 9 -- one pass of 512 threads per product instead of two of 256)
  hipLaunchKernelGGL(k_ortho_bwd, dim3(G), dim3(4 * n * n > 256 ? 512 : 256), ortho_bwd_lds(n), (hipStream_t)stream, M, n, gQ, gQQ, Q, gM_add,
                     gM);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_ortho_bwd(const float* M, int G, int n, const float* gQ, const float* gQQ,
                              const float* Q, float* gM, void* stream) {
  return gfdn_ortho_bwd_add(M, G, n, gQ, gQQ, Q, nullptr, gM, stream);
}
