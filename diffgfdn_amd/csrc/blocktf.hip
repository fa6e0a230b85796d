// Block transfer functions in polynomial form, for gfx950 (blocks of n <= 4 delay lines, zero coupling).
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn): with zero inter-group coupling the feedback matrix is
// block diagonal (feedback_loop.py:393-404, :439-443) and every receiver sees the loop only through the
// group transfer functions
//        T_g(z) = c_g^T (D_g(z) Gamma_g^-1 - A_g)^-1 b_g          (model.py:583-619: H = sum_g gain_g T_g + d),
// the sub-FDN responses of the colorless loss being the same expression with the raw M_g and no absorption
// (model.py:209-252).  For a block of n <= 4 lines this is a ratio of two MULTILINEAR polynomials in the n
// phasors zeta_i = z^{m_i} / gamma_i:
//        det(D' - A)           = sum_{S subset [n]} q_S prod_{i in S} zeta_i ,   q_S = (-1)^{|S^c|} det A[S^c, S^c]
//        c^T adj(D' - A) b     = det(D' - A + b c^T) - det(D' - A)               (matrix determinant lemma)
//                              = sum_S (q_S(A - b c^T) - q_S(A)) prod_{i in S} zeta_i ,
// 2 x 16 real coefficients per block that do not depend on the bin.  The per-bin work drops from a pivoted
// 4 x 4 complex elimination (~1000 VALU instructions, csrc/solve.hip k_solve4_*) to 4 sincos + 11 complex
// products + two 16-term forms + one division (~260), the delay-line responses Y (K, N) never exist, and the
// backward is 30 real accumulators per block (dL/dcoef) followed by a bin-independent map coef -> (A, b, c)
// done once per block in float64.  The coefficient records carry the products of 1 / gamma_i, so that the
// per-bin phasors are pure powers of z: prod_{i in S} z^{m_i}.
//
// Record layout: coef (nblk, 32) float32 = [P_S, S = 0..15 | Q_S, S = 0..15], S a bit mask over the lines of
// the block (bit i = line i; masks with bits >= n hold zeros).  T = (sum_S P_S e_S) / (sum_S Q_S e_S).
#include "common.h"
#include "ortho_dev.h"

#define TF_MAXBLK 64
#define TF_REC 32
#define TF_MAXG 4                // groups per band the output-stage kernels take (one wavefront per group)
#define TF_MAX_PARTS 2048

__device__ __forceinline__ float2 tf_zpow(const double* __restrict__ turns, const double* __restrict__ logr,
                                          int k, float m) {
  double t = (double)m * turns[k];
  t -= rint(t);
  float s, c;
  sincospif(2.0f * (float)t, &s, &c);
  if (logr) {
    const float mag = (float)exp((double)m * logr[k]);
    c *= mag;
    s *= mag;
  }
  return make_float2(c, s);
}

// e[S] = prod_{i in S} e_i from the four single-line phasors e[1], e[2], e[4], e[8]  (e[0] is never read)
__device__ __forceinline__ void tf_subsets(float2 (&e)[16]) {
  e[3] = cmul(e[1], e[2]);
  e[5] = cmul(e[1], e[4]);
  e[6] = cmul(e[2], e[4]);
  e[9] = cmul(e[1], e[8]);
  e[10] = cmul(e[2], e[8]);
  e[12] = cmul(e[4], e[8]);
  e[7] = cmul(e[3], e[4]);
  e[11] = cmul(e[3], e[8]);
  e[13] = cmul(e[5], e[8]);
  e[14] = cmul(e[6], e[8]);
  e[15] = cmul(e[3], e[12]);
}

// e[S] = prod_{i in S} z_k^{m_i}
__device__ __forceinline__ void tf_phasors(const double* __restrict__ turns, const double* __restrict__ logr,
                                           int k, const float (&m)[4], float2 (&e)[16]) {
  e[1] = tf_zpow(turns, logr, k, m[0]);
  e[2] = tf_zpow(turns, logr, k, m[1]);
  e[4] = tf_zpow(turns, logr, k, m[2]);
  e[8] = tf_zpow(turns, logr, k, m[3]);
  e[3] = cmul(e[1], e[2]);
  e[5] = cmul(e[1], e[4]);
  e[6] = cmul(e[2], e[4]);
  e[9] = cmul(e[1], e[8]);
  e[10] = cmul(e[2], e[8]);
  e[12] = cmul(e[4], e[8]);
  e[7] = cmul(e[3], e[4]);
  e[11] = cmul(e[3], e[8]);
  e[13] = cmul(e[5], e[8]);
  e[14] = cmul(e[6], e[8]);
  e[15] = cmul(e[3], e[12]);
}

__device__ __forceinline__ void tf_numden(const float (&P)[16], const float (&Q)[16], const float2 (&e)[16],
                                          float2& num, float2& den) {
  num = make_float2(P[0], 0.f);
  den = make_float2(Q[0], 0.f);
#pragma unroll
  for (int S = 1; S < 16; ++S) {
    num.x += P[S] * e[S].x;
    num.y += P[S] * e[S].y;
    den.x += Q[S] * e[S].x;
    den.y += Q[S] * e[S].y;
  }
}

struct TfBlock { float P[16], Q[16], m[4]; };      // 36 floats

// record of block blk from global memory (delays zero-padded: absent lines get the phasor 1 and zero coefficients)
__device__ __forceinline__ void tf_load(const float* __restrict__ coef, const float* __restrict__ delays,
                                        int blk, int n, float (&P)[16], float (&Q)[16], float (&m)[4]) {
#pragma unroll
  for (int S = 0; S < 16; ++S) {
    P[S] = coef[(size_t)blk * TF_REC + S];
    Q[S] = coef[(size_t)blk * TF_REC + 16 + S];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = r < n ? delays[blk * n + r] : 0.f;
}

__device__ __forceinline__ void tf_stage(const float* __restrict__ coef, const float* __restrict__ delays,
                                         int blk0, int nb, int n, TfBlock* tab) {
  for (int t = threadIdx.x; t < nb * 36; t += blockDim.x) {
    const int q = t / 36, f = t - q * 36;
    float v;
    if (f < 32) v = coef[(size_t)(blk0 + q) * TF_REC + f];
    else v = (f - 32) < n ? delays[(blk0 + q) * n + (f - 32)] : 0.f;
    ((float*)&tab[q])[f] = v;
  }
}

// ------------------------------------------------------------------------------------------
// coefficient records (float64 inside: the numerator is a difference of determinants)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double tf_det3(const double* a, int r0, int r1, int r2, int c0, int c1, int c2) {
  return a[r0 * 4 + c0] * (a[r1 * 4 + c1] * a[r2 * 4 + c2] - a[r1 * 4 + c2] * a[r2 * 4 + c1]) -
         a[r0 * 4 + c1] * (a[r1 * 4 + c0] * a[r2 * 4 + c2] - a[r1 * 4 + c2] * a[r2 * 4 + c0]) +
         a[r0 * 4 + c2] * (a[r1 * 4 + c0] * a[r2 * 4 + c1] - a[r1 * 4 + c1] * a[r2 * 4 + c0]);
}
// det of a[ri[0..t-1]][ci[0..t-1]], a row-major 4 x 4, t <= 4
__device__ double tf_det(const double* a, const int* ri, const int* ci, int t) {
  if (t == 0) return 1.0;
  if (t == 1) return a[ri[0] * 4 + ci[0]];
  if (t == 2) return a[ri[0] * 4 + ci[0]] * a[ri[1] * 4 + ci[1]] - a[ri[0] * 4 + ci[1]] * a[ri[1] * 4 + ci[0]];
  if (t == 3) return tf_det3(a, ri[0], ri[1], ri[2], ci[0], ci[1], ci[2]);
  return a[ri[0] * 4 + ci[0]] * tf_det3(a, ri[1], ri[2], ri[3], ci[1], ci[2], ci[3]) -
         a[ri[0] * 4 + ci[1]] * tf_det3(a, ri[1], ri[2], ri[3], ci[0], ci[2], ci[3]) +
         a[ri[0] * 4 + ci[2]] * tf_det3(a, ri[1], ri[2], ri[3], ci[0], ci[1], ci[3]) -
         a[ri[0] * 4 + ci[3]] * tf_det3(a, ri[1], ri[2], ri[3], ci[0], ci[1], ci[2]);
}

// sA[0] = A (zero-padded 4 x 4), sA[1] = A - b c^T, sig[i] = 1 / gamma_i, all float64
// (Ablk: the block's n x n matrix; bblk, cblk, igblk: the block's n gains -- global or LDS)
__device__ __forceinline__ void tf_stage_block(const float* Ablk, const float* bblk, const float* cblk,
                                               const float* igblk, int n, int tid, double (*sA)[16], double* sig) {
  if (tid < 16) {
    const int i = tid >> 2, j = tid & 3;
    const bool in = i < n && j < n;
    const double a = in ? (double)Ablk[i * n + j] : 0.0;
    const double bc = in ? (double)bblk[i] * (double)cblk[j] : 0.0;
    sA[0][tid] = a;
    sA[1][tid] = a - bc;
  }
  if (tid < 4) sig[tid] = (tid < n && igblk) ? (double)igblk[tid] : 1.0;
}

// Records of block blk for one or two sets (A1blk = NULL: one) sharing b, c, by threads tid < 64 of a workgroup (every
// thread must call it: barriers).  A0blk / A1blk: the block's matrices, bblk / cblk: its gains (global or LDS).
__device__ __forceinline__ void tf_coefs_block(const float* A0blk, const float* __restrict__ ig0,
                                               float* __restrict__ coef0, const float* A1blk,
                                               const float* __restrict__ ig1, float* __restrict__ coef1,
                                               const float* bblk, const float* cblk, int blk, int n, int tid) {
  __shared__ double sA[2][2][16], sq[2][2][16], sig[2][4];
  const int nsets = A1blk ? 2 : 1;
  if (tid < 64) {
    tf_stage_block(A0blk, bblk, cblk, ig0 ? ig0 + blk * n : nullptr, n, tid, sA[0], sig[0]);
    if (nsets == 2) tf_stage_block(A1blk, bblk, cblk, ig1 ? ig1 + blk * n : nullptr, n, tid, sA[1], sig[1]);
  }
  __syncthreads();
  if (tid < 32 * nsets) {
    const int st = tid >> 5, v = (tid >> 4) & 1, S = tid & 15;
    double q = 0.0;
    if ((S >> n) == 0) {
      int T[4], t = 0;
      for (int i = 0; i < n; ++i)
        if (!((S >> i) & 1)) T[t++] = i;
      q = tf_det(sA[st][v], T, T, t);
      if (t & 1) q = -q;
    }
    sq[st][v][S] = q;
  }
  __syncthreads();
  if (tid < 32 * nsets && (tid & 31) < 16) {
    const int st = tid >> 5, S = tid & 15;
    double igp = 1.0;
    for (int i = 0; i < 4; ++i)
      if ((S >> i) & 1) igp *= sig[st][i];
    float* coef = st ? coef1 : coef0;
    coef[(size_t)blk * TF_REC + S] = (float)((sq[st][1][S] - sq[st][0][S]) * igp);
    coef[(size_t)blk * TF_REC + 16 + S] = (float)(sq[st][0][S] * igp);
  }
}

// workgroup blk < nblk: the records of block blk for set 0 (A0, ig0 -> coef0) and, with A1, set 1 (A1, ig1 -> coef1)
__global__ __launch_bounds__(64) void k_tf_coefs(const float* __restrict__ A0, const float* __restrict__ ig0,
                                                 float* __restrict__ coef0, const float* __restrict__ A1,
                                                 const float* __restrict__ ig1, float* __restrict__ coef1,
                                                 const float* __restrict__ b, const float* __restrict__ c, int n) {
  const int blk = blockIdx.x;
  tf_coefs_block(A0 + (size_t)blk * n * n, ig0, coef0, A1 ? A1 + (size_t)blk * n * n : nullptr, ig1, coef1, b + blk * n,
                 c + blk * n, blk, n, threadIdx.x);
}

// Head of the band bank's step in ONE launch, one workgroup per group: Q = expm(skew(M)), QQ = Q Q (k_ortho_fwd), then
// the records of the damped loop (A = QQ with 1 / gamma -> coef0) and of the sub-FDN (A = raw M -> coef1) as k_tf_coefs;
// QQ passes through its float32 rounding in between, so the records equal those of the two separate launches bit for bit.
__global__ __launch_bounds__(256) void k_tf_ortho_coefs(const float* __restrict__ M, int n, const float* __restrict__ ig0,
                                                        const float* __restrict__ b, const float* __restrict__ c,
                                                        float* __restrict__ Q, float* __restrict__ QQ,
                                                        float* __restrict__ coef0, float* __restrict__ coef1) {
  extern __shared__ double tfh_lds[];
  __shared__ float lQQ[16];
  double* A = tfh_lds;
  const int blk = blockIdx.x;
  const float* Mg = M + (size_t)blk * n * n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) A[e] = skew_elem(Mg, n, e / n, e % n);
  __syncthreads();
  const double* E = expm_lds(A, n);
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    Q[(size_t)blk * n * n + e] = (float)E[e];
    const int i = e / n, j = e - i * n;
    double acc = 0.0;
    for (int q = 0; q < n; ++q) acc += E[i * n + q] * E[q * n + j];
    const float v = (float)acc;
    QQ[(size_t)blk * n * n + e] = v;
    lQQ[e] = v;
  }
  __syncthreads();
  tf_coefs_block(lQQ, ig0, coef0, coef1 ? Mg : nullptr, nullptr, coef1, b + blk * n, c + blk * n, blk, n, threadIdx.x);
}

extern "C" int gfdn_tf_coefs_fwd(const float* A, const float* b, const float* c, const float* inv_gamma,
                                 int nblk, int nper, float* coef, void* stream) {
  if (!A || !b || !c || !coef || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_tf_coefs, dim3(nblk), dim3(64), 0, (hipStream_t)stream, A, inv_gamma, coef, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr, b, c, nper);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// two record sets sharing b, c in ONE launch (the damped loop's Q Q with 1 / gamma and the sub-FDNs' raw M)
extern "C" int gfdn_tf_coefs_fwd2(const float* A0, const float* inv_gamma0, float* coef0, const float* A1,
                                  const float* inv_gamma1, float* coef1, const float* b, const float* c, int nblk,
                                  int nper, void* stream) {
  if (!A0 || !A1 || !b || !c || !coef0 || !coef1 || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_tf_coefs, dim3(nblk), dim3(64), 0, (hipStream_t)stream, A0, inv_gamma0, coef0, A1, inv_gamma1,
                     coef1, b, c, nper);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_tf_ortho_coefs(const float* M, const float* inv_gamma, const float* b, const float* c, int nblk,
                                   int nper, float* Q, float* QQ, float* coef, float* coef_sub, void* stream) {
  if (!M || !b || !c || !Q || !QQ || !coef || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  const size_t lds = expm_lds_doubles(nper) * sizeof(double);
  hipLaunchKernelGGL(k_tf_ortho_coefs, dim3(nblk), dim3(256), lds, (hipStream_t)stream, M, nper, inv_gamma, b, c, Q, QQ,
                     coef, coef_sub);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// dL/dcoef records -> dL/dA, dL/db, dL/dc for up to two coefficient sets that share b, c (set 0: the damped loop,
// A0 = Q Q with 1 / gamma; set 1: the sub-FDNs, A1 = raw M).  grec_s (nblk, 32): the summed records.
struct TfBwdSet {
  const float* A;
  const float* ig;
  const float* grec;
  float* gA;
};
// One block's share, run by threads tid < 64 of a workgroup (every thread of the workgroup must call it: barriers).
// grec0 / grec1: this block's 32 summed records (global or LDS).  lA0 / lA1 (optional, LDS, n x n row-major):
// dL/dA of the two sets rounded to float32 as the global outputs are, for a consumer in the same kernel.
__device__ __forceinline__ void tf_coefs_bwd_block(const TfBwdSet& s0, const TfBwdSet& s1, const float* grec0,
                                                   const float* grec1, const float* __restrict__ b,
                                                   const float* __restrict__ c, int n, int blk, int tid,
                                                   float* __restrict__ gb, float* __restrict__ gc, float* lA0,
                                                   float* lA1) {
  __shared__ double sA[2][2][16], sig[2][4], sgq[2][2][16], sgA[2][2][16], spart[4][64];
  const int nsets = s1.A ? 2 : 1;
  if (tid < 64) {
    tf_stage_block(s0.A + (size_t)blk * n * n, b + blk * n, c + blk * n, s0.ig ? s0.ig + blk * n : nullptr, n, tid, sA[0],
                   sig[0]);
    if (nsets == 2)
      tf_stage_block(s1.A + (size_t)blk * n * n, b + blk * n, c + blk * n, s1.ig ? s1.ig + blk * n : nullptr, n, tid,
                     sA[1], sig[1]);
  }
  __syncthreads();
  if (tid < 32 * nsets) {
    const int s = tid >> 5, v = (tid >> 4) & 1, S = tid & 15;
    const float* grec = s ? grec1 : grec0;
    double igp = 1.0;
    for (int i = 0; i < 4; ++i)
      if ((S >> i) & 1) igp *= sig[s][i];
    double g = 0.0;
    if ((S >> n) == 0 && S != (1 << n) - 1)            // (the full set's determinant is the constant 1)
      g = v ? (double)grec[S] * igp : ((double)grec[16 + S] - (double)grec[S]) * igp;
    sgq[s][v][S] = g;
  }
  __syncthreads();
  {
    // cofactor sums: thread (chunk, s, v, i, j) takes the subsets S = chunk, chunk + nch, ... (nch = wavefronts of the
    // workgroup: the 16 subsets of an entry are a chain of LDS-latency-bound determinants), fixed-order fold after
    const int t64 = tid & 63, chunk = tid >> 6, nch = ((int)blockDim.x >> 6) < 4 ? ((int)blockDim.x >> 6) : 4;
    const int s = t64 >> 5, v = (t64 >> 4) & 1, i = (t64 >> 2) & 3, j = t64 & 3;
    double acc = 0.0;
    if (chunk < nch && s < nsets && i < n && j < n) {
      for (int S = chunk; S < (1 << n); S += nch) {
        if (((S >> i) & 1) || ((S >> j) & 1)) continue;
        int Ti[4], Tj[4], ti = 0, tj = 0, pi = 0, pj = 0, t = 0;
        for (int l = 0; l < n; ++l) {
          if ((S >> l) & 1) continue;
          if (l == i) pi = t; else Ti[ti++] = l;
          if (l == j) pj = t; else Tj[tj++] = l;
          ++t;
        }
        double cof = tf_det(sA[s][v], Ti, Tj, t - 1);
        if ((pi + pj + t) & 1) cof = -cof;                 // cofactor sign (-1)^(pi+pj) times (-1)^|T|
        acc += sgq[s][v][S] * cof;
      }
    }
    if (chunk < 4) spart[chunk][t64] = acc;
  }
  __syncthreads();
  if (tid < 32 * nsets) {
    const int s = tid >> 5, v = (tid >> 4) & 1, i = (tid >> 2) & 3, j = tid & 3;
    const int nch = ((int)blockDim.x >> 6) < 4 ? ((int)blockDim.x >> 6) : 4;
    double acc = spart[0][tid];
    for (int ch = 1; ch < nch; ++ch) acc += spart[ch][tid];
    sgA[s][v][i * 4 + j] = acc;
  }
  __syncthreads();
  if (tid < 16 * nsets) {
    const int s = tid >> 4, i = (tid >> 2) & 3, j = tid & 3;
    const TfBwdSet& st = s ? s1 : s0;
    if (i < n && j < n) {
      const float v = (float)(sgA[s][0][i * 4 + j] + sgA[s][1][i * 4 + j]);
      if (st.gA) st.gA[(size_t)blk * n * n + i * n + j] = v;
      float* l = s ? lA1 : lA0;
      if (l) l[i * n + j] = v;
    }
  } else if (tid >= 32 && tid < 36) {
    const int i = tid - 32;
    if (i < n && gb) {
      double acc = 0.0;
      for (int s = 0; s < nsets; ++s)
        for (int j = 0; j < n; ++j) acc -= sgA[s][1][i * 4 + j] * (double)c[blk * n + j];
      gb[blk * n + i] = (float)acc;
    }
  } else if (tid >= 48 && tid < 52) {
    const int j = tid - 48;
    if (j < n && gc) {
      double acc = 0.0;
      for (int s = 0; s < nsets; ++s)
        for (int i = 0; i < n; ++i) acc -= sgA[s][1][i * 4 + j] * (double)b[blk * n + i];
      gc[blk * n + j] = (float)acc;
    }
  }
}

__global__ __launch_bounds__(256) void k_tf_coefs_bwd(TfBwdSet s0, TfBwdSet s1, const float* __restrict__ b,
                                                     const float* __restrict__ c, int n, float* __restrict__ gb,
                                                     float* __restrict__ gc) {
  const int blk = blockIdx.x;
  tf_coefs_bwd_block(s0, s1, s0.grec + (size_t)blk * TF_REC, s1.grec ? s1.grec + (size_t)blk * TF_REC : nullptr, b, c, n,
                     blk, threadIdx.x, gb, gc, nullptr, nullptr);
}

extern "C" int gfdn_tf_coefs_bwd(const float* A0, const float* inv_gamma0, const float* grec0, const float* A1,
                                 const float* inv_gamma1, const float* grec1, const float* b, const float* c,
                                 int nblk, int nper, float* gA0, float* gA1, float* gb, float* gc, void* stream) {
  if (!A0 || !grec0 || !b || !c || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (A1 && !grec1) return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  TfBwdSet s0{A0, inv_gamma0, grec0, gA0};
  TfBwdSet s1{A1, inv_gamma1, grec1, gA1};
  hipLaunchKernelGGL(k_tf_coefs_bwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, s0, s1, b, c, nper, gb, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// srec[r] = sum_p rows[r * nparts + p] for the TF_REC record rows of one block, by the 4 wavefronts of a 256-thread
// workgroup: wave w takes rows w, w + 4, ... -- the loads of its eight rows are issued together (they were eight dependent
// load -> crossbar-sum round trips, a third of the tail kernel's time), the sums in the order of k_tf_rows_sum.
__device__ __forceinline__ void tf_sum_record_rows(const float* __restrict__ rows, int nparts, float* srec) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[TF_REC / 4];
#pragma unroll
  for (int q = 0; q < TF_REC / 4; ++q) {
    const float* row = rows + (size_t)(w + 4 * q) * nparts;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = lane;
    for (; p + 192 < nparts; p += 256) {
      a0 += row[p];
      a1 += row[p + 64];
      a2 += row[p + 128];
      a3 += row[p + 192];
    }
    for (; p < nparts; p += 64) a0 += row[p];
    acc[q] = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int q = 0; q < TF_REC / 4; ++q) {
    const float sum = wave_sum_full(acc[q]);            // (whole waves: VALU sums)
    if (lane == 0) srec[w + 4 * q] = sum;
  }
}

// The tail of the band bank's backward in ONE launch, one workgroup per block (group): sum the partial record rows of
// the output-stage adjoint (gpart0[(blk * 32 + e) * nparts0 + p], as k_tf_compose_bwd_rec leaves them; nparts0 = 1:
// summed records), records -> (dL/dQQ, dL/dM_raw, dL/db, dL/dc) as k_tf_coefs_bwd, then the adjoint of
// Q = expm(skew(M)), QQ = Q Q as k_ortho_bwd with dL/dM_raw added: dL/dM.  Same arithmetic as the three separate
// launches (row sums in the order of k_tf_rows_sum, dL/dQQ and dL/dM_raw rounded to float32 in between): the results
// are bit-identical, which is how tests/test_gpu_blocktf.py checks it.
__global__ __launch_bounds__(256) void k_tf_param_grads(TfBwdSet s0, TfBwdSet s1, int nparts0, const float* __restrict__ b,
                                                        const float* __restrict__ c, int n, const float* __restrict__ M,
                                                        const float* __restrict__ gQ, const float* __restrict__ Q,
                                                        float* __restrict__ gb, float* __restrict__ gc,
                                                        float* __restrict__ gM) {
  extern __shared__ double tfp_lds[];
  __shared__ float srec[TF_REC], sG[2][16];
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (nparts0 > 1) {
    tf_sum_record_rows(s0.grec + (size_t)blk * TF_REC * nparts0, nparts0, srec);
  } else if (tid < TF_REC) {
    srec[tid] = s0.grec[(size_t)blk * TF_REC + tid];
  }
  __syncthreads();
  tf_coefs_bwd_block(s0, s1, srec, s1.grec ? s1.grec + (size_t)blk * TF_REC : nullptr, b, c, n, blk, tid, gb, gc, sG[0],
                     sG[1]);
  __syncthreads();
  const size_t off = (size_t)blk * n * n;
  ortho_bwd_group(tfp_lds, M + off, n, gQ ? gQ + off : nullptr, sG[0], Q ? Q + off : nullptr, s1.A ? sG[1] : nullptr,
                  gM + off);
}

extern "C" int gfdn_tf_param_grads(const float* A0, const float* inv_gamma0, const float* grec0, int nparts0,
                                   const float* A1, const float* inv_gamma1, const float* grec1, const float* b,
                                   const float* c, int nblk, int nper, const float* M, const float* gQ, const float* Q,
                                   float* gb, float* gc, float* gM, void* stream) {
  if (!A0 || !grec0 || !b || !c || !M || !gM || nblk <= 0 || nper <= 0 || nparts0 <= 0) return GFDN_E_BADARG;
  if (A1 && !grec1) return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  TfBwdSet s0{A0, inv_gamma0, grec0, nullptr};
  TfBwdSet s1{A1, inv_gamma1, grec1, nullptr};
  const size_t lds = ortho_bwd_lds_doubles(nper) * sizeof(double);
  // (four wavefronts: measured 29 us against 50 with one -- the row sums want the loads of several rows in flight, and
  // the barriers of the later stages cost nothing measurable)
  hipLaunchKernelGGL(k_tf_param_grads, dim3(nblk), dim3(256), lds, (hipStream_t)stream, s0, s1, nparts0, b, c, nper, M, gQ,
                     Q, gb, gc, gM);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// The tail of the band bank's step AND the head of the next one in ONE launch (single-process training: nothing sits
// between the gradients and the update), one workgroup per block (group):
//   k_tf_param_grads (above) -> Adam on the block's own entries of M, b, c (trainer.py:475; maths of optim.hip's k_adam:
//   the flat buffers' element i of the three leaves at offM + blk n n + e, offb + blk n + i, offc + blk n + j)
//   -> k_tf_ortho_coefs on the UPDATED block: Q, QQ and both record sets of the next step.
// A step then neither starts with the records launch nor ends with the optimiser launch for these leaves: the chain
// records pass -> [this] -> next step's group responses has one launch where it had three and two stream joins.  The update
// reads t = step_count + 1; the LAST workgroup to finish writes t back (every workgroup read it before reporting in) and
// re-arms the counter.  The rest of the flat buffer (the gain network) is stepped by its own launch on its own counter
// (FlatAdam.step_range(second=True)).  The updated values go to the next records through LDS, not back through memory.
// (probe builds only, tools/build_probe_lib.sh ... -DTFT_TIMING: wall-clock stamps of the tail's stages, 100 MHz)
#ifdef TFT_TIMING
__device__ unsigned long long tft_times[64 * 8];
#define TFT_STAMP(slot)                                                                                              \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < 64) tft_times[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
extern "C" int gfdn_probe_tf_tail_times(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tft_times), sizeof(unsigned long long) * n);
}
#else
#define TFT_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_tf_tail(TfBwdSet s0, TfBwdSet s1, int nparts0, const float* b, const float* c,
                                                 int n, const float* M, const float* gQ, const float* Q, float* gb,
                                                 float* gc, float* gM, TfAdam ad, float* Qn, float* QQn, float* coef0,
                                                 float* coef1) {
  extern __shared__ double tfp_lds[];
  __shared__ float srec[TF_REC], sG[2][16], sM[16], sb[4], sc[4], lQQ[16];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const float t = ad.step_count[0] + 1.0f;
  TFT_STAMP(0);
  const size_t off = (size_t)blk * n * n;
  if (nparts0 > 1) {
    tf_sum_record_rows(s0.grec + (size_t)blk * TF_REC * nparts0, nparts0, srec);
  } else if (tid < TF_REC) {
    srec[tid] = s0.grec[(size_t)blk * TF_REC + tid];
  }
  __syncthreads();
  TFT_STAMP(1);
  tf_coefs_bwd_block(s0, s1, srec, s1.grec ? s1.grec + (size_t)blk * TF_REC : nullptr, b, c, n, blk, tid, gb, gc, sG[0],
                     sG[1]);
  __syncthreads();
  TFT_STAMP(2);
  ortho_bwd_group(tfp_lds, M + off, n, gQ ? gQ + off : nullptr, sG[0], Q ? Q + off : nullptr, s1.A ? sG[1] : nullptr,
                  gM + off);
  TFT_STAMP(3);
  // ---- Adam on the block's own entries: every element is updated by the thread that wrote its gradient
  // (tf_coefs_bwd_block: dL/db[i] by thread 32 + i, dL/dc[j] by thread 48 + j; ortho_bwd_group: dL/dM[e] by thread e)
  const float bc1 = 1.0f - powf(ad.b1, t), bc2_sqrt = sqrtf(1.0f - powf(ad.b2, t));
  if (tid < n * n) sM[tid] = tf_adam_elem(ad, ad.offM + (int)off + tid, gM[off + tid], bc1, bc2_sqrt);
  if (tid >= 32 && tid < 32 + n) sb[tid - 32] = tf_adam_elem(ad, ad.offb + blk * n + tid - 32, gb[blk * n + tid - 32], bc1, bc2_sqrt);
  if (tid >= 48 && tid < 48 + n) sc[tid - 48] = tf_adam_elem(ad, ad.offc + blk * n + tid - 48, gc[blk * n + tid - 48], bc1, bc2_sqrt);
  __syncthreads();
  TFT_STAMP(4);
  // ---- the next step's head on the updated block (k_tf_ortho_coefs)
  {
    double* A = tfp_lds;
    for (int e = tid; e < n * n; e += blockDim.x) A[e] = skew_elem(sM, n, e / n, e % n);
    __syncthreads();
    const double* E = expm_lds(A, n);
    for (int e = tid; e < n * n; e += blockDim.x) {
      Qn[off + e] = (float)E[e];
      const int i = e / n, j = e - i * n;
      double acc = 0.0;
      for (int q = 0; q < n; ++q) acc += E[i * n + q] * E[q * n + j];
      const float v = (float)acc;
      QQn[off + e] = v;
      lQQ[e] = v;
    }
    __syncthreads();
    TFT_STAMP(5);
    tf_coefs_block(lQQ, s0.ig, coef0, sM, nullptr, coef1, sb, sc, blk, n, tid);
  }
  TFT_STAMP(6);
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(ad.block_counter, 1u) == gridDim.x - 1) {
      ad.step_count[0] = t;
      ad.block_counter[0] = 0u;
    }
  }
}

extern "C" int gfdn_tf_tail(const float* A0, const float* inv_gamma0, const float* grec0, int nparts0, const float* A1,
                            const float* grec1, const float* b, const float* c, int nblk, int nper, const float* M,
                            const float* gQ, const float* Q, float* gb, float* gc, float* gM, float* flat_p, float* flat_m,
                            float* flat_v, const unsigned char* seg, const float* lr_seg, float* step_count,
                            unsigned int* block_counter, int offM, int offb, int offc, float beta1, float beta2, float eps,
                            float* Q_next, float* QQ_next, float* coef_next, float* coef_sub_next, void* stream) {
  if (!A0 || !grec0 || !A1 || !grec1 || !b || !c || !M || !gM || !gb || !gc || !flat_p || !flat_m || !flat_v || !seg ||
      !lr_seg || !step_count || !block_counter || !Q_next || !QQ_next || !coef_next || !coef_sub_next || nblk <= 0 ||
      nper <= 0 || nparts0 <= 0 || offM < 0 || offb < 0 || offc < 0)
    return GFDN_E_BADARG;
  if (nper > 4) return GFDN_E_UNSUPPORTED;
  TfBwdSet s0{A0, inv_gamma0, grec0, nullptr};
  TfBwdSet s1{A1, nullptr, grec1, nullptr};
  TfAdam ad{flat_p, flat_m, flat_v, seg, lr_seg, step_count, block_counter, offM, offb, offc, beta1, beta2, eps};
  size_t lds = ortho_bwd_lds_doubles(nper);
  const size_t lds_head = expm_lds_doubles(nper);
  if (lds_head > lds) lds = lds_head;
  hipLaunchKernelGGL(k_tf_tail, dim3(nblk), dim3(256), lds * sizeof(double), (hipStream_t)stream, s0, s1, nparts0, b, c, nper,
                     M, gQ, Q, gb, gc, gM, ad, Q_next, QQ_next, coef_next, coef_sub_next);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// out0[r] (r < n0) / out1[r - n0] = sum_p part[r][p]: one wavefront per row, fixed order
// pick_out (optional): rows r < n0 with r % 32 == 15 also go to pick_out[r / 32] (the loss partials of the
// colorless pass, parked in the slot of P_full).
__global__ __launch_bounds__(256) void k_tf_rows_sum(const float* __restrict__ part, int cols, int rows,
                                                     float* __restrict__ out0, int n0, float* __restrict__ out1,
                                                     float* __restrict__ pick_out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* row = part + (size_t)r * cols;
  int p = lane;
  for (; p + 192 < cols; p += 256) {
    s0 += row[p];
    s1 += row[p + 64];
    s2 += row[p + 128];
    s3 += row[p + 192];
  }
  for (; p < cols; p += 64) s0 += row[p];
  const float s = wave_sum_full((s0 + s1) + (s2 + s3));            // (a wave is wholly in or out of range)
  if (lane == 0) {
    if (r < n0) {
      out0[r] = s;
      if (pick_out && (r & 31) == 15) pick_out[r >> 5] = s;
    } else {
      out1[r - n0] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// thread-per-(bin, block) launches
// ------------------------------------------------------------------------------------------
struct TfArgs {
  const double* turns;
  const double* logr;
  int K, nblk, nper;
  const float* coef;
  const float* delays;
  const float* scale;      // per block, multiplies T (NULL: 1)
};

static int tf_items(int nblk) { return (256 / nblk) * nblk; }
static int tf_parts_host(int K, int nblk) {
  const int rows = 256 / nblk;
  const int full = (K + rows - 1) / rows;
  int parts = (K + 8 * rows - 1) / (8 * rows);
  if (parts < 1024) parts = full < 1024 ? full : 1024;
  if (parts > TF_MAX_PARTS) parts = TF_MAX_PARTS;
  return parts;
}
// out[r] = sum_p part[r * cols + p], one wavefront per row, fixed order (the second stage of every partial-sum scheme
// of this file; also used for the gains partials of gfdn_irfft_odd_pairs_gains_bwd)
extern "C" int gfdn_tf_rows_sum(const float* part, int cols, int rows, float* out, void* stream) {
  if (!part || !out || cols <= 0 || rows <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_tf_rows_sum, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, cols, rows, out, rows,
                     (float*)nullptr, (float*)nullptr);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_tf_parts(int K, int nblk) {
  if (K <= 0 || nblk <= 0 || nblk > TF_MAXBLK) return 0;
  return tf_parts_host(K, nblk);
}

static int tf_args_ok(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays) {
  if (!turns || !coef || !delays || K <= 0 || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 4 || nblk > TF_MAXBLK) return GFDN_E_UNSUPPORTED;
  return 0;
}

// T[k][blk] (bin-major) = scale_blk * Num / Den
__global__ __launch_bounds__(256) void k_tf_eval(TfArgs a, float2* __restrict__ T) {
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= (long long)a.K * a.nblk) return;
  const int k = (int)(w / a.nblk), blk = (int)(w - (long long)k * a.nblk);
  float P[16], Q[16], m[4];
  tf_load(a.coef, a.delays, blk, a.nper, P, Q, m);
  float2 e[16], num, den;
  tf_phasors(a.turns, a.logr, k, m, e);
  tf_numden(P, Q, e, num, den);
  float2 t = cmul(num, cinv(den));
  if (a.scale) t = cscale(t, a.scale[blk]);
  T[w] = t;
}

extern "C" int gfdn_tf_eval(const double* turns, const double* logr, int K, int nblk, int nper,
                            const float* coef, const float* delays, const float* scale, float* T,
                            void* stream) {
  int rc = tf_args_ok(turns, K, nblk, nper, coef, delays);
  if (rc) return rc;
  if (!T) return GFDN_E_BADARG;
  TfArgs a{turns, logr, K, nblk, nper, coef, delays, scale};
  const long long items = (long long)K * nblk;
  hipLaunchKernelGGL(k_tf_eval, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     (float2*)T);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// partial[blk * gridDim.x + part] = sum over this workgroup's bins of |T|^2
__global__ __launch_bounds__(256) void k_tf_energy(TfArgs a, float* __restrict__ partial, int items) {
  __shared__ float s_e[256];
  const int nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  float P[16], Q[16], m[4];
  tf_load(a.coef, a.delays, blk, a.nper, P, Q, m);
  float acc = 0.f;
#pragma unroll 1
  for (int k = blockIdx.x * rows + krow; live && k < a.K; k += gridDim.x * rows) {
    float2 e[16], num, den;
    tf_phasors(a.turns, a.logr, k, m, e);
    tf_numden(P, Q, e, num, den);
    acc += (num.x * num.x + num.y * num.y) / (den.x * den.x + den.y * den.y);
  }
  s_e[threadIdx.x] = live ? acc : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < nblk; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}

// Uniform grids on the unit circle (turns[k] = turns[0] + k dturn: the reference's z = exp(2 pi i rfftfreq), dataloader.py:
// 552-566): a thread takes RUNS of TF_RUN consecutive bins, evaluates the four phasors exactly at the head of a run
// and steps them by the constant rotations z_1^{m_i} inside it -- 4 complex products per bin instead of 4 float64
// range reductions + sincos (the rounding of TF_RUN - 1 products, ~1e-7 per step, stays far below the 1e-5 the
// evaluation itself carries).
#define TF_RUN 8
__device__ __forceinline__ float2 tf_rot(double dturn, float m) {
  double t = (double)m * dturn;
  t -= rint(t);
  float s, c;
  sincospif(2.0f * (float)t, &s, &c);
  return make_float2(c, s);
}

__global__ __launch_bounds__(256) void k_tf_energy_runs(TfArgs a, double dturn, float* __restrict__ partial, int items) {
  __shared__ float s_e[256];
  const int nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  float P[16], Q[16], m[4];
  tf_load(a.coef, a.delays, blk, a.nper, P, Q, m);
  const float2 r0 = tf_rot(dturn, m[0]), r1 = tf_rot(dturn, m[1]), r2 = tf_rot(dturn, m[2]), r3 = tf_rot(dturn, m[3]);
  const int nruns = (a.K + TF_RUN - 1) / TF_RUN;
  float acc = 0.f;
#pragma unroll 1
  for (int run = blockIdx.x * rows + krow; live && run < nruns; run += gridDim.x * rows) {
    const int k0 = run * TF_RUN;
    float2 e[16];
    e[1] = tf_zpow(a.turns, nullptr, k0, m[0]);
    e[2] = tf_zpow(a.turns, nullptr, k0, m[1]);
    e[4] = tf_zpow(a.turns, nullptr, k0, m[2]);
    e[8] = tf_zpow(a.turns, nullptr, k0, m[3]);
#pragma unroll 2
    for (int j = 0; j < TF_RUN; ++j) {
      if (k0 + j < a.K) {
        float2 num, den;
        tf_subsets(e);
        tf_numden(P, Q, e, num, den);
        acc += (num.x * num.x + num.y * num.y) / (den.x * den.x + den.y * den.y);
        e[1] = cmul(e[1], r0);
        e[2] = cmul(e[2], r1);
        e[4] = cmul(e[4], r2);
        e[8] = cmul(e[8], r3);
      }
    }
  }
  s_e[threadIdx.x] = live ? acc : 0.f;
  __syncthreads();
  for (int bq = threadIdx.x; bq < nblk; bq += 256) {
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += s_e[t];
    partial[(size_t)bq * gridDim.x + blockIdx.x] = sum;
  }
}

// E = sum / K -> energy, scale = E^(-1/2) (what T and the numerator coefficients scale by once b, c are
// divided by E^(1/4): trainer.py:317-332), and the in-place rescale of b, c
// gains / gains_scaled ((nblk / G) Bper, G) or NULL: column g % G of band g / G of the receiver gains times the block's scale
// (the scale folded into the gains: see gfdn_tf_energy_gains)
__global__ __launch_bounds__(256) void k_tf_energy_finish(const float* __restrict__ partial, int nparts, int K,
                                                          int nper, float* __restrict__ b, float* __restrict__ c,
                                                          float* __restrict__ energy, float* __restrict__ scale,
                                                          const float* __restrict__ gains, float* __restrict__ gains_scaled,
                                                          int Bper, int G) {
  __shared__ float s_red[16];
  const int g = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)g * nparts + p];
  s = block_sum(s, s_red);
  const float E = s / (float)K;
  if (threadIdx.x == 0) {
    if (energy) energy[g] = E;
    if (scale) scale[g] = 1.0f / sqrtf(E);
  }
  if (gains_scaled) {
    const float sc = 1.0f / sqrtf(E);
    const int band = g / G, col = g - band * G;
    for (int r = threadIdx.x; r < Bper; r += 256) {
      const size_t i = ((size_t)band * Bper + r) * G + col;
      gains_scaled[i] = gains[i] * sc;
    }
  }
  if (b && c) {
    const float d = powf(E, 0.25f);
    for (int i = threadIdx.x; i < nper; i += 256) {
      b[g * nper + i] /= d;
      c[g * nper + i] /= d;
    }
  }
}

extern "C" size_t gfdn_tf_work_bytes(int nblk) {
  return (size_t)TF_MAX_PARTS * (nblk > 0 ? nblk : 1) * sizeof(float);
}

static int tf_energy_run(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                         const float* delays, float* b, float* c, float* energy, float* scale, void* work, int phase,
                         double dturn, const float* gains, float* gains_scaled, int Bper, int G, void* stream);
extern "C" int gfdn_tf_energy(const double* turns, const double* logr, int K, int nblk, int nper,
                              const float* coef, const float* delays, float* b, float* c, float* energy,
                              float* scale, void* work, int phase, double dturn, void* stream) {
  return tf_energy_run(turns, logr, K, nblk, nper, coef, delays, b, c, energy, scale, work, phase, dturn, nullptr, nullptr, 0, 0,
                       stream);
}
// ... whose finish also stores gains_scaled[band Bper + r][g] = gains[..][g] scale[band G + g] (nblk = bands x G blocks): the
// normalisation scale folded into the receiver gains -- H = sum_g gain[b][g] (s_g T_g) + direct is linear in both -- so that
// the launches of the linear step can run on group signals of the UNSCALED functions and nothing waits for the scale
extern "C" int gfdn_tf_energy_gains(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                                    const float* delays, float* b, float* c, float* energy, float* scale, void* work,
                                    int phase, double dturn, const float* gains, float* gains_scaled, int Bper, int G,
                                    void* stream) {
  if (!gains || !gains_scaled || Bper <= 0 || G <= 0 || nblk % G || !(phase & 2)) return GFDN_E_BADARG;
  return tf_energy_run(turns, logr, K, nblk, nper, coef, delays, b, c, energy, scale, work, phase, dturn, gains, gains_scaled, Bper,
                       G, stream);
}
static int tf_energy_run(const double* turns, const double* logr, int K, int nblk, int nper, const float* coef,
                         const float* delays, float* b, float* c, float* energy, float* scale, void* work, int phase,
                         double dturn, const float* gains, float* gains_scaled, int Bper, int G, void* stream) {
  int rc = tf_args_ok(turns, K, nblk, nper, coef, delays);
  if (rc) return rc;
  if (!work || (!b) != (!c) || !(phase & 3)) return GFDN_E_BADARG;
  TfArgs a{turns, logr, K, nblk, nper, coef, delays, nullptr};
  const int nparts = tf_parts_host(K, nblk);
  hipStream_t s = (hipStream_t)stream;
  if (phase & 1) {
    if (dturn != 0.0 && !logr)
      hipLaunchKernelGGL(k_tf_energy_runs, dim3(nparts), dim3(256), 0, s, a, dturn, (float*)work, tf_items(nblk));
    else
      hipLaunchKernelGGL(k_tf_energy, dim3(nparts), dim3(256), 0, s, a, (float*)work, tf_items(nblk));
    GFDN_LAUNCH_CHECK();
  }
  if (phase & 2) {
    hipLaunchKernelGGL(k_tf_energy_finish, dim3(nblk), dim3(256), 0, s, (const float*)work, nparts, K, nper, b, c,
                       energy, scale, gains, gains_scaled, Bper, G);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}

// Colorless pass (colorless_fdn/losses.py:20-73 on Hout[:, g], trainer.py:298-304): S' = scale T,
// loss_g = mean_k (|S'| - 1)^p (p = 4 where asym and |S'| - 1 > 1, else 2), and dL/dcoef of
// gscale * sum_g loss_g with respect to the records of S' (numerator coefficients P' = scale P):
//     dL/dP'_S = Re(gS' conj(e_S / Den)) ,   dL/dQ_S = -Re(gS' conj(S' e_S / Den)).
// gpart[(blk * 32 + e) * nparts + part]: e < 15 dL/dP'_S, e = 15 the loss partial, 16 + S dL/dQ_S.
template <bool RUNS>
__global__ __launch_bounds__(256) void k_tf_colorless(TfArgs a, double dturn, int asym, float gscale,
                                                      float* __restrict__ gpart, int items) {
  extern __shared__ float tf_acc[];        // [items][33]
  const int nblk = a.nblk;
  const bool live = (int)threadIdx.x < items;
  const int blk = live ? threadIdx.x % nblk : 0;
  const int krow = threadIdx.x / nblk, rows = items / nblk;
  float P[16], Q[16], m[4];
  tf_load(a.coef, a.delays, blk, a.nper, P, Q, m);
  const float sc = a.scale ? a.scale[blk] : 1.0f;
  const float invK = 1.0f / (float)a.K;
  float2 r0, r1, r2, r3;
  if (RUNS) {
    r0 = tf_rot(dturn, m[0]);
    r1 = tf_rot(dturn, m[1]);
    r2 = tf_rot(dturn, m[2]);
    r3 = tf_rot(dturn, m[3]);
  }
  float aP[16], aQ[16];
#pragma unroll
  for (int S = 0; S < 16; ++S) aP[S] = aQ[S] = 0.f;
  constexpr int RUN = RUNS ? TF_RUN : 1;
  const int nruns = (a.K + RUN - 1) / RUN;
#pragma unroll 1
  for (int run = blockIdx.x * rows + krow; live && run < nruns; run += gridDim.x * rows) {
    const int k0 = run * RUN;
    float2 e[16];
    e[1] = tf_zpow(a.turns, RUNS ? nullptr : a.logr, k0, m[0]);
    e[2] = tf_zpow(a.turns, RUNS ? nullptr : a.logr, k0, m[1]);
    e[4] = tf_zpow(a.turns, RUNS ? nullptr : a.logr, k0, m[2]);
    e[8] = tf_zpow(a.turns, RUNS ? nullptr : a.logr, k0, m[3]);
#pragma unroll 1
    for (int j = 0; j < RUN; ++j) {
      if (k0 + j >= a.K) break;
      float2 num, den;
      tf_subsets(e);
      tf_numden(P, Q, e, num, den);
      const float2 dinv = cinv(den);
      const float2 s = cscale(cmul(num, dinv), sc);
      const float mag = sqrtf(s.x * s.x + s.y * s.y);
      const float d = mag - 1.0f, d2 = d * d;
      const bool four = asym && (d > 1.0f);
      aP[15] += (four ? d2 * d2 : d2) * invK;
      const float dl = four ? 4.0f * d2 * d : 2.0f * d;
      const float f = (mag > 0.f) ? gscale * invK * dl / mag : 0.f;
      const float2 gs = make_float2(f * s.x, f * s.y);
      const float2 u = cmulc(gs, dinv);                  // gS' conj(1 / Den)
      const float2 v = cmulc(u, s);                      // u conj(S')
      aP[0] += u.x;
      aQ[0] -= v.x;
#pragma unroll
      for (int S = 1; S < 16; ++S) {
        if (S < 15) aP[S] += u.x * e[S].x + u.y * e[S].y;
        aQ[S] -= v.x * e[S].x + v.y * e[S].y;
      }
      if (RUNS) {
        e[1] = cmul(e[1], r0);
        e[2] = cmul(e[2], r1);
        e[4] = cmul(e[4], r2);
        e[8] = cmul(e[8], r3);
      }
    }
  }
  if (live) {
#pragma unroll
    for (int S = 0; S < 16; ++S) {
      tf_acc[threadIdx.x * 33 + S] = aP[S];
      tf_acc[threadIdx.x * 33 + 16 + S] = aQ[S];
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < nblk * TF_REC; o += 256) {
    const int bq = o >> 5, e = o & 31;
    float sum = 0.f;
    for (int t = bq; t < items; t += nblk) sum += tf_acc[t * 33 + e];
    gpart[(size_t)o * gridDim.x + blockIdx.x] = sum;
  }
}

extern "C" size_t gfdn_tf_gpart_bytes(int nblk) {
  return (size_t)TF_MAX_PARTS * (nblk > 0 ? nblk : 1) * TF_REC * sizeof(float);
}

extern "C" int gfdn_tf_colorless(const double* turns, const double* logr, int K, int nblk, int nper,
                                 const float* coef, const float* delays, const float* scale, int asym,
                                 float gscale, float* grec, float* loss, void* work, double dturn, void* stream) {
  int rc = tf_args_ok(turns, K, nblk, nper, coef, delays);
  if (rc) return rc;
  if (!grec || !work) return GFDN_E_BADARG;
  TfArgs a{turns, logr, K, nblk, nper, coef, delays, scale};
  const int nparts = tf_parts_host(K, nblk), items = tf_items(nblk);
  hipStream_t s = (hipStream_t)stream;
  if (dturn != 0.0 && !logr)
    hipLaunchKernelGGL(k_tf_colorless<true>, dim3(nparts), dim3(256), (size_t)items * 33 * sizeof(float), s, a, dturn,
                       asym, gscale, (float*)work, items);
  else
    hipLaunchKernelGGL(k_tf_colorless<false>, dim3(nparts), dim3(256), (size_t)items * 33 * sizeof(float), s, a, 0.0,
                       asym, gscale, (float*)work, items);
  GFDN_LAUNCH_CHECK();
  const int rows = nblk * TF_REC;
  hipLaunchKernelGGL(k_tf_rows_sum, dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)work, nparts, rows, grec, rows,
                     (float*)nullptr, loss);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// output stage straight from the coefficient records (model.py:583-619 with trainer.py:459 folded in):
//   H[b][k] = (sum_g rgain[b][g] scale_g T_g(z_k) + direct[rows[b]][k]) * filt[k]
// band-stacked: blocks band * G + g, items band * B + b, filter row band.
// ------------------------------------------------------------------------------------------
struct TfCompose {
  const double* turns;
  const double* logr;
  int K, G, nper, B;
  const float* coef;
  const float* delays;
  const float* scale;
  const float* rgain;
  const float2* filt;
  int ldf;
  int fold;                  // records pass with a.scale: gH is dL/d(T filt) of the UNSCALED functions (the scale sits in the
                             // receiver gains): dL/dT' = dL/dT / scale
};

#define TFC_BCH 8
__global__ __launch_bounds__(256) void k_tf_compose_fwd(TfCompose a, const float2* __restrict__ direct, int ldd,
                                                        const long long* __restrict__ drows,
                                                        float2* __restrict__ H, int ldh, float2* __restrict__ Tsave,
                                                        float2* __restrict__ Tquad) {
  __shared__ TfBlock tab[TF_MAXG];
  const int band = blockIdx.y, G = a.G, B = a.B;
  tf_stage(a.coef, a.delays, band * G, G, a.nper, tab);
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.K) return;
  const float* rgain = a.rgain + (size_t)band * B * G;
  if (drows) drows += (size_t)band * B;
  else if (direct) direct += (size_t)band * B * ldd;
  if (H) H += (size_t)band * B * ldh;
  else direct = nullptr;
  // the first receivers' direct-path loads fly while the transfer functions are evaluated (with one wave of
  // workgroups on the chip the launch would otherwise run as a compute phase followed by a memory phase)
  float2 d[TFC_BCH];
#pragma unroll
  for (int bb = 0; bb < TFC_BCH; ++bb)
    d[bb] = (direct && bb < B) ? direct[(size_t)(drows ? drows[bb] : bb) * ldd + k] : make_float2(0.f, 0.f);
  float2 T[TF_MAXG];
#pragma unroll
  for (int g = 0; g < TF_MAXG; ++g) {
    T[g] = make_float2(0.f, 0.f);
    if (g < G) {
      float2 e[16], num, den;
      tf_phasors(a.turns, a.logr, k, tab[g].m, e);
      tf_numden(tab[g].P, tab[g].Q, e, num, den);
      T[g] = cmul(num, cinv(den));
      if (a.scale) T[g] = cscale(T[g], a.scale[band * G + g]);
      if (Tsave) Tsave[(size_t)(band * G + g) * a.K + k] = T[g];
    }
  }
  if (Tquad) {                         // the band's four transfer functions of a bin side by side (zeros beyond G)
    float4* q = (float4*)(Tquad + ((size_t)band * a.K + k) * 4);
    q[0] = make_float4(T[0].x, T[0].y, T[1].x, T[1].y);
    q[1] = make_float4(T[2].x, T[2].y, T[3].x, T[3].y);
  }
  if (!H) return;                      // transfer functions only (the transform's first pass forms H itself)
  const float2 f = a.filt ? a.filt[(size_t)band * a.ldf + k] : make_float2(1.f, 0.f);
  for (int b0 = 0; b0 < B; b0 += TFC_BCH) {
    float2 dn[TFC_BCH];
#pragma unroll
    for (int bb = 0; bb < TFC_BCH; ++bb) {               // next chunk's loads in flight over this chunk's stores
      const int b = b0 + TFC_BCH + bb;
      dn[bb] = (direct && b < B) ? direct[(size_t)(drows ? drows[b] : b) * ldd + k] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int bb = 0; bb < TFC_BCH; ++bb) {
      const int b = b0 + bb;
      if (b < B) {
        float2 h = d[bb];
#pragma unroll
        for (int g = 0; g < TF_MAXG; ++g) {
          if (g < G) {
            const float rg = rgain[b * G + g];
            h.x += rg * T[g].x;
            h.y += rg * T[g].y;
          }
        }
        if (a.filt) h = cmul(h, f);
        H[(size_t)b * ldh + k] = h;
      }
    }
#pragma unroll
    for (int bb = 0; bb < TFC_BCH; ++bb) d[bb] = dn[bb];
  }
}

static int tf_compose_ok(const double* turns, int K, int nbands, int G, int nper, int B, const float* coef,
                         const float* delays, const float* rgain) {
  if (!turns || !coef || !delays || !rgain || K <= 0 || nbands <= 0 || G <= 0 || nper <= 0 || B <= 0)
    return GFDN_E_BADARG;
  if (nper > 4 || G > TF_MAXG || nbands > 65535) return GFDN_E_UNSUPPORTED;
  return 0;
}

extern "C" int gfdn_tf_compose_fwd(const double* turns, const double* logr, int K, int nbands, int G, int nper,
                                   const float* coef, const float* delays, const float* scale,
                                   const float* rgain, int B, const float* direct, int ldd,
                                   const long long* direct_rows, const float* filt, int ldf, float* H, int ldh,
                                   float* Tsave, float* Tquad, void* stream) {
  int rc = tf_compose_ok(turns, K, nbands, G, nper, B, coef, delays, rgain);
  if (rc) return rc;
  if ((!H && !Tsave && !Tquad) || (H && ldh < K) || (direct && ldd < K) || (filt && nbands > 1 && ldf < K)) return GFDN_E_BADARG;
  TfCompose a{turns, logr, K, G, nper, B, coef, delays, scale, rgain, (const float2*)filt, ldf};
  hipLaunchKernelGGL(k_tf_compose_fwd, dim3((K + 255) / 256, nbands), dim3(256), 0, (hipStream_t)stream, a,
                     (const float2*)direct, ldd, direct ? direct_rows : nullptr, (float2*)H, ldh, (float2*)Tsave,
                     (float2*)Tquad);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// Backward of the output stage, two streaming launches over dL/dH (the second read comes from the last-level cache):
//   gains  : rg_partial[(band * B * G + b * G + g) * nchunk + chunk] = partial of
//            dL/drgain[b][g] = sum_k Re(dL/dH[b][k] conj(filt[k] T'_g[k]))  -- 8 receivers x a stripe of bins per
//            workgroup, 32 per-thread sums, one block reduction at the end;
//   records: gpart[((band * G + g) * 32 + e) * nparts + part] = dL/dcoef partials of the (scaled) records, as
//            k_tf_colorless -- wavefront g of a workgroup owns group g for tiles of 64 bins: it folds the receivers'
//            dL/dH into dL/dT'_g = conj(filt) sum_b rgain[b][g] dL/dH[b] (the four wavefronts read the same lines:
//            L1 hits), rebuilds the phasors and the denominator, and adds into its 30 accumulators; no LDS, no
//            barrier.  T' = the forward's saved (scaled, unfiltered) group transfer functions (nbands * G, K).
#define TFB_T 64
#define TFG_R 8
#define TFG_PRIO 0       // wave priority of the gains pass (it heads the longer branch of the tail)
__global__ __launch_bounds__(256) void k_tf_gain_grad(const float2* __restrict__ Tsave, int K, int G, int B,
                                                      const float2* __restrict__ filt, int ldf,
                                                      const float2* __restrict__ gH, int ldh,
                                                      float* __restrict__ rg_partial, int ngrp) {
  if (TFG_PRIO) __builtin_amdgcn_s_setprio(TFG_PRIO);
  __shared__ float s_red[4][TFG_R * TF_MAXG];
  const int band = blockIdx.y / ngrp, b0 = (blockIdx.y - band * ngrp) * TFG_R;
  const int nr = B - b0 < TFG_R ? B - b0 : TFG_R;
  const float2* T0 = Tsave + (size_t)band * G * K;
  const float2* g0 = gH + ((size_t)band * B + b0) * ldh;
  float acc[TFG_R][TF_MAXG];
#pragma unroll
  for (int r = 0; r < TFG_R; ++r)
#pragma unroll
    for (int g = 0; g < TF_MAXG; ++g) acc[r][g] = 0.f;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
    float2 gh[TFG_R];
#pragma unroll
    for (int r = 0; r < TFG_R; ++r) gh[r] = r < nr ? g0[(size_t)r * ldh + k] : make_float2(0.f, 0.f);
    const float2 f = filt ? filt[(size_t)band * ldf + k] : make_float2(1.f, 0.f);
    float2 w[TF_MAXG];
#pragma unroll
    for (int g = 0; g < TF_MAXG; ++g) w[g] = g < G ? cmul(f, T0[(size_t)g * K + k]) : make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < TFG_R; ++r)
#pragma unroll
      for (int g = 0; g < TF_MAXG; ++g) acc[r][g] += gh[r].x * w[g].x + gh[r].y * w[g].y;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < TFG_R; ++r)
#pragma unroll
    for (int g = 0; g < TF_MAXG; ++g) {
      const float v = wave_sum(acc[r][g]);
      if (lane == 0) s_red[wv][r * TF_MAXG + g] = v;
    }
  __syncthreads();
  if (threadIdx.x < TFG_R * TF_MAXG) {
    const int r = threadIdx.x / TF_MAXG, g = threadIdx.x % TF_MAXG;
    if (r < nr && g < G) {
      const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
      rg_partial[((size_t)band * B * G + (size_t)(b0 + r) * G + g) * gridDim.x + blockIdx.x] = v;
    }
  }
}

#define TFR_BCH 8
__global__ __launch_bounds__(256) void k_tf_compose_bwd_rec(TfCompose a, const float2* __restrict__ Tsave,
                                                            const float2* __restrict__ gH, int ldh,
                                                            float* __restrict__ gpart) {
  __shared__ TfBlock tab[TF_MAXG];
  const int band = blockIdx.y, G = a.G, B = a.B, nparts = gridDim.x;
  tf_stage(a.coef, a.delays, band * G, G, a.nper, tab);
  __syncthreads();
  // (wave index made wave-uniform for the compiler: the receiver-gain reads become scalar loads)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w >= G) return;
  const float* rgain = a.rgain + (size_t)band * B * G + w;
  gH += (size_t)band * B * ldh;
  const float2* Tp = Tsave + (size_t)(band * G + w) * a.K;
  float Q[16], m[4];
#pragma unroll
  for (int S = 0; S < 16; ++S) Q[S] = tab[w].Q[S];
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = tab[w].m[r];
  float aP[16], aQ[16];
#pragma unroll
  for (int S = 0; S < 16; ++S) aP[S] = aQ[S] = 0.f;
  // (a.scale: Tsave holds the UNSCALED functions -- the step's normalisation scale joined the group signals behind the
  // transform -- and T' = scale T is formed here)
  const float tsc = a.scale ? a.scale[band * G + w] : 1.0f;
  const float itsc = 1.0f / tsc;
  const int ntiles = (a.K + TFB_T - 1) / TFB_T;
  for (int tile = blockIdx.x; tile < ntiles; tile += nparts) {
    const int k = tile * TFB_T + lane;
    const bool live = k < a.K;
    const int kk = live ? k : a.K - 1;
    float2 acc = make_float2(0.f, 0.f);
    for (int b0 = 0; b0 < B; b0 += TFR_BCH) {
      float2 gh[TFR_BCH];
#pragma unroll
      for (int i = 0; i < TFR_BCH; ++i)
        gh[i] = (live && b0 + i < B) ? gH[(size_t)(b0 + i) * ldh + kk] : make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < TFR_BCH; ++i) {
        const float rg = b0 + i < B ? rgain[(b0 + i) * G] : 0.f;
        acc.x += rg * gh[i].x;
        acc.y += rg * gh[i].y;
      }
    }
    if (a.filt) acc = cmulc(acc, a.filt[(size_t)band * a.ldf + kk]);      // dL/dT' (zero beyond K)
    float2 e[16];
    tf_phasors(a.turns, a.logr, kk, m, e);
    float2 den = make_float2(Q[0], 0.f);
#pragma unroll
    for (int S = 1; S < 16; ++S) {
      den.x += Q[S] * e[S].x;
      den.y += Q[S] * e[S].y;
    }
    float2 u = cmulc(acc, cinv(den));                   // dL/dT' conj(1 / Den)
    float2 v;
    if (a.fold) {                                       // (acc = scale dL/dT': v = u' conj(scale T) = u conj(T), u' = u / scale)
      v = cmulc(u, Tp[kk]);
      u = cscale(u, itsc);
    } else {
      v = cmulc(u, a.scale ? cscale(Tp[kk], tsc) : Tp[kk]);
    }
    aP[0] += u.x;
    aQ[0] -= v.x;
#pragma unroll
    for (int S = 1; S < 16; ++S) {
      if (S < 15) aP[S] += u.x * e[S].x + u.y * e[S].y;
      aQ[S] -= v.x * e[S].x + v.y * e[S].y;
    }
  }
  float* out = gpart + (size_t)(band * G + w) * TF_REC * nparts + blockIdx.x;
#pragma unroll
  for (int S = 0; S < 16; ++S) {
    const float p = wave_sum_full(aP[S]), q = wave_sum_full(aQ[S]);       // (whole waves reach this point: VALU sums)
    if (lane == 0) {
      out[(size_t)S * nparts] = p;
      out[(size_t)(16 + S) * nparts] = q;
    }
  }
}

static int tf_compose_parts_host(int K) {
  const int tiles = (K + TFB_T - 1) / TFB_T;
  int parts = (tiles + 3) / 4;                   // ~4 tiles per workgroup
  if (parts > TF_MAX_PARTS) parts = TF_MAX_PARTS;
  return parts < 1 ? 1 : parts;
}
static int tf_gain_chunks_host(int K) {
  int c = (K + 1023) / 1024;                     // ~4 bins per thread
  if (c > 32) c = 32;
  return c < 1 ? 1 : c;
}
extern "C" int gfdn_tf_compose_parts(int K) { return K > 0 ? tf_compose_parts_host(K) : 0; }
extern "C" size_t gfdn_tf_compose_bwd_work_bytes(int K, int nbands, int G) {
  if (K <= 0 || nbands <= 0 || G <= 0) return 0;
  return (size_t)nbands * G * TF_REC * tf_compose_parts_host(K) * sizeof(float);
}
extern "C" size_t gfdn_tf_gain_grad_work_bytes(int K, int nbands, int G, int B) {
  if (K <= 0 || nbands <= 0 || G <= 0 || B <= 0) return 0;
  return (size_t)nbands * G * B * tf_gain_chunks_host(K) * sizeof(float);
}

extern "C" int gfdn_tf_gain_grad(int K, int nbands, int G, int B, const float* Tsave, const float* filt, int ldf,
                                 const float* gH, int ldh, float* grgain, void* work, void* stream) {
  if (!Tsave || !gH || !work || K <= 0 || nbands <= 0 || G <= 0 || B <= 0 || ldh < K ||
      (filt && nbands > 1 && ldf < K))
    return GFDN_E_BADARG;
  if (G > TF_MAXG || nbands * ((B + TFG_R - 1) / TFG_R) > 65535) return GFDN_E_UNSUPPORTED;
  const int nchunk = tf_gain_chunks_host(K), ngrp = (B + TFG_R - 1) / TFG_R;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tf_gain_grad, dim3(nchunk, nbands * ngrp), dim3(256), 0, s, (const float2*)Tsave, K, G, B,
                     (const float2*)filt, ldf, (const float2*)gH, ldh, (float*)work, ngrp);
  GFDN_LAUNCH_CHECK();
  if (!grgain) return 0;            // the (nbands B G, gfdn_tf_gain_chunks(K)) partial rows stay in ``work`` for a consumer that
                                    // sums them itself (gfdn_mlp_gains_banded_bwd_parts)
  const int nrg = nbands * B * G;
  hipLaunchKernelGGL(k_tf_rows_sum, dim3((nrg + 3) / 4), dim3(256), 0, s, (const float*)work, nchunk, nrg, grgain, nrg,
                     (float*)nullptr, (float*)nullptr);
  GFDN_LAUNCH_CHECK();
  return 0;
}
extern "C" int gfdn_tf_gain_chunks(int K) { return K > 0 ? tf_gain_chunks_host(K) : 0; }

extern "C" int gfdn_tf_compose_bwd(const double* turns, const double* logr, int K, int nbands, int G, int nper,
                                   const float* coef, const float* delays, const float* Tsave, const float* tscale,
                                   const float* rgain, int B, const float* filt, int ldf, const float* gH,
                                   int ldh, float* grec, void* work, int gain_fold, void* stream) {
  int rc = tf_compose_ok(turns, K, nbands, G, nper, B, coef, delays, rgain);
  if (rc) return rc;
  if (!Tsave || !gH || !work || ldh < K || (filt && nbands > 1 && ldf < K) || (gain_fold && !tscale)) return GFDN_E_BADARG;
  TfCompose a{turns, logr, K, G, nper, B, coef, delays, tscale, rgain, (const float2*)filt, ldf, gain_fold ? 1 : 0};
  const int nparts = tf_compose_parts_host(K);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tf_compose_bwd_rec, dim3(nparts, nbands), dim3(256), 0, s, a, (const float2*)Tsave,
                     (const float2*)gH, ldh, (float*)work);
  GFDN_LAUNCH_CHECK();
  if (!grec) return 0;          // the partial rows stay in work: gfdn_tf_param_grads(..., nparts0 = gfdn_tf_compose_parts(K))
  const int n0 = nbands * G * TF_REC;
  hipLaunchKernelGGL(k_tf_rows_sum, dim3((n0 + 3) / 4), dim3(256), 0, s, (const float*)work, nparts, n0, grec, n0,
                     (float*)nullptr, (float*)nullptr);
  GFDN_LAUNCH_CHECK();
  return 0;
}
