// Block transfer functions in polynomial form for blocks of 5..8 delay lines (the N = 32 = 4 x 8 layout of
// BASELINE.json configs[4]), zero coupling, for gfx950 -- with the polynomial evaluation AND the gradient accumulation on
// the matrix cores.
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn): the resolvent of one block, X(z) = D(z) Gamma^-1 - A
// (feedback_loop.py:326-391), enters the model only through the group transfer function T = c^T X^-1 b (model.py:583-619;
// sub-FDN responses with the raw M and no absorption, model.py:209-252).  As for the blocks of <= 4 lines
// (csrc/blocktf.hip) T is a ratio of MULTILINEAR polynomials in the n phasors zeta_i = z^{m_i} / gamma_i,
//     det X           = sum_S Q_S e_S ,   Q_S = det((-A)[S^c, S^c]) prod_{i in S} 1/gamma_i ,   e_S = prod_{i in S} z^{m_i}
//     c^T adj(X) b    = sum_S P_S e_S ,   P_S = sum_i c_i Y_{i,S} ,  Y_{i,S} = det(((-A) with column i := b)[S^c, S^c]) prod 1/gamma
//                                         = -det([[(-A), b], [c^T, 0]] restricted to S^c and the border) prod 1/gamma ,
// 2 x 256 real coefficients per block (records: the determinant polynomial and the eight Y_i, built in float64).
// With the subsets split as S = (S1 over lines 0..3, S2 over lines 4..7) a polynomial is the bilinear form e1^T C e2 of a
// REAL 16 x 16 coefficient matrix with the two complex subset-phasor vectors of a bin, and C e2 for 64 bins at a time is a
// GEMM: v_mfma_f32_16x16x4_f32 (exact float32 products, float32 accumulate), A = C (lane l: C[S1 = l & 15][S2 = 4 ks + (l >> 4)]
// = coef[64 ks + l]: one coalesced load), B = Re / Im of e2 of sixteen bins.  The backward is the transpose of the same
// product: with dL = Re(conj(g) dT), T = P / Q,
//     dL/dP_S = sum_k Re(u_k e_S(k)) ,  u = conj(g) / Q ;     dL/dQ_S = -sum_k Re(u_k T_k e_S(k)) ,
// i.e. the (16 x 2 bins) x (2 bins x 16) products (u e1) e2^T summed over the bins -- two 16 x 16 MFMA accumulator tiles
// per wavefront hold all 512 gradient records.  The records -> (dL/dA, dL/db, dL/dc) map is bin independent: cofactors of
// the 256 masked 8 x 8 and bordered 9 x 9 matrices of a block, in float64 (k_tf8_rec_grads).  The thread-per-system 8 x 8
// complex eliminations of csrc/solve.hip (k_solve8_fwd / _bwd, k_subfdn8_energy: ~2500 VALU instructions per system at one
// wave per SIMD) become ~128 MFMAs + ~1400 VALU instructions per 64 bins, forward and backward together.
//
// Scaling convention (trainer.py:317-332 normalize): the records are built from the gains BEFORE the rescale (b_old,
// c_old); afterwards the gain buffers hold b' = b_old sqrt(s), c' = c_old sqrt(s), s = scale = E^(-1/2), and
// T' = T(b', c') = P(b', c') / Q with P(b', c')_S = sqrt(s) sum_i c'_i Y_{i,S}(b_old): the passes take the CURRENT c and s and
// work on the records of the current gains; the gradient map reads the current gains as well.
#include "common.h"
#include "ortho_dev.h"

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

#define T8_NPOLY 9
#define T8_REC (T8_NPOLY * 256)        // floats per block: [poly][S], S = S1 + 16 S2
#define T8_ACC 80                      // 64 dL/dA (row-major 8 x 8) | 8 dL/db | 8 dL/dc (output of the gradient map)
#define T8_GREC 512                    // gradient records of a block: dL/dP_S (256) | dL/dQ_S (256), S = S1 + 16 S2
#define T8_MAXBLK 64
#define T8_MAX_PARTS 256
#define T8_WAVES 4                     // wavefronts per workgroup (measured: 2 per workgroup, four workgroups per CU, is 10 % slower)
#define T8_P 66                        // row pitch (float2) of the waves' [S][bin] images: rows 16 B apart in the banks, so
                                       // that the 16 rows x 2 bins a gradient MFMA step reads hit 64 distinct banks
#define T8_IMG (2 * 16 * T8_P + 64)     // float2 per wavefront: the two [S][bin] images and one row of 64 per-bin values
#define T8_LDS_BYTES ((size_t)T8_WAVES * T8_IMG * sizeof(float2))

// ------------------------------------------------------------------------------------------
// coefficient records (float64): determinants of masked 8 x 8 matrices, partial pivoting by row selects
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double t8_det(double (&m)[8][8]) {
  double det = 1.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double best = fabs(m[j][j]);
    int bi = j;
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      const double v = fabs(m[i][j]);
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
      const bool sw = bi == i;
#pragma unroll
      for (int c = j; c < 8; ++c) {
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
    for (int i = j + 1; i < 8; ++i) {
      const double f = m[i][j] * inv;
#pragma unroll
      for (int c = j + 1; c < 8; ++c) m[i][c] -= f * m[j][c];
    }
  }
  return det;
}

// grid (nblk, sets, 9): z = 0 the determinant polynomial, z = 1 + i the numerator polynomial of y_i = (X^-1 b)_i; thread S = subset
__global__ __launch_bounds__(256) void k_tf8_coefs(const float* __restrict__ A0, const float* __restrict__ ig0,
                                                   float* __restrict__ coef0, const float* __restrict__ A1,
                                                   const float* __restrict__ ig1, float* __restrict__ coef1,
                                                   const float* __restrict__ b, const float* __restrict__ c, int n) {
  __shared__ double sA[64], sb[8], sig[8];
  const int blk = blockIdx.x, set = blockIdx.y, task = blockIdx.z, S = threadIdx.x;
  const float* A = (set ? A1 : A0) + (size_t)blk * n * n;
  const float* ig = set ? ig1 : ig0;
  float* coef = (set ? coef1 : coef0) + (size_t)blk * T8_REC;
  if (S < 64) {
    const int i = S >> 3, j = S & 7;
    sA[S] = (i < n && j < n) ? -(double)A[i * n + j] : 0.0;
  }
  if (S < 8) {
    sb[S] = S < n ? (double)b[blk * n + S] : 0.0;
    sig[S] = (S < n && ig) ? (double)ig[blk * n + S] : 1.0;
  }
  __syncthreads();
  const bool absent = (S >> n) != 0;              // the subset names a line the block does not have: coefficient 0
  double igp = 1.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if ((S >> i) & 1) igp *= sig[i];
  // (-A) with column `col` := b (-1: untouched), restricted to the complement of S: lines of S -- and
  // the lines beyond n -- become unit rows / columns, which leaves the determinant of the restriction
  auto masked_det = [&](int col) {
    double m[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const bool in = !((S >> r) & 1) && !((S >> cc) & 1) && r < n && cc < n;
        double v = sA[r * 8 + cc];
        if (cc == col) v = sb[r];
        m[r][cc] = in ? v : (r == cc ? 1.0 : 0.0);
      }
    return t8_det(m);
  };
  if (task == 0) {
    coef[S] = absent ? 0.f : (float)(masked_det(-1) * igp);
  } else {
    const int i = task - 1;
    const bool none = absent || i >= n || ((S >> i) & 1);
    coef[(1 + i) * 256 + S] = none ? 0.f : (float)(masked_det(i) * igp);
  }
}

extern "C" int gfdn_tf8_coefs(const float* A0, const float* inv_gamma0, float* coef0, const float* A1,
                              const float* inv_gamma1, float* coef1, const float* b, const float* c, int nblk, int nper,
                              void* stream) {
  if (!A0 || !coef0 || !b || !c || nblk <= 0 || nper <= 0 || (A1 && !coef1)) return GFDN_E_BADARG;
  if (nper > 8 || nblk > T8_MAXBLK) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_tf8_coefs, dim3(nblk, A1 ? 2 : 1, 9), dim3(256), 0, (hipStream_t)stream, A0, inv_gamma0, coef0, A1,
                     inv_gamma1, coef1, b, c, nper);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// passes over the bins: one workgroup = four wavefronts on ONE block, a wavefront = tiles of 64 bins (lane = bin)
// ------------------------------------------------------------------------------------------
struct T8Args {
  const double* turns;
  double dturn;              // != 0: the grid is uniform on the unit circle with this step (turns[k] = turns[0] + k dturn):
                             // a wavefront then steps its phasors from tile to tile by constant rotations
  int K, nblk, nper;
  const float* coef;         // (nblk, 9, 256)
  const float* delays;       // (nblk * nper)
  const float* c;            // (nblk * nper) output gains as they are NOW (see the scaling convention)
  const float* scale;        // (nblk) s = E^(-1/2), or NULL (the energy pass: gains not yet rescaled, factor 1)
  // T8_TSAVE
  float2* Tsave;             // (nblk, K)
  float2* Tquad;             // (nblk / G, K, 4) or NULL
  float2* Hout;              // (nblk, K) or NULL: T' filt (band row nblk / G of filt) -- the group responses through the band's filter
  float2* Dsave;             // (nblk, K) or NULL: 1 / Q of the bin, for the adjoint pass of the same grid (Tin / Din below)
  const int* hslot;          // NULL, or (K): column of bin k in Hout (and in filt), bit 31: the column holds the conjugate --
                             // the grid in bin order, the group responses in the slot order of the odd-length transform
  int G;
  // T8_COLORLESS
  int asym;
  float gscale;
  float* lossp;              // (nblk, nparts) partial losses
  // T8_BWD
  const float* rgain;        // (nblk / G * B, G)
  const float2* gH;          // (nblk / G * B, ldh)
  int ldh;
  const float2* filt;        // (nblk / G, ldf) or NULL
  int ldf, B;
  float* part;               // T8_ENERGY: [blk * nparts + p]; heavy modes: [(blk * T8_GREC + e) * nparts + p]
  // T8_BWD: the forward pass's saved T' and 1 / Q of the same grid (both or neither): the two polynomials are not evaluated again
  const float2* Tin;
  const float2* Din;
};

enum { T8_ENERGY = 0, T8_TSAVE = 1, T8_COLORLESS = 2, T8_BWD = 3 };

extern __shared__ float2 t8_lds[];

__device__ __forceinline__ float2 t8_zpow(const double* __restrict__ turns, int k, float m) {
  double t = (double)m * turns[k];
  t -= rint(t);
  float s, c;
  sincospif(2.0f * (float)t, &s, &c);
  return make_float2(c, s);
}

// subset products of four phasors, written to the wave's [S][lane] image
__device__ __forceinline__ void t8_stage_subsets(float2 p0, float2 p1, float2 p2, float2 p3, float2* img, int lane) {
  float2 e[16];
  e[0] = make_float2(1.f, 0.f);
  e[1] = p0; e[2] = p1; e[4] = p2; e[8] = p3;
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
#pragma unroll
  for (int S = 0; S < 16; ++S) img[S * T8_P + lane] = e[S];
}

// NP polynomials of the wave's 64 bins: val[p] (lane = bin) = e1^T C_p e2, the A operands A[p][ks] already in registers
template <int NP>
__device__ __forceinline__ void t8_polys(const float (&A)[NP][4], const float2* E1, const float2* E2, int lane,
                                         float2 (&val)[NP]) {
  const int q = lane >> 4, cidx = lane & 15;
  // (unrolled: the scheduler interleaves the four column groups' MFMA chains with the previous groups' dot products --
  // 4 x 2 x 4 NP accumulator registers alive at once, 64 for the two polynomials of a pass)
#pragma unroll
  for (int cg = 0; cg < 4; ++cg) {
    f32x4 are[NP], aim[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      are[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
      aim[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float2 b2 = E2[(4 * ks + q) * T8_P + 16 * cg + cidx];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#ifdef T8_DIAG_NO_MFMA
        are[p][0] += A[p][ks] * b2.x;
        aim[p][0] += A[p][ks] * b2.y;
#else
        are[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p][ks], b2.x, are[p], 0, 0, 0);
        aim[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[p][ks], b2.y, aim[p], 0, 0, 0);
#endif
      }
    }
    float2 e1v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) e1v[r] = E1[(4 * q + r) * T8_P + 16 * cg + cidx];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float2 s = make_float2(0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // D[S1 = 4 q + r][bin] = (C e2)[S1]: times e1[S1], summed over the S1 of this lane ...
        s.x += are[p][r] * e1v[r].x - aim[p][r] * e1v[r].y;
        s.y += are[p][r] * e1v[r].y + aim[p][r] * e1v[r].x;
      }
      // ... and over the four lanes that hold the other S1 of the same bin
#ifndef T8_DIAG_NO_SHFL
      s.x = xor32_sum(xor16_sum(s.x));          // (v_permlane16/32_swap: VALU, no LDS crossbar)
      s.y = xor32_sum(xor16_sum(s.y));
#endif
      if (q == cg) val[p] = s;          // bin 16 cg + cidx = this lane
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(64 * T8_WAVES, 8 / T8_WAVES) void k_tf8_pass(T8Args a) {
  __shared__ float s_red[T8_WAVES][T8_GREC + 1];
  // (the adjoint pass and the forward pass of the group responses are links of the step's critical chain; the colorless pass
  // -- the same VALU-bound template on twice the bins -- runs beside them on the side stream: they take issue priority)
  if (MODE == T8_BWD || MODE == T8_TSAVE) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int blk = blockIdx.y, n = a.nper, K = a.K;
  float2* E1 = t8_lds + wv * T8_IMG;
  float2* E2 = E1 + 16 * T8_P;
  float2* TT = E2 + 16 * T8_P;
  const float* coef = a.coef + (size_t)blk * T8_REC;
  float m[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) m[r] = r < n ? a.delays[blk * n + r] : 0.f;
  const float sc = a.scale ? a.scale[blk] : 1.0f;
  const float tmul = a.scale ? sqrtf(sc) : 1.0f;
  constexpr bool HEAVY = MODE == T8_COLORLESS || MODE == T8_BWD;
  // the two coefficient matrices of the CURRENT gains as MFMA A operands, in registers for the whole launch:
  // AL[0] = Q (determinant), AL[1] = P = tmul sum_i c_i Y_i (scaling convention above)
  float AL[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    AL[0][ks] = coef[64 * ks + lane];
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p += (i < n ? a.c[blk * n + i] : 0.f) * coef[(1 + i) * 256 + 64 * ks + lane];
    AL[1][ks] = tmul * p;
  }
  float acc0 = 0.f;                                      // energy / loss partial
  f32x4 accP = (f32x4){0.f, 0.f, 0.f, 0.f};              // heavy modes: dL/dP[S1][S2] and dL/dQ[S1][S2], lane l register r
  f32x4 accQ = (f32x4){0.f, 0.f, 0.f, 0.f};              // = entry [S1 = 4 (l >> 4) + r][S2 = l & 15]
  const int ntiles = (K + 63) >> 6;
  const float invK = 1.0f / (float)K;
  float2 ph[8], rot[8];
  int iter = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    ph[r] = make_float2(1.f, 0.f);
    double tr = (double)m[r] * a.dturn * 64.0 * (double)(gridDim.x * T8_WAVES);
    tr -= rint(tr);
    float sn, cs;
    sincospif(2.0f * (float)tr, &sn, &cs);
    rot[r] = make_float2(cs, sn);
  }
  const int band = a.G > 0 ? blk / a.G : 0, g = a.G > 0 ? blk - band * a.G : 0;
#pragma unroll 1
  for (int tile = blockIdx.x * T8_WAVES + wv; tile < ntiles; tile += gridDim.x * T8_WAVES) {
    const int k = tile * 64 + lane;
    const bool live = k < K;
    const int kk = live ? k : K - 1;
    // the output-stage adjoint folds the receivers' dL/dH first: the loads fly over the phasors
    float2 gs = make_float2(0.f, 0.f);
    if (MODE == T8_BWD && live) {
      const float2* gh = a.gH + (size_t)band * a.B * a.ldh + kk;
      const float* rg = a.rgain + (size_t)band * a.B * a.G + g;
      for (int bb = 0; bb < a.B; ++bb) {
        const float2 v = gh[(size_t)bb * a.ldh];
        const float r = rg[bb * a.G];
        gs.x += r * v.x;
        gs.y += r * v.y;
      }
      if (a.filt) gs = cmulc(gs, a.filt[(size_t)band * a.ldf + kk]);      // dL/dT' = conj(filt) sum_b rgain dL/dH
    }
    asm volatile("" ::: "memory");
    // phasors z_k^{m_i}: exact (float64 range reduction + sincospif) -- or, on a uniform grid, exact every eighth tile of
    // the wavefront and stepped by the constant rotation z_1^{m_i (tile stride)} in between (eight complex products per
    // bin instead of eight range reductions; the rounding of seven steps, ~1e-7 each, stays far below the 1e-5 the
    // evaluation itself carries -- as the TF_RUN runs of csrc/blocktf.hip)
#ifdef T8_DIAG_NO_PHASE
    if (iter == 0) {
#else
    if (a.dturn == 0.0 || (iter & 7) == 0 || !live) {
#endif
#pragma unroll
      for (int r = 0; r < 8; ++r) ph[r] = t8_zpow(a.turns, kk, m[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) ph[r] = cmul(ph[r], rot[r]);
    }
    ++iter;
    t8_stage_subsets(ph[0], ph[1], ph[2], ph[3], E1, lane);
    t8_stage_subsets(ph[4], ph[5], ph[6], ph[7], E2, lane);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave's own LDS image: no block barrier)
    float2 dinv, t;
    if (MODE == T8_BWD && a.Tin) {
      // (the forward pass of this grid left them: 64 MFMAs and the dot products with e1 less per 64 bins)
      t = a.Tin[(size_t)blk * K + kk];
      dinv = a.Din[(size_t)blk * K + kk];
    } else {
      float2 val[2];
      t8_polys<2>(AL, E1, E2, lane, val);
      dinv = cinv(val[0]);
      t = cmul(val[1], dinv);                                // T' = P / Q at the current gains
    }
    if (MODE == T8_ENERGY) {
      if (live) acc0 += t.x * t.x + t.y * t.y;
    } else if (MODE == T8_TSAVE) {
      if (live) {
        a.Tsave[(size_t)blk * K + k] = t;
        if (a.Hout) {
          const int so = a.hslot ? a.hslot[k] : k, col = so & 0x7fffffff;
          const float2 ts = so < 0 ? cconj(t) : t;               // (the response at conj(z) is the conjugate)
          a.Hout[(size_t)blk * K + col] = a.filt ? cmul(ts, a.filt[(size_t)band * a.ldf + col]) : ts;
        }
        if (a.Dsave) a.Dsave[(size_t)blk * K + k] = dinv;
        if (a.Tquad) {
          float2* qd = a.Tquad + ((size_t)band * K + k) * 4;
          qd[g] = t;
          if (g == a.G - 1)
            for (int z = a.G; z < 4; ++z) qd[z] = make_float2(0.f, 0.f);
        }
      }
    } else {
      if (MODE == T8_COLORLESS) {
        // colorless_fdn/losses.py:20-73 on the scaled sub-FDN response, as k_tf_colorless
        const float mag = sqrtf(t.x * t.x + t.y * t.y);
        const float d = mag - 1.0f, d2 = d * d;
        const bool four = a.asym && (d > 1.0f);
        const float dl = four ? 4.0f * d2 * d : 2.0f * d;
        const float f = (mag > 0.f && live) ? a.gscale * invK * dl / mag : 0.f;
        gs = make_float2(f * t.x, f * t.y);
        if (live) acc0 += (four ? d2 * d2 : d2) * invK;
      }
      // dL = Re(conj(gs) dT'), dT' = (dP - T' dQ) / Q:  dL/dP[S1][S2] += Re(u e1[S1] e2[S2]), u = conj(gs) / Q, and
      // dL/dQ[S1][S2] += Re(-u T' e1[S1] e2[S2]) -- (16 x 2 bins) x (2 bins x 16) products with k = (bin, Re / Im): the
      // rows u e1[S1] replace e1 in the wave's LDS image (every lane rewrites its own column), the columns are e2
      const float2 u = make_float2(gs.x * dinv.x + gs.y * dinv.y, gs.x * dinv.y - gs.y * dinv.x);    // conj(gs) / Q
      const int rc = lane & 15, kq = lane >> 4, part = kq & 1;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int S1 = 0; S1 < 16; ++S1) E1[S1 * T8_P + lane] = cmul(E1[S1 * T8_P + lane], u);           // rows u e1[S1]
      TT[lane] = live ? make_float2(-t.x, -t.y) : make_float2(0.f, 0.f);                               // -T' of the bin
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // one loop for both record sets, two accumulators each (even / odd steps): four independent MFMA chains; the
      // rows of dL/dQ are the rows of dL/dP times -T' of the bin
      f32x4 p0 = accP, p1 = (f32x4){0.f, 0.f, 0.f, 0.f}, q0 = accQ, q1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int t2 = 0; t2 < 32; t2 += 2) {
        const int bin0 = 2 * t2 + (kq >> 1), bin1 = bin0 + 2;
        const float2 a0 = E1[rc * T8_P + bin0], b0 = E2[rc * T8_P + bin0], m0 = TT[bin0];
        const float2 a1 = E1[rc * T8_P + bin1], b1 = E2[rc * T8_P + bin1], m1 = TT[bin1];
        const float2 c0 = cmul(a0, m0), c1 = cmul(a1, m1);
        const float bb0 = part ? b0.y : b0.x, bb1 = part ? b1.y : b1.x;
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(part ? -a0.y : a0.x, bb0, p0, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(part ? -c0.y : c0.x, bb0, q0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(part ? -a1.y : a1.x, bb1, p1, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(part ? -c1.y : c1.x, bb1, q1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accP[r] = p0[r] + p1[r];
        accQ[r] = q0[r] + q1[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  // fixed-order sums over the four waves
  if (MODE == T8_TSAVE) return;
  const int nparts = gridDim.x;
  {
    const float v = wave_sum(acc0);
    if (lane == 0) s_red[wv][T8_GREC] = v;
  }
  if (HEAVY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int S1 = 4 * (lane >> 4) + r, S2 = lane & 15;
      s_red[wv][S1 + 16 * S2] = accP[r];
      s_red[wv][256 + S1 + 16 * S2] = accQ[r];
    }
  }
  __syncthreads();
  if (!HEAVY) {
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < T8_WAVES; ++w2) s += s_red[w2][T8_GREC];
      a.part[(size_t)blk * nparts + blockIdx.x] = s;
    }
    return;
  }
  for (int e = tid; e <= T8_GREC; e += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < T8_WAVES; ++w2) s += s_red[w2][e];
    if (e < T8_GREC) a.part[((size_t)blk * T8_GREC + e) * nparts + blockIdx.x] = s;
    else if (a.lossp) a.lossp[(size_t)blk * nparts + blockIdx.x] = s;
  }
}

static int t8_parts_host(int K) {
  const int tiles = (K + 63) / 64, wg = (tiles + T8_WAVES - 1) / T8_WAVES;
  int parts = (wg + 3) / 4;                    // ~4 tiles per wavefront
  if (parts < 1) parts = 1;
  if (parts > T8_MAX_PARTS) parts = T8_MAX_PARTS;
  return parts;
}

extern "C" int gfdn_tf8_parts(int K) { return K > 0 ? t8_parts_host(K) : 0; }

static int t8_ok(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays, const float* c) {
  if (!turns || !coef || !delays || !c || K <= 0 || nblk <= 0 || nper <= 0) return GFDN_E_BADARG;
  if (nper > 8 || nblk > T8_MAXBLK) return GFDN_E_UNSUPPORTED;
  return 0;
}

template <int MODE>
static int t8_launch(const T8Args& a, hipStream_t s) {
  int rc = ensure_dyn_lds(k_tf8_pass<MODE>, T8_LDS_BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tf8_pass<MODE>, dim3(t8_parts_host(a.K), a.nblk), dim3(64 * T8_WAVES), T8_LDS_BYTES, s, a);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// energy partials of the (unscaled) responses sum_i c_i y_i: part[blk * gfdn_tf8_parts(K) + p]; finish with
// gfdn_tf_energy(..., phase = 2) semantics through gfdn_tf8_energy_finish below
extern "C" int gfdn_tf8_energy(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays,
                               float* b, float* c, float* energy, float* scale, void* work, double dturn, void* stream);

// finish of the energy pass: E, scale = E^(-1/2), in-place rescale of b, c (trainer.py:317-332)
__global__ __launch_bounds__(256) void k_tf8_energy_finish(const float* __restrict__ partial, int nparts, int K, int nper,
                                                           float* __restrict__ b, float* __restrict__ c,
                                                           float* __restrict__ energy, float* __restrict__ scale) {
  __shared__ float s_r[16];
  const int g = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += partial[(size_t)g * nparts + p];
  s = block_sum(s, s_r);
  const float E = s / (float)K;
  if (threadIdx.x == 0) {
    if (energy) energy[g] = E;
    if (scale) scale[g] = 1.0f / sqrtf(E);
  }
  if (b && c) {
    const float d = powf(E, 0.25f);
    for (int i = threadIdx.x; i < nper; i += 256) {
      b[g * nper + i] /= d;
      c[g * nper + i] /= d;
    }
  }
}

extern "C" int gfdn_tf8_energy(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays,
                               float* b, float* c, float* energy, float* scale, void* work, double dturn, void* stream) {
  int rc = t8_ok(turns, K, nblk, nper, coef, delays, c);
  if (rc) return rc;
  if (!work || !b) return GFDN_E_BADARG;
  T8Args a{};
  a.turns = turns; a.dturn = dturn; a.K = K; a.nblk = nblk; a.nper = nper; a.coef = coef; a.delays = delays; a.c = c; a.scale = nullptr;
  a.part = (float*)work;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = t8_launch<T8_ENERGY>(a, s))) return rc;
  hipLaunchKernelGGL(k_tf8_energy_finish, dim3(nblk), dim3(256), 0, s, (const float*)work, t8_parts_host(K), K, nper, b, c,
                     energy, scale);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_tf8_tsave(const double* turns, int K, int nbands, int G, int nper, const float* coef,
                              const float* delays, const float* c, const float* scale, float* Tsave, float* Tquad,
                              const float* filt_c64, int ldf, float* Hout_c64, float* Dinv_c64, const int* hslot,
                              void* stream) {
  int rc = t8_ok(turns, K, nbands * G, nper, coef, delays, c);
  if (rc) return rc;
  if (!Tsave || G <= 0 || G > 4 || (filt_c64 && ldf < K)) return GFDN_E_BADARG;
  T8Args a{};
  a.turns = turns; a.K = K; a.nblk = nbands * G; a.nper = nper; a.coef = coef; a.delays = delays; a.c = c; a.scale = scale;
  a.Tsave = (float2*)Tsave; a.Tquad = (float2*)Tquad; a.G = G;
  a.Hout = (float2*)Hout_c64; a.filt = (const float2*)filt_c64; a.ldf = ldf; a.Dsave = (float2*)Dinv_c64; a.hslot = hslot;
  return t8_launch<T8_TSAVE>(a, (hipStream_t)stream);
}

// work: gfdn_tf8_part_bytes(nblk, K) for part, nblk * gfdn_tf8_parts(K) floats for lossp
extern "C" size_t gfdn_tf8_part_bytes(int nblk, int K) {
  return (size_t)(nblk > 0 ? nblk : 1) * T8_GREC * t8_parts_host(K > 0 ? K : 1) * sizeof(float);
}

extern "C" int gfdn_tf8_colorless(const double* turns, int K, int nblk, int nper, const float* coef, const float* delays,
                                  const float* c, const float* scale, int asym, float gscale, float* part, float* lossp,
                                  float* loss, double dturn, void* stream) {
  int rc = t8_ok(turns, K, nblk, nper, coef, delays, c);
  if (rc) return rc;
  if (!part || !lossp || !loss) return GFDN_E_BADARG;
  T8Args a{};
  a.turns = turns; a.dturn = dturn; a.K = K; a.nblk = nblk; a.nper = nper; a.coef = coef; a.delays = delays; a.c = c; a.scale = scale;
  a.asym = asym; a.gscale = gscale; a.part = part; a.lossp = lossp;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = t8_launch<T8_COLORLESS>(a, s))) return rc;
  return gfdn_tf_rows_sum(lossp, t8_parts_host(K), nblk, loss, stream);
}

extern "C" int gfdn_tf8_compose_bwd(const double* turns, int K, int nbands, int G, int nper, const float* coef,
                                    const float* delays, const float* c, const float* scale, const float* rgain, int B,
                                    const float* filt_c64, int ldf, const float* gH_c64, int ldh,
                                    const float* Tsave_c64, const float* Dinv_c64, float* part, void* stream) {
  int rc = t8_ok(turns, K, nbands * G, nper, coef, delays, c);
  if (rc) return rc;
  if (!rgain || !gH_c64 || !part || G <= 0 || G > 4 || B <= 0 || ldh < K || (filt_c64 && ldf < K) ||
      (!Tsave_c64) != (!Dinv_c64))
    return GFDN_E_BADARG;
  T8Args a{};
  a.turns = turns; a.K = K; a.nblk = nbands * G; a.nper = nper; a.coef = coef; a.delays = delays; a.c = c; a.scale = scale;
  a.G = G; a.rgain = rgain; a.gH = (const float2*)gH_c64; a.ldh = ldh; a.filt = (const float2*)filt_c64; a.ldf = ldf;
  a.B = B; a.part = part; a.Tin = (const float2*)Tsave_c64; a.Din = (const float2*)Dinv_c64;
  return t8_launch<T8_BWD>(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Gradient records -> (dL/dA, dL/db, dL/dc), float64, one workgroup per (block, set), thread S = subset.
//   Q_S = igp_S det(M_S) ,             M_S = (-A) restricted to S^c (unit rows / columns on S and beyond n)
//   P_S = -igp_S det(B_S) ,            B_S = [[M_S, b on S^c], [c^T on S^c, 0]]   (9 x 9, b, c the CURRENT gains)
//   dL/dA_ij = sum_S igp_S (gP_S Cof(B_S)_ij - gQ_S Cof(M_S)_ij)        (i, j in S^c;  d(-A) = -dA)
//   dL/db_i  = -sum_S igp_S gP_S Cof(B_S)_{i,8} ,   dL/dc_j = -sum_S igp_S gP_S Cof(B_S)_{8,j}
// Cofactor matrices as det * inverse^T from an in-register Gauss-Jordan inversion with partial pivoting (row swaps by
// selects: compile-time indices only); a vanishing pivot is replaced by 1e-150 (the product det * inverse keeps the
// finite limit).  out[(blk * 80 + e)]: e < 64 dL/dA, 64 + i dL/db_i, 72 + j dL/dc_j -- the layout k_tf8_param_grads sums.
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ double t8_inverse(double (&m)[N][N]) {
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

__device__ __forceinline__ double t8_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_tf8_rec_grads(const float* __restrict__ A0, const float* __restrict__ ig0,
                                                       const float* __restrict__ part0, int np0,
                                                       const float* __restrict__ A1, const float* __restrict__ ig1,
                                                       const float* __restrict__ part1, int np1,
                                                       const float* __restrict__ b, const float* __restrict__ c, int n,
                                                       float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ double sA[64], sb[8], sc[8], sig[8], sred[4][T8_ACC];
  // blockIdx.z = 0: the masked 8 x 8 matrices (dL/dQ_S -> dL/dA), 1: the bordered 9 x 9 ones (dL/dP_S -> dL/dA, dL/db,
  // dL/dc); the two halves land in out[.. + 0] and out[.. + nblk * 2 * 80] and are added by k_tf8_param_grads
  const int blk = blockIdx.x, set = blockIdx.y, half = blockIdx.z, S = threadIdx.x, lane = S & 63, wv = S >> 6;
  const float* A = (set ? A1 : A0) + (size_t)blk * n * n;
  const float* ig = set ? ig1 : ig0;
  const float* part = set ? part1 : part0;
  const int np = set ? np1 : np0;
  float* out = (set ? out1 : out0) + (size_t)blk * T8_ACC + (size_t)half * gridDim.x * 2 * T8_ACC;
  if (S < 64) {
    const int i = S >> 3, j = S & 7;
    sA[S] = (i < n && j < n) ? -(double)A[i * n + j] : 0.0;
  }
  if (S < 8) {
    sb[S] = S < n ? (double)b[blk * n + S] : 0.0;
    sc[S] = S < n ? (double)c[blk * n + S] : 0.0;
    sig[S] = (S < n && ig) ? (double)ig[blk * n + S] : 1.0;
  }
  // this subset's two gradient records, summed over the passes' partial rows in a fixed order
  double gP = 0.0, gQ = 0.0;
  {
    const float* rp = part + ((size_t)blk * T8_GREC + S) * np;
    const float* rq = part + ((size_t)blk * T8_GREC + 256 + S) * np;
    float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
    int i = 0;
    for (; i + 1 < np; i += 2) { p0 += rp[i]; p1 += rp[i + 1]; q0 += rq[i]; q1 += rq[i + 1]; }
    if (i < np) { p0 += rp[i]; q0 += rq[i]; }
    gP = (double)(p0 + p1);
    gQ = (double)(q0 + q1);
  }
  __syncthreads();
  const bool absent = (S >> n) != 0, full = S == (1 << n) - 1;      // (the full set's coefficients are constants)
  double igp = 1.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if ((S >> i) & 1) igp *= sig[i];
  const double wq = (absent || full) ? 0.0 : -gQ * igp, wp = (absent || full) ? 0.0 : gP * igp;
  if (S < T8_ACC) { sred[0][S] = 0.0; sred[1][S] = 0.0; sred[2][S] = 0.0; sred[3][S] = 0.0; }
  __syncthreads();
  if (half == 0) {
    double m[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const bool in = !((S >> r) & 1) && !((S >> cc) & 1) && r < n && cc < n;
        m[r][cc] = in ? sA[r * 8 + cc] : (r == cc ? 1.0 : 0.0);
      }
    const double wd = wq * t8_inverse<8>(m);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = !((S >> i) & 1) && !((S >> j) & 1) && i < n && j < n;
        const double v = t8_wave_sum_d(in ? wd * m[j][i] : 0.0);        // Cof = det inverse^T
        if (lane == 0) sred[wv][i * 8 + j] = v;
      }
  } else {
    double m[9][9];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int cc = 0; cc < 9; ++cc) {
        const bool rin = r == 8 || (!((S >> r) & 1) && r < n), cin = cc == 8 || (!((S >> cc) & 1) && cc < n);
        double v;
        if (r < 8 && cc < 8) v = (rin && cin) ? sA[r * 8 + cc] : (r == cc ? 1.0 : 0.0);
        else if (r < 8) v = rin ? sb[r] : 0.0;                       // border column: b on S^c
        else if (cc < 8) v = cin ? sc[cc] : 0.0;                     // border row: c on S^c
        else v = 0.0;
        m[r][cc] = v;
      }
    const double wd = wp * t8_inverse<9>(m);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool iin = !((S >> i) & 1) && i < n;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = iin && !((S >> j) & 1) && j < n;
        const double v = t8_wave_sum_d(in ? wd * m[j][i] : 0.0);
        if (lane == 0) sred[wv][i * 8 + j] = v;
      }
      const double vb = t8_wave_sum_d(iin ? -wd * m[8][i] : 0.0);      // Cof(B)_{i,8} = det inverse[8][i]
      const double vc = t8_wave_sum_d(iin ? -wd * m[i][8] : 0.0);      // Cof(B)_{8,i} = det inverse[i][8]
      if (lane == 0) { sred[wv][64 + i] = vb; sred[wv][72 + i] = vc; }
    }
  }
  __syncthreads();
  if (S < T8_ACC) out[S] = (float)((sred[0][S] + sred[1][S]) + (sred[2][S] + sred[3][S]));
}

// ------------------------------------------------------------------------------------------
// Tail: partial rows -> dL/dQQ (set 0), dL/dM_raw (set 1), dL/db, dL/dc, then the adjoint of Q = expm(skew(M)), QQ = Q Q
// (k_ortho_bwd with dL/dM_raw added) -- one workgroup per block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tf8_param_grads(const float* __restrict__ part0, int np0,
                                                         const float* __restrict__ part1, int np1,
                                                         const float* __restrict__ scale, int n,
                                                         const float* __restrict__ M, const float* __restrict__ gQ,
                                                         const float* __restrict__ Q, float* __restrict__ gb,
                                                         float* __restrict__ gc, float* __restrict__ gM, int half2) {
  // half2 > 0: every row has a second half at + half2 floats (the 9 x 9 contributions of k_tf8_rec_grads), added here
  extern __shared__ double t8p_lds[];
  __shared__ float srec[2][T8_ACC], sG[2][64];
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  for (int r = tid >> 6; r < 2 * T8_ACC; r += nw) {
    const int set = r / T8_ACC, e = r - set * T8_ACC;
    const float* p = set ? part1 : part0;
    const int np = set ? np1 : np0;
    float s = 0.f;
    if (p) {
      const float* row = p + ((size_t)blk * T8_ACC + e) * np;
      for (int i = lane; i < np; i += 64) s += row[i] + (half2 > 0 ? row[i + half2] : 0.f);
    }
    s = wave_sum(s);
    if (lane == 0) srec[set][e] = s;
  }
  __syncthreads();
  const float sc = scale ? scale[blk] : 1.0f, rs = sqrtf(sc);
  if (tid < 2 * n * n) {
    const int set = tid / (n * n), e = tid - set * n * n, i = e / n, j = e - i * n;
    sG[set][e] = sc * srec[set][i * 8 + j];
  } else if (tid >= 128 && tid < 128 + n) {
    const int i = tid - 128;
    gb[blk * n + i] = rs * (srec[0][64 + i] + srec[1][64 + i]);
  } else if (tid >= 192 && tid < 192 + n) {
    const int j = tid - 192;
    gc[blk * n + j] = rs * (srec[0][72 + j] + srec[1][72 + j]);
  }
  __syncthreads();
  const size_t off = (size_t)blk * n * n;
  ortho_bwd_group(t8p_lds, M + off, n, gQ ? gQ + off : nullptr, sG[0], Q ? Q + off : nullptr, part1 ? sG[1] : nullptr,
                  gM + off);
}

// The same tail with the optimiser and the next step's rotations inside (as k_tf_tail for the 4-line blocks): partial rows ->
// gradients -> Adam on the block's own entries of M, b, c (every element by the thread that wrote its gradient:
// dL/db[i] thread 128 + i, dL/dc[j] thread 192 + j, dL/dM[e] thread e) -> Q = expm(skew(M_new)), Q Q and a snapshot of the
// updated output gains (what the next step's forward pass reads while its normalize rescales them) -- one workgroup per
// block; the last one advances the step counter.
// (no __restrict__ on M, Q, gM, Qn, QQn: M is a view of the flat parameter buffer this launch steps through ``ad``, and the
// caller hands the SAME buffers as Q / Qn (the records that outlive the step) -- as k_tf_tail of blocktf.hip)
__global__ __launch_bounds__(256) void k_tf8_tail(const float* __restrict__ part0, const float* __restrict__ part1, int n,
                                                  const float* M, const float* __restrict__ gQ, const float* Q,
                                                  float* __restrict__ gb, float* __restrict__ gc, float* gM, int half2,
                                                  TfAdam ad, float* Qn, float* QQn, float* __restrict__ c_next) {
  extern __shared__ double t8p_lds[];
  __shared__ float srec[2][T8_ACC], sG[2][64], sM[64];
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const float t = ad.step_count[0] + 1.0f;
  for (int r = tid >> 6; r < 2 * T8_ACC; r += nw) {
    const int set = r / T8_ACC, e = r - set * T8_ACC;
    const float* p = set ? part1 : part0;
    float s = 0.f;
    if (lane == 0) s = p[(size_t)blk * T8_ACC + e] + p[(size_t)blk * T8_ACC + e + half2];
    if (lane == 0) srec[set][e] = s;
  }
  __syncthreads();
  if (tid < 2 * n * n) {
    const int set = tid / (n * n), e = tid - set * n * n, i = e / n, j = e - i * n;
    sG[set][e] = srec[set][i * 8 + j];
  } else if (tid >= 128 && tid < 128 + n) {
    const int i = tid - 128;
    gb[blk * n + i] = srec[0][64 + i] + srec[1][64 + i];
  } else if (tid >= 192 && tid < 192 + n) {
    const int j = tid - 192;
    gc[blk * n + j] = srec[0][72 + j] + srec[1][72 + j];
  }
  __syncthreads();
  const size_t off = (size_t)blk * n * n;
  ortho_bwd_group(t8p_lds, M + off, n, gQ ? gQ + off : nullptr, sG[0], Q ? Q + off : nullptr, sG[1], gM + off);
  const float bc1 = 1.0f - powf(ad.b1, t), bc2_sqrt = sqrtf(1.0f - powf(ad.b2, t));
  if (tid < n * n) sM[tid] = tf_adam_elem(ad, ad.offM + (int)off + tid, gM[off + tid], bc1, bc2_sqrt);
  if (tid >= 128 && tid < 128 + n) tf_adam_elem(ad, ad.offb + blk * n + tid - 128, gb[blk * n + tid - 128], bc1, bc2_sqrt);
  if (tid >= 192 && tid < 192 + n)
    c_next[blk * n + tid - 192] = tf_adam_elem(ad, ad.offc + blk * n + tid - 192, gc[blk * n + tid - 192], bc1, bc2_sqrt);
  __syncthreads();
  {
    double* A = t8p_lds;
    for (int e = tid; e < n * n; e += blockDim.x) A[e] = skew_elem(sM, n, e / n, e % n);
    __syncthreads();
    const double* E = expm_lds(A, n);
    for (int e = tid; e < n * n; e += blockDim.x) {
      Qn[off + e] = (float)E[e];
      const int i = e / n, j = e - i * n;
      double acc = 0.0;
      for (int q = 0; q < n; ++q) acc += E[i * n + q] * E[q * n + j];
      QQn[off + e] = (float)acc;
    }
  }
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(ad.block_counter, 1u) == gridDim.x - 1) {
      ad.step_count[0] = t;
      ad.block_counter[0] = 0u;
    }
  }
}

extern "C" size_t gfdn_tf8_param_grads_work_bytes(int nblk) { return (size_t)4 * (nblk > 0 ? nblk : 1) * T8_ACC * sizeof(float); }

extern "C" int gfdn_tf8_param_grads(const float* A0, const float* inv_gamma0, const float* part0, int nparts0,
                                    const float* A1, const float* inv_gamma1, const float* part1, int nparts1,
                                    const float* b, const float* c, int nblk, int nper, const float* M, const float* gQ,
                                    const float* Q, float* gb, float* gc, float* gM, void* work, void* stream) {
  if (!A0 || !part0 || !b || !c || !M || !gb || !gc || !gM || !work || nblk <= 0 || nper <= 0 || nparts0 <= 0 ||
      (A1 && (!part1 || nparts1 <= 0)))
    return GFDN_E_BADARG;
  if (nper > 8) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* out0 = (float*)work;
  float* out1 = out0 + (size_t)nblk * T8_ACC;
  hipLaunchKernelGGL(k_tf8_rec_grads, dim3(nblk, A1 ? 2 : 1, 2), dim3(256), 0, s, A0, inv_gamma0, part0, nparts0, A1,
                     inv_gamma1, part1, nparts1, b, c, nper, out0, out1);
  GFDN_LAUNCH_CHECK();
  const size_t lds = ortho_bwd_lds_doubles(nper) * sizeof(double);
  int rc = ensure_dyn_lds(k_tf8_param_grads, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tf8_param_grads, dim3(nblk), dim3(256), lds, s, (const float*)out0, 1, A1 ? (const float*)out1 : nullptr,
                     1, (const float*)nullptr, nper, M, gQ, Q, gb, gc, gM, nblk * 2 * T8_ACC);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// gfdn_tf8_param_grads with the optimiser update of the blocks' own M, b, c (flat buffers of gfdn_adam_step at the element
// offsets offM / offb / offc; same roundings: adam_elem) and the next step's Q, Q Q and output-gain snapshot in the same
// launch (k_tf8_tail).  Both record sets are required.
extern "C" int gfdn_tf8_tail(const float* A0, const float* inv_gamma0, const float* part0, int nparts0, const float* A1,
                             const float* part1, int nparts1, const float* b, const float* c, int nblk, int nper,
                             const float* M, const float* gQ, const float* Q, float* gb, float* gc, float* gM, void* work,
                             float* flat_p, float* flat_m, float* flat_v, const unsigned char* seg, const float* lr_seg,
                             float* step_count, unsigned int* block_counter, int offM, int offb, int offc, float beta1,
                             float beta2, float eps, float* Q_next, float* QQ_next, float* c_next, void* stream) {
  if (!A0 || !part0 || !A1 || !part1 || !b || !c || !M || !gb || !gc || !gM || !work || !flat_p || !flat_m || !flat_v || !seg ||
      !lr_seg || !step_count || !block_counter || !Q_next || !QQ_next || !c_next || nblk <= 0 || nper <= 0 || nparts0 <= 0 ||
      nparts1 <= 0 || offM < 0 || offb < 0 || offc < 0)
    return GFDN_E_BADARG;
  if (nper > 8) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* out0 = (float*)work;
  float* out1 = out0 + (size_t)nblk * T8_ACC;
  hipLaunchKernelGGL(k_tf8_rec_grads, dim3(nblk, 2, 2), dim3(256), 0, s, A0, inv_gamma0, part0, nparts0, A1,
                     (const float*)nullptr, part1, nparts1, b, c, nper, out0, out1);
  GFDN_LAUNCH_CHECK();
  TfAdam ad{flat_p, flat_m, flat_v, seg, lr_seg, step_count, block_counter, offM, offb, offc, beta1, beta2, eps};
  const size_t lds = ortho_bwd_lds_doubles(nper) * sizeof(double);
  int rc = ensure_dyn_lds(k_tf8_tail, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tf8_tail, dim3(nblk), dim3(256), lds, s, (const float*)out0, (const float*)out1, nper, M, gQ, Q, gb, gc,
                     gM, nblk * 2 * T8_ACC, ad, Q_next, QQ_next, c_next);
  GFDN_LAUNCH_CHECK();
  return 0;
}
