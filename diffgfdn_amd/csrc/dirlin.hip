// The directional model's output stage in the time domain (reference model.py:1056-1088 behind trainer.py:853-865 and
// losses.py:333-371), gfx950.
//
// The reference forms, per receiver b of a batch and SH channel l,
//     H_sh[b][l][k] = filt_k sum_g w[b][g nper + l] c[g nper + l] Y[k][g nper + l]
// (Y: the N = G nper delay-line responses of the transposed solve, c: the output gains, w: the receiver's SH weights)
// and transforms every one of the B nper responses to time.  The receiver enters through N real scalars, everything
// else is linear:
//     x_sh[b][l] = irfft(H_sh[b][l]) = sum_g w[b][g nper + l] tau[g nper + l],     tau[n] = irfft(c_n filt Y[:, n]).
// The step transforms the N line signals (27 at order 2 with three groups) instead of the B nper receiver signals (288),
// forms the receivers' SH signals on the EDC window only, and takes the adjoint the same way:
//     dL/dtau[n] = sum_b w[b][n] dL/dx_sh[b][l(n)],        dL/dw[b][n] = <dL/dx_sh[b][l(n)], tau[n]>.
// The (B, nper, K) responses, their gradient and the SH output stage's two passes over them (gfdn_compose_sh_fwd / _bwd)
// do not exist on this path.
//
//   k_dl_lines_fwd  : Z[n][k] = c_n filt_k Y[k][n]                         (bin-major Y -> line-major rows for the transform)
//   k_dl_combine    : x_sh[b nper + l][t] = sum_g w[b][g nper + l] tau[g nper + l][start + t],  t in [0, len)
//   k_dl_gamma_dots : gtau[n][start + t] = sum_b w[b][n] gx[b nper + l][t]  and per-tile partial sums of dL/dw
//   k_dl_lines_bwd  : gY[k][n] = c_n conj(filt_k) gZ[n][k],  per-tile partial sums of dL/dc_n = sum_k Re(gZ conj(filt Y))
//   k_dl_rowsum     : fixed-order sums of the partial rows
// All sums in fixed order (no atomics): the step is bitwise reproducible.
#include "common.h"
#include "../../include/diffgfdn_hip.h"

#define DL_TB 64          // bins per tile of the line kernels
#define DL_T 256          // threads of the time-domain kernels
#define DL_V 4            // samples per thread
#define DL_TILE (DL_T * DL_V)
#define DL_RB 8           // receivers per register chunk of k_dl_gamma_dots
#define DL_GMAX 4

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

extern __shared__ float2 dl_lds[];

// ---- lines: Y (K, N) bin-major -> Z (N, ldz) --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dl_lines_fwd(const float2* __restrict__ Y, int K, int N,
                                                      const float* __restrict__ c, const float2* __restrict__ filt,
                                                      float2* __restrict__ Z, int ldz) {
  const int NS = N + 1 + (N & 1);                       // odd row stride
  float2* yt = dl_lds;                                  // [DL_TB][NS]
  const int k0 = blockIdx.x * DL_TB;
  const int nbin = K - k0 < DL_TB ? K - k0 : DL_TB;
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < nbin * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = Y[base + e];
  }
  __syncthreads();
  const int kq = threadIdx.x & (DL_TB - 1);
  if (kq >= nbin) return;
  const float2 f = filt ? filt[k0 + kq] : make_float2(1.f, 0.f);
  for (int n = threadIdx.x / DL_TB; n < N; n += 256 / DL_TB) {
    const float2 y = yt[kq * NS + n];
    Z[(size_t)n * ldz + k0 + kq] = cscale(cmul(y, f), c[n]);
  }
}

// gY[k][n] = c_n conj(f_k) gZ[n][k];   gc_part[n][tile] = sum over the tile's bins of Re(gZ conj(f Y))
__global__ __launch_bounds__(256) void k_dl_lines_bwd(const float2* __restrict__ Y, int K, int N,
                                                      const float* __restrict__ c, const float2* __restrict__ filt,
                                                      const float2* __restrict__ gZ, int ldz, float2* __restrict__ gY,
                                                      float* __restrict__ gc_part, int ntiles) {
  const int NS = N + 1 + (N & 1);
  float2* gt = dl_lds;                                  // [DL_TB][NS]: gZ conj(f) of the tile, bin-major
  float* pr = (float*)(gt + DL_TB * NS);                // [DL_TB][N + 1]: Re(gZ conj(f Y))
  const int k0 = blockIdx.x * DL_TB;
  const int nbin = K - k0 < DL_TB ? K - k0 : DL_TB;
  const int kq = threadIdx.x & (DL_TB - 1);
  if (kq < nbin) {
    const float2 f = filt ? filt[k0 + kq] : make_float2(1.f, 0.f);
    for (int n = threadIdx.x / DL_TB; n < N; n += 256 / DL_TB)
      gt[kq * NS + n] = cmulc(gZ[(size_t)n * ldz + k0 + kq], f);
  }
  __syncthreads();
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < DL_TB * N; e += 256) {
    const int q = e / N, n = e - q * N;
    float p = 0.f;
    if (q < nbin) {
      const float2 g = gt[q * NS + n], y = Y[base + e];
      gY[base + e] = cscale(g, c[n]);
      p = g.x * y.x + g.y * y.y;
    }
    pr[q * (N + 1) + n] = p;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
    for (int q = 0; q < DL_TB; ++q) s += pr[q * (N + 1) + threadIdx.x];
    gc_part[(size_t)threadIdx.x * ntiles + blockIdx.x] = s;
  }
}

// one wave per row: out[r] = sum of part[r][0 .. cols) in a fixed order
__global__ __launch_bounds__(64) void k_dl_rowsum(const float* __restrict__ part, int cols, int ld, float* __restrict__ out) {
  const float* p = part + (size_t)blockIdx.x * ld;
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 64) s += p[j];
  s = wave_sum_full(s);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---- the receivers' SH signals on the window from the line signals ---------------------------------------------------
// grid (tiles of DL_TILE window samples, nper): the thread holds its 4 samples of the channel's G line signals and walks
// the receivers (w[b][n] is uniform: scalar loads).
template <int G>
__global__ __launch_bounds__(DL_T) void k_dl_combine(const float* __restrict__ tau, int ld_tau, int start, int len,
                                                     const float* __restrict__ w, int B, int nper,
                                                     float* __restrict__ x, int ld_x) {
  const int l = blockIdx.y, N = G * nper;
  const int t0 = blockIdx.x * DL_TILE + threadIdx.x * DL_V;
  if (t0 >= len) return;
  const bool full = t0 + DL_V <= len;
  float tv[G][DL_V];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* tp = tau + (size_t)(g * nper + l) * ld_tau + start + t0;
    if (full) {
      const f4u v = *(const f4u*)tp;
      tv[g][0] = v.x; tv[g][1] = v.y; tv[g][2] = v.z; tv[g][3] = v.w;
    } else {
#pragma unroll
      for (int u = 0; u < DL_V; ++u) tv[g][u] = t0 + u < len ? tp[u] : 0.f;
    }
  }
  for (int b = 0; b < B; ++b) {
    float o[DL_V] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float wg = w[(size_t)b * N + g * nper + l];
#pragma unroll
      for (int u = 0; u < DL_V; ++u) o[u] += wg * tv[g][u];
    }
    float* xp = x + (size_t)(b * nper + l) * ld_x + t0;       // (ld_x multiple of 4, t0 multiple of 4: aligned)
    *(float4*)xp = make_float4(o[0], o[1], o[2], o[3]);       // (samples >= len of the last group: zeros, inside the pitch)
  }
}

// ---- adjoint: gradient of the line signals + partial sums of dL/dw --------------------------------------------------
// Same grid.  part[(b N + n) tiles + tile] = sum over the tile's samples of gx[b nper + l][t] tau[n][start + t].
template <int G>
__global__ __launch_bounds__(DL_T) void k_dl_gamma_dots(const float* __restrict__ gx, int ld_g, int len,
                                                        const float* __restrict__ tau, int ld_tau, int start,
                                                        const float* __restrict__ w, int B, int nper,
                                                        float* __restrict__ gtau, int ld_o, float* __restrict__ part,
                                                        int tiles) {
  __shared__ float red[DL_T / 64][DL_RB * DL_GMAX];
  const int l = blockIdx.y, N = G * nper;
  const int t0 = blockIdx.x * DL_TILE + threadIdx.x * DL_V;
  const bool live = t0 < len, full = t0 + DL_V <= len;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float tv[G][DL_V], acc[G][DL_V];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* tp = tau + (size_t)(g * nper + l) * ld_tau + start + t0;
    if (full) {
      const f4u v = *(const f4u*)tp;
      tv[g][0] = v.x; tv[g][1] = v.y; tv[g][2] = v.z; tv[g][3] = v.w;
    } else {
#pragma unroll
      for (int u = 0; u < DL_V; ++u) tv[g][u] = (live && t0 + u < len) ? tp[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DL_V; ++u) acc[g][u] = 0.f;
  }
  for (int b0 = 0; b0 < B; b0 += DL_RB) {
    float dots[DL_RB][G];
#pragma unroll
    for (int bb = 0; bb < DL_RB; ++bb) {
      const int b = b0 + bb;
      float v[DL_V] = {0.f, 0.f, 0.f, 0.f};
      if (b < B && live) {                         // (b < B is uniform; the window buffer's pitch covers the last group)
        const float4 q = *(const float4*)(gx + (size_t)(b * nper + l) * ld_g + t0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        if (!full) {
#pragma unroll
          for (int u = 0; u < DL_V; ++u)
            if (t0 + u >= len) v[u] = 0.f;
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float wg = b < B ? w[(size_t)b * N + g * nper + l] : 0.f;
        float d = 0.f;
#pragma unroll
        for (int u = 0; u < DL_V; ++u) {
          acc[g][u] += wg * v[u];
          d += v[u] * tv[g][u];
        }
        dots[bb][g] = wave_sum_full(d);
      }
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int bb = 0; bb < DL_RB; ++bb)
#pragma unroll
        for (int g = 0; g < G; ++g) red[wv][bb * DL_GMAX + g] = dots[bb][g];
    }
    __syncthreads();
    if (threadIdx.x < DL_RB * G) {
      const int bb = threadIdx.x / G, g = threadIdx.x - bb * G;
      if (b0 + bb < B) {
        const int e = bb * DL_GMAX + g;
        part[((size_t)(b0 + bb) * N + g * nper + l) * tiles + blockIdx.x] =
            ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* op = gtau + (size_t)(g * nper + l) * ld_o + start + t0;
    if (full) {
      f4u v;
      v.x = acc[g][0]; v.y = acc[g][1]; v.z = acc[g][2]; v.w = acc[g][3];
      *(f4u*)op = v;
    } else {
#pragma unroll
      for (int u = 0; u < DL_V; ++u)
        if (t0 + u < len) op[u] = acc[g][u];
    }
  }
}

// ---- group sums of the raw sub-FDN responses (reference model.py:243-250: Hout[k][g] = sum_{n in g} c_n y_n[k]) --------
// The colorless branch of every model needs only these G sums; through the general output stage (G "receivers" with identity
// gains, gfdn_compose_fwd / _bwd) the backward alone took 48 us per band-step of the directional model for 28 MB of traffic.
//   k_gs_fwd : S (G, K) = sum over the group's lines of c_n Y[k][n]          (Y (K, N) bin-major, N = G nper)
//   k_gs_bwd : gY[k][n] = c_n gS[g(n)][k],  per-tile partial sums of gc_n = sum_k Re(conj(gS[g(n)][k]) Y[k][n])
__global__ __launch_bounds__(256) void k_gs_fwd(const float2* __restrict__ Y, int K, int G, int nper,
                                                const float* __restrict__ c, float2* __restrict__ S) {
  const int N = G * nper, NS = N + 1 + (N & 1);
  float2* yt = dl_lds;                                  // [DL_TB][NS]
  const int k0 = blockIdx.x * DL_TB;
  const int nbin = K - k0 < DL_TB ? K - k0 : DL_TB;
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < nbin * N; e += 256) {
    const int kq = e / N, n = e - kq * N;
    yt[kq * NS + n] = cscale(Y[base + e], c[n]);
  }
  __syncthreads();
  const int kq = threadIdx.x & (DL_TB - 1);
  if (kq >= nbin) return;
  for (int g = threadIdx.x / DL_TB; g < G; g += 256 / DL_TB) {
    float2 acc = make_float2(0.f, 0.f);
    for (int l = 0; l < nper; ++l) acc = cadd(acc, yt[kq * NS + g * nper + l]);
    S[(size_t)g * K + k0 + kq] = acc;
  }
}

__global__ __launch_bounds__(256) void k_gs_bwd(const float2* __restrict__ Y, int K, int G, int nper,
                                                const float* __restrict__ c, const float2* __restrict__ gS,
                                                float2* __restrict__ gY, float* __restrict__ gc_part, int ntiles) {
  const int N = G * nper;
  float2* gt = dl_lds;                                  // [G][DL_TB]: gS of the tile
  float* pr = (float*)(gt + G * DL_TB);                 // [DL_TB][N + 1]
  const int k0 = blockIdx.x * DL_TB;
  const int nbin = K - k0 < DL_TB ? K - k0 : DL_TB;
  for (int e = threadIdx.x; e < G * DL_TB; e += 256) {
    const int g = e / DL_TB, q = e - g * DL_TB;
    gt[e] = q < nbin ? gS[(size_t)g * K + k0 + q] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const size_t base = (size_t)k0 * N;
  for (int e = threadIdx.x; e < DL_TB * N; e += 256) {
    const int q = e / N, n = e - q * N;
    float p = 0.f;
    if (q < nbin) {
      const float2 g = gt[(n / nper) * DL_TB + q], y = Y[base + e];
      gY[base + e] = cscale(g, c[n]);
      p = g.x * y.x + g.y * y.y;
    }
    pr[q * (N + 1) + n] = p;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
    for (int q = 0; q < DL_TB; ++q) s += pr[q * (N + 1) + threadIdx.x];
    gc_part[(size_t)threadIdx.x * ntiles + blockIdx.x] = s;
  }
}

extern "C" int gfdn_group_sums_fwd(const float* Y, int K, int G, int nper, const float* c, float* S, void* stream) {
  if (!Y || !c || !S || K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  const int N = G * nper;
  if (N > 64) return GFDN_E_UNSUPPORTED;
  const int NS = N + 1 + (N & 1);
  hipLaunchKernelGGL(k_gs_fwd, dim3((K + DL_TB - 1) / DL_TB), dim3(256), (size_t)DL_TB * NS * sizeof(float2),
                     (hipStream_t)stream, (const float2*)Y, K, G, nper, c, (float2*)S);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_group_sums_bwd(const float* Y, int K, int G, int nper, const float* c, const float* gS, float* gY,
                                   float* gc, float* gc_part, void* stream) {
  if (!Y || !c || !gS || !gY || !gc || !gc_part || K <= 0 || G <= 0 || nper <= 0) return GFDN_E_BADARG;
  const int N = G * nper;
  if (N > 64) return GFDN_E_UNSUPPORTED;
  const int ntiles = (K + DL_TB - 1) / DL_TB;
  const size_t lds = (size_t)G * DL_TB * sizeof(float2) + (size_t)DL_TB * (N + 1) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gs_bwd, dim3(ntiles), dim3(256), lds, s, (const float2*)Y, K, G, nper, c, (const float2*)gS,
                     (float2*)gY, gc_part, ntiles);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dl_rowsum, dim3(N), dim3(64), 0, s, (const float*)gc_part, ntiles, ntiles, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int gfdn_dirlin_tiles(int len) { return len > 0 ? (len + DL_TILE - 1) / DL_TILE : 0; }
extern "C" int gfdn_dirlin_line_tiles(int K) { return K > 0 ? (K + DL_TB - 1) / DL_TB : 0; }

extern "C" int gfdn_dirlin_lines_fwd(const float* Y, int K, int N, const float* c, const float* filt, float* Z, int ldz,
                                     void* stream) {
  if (!Y || !c || !Z || K <= 0 || N <= 0 || ldz < K) return GFDN_E_BADARG;
  if (N > 64) return GFDN_E_UNSUPPORTED;
  const int NS = N + 1 + (N & 1);
  const size_t lds = (size_t)DL_TB * NS * sizeof(float2);
  hipLaunchKernelGGL(k_dl_lines_fwd, dim3((K + DL_TB - 1) / DL_TB), dim3(256), lds, (hipStream_t)stream,
                     (const float2*)Y, K, N, c, (const float2*)filt, (float2*)Z, ldz);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_dirlin_lines_bwd(const float* Y, int K, int N, const float* c, const float* filt, const float* gZ,
                                     int ldz, float* gY, float* gc, float* gc_part, void* stream) {
  if (!Y || !c || !gZ || !gY || !gc || !gc_part || K <= 0 || N <= 0 || ldz < K) return GFDN_E_BADARG;
  if (N > 64) return GFDN_E_UNSUPPORTED;
  const int NS = N + 1 + (N & 1), ntiles = (K + DL_TB - 1) / DL_TB;
  const size_t lds = (size_t)DL_TB * NS * sizeof(float2) + (size_t)DL_TB * (N + 1) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_dyn_lds(k_dl_lines_bwd, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_dl_lines_bwd, dim3(ntiles), dim3(256), lds, s, (const float2*)Y, K, N, c, (const float2*)filt,
                     (const float2*)gZ, ldz, (float2*)gY, gc_part, ntiles);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dl_rowsum, dim3(N), dim3(64), 0, s, (const float*)gc_part, ntiles, ntiles, gc);
  GFDN_LAUNCH_CHECK();
  return 0;
}

template <int G>
static int dl_combine(const float* tau, int ld_tau, int start, int len, const float* w, int B, int nper, float* x, int ld_x,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_dl_combine<G>, dim3((len + DL_TILE - 1) / DL_TILE, nper), dim3(DL_T), 0, s, tau, ld_tau, start, len,
                     w, B, nper, x, ld_x);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_dirlin_combine(const float* tau, int ld_tau, int start, int len, const float* w, int B, int G, int nper,
                                   float* x, int ld_x, void* stream) {
  if (!tau || !w || !x || start < 0 || len <= 0 || start + len > ld_tau || B <= 0 || G <= 0 || nper <= 0 ||
      ld_x < ((len + 3) & ~3) || (ld_x & 3) || ((uintptr_t)x & 15))
    return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  switch (G) {
    case 1: return dl_combine<1>(tau, ld_tau, start, len, w, B, nper, x, ld_x, s);
    case 2: return dl_combine<2>(tau, ld_tau, start, len, w, B, nper, x, ld_x, s);
    case 3: return dl_combine<3>(tau, ld_tau, start, len, w, B, nper, x, ld_x, s);
    case 4: return dl_combine<4>(tau, ld_tau, start, len, w, B, nper, x, ld_x, s);
    default: return GFDN_E_UNSUPPORTED;
  }
}

template <int G>
static int dl_gamma_dots(const float* gx, int ld_g, int len, const float* tau, int ld_tau, int start, const float* w, int B,
                         int nper, float* gtau, int ld_o, float* part, float* gw, hipStream_t s) {
  const int tiles = (len + DL_TILE - 1) / DL_TILE;
  hipLaunchKernelGGL(k_dl_gamma_dots<G>, dim3(tiles, nper), dim3(DL_T), 0, s, gx, ld_g, len, tau, ld_tau, start, w, B, nper,
                     gtau, ld_o, part, tiles);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dl_rowsum, dim3(B * G * nper), dim3(64), 0, s, (const float*)part, tiles, tiles, gw);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_dirlin_gamma_dots(const float* gx, int ld_g, int len, const float* tau, int ld_tau, int start,
                                      const float* w, int B, int G, int nper, float* gtau, int ld_o, float* gw, float* part,
                                      void* stream) {
  if (!gx || !tau || !w || !gtau || !gw || !part || start < 0 || len <= 0 || start + len > ld_tau || start + len > ld_o ||
      B <= 0 || G <= 0 || nper <= 0 || ld_g < ((len + 3) & ~3) || (ld_g & 3) || ((uintptr_t)gx & 15))
    return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  switch (G) {
    case 1: return dl_gamma_dots<1>(gx, ld_g, len, tau, ld_tau, start, w, B, nper, gtau, ld_o, part, gw, s);
    case 2: return dl_gamma_dots<2>(gx, ld_g, len, tau, ld_tau, start, w, B, nper, gtau, ld_o, part, gw, s);
    case 3: return dl_gamma_dots<3>(gx, ld_g, len, tau, ld_tau, start, w, B, nper, gtau, ld_o, part, gw, s);
    case 4: return dl_gamma_dots<4>(gx, ld_g, len, tau, ld_tau, start, w, B, nper, gtau, ld_o, part, gw, s);
    default: return GFDN_E_UNSUPPORTED;
  }
}
