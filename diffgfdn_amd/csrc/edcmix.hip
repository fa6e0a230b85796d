// Directional EDC loss on SH-domain time signals (reference losses.py:333-371 behind trainer.py:853-865), gfx950.
//
// The reference converts the SH-domain responses to J directional ones (einsum 'jl,blk->bjk' with the real analysis
// matrix A), transforms those to time and compares their energy-decay curves with the common-slope model.  A is real
// and the transform linear, so x_dir[b][j][t] = sum_c A[j][c] x_sh[b][c][t] with x_sh = irfft(H_sh): these kernels read
// the C SH-domain signals of a receiver and form the J directional samples in registers.  The directional signals,
// their gradient and the staged dL/dEDC never exist in memory (a step of 32 receivers x 12 directions moved
// 2.1 GB through the mix kernels and the three EDC scans; this moves 0.4 GB).
//
// One workgroup = one receiver x one segment of EM_SEG = 512 samples of the window, everything in registers:
//   k_em_segsum : seg[b][j][s]   = sum over the segment of x_dir^2
//   k_em_carry  : exclusive SUFFIX sums of seg over s (the energy behind the segment)
//   k_em_fwd    : EDC_j(t) = carry + suffix scan inside the segment -> dB, |target - EDC| -> partial loss, dL/dEDC;
//                 part[b][j][s] = loss of the segment, gs[b][j][s] = sum of dL/dEDC over the segment
//   k_em_carry  : exclusive PREFIX sums of gs over s; loss_item[b][j] = gscale inv_count sum_s part (the weighted term)
//   k_em_bwd    : dL/dEDC recomputed as in k_em_fwd (not staged), prefix scan inside the segment + carry
//                 -> dL/dx_dir = 2 x_dir prefix -> dL/dx_sh[c] = sum_j A[j][c] dL/dx_dir[j], stored for the window only
//                 (the adjoint transform is told the window, gfdn_irfft_pow2_bwd_window: nothing outside is read).
// Same arithmetic per sample as k_edc_seg_fwd / k_edc_seg_bwd (losses.hip); the summation order of the scans differs
// (segments of 512 instead of eighths of the window): rounding-level differences.
#include "common.h"
#include "../../include/diffgfdn_hip.h"

#define EM_T 256
#define EM_V 2
#define EM_SEG (EM_T * EM_V)
#define EM_JMAX 16      // directions (the template parameter JT = 8, 12 or 16 sizes the registers)
#define EM_SMAX 8       // slopes of the common-slope model
#define EM_DB 3.0102999566398120f      // 10 log10(x) = EM_DB log2(x)

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

struct EmArgs {
  const float* x;        // (B C, ld)
  int ld, B, J, start, len, nseg;
  const float* A;        // (J, C)
  const float* amps;     // (B J, S)
  int S;
  const float* env;      // (S, ld_env)
  int ld_env;
  const float* maskw;    // (len) or null
  float inv_count, gscale;
  float* seg;            // [B][J][nseg] energies, then their exclusive suffix sums
  float* part;           // [B][J][nseg] loss partials
  float* gs;             // [B][J][nseg] sums of dL/dEDC, then their exclusive prefix sums
  float* loss_item;      // (B J)
  float* gx;             // (B C, ld) or null
};

// the thread's EM_V consecutive samples of the C channels -> J directional samples; i0 = first sample (window index)
template <int C, int JT>
__device__ __forceinline__ void em_load(const EmArgs& a, const float* sA, int b, int i0, float (&xd)[JT][EM_V]) {
  const float* xb = a.x + (size_t)b * C * a.ld + a.start + i0;
  const bool full = i0 + EM_V <= a.len;
  float xs[C][EM_V];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if (full) {
      const f2u v = *(const f2u*)(xb + (size_t)c * a.ld);
      xs[c][0] = v.x; xs[c][1] = v.y;
    } else {
#pragma unroll
      for (int u = 0; u < EM_V; ++u) xs[c][u] = i0 + u < a.len ? xb[(size_t)c * a.ld + u] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < JT; ++j) {
#pragma unroll
    for (int u = 0; u < EM_V; ++u) xd[j][u] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float w = sA[j * C + c];              // (rows j >= J of the staged matrix are zero)
#pragma unroll
      for (int u = 0; u < EM_V; ++u) xd[j][u] += w * xs[c][u];
    }
  }
}

// sum of v over the workgroup for each of the J values; result in every thread.  lds: 4 * EM_JMAX floats
template <int JT>
__device__ __forceinline__ void em_block_sums(float (&v)[JT], float* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < JT; ++j) v[j] = wave_sum_valu(v[j]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < JT; ++j) lds[w * EM_JMAX + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < JT; ++j) v[j] = lds[j] + lds[EM_JMAX + j] + lds[2 * EM_JMAX + j] + lds[3 * EM_JMAX + j];
}

// For each j: the thread's value v[j] (sum of its own samples) -> sum of the values of the threads BEHIND it (REV: of
// higher thread index; else of lower), exclusive.  lds: 4 * EM_JMAX floats.
template <bool REV, int JT>
__device__ __forceinline__ void em_block_scan_excl(float (&v)[JT], float* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float own[JT], tot[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    own[j] = v[j];
    v[j] = REV ? wave_scan_incl_rev(v[j], tot[j]) : wave_scan_incl(v[j], tot[j]);      // (VALU scans, common.h)
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < JT; ++j) lds[w * EM_JMAX + j] = tot[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    float pre = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (REV ? (q > w) : (q < w)) pre += lds[q * EM_JMAX + j];
    v[j] = v[j] - own[j] + pre;
  }
}

// value v[threadIdx.x] of a register array (threads 0 .. J-1)
template <int JT>
__device__ __forceinline__ float em_pick(const float (&v)[JT]) {
  float mine = 0.f;
#pragma unroll
  for (int j = 0; j < JT; ++j)
    if (j == (int)threadIdx.x) mine = v[j];
  return mine;
}

template <int C, int JT>
__global__ __launch_bounds__(EM_T, 4) void k_em_segsum(EmArgs a) {
  __shared__ float sA[JT * C];
  __shared__ float red[4 * EM_JMAX];
  const int s = blockIdx.x, b = blockIdx.y;
  for (int e = threadIdx.x; e < JT * C; e += EM_T) sA[e] = e < a.J * C ? a.A[e] : 0.f;
  __syncthreads();
  float xd[JT][EM_V], v[JT];
  em_load<C, JT>(a, sA, b, s * EM_SEG + threadIdx.x * EM_V, xd);
#pragma unroll
  for (int j = 0; j < JT; ++j) v[j] = xd[j][0] * xd[j][0] + xd[j][1] * xd[j][1];
  em_block_sums<JT>(v, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < JT; ++j)
      if (j < a.J) a.seg[((size_t)b * a.J + j) * a.nseg + s] = v[j];
  }
}

// one wave per row (b, j) of nseg values: exclusive suffix (suffix != 0) or prefix sums in place; with part / loss:
// loss[row] = scale * sum part[row][:]
__global__ __launch_bounds__(64) void k_em_carry(float* __restrict__ v, int nseg, int suffix, const float* __restrict__ part,
                                                 float scale, float* __restrict__ loss) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float* r = v + (size_t)row * nseg;
  float carry = 0.f;
  for (int base = 0; base < nseg; base += 64) {
    const int i = base + lane;
    const int idx = suffix ? nseg - 1 - i : i;
    const float own = i < nseg ? r[idx] : 0.f;
    float incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (i < nseg) r[idx] = carry + incl - own;
    carry += __shfl(incl, 63, 64);
  }
  if (part) {
    float s = 0.f;
    for (int i = lane; i < nseg; i += 64) s += part[(size_t)row * nseg + i];
    s = wave_sum(s);
    if (lane == 0) loss[row] = s * scale;
  }
}

// dL/dEDC (g) of the thread's samples for every direction and the loss terms summed over the thread's samples (lsum).
// xd: directional samples; i0: first sample of the thread (window index).  EDC_i = energy behind the segment (seg, after
// its suffix scan) + later threads of the segment + the thread's own later sample + x_i^2; dB, target and gradient as in
// k_edc_seg_fwd (losses.hip), 10 log10 through the hardware log2.
template <int JT>
__device__ __forceinline__ void em_edc_grad(const EmArgs& a, int b, int s, int i0, const float (&xd)[JT][EM_V],
                                            float (&g)[JT][EM_V], float (&lsum)[JT], float* red) {
  float v[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) v[j] = xd[j][0] * xd[j][0] + xd[j][1] * xd[j][1];
  em_block_scan_excl<true, JT>(v, red);
  float m[EM_V], ev[EM_SMAX][EM_V];
#pragma unroll
  for (int u = 0; u < EM_V; ++u) {
    const bool in = i0 + u < a.len;
    m[u] = in ? (a.maskw ? a.maskw[i0 + u] : 1.0f) : 0.f;
#pragma unroll
    for (int k = 0; k < EM_SMAX; ++k) ev[k][u] = (in && k < a.S) ? a.env[(size_t)k * a.ld_env + i0 + u] : 0.f;
  }
  // (the dB stage written for instruction count, as k_edc_lin_one's (edcone.hip): one compare shared by the -200 dB floor and
  // its gradient, the constants folded into the time weight, the sign of the difference copied as a bit)
  const float gneg = -(a.inv_count * a.gscale) * TEN_OVER_LN10;
  float mg[EM_V];
#pragma unroll
  for (int u = 0; u < EM_V; ++u) mg[u] = m[u] * gneg;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    // (a direction j >= J of the register tile has a zero row of A: x_dir = 0, and is given the amplitudes of direction 0
    // times zero: target = EDC = eps, difference 0, gradient 0 -- no branches on J in the body)
    const int jj = j < a.J ? j : 0;
    const float live = j < a.J ? 1.0f : 0.f;
    const float* am = a.amps + ((size_t)b * a.J + jj) * a.S;        // (uniform: scalar loads)
    float run = live * a.seg[((size_t)b * a.J + jj) * a.nseg + s] + v[j];
    lsum[j] = 0.f;
#pragma unroll
    for (int u = EM_V - 1; u >= 0; --u) {
      run += xd[j][u] * xd[j][u];
      float tl = 0.f;
#pragma unroll
      for (int k = 0; k < EM_SMAX; ++k)
        if (k < a.S) tl += am[k] * ev[k][u];
      const float traw = EM_DB * __log2f(fabsf(live * tl) + F32_EPS);
      const float tdb = traw > -200.0f ? traw : -200.0f;
      const float lin = fabsf(run) + F32_EPS;
      const float raw = EM_DB * __log2f(lin);
      const bool above = raw > -200.0f;
      const float diff = tdb - (above ? raw : -200.0f);
      lsum[j] += m[u] * fabsf(diff);
      // dL/dEDC = -sign(diff) (10 / ln 10) / lin x weight x scale; zero on the floor and where the difference vanishes
      const float mag = __builtin_amdgcn_rcpf(lin) * mg[u];
      const float sm = __uint_as_float((__float_as_uint(diff) & 0x80000000u) ^ __float_as_uint(mag));
      g[j][u] = (above && diff != 0.f) ? sm : 0.f;
    }
  }
}

template <int C, int JT>
__global__ __launch_bounds__(EM_T, 3) void k_em_fwd(EmArgs a) {
  __shared__ float sA[JT * C];
  __shared__ float red[4 * EM_JMAX];
  const int s = blockIdx.x, b = blockIdx.y;
  for (int e = threadIdx.x; e < JT * C; e += EM_T) sA[e] = e < a.J * C ? a.A[e] : 0.f;
  __syncthreads();
  const int i0 = s * EM_SEG + threadIdx.x * EM_V;
  float xd[JT][EM_V], g[JT][EM_V], lsum[JT], gsum[JT];
  em_load<C, JT>(a, sA, b, i0, xd);
  em_edc_grad<JT>(a, b, s, i0, xd, g, lsum, red);
#pragma unroll
  for (int j = 0; j < JT; ++j) gsum[j] = g[j][0] + g[j][1];
  em_block_sums<JT>(lsum, red);
  em_block_sums<JT>(gsum, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      if (j < a.J) {
        const size_t o = ((size_t)b * a.J + j) * a.nseg + s;
        a.part[o] = lsum[j];
        a.gs[o] = gsum[j];
      }
    }
  }
}

template <int C, int JT>
__global__ __launch_bounds__(EM_T, 3) void k_em_bwd(EmArgs a) {
  __shared__ float sA[JT * C];
  __shared__ float red[4 * EM_JMAX];
  const int s = blockIdx.x, b = blockIdx.y;
  for (int e = threadIdx.x; e < JT * C; e += EM_T) sA[e] = e < a.J * C ? a.A[e] : 0.f;
  __syncthreads();
  const int i0 = s * EM_SEG + threadIdx.x * EM_V;
  float xd[JT][EM_V], g[JT][EM_V], lsum[JT], v[JT];
  em_load<C, JT>(a, sA, b, i0, xd);
  em_edc_grad<JT>(a, b, s, i0, xd, g, lsum, red);
  // EDC_i = sum_{t >= i} x_t^2  =>  dL/dx_t = 2 x_t sum_{i <= t} dL/dEDC_i : prefix sums of g
#pragma unroll
  for (int j = 0; j < JT; ++j) v[j] = g[j][0] + g[j][1];
  em_block_scan_excl<false, JT>(v, red);
  float out[C][EM_V];
#pragma unroll
  for (int c = 0; c < C; ++c) { out[c][0] = 0.f; out[c][1] = 0.f; }
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    float run = a.gs[((size_t)b * a.J + (j < a.J ? j : 0)) * a.nseg + s] + v[j];      // (j >= J: times a zero row of A)
#pragma unroll
    for (int u = 0; u < EM_V; ++u) {
      run += g[j][u];
      const float gd = 2.0f * xd[j][u] * run;
#pragma unroll
      for (int c = 0; c < C; ++c) out[c][u] += sA[j * C + c] * gd;
    }
  }
  float* gb = a.gx + (size_t)b * C * a.ld + a.start + i0;
  if (i0 + EM_V <= a.len) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      f2u o; o.x = out[c][0]; o.y = out[c][1];
      *(f2u*)(gb + (size_t)c * a.ld) = o;
    }
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int u = 0; u < EM_V; ++u)
        if (i0 + u < a.len) gb[(size_t)c * a.ld + u] = out[c][u];
  }
}

extern "C" size_t gfdn_edc_mixed_work_bytes(int B, int J, int len) {
  if (B <= 0 || J <= 0 || len <= 0) return 0;
  const size_t nseg = ((size_t)len + EM_SEG - 1) / EM_SEG;
  return (size_t)3 * B * J * nseg * sizeof(float);
}

// stages: 1 = the forward chain (energies, carries, loss, carries), 2 = the backward kernel
template <int C, int JT>
static int em_run(EmArgs a, hipStream_t s, int stages) {
  dim3 grid(a.nseg, a.B), block(EM_T);
  if (!(stages & 1)) {
    if ((stages & 2) && a.gx) {
      hipLaunchKernelGGL((k_em_bwd<C, JT>), grid, block, 0, s, a);
      GFDN_LAUNCH_CHECK();
    }
    return 0;
  }
  hipLaunchKernelGGL((k_em_segsum<C, JT>), grid, block, 0, s, a);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_em_carry, dim3(a.B * a.J), dim3(64), 0, s, a.seg, a.nseg, 1, (const float*)nullptr, 0.f,
                     (float*)nullptr);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_em_fwd<C, JT>), grid, block, 0, s, a);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_em_carry, dim3(a.B * a.J), dim3(64), 0, s, a.gs, a.nseg, 0, (const float*)a.part,
                     a.inv_count * a.gscale, a.loss_item);
  GFDN_LAUNCH_CHECK();
  if ((stages & 2) && a.gx) {
    hipLaunchKernelGGL((k_em_bwd<C, JT>), grid, block, 0, s, a);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}
template <int C>
static int em_run_c(const EmArgs& a, hipStream_t s, int stages) {
  if (a.J <= 8) return em_run<C, 8>(a, s, stages);
  if (a.J <= 12) return em_run<C, 12>(a, s, stages);
  return em_run<C, 16>(a, s, stages);
}

extern "C" int gfdn_edc_loss_model_mixed_stages(const float* x_sh, int ld, int B, int C, const float* A, int J, int start,
                                                int len, const float* amps, int S, const float* env, int ld_env,
                                                const float* maskw, float inv_count, float gscale, float* loss_item,
                                                float* gx_sh, void* work, int stages, void* stream) {
  if ((stages & ~3) || !stages) return GFDN_E_BADARG;
  if (!x_sh || !A || !amps || !env || !loss_item || !work || B <= 0 || C <= 0 || J <= 0 || start < 0 || len <= 0 ||
      start + len > ld || S <= 0 || ld_env < len)
    return GFDN_E_BADARG;
  if (J > EM_JMAX || S > EM_SMAX) return GFDN_E_UNSUPPORTED;
  EmArgs a;
  a.x = x_sh; a.ld = ld; a.B = B; a.J = J; a.start = start; a.len = len;
  a.nseg = (len + EM_SEG - 1) / EM_SEG;
  a.A = A; a.amps = amps; a.S = S; a.env = env; a.ld_env = ld_env; a.maskw = maskw;
  a.inv_count = inv_count; a.gscale = gscale;
  const size_t n = (size_t)B * J * a.nseg;
  a.seg = (float*)work; a.part = a.seg + n; a.gs = a.part + n;
  a.loss_item = loss_item; a.gx = gx_sh;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: return em_run_c<1>(a, s, stages);
    case 4: return em_run_c<4>(a, s, stages);
    case 9: return em_run_c<9>(a, s, stages);
    case 16: return em_run_c<16>(a, s, stages);
    default: return GFDN_E_UNSUPPORTED;
  }
}

extern "C" int gfdn_edc_loss_model_mixed(const float* x_sh, int ld, int B, int C, const float* A, int J, int start, int len,
                                         const float* amps, int S, const float* env, int ld_env, const float* maskw,
                                         float inv_count, float gscale, float* loss_item, float* gx_sh, void* work,
                                         void* stream) {
  return gfdn_edc_loss_model_mixed_stages(x_sh, ld, B, C, A, J, start, len, amps, S, env, ld_env, maskw, inv_count, gscale,
                                          loss_item, gx_sh, work, 3, stream);
}
