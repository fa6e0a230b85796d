// Shared device helpers for the gfx950 kernels of the DiffGFDN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stddef.h>
#include <stdint.h>

#include "../../include/diffgfdn_hip.h"

#define GFDN_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

#define F32_EPS 1.1920928955078125e-07f
#define TEN_OVER_LN10 4.3429448190325175f

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cinv(float2 a) {
  float d = 1.0f / (a.x * a.x + a.y * a.y);
  return make_float2(a.x * d, -a.y * d);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// 10*log10(|x| + eps_f32) clipped below at -200 dB  (reference utils.py:32-40)
__device__ __forceinline__ float db_pow(float x) {
  float y = 10.0f * log10f(fabsf(x) + F32_EPS);
  return fmaxf(y, -200.0f);
}

// One element of the Adam update (torch.optim.Adam without amsgrad / weight decay; reference trainer.py:475) with every
// rounding pinned -- two kernels step ranges of one flat buffer (optim.hip k_adam, blocktf.hip k_tf_tail) and must agree
// to the bit whatever the compiler would contract:  m <- m + (g - m)(1 - b1);  v <- v b2 + g g (1 - b2);
// p <- p - (lr / bc1) m / (sqrt(v) / bc2_sqrt + eps).  Returns the new parameter.
__device__ __forceinline__ float adam_elem(float p, float g, float& m, float& v, float lr, float bc1, float bc2_sqrt,
                                           float b1, float b2, float eps) {
  m = __fmaf_rn(g - m, 1.0f - b1, m);
  v = __fmaf_rn(__fmul_rn(g, g), 1.0f - b2, __fmul_rn(v, b2));
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
  return __fsub_rn(p, __fmul_rn(__fdiv_rn(lr, bc1), __fdiv_rn(m, denom)));
}

// The flat optimiser buffers (optim.hip) as a kernel argument: the tails of the bank's step (blocktf.hip k_tf_tail,
// blocktf8.hip k_tf8_tail) step the blocks' own entries of M, b, c themselves, straight behind their gradients.
struct TfAdam {
  float* p;                    // flat parameters
  float* m;
  float* v;
  const unsigned char* seg;
  const float* lr_seg;
  float* step_count;
  unsigned int* block_counter;
  int offM, offb, offc;
  float b1, b2, eps;
};

__device__ __forceinline__ float tf_adam_elem(const TfAdam& ad, int i, float gi, float bc1, float bc2_sqrt) {
  float mi = ad.m[i], vi = ad.v[i];
  const float pn = adam_elem(ad.p[i], gi, mi, vi, ad.lr_seg[ad.seg[i]], bc1, bc2_sqrt, ad.b1, ad.b2, ad.eps);
  ad.m[i] = mi;
  ad.v[i] = vi;
  ad.p[i] = pn;
  return pn;
}


// sum over the 64 lanes of a wavefront
// Cross-lane sums on the VALU (DPP inside the rows of 16 lanes, v_permlane16_swap / v_permlane32_swap of gfx950 between
// rows and half-waves).  __shfl_* compile to ds_bpermute_b32, which issues on the LDS pipe with its latency: a butterfly of
// six of them per value made the reductions of several kernels LDS-bound.
template <int CTRL>
__device__ __forceinline__ float dpp_pair_sum(float v) {          // v + v of the lane CTRL pairs this one with
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float xor16_sum(float x) {             // x(l) + x(l ^ 16)
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {             // x(l) + x(l ^ 32)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 64 lanes, the same bits in every lane (every level adds two values that its lanes share).  ALL 64 lanes
// must be active: a lane whose partner is switched off keeps its own value in the swaps.
__device__ __forceinline__ float wave_sum_full(float v) {
  v = dpp_pair_sum<0x128>(v);           // row_ror:8
  v = dpp_pair_sum<0x124>(v);           // row_ror:4
  v = dpp_pair_sum<0x122>(v);           // row_ror:2
  v = dpp_pair_sum<0x121>(v);           // row_ror:1
  return xor32_sum(xor16_sum(v));
}
// the general form (partial waves, divergent callers: a switched-off lane counts as zero)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Prefix / suffix sums over the 64 lanes of a wave on the VALU: DPP row shifts inside the rows of 16 lanes, the row
// totals through v_readlane -- no LDS crossbar (ds_bpermute, which __shfl_* compile to, issues on the LDS pipe: a kernel
// that scans a dozen values per thread is bound by it).  Fixed summation order.
template <int CTRL>
__device__ __forceinline__ float dpp_add_shifted(float v) {       // v + (lane shifted by CTRL inside its row, 0 where none)
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// inclusive prefix sum (lane i: v_0 + ... + v_i); total = sum of the wave
__device__ __forceinline__ float wave_scan_incl(float v, float& total) {
  v = dpp_add_shifted<0x111>(v);        // row_shr:1
  v = dpp_add_shifted<0x112>(v);        // row_shr:2
  v = dpp_add_shifted<0x114>(v);        // row_shr:4
  v = dpp_add_shifted<0x118>(v);        // row_shr:8
  const float t0 = readlane_f(v, 15), t1 = readlane_f(v, 31), t2 = readlane_f(v, 47), t3 = readlane_f(v, 63);
  const int row = (threadIdx.x & 63) >> 4;
  const float add = (row >= 1 ? t0 : 0.f) + (row >= 2 ? t1 : 0.f) + (row >= 3 ? t2 : 0.f);
  total = ((t0 + t1) + t2) + t3;
  return v + add;
}
// inclusive suffix sum (lane i: v_i + ... + v_63)
__device__ __forceinline__ float wave_scan_incl_rev(float v, float& total) {
  v = dpp_add_shifted<0x101>(v);        // row_shl:1
  v = dpp_add_shifted<0x102>(v);
  v = dpp_add_shifted<0x104>(v);
  v = dpp_add_shifted<0x108>(v);
  const float t0 = readlane_f(v, 0), t1 = readlane_f(v, 16), t2 = readlane_f(v, 32), t3 = readlane_f(v, 48);
  const int row = (threadIdx.x & 63) >> 4;
  const float add = (row <= 2 ? t3 : 0.f) + (row <= 1 ? t2 : 0.f) + (row <= 0 ? t1 : 0.f);
  total = ((t3 + t2) + t1) + t0;
  return v + add;
}
__device__ __forceinline__ float wave_sum_valu(float v) {         // the wave's sum in every lane
  float total;
  wave_scan_incl(v, total);
  return total;
}

// deterministic block sum (blockDim.x multiple of 64, <= 1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* lds /* >= 16 floats */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += lds[i];
  return s;
}

// dynamic LDS above the 64 KB default needs an explicit opt-in per kernel
// The largest size already granted per kernel is remembered (a process-wide cache, the library's only global
// state, guarded by a mutex: launches may come from several host threads) so that steady-state launches -- and
// launches under HIP graph capture -- make no runtime API call.
static inline int ensure_dyn_lds_ptr(const void* fn, size_t bytes) {
  if (bytes <= 48 * 1024) return 0;
  // (a kernel that does not fit in the table would call hipFuncSetAttribute on EVERY launch -- also under graph capture,
  // where the runtime refuses runtime API calls: the table is per translation unit and sized far above the number of
  // kernels with more than 48 KB of dynamic LDS in any of them)
  constexpr int SLOTS = 512;
  static const void* fns[SLOTS];
  static size_t granted[SLOTS];
  static int count = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int slot = -1;
  for (int i = 0; i < count; ++i)
    if (fns[i] == fn) { slot = i; break; }
  if (slot >= 0 && bytes <= granted[slot]) return 0;
  int rc = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc != 0) return rc;
  if (slot < 0 && count < SLOTS) { slot = count++; fns[slot] = fn; }
  if (slot >= 0) granted[slot] = bytes;
  return 0;
}
template <typename K>
static inline int ensure_dyn_lds(K kernel, size_t bytes) {
  return ensure_dyn_lds_ptr(reinterpret_cast<const void*>(kernel), bytes);
}

static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int next_pow2(int v) { return 1 << ilog2(v); }
