// Block-scan helpers of the EDC kernels (losses.hip) and the fused decay-loss kernel (decay.hip).
#pragma once
#include "common.h"

#define EDC_THREADS 256
#define EDC_V 4
#define EDC_S 4
#define EDC_SUB (EDC_THREADS * EDC_V)
#define EDC_TILE (EDC_SUB * EDC_S)
#define EDC_NSEG 8

__device__ __forceinline__ void block_scan_multi(float (&v)[EDC_S], float (&tot)[EDC_S],
                                                 float* lds /* >= 16*EDC_S floats */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < EDC_S; ++s) {
    float t;
    v[s] = wave_scan_incl(v[s], t);      // (DPP + readlane: VALU, no LDS crossbar -- common.h)
  }
  __syncthreads();
  if (lane == 63) {
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) lds[s * 16 + w] = v[s];
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
#pragma unroll
  for (int s = 0; s < EDC_S; ++s) {
    float pre = 0.f, t = 0.f;
    for (int i = 0; i < nw; ++i) {
      float q = lds[s * 16 + i];
      if (i < w) pre += q;
      t += q;
    }
    v[s] += pre;
    tot[s] = t;
  }
}

// running sums over j = 0..len-1 (index order) starting from `carry`; fn(j, inclusive_sum_j, value_j)
// 16-byte accesses at 4-byte alignment (rows of odd length start on 8-byte boundaries only): four consecutive
// pair samples / four consecutive floats per thread in two / one memory instructions
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void ld4_f2(const float2* p, float2 (&o)[4]) {
  const f4u a = *(const f4u*)p, b = *(const f4u*)(p + 2);
  o[0] = make_float2(a.x, a.y); o[1] = make_float2(a.z, a.w);
  o[2] = make_float2(b.x, b.y); o[3] = make_float2(b.z, b.w);
}
__device__ __forceinline__ void st4_f2(float2* p, const float2 (&v)[4]) {
  f4u a, b;
  a.x = v[0].x; a.y = v[0].y; a.z = v[1].x; a.w = v[1].y;
  b.x = v[2].x; b.y = v[2].y; b.z = v[3].x; b.w = v[3].y;
  *(f4u*)p = a;
  *(f4u*)(p + 2) = b;
}
__device__ __forceinline__ void ld4_f(const float* p, float (&o)[4]) {
  const f4u a = *(const f4u*)p;
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
__device__ __forceinline__ void st4_f(float* p, const float (&v)[4]) {
  f4u a;
  a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3];
  *(f4u*)p = a;
}

