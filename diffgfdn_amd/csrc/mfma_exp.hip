// BASELINE.json configs[4]: "fp32 vs bf16 feedback-matmul on MFMA" -- the measurement kernel.
//
// The reference forms the resolvent P_k = (D_k Gamma^-1 - A)^-1 as a dense (K, N, N) tensor (feedback_loop.py:389-391)
// and contracts it with the receiver-dependent output gains and the input gains (model.py:615-619:
// einsum('knb,knm->kmb') over (B, N, K), then the product with b).  That "feedback multiply promoted to a batched
// dense contraction" is, per bin, D_k = C (B x N) . P_k (N x N) followed by H[b][k] = sum_m D_k[b][m] b_m -- the one
// GEMM-shaped formulation of this path.  This file runs exactly that contraction for N = B = 32 on the matrix cores,
// one wavefront per bin, in two precisions:
//   bf16 : v_mfma_f32_32x32x16_bf16, operands rounded to bfloat16 (round to nearest even), float32 accumulate;
//   f32  : v_mfma_f32_32x32x2_f32 (exact float32 products).
// The product path never forms P (it SOLVES per bin, csrc/solve.hip, or evaluates the block transfer functions,
// csrc/blocktf.hip); tools/mfma_experiment.py times this kernel against that path and reports the deviation of H
// from a complex128 evaluation, next to the 1e-4 bar of the north star.
#include "common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ __bf16 to_bf16(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);          // round to nearest even (finite inputs)
  const unsigned short h = (unsigned short)(u >> 16);
  return __builtin_bit_cast(__bf16, h);
}

// P (K, 32, 32) complex64 row-major [k][n][m]; C (32, 32) float32 [b][n]; bvec (32); H (32, K) complex64.
// Lane l = (r = l & 31, h = l >> 5).  32x32x16 bf16: A fragment = C[row r][n = 8h + j (+16)], B fragment =
// P_k[n = 8h + j (+16)][col r]; accumulator register q holds row (q & 3) + 8 (q >> 2) + 4 h, column r.
template <bool BF16>
__global__ __launch_bounds__(256) void k_exp_contract(const float2* __restrict__ P, int K,
                                                      const float* __restrict__ C, const float* __restrict__ bvec,
                                                      float2* __restrict__ H) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const float2* Pk = P + (size_t)k * 1024;
  f32x16 dre, dim;
#pragma unroll
  for (int q = 0; q < 16; ++q) dre[q] = dim[q] = 0.f;
  if (BF16) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 a, bre, bim;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = 16 * half + 8 * h + j;
        a[j] = to_bf16(C[r * 32 + n]);
        const float2 p = Pk[n * 32 + r];
        bre[j] = to_bf16(p.x);
        bim[j] = to_bf16(p.y);
      }
      dre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bre, dre, 0, 0, 0);
      dim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bim, dim, 0, 0, 0);
    }
  } else {
    // 32x32x2 f32: A = C[row r][n = 2 s + h], B = P_k[n = 2 s + h][col r], s = 0..15
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int n = 2 * s + h;
      const float a = C[r * 32 + n];
      const float2 p = Pk[n * 32 + r];
      dre = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p.x, dre, 0, 0, 0);
      dim = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p.y, dim, 0, 0, 0);
    }
  }
  // H[b][k] = sum_m D[b][m] b_m: column m = r lives on the lane -> reduce over the 32 lanes of each half
  const float bm = bvec[r];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float vr = dre[q] * bm, vi = dim[q] * bm;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      vr += __shfl_xor(vr, off, 64);
      vi += __shfl_xor(vi, off, 64);
    }
    if (r == 0) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
      H[(size_t)row * K + k] = make_float2(vr, vi);
    }
  }
}

extern "C" int gfdn_exp_contract_mfma(const float* P_c64, int K, const float* C, const float* bvec, int use_bf16,
                                      float* H_c64, void* stream) {
  if (!P_c64 || !C || !bvec || !H_c64 || K <= 0) return GFDN_E_BADARG;
  dim3 grid((K + 3) / 4), block(256);
  if (use_bf16)
    hipLaunchKernelGGL(k_exp_contract<true>, grid, block, 0, (hipStream_t)stream, (const float2*)P_c64, K, C, bvec,
                       (float2*)H_c64);
  else
    hipLaunchKernelGGL(k_exp_contract<false>, grid, block, 0, (hipStream_t)stream, (const float2*)P_c64, K, C, bvec,
                       (float2*)H_c64);
  GFDN_LAUNCH_CHECK();
  return 0;
}
