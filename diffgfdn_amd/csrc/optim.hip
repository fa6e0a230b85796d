// Fused Adam over one flat parameter buffer, for gfx950.
//
// reference: trainer.py:152-228 builds torch.optim.Adam with per-name learning-rate groups and
// steps it once per batch (:475).  On the MI355X path all parameters of the model are views into
// ONE flat fp32 buffer (and their gradients into another): the data-parallel all-reduce runs on the
// flat gradient buffer in place, and the update of every parameter of every group is a single
// launch instead of ~100 tiny foreach kernels.  Maths = torch.optim.Adam (no amsgrad, no weight
// decay):  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
#include "common.h"

// step_count: device float holding t-1 on entry; every thread uses t = step_count + 1.
// A separate 1-thread kernel advances the counter afterwards (no intra-launch race).
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              const unsigned char* __restrict__ seg,
                                              const float* __restrict__ lr_seg,
                                              const float* __restrict__ step_count, int n, float b1,
                                              float b2, float eps) {
  const float t = step_count[0] + 1.0f;
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // lerp, as torch does
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr_seg[seg[i]] / bc1) * (mi / denom);
  }
}
__global__ void k_adam_advance(float* step_count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_count[0] += 1.0f;
}

extern "C" int gfdn_adam_step(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                              const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                              float eps, void* stream) {
  if (!p || !g || !m || !v || !seg || !lr_seg || !step_count || n <= 0) return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, p, g, m, v, seg, lr_seg, step_count, n,
                     beta1, beta2, eps);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(64), 0, s, step_count);
  GFDN_LAUNCH_CHECK();
  return 0;
}
