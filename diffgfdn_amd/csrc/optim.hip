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
// advance != 0 (single-block launches only): the block writes t back after a barrier that every
// thread reaches after its read; multi-block launches leave it to the 1-thread kernel below
// (no intra-launch race).
__global__ __launch_bounds__(1024) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              const unsigned char* __restrict__ seg,
                                              const float* __restrict__ lr_seg,
                                              float* __restrict__ step_count, int n, float b1,
                                              float b2, float eps, int advance,
                                              unsigned int* __restrict__ block_counter,
                                              float* __restrict__ mirror) {
  const float t = step_count[0] + 1.0f;
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    p[i] = adam_elem(p[i], g[i], mi, vi, lr_seg[seg[i]], bc1, bc2_sqrt, b1, b2, eps);      // (lerp, as torch does)
    m[i] = mi;
    v[i] = vi;
  }
  if (advance) {
    __syncthreads();
    if (threadIdx.x == 0) {
      step_count[0] = t;
      if (mirror) mirror[0] = t;
    }
  } else if (block_counter) {
    // multi-block launch: the LAST workgroup to finish writes t back -- every workgroup has read the counter by the
    // time it reports in -- and re-arms the block counter for the next launch (no second kernel for the increment)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(block_counter, 1u) == gridDim.x - 1) {
        step_count[0] = t;
        if (mirror) mirror[0] = t;
        block_counter[0] = 0u;
      }
    }
  }
}
__global__ void k_adam_advance(float* step_count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_count[0] += 1.0f;
}

static int adam_step_run(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                         const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                         float eps, unsigned int* block_counter, void* stream, float* mirror = nullptr) {
  if (!p || !g || !m || !v || !seg || !lr_seg || !step_count || n <= 0) return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 16 * 1024) {          // one block: update + counter advance in a single launch
    hipLaunchKernelGGL(k_adam, dim3(1), dim3(1024), 0, s, p, g, m, v, seg, lr_seg, step_count, n,
                       beta1, beta2, eps, 1, (unsigned int*)nullptr, mirror);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  // (eight elements per thread: a workgroup ends with a fence and an atomic on ONE counter -- 800 workgroups of one
  // element per thread took 31 us for 200 000 parameters, most of it the 800 fences and serialised counter adds)
  int blocks = (n + 2047) / 2048;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, p, g, m, v, seg, lr_seg, step_count, n,
                     beta1, beta2, eps, 0, block_counter, mirror);
  GFDN_LAUNCH_CHECK();
  if (!block_counter) {
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(64), 0, s, step_count);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gfdn_adam_step(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                              const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                              float eps, void* stream) {
  return adam_step_run(p, g, m, v, seg, lr_seg, step_count, n, beta1, beta2, eps, nullptr, stream);
}

// block_counter: one zero-initialised uint32 owned by the optimiser (re-armed by every launch): the update and the
// advance of step_count are then ONE launch whatever n is
extern "C" int gfdn_adam_step_counted(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                                      const float* lr_seg, float* step_count, int n, float beta1, float beta2,
                                      float eps, unsigned int* block_counter, void* stream) {
  if (!block_counter) return GFDN_E_BADARG;
  return adam_step_run(p, g, m, v, seg, lr_seg, step_count, n, beta1, beta2, eps, block_counter, stream);
}

// The same with a second counter kept equal to step_count (mirror <- t).  An optimiser that sometimes steps two ranges of
// its buffer from two streams gives each range its own counter (a range reads and advances only its own: no ordering
// between the two launches is needed) and keeps them equal through this entry point when it steps the whole buffer.
extern "C" int gfdn_adam_step_mirrored(float* p, const float* g, float* m, float* v, const unsigned char* seg,
                                       const float* lr_seg, float* step_count, float* mirror, int n, float beta1,
                                       float beta2, float eps, unsigned int* block_counter, void* stream) {
  if (!block_counter || !mirror) return GFDN_E_BADARG;
  return adam_step_run(p, g, m, v, seg, lr_seg, step_count, n, beta1, beta2, eps, block_counter, stream, mirror);
}

// ------------------------------------------------------------------------------------------
// Receiver schedule of a graph-replayed epoch (trainer.py:373-379: the DataLoader fixes the batches of an epoch when
// the epoch starts).  The host uploads the epoch's batches once -- table (len, B) of dataset rows -- and every replayed
// step ends with this launch: idx <- table[pos mod len], pos <- pos + 1, on the device.  The next step then finds its
// receivers in the static index buffer without a host copy in front of it.
// state[0] = pos, state[1] = len (>= 1).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pick_rows(const long long* __restrict__ table, long long* __restrict__ state,
                                                   long long* __restrict__ idx, int B) {
  const long long len = state[1] > 0 ? state[1] : 1;
  const long long row = state[0] % len;
  for (int i = threadIdx.x; i < B; i += blockDim.x) idx[i] = table[row * B + i];
  __syncthreads();
  if (threadIdx.x == 0) state[0] = state[0] + 1;
}

extern "C" int gfdn_pick_rows(const long long* table, long long* state, long long* idx, int B, void* stream) {
  if (!table || !state || !idx || B <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_pick_rows, dim3(1), dim3(256), 0, (hipStream_t)stream, table, state, idx, B);
  GFDN_LAUNCH_CHECK();
  return 0;
}
