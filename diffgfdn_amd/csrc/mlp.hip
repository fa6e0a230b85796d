// Receiver-position -> group-gain network in two launches, for gfx950.
//
// reference: gain_filters.py:497-534 (Gains_from_MLP.forward), dnn.py:89-126 (SinusoidalEncoding),
// dnn.py:331-400 (MLP = [Linear, LayerNorm, ReLU] x nl, then Linear), dnn.py:21-36 (ScaledSigmoid).
// The network is tiny (batch 32; 120 -> 16 x 6 -> 4 on the 500 Hz band, at most 120 -> 128 x 4 -> G on the
// top bands), so on the GPU it is pure launch overhead: ~30 launches forward and ~60 backward in
// torch.  Here one workgroup per receiver runs the whole stack with the activations in LDS:
//   forward : encoding (float64 sin/cos of f32(freq*pi)*x, as the reference), layers, sigmoid scaling;
//             saves xhat (normalised pre-affine activations) and rstd per layer for the backward
//   backward: per-receiver parameter-gradient partials, summed over receivers by a fixed-order pass
// Parameters arrive packed in ONE flat float buffer in named_parameters() order:
//   [W0 (H x in) | b0 | gamma0 | beta0 | W1 (H x H) | b1 | gamma1 | beta1 | ... | Wout (G x H) | bout].
#include "common.h"

#define MLP_T 256
#define MLP_STAGE_MAX 7168     // floats: parameter sets up to 28 KB are staged in LDS (twice in the backward)
#define LN_EPS 1e-5f

struct MlpDims {
  int B, F, in_dim, H, nl, G;   // F fourier features, in_dim = 6 F, nl = 1 + hidden layers
  float lo, hi;                 // ScaledSigmoid limits
  const long long* rows;        // item b encodes pos[rows[b]] (NULL: pos[b])
  int stage;                    // 1: the packed parameters (+ the receiver's saved activations) fit in LDS
  int Bper;                     // items per band: item b uses the parameter set b / Bper (w is (bands, P))
  int gparts;                   // > 0: ``ggains`` holds (B G, gparts) partial rows (gfdn_tf_gain_grad without its row sums)
  // (wave-per-receiver backward) column scales s[band G + g] (normalize's scale of the band's groups, trainer.py:317-332,
  // folded into the receiver gains: gfdn_tf_energy_gains): ``ggains`` holds dL/d(gains s), multiplied by s first
  const float* colscale;
};

__device__ __forceinline__ size_t mlp_layer_off(const MlpDims& d, int l) {
  // offset of W_l in the flat parameter buffer
  if (l == 0) return 0;
  size_t first = (size_t)d.H * d.in_dim + 3 * (size_t)d.H;
  return first + (size_t)(l - 1) * ((size_t)d.H * d.H + 3 * (size_t)d.H);
}
__host__ __device__ static inline size_t mlp_param_count(const MlpDims& d) {
  return (size_t)d.H * d.in_dim + 3 * (size_t)d.H + (size_t)(d.nl - 1) * ((size_t)d.H * d.H + 3 * (size_t)d.H) +
         (size_t)d.G * d.H + d.G;
}

extern __shared__ float mlp_lds[];

__device__ __forceinline__ void mlp_encode(const MlpDims& d, int b, const double* __restrict__ pos,
                                           const float* __restrict__ freq_pi, float* a) {
  // dnn.py:112-124: per frequency k: [sin(f_k pi x) (3) | cos(f_k pi x) (3)], stored as float32
  const double* p = pos + (d.rows ? (size_t)d.rows[b] : (size_t)b) * 3;
  for (int e = threadIdx.x; e < d.in_dim; e += blockDim.x) {
    const int k = e / 6, r = e - 6 * k;
    const double arg = (double)freq_pi[k] * p[r % 3];
    a[e] = (float)(r < 3 ? sin(arg) : cos(arg));
  }
}

// Parameters that stay in memory (a band of 128-neuron layers: 266 KB): ONE round of loads touches every cache line of the
// set up front.  A step starts with the parameters cold (the optimiser rewrote them on other CUs), and the layer chain then
// paid an HBM miss per dependent access -- rows, bias, gamma, beta of every layer: 83 us for the launch from cold caches
// against 32 warm.  The value is kept alive through a condition that never holds.
__device__ __forceinline__ float mlp_touch(const float* __restrict__ w, int P) {
  float t = 0.f;
  for (int p = threadIdx.x * 32; p < P; p += (int)blockDim.x * 32) t += w[p];
  return t;
}

// One receiver (item b) by a whole workgroup.  w: the band's packed parameters; xout / rout: where this receiver's saved
// activations (nl, H) / (nl) go.
__device__ __forceinline__ void mlp_fwd_block(const MlpDims& d, int b, const double* __restrict__ pos,
                                              const float* __restrict__ freq_pi, const float* __restrict__ w,
                                              float* __restrict__ gains, float* __restrict__ xout,
                                              float* __restrict__ rout) {
  // (a chain of dependent steps launched beside kernels that fill every SIMD: see k_mlp_fwd_waves)
  __builtin_amdgcn_s_setprio(3);
  const int amax = d.in_dim > d.H ? d.in_dim : d.H;
  float* a = mlp_lds;            // current activations
  float* h = a + amax;           // pre-norm outputs
  float* red = h + d.H;          // 16 floats
  const int H = d.H;
  // small networks: ONE round of global loads brings every parameter into LDS; the layer chain then
  // never waits on memory again (it was one dependent global-load latency per layer)
  // Likewise the saved activations go to LDS first and to memory once at the end: every
  // __syncthreads() waits for ALL outstanding global stores of the wave (vmcnt covers stores on
  // gfx9), so a store inside the layer loop costs a full write round trip per barrier.
  float* xs = red + 16;          // (parameters in memory: the saved activations still wait in LDS -- a global store inside the
  float* rsv = xs + d.nl * H;    // layer loop is a write round trip at every barrier behind it)
  if (!d.stage) {
    const float t = mlp_touch(w, (int)mlp_param_count(d));
    if (t == 1.2345678e38f) rout[0] = t;
  }
  if (d.stage) {
    float* wl = red + 16;
    const int P = (int)mlp_param_count(d);
    for (int p = threadIdx.x; p < P; p += blockDim.x) wl[p] = w[p];
    w = wl;
    xs = wl + P;
    rsv = xs + d.nl * H;
  }
  mlp_encode(d, b, pos, freq_pi, a);
  __syncthreads();
  for (int l = 0; l < d.nl; ++l) {
    const int n_in = l == 0 ? d.in_dim : H;
    const float* W = w + mlp_layer_off(d, l);
    const float* bias = W + (size_t)H * n_in;
    const float* gamma = bias + H;
    const float* beta = gamma + H;
    float hv = 0.f;
    const int j = threadIdx.x;
    if (!d.stage) {
      // parameters in memory (layers of 128 neurons: 266 KB per band): the waves take the neurons, the LANES the inputs --
      // one coalesced row read per neuron (a thread per neuron walks its own row: every load touches H cache lines)
      // (eight neurons per round: their row loads are in flight together -- one neuron at a time was one L2 round trip
      // per neuron, 16 us per layer -- and the wave sums run on the VALU, not through the LDS crossbar)
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
      for (int j0 = wv * 8; j0 < H; j0 += nw * 8) {
        float part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          part[u] = 0.f;
          if (j0 + u < H)
            for (int i = lane; i < n_in; i += 64) part[u] += W[(size_t)(j0 + u) * n_in + i] * a[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float sum = wave_sum_full(part[u]);
          if (lane == 0 && j0 + u < H) h[j0 + u] = sum + bias[j0 + u];
        }
      }
      __syncthreads();
      if (j < H) hv = h[j];
    } else if (j < H) {
      hv = bias[j];
      for (int i = 0; i < n_in; ++i) hv += W[(size_t)j * n_in + i] * a[i];
    }
    const float mean = block_sum(j < H ? hv : 0.f, red) / (float)H;
    const float dv = j < H ? hv - mean : 0.f;
    const float var = block_sum(dv * dv, red) / (float)H;
    const float rs = rsqrtf(var + LN_EPS);
    __syncthreads();
    if (j < H) {
      const float xh = dv * rs;
      xs[l * H + j] = xh;
      a[j] = fmaxf(xh * gamma[j] + beta[j], 0.f);
    }
    if (j == 0) rsv[l] = rs;
    __syncthreads();
  }
  const float* Wout = w + mlp_layer_off(d, d.nl);
  const float* bout = Wout + (size_t)d.G * H;
  if (!d.stage) {
    // (parameters in memory: a wave per output, the lanes over its row -- one thread walking a row of 128 weights was 128
    // dependent loads, 60 us of the launch)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    for (int g = wv; g < d.G; g += nw) {
      float part = 0.f;
      for (int i = lane; i < H; i += 64) part += Wout[(size_t)g * H + i] * a[i];
      const float raw = wave_sum_full(part) + bout[g];
      if (lane == 0)
        gains[(size_t)b * d.G + g] = d.hi > d.lo ? d.lo + (d.hi - d.lo) * (1.0f / (1.0f + expf(-raw))) : raw;
    }
  } else {
    for (int g = threadIdx.x; g < d.G; g += blockDim.x) {
      float raw = bout[g];
      for (int i = 0; i < H; ++i) raw += Wout[(size_t)g * H + i] * a[i];
      // hi <= lo: no output activation (the SVF network hands its raw outputs to the SVF -> biquad map)
      gains[(size_t)b * d.G + g] = d.hi > d.lo ? d.lo + (d.hi - d.lo) * (1.0f / (1.0f + expf(-raw))) : raw;
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < d.nl * H; p += blockDim.x) xout[p] = xs[p];
  for (int p = threadIdx.x; p < d.nl; p += blockDim.x) rout[p] = rsv[p];
}

__global__ __launch_bounds__(MLP_T) void k_mlp_fwd(MlpDims d, const double* __restrict__ pos,
                                                   const float* __restrict__ freq_pi,
                                                   const float* __restrict__ w,
                                                   float* __restrict__ gains,     // (B, G)
                                                   float* __restrict__ xhat,      // (B, nl, H)
                                                   float* __restrict__ rstd) {    // (B, nl)
  const int b = blockIdx.x;
  mlp_fwd_block(d, b, pos, freq_pi, w + (size_t)(b / d.Bper) * mlp_param_count(d), gains,
                xhat + (size_t)b * d.nl * d.H, rstd + (size_t)b * d.nl);
}

// ------------------------------------------------------------------------------------------
// Layers of at most 64 neurons (the north-star network: 120 -> 16 x 6 -> 4): ONE WAVEFRONT per receiver, MLP_RB
// receivers per workgroup, the packed parameters staged in LDS once per workgroup.  A receiver's layer chain is a
// serial string of 16-wide steps: on a 256-thread workgroup per receiver every step pays block barriers and two block
// reductions; here they are wave-level (the LDS traffic of one wave executes in order).  Same arithmetic, same order.
// ------------------------------------------------------------------------------------------
#define MLP_RB 8
// LDS hand-over between the lanes of ONE wavefront: its LDS operations execute in order, so all that is needed is that
// the writes have been issued (lgkmcnt) and that the compiler keeps the order.  Deliberately NOT a fence: a release
// fence also waits for the wave's outstanding GLOBAL stores (vmcnt counts stores on gfx9) -- a write round trip per use.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// MLP_RB receivers b0 .. b0 + MLP_RB - 1 of ONE band, a wave each.  w: the band's packed parameters; xhat_b0 / rstd_b0: where
// receiver b0's saved activations go (the others follow at (nl, H) / (nl) each).
__device__ __forceinline__ void mlp_fwd_waves(const MlpDims& d, int b0, const double* __restrict__ pos,
                                              const float* __restrict__ freq_pi, const float* __restrict__ w,
                                              float* __restrict__ gains, float* __restrict__ xhat_b0,
                                              float* __restrict__ rstd_b0) {
  // a few workgroups of dependent 16-wide steps, launched beside kernels that fill every SIMD with VALU work (the
  // energy pass; the output-stage adjoint): without priority the chain advances at the SIMD's round-robin share
  __builtin_amdgcn_s_setprio(3);
  const int H = d.H, nl = d.nl, P = (int)mlp_param_count(d);
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int b = b0 + r;
  const int amax = d.in_dim > H ? d.in_dim : H;
  float* wl = mlp_lds;
  float* a = wl + P + (size_t)r * (amax + nl * H + nl);      // this receiver's current activations
  float* xs = a + amax;                          // saved activations: collected here, written once at the end
  float* rsv = xs + nl * H;
  for (int p = threadIdx.x; p < P; p += blockDim.x) wl[p] = w[p];
  {
    const double* pp = pos + (d.rows ? (size_t)d.rows[b] : (size_t)b) * 3;
    for (int e = lane; e < d.in_dim; e += 64) {
      const int k = e / 6, q = e - 6 * k;
      const double arg = (double)freq_pi[k] * pp[q % 3];
      a[e] = (float)(q < 3 ? sin(arg) : cos(arg));
    }
  }
  __syncthreads();
  for (int l = 0; l < nl; ++l) {
    const int n_in = l == 0 ? d.in_dim : H;
    const float* W = wl + mlp_layer_off(d, l);
    const float* bias = W + (size_t)H * n_in;
    const float* gamma = bias + H;
    const float* beta = gamma + H;
    float hv = 0.f;
    if (lane < H) {
      hv = bias[lane];
      for (int i = 0; i < n_in; ++i) hv += W[(size_t)lane * n_in + i] * a[i];
    }
    const float mean = wave_sum(lane < H ? hv : 0.f) / (float)H;
    const float dv = lane < H ? hv - mean : 0.f;
    const float var = wave_sum(dv * dv) / (float)H;
    const float rs = rsqrtf(var + LN_EPS);
    wave_lds_sync();                             // every lane has read a[] of this layer
    if (lane < H) {
      const float xh = dv * rs;
      xs[l * H + lane] = xh;
      a[lane] = fmaxf(xh * gamma[lane] + beta[lane], 0.f);
    }
    if (lane == 0) rsv[l] = rs;
    wave_lds_sync();
  }
  const float* Wout = wl + mlp_layer_off(d, nl);
  const float* bout = Wout + (size_t)d.G * H;
  if (lane < d.G) {
    float raw = bout[lane];
    for (int i = 0; i < H; ++i) raw += Wout[(size_t)lane * H + i] * a[i];
    gains[(size_t)b * d.G + lane] = d.hi > d.lo ? d.lo + (d.hi - d.lo) * (1.0f / (1.0f + expf(-raw))) : raw;
  }
  for (int p = lane; p < nl * H; p += 64) xhat_b0[(size_t)r * nl * H + p] = xs[p];
  if (lane < nl) rstd_b0[(size_t)r * nl + lane] = rsv[lane];
}

__global__ __launch_bounds__(64 * MLP_RB) void k_mlp_fwd_waves(MlpDims d, const double* __restrict__ pos,
                                                               const float* __restrict__ freq_pi,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ gains,     // (B, G)
                                                               float* __restrict__ xhat,      // (B, nl, H)
                                                               float* __restrict__ rstd) {    // (B, nl)
  const int b0 = blockIdx.x * MLP_RB;            // (all receivers of a workgroup belong to one band)
  mlp_fwd_waves(d, b0, pos, freq_pi, w + (size_t)(b0 / d.Bper) * mlp_param_count(d), gains,
                xhat + (size_t)b0 * d.nl * d.H, rstd + (size_t)b0 * d.nl);
}

// One receiver (item b) by a whole workgroup.  w: the band's packed parameters; xhat / rstd: this receiver's saved
// activations; gout: its row of parameter-gradient partials (P).  ggains: (B, G) summed gradients, or with d.gparts > 0 the
// (B G, gparts) partial rows, summed here by one thread per group in column order (fixed, but not the wave form's order).
__device__ __forceinline__ void mlp_bwd_block(const MlpDims& d, int b, const double* __restrict__ pos,
                                              const float* __restrict__ freq_pi, const float* __restrict__ w,
                                              const float* __restrict__ gains, const float* __restrict__ xhat,
                                              const float* __restrict__ rstd, const float* __restrict__ ggains,
                                              float* __restrict__ gout) {
  __builtin_amdgcn_s_setprio(3);
  const int amax = d.in_dim > d.H ? d.in_dim : d.H;
  float* aprev = mlp_lds;        // activations entering the current layer
  float* da = aprev + amax;      // gradient w.r.t. the current layer's output (H)
  float* dh = da + d.H;          // gradient w.r.t. the linear output (H)
  float* draw = dh + d.H;        // (G)
  float* red = draw + d.G;       // 16
  const int H = d.H, G = d.G;
  const size_t P = mlp_param_count(d);
  float* gp = gout;
  if (!d.stage) {
    const float t = mlp_touch(w, (int)P);
    if (t == 1.2345678e38f) gout[0] = t;
  }
  if (d.stage) {                 // parameters and this receiver's saved activations -> LDS, one load round
    float* wl = red + 16;
    float* xl = wl + P;
    float* rl = xl + d.nl * H;
    for (int p = threadIdx.x; p < (int)P; p += blockDim.x) wl[p] = w[p];
    for (int p = threadIdx.x; p < d.nl * H; p += blockDim.x) xl[p] = xhat[p];
    for (int p = threadIdx.x; p < d.nl; p += blockDim.x) rl[p] = rstd[p];
    w = wl; xhat = xl; rstd = rl;
    gp = rl + d.nl;              // gradient partials are collected in LDS and written once at the end
    __syncthreads();
  }
  // output layer
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float gg;
    if (d.gparts > 0) {
      const float* row = ggains + ((size_t)b * G + g) * d.gparts;
      gg = 0.f;
      for (int p = 0; p < d.gparts; ++p) gg += row[p];
    } else {
      gg = ggains[(size_t)b * G + g];
    }
    if (d.colscale) gg *= d.colscale[(b / d.Bper) * G + g];
    if (d.hi > d.lo) {
      const float sg = (gains[(size_t)b * G + g] - d.lo) / (d.hi - d.lo);
      draw[g] = gg * (d.hi - d.lo) * sg * (1.0f - sg);
    } else {
      draw[g] = gg;
    }
  }
  {
    const int l = d.nl - 1;
    const float* W = w + mlp_layer_off(d, l);
    const int n_in = l == 0 ? d.in_dim : H;
    const float* gamma = W + (size_t)H * n_in + H;
    const float* beta = gamma + H;
    for (int i = threadIdx.x; i < H; i += blockDim.x)
      aprev[i] = fmaxf(xhat[(size_t)l * H + i] * gamma[i] + beta[i], 0.f);
  }
  __syncthreads();
  const size_t offOut = mlp_layer_off(d, d.nl);
  const float* Wout = w + offOut;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float acc = 0.f;
    for (int g = 0; g < G; ++g) {
      gp[offOut + (size_t)g * H + i] = draw[g] * aprev[i];
      acc += Wout[(size_t)g * H + i] * draw[g];
    }
    da[i] = acc;
  }
  for (int g = threadIdx.x; g < G; g += blockDim.x) gp[offOut + (size_t)G * H + g] = draw[g];
  __syncthreads();
  // hidden blocks, last to first
  for (int l = d.nl - 1; l >= 0; --l) {
    const int n_in = l == 0 ? d.in_dim : H;
    const size_t offW = mlp_layer_off(d, l);
    const float* W = w + offW;
    const float* gamma = W + (size_t)H * n_in + H;
    const float* beta = gamma + H;
    const size_t offb = offW + (size_t)H * n_in, offg = offb + H, offbeta = offg + H;
    const int j = threadIdx.x;
    float xh = 0.f, dxh = 0.f;
    if (j < H) {
      xh = xhat[(size_t)l * H + j];
      const float y = xh * gamma[j] + beta[j];
      const float dy = y > 0.f ? da[j] : 0.f;
      gp[offg + j] = dy * xh;
      gp[offbeta + j] = dy;
      dxh = dy * gamma[j];
    }
    const float m1 = block_sum(dxh, red) / (float)H;
    const float m2 = block_sum(dxh * xh, red) / (float)H;
    const float rs = rstd[l];
    __syncthreads();
    if (j < H) {
      const float v = rs * (dxh - m1 - xh * m2);
      dh[j] = v;
      gp[offb + j] = v;
    }
    // activations that entered this layer
    if (l == 0) {
      mlp_encode(d, b, pos, freq_pi, aprev);
    } else {
      const float* Wp = w + mlp_layer_off(d, l - 1);
      const int n_in_p = (l - 1) == 0 ? d.in_dim : H;
      const float* gam_p = Wp + (size_t)H * n_in_p + H;
      const float* bet_p = gam_p + H;
      for (int i = threadIdx.x; i < H; i += blockDim.x)
        aprev[i] = fmaxf(xhat[(size_t)(l - 1) * H + i] * gam_p[i] + bet_p[i], 0.f);
    }
    __syncthreads();
    if (!d.stage && (int)blockDim.x >= 2 * n_in) {
      // parameters in memory (128-neuron layers): the workgroup's threads as (neuron group, input) -- with one thread per
      // input walking all H neurons, 128 threads of 512 made 128 dependent load / store round trips per layer (the launch
      // took 219 us); here every thread takes H / groups neurons, eight rows in flight, and the partial sums of dL/da
      // meet in LDS (``red2``: groups x n_in floats behind the vectors, which the launcher sizes)
      const int ngrp = (int)blockDim.x / n_in, grp = threadIdx.x / n_in, i = threadIdx.x - grp * n_in;
      float* red2 = red + 16;
      float acc = 0.f;
      if (grp < ngrp) {
        const float ai = aprev[i];
        for (int j0 = grp * 8; j0 < H; j0 += ngrp * 8) {
          float wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = j0 + u < H ? W[(size_t)(j0 + u) * n_in + i] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (j0 + u < H) {
              const float v = dh[j0 + u];
              gp[offW + (size_t)(j0 + u) * n_in + i] = v * ai;
              acc += wv[u] * v;
            }
          }
        }
        red2[grp * n_in + i] = acc;
      }
      __syncthreads();
      if (l > 0 && threadIdx.x < n_in) {
        float sum = 0.f;
        for (int q = 0; q < ngrp; ++q) sum += red2[q * n_in + threadIdx.x];
        da[threadIdx.x] = sum;
      }
    } else {
      for (int i = threadIdx.x; i < n_in; i += blockDim.x) {
        const float ai = aprev[i];
        float acc = 0.f;
        for (int jj = 0; jj < H; ++jj) {
          const float v = dh[jj];
          gp[offW + (size_t)jj * n_in + i] = v * ai;      // coalesced over i
          acc += W[(size_t)jj * n_in + i] * v;
        }
        if (l > 0) da[i] = acc;
      }
    }
    __syncthreads();
  }
  if (d.stage) {
    for (int p = threadIdx.x; p < (int)P; p += blockDim.x) gout[p] = gp[p];
  }
}

__global__ __launch_bounds__(MLP_T) void k_mlp_bwd(MlpDims d, const double* __restrict__ pos,
                                                   const float* __restrict__ freq_pi,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ gains,
                                                   const float* __restrict__ xhat,
                                                   const float* __restrict__ rstd,
                                                   const float* __restrict__ ggains,   // (B, G)
                                                   float* __restrict__ partial) {      // (B, P)
  const int b = blockIdx.x;
  const size_t P = mlp_param_count(d);
  mlp_bwd_block(d, b, pos, freq_pi, w + (size_t)(b / d.Bper) * P, gains, xhat + (size_t)b * d.nl * d.H,
                rstd + (size_t)b * d.nl, ggains, partial + (size_t)b * P);
}

// ------------------------------------------------------------------------------------------
// Backward, one wavefront per receiver as k_mlp_fwd_waves.  The chain only leaves per-layer vectors in LDS (activations
// entering the layer, dL/d(linear output), dL/d(affine output)); the parameter gradients are outer products of
// those, assembled at the end by the whole workgroup and summed over its receivers in a fixed order: one partial
// row per workgroup instead of one per receiver for the reduction pass.
// ------------------------------------------------------------------------------------------
__host__ __device__ static inline int mlp_rb_stride(const MlpDims& d) {
  // per receiver: A0 (in_dim) | A (nl, H) | XH (nl, H) | DH (nl, H) | DY (nl, H) | RS (nl) | DRAW (G)
  return d.in_dim + 4 * d.nl * d.H + d.nl + d.G;
}

// MLP_RB receivers b0 .. of ONE band, a wave each (w: the band's packed parameters; xhat_b0 / rstd_b0: receiver b0's saved
// activations; gout: the workgroup's row of parameter-gradient partials (P))
__device__ __forceinline__ void mlp_bwd_waves(const MlpDims& d, int b0, const double* __restrict__ pos,
                                              const float* __restrict__ freq_pi, const float* __restrict__ w,
                                              const float* __restrict__ gains, const float* __restrict__ xhat_b0,
                                              const float* __restrict__ rstd_b0, const float* __restrict__ ggains,
                                              float* __restrict__ gout) {
  __builtin_amdgcn_s_setprio(3);                 // (as k_mlp_fwd_waves)
  const int H = d.H, G = d.G, nl = d.nl, P = (int)mlp_param_count(d);
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int b = b0 + r;
  float* wl = mlp_lds;
  const int RS_ = mlp_rb_stride(d);
  float* mine = wl + P + (size_t)r * RS_;
  float* A0 = mine;
  float* A = A0 + d.in_dim;
  float* XH = A + nl * H;
  float* DH = XH + nl * H;
  float* DY = DH + nl * H;
  float* RS = DY + nl * H;
  float* DRAW = RS + nl;
  for (int p = threadIdx.x; p < P; p += blockDim.x) wl[p] = w[p];
  // this receiver's saved activations and encoding
  for (int p = lane; p < nl * H; p += 64) XH[p] = xhat_b0[(size_t)r * nl * H + p];
  if (lane < nl) RS[lane] = rstd_b0[(size_t)r * nl + lane];
  {
    const double* pp = pos + (d.rows ? (size_t)d.rows[b] : (size_t)b) * 3;
    for (int e = lane; e < d.in_dim; e += 64) {
      const int k = e / 6, q = e - 6 * k;
      const double arg = (double)freq_pi[k] * pp[q % 3];
      A0[e] = (float)(q < 3 ? sin(arg) : cos(arg));
    }
  }
  float gg_row = 0.f;
  if (d.gparts > 0) {
    // the gains pass of the output stage's adjoint left (B G, gparts) partial rows: each row summed here by the whole
    // wave, same terms in the same order as k_tf_rows_sum (blocktf.hip) -- that launch sat on the path to the update
    for (int g = 0; g < G; ++g) {
      const float* row = ggains + ((size_t)b * G + g) * d.gparts;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int p = lane;
      for (; p + 192 < d.gparts; p += 256) {
        s0 += row[p];
        s1 += row[p + 64];
        s2 += row[p + 128];
        s3 += row[p + 192];
      }
      for (; p < d.gparts; p += 64) s0 += row[p];
      const float sr = wave_sum((s0 + s1) + (s2 + s3));
      if (lane == g) gg_row = sr;
    }
  }
  if (lane < G) {
    float gg = d.gparts > 0 ? gg_row : ggains[(size_t)b * G + lane];
    if (d.colscale) gg *= d.colscale[(b / d.Bper) * G + lane];
    if (d.hi > d.lo) {
      const float sg = (gains[(size_t)b * G + lane] - d.lo) / (d.hi - d.lo);
      DRAW[lane] = gg * (d.hi - d.lo) * sg * (1.0f - sg);
    } else {
      DRAW[lane] = gg;
    }
  }
  __syncthreads();                               // parameters staged
  for (int p = lane; p < nl * H; p += 64) {      // A[l] = relu(xhat_l gamma_l + beta_l): what leaves layer l
    const int l = p / H, j = p - l * H;
    const float* W = wl + mlp_layer_off(d, l);
    const int n_in = l == 0 ? d.in_dim : H;
    const float* gamma = W + (size_t)H * n_in + H;
    A[p] = fmaxf(XH[p] * gamma[j] + gamma[H + j], 0.f);
  }
  wave_lds_sync();
  // output layer: dL/dA[nl-1]
  float da = 0.f;
  {
    const float* Wout = wl + mlp_layer_off(d, nl);
    if (lane < H)
      for (int g = 0; g < G; ++g) da += Wout[(size_t)g * H + lane] * DRAW[g];
  }
  for (int l = nl - 1; l >= 0; --l) {
    const int n_in = l == 0 ? d.in_dim : H;
    const float* W = wl + mlp_layer_off(d, l);
    const float* gamma = W + (size_t)H * n_in + H;
    float xh = 0.f, dxh = 0.f, dy = 0.f;
    if (lane < H) {
      xh = XH[l * H + lane];
      const float y = xh * gamma[lane] + gamma[H + lane];
      dy = y > 0.f ? da : 0.f;
      dxh = dy * gamma[lane];
    }
    const float m1 = wave_sum(dxh) / (float)H;
    const float m2 = wave_sum(dxh * xh) / (float)H;
    if (lane < H) {
      DY[l * H + lane] = dy;
      DH[l * H + lane] = RS[l] * (dxh - m1 - xh * m2);
    }
    wave_lds_sync();
    if (l > 0) {
      da = 0.f;
      if (lane < H)
        for (int jj = 0; jj < H; ++jj) da += W[(size_t)jj * n_in + lane] * DH[l * H + jj];
    }
  }
  __syncthreads();
  // parameter gradients of the workgroup's receivers, fixed order over r
  const float* base = wl + P;
  const int offOut = (int)mlp_layer_off(d, nl);
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float sum = 0.f;
    if (p >= offOut) {
      const int q = p - offOut;
      if (q < G * H) {
        const int g = q / H, i = q - g * H;
        for (int rr = 0; rr < MLP_RB; ++rr) {
          const float* m = base + (size_t)rr * RS_;
          sum += m[d.in_dim + 4 * nl * H + nl + g] * m[d.in_dim + (nl - 1) * H + i];
        }
      } else {
        const int g = q - G * H;
        for (int rr = 0; rr < MLP_RB; ++rr) sum += (base + (size_t)rr * RS_)[d.in_dim + 4 * nl * H + nl + g];
      }
    } else {
      const int first = H * d.in_dim + 3 * H, per = H * H + 3 * H;
      const int l = p < first ? 0 : 1 + (p - first) / per;
      const int q = p < first ? p : (p - first) - (l - 1) * per;
      const int n_in = l == 0 ? d.in_dim : H;
      const int oA = d.in_dim, oXH = oA + nl * H, oDH = oXH + nl * H, oDY = oDH + nl * H;
      if (q < H * n_in) {
        const int j = q / n_in, i = q - j * n_in;
        const int oact = l == 0 ? i : oA + (l - 1) * H + i;
        for (int rr = 0; rr < MLP_RB; ++rr) {
          const float* m = base + (size_t)rr * RS_;
          sum += m[oDH + l * H + j] * m[oact];
        }
      } else {
        const int kind = (q - H * n_in) / H, j = (q - H * n_in) - kind * H;
        for (int rr = 0; rr < MLP_RB; ++rr) {
          const float* m = base + (size_t)rr * RS_;
          sum += kind == 0 ? m[oDH + l * H + j]
                           : (kind == 1 ? m[oDY + l * H + j] * m[oXH + l * H + j] : m[oDY + l * H + j]);
        }
      }
    }
    gout[p] = sum;
  }
}

__global__ __launch_bounds__(64 * MLP_RB) void k_mlp_bwd_waves(MlpDims d, const double* __restrict__ pos,
                                                               const float* __restrict__ freq_pi,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ gains,
                                                               const float* __restrict__ xhat,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ ggains,   // (B, G)
                                                               float* __restrict__ partial) {      // (B / MLP_RB, P)
  const int b0 = blockIdx.x * MLP_RB;            // (all receivers of a workgroup belong to one band)
  const size_t P = mlp_param_count(d);
  mlp_bwd_waves(d, b0, pos, freq_pi, w + (size_t)(b0 / d.Bper) * P, gains, xhat + (size_t)b0 * d.nl * d.H,
                rstd + (size_t)b0 * d.nl, ggains, partial + (size_t)blockIdx.x * P);
}

// gflat[band][p] = sum_{b in band} partial[b][p]   (B items per band, band = blockIdx.y)
__global__ void k_mlp_reduce(const float* __restrict__ partial, int B, size_t P,
                             float* __restrict__ gflat) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  partial += (size_t)blockIdx.y * B * P;
  gflat += (size_t)blockIdx.y * P;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += partial[(size_t)b * P + p];
  gflat[p] = s;
}

static int mlp_dims(int B, int F, int H, int n_hidden, int G, float lo, float hi, MlpDims* d) {
  if (B <= 0 || F <= 0 || H <= 0 || n_hidden < 0 || G <= 0) return GFDN_E_BADARG;
  if (H > MLP_T || G > MLP_T || 6 * F > 4096) return GFDN_E_UNSUPPORTED;
  d->B = B; d->F = F; d->in_dim = 6 * F; d->H = H; d->nl = 1 + n_hidden; d->G = G; d->lo = lo; d->hi = hi;
  d->rows = nullptr;
  d->Bper = B;
  d->gparts = 0;
  d->colscale = nullptr;
  d->stage = mlp_param_count(*d) <= MLP_STAGE_MAX ? 1 : 0;
  return 0;
}
// threads per receiver: the layer width rounded up to whole wavefronts.  A 16-neuron layer on one
// wavefront turns every block barrier / block reduction of the layer chain into wave-level ops
// (measured: 31 -> see profiles/ for k_mlp_bwd at H = 16).
static int mlp_threads(const MlpDims& d) {
  if (d.stage) return MLP_T;     // the staging loop wants every lane issuing loads
  const int wdt = d.H > d.G ? d.H : d.G;
  int t = 64 * ((wdt + 63) / 64);
  return t > MLP_T ? MLP_T : t;
}
static size_t mlp_lds_bytes(const MlpDims& d) {
  const int amax = d.in_dim > d.H ? d.in_dim : d.H;
  size_t n = (size_t)amax + 2 * (size_t)d.H + d.G + 16;
  if (d.stage) n += 2 * mlp_param_count(d) + (size_t)d.nl * d.H + d.nl;   // weights [+ gradient partials] + saved activations
  else n += 64 * MLP_RB + (size_t)d.nl * d.H + d.nl;                      // partial sums of dL/da (one per thread) /
                                                                          // saved activations of the forward
  return n * sizeof(float);
}

extern "C" size_t gfdn_mlp_param_count(int F, int H, int n_hidden, int G) {
  MlpDims d;
  if (mlp_dims(1, F, H, n_hidden, G, -1.f, 1.f, &d)) return 0;
  return mlp_param_count(d);
}
extern "C" size_t gfdn_mlp_bwd_work_bytes(int B, int F, int H, int n_hidden, int G) {
  return (size_t)B * gfdn_mlp_param_count(F, H, n_hidden, G) * sizeof(float);
}

extern "C" int gfdn_mlp_gains_fwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                                  const float* w, int B, int F, int H, int n_hidden, int G, float lo,
                                  float hi, float* gains, float* xhat, float* rstd, void* stream) {
  return gfdn_mlp_gains_banded_fwd(pos, pos_rows, freq_pi, w, 1, B, F, H, n_hidden, G, lo, hi, gains, xhat,
                                   rstd, stream);
}

extern "C" int gfdn_mlp_gains_banded_fwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                                         const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                                         int G, float lo, float hi, float* gains, float* xhat, float* rstd,
                                         void* stream) {
  MlpDims d;
  if (nbands <= 0 || Bper <= 0) return GFDN_E_BADARG;
  const int B = nbands * Bper;
  int rc = mlp_dims(B, F, H, n_hidden, G, lo, hi, &d);
  if (rc) return rc;
  d.Bper = Bper;
  d.rows = pos_rows;
  if (!pos || !freq_pi || !w || !gains || !xhat || !rstd) return GFDN_E_BADARG;
  if (d.stage && H <= 64 && G <= 64 && Bper % MLP_RB == 0) {
    const int amax = d.in_dim > H ? d.in_dim : H;
    const size_t lds = (mlp_param_count(d) + (size_t)MLP_RB * (amax + d.nl * H + d.nl)) * sizeof(float);
    rc = ensure_dyn_lds(k_mlp_fwd_waves, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mlp_fwd_waves, dim3(B / MLP_RB), dim3(64 * MLP_RB), lds, (hipStream_t)stream, d, pos, freq_pi, w,
                       gains, xhat, rstd);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_mlp_fwd, dim3(B), dim3(mlp_threads(d)), mlp_lds_bytes(d), (hipStream_t)stream, d, pos,
                     freq_pi, w, gains, xhat, rstd);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_mlp_gains_bwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                                  const float* w, int B, int F, int H, int n_hidden, int G, float lo,
                                  float hi, const float* gains, const float* xhat, const float* rstd,
                                  const float* ggains, float* gw, void* work, void* stream) {
  return gfdn_mlp_gains_banded_bwd(pos, pos_rows, freq_pi, w, 1, B, F, H, n_hidden, G, lo, hi, gains, xhat,
                                   rstd, ggains, gw, work, stream);
}

static int mlp_banded_bwd_run(const double* pos, const long long* pos_rows, const float* freq_pi,
                              const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                              int G, float lo, float hi, const float* gains, const float* xhat,
                              const float* rstd, const float* ggains, int gparts, float* gw, void* work,
                              void* stream, const float* colscale = nullptr);

// 1 when gfdn_mlp_gains_banded_bwd_parts takes this network (the wave-per-receiver form applies), else 0
extern "C" int gfdn_mlp_bwd_takes_parts(int F, int H, int n_hidden, int G, int Bper) {
  MlpDims d;
  if (Bper <= 0 || mlp_dims(Bper, F, H, n_hidden, G, 0.f, 0.f, &d)) return 0;
  return (d.stage && H <= 64 && G <= 64 && Bper % MLP_RB == 0) ? 1 : 0;
}

extern "C" int gfdn_mlp_gains_banded_bwd(const double* pos, const long long* pos_rows, const float* freq_pi,
                                         const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                                         int G, float lo, float hi, const float* gains, const float* xhat,
                                         const float* rstd, const float* ggains, float* gw, void* work,
                                         void* stream) {
  return mlp_banded_bwd_run(pos, pos_rows, freq_pi, w, nbands, Bper, F, H, n_hidden, G, lo, hi, gains, xhat, rstd, ggains,
                            0, gw, work, stream);
}
// ... with dL/dgains as the (nbands Bper G, gparts) partial rows gfdn_tf_gain_grad leaves when it is called without an
// output: the row sums happen inside this launch (wave-per-receiver form only: GFDN_E_UNSUPPORTED otherwise)
extern "C" int gfdn_mlp_gains_banded_bwd_parts(const double* pos, const long long* pos_rows, const float* freq_pi,
                                               const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                                               int G, float lo, float hi, const float* gains, const float* xhat,
                                               const float* rstd, const float* ggains_parts, int gparts, float* gw,
                                               void* work, void* stream) {
  if (gparts <= 0) return GFDN_E_BADARG;
  return mlp_banded_bwd_run(pos, pos_rows, freq_pi, w, nbands, Bper, F, H, n_hidden, G, lo, hi, gains, xhat, rstd,
                            ggains_parts, gparts, gw, work, stream);
}
// ... where the partial rows hold dL/d(gains colscale): every row sum is multiplied by colscale[band G + g] first
extern "C" int gfdn_mlp_gains_banded_bwd_parts_scaled(const double* pos, const long long* pos_rows, const float* freq_pi,
                                                      const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                                                      int G, float lo, float hi, const float* gains, const float* xhat,
                                                      const float* rstd, const float* ggains_parts, int gparts,
                                                      const float* colscale, float* gw, void* work, void* stream) {
  if (gparts <= 0 || !colscale) return GFDN_E_BADARG;
  return mlp_banded_bwd_run(pos, pos_rows, freq_pi, w, nbands, Bper, F, H, n_hidden, G, lo, hi, gains, xhat, rstd,
                            ggains_parts, gparts, gw, work, stream, colscale);
}

static int mlp_banded_bwd_run(const double* pos, const long long* pos_rows, const float* freq_pi,
                              const float* w, int nbands, int Bper, int F, int H, int n_hidden,
                              int G, float lo, float hi, const float* gains, const float* xhat,
                              const float* rstd, const float* ggains, int gparts, float* gw, void* work,
                              void* stream, const float* colscale) {
  MlpDims d;
  if (nbands <= 0 || Bper <= 0) return GFDN_E_BADARG;
  const int B = nbands * Bper;
  int rc = mlp_dims(B, F, H, n_hidden, G, lo, hi, &d);
  if (rc) return rc;
  d.Bper = Bper;
  d.gparts = gparts;
  d.rows = pos_rows;
  d.colscale = colscale;
  if (!pos || !freq_pi || !w || !gains || !xhat || !rstd || !ggains || !gw || !work) return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t P = mlp_param_count(d);
  if (d.stage && H <= 64 && G <= 64 && Bper % MLP_RB == 0) {
    const size_t lds = (P + (size_t)MLP_RB * mlp_rb_stride(d)) * sizeof(float);
    int erc = ensure_dyn_lds(k_mlp_bwd_waves, lds);
    if (erc) return erc;
    hipLaunchKernelGGL(k_mlp_bwd_waves, dim3(B / MLP_RB), dim3(64 * MLP_RB), lds, s, d, pos, freq_pi, w, gains, xhat,
                       rstd, ggains, (float*)work);
    GFDN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((P + 255) / 256), nbands), dim3(256), 0, s, (const float*)work,
                       Bper / MLP_RB, P, gw);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  if (gparts > 0) return GFDN_E_UNSUPPORTED;      // (the block-per-receiver form reads summed gradients)
  hipLaunchKernelGGL(k_mlp_bwd, dim3(B), dim3(mlp_threads(d)), mlp_lds_bytes(d), s, d, pos, freq_pi, w, gains, xhat,
                     rstd, ggains, (float*)work);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((P + 255) / 256), nbands), dim3(256), 0, s, (const float*)work, Bper, P, gw);
  GFDN_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------
// Bands with DIFFERENT layer sizes in one launch (round 6).  The reference's sub-band driver gives every band its own gain
// network -- 63 Hz: 1 x 8, 125 Hz: 1 x 16, 250-1000 Hz: 5 x 16, 2-8 kHz: 3 x 128 hidden layers x neurons
// (src/run_subband_training_treble.py:61-73) -- which the entry points above (one MlpDims for all bands) cannot take.
// Here a table holds every band's sizes and offsets; a workgroup finds its band from its id and runs the same device
// bodies: bands whose packed parameters fit in LDS (<= 64 neurons) a wave per receiver, the others a workgroup per receiver
// with the parameters read from memory (waves over the neurons, lanes over the inputs: coalesced rows).
//   w, gw : the bands' packed parameter sets one after the other (band q at woff[q])
//   xhat  : the bands' (Bper, nl_q, H_q) blocks one after the other;  rstd: their (Bper, nl_q) blocks
// ------------------------------------------------------------------------------------------
#define MLP_MAXBANDS 16
struct MlpBandTab {
  int nbands;
  int H[MLP_MAXBANDS], nl[MLP_MAXBANDS], mode[MLP_MAXBANDS];      // mode 0: a wave per receiver, 1: a workgroup per receiver
  int wg0[MLP_MAXBANDS + 1];                                      // first workgroup of band q
  int prow[MLP_MAXBANDS];                                         // rows of parameter-gradient partials of band q
  unsigned woff[MLP_MAXBANDS], xoff[MLP_MAXBANDS], roff[MLP_MAXBANDS], poff[MLP_MAXBANDS], P[MLP_MAXBANDS];
};

__device__ __forceinline__ int mlp_band_of(const MlpBandTab& t, int wg) {
  int q = 0;
  while (q + 1 < t.nbands && wg >= t.wg0[q + 1]) ++q;
  return q;
}

__global__ __launch_bounds__(64 * MLP_RB) void k_mlp_bands_fwd(MlpDims d, MlpBandTab t, const double* __restrict__ pos,
                                                               const float* __restrict__ freq_pi,
                                                               const float* __restrict__ w, float* __restrict__ gains,
                                                               float* __restrict__ xhat, float* __restrict__ rstd) {
  const int q = mlp_band_of(t, blockIdx.x), local = blockIdx.x - t.wg0[q];
  d.H = t.H[q];
  d.nl = t.nl[q];
  d.stage = (t.mode[q] == 0 || t.P[q] <= MLP_STAGE_MAX) ? 1 : 0;
  const size_t per = (size_t)d.nl * d.H;
  if (t.mode[q] == 0) {
    const int b0 = q * d.Bper + local * MLP_RB;
    mlp_fwd_waves(d, b0, pos, freq_pi, w + t.woff[q], gains, xhat + t.xoff[q] + (size_t)local * MLP_RB * per,
                  rstd + t.roff[q] + (size_t)local * MLP_RB * d.nl);
  } else {
    mlp_fwd_block(d, q * d.Bper + local, pos, freq_pi, w + t.woff[q], gains, xhat + t.xoff[q] + (size_t)local * per,
                  rstd + t.roff[q] + (size_t)local * d.nl);
  }
}

__global__ __launch_bounds__(64 * MLP_RB) void k_mlp_bands_bwd(MlpDims d, MlpBandTab t, const double* __restrict__ pos,
                                                               const float* __restrict__ freq_pi,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ gains,
                                                               const float* __restrict__ xhat,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ ggains,
                                                               float* __restrict__ partial) {
  const int q = mlp_band_of(t, blockIdx.x), local = blockIdx.x - t.wg0[q];
  d.H = t.H[q];
  d.nl = t.nl[q];
  d.stage = (t.mode[q] == 0 || t.P[q] <= MLP_STAGE_MAX) ? 1 : 0;
  const size_t per = (size_t)d.nl * d.H;
  float* gout = partial + t.poff[q] + (size_t)local * t.P[q];
  if (t.mode[q] == 0) {
    const int b0 = q * d.Bper + local * MLP_RB;
    mlp_bwd_waves(d, b0, pos, freq_pi, w + t.woff[q], gains, xhat + t.xoff[q] + (size_t)local * MLP_RB * per,
                  rstd + t.roff[q] + (size_t)local * MLP_RB * d.nl, ggains, gout);
  } else {
    mlp_bwd_block(d, q * d.Bper + local, pos, freq_pi, w + t.woff[q], gains, xhat + t.xoff[q] + (size_t)local * per,
                  rstd + t.roff[q] + (size_t)local * d.nl, ggains, gout);
  }
}

// gflat[woff[q] + p] = sum over band q's rows of partials, in row order (band = blockIdx.y)
__global__ void k_mlp_bands_reduce(MlpBandTab t, const float* __restrict__ partial, float* __restrict__ gflat) {
  const int q = blockIdx.y;
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= t.P[q]) return;
  const float* src = partial + t.poff[q] + p;
  float sum = 0.f;
  for (int r = 0; r < t.prow[q]; ++r) sum += src[(size_t)r * t.P[q]];
  gflat[t.woff[q] + p] = sum;
}

// the table of a bank; returns 0 or an error code.  lds_fwd / lds_bwd: dynamic LDS of the two launches (bytes)
static int mlp_band_tab(int nbands, int Bper, int F, const int* H, const int* n_hidden, int G, float lo, float hi,
                        MlpDims* d, MlpBandTab* t, size_t* lds_fwd, size_t* lds_bwd, size_t* nparam, size_t* nxhat,
                        size_t* nrstd, size_t* npart) {
  if (nbands <= 0 || Bper <= 0 || !H || !n_hidden) return GFDN_E_BADARG;
  if (nbands > MLP_MAXBANDS) return GFDN_E_UNSUPPORTED;
  size_t wo = 0, xo = 0, ro = 0, po = 0, lf = 0, lb = 0;
  int wg = 0;
  t->nbands = nbands;
  for (int q = 0; q < nbands; ++q) {
    MlpDims dq;
    int rc = mlp_dims(nbands * Bper, F, H[q], n_hidden[q], G, lo, hi, &dq);
    if (rc) return rc;
    if (q == 0) *d = dq;
    const size_t P = mlp_param_count(dq);
    const bool waves = dq.stage && H[q] <= 64 && G <= 64 && Bper % MLP_RB == 0;
    t->H[q] = H[q];
    t->nl[q] = dq.nl;
    t->mode[q] = waves ? 0 : 1;
    t->wg0[q] = wg;
    t->prow[q] = waves ? Bper / MLP_RB : Bper;
    t->woff[q] = (unsigned)wo; t->xoff[q] = (unsigned)xo; t->roff[q] = (unsigned)ro; t->poff[q] = (unsigned)po;
    t->P[q] = (unsigned)P;
    wg += t->prow[q];
    wo += P;
    xo += (size_t)Bper * dq.nl * H[q];
    ro += (size_t)Bper * dq.nl;
    po += (size_t)t->prow[q] * P;
    const int amax = dq.in_dim > H[q] ? dq.in_dim : H[q];
    const size_t f = waves ? (P + (size_t)MLP_RB * (amax + dq.nl * H[q] + dq.nl)) * sizeof(float) : mlp_lds_bytes(dq);
    const size_t b = waves ? (P + (size_t)MLP_RB * mlp_rb_stride(dq)) * sizeof(float) : mlp_lds_bytes(dq);
    if (f > lf) lf = f;
    if (b > lb) lb = b;
  }
  if (po > 0xFFFFFFFFull || xo > 0xFFFFFFFFull) return GFDN_E_UNSUPPORTED;
  t->wg0[nbands] = wg;
  d->Bper = Bper;
  if (lds_fwd) *lds_fwd = lf;
  if (lds_bwd) *lds_bwd = lb;
  if (nparam) *nparam = wo;
  if (nxhat) *nxhat = xo;
  if (nrstd) *nrstd = ro;
  if (npart) *npart = po;
  return 0;
}

// sizes[0..3] = floats of the packed parameters, of xhat, of rstd, of the backward's work buffer
extern "C" int gfdn_mlp_bands_sizes(int nbands, int Bper, int F, const int* H, const int* n_hidden, int G, size_t* sizes) {
  MlpDims d;
  MlpBandTab t;
  if (!sizes) return GFDN_E_BADARG;
  return mlp_band_tab(nbands, Bper, F, H, n_hidden, G, -1.f, 1.f, &d, &t, nullptr, nullptr, sizes, sizes + 1, sizes + 2,
                      sizes + 3);
}

extern "C" int gfdn_mlp_gains_bands_fwd(const double* pos, const long long* pos_rows, const float* freq_pi, const float* w,
                                        int nbands, int Bper, int F, const int* H, const int* n_hidden, int G, float lo,
                                        float hi, float* gains, float* xhat, float* rstd, void* stream) {
  MlpDims d;
  MlpBandTab t;
  size_t lds = 0;
  int rc = mlp_band_tab(nbands, Bper, F, H, n_hidden, G, lo, hi, &d, &t, &lds, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (!pos || !freq_pi || !w || !gains || !xhat || !rstd) return GFDN_E_BADARG;
  d.rows = pos_rows;
  rc = ensure_dyn_lds(k_mlp_bands_fwd, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mlp_bands_fwd, dim3(t.wg0[nbands]), dim3(64 * MLP_RB), lds, (hipStream_t)stream, d, t, pos, freq_pi, w,
                     gains, xhat, rstd);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ggains: (nbands Bper, G) summed gradients (gparts = 0) or the (nbands Bper G, gparts) partial rows of the output stage's
// adjoint, summed inside the launch; colscale (optional, (nbands G)): every row sum is multiplied by colscale[band G + g]
extern "C" int gfdn_mlp_gains_bands_bwd(const double* pos, const long long* pos_rows, const float* freq_pi, const float* w,
                                        int nbands, int Bper, int F, const int* H, const int* n_hidden, int G, float lo,
                                        float hi, const float* gains, const float* xhat, const float* rstd,
                                        const float* ggains, int gparts, const float* colscale, float* gw, void* work,
                                        void* stream) {
  MlpDims d;
  MlpBandTab t;
  size_t lds = 0;
  int rc = mlp_band_tab(nbands, Bper, F, H, n_hidden, G, lo, hi, &d, &t, nullptr, &lds, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (!pos || !freq_pi || !w || !gains || !xhat || !rstd || !ggains || !gw || !work || gparts < 0) return GFDN_E_BADARG;
  d.rows = pos_rows;
  d.gparts = gparts;
  d.colscale = colscale;
  rc = ensure_dyn_lds(k_mlp_bands_bwd, lds);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mlp_bands_bwd, dim3(t.wg0[nbands]), dim3(64 * MLP_RB), lds, s, d, t, pos, freq_pi, w, gains, xhat,
                     rstd, ggains, (float*)work);
  GFDN_LAUNCH_CHECK();
  unsigned maxP = 0;
  for (int q = 0; q < nbands; ++q) maxP = t.P[q] > maxP ? t.P[q] : maxP;
  hipLaunchKernelGGL(k_mlp_bands_reduce, dim3((maxP + 255) / 256, nbands), dim3(256), 0, s, t, (const float*)work, gw);
  GFDN_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------
// Row normalisation of the SH-domain receiver weights, y = w / (||w||_2 + eps) over the last axis (reference
// spatial_sampling/model.py:117-190 ``normalise_weights``: weights / (torch.norm(weights, dim=-1, keepdim=True) + 1e-6))
// and its adjoint, one thread per row: the tensor-operator form is 3 launches forward and 11 backward on 96 rows.
//   gw_j = g_j / (n + eps) - w_j (g . w) / (n (n + eps)^2),  the second term dropped at n = 0 (as torch's norm backward)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rownorm(const float* __restrict__ w, int rows, int len, float eps,
                                                 const float* __restrict__ g, float* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* wr = w + (size_t)r * len;
  float ss = 0.f, gw = 0.f;
  for (int i = 0; i < len; ++i) {
    ss += wr[i] * wr[i];
    if (g) gw += g[(size_t)r * len + i] * wr[i];
  }
  const float n = sqrtf(ss), d = 1.0f / (n + eps);
  if (!g) {
    for (int i = 0; i < len; ++i) out[(size_t)r * len + i] = wr[i] * d;
  } else {
    const float k = n > 0.f ? gw * d * d / n : 0.f;
    for (int i = 0; i < len; ++i) out[(size_t)r * len + i] = g[(size_t)r * len + i] * d - wr[i] * k;
  }
}

extern "C" int gfdn_rownorm_fwd(const float* w, int rows, int len, float eps, float* y, void* stream) {
  if (!w || !y || rows <= 0 || len <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_rownorm, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, rows, len, eps,
                     (const float*)nullptr, y);
  GFDN_LAUNCH_CHECK();
  return 0;
}
extern "C" int gfdn_rownorm_bwd(const float* w, int rows, int len, float eps, const float* gy, float* gw, void* stream) {
  if (!w || !gy || !gw || rows <= 0 || len <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_rownorm, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, rows, len, eps, gy, gw);
  GFDN_LAUNCH_CHECK();
  return 0;
}
