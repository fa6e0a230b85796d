// Energy-decay losses on the time-domain RIRs, for gfx950.
//
// EDR (losses.py:556-575, :478-492): EDR[f][m] = 10 log10(sum_{tau >= m} |S[f][tau]|^2 + eps),
//   per item sum_{f,m} wf[f] |EDR_t - EDR_a| / sum |EDR_t|, summed over the batch.
// EDC (losses.py:187-238): Schroeder integral of x[start:start+len]^2, dB, mean |difference|
//   over batch x kept indices.
// Both are evaluated fused with their gradients: the loss is a scalar, so the backward of
// the dB / L1 stage is produced in the same pass that produces the value.
// All cross-thread sums use fixed-order reductions (no float atomics): bitwise reproducible.
#include "common.h"
#include "scan_dev.h"
#include "xlin_dev.h"

// out[r] = scale * sum_c part[r][c] * (rowscale ? 1/rowscale[r] : 1)
__global__ void k_row_sum(const float* __restrict__ part, int rows, int cols,
                          const float* __restrict__ rowdiv, const long long* __restrict__ divrows,
                          float* __restrict__ out) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = 0; c < cols; ++c) s += part[(size_t)r * cols + c];
  out[r] = rowdiv ? s / rowdiv[divrows ? divrows[r] : r] : s;
}

// ------------------------------------------------------------------------------------------
// EDR
// ------------------------------------------------------------------------------------------
#define EDR_MAX_FBLK 64   // nfreq <= 64*256

// target: P -> EDR dB in place; part[b][fblk] = sum |EDR|
__global__ __launch_bounds__(256) void k_edr_target(float* __restrict__ P, int nframes, int nfreq,
                                                    float* __restrict__ part) {
  __shared__ float s_red[16];
  const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  float sabs = 0.f;
  if (f < nfreq) {
    float* p = P + (size_t)b * nframes * nfreq + f;
    float E = 0.f;
    for (int m = nframes - 1; m >= 0; --m) {
      E += p[(size_t)m * nfreq];
      float d = db_pow(E);
      p[(size_t)m * nfreq] = d;
      sabs += fabsf(d);
    }
  }
  sabs = block_sum(sabs, s_red);
  if (threadIdx.x == 0) part[b * gridDim.x + blockIdx.x] = sabs;
}

// achieved: part[b][fblk] = sum_{f,m} wf |T - EDR| ; optionally P <- gscale/sum_abs[b] * dloss/dP
__global__ __launch_bounds__(256) void k_edr_loss(float* __restrict__ P,
                                                  const float* __restrict__ Tdb,
                                                  const float* __restrict__ sum_abs,
                                                  const long long* __restrict__ trows,
                                                  const float* __restrict__ wf, int nframes,
                                                  int nfreq, float gscale, int want_grad,
                                                  float* __restrict__ part) {
  __shared__ float s_red[16];
  const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (f < nfreq) {
    float* p = P + (size_t)b * nframes * nfreq + f;
    const size_t tb = trows ? (size_t)trows[b] : (size_t)b;
    const float* t = Tdb + tb * nframes * nfreq + f;
    const float w = wf ? wf[f] : 1.0f;
    const float gs = want_grad ? gscale * w / sum_abs[tb] : 0.f;
    float E = 0.f;
    for (int m = nframes - 1; m >= 0; --m) {
      E += p[(size_t)m * nfreq];
      const float lin = fabsf(E) + F32_EPS;
      const float raw = 10.0f * log10f(lin);
      const float d = fmaxf(raw, -200.0f);
      const float diff = t[(size_t)m * nfreq] - d;
      acc += w * fabsf(diff);
      if (want_grad) {
        // d|diff|/dEDR = -sign(diff); dEDR/dE = (10/ln10)/(E+eps) unless clipped
        const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
        const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
        p[(size_t)m * nfreq] = -sg * dE * gs;
      }
    }
    if (want_grad) {
      // E_m = sum_{tau >= m} P_tau  =>  gP_tau = sum_{m <= tau} gE_m
      float run = 0.f;
      for (int m = 0; m < nframes; ++m) {
        run += p[(size_t)m * nfreq];
        p[(size_t)m * nfreq] = run;
      }
    }
  }
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) part[b * gridDim.x + blockIdx.x] = acc;
}

// Same maths for nframes <= EDR_RF with the whole frame column of P and of the target in registers:
// all 2 x nframes loads of a thread are in flight before the dependent dB / prefix chains start
// (the loop version above waits one memory latency per frame; measured 27 -> see profiles/).
// (One block per item, which would make the second launch unnecessary, is bound by what a single CU
// can stream: 35 us.)
#define EDR_RF 32
__global__ __launch_bounds__(256) void k_edr_loss_cols(float* __restrict__ P,
                                                       const float* __restrict__ Tdb,
                                                       const float* __restrict__ sum_abs,
                                                       const long long* __restrict__ trows,
                                                       const float* __restrict__ wf, int nframes,
                                                       int nfreq, float gscale, int want_grad,
                                                       float* __restrict__ part) {
  __shared__ float s_red[16];
  const int b = blockIdx.y;
  const size_t tb = trows ? (size_t)trows[b] : (size_t)b;
  const float inv_norm = 1.0f / sum_abs[tb];
  float acc = 0.f;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f < nfreq) {
    float* p = P + (size_t)b * nframes * nfreq + f;
    const float* t = Tdb + tb * nframes * nfreq + f;
    float pv[EDR_RF], tv[EDR_RF];
#pragma unroll
    for (int m = 0; m < EDR_RF; ++m) {
      pv[m] = m < nframes ? p[(size_t)m * nfreq] : 0.f;
      tv[m] = m < nframes ? t[(size_t)m * nfreq] : 0.f;
    }
    const float w = wf ? wf[f] : 1.0f;
    const float gs = want_grad ? gscale * w * inv_norm : 0.f;
    float E = 0.f;
#pragma unroll
    for (int m = EDR_RF - 1; m >= 0; --m) {
      if (m < nframes) {
        E += pv[m];
        const float lin = fabsf(E) + F32_EPS;
        const float raw = 10.0f * log10f(lin);
        const float d = fmaxf(raw, -200.0f);
        const float diff = tv[m] - d;
        acc += w * fabsf(diff);
        const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
        const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
        pv[m] = -sg * dE * gs;
      }
    }
    if (want_grad) {
      float run = 0.f;
#pragma unroll
      for (int m = 0; m < EDR_RF; ++m) {
        if (m < nframes) {
          run += pv[m];
          p[(size_t)m * nfreq] = run;
        }
      }
    }
  }
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) part[b * gridDim.x + blockIdx.x] = acc;
}

// work: per-(item, frequency-block) partial sums, reduced in a fixed order by k_row_sum
extern "C" size_t gfdn_edr_work_bytes(int batch, int nfreq) {
  return (size_t)batch * ((nfreq + 255) / 256) * sizeof(float);
}

extern "C" int gfdn_edr_target(float* P, int batch, int nframes, int nfreq, float* sum_abs,
                               void* work, void* stream) {
  if (!P || !sum_abs || !work || batch <= 0 || nframes <= 0 || nfreq <= 0) return GFDN_E_BADARG;
  const int fblk = (nfreq + 255) / 256;
  if (fblk > EDR_MAX_FBLK) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)work;
  hipLaunchKernelGGL(k_edr_target, dim3(fblk, batch), dim3(256), 0, s, P, nframes, nfreq, part);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_row_sum, dim3((batch + 63) / 64), dim3(64), 0, s, part, batch, fblk,
                     (const float*)nullptr, (const long long*)nullptr, sum_abs);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_edr_loss(float* P, const float* T_db, const float* sum_abs,
                             const long long* target_rows, const float* wf, int batch, int nframes, int nfreq, float gscale, int want_grad,
                             float* loss_item, void* work, void* stream) {
  if (!P || !T_db || !sum_abs || !work || batch <= 0 || nframes <= 0 || nfreq <= 0)
    return GFDN_E_BADARG;
  const int fblk = (nfreq + 255) / 256;
  if (fblk > EDR_MAX_FBLK) return GFDN_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)work;
  if (nframes <= EDR_RF)
    hipLaunchKernelGGL(k_edr_loss_cols, dim3(fblk, batch), dim3(256), 0, s, P, T_db, sum_abs, target_rows, wf,
                       nframes, nfreq, gscale, want_grad, part);
  else
    hipLaunchKernelGGL(k_edr_loss, dim3(fblk, batch), dim3(256), 0, s, P, T_db, sum_abs, target_rows, wf,
                       nframes, nfreq, gscale, want_grad, part);
  GFDN_LAUNCH_CHECK();
  if (!loss_item) return 0;       // deferred: the caller sums the partials (gfdn_weighted_sums)
  hipLaunchKernelGGL(k_row_sum, dim3((batch + 63) / 64), dim3(64), 0, s, part, batch, fblk, sum_abs,
                     target_rows, loss_item);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// EDC.  The window of every item is cut into EDC_NSEG contiguous segments, one 256-thread block each
// (32 items x 8 segments = 256 blocks, one per CU).  Scans are made global by carries:
//   k_edc_segsum : energy of every segment
//   k_edc_seg_fwd: carry-in = energy of all LATER segments; suffix scan inside the segment ->
//                  EDC, dB, |diff| partial sum, dL/dEDC staged in gx, segment sum of dL/dEDC
//   k_edc_seg_bwd: carry-in = dL/dEDC of all EARLIER segments; prefix scan -> dL/dx; item loss
// Inside a block every iteration covers EDC_S striped sub-tiles (thread t owns samples
// [s*1024 + 4t, +4) of sub-tile s): coalesced 16 B per lane, EDC_S loads in flight, and the EDC_S
// block scans share one pair of barriers.  All sums are fixed-order (bitwise reproducible).
// ------------------------------------------------------------------------------------------
// Group form of edc_scan: ``load(s, j0, nv, val)`` fills the EDC_V consecutive scan elements j0 .. j0 + 3 of sub-tile
// s (nv of them inside the segment; it may also fetch whatever else ``fn`` needs for them into arrays of its own) and
// ``fn(s, j0, nv, excl, val)`` receives the exclusive prefix in front of the group.  Every load of a tile is issued
// before the block scan, four consecutive elements per thread as one 16-byte access; the running sums inside a
// group are re-accumulated by ``fn`` in the same order instead of being kept in registers.
template <typename L, typename F>
__device__ __forceinline__ void edc_scan_v(int len, float carry, float* lds, L load, F fn) {
  const int ntiles = (len + EDC_TILE - 1) / EDC_TILE;
  for (int tile = 0; tile < ntiles; ++tile) {
    float val[EDC_S][EDC_V], loc[EDC_S], tot[EDC_S], incl[EDC_S];
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      const int nv = len - j0 >= EDC_V ? EDC_V : (len > j0 ? len - j0 : 0);
      load(s, j0, nv, val[s]);
      float run = 0.f;
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) run += val[s][u];
      loc[s] = run;
      incl[s] = run;
    }
    block_scan_multi(incl, tot, lds);
    float base = carry;
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      const int nv = len - j0 >= EDC_V ? EDC_V : (len > j0 ? len - j0 : 0);
      fn(s, j0, nv, base + incl[s] - loc[s], val[s]);
      base += tot[s];
    }
    carry = base;
    __syncthreads();     // lds is reused by the next iteration
  }
}

template <typename G, typename F>
__device__ __forceinline__ void edc_scan(int len, float carry, float* lds, G get, F fn) {
  const int ntiles = (len + EDC_TILE - 1) / EDC_TILE;
  for (int tile = 0; tile < ntiles; ++tile) {
    float val[EDC_S][EDC_V], pre[EDC_S][EDC_V], loc[EDC_S], tot[EDC_S], incl[EDC_S];
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      float run = 0.f;
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) {
        const int j = j0 + u;
        val[s][u] = (j < len) ? get(j) : 0.f;
        run += val[s][u];
        pre[s][u] = run;
      }
      loc[s] = run;
      incl[s] = run;
    }
    block_scan_multi(incl, tot, lds);
    float base = carry;
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const float excl = base + incl[s] - loc[s];
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) {
        const int j = j0 + u;
        if (j < len) fn(j, excl + pre[s][u], val[s][u]);
      }
      base += tot[s];
    }
    carry = base;
    __syncthreads();     // lds is reused by the next iteration
  }
}

// Per-item windows of a band bank (gfdn_edc_loss*_banded): item b's window is [start, start + item_len[b]), its target
// row has pitch ld_T, and band b / items_per_band reads mask row maskw + band * ld_mask (ld_mask = 0: one shared row).
// item_len == nullptr: one window ``len`` for every item, targets of pitch ``len`` (the plain entry points).
struct EdcBands {
  const int* item_len;
  int ld_T, items_per_band, ld_mask;
};

__device__ __forceinline__ void edc_segment(int len, int seg, int* s0, int* slen) {
  int per = (len + EDC_NSEG - 1) / EDC_NSEG;
  per = (per + 3) & ~3;
  const int a = seg * per;
  *s0 = a < len ? a : len;
  const int e = a + per < len ? a + per : len;
  *slen = e > *s0 ? e - *s0 : 0;
}

// segsum[b][seg] = sum of x^2 over the segment
__global__ __launch_bounds__(EDC_THREADS) void k_edc_segsum(const float* __restrict__ x, int ld,
                                                            int start, int len,
                                                            float* __restrict__ segsum,
                                                            const int* __restrict__ item_len) {
  __shared__ float s_red[16];
  const int seg = blockIdx.x, b = blockIdx.y;
  if (item_len) len = item_len[b];          // (per-item windows: band banks whose bands differ in T60max)
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float* xw = x + (size_t)b * ld + start + s0;
  float acc = 0.f;
  for (int i = threadIdx.x; i < sl; i += blockDim.x) acc += xw[i] * xw[i];
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) segsum[b * EDC_NSEG + seg] = acc;
}

__device__ __forceinline__ float later_segments(const float* segsum, int b, int seg) {
  float c = 0.f;
  for (int s2 = EDC_NSEG - 1; s2 > seg; --s2) c += segsum[b * EDC_NSEG + s2];
  return c;
}

__global__ __launch_bounds__(EDC_THREADS) void k_edc_target(const float* __restrict__ x, int ld,
                                                            int start, int len,
                                                            const float* __restrict__ segsum,
                                                            float* __restrict__ Tdb) {
  __shared__ float s_scan[16 * EDC_S];
  const int seg = blockIdx.x, b = blockIdx.y;
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float* xw = x + (size_t)b * ld + start + s0;
  float* t = Tdb + (size_t)b * len + s0;
  edc_scan(sl, later_segments(segsum, b, seg), s_scan,
           [&](int j) { const float v = xw[sl - 1 - j]; return v * v; },
           [&](int j, float edc, float) { t[sl - 1 - j] = db_pow(edc); });
}

// work layout: segsum[B][NSEG] | partial[B][NSEG] | gsum[B][NSEG]
__global__ __launch_bounds__(EDC_THREADS) void k_edc_seg_fwd(const float* __restrict__ x, int ld,
                                                             int start, int len,
                                                             const float* __restrict__ Tdb,
                                                             const long long* __restrict__ trows,
                                                             const float* __restrict__ maskw,
                                                             float inv_count, float gscale,
                                                             float* __restrict__ work,
                                                             float* __restrict__ gx, int batch,
                                                             const float* __restrict__ amps, int S,
                                                             const float* __restrict__ env, int ld_env,
                                                             EdcBands eb) {
  __shared__ float s_scan[16 * EDC_S];
  __shared__ float s_red[16];
  const int seg = blockIdx.x, b = blockIdx.y;
  const float* segsum = work;
  float* partial = work + (size_t)batch * EDC_NSEG;
  float* gsum = partial + (size_t)batch * EDC_NSEG;
  const int ld_T = eb.item_len ? eb.ld_T : len;
  if (eb.item_len) len = eb.item_len[b];
  if (maskw && eb.ld_mask) maskw += (size_t)(b / eb.items_per_band) * eb.ld_mask;
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float* xw = x + (size_t)b * ld + start + s0;
  // target: a stored EDC in dB (Tdb), or the common-slope model evaluated on the fly (losses.py:354-359: einsum
  // 'bjk,kt->bjt' of the item's amplitudes with the slope envelopes, then dB) -- amps (items, S), env (S, ld_env)
  const float* t = Tdb ? Tdb + (trows ? (size_t)trows[b] : (size_t)b) * ld_T + s0 : nullptr;
  const float* am = amps ? amps + (size_t)b * S : nullptr;
  const float* ev = env ? env + s0 : nullptr;
  const float* mw = maskw ? maskw + s0 : nullptr;
  float* gw = gx ? gx + (size_t)b * ld + start + s0 : nullptr;
  float acc = 0.f, gacc = 0.f;
  // suffix scan from the END of the segment: scan element j = sample sl - 1 - j
  float tv[EDC_S][EDC_V];
  edc_scan_v(sl, later_segments(segsum, b, seg), s_scan,
             [&](int s, int j0, int nv, float (&val)[EDC_V]) {
               const int ilo = sl - 1 - j0 - (EDC_V - 1);
               if (nv == EDC_V) {
                 float xv[4], t4[4];
                 ld4_f(xw + ilo, xv);
                 if (t) {
                   ld4_f(t + ilo, t4);
                 } else {
                   float lin[4] = {0.f, 0.f, 0.f, 0.f};
                   for (int k = 0; k < S; ++k) {
                     float e4[4];
                     ld4_f(ev + (size_t)k * ld_env + ilo, e4);
                     const float ak = am[k];
#pragma unroll
                     for (int u = 0; u < 4; ++u) lin[u] += ak * e4[u];
                   }
#pragma unroll
                   for (int u = 0; u < 4; ++u) t4[u] = fmaxf(10.0f * log10f(fabsf(lin[u]) + F32_EPS), -200.0f);
                 }
#pragma unroll
                 for (int u = 0; u < EDC_V; ++u) { val[u] = xv[3 - u] * xv[3 - u]; tv[s][u] = t4[3 - u]; }
               } else {
#pragma unroll
                 for (int u = 0; u < EDC_V; ++u) {
                   const int i = sl - 1 - j0 - u;
                   const float v = u < nv ? xw[i] : 0.f;
                   val[u] = v * v;
                   float tt = 0.f;
                   if (u < nv) {
                     if (t) tt = t[i];
                     else {
                       float lin = 0.f;
                       for (int k = 0; k < S; ++k) lin += am[k] * ev[(size_t)k * ld_env + i];
                       tt = fmaxf(10.0f * log10f(fabsf(lin) + F32_EPS), -200.0f);
                     }
                   }
                   tv[s][u] = tt;
                 }
               }
             },
             [&](int s, int j0, int nv, float excl, const float (&val)[EDC_V]) {
               const int ilo = sl - 1 - j0 - (EDC_V - 1);
               float m4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
               if (mw) {
                 if (nv == EDC_V) ld4_f(mw + ilo, m4);
                 else {
#pragma unroll
                   for (int u = 0; u < EDC_V; ++u) m4[u] = ilo + u >= 0 ? mw[ilo + u] : 0.f;
                 }
               }
               float gq[4] = {0.f, 0.f, 0.f, 0.f};
               float run = 0.f;
#pragma unroll
               for (int u = 0; u < EDC_V; ++u) {
                 run += val[u];
                 if (u < nv) {
                   const float edc = excl + run;
                   const float lin = fabsf(edc) + F32_EPS;
                   const float raw = 10.0f * log10f(lin);
                   const float d = fmaxf(raw, -200.0f);
                   const float diff = tv[s][u] - d;
                   const float m = m4[EDC_V - 1 - u];
                   acc += m * fabsf(diff);
                   const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
                   const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
                   const float g = -sg * dE * m * inv_count * gscale;   // dL/dEDC_i, staged in place
                   gq[EDC_V - 1 - u] = g;
                   gacc += g;
                 }
               }
               if (gw) {
                 if (nv == EDC_V) st4_f(gw + ilo, gq);
                 else {
#pragma unroll
                   for (int u = 0; u < EDC_V; ++u) if (ilo + u >= 0) gw[ilo + u] = gq[u];
                 }
               }
             });
  if (!gw) gacc = 0.f;
  acc = block_sum(acc, s_red);
  gacc = block_sum(gacc, s_red);
  if (threadIdx.x == 0) {
    partial[b * EDC_NSEG + seg] = acc;
    gsum[b * EDC_NSEG + seg] = gacc;
  }
}

__global__ __launch_bounds__(EDC_THREADS) void k_edc_seg_bwd(const float* __restrict__ x, int ld,
                                                             int start, int len, float inv_count,
                                                             const float* __restrict__ work,
                                                             float* __restrict__ loss_item,
                                                             float* __restrict__ gx, int batch,
                                                             const int* __restrict__ item_len) {
  __shared__ float s_scan[16 * EDC_S];
  const int seg = blockIdx.x, b = blockIdx.y;
  if (item_len) len = item_len[b];
  const float* partial = work + (size_t)batch * EDC_NSEG;
  const float* gsum = partial + (size_t)batch * EDC_NSEG;
  if (seg == 0 && threadIdx.x == 0) {
    float l = 0.f;
    for (int s2 = 0; s2 < EDC_NSEG; ++s2) l += partial[b * EDC_NSEG + s2];
    loss_item[b] = l * inv_count;
  }
  if (!gx) return;
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float* xw = x + (size_t)b * ld + start + s0;
  float* gw = gx + (size_t)b * ld + start + s0;
  float carry = 0.f;
  for (int s2 = 0; s2 < seg; ++s2) carry += gsum[b * EDC_NSEG + s2];
  // EDC_i = sum_{j >= i} x_j^2  =>  dL/dx_j = 2 x_j sum_{i <= j} dL/dEDC_i   (forward prefix scan)
  float xs[EDC_S][EDC_V];
  edc_scan_v(sl, carry, s_scan,
             [&](int s, int j0, int nv, float (&val)[EDC_V]) {
               if (nv == EDC_V) {
                 ld4_f(gw + j0, val);
                 ld4_f(xw + j0, xs[s]);
               } else {
#pragma unroll
                 for (int u = 0; u < EDC_V; ++u) {
                   val[u] = u < nv ? gw[j0 + u] : 0.f;
                   xs[s][u] = u < nv ? xw[j0 + u] : 0.f;
                 }
               }
             },
             [&](int s, int j0, int nv, float excl, const float (&val)[EDC_V]) {
               float out[4];
               float run = 0.f;
#pragma unroll
               for (int u = 0; u < EDC_V; ++u) {
                 run += val[u];
                 out[u] = 2.0f * xs[s][u] * (excl + run);
               }
               if (nv == EDC_V) st4_f(gw + j0, out);
               else {
#pragma unroll
                 for (int u = 0; u < EDC_V; ++u) if (u < nv) gw[j0 + u] = out[u];
               }
             });
  // zeros outside the window
  float* g = gx + (size_t)b * ld;
  if (seg == 0)
    for (int i = threadIdx.x; i < start; i += blockDim.x) g[i] = 0.f;
  if (seg == EDC_NSEG - 1)
    for (int i = start + len + threadIdx.x; i < ld; i += blockDim.x) g[i] = 0.f;
}


// ------------------------------------------------------------------------------------------
// EDC on pair-interleaved signals (x2 (pairs, ld) float2: item 2p in .x, item 2p + 1 in .y): one block
// scans BOTH items of a pair -- every quantity of the scans is a float2, loads and stores are coalesced
// 8-byte accesses (running the per-item kernels over the interleaved layout with an element stride of 2
// halved the efficiency of every access: 127 + 98 us vs 95 + 79 us beside the STFT kernels).
// Same segmentation, carries and summation order per item as the kernels above: identical numbers.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 f2add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

__device__ __forceinline__ void block_scan_multi2(float2 (&v)[EDC_S], float2 (&tot)[EDC_S], float2* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < EDC_S; ++s) {
    float tx, ty;
    v[s].x = wave_scan_incl(v[s].x, tx);
    v[s].y = wave_scan_incl(v[s].y, ty);
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
    float2 pre = make_float2(0.f, 0.f), t = make_float2(0.f, 0.f);
    for (int i = 0; i < nw; ++i) {
      const float2 q = lds[s * 16 + i];
      if (i < w) pre = f2add(pre, q);
      t = f2add(t, q);
    }
    v[s] = f2add(v[s], pre);
    tot[s] = t;
  }
}

__device__ __forceinline__ float block_sum2(float2& v, float* lds /* >= 32 floats */) {
  v.x = wave_sum(v.x);
  v.y = wave_sum(v.y);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { lds[w] = v.x; lds[16 + w] = v.y; }
  __syncthreads();
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < nw; ++i) { sx += lds[i]; sy += lds[16 + i]; }
  v = make_float2(sx, sy);
  return sx;
}

// work layout (items padded to 2 * pairs): segsum[I][NSEG] | partial[I][NSEG] | gsum[I][NSEG]
template <bool LIN>
__global__ __launch_bounds__(EDC_THREADS) void k_edc_pair_segsum(const float2* __restrict__ x2, int ld, int start,
                                                                 int len, float* __restrict__ segsum,
                                                                 const int* __restrict__ item_len, XLin xl, int items,
                                                                 float2* __restrict__ xout) {
  __shared__ float s_red[32];
  const int seg = blockIdx.x, p = blockIdx.y;
  if (item_len) len = item_len[2 * p];      // (both items of a pair belong to one band: one window)
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float2* xw = x2 + (size_t)p * ld + start + s0;
  float2 acc = make_float2(0.f, 0.f);
  if (LIN) {
    // the samples formed on the fly (xlin_dev.h) and -- the one place they are ever written -- stored on the window for
    // the two scans behind this launch (the scans themselves composing them: 189 registers, 138 us against 47)
    const XPair xp = xpair_init(xl, p, items);
    float2* xo = xout + (size_t)p * ld + start + s0;
    for (int i = threadIdx.x; i < sl; i += blockDim.x) {
      const float2 v = xpair_load1(xp, start + s0 + i);
      xo[i] = v;
      acc.x += v.x * v.x;
      acc.y += v.y * v.y;
    }
  } else
  for (int i = threadIdx.x; i < sl; i += blockDim.x) {
    const float2 v = xw[i];
    acc.x += v.x * v.x;
    acc.y += v.y * v.y;
  }
  block_sum2(acc, s_red);
  if (threadIdx.x == 0) {
    segsum[(2 * p) * EDC_NSEG + seg] = acc.x;
    segsum[(2 * p + 1) * EDC_NSEG + seg] = acc.y;
  }
}

template <bool LIN>
__global__ __launch_bounds__(EDC_THREADS, LIN ? 2 : 4) void k_edc_pair_seg_fwd(const float2* __restrict__ x2, int ld, int start,
                                                                  int len, const float* __restrict__ Tdb,
                                                                  const long long* __restrict__ trows,
                                                                  const float* __restrict__ maskw,
                                                                  float inv_count, float gscale,
                                                                  float* __restrict__ work,
                                                                  float2* __restrict__ gx2, int items, EdcBands eb,
                                                                  XLin xl) {
  __shared__ float2 s_scan[16 * EDC_S];
  __shared__ float s_red[32];
  const int seg = blockIdx.x, p = blockIdx.y;
  XPair xp;
  if (LIN) xp = xpair_init(xl, p, items);
  const int I = 2 * (int)gridDim.y, b1 = 2 * p, b2 = b1 + 1;
  const bool two = b2 < items;
  const float* segsum = work;
  float* partial = work + (size_t)I * EDC_NSEG;
  float* gsum = partial + (size_t)I * EDC_NSEG;
  const int ld_T = eb.item_len ? eb.ld_T : len;
  if (eb.item_len) len = eb.item_len[b1];
  if (maskw && eb.ld_mask) maskw += (size_t)(b1 / eb.items_per_band) * eb.ld_mask;
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float2* xw = x2 + (size_t)p * ld + start + s0;
  const float* t1 = Tdb + (trows ? (size_t)trows[b1] : (size_t)b1) * ld_T + s0;
  const float* t2 = two ? Tdb + (trows ? (size_t)trows[b2] : (size_t)b2) * ld_T + s0 : t1;
  const float* mw = maskw ? maskw + s0 : nullptr;
  float2* gw = gx2 ? gx2 + (size_t)p * ld + start + s0 : nullptr;
  float2 acc = make_float2(0.f, 0.f), gacc = make_float2(0.f, 0.f);
  float2 carry = make_float2(later_segments(segsum, b1, seg), later_segments(segsum, b2, seg));
  // Suffix scan of x^2 over the segment, tile by tile from its END (scan element j = sample sl - 1 - j).  Every
  // load of a tile (samples, both targets) is issued before the block scan: four consecutive samples per thread
  // and sub-tile as 16-byte accesses; the running sums inside a thread are re-accumulated after the scan in the
  // same order instead of being kept in registers.
  const int ntiles = (sl + EDC_TILE - 1) / EDC_TILE;
  for (int tile = 0; tile < ntiles; ++tile) {
    float2 val[EDC_S][EDC_V], loc[EDC_S], incl[EDC_S], tot[EDC_S];
    float ta[EDC_S][EDC_V], tb[EDC_S][EDC_V];
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      const int ilo = sl - 1 - j0 - (EDC_V - 1);            // lowest sample index of the group
      if (ilo >= 0) {
        float2 xv[4];
        float t4[4];
        if (LIN) xpair_load4(xp, start + s0 + ilo, xv);
        else ld4_f2(xw + ilo, xv);
#pragma unroll
        for (int u = 0; u < EDC_V; ++u) val[s][u] = make_float2(xv[3 - u].x * xv[3 - u].x, xv[3 - u].y * xv[3 - u].y);
        ld4_f(t1 + ilo, t4);
#pragma unroll
        for (int u = 0; u < EDC_V; ++u) ta[s][u] = t4[3 - u];
        if (two) {
          ld4_f(t2 + ilo, t4);
#pragma unroll
          for (int u = 0; u < EDC_V; ++u) tb[s][u] = t4[3 - u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < EDC_V; ++u) {
          const int i = sl - 1 - j0 - u;
          const float2 v = i >= 0 ? (LIN ? xpair_load1(xp, start + s0 + i) : xw[i]) : make_float2(0.f, 0.f);
          val[s][u] = make_float2(v.x * v.x, v.y * v.y);
          ta[s][u] = i >= 0 ? t1[i] : 0.f;
          tb[s][u] = (i >= 0 && two) ? t2[i] : 0.f;
        }
      }
      float2 run = make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) run = f2add(run, val[s][u]);
      loc[s] = run;
      incl[s] = run;
    }
    block_scan_multi2(incl, tot, s_scan);
    float2 base = carry;
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const float2 excl = make_float2(base.x + incl[s].x - loc[s].x, base.y + incl[s].y - loc[s].y);
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      const int ilo = sl - 1 - j0 - (EDC_V - 1);
      float m4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if (mw) {
        if (ilo >= 0) ld4_f(mw + ilo, m4);
        else {
#pragma unroll
          for (int u = 0; u < EDC_V; ++u) { const int i = ilo + u; m4[u] = i >= 0 ? mw[i] : 0.f; }
        }
      }
      float2 gq[4];
      float2 run = make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) {
        run = f2add(run, val[s][u]);
        const float2 edc = f2add(excl, run);
        const bool in = ilo + (EDC_V - 1 - u) >= 0;
        const float m = m4[EDC_V - 1 - u];
        float2 g = make_float2(0.f, 0.f);
        if (in) {
          {
            const float lin = fabsf(edc.x) + F32_EPS;
            const float raw = 10.0f * log10f(lin);
            const float diff = ta[s][u] - fmaxf(raw, -200.0f);
            acc.x += m * fabsf(diff);
            const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
            const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
            g.x = -sg * dE * m * inv_count * gscale;
          }
          if (two) {
            const float lin = fabsf(edc.y) + F32_EPS;
            const float raw = 10.0f * log10f(lin);
            const float diff = tb[s][u] - fmaxf(raw, -200.0f);
            acc.y += m * fabsf(diff);
            const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
            const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
            g.y = -sg * dE * m * inv_count * gscale;
          }
          gacc = f2add(gacc, g);
        }
        gq[EDC_V - 1 - u] = g;
      }
      if (gw) {                                             // dL/dEDC_i, staged in place
        if (ilo >= 0) st4_f2(gw + ilo, gq);
        else {
#pragma unroll
          for (int u = 0; u < EDC_V; ++u) if (ilo + u >= 0) gw[ilo + u] = gq[u];
        }
      }
      base = f2add(base, tot[s]);
    }
    carry = base;
    __syncthreads();
  }
  block_sum2(acc, s_red);
  block_sum2(gacc, s_red);
  if (threadIdx.x == 0) {
    partial[b1 * EDC_NSEG + seg] = acc.x;
    partial[b2 * EDC_NSEG + seg] = acc.y;
    gsum[b1 * EDC_NSEG + seg] = gacc.x;
    gsum[b2 * EDC_NSEG + seg] = gacc.y;
  }
}

template <bool LIN>
__global__ __launch_bounds__(EDC_THREADS, LIN ? 3 : 4) void k_edc_pair_seg_bwd(const float2* __restrict__ x2, int ld, int start,
                                                                  int len, float inv_count,
                                                                  const float* __restrict__ work,
                                                                  float* __restrict__ loss_item,
                                                                  float2* __restrict__ gx2, int items,
                                                                  const int* __restrict__ item_len, XLin xl,
                                                                  int fill_outside) {
  __shared__ float2 s_scan[16 * EDC_S];
  const int seg = blockIdx.x, p = blockIdx.y;
  XPair xp;
  if (LIN) xp = xpair_init(xl, p, items);
  const int I = 2 * (int)gridDim.y, b1 = 2 * p, b2 = b1 + 1;
  const bool two = b2 < items;
  if (item_len) len = item_len[b1];
  const float* partial = work + (size_t)I * EDC_NSEG;
  const float* gsum = partial + (size_t)I * EDC_NSEG;
  if (seg == 0 && threadIdx.x == 0) {
    float l1 = 0.f, l2 = 0.f;
    for (int s2 = 0; s2 < EDC_NSEG; ++s2) { l1 += partial[b1 * EDC_NSEG + s2]; l2 += partial[b2 * EDC_NSEG + s2]; }
    loss_item[b1] = l1 * inv_count;
    if (two) loss_item[b2] = l2 * inv_count;
  }
  if (!gx2) return;
  int s0, sl;
  edc_segment(len, seg, &s0, &sl);
  const float2* xw = x2 + (size_t)p * ld + start + s0;
  float2* gw = gx2 + (size_t)p * ld + start + s0;
  float2 carry = make_float2(0.f, 0.f);
  for (int s2 = 0; s2 < seg; ++s2) {
    carry.x += gsum[b1 * EDC_NSEG + s2];
    carry.y += gsum[b2 * EDC_NSEG + s2];
  }
  // prefix scan of the staged dL/dEDC, times 2 x: samples and staged terms of a tile are loaded together (16-byte
  // accesses) before the block scan
  const int ntiles = (sl + EDC_TILE - 1) / EDC_TILE;
  for (int tile = 0; tile < ntiles; ++tile) {
    float2 val[EDC_S][EDC_V], xs[EDC_S][EDC_V], loc[EDC_S], incl[EDC_S], tot[EDC_S];
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      if (j0 + EDC_V <= sl) {
        ld4_f2(gw + j0, val[s]);
        if (LIN) xpair_load4(xp, start + s0 + j0, xs[s]);
        else ld4_f2(xw + j0, xs[s]);
      } else {
#pragma unroll
        for (int u = 0; u < EDC_V; ++u) {
          const bool in = j0 + u < sl;
          val[s][u] = in ? gw[j0 + u] : make_float2(0.f, 0.f);
          xs[s][u] = in ? (LIN ? xpair_load1(xp, start + s0 + j0 + u) : xw[j0 + u]) : make_float2(0.f, 0.f);
        }
      }
      float2 run = make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) run = f2add(run, val[s][u]);
      loc[s] = run;
      incl[s] = run;
    }
    block_scan_multi2(incl, tot, s_scan);
    float2 base = carry;
#pragma unroll
    for (int s = 0; s < EDC_S; ++s) {
      const float2 excl = make_float2(base.x + incl[s].x - loc[s].x, base.y + incl[s].y - loc[s].y);
      const int j0 = tile * EDC_TILE + s * EDC_SUB + threadIdx.x * EDC_V;
      float2 out[4];
      float2 run = make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < EDC_V; ++u) {
        run = f2add(run, val[s][u]);
        const float2 cum = f2add(excl, run);
        out[u] = make_float2(2.0f * xs[s][u].x * cum.x, 2.0f * xs[s][u].y * cum.y);
      }
      if (j0 + EDC_V <= sl) st4_f2(gw + j0, out);
      else {
#pragma unroll
        for (int u = 0; u < EDC_V; ++u) if (j0 + u < sl) gw[j0 + u] = out[u];
      }
      base = f2add(base, tot[s]);
    }
    carry = base;
    __syncthreads();
  }
  if (!fill_outside) return;            // (a consumer that reads the window only: gfdn_lin_gamma_dots)
  float2* g = gx2 + (size_t)p * ld;
  if (seg == 0)
    for (int i = threadIdx.x; i < start; i += blockDim.x) g[i] = make_float2(0.f, 0.f);
  if (seg == EDC_NSEG - 1)
    for (int i = start + len + threadIdx.x; i < ld; i += blockDim.x) g[i] = make_float2(0.f, 0.f);
}

extern "C" size_t gfdn_edc_work_bytes(int batch) { return (size_t)3 * batch * EDC_NSEG * sizeof(float); }

extern "C" int gfdn_edc_target(const float* x, int ld, int batch, int start, int len, float* T_db,
                               void* work, void* stream) {
  if (!x || !T_db || !work || batch <= 0 || start < 0 || len <= 0 || start + len > ld)
    return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_edc_segsum, dim3(EDC_NSEG, batch), dim3(EDC_THREADS), 0, s, x, ld, start, len,
                     (float*)work, (const int*)nullptr);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_target, dim3(EDC_NSEG, batch), dim3(EDC_THREADS), 0, s, x, ld, start, len,
                     (const float*)work, T_db);
  GFDN_LAUNCH_CHECK();
  return 0;
}

static int edc_loss_launch(const float* x, int ld, int batch, int start, int len, const float* T_db,
                           const long long* target_rows, const float* maskw, float inv_count, float gscale,
                           float* loss_item, float* gx, void* work, EdcBands eb, hipStream_t s) {
  dim3 grid(EDC_NSEG, batch), block(EDC_THREADS);
  hipLaunchKernelGGL(k_edc_segsum, grid, block, 0, s, x, ld, start, len, (float*)work, eb.item_len);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_seg_fwd, grid, block, 0, s, x, ld, start, len, T_db, target_rows, maskw, inv_count, gscale,
                     (float*)work, gx, batch, (const float*)nullptr, 0, (const float*)nullptr, 0, eb);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_seg_bwd, grid, block, 0, s, x, ld, start, len, inv_count, (const float*)work,
                     loss_item, gx, batch, eb.item_len);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_edc_loss(const float* x, int ld, int batch, int start, int len,
                             const float* T_db, const long long* target_rows, const float* maskw,
                             float inv_count, float gscale,
                             float* loss_item, float* gx, void* work, void* stream) {
  if (!x || !T_db || !loss_item || !work || batch <= 0 || start < 0 || len <= 0 || start + len > ld)
    return GFDN_E_BADARG;
  return edc_loss_launch(x, ld, batch, start, len, T_db, target_rows, maskw, inv_count, gscale, loss_item, gx, work,
                         EdcBands{nullptr, len, 1, 0}, (hipStream_t)stream);
}

// Band bank whose bands have different longest decay times (src/diff_gfdn/trainer.py:56-59: every band's trainer derives
// its EDC window from ITS T60max; run_subband_training_treble.py:286 gives every band its own decay times): item b's
// window is [start, start + item_len[b]) with item_len[b] <= max_len (device int32, one entry per item), its target row has
// pitch ld_T >= max_len, and band b / items_per_band reads the mask row maskw + band * ld_mask (ld_mask = 0: one row).
extern "C" int gfdn_edc_loss_banded(const float* x, int ld, int batch, int start, int max_len, const int* item_len,
                                    const float* T_db, int ld_T, const long long* target_rows, const float* maskw,
                                    int ld_mask, int items_per_band, float inv_count, float gscale, float* loss_item,
                                    float* gx, void* work, void* stream) {
  if (!x || !T_db || !item_len || !loss_item || !work || batch <= 0 || start < 0 || max_len <= 0 || start + max_len > ld ||
      ld_T < max_len || items_per_band <= 0 || ld_mask < 0 || (ld_mask > 0 && ld_mask < max_len))
    return GFDN_E_BADARG;
  return edc_loss_launch(x, ld, batch, start, max_len, T_db, target_rows, maskw, inv_count, gscale, loss_item, gx, work,
                         EdcBands{item_len, ld_T, items_per_band, ld_mask}, (hipStream_t)stream);
}

// gfdn_edc_loss against the common-slope MODEL instead of a stored target (directional loss, losses.py:354-359):
// target EDC of item b = sum_k amps[b][k] env[k][t] over the window samples t < len, in dB (|.| + eps, clipped at
// -200) -- evaluated inside the scan; the (items, len) target and the five passes that build it never exist.
extern "C" int gfdn_edc_loss_model(const float* x, int ld, int batch, int start, int len, const float* amps, int S,
                                   const float* env, int ld_env, const float* maskw, float inv_count, float gscale,
                                   float* loss_item, float* gx, void* work, void* stream) {
  if (!x || !amps || !env || !loss_item || !work || batch <= 0 || start < 0 || len <= 0 || start + len > ld || S <= 0 ||
      ld_env < len)
    return GFDN_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(EDC_NSEG, batch), block(EDC_THREADS);
  hipLaunchKernelGGL(k_edc_segsum, grid, block, 0, s, x, ld, start, len, (float*)work, (const int*)nullptr);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_seg_fwd, grid, block, 0, s, x, ld, start, len, (const float*)nullptr,
                     (const long long*)nullptr, maskw, inv_count, gscale, (float*)work, gx, batch, amps, S, env, ld_env,
                     EdcBands{nullptr, len, 1, 0});
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_seg_bwd, grid, block, 0, s, x, ld, start, len, inv_count, (const float*)work,
                     loss_item, gx, batch, (const int*)nullptr);
  GFDN_LAUNCH_CHECK();
  return 0;
}

static int edc_loss_pairs_launch(const float* x2, int ld, int items, int start, int len, const float* T_db,
                                 const long long* target_rows, const float* maskw, float inv_count, float gscale,
                                 float* loss_item, float* gx2, void* work, EdcBands eb, hipStream_t s,
                                 XLin xl = XLin{nullptr, 0, nullptr, nullptr, 0, nullptr, 1, 0}, int fill_outside = 1) {
  dim3 grid(EDC_NSEG, (items + 1) / 2), block(EDC_THREADS);
  if (xl.xd) {
    // x2 here: scratch (pairs, ld) that receives the window samples from the first launch
    hipLaunchKernelGGL(k_edc_pair_segsum<true>, grid, block, 0, s, (const float2*)x2, ld, start, len, (float*)work,
                       eb.item_len, xl, items, (float2*)x2);
    GFDN_LAUNCH_CHECK();
    xl.xd = nullptr;
  }
  else {
  hipLaunchKernelGGL(k_edc_pair_segsum<false>, grid, block, 0, s, (const float2*)x2, ld, start, len, (float*)work,
                     eb.item_len, xl, items, (float2*)nullptr);
  }
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_pair_seg_fwd<false>, grid, block, 0, s, (const float2*)x2, ld, start, len, T_db, target_rows,
                     maskw, inv_count, gscale, (float*)work, (float2*)gx2, items, eb, xl);
  GFDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edc_pair_seg_bwd<false>, grid, block, 0, s, (const float2*)x2, ld, start, len, inv_count,
                     (const float*)work, loss_item, (float2*)gx2, items, eb.item_len, xl, fill_outside);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// gfdn_edc_loss_pairs[_banded] on signals x[b] = xd[rows[b]] + sum_g rgain[b][g] tau[band G + g] (the time-domain output
// stage, gfdn_lin_combine_fwd) formed by the first of the three launches, which stores them ON THE WINDOW ONLY into the
// scratch xwin2 (pairs, ld) for the two scans -- xd (R, ld_xd >= ld) float with
// row indirection xrows, tau2 pair-interleaved (ld_tau >= ld), rgain (items, G), items = nbands B.  ld = the signals'
// length (the pitch of gx2); item_len / ld_T / ld_mask / B as in gfdn_edc_loss_pairs_banded (item_len NULL: one window
// max_len, T_db rows of pitch max_len).
extern "C" int gfdn_edc_loss_pairs_lin(const float* xd, int ld_xd, const long long* xrows, const float* tau2, int ld_tau,
                                       const float* rgain, int nbands, int B, int G, int ld, int start, int max_len,
                                       const int* item_len, const float* T_db, int ld_T, const long long* target_rows,
                                       const float* maskw, int ld_mask, float inv_count, float gscale, float* loss_item,
                                       float* gx2, int fill_outside, float* xwin2, void* work, void* stream) {
  if (!xd || !tau2 || !rgain || !T_db || !loss_item || !work || !xwin2 || nbands <= 0 || B <= 0 || G <= 0 || G > 4 || start < 0 ||
      max_len <= 0 || start + max_len > ld || ld_xd < ld || ld_tau < ld || ld_mask < 0 || (ld_mask > 0 && ld_mask < max_len))
    return GFDN_E_BADARG;
  if (item_len && (ld_T < max_len || (B & 1))) return GFDN_E_BADARG;
  const int items = nbands * B;
  XLin xl{xd, ld_xd, xrows, (const float2*)tau2, ld_tau, rgain, B, G};
  return edc_loss_pairs_launch(xwin2, ld, items, start, max_len, T_db, target_rows, maskw, inv_count, gscale, loss_item,
                               gx2, work, item_len ? EdcBands{item_len, ld_T, B, ld_mask} : EdcBands{nullptr, max_len, 1, 0},
                               (hipStream_t)stream, xl, fill_outside);
}

// work: gfdn_edc_work_bytes(items + 1)  (per-item slots for 2 * ceil(items / 2) items)
extern "C" int gfdn_edc_loss_pairs(const float* x2, int ld, int items, int start, int len,
                                   const float* T_db, const long long* target_rows, const float* maskw,
                                   float inv_count, float gscale,
                                   float* loss_item, float* gx2, void* work, void* stream) {
  if (!x2 || !T_db || !loss_item || !work || items <= 0 || start < 0 || len <= 0 || start + len > ld)
    return GFDN_E_BADARG;
  return edc_loss_pairs_launch(x2, ld, items, start, len, T_db, target_rows, maskw, inv_count, gscale, loss_item, gx2,
                               work, EdcBands{nullptr, len, 1, 0}, (hipStream_t)stream);
}

// gfdn_edc_loss_banded on pair-interleaved signals: both items of a pair belong to one band (items_per_band even), so a
// pair has ONE window.
extern "C" int gfdn_edc_loss_pairs_banded(const float* x2, int ld, int items, int start, int max_len,
                                          const int* item_len, const float* T_db, int ld_T,
                                          const long long* target_rows, const float* maskw, int ld_mask,
                                          int items_per_band, float inv_count, float gscale, float* loss_item,
                                          float* gx2, void* work, void* stream) {
  if (!x2 || !T_db || !item_len || !loss_item || !work || items <= 0 || start < 0 || max_len <= 0 ||
      start + max_len > ld || ld_T < max_len || items_per_band <= 0 || (items_per_band & 1) || items % items_per_band ||
      ld_mask < 0 || (ld_mask > 0 && ld_mask < max_len))
    return GFDN_E_BADARG;
  return edc_loss_pairs_launch(x2, ld, items, start, max_len, T_db, target_rows, maskw, inv_count, gscale, loss_item,
                               gx2, work, EdcBands{item_len, ld_T, items_per_band, ld_mask}, (hipStream_t)stream);
}

extern "C" int gfdn_abi_version(void) { return GFDN_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// EDC time mask drawn on the device (losses.py:221-227): iid fair bits from Philox4x32-10.
// One block; thread i produces bits 128 i .. 128 i + 127; the kept count is a fixed-order block sum.
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

#define MASK_MAX_LEN 131072
__global__ __launch_bounds__(1024) void k_draw_mask(unsigned long long seed, unsigned long long* state,
                                                    int len, float scale, float* maskw) {
  __shared__ unsigned words[MASK_MAX_LEN / 32];
  __shared__ float red[16];
  const unsigned long long step = state[0];
  const int nchunk = (len + 127) >> 7;
  float cnt = 0.f;
  for (int i = threadIdx.x; i < nchunk; i += blockDim.x) {
    unsigned c[4] = {(unsigned)i, 0u, (unsigned)step, (unsigned)(step >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int t0 = i * 128 + w * 32, left = len - t0;
      unsigned v = c[w];
      if (left < 32) v = left <= 0 ? 0u : (v & ((1u << left) - 1u));
      words[i * 4 + w] = v;
      cnt += (float)__popc(v);
    }
  }
  const float count = block_sum(cnt, red);          // exact: integers below 2^24
  const float wgt = count > 0.f ? scale / count : 0.f;
  for (int t = threadIdx.x; t < len; t += blockDim.x)
    maskw[t] = ((words[t >> 5] >> (t & 31)) & 1u) ? wgt : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) state[0] = step + 1ull;
}

// One row of weights per band from ONE draw of max(band_len) fair bits (the bits of gfdn_draw_mask at the same seed and
// step): band q keeps the first band_len[q] of them, maskw[q][t] = kept * scale / (kept among the first band_len[q]), zero
// behind its window up to the row pitch.  One workgroup: the bits are generated once, then one fixed-order count and
// one row of stores per band.
__global__ __launch_bounds__(1024) void k_draw_mask_banded(unsigned long long seed, unsigned long long* state,
                                                           const int* __restrict__ band_len, int nbands, int max_len,
                                                           int ld_mask, float scale, float* maskw) {
  __shared__ unsigned words[MASK_MAX_LEN / 32];
  __shared__ float red[16];
  const unsigned long long step = state[0];
  const int nchunk = (max_len + 127) >> 7;
  for (int i = threadIdx.x; i < nchunk; i += blockDim.x) {
    unsigned c[4] = {(unsigned)i, 0u, (unsigned)step, (unsigned)(step >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int w = 0; w < 4; ++w) words[i * 4 + w] = c[w];
  }
  __syncthreads();
  for (int q = 0; q < nbands; ++q) {
    const int len = band_len[q] < max_len ? band_len[q] : max_len;
    const int nw = (len + 31) >> 5;
    float cnt = 0.f;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
      const int left = len - i * 32;
      unsigned v = words[i];
      if (left < 32) v &= (1u << left) - 1u;
      cnt += (float)__popc(v);
    }
    const float count = block_sum(cnt, red);          // exact: integers below 2^24
    const float wgt = count > 0.f ? scale / count : 0.f;
    float* row = maskw + (size_t)q * ld_mask;
    for (int t = threadIdx.x; t < ld_mask; t += blockDim.x)
      row[t] = (t < len && ((words[t >> 5] >> (t & 31)) & 1u)) ? wgt : 0.f;
    __syncthreads();                                  // (red is reused by the next band's count)
  }
  if (threadIdx.x == 0) state[0] = step + 1ull;
}

extern "C" int gfdn_draw_mask_banded(unsigned long long seed, unsigned long long* state, const int* band_len, int nbands,
                                     int max_len, int ld_mask, float scale, float* maskw, void* stream) {
  if (!state || !band_len || !maskw || nbands <= 0 || max_len <= 0 || max_len > MASK_MAX_LEN || ld_mask < max_len)
    return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_draw_mask_banded, dim3(1), dim3(1024), 0, (hipStream_t)stream, seed, state, band_len, nbands,
                     max_len, ld_mask, scale, maskw);
  return (int)hipGetLastError();
}

extern "C" int gfdn_draw_mask(unsigned long long seed, unsigned long long* state, int len, float scale,
                              float* maskw, void* stream) {
  if (!state || !maskw || len <= 0 || len > MASK_MAX_LEN) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_draw_mask, dim3(1), dim3(1024), 0, (hipStream_t)stream, seed, state, len, scale, maskw);
  return (int)hipGetLastError();
}
