// The EDC term of the linear step as ONE register-resident launch, for gfx950.
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn/losses.py:187-238): rir = irfft(H, n = K)[start : start + L],
// EDC_t = sum_{s >= t} rir_s^2 (Schroeder), loss = mean over (items, kept t) of |10 log10(EDC_t + eps) - target_t|, with
// the bernoulli time mask of :221-227 as per-sample weights.  With the output stage in the time domain (linear.hip) a
// receiver's samples are x[b][t] = xd[row_b][t] + sum_g gain[b][g] tau_g[t], formed where they are read.
//
// k_edc_lin_one: ONE workgroup of 1024 threads per receiver keeps the receiver's whole window in registers (4 consecutive
// samples x 12 tiles of 4096 per thread) and does, without touching memory in between,
//     compose x -> squares -> suffix sums (EDC) -> dB vs the target (read once) -> |difference| (loss) and dL/dEDC
//     -> prefix sums -> dL/dx = 2 x sum_{s <= t} dL/dEDC_s (written once, window only)
//     -> the EDC part of dL/dgain[b][g] = <dL/dx, tau_g>   (tau re-read from the cache)
// Both scans are two-level: inside a wave on the VALU (DPP, common.h), across the 16 waves and the 12 tiles through one
// small LDS table each -- two workgroup barriers per scan for the whole window, against one block scan per 4096-sample tile
// and three launches (segment energies, forward scan, backward scan; losses.hip k_edc_pair_*) that staged the window
// samples and dL/dEDC through memory: per receiver 4 L B written + 12 L B read there, L = window length, against 8 L B
// read (xd, target) + 4 L B written here.  Nine of the twelve tiles of staged dL/dEDC values live in LDS (144 KB, 16-byte slots; a workgroup owns its CU)
// so that the thread's registers stay below 128 (four waves per SIMD).
// The suffix scan runs from the END of the window (small tail energies are summed first) and the prefix scan from its START
// (the small early terms of dL/dEDC first), as the three-launch form does; sums are in fixed order (bitwise reproducible).
//
// k_lin_gamma_win: gamma_g = base_a + base_b + sum_b gain[b][g] dL/dx[b] on the window-only gradient rows the launch above
// leaves, in the adjoint transform's slot order (the light form of linear.hip's k_lin_gamma_dots: the dot products are
// already done).
#include "common.h"
#include "scan_dev.h"

#define E1_T 1024
#define E1_V 4
#define E1_TILE (E1_T * E1_V)        // 4096 samples
#define E1_W (E1_T / 64)             // 16 waves
#define E1_NT 12                     // tiles per window: L <= 49152
#ifndef E1_NL
#define E1_NL 9                      // tiles whose dL/dEDC values wait in LDS (144 KB; the others in registers)
#endif
#define E1_MAXG 4
#ifndef E1_SB
#define E1_SB 2                       // tiles per scheduling group (their loads are in flight together)
#endif

struct Edc1Args {
  const float* xd;             // (R, ld_xd): transformed direct paths
  int ld_xd;
  const long long* xrows;      // item -> row of xd (NULL: identity)
  const float2* tau2;          // (ceil(nbands G / 2), ld_tau) pair-interleaved group signals
  int ld_tau;
  const float* rgain;          // (items, G)
  int B, G, items;
  int start, max_len;
  const int* item_len;         // per-item window lengths (NULL: max_len)
  const float* Tdb;            // target EDC in dB, rows of pitch ld_T
  int ld_T;
  const long long* trows;
  const float* maskw;          // time weights (NULL: 1), row band * ld_mask
  int ld_mask;
  float inv_count, gscale;
  float* loss_item;            // (items)
  float* gx;                   // (items, ld_gx): dL/dx on the window, sample start + j at column j (NULL: loss only)
  int ld_gx;
  float* dots;                 // dots[(item G + g) ld_dots + col] = <dL/dx, tau_g>
  int ld_dots, col;
};

// Addressing: every array of the launch is walked through ONE per-thread 32-bit sample index on top of workgroup-uniform
// base pointers (scalar base + 32-bit vector offset loads).  With 64-bit per-thread pointers per (array, tile) the compiler
// kept 12 x 6 address pairs alive across the phases and spilled 500 registers.
typedef float e1f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float e1q __attribute__((ext_vector_type(4)));                              // (16-byte aligned: LDS slots)
__device__ __forceinline__ e1f4 e1_ld(const float* base, unsigned idx) {            // base[idx .. idx + 3], base uniform
  return *(const e1f4*)((const char*)base + (size_t)(idx * 4u));
}
// four values of row `base` at the samples j .. j + 3; samples j + v < 0 do not exist (zeros): only the group that holds the
// window's first sample can be partial -- a rarely taken branch with guarded scalar loads
__device__ __forceinline__ void e1_load4(const float* base, int j, float (&o)[E1_V]) {
  if (j >= 0) {
    const e1f4 w = e1_ld(base, (unsigned)j);
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
    return;
  }
#pragma unroll
  for (int v = 0; v < E1_V; ++v) o[v] = j + v >= 0 ? base[j + v] : 0.f;
}
// The band's group signals live in the pair-interleaved store tau2 (signal s = row s >> 1, component s & 1).  A band's G <= 4
// signals s0 .. s0 + G - 1 always lie inside TWO consecutive pair rows -- four SLOTS (row A .x, .y, row B .x, .y) starting at
// signal s0 & ~1 -- whatever the parity of s0 (an odd G puts every second band on an odd signal: the reference's own layout,
// N = 12 = 3 groups x 4 lines, run_subband_training_treble.py:109): the kernel works on slots, with gain 0 on a slot that
// belongs to a neighbouring band, and every load is a 16-byte load of two samples x two signals.  (Round 5 fell back to
// 4-byte loads for an odd G: 206 us for the launch instead of 100 at 8 bands x 3 groups.)
// Slot pair h (row A: h = 0, row B: h = 1) at the samples j .. j + 3; samples j + v < 0 do not exist (zeros).
__device__ __forceinline__ void e1_load_tau2(const float* tp, int j, float (&ta)[E1_V], float (&tb)[E1_V]) {
  if (j >= 0) {
    const e1f4 q0 = e1_ld(tp, 2u * (unsigned)j), q1 = e1_ld(tp, 2u * (unsigned)j + 4u);
    ta[0] = q0.x; tb[0] = q0.y; ta[1] = q0.z; tb[1] = q0.w;
    ta[2] = q1.x; tb[2] = q1.y; ta[3] = q1.z; tb[3] = q1.w;
    return;
  }
#pragma unroll
  for (int v = 0; v < E1_V; ++v) {
    const bool in = j + v >= 0;
    ta[v] = in ? tp[2 * (j + v)] : 0.f;
    tb[v] = in ? tp[2 * (j + v) + 1] : 0.f;
  }
}

// (the compiler must not recognise the per-tile sample indices of one phase in the next one: kept alive across the phases
// they cost more registers than recomputing them)
__device__ __forceinline__ int e1_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// exclusive prefix (REV: suffix) sums of the E1_NT x E1_W table of wave totals in its flat order, by wave 0
static_assert(E1_NT * E1_W == 3 * 64, "e1_table_scan: three entries per lane");
template <bool REV>
__device__ __forceinline__ void e1_table_scan(const float* s_w, float* s_c, int w, int lane) {
  if (w != 0) return;
  const float a0 = s_w[3 * lane], a1 = s_w[3 * lane + 1], a2 = s_w[3 * lane + 2];
  float tot;
  if (!REV) {
    float prev = __shfl_up((a0 + a1) + a2, 1, 64);
    if (lane == 0) prev = 0.f;
    const float pre = wave_scan_incl(prev, tot);             // the entries of the lanes in front
    s_c[3 * lane] = pre;
    s_c[3 * lane + 1] = pre + a0;
    s_c[3 * lane + 2] = (pre + a0) + a1;
  } else {
    float next = __shfl_down((a2 + a1) + a0, 1, 64);
    if (lane == 63) next = 0.f;
    const float post = wave_scan_incl_rev(next, tot);        // the entries of the lanes behind
    s_c[3 * lane + 2] = post;
    s_c[3 * lane + 1] = post + a2;
    s_c[3 * lane] = (post + a2) + a1;
  }
}

// (probe builds only, tools/build_probe_lib.sh ... -DE1_TIMING: wall-clock stamps of the phases, 100 MHz)
#ifdef E1_TIMING
__device__ unsigned long long e1_times[1024 * 8];
#define E1_STAMP(slot)                                                                                                   \
  do {                                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 1024) e1_times[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
extern "C" int gfdn_probe_edc_one_times(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(e1_times), sizeof(unsigned long long) * n);
}
#else
#define E1_STAMP(slot) do { } while (0)
#endif

__global__ __launch_bounds__(E1_T) void k_edc_lin_one(Edc1Args a) {
  extern __shared__ __attribute__((aligned(16))) float e1_lds[];   // [E1_NL][E1_T][E1_V] staged dL/dEDC | scan tables
  e1q* s_gq = (e1q*)e1_lds;                        // slot (k, tid): the dL/dEDC values of the thread's group
  float* s_w = e1_lds + (size_t)E1_NL * E1_V * E1_T;            // [E1_NT][E1_W]
  float* s_c = s_w + E1_NT * E1_W;                              // [E1_NT][E1_W] sums in front of (tile, wave)
  float* s_red = s_c + E1_NT * E1_W;                            // [E1_W][8]
  // XCD-aware item map: workgroups are dealt round-robin over the 8 XCDs by linear id; consecutive ITEMS go to one XCD, so
  // that an XCD's L2 holds the group signals of one or two bands instead of all of them
  int item = blockIdx.x;
  if ((a.items & 7) == 0) {
    const int per = a.items >> 3;
    item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  E1_STAMP(0);
  const int band = item / a.B, G = a.G;
  const int L = a.item_len ? a.item_len[item] : a.max_len;
  const float* xrow = a.xd + (size_t)(a.xrows ? a.xrows[item] : item) * a.ld_xd + a.start;
  const float* trow = a.Tdb + (size_t)(a.trows ? a.trows[item] : item) * a.ld_T;
  const float* mrow = a.maskw ? a.maskw + (size_t)band * a.ld_mask : nullptr;
  const int s0 = band * G, sh = s0 & 1, Gs = G + sh;         // slots in use (<= 4: the launcher checks)
  const float* t01 = (const float*)(a.tau2 + (size_t)(s0 >> 1) * a.ld_tau + a.start);
  const float* t23 = t01 + 2 * (size_t)a.ld_tau;
  float rg[E1_MAXG];                                          // gains by SLOT: zero on a neighbouring band's signal
#pragma unroll
  for (int g = 0; g < E1_MAXG; ++g) rg[g] = (g >= sh && g - sh < G) ? a.rgain[(size_t)item * G + g - sh] : 0.f;

  // scan element e = k TILE + tid V + u  <->  window sample j = L - 1 - e: the thread's group of tile k holds the samples
  // jlo .. jlo + 3 (sample order v), jlo = L - 4 - k TILE - tid V; samples j < 0 do not exist (zeros)
  const int jtop = L - E1_V - tid * E1_V;
  float xs[E1_NT][E1_V], ex[E1_NT];
  {
    const int jb = e1_opaque(jtop);
#pragma unroll
    for (int k = 0; k < E1_NT; ++k) {
      const int jlo = jb - k * E1_TILE;
      ex[k] = 0.f;
      if (jlo > -E1_V) {
        e1_load4(xrow, jlo, xs[k]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (2 * h < Gs) {
            float ta[E1_V], tb[E1_V];
            e1_load_tau2(h ? t23 : t01, jlo, ta, tb);
#pragma unroll
            for (int v = 0; v < E1_V; ++v) xs[k][v] += rg[2 * h] * ta[v] + rg[2 * h + 1] * tb[v];
          }
        }
        float run = 0.f;
#pragma unroll
        for (int u = 0; u < E1_V; ++u) run += xs[k][E1_V - 1 - u] * xs[k][E1_V - 1 - u];
        ex[k] = run;
      } else {
#pragma unroll
        for (int v = 0; v < E1_V; ++v) xs[k][v] = 0.f;
      }
      // (one workgroup = 1024 threads per CU: one tile's loads per thread in flight saturate the memory system; the
      // scheduler hoisting several tiles' loads only cost registers)
      if ((k % E1_SB) == E1_SB - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- suffix sums of x^2 (prefix over the scan elements): wave level on the VALU, waves and tiles through s_w
#pragma unroll
  for (int k = 0; k < E1_NT; ++k) {
    float tot;
    const float inc = wave_scan_incl(ex[k], tot);
    if (lane == 0) s_w[k * E1_W + w] = tot;
    ex[k] = inc - ex[k];                           // elements of the wave in front of this thread's group
  }
  __syncthreads();
  E1_STAMP(1);
  // s_c[k W + wv] = the sum of the wave totals in front of (tile k, wave wv) = the exclusive prefix of the E1_NT x E1_W
  // table in its flat order: ONE wave, three consecutive entries per lane, the lanes' sums scanned on the VALU.  (Round 5:
  // thread (k, wv) walked its up to 191 predecessors itself -- dependent LDS reads, 2.7 + 3.4 us of a 46 us launch by the
  // wall-clock stamps of a probe build; now 0.3 + 0.4.)
  e1_table_scan<false>(s_w, s_c, w, lane);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < E1_NT; ++k) ex[k] += s_c[k * E1_W + w];
  E1_STAMP(2);
  // ---- EDC, dB, |difference|, dL/dEDC (staged: tiles < E1_NL in LDS, the others in registers)
  float gq[E1_NT - E1_NL][E1_V];
  float acc = 0.f;
  // (the dB stage is the launch's largest block of arithmetic -- SQ counters of round 6: 3740 vector instructions per wave,
  // the vector unit busy in 64 % of the launch's cycles -- so it is written for instruction count: hardware log2 and
  // reciprocal, 1 ulp each (round 5's __frcp_rn is a full division: 11 instructions behind a branch per sample), one
  // compare shared by the -200 dB floor and its gradient, the constants folded into the time weight, the sign of the
  // difference copied as a bit, no per-sample branch)
  const float gneg = -(a.inv_count * a.gscale) * TEN_OVER_LN10;
  {
    const int jb = e1_opaque(jtop);
#pragma unroll
    for (int k = 0; k < E1_NT; ++k) {
      const int jlo = jb - k * E1_TILE;
      float gv[E1_V] = {0.f, 0.f, 0.f, 0.f};
      if (jlo > -E1_V) {
        float t4[E1_V], m4[E1_V] = {1.f, 1.f, 1.f, 1.f};
        e1_load4(trow, jlo, t4);
        if (mrow) e1_load4(mrow, jlo, m4);
        if (jlo < 0) {
          // (the group that holds the window's first sample: a sample that does not exist is x = 0 with time weight 0 --
          // the arithmetic below needs no per-sample branch)
#pragma unroll
          for (int v = 0; v < E1_V; ++v) if (jlo + v < 0) m4[v] = 0.f;
        }
        float run = 0.f;
#pragma unroll
        for (int u = 0; u < E1_V; ++u) {
          const int v = E1_V - 1 - u;
          run += xs[k][v] * xs[k][v];
          const float lin = fabsf(ex[k] + run) + F32_EPS;
          const float raw = 3.0102999566398120f * __builtin_amdgcn_logf(lin);        // 10 log10(lin)
          const bool above = raw > -200.0f;
          const float diff = t4[v] - (above ? raw : -200.0f);
          acc += m4[v] * fabsf(diff);
          // dL/dEDC = -sign(diff) (10 / ln 10) / lin x weight x scale; zero on the floor and where the difference vanishes
          const float mag = __builtin_amdgcn_rcpf(lin) * (m4[v] * gneg);
          const float sm = __uint_as_float((__float_as_uint(diff) & 0x80000000u) ^ __float_as_uint(mag));   // sign(diff) mag
          gv[v] = (above && diff != 0.f) ? sm : 0.f;
        }
      }
      ex[k] = ((gv[0] + gv[1]) + gv[2]) + gv[3];   // the group's sum, in sample order
      if (k < E1_NL) {
        e1q q;
        q.x = gv[0]; q.y = gv[1]; q.z = gv[2]; q.w = gv[3];
        s_gq[(size_t)k * E1_T + tid] = q;
      } else {
#pragma unroll
        for (int v = 0; v < E1_V; ++v) gq[k >= E1_NL ? k - E1_NL : 0][v] = gv[v];
      }
      if ((k % E1_SB) == E1_SB - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (a.gx || a.dots) {
    // ---- prefix sums of dL/dEDC in SAMPLE order: tiles descending, threads descending, samples ascending
#pragma unroll
    for (int k = 0; k < E1_NT; ++k) {
      float tot;
      const float inc = wave_scan_incl_rev(ex[k], tot);
      if (lane == 0) s_w[k * E1_W + w] = tot;
      ex[k] = inc - ex[k];                         // groups of the wave with smaller sample indices (higher lanes)
    }
    __syncthreads();
    E1_STAMP(3);
    e1_table_scan<true>(s_w, s_c, w, lane);              // (sums BEHIND (tile, wave), walking down from the last entry)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E1_NT; ++k) ex[k] += s_c[k * E1_W + w];
    E1_STAMP(4);
    // ---- dL/dx = 2 x cum, stored once; the EDC part of dL/dgain as dot products with the group signals
    float d[E1_MAXG] = {0.f, 0.f, 0.f, 0.f};
    float* grow = a.gx ? a.gx + (size_t)item * a.ld_gx : nullptr;
    const int jb = e1_opaque(jtop);
#pragma unroll
    for (int k = 0; k < E1_NT; ++k) {
      const int jlo = jb - k * E1_TILE;
      if (jlo > -E1_V) {
        float gv[E1_V], out[E1_V];
        if (k < E1_NL) {
          const e1q q = s_gq[(size_t)k * E1_T + tid];
          gv[0] = q.x; gv[1] = q.y; gv[2] = q.z; gv[3] = q.w;
        } else {
#pragma unroll
          for (int v = 0; v < E1_V; ++v) gv[v] = gq[k >= E1_NL ? k - E1_NL : 0][v];
        }
        float run = 0.f;
#pragma unroll
        for (int v = 0; v < E1_V; ++v) {
          run += gv[v];
          out[v] = 2.0f * xs[k][v] * (ex[k] + run);
        }
        if (grow) {
          if (jlo >= 0) {
            e1f4 o4;
            o4.x = out[0]; o4.y = out[1]; o4.z = out[2]; o4.w = out[3];
            *(e1f4*)((char*)grow + (size_t)((unsigned)jlo * 4u)) = o4;
          } else {
#pragma unroll
            for (int v = 0; v < E1_V; ++v) if (jlo + v >= 0) grow[jlo + v] = out[v];
          }
        }
        if (a.dots) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (2 * h < Gs) {
              float ta[E1_V], tb[E1_V];
              e1_load_tau2(h ? t23 : t01, jlo, ta, tb);
#pragma unroll
              for (int v = 0; v < E1_V; ++v) {
                d[2 * h] += out[v] * ta[v];
                d[2 * h + 1] += out[v] * tb[v];
              }
            }
          }
        }
      }
      if ((k % E1_SB) == E1_SB - 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (a.dots) {
#pragma unroll
      for (int g = 0; g < E1_MAXG; ++g) {
        const float s = wave_sum_full(d[g]);
        if (lane == 0) s_red[w * 8 + 1 + g] = s;
      }
    }
  }
  {
    const float s = wave_sum_full(acc);
    if (lane == 0) s_red[w * 8] = s;
  }
  __syncthreads();
  E1_STAMP(5);
  if (tid < 1 + E1_MAXG) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < E1_W; ++i) s += s_red[i * 8 + tid];
    if (tid == 0) a.loss_item[item] = s * a.inv_count;
    else if (a.dots && tid - 1 >= sh && tid - 1 - sh < G)     // (slot tid - 1 = group tid - 1 - sh)
      a.dots[((size_t)item * G + (tid - 1 - sh)) * a.ld_dots + a.col] = s;
  }
}

extern "C" int gfdn_edc_lin_one_max_len(void) { return E1_NT * E1_TILE; }

extern "C" int gfdn_edc_lin_one(const float* xd, int ld_xd, const long long* xrows, const float* tau2, int ld_tau,
                                const float* rgain, int nbands, int B, int G, int start, int max_len, const int* item_len,
                                const float* T_db, int ld_T, const long long* target_rows, const float* maskw, int ld_mask,
                                float inv_count, float gscale, float* loss_item, float* gx, int ld_gx, float* dots,
                                int ld_dots, int col, void* stream) {
  if (!xd || !tau2 || !rgain || !T_db || !loss_item || nbands <= 0 || B <= 0 || G <= 0 || start < 0 || max_len <= 0 ||
      ld_xd < start + max_len || ld_tau < start + max_len || ld_T < max_len || ld_mask < 0 ||
      (maskw && ld_mask > 0 && ld_mask < max_len) || (gx && ld_gx < max_len) || (dots && (ld_dots <= 0 || col < 0 || col >= ld_dots)))
    return GFDN_E_BADARG;
  if (G > E1_MAXG || max_len > E1_NT * E1_TILE) return GFDN_E_UNSUPPORTED;
  Edc1Args a{xd, ld_xd, xrows, (const float2*)tau2, ld_tau, rgain, B, G, nbands * B, start, max_len, item_len, T_db, ld_T,
             target_rows, maskw, ld_mask, inv_count, gscale, loss_item, gx, ld_gx, dots, ld_dots, col};
  const size_t lds = ((size_t)E1_NL * E1_V * E1_T + 2 * E1_NT * E1_W + E1_W * 8) * sizeof(float);
  // (G = 4: every band starts on an even signal; G <= 3: at most one slot of a neighbouring band in front -- four slots)
  int rc = ensure_dyn_lds(k_edc_lin_one, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_edc_lin_one, dim3(nbands * B), dim3(E1_T), lds, (hipStream_t)stream, a);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// gamma[band G + g][pos(t)] = base_a[band G + g][t] + base_b[..][t] + sum_{b in band} rgain[b][g] gx[b][t - w0]   (t in the
// band's window; the bases alone outside it), pos = the adjoint pair transform's slot order when slot_of_time is given.
// gx (items, ld_g): window-only rows as k_edc_lin_one leaves them; gamma / bases pair-interleaved (ceil(S / 2), ld, 2).
// One workgroup = one band x 1024 samples, the band's receivers in index order, eight rows in flight.
#define GW_V 4
#define GW_U 8
__global__ __launch_bounds__(256) void k_lin_gamma_win(const float* __restrict__ gx, int ld_g,
                                                       const float* __restrict__ rgain, int B, int G, int n, int w0,
                                                       int wlen, const int* __restrict__ wlen_band,
                                                       const float* __restrict__ base_a, const float* __restrict__ base_b,
                                                       int ld_b, const int* __restrict__ slot_of_time,
                                                       float* __restrict__ gamma, int ld_o) {
  __shared__ float s_rg[256];
  const int band = blockIdx.y;
  for (int i = threadIdx.x; i < B * G; i += 256) s_rg[i] = rgain[(size_t)band * B * G + i];
  __syncthreads();
  if (wlen_band) wlen = wlen_band[band];
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * GW_V;
  if (t0 >= n) return;
  float acc[E1_MAXG][GW_V];
#pragma unroll
  for (int g = 0; g < E1_MAXG; ++g)
#pragma unroll
    for (int u = 0; u < GW_V; ++u) acc[g][u] = 0.f;
  const int j0 = t0 - w0;
  const bool any = j0 + GW_V > 0 && j0 < wlen;
  const bool full = j0 >= 0 && j0 + GW_V <= wlen;
  if (any) {
    const float* g0 = gx + (size_t)band * B * ld_g;
    for (int b0 = 0; b0 < B; b0 += GW_U) {
      float q[GW_U][GW_V];
#pragma unroll
      for (int r = 0; r < GW_U; ++r) {
        const int b = b0 + r;
        if (b < B) {
          if (full) ld4_f(g0 + (size_t)b * ld_g + j0, q[r]);
          else {
#pragma unroll
            for (int u = 0; u < GW_V; ++u) q[r][u] = (j0 + u >= 0 && j0 + u < wlen) ? g0[(size_t)b * ld_g + j0 + u] : 0.f;
          }
        } else {
#pragma unroll
          for (int u = 0; u < GW_V; ++u) q[r][u] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < GW_U; ++r) {
        const int b = b0 + r;
        if (b < B) {
#pragma unroll
          for (int g = 0; g < E1_MAXG; ++g) {
            if (g < G) {
              const float ra = s_rg[b * G + g];
#pragma unroll
              for (int u = 0; u < GW_V; ++u) acc[g][u] += ra * q[r][u];
            }
          }
        }
      }
    }
  }
  if (base_a) {
#pragma unroll
    for (int g = 0; g < E1_MAXG; ++g) {
      if (g < G) {
        const int s = band * G + g;
#pragma unroll
        for (int u = 0; u < GW_V; ++u)
          if (t0 + u < n) {
            const size_t o = ((size_t)(s >> 1) * ld_b + t0 + u) * 2 + (s & 1);
            acc[g][u] += base_b ? base_a[o] + base_b[o] : base_a[o];
          }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < GW_V; ++u) {
    const int t = t0 + u;
    if (t < n) {
      const size_t pos = slot_of_time ? (t == 0 ? 0 : 1 + (size_t)slot_of_time[t]) : (size_t)t;
      if (!((band * G) & 1) && !(G & 1)) {
#pragma unroll
        for (int g = 0; g < E1_MAXG; g += 2)
          if (g < G)
            ((float2*)gamma)[(size_t)((band * G + g) >> 1) * ld_o + pos] = make_float2(acc[g][u], acc[g + 1][u]);
      } else {
#pragma unroll
        for (int g = 0; g < E1_MAXG; ++g) {
          if (g < G) {
            const int s = band * G + g;
            gamma[((size_t)(s >> 1) * ld_o + pos) * 2 + (s & 1)] = acc[g][u];
          }
        }
      }
    }
  }
}

extern "C" int gfdn_lin_gamma_win(const float* gx, int ld_g, const float* rgain, int nbands, int B, int G, int n,
                                  int win_start, int win_len, const int* band_win_len, const float* base2a,
                                  const float* base2b, int ld_b, const int* slot_of_time, float* gamma2, int ld_o,
                                  void* stream) {
  if (!gx || !rgain || !gamma2 || nbands <= 0 || B <= 0 || G <= 0 || n <= 0 || win_start < 0 || win_len <= 0 ||
      win_start + win_len > n || ld_g < win_len || ld_o < n || (base2a && (ld_b < n || base2a == gamma2)) ||
      (base2b && (!base2a || base2b == gamma2)))
    return GFDN_E_BADARG;
  if (G > E1_MAXG || B * G > 256 || nbands > 65535) return GFDN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_lin_gamma_win, dim3((n + 256 * GW_V - 1) / (256 * GW_V), nbands), dim3(256), 0, (hipStream_t)stream,
                     gx, ld_g, rgain, B, G, n, win_start, win_len, band_win_len, base2a, base2b, ld_b, slot_of_time, gamma2,
                     ld_o);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// out[r][pos(t)] = a[r][t] + b[r][t] + c[r][t] for pair-interleaved signal rows (float2 per sample; b, c optional), pos = the
// adjoint pair transform's slot order (sample 0 first, the sample of time t >= 1 at 1 + slot_of_time[t]): the parts of
// dL/dtau that different launches left (EDC part summed over the receivers, the adjoint STFT's even / odd frames) merged
// and permuted in one small pass in front of the adjoint transform.
__global__ __launch_bounds__(256) void k_lin_merge_slots(const float2* __restrict__ a, const float2* __restrict__ b,
                                                         const float2* __restrict__ c, int rows, int n, int ld,
                                                         const int* __restrict__ slot_of_time, float2* __restrict__ out,
                                                         int ld_o) {
  // XCD-aware map (workgroups are dealt round-robin over the 8 XCDs by linear id, each XCD has its own L2): all tiles of a
  // signal row get ids of one residue class mod 8, so that the 8-byte slots a row's tiles scatter over its output (512 KB)
  // meet in ONE L2 and lines leave it complete -- with the plain (tile, row) grid PMC showed 56.6 MB for 29 MB algorithmic
  const int ntile = (n + 1023) / 1024;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int r = xcd + 8 * (jj / ntile), tile = jj - (jj / ntile) * ntile;
  if (r >= rows) return;
  const int t0 = (tile * 256 + threadIdx.x) * 4;
  if (t0 >= n) return;
  const size_t o = (size_t)r * ld;
  float2 v[4];
  if (t0 + 4 <= n) {
    ld4_f2(a + o + t0, v);
    if (b) {
      float2 w[4];
      ld4_f2(b + o + t0, w);
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[u].x += w[u].x; v[u].y += w[u].y; }
    }
    if (c) {
      float2 w[4];
      ld4_f2(c + o + t0, w);
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[u].x += w[u].x; v[u].y += w[u].y; }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = make_float2(0.f, 0.f);
      if (t0 + u < n) {
        v[u] = a[o + t0 + u];
        if (b) { v[u].x += b[o + t0 + u].x; v[u].y += b[o + t0 + u].y; }
        if (c) { v[u].x += c[o + t0 + u].x; v[u].y += c[o + t0 + u].y; }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = t0 + u;
    if (t < n) out[(size_t)r * ld_o + (slot_of_time ? (t == 0 ? 0 : 1 + slot_of_time[t]) : t)] = v[u];
  }
}

extern "C" int gfdn_lin_merge_slots(const float* a2, const float* b2, const float* c2, int rows, int n, int ld,
                                    const int* slot_of_time, float* out2, int ld_o, void* stream) {
  if (!a2 || !out2 || rows <= 0 || n <= 0 || ld < n || ld_o < n || out2 == a2 || out2 == b2 || out2 == c2 || (c2 && !b2))
    return GFDN_E_BADARG;
  if (rows > 4096) return GFDN_E_UNSUPPORTED;
  const int ntile = (n + 1023) / 1024, rows8 = (rows + 7) / 8 * 8;
  hipLaunchKernelGGL(k_lin_merge_slots, dim3(ntile * rows8), dim3(256), 0, (hipStream_t)stream, (const float2*)a2,
                     (const float2*)b2, (const float2*)c2, rows, n, ld, slot_of_time, (float2*)out2, ld_o);
  GFDN_LAUNCH_CHECK();
  return 0;
}
