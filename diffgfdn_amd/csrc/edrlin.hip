// The EDR loss on LINEARLY COMPOSED short-time spectra, for gfx950.
//
// Reference maths (orchidas/DiffGFDN, src/diff_gfdn/losses.py:430-495, :501-575): the EDR loss takes the STFT (Hann 4096,
// hop 2048) of x[b] = irfft(H[b], n = K), its power |S|^2, Schroeder-integrates it over the frames, compares in dB.
// With the output stage in the time domain (linear.hip), x[b] = xd[row_b] + sum_g gain[b][g] tau_g, and the STFT is linear:
//        S[b] = STFT(xd[row_b]) + sum_g gain[b][g] STFT(tau_g) = Sd[row_b] + sum_g gain[b][g] Stau_g .
// Sd is a constant of the dataset (one STFT per receiver, once, like the decay targets) and Stau are G short STFTs per
// band and step.  The 224 forward STFTs and the 224 adjoint STFTs of a step (two 4096-point FFTs per frame pair each way)
// do not run; what runs per receiver is streaming arithmetic on (frame, frequency) cells:
//   k_edr_lin_cols : per (receiver, frequency): compose S over the frames, |S|^2, tail sums, dB, |difference| -> loss
//                    partials, dL/d|S|^2 (gP, real) and -- in the same descending sweep -- the EDR part of
//                    dL/dgain[b][g] = Re sum_cells conj(Stau_g) dL/dS,  dL/dS = 2 gP S   (tail sums of Re(conj(Stau_g) S))
//   k_edr_lin_gsum : per (band, cell): Gsum_g = sum_b gain[b][g] dL/dS[b] -- the G gradient spectra per band whose adjoint
//                    STFT (k_stft4k_pair_spec_bwd, 14 signal pairs) is the EDR part of dL/dtau_g
//   k_edr_lin_wave : both of the above in ONE launch (the training step's): threads own cells of the band's plane and walk
//                    the band's receivers; dL/d|S|^2 never reaches memory, the direct-path spectra are read once
//   k_stft4k_pair_spec(_bwd) : STFT of pair-interleaved signals to complex one-sided spectra, and its adjoint from
//                    gradient spectra (the forms of fft.hip's k_stft4k_pair_power(_bwd) without the |.|^2 stage).
// All sums in fixed order (bitwise reproducible).
#include "common.h"
#include "fft4k_dev.h"

extern __shared__ float2 edl_lds[];

#define EDL_RF 32
#define EDL_MAXG 4

// Cell order of the (nframes, nfreq) planes.  Plain: cell = m nfreq + f.  TILED: the frequencies are cut into blocks of 256
// and a block's frames are contiguous -- cell = fb nframes 256 + m w + (f - 256 fb), w = the block's width (256; the last
// block holds the rest) -- so that everything a (receiver, frequency-block) workgroup of k_edr_lin_cols touches is ONE
// contiguous run (64 KB of Sd, 32 KB of target, 32 KB of dL/d|S|^2) instead of 32 runs of 2 KB / 1 KB at a 16 KB / 8 KB
// stride.  The planes have the same number of cells either way.
__device__ __forceinline__ size_t edl_cell(int m, int f, int nframes, int nfreq, int tiled) {
  if (!tiled) return (size_t)m * nfreq + f;
  const int fb = f >> 8, w = nfreq - (fb << 8) < 256 ? nfreq - (fb << 8) : 256;
  return (size_t)fb * nframes * 256 + (size_t)m * w + (f & 255);
}

struct EdrLin {
  const float2* Sd;            // (R, nframes, nfreq) complex: STFT of the transformed direct paths
  const long long* rows;       // item -> row of Sd, Tdb, sum_abs (NULL: identity)
  const float2* Stau;          // (nbands * G, nframes, nfreq) complex: STFT of the band's group signals
  const float* rgain;          // (items, G)
  int B, G;
  const float* Tdb;            // (R, nframes, nfreq) target EDR in dB
  const float* sum_abs;        // (R) sum |target EDR|
  int tiled;                   // cell order of Sd, Stau, Tdb, gP, Gsum (edl_cell)
};

// One thread per (receiver, frequency) column, one descending sweep over the frames (see the body).
// The training step runs k_edr_lin_wave below instead (receivers summed in the launch); this kernel serves the value-only pass
// (validation) and is the cross-check of the tests.
// (Measured alternatives, same box: the group spectra of a 64-frequency tile staged in LDS for 8 receivers per workgroup --
// 64 KB, two workgroups per CU -- 175 us against 99; the whole band per workgroup with the receivers' gradient spectra summed
// through a 64 KB LDS exchange beside 64 KB of staged group spectra -- one workgroup per CU -- 213 us, and it starves the
// colorless pass beside it of LDS: both keep far fewer loads in flight than 2016 independent workgroups do.)
//   part[b][fblk] = sum_{f in block, m} |T - EDR| (to be divided by sum_abs, as gfdn_edr_loss(defer))
//   gP (items, nframes, nfreq) = gscale / sum_abs dloss/d|S|^2
//   dots[(b G + g) ld_dots + col0 + fblk] = partial of the EDR part of dL/dgain[b][g]
#define EDL_FT 64
__global__ __launch_bounds__(256) void k_edr_lin_cols(EdrLin a, int nframes, int nfreq, float gscale, int want_grad,
                                                      float* __restrict__ gP, float* __restrict__ part,
                                                      float* __restrict__ dots, int ld_dots, int col0, int nslices) {
  __shared__ float s_red[16];
  // XCD-aware map: workgroups are dealt round-robin over the 8 XCDs by linear id, and each XCD has its own 4 MB L2.  The B
  // receivers that share a (band, 256-frequency block) slice of the group spectra (G x nframes x 256 complex = 262 KB) are
  // given ids of ONE residue class mod 8, slice after slice -- with the plain (block, receiver) grid every XCD touched every
  // slice (14.7 MB of group spectra against a 4 MB L2: 0.9 GB of refetches per launch, 309 us, and the EDC scans beside
  // it slowed fourfold).
  const int G = a.G, B = a.B, fblk = (nfreq + 255) / 256;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slice = xcd + 8 * (j / B), bl = j - (j / B) * B;
  if (slice >= nslices) return;
  const int band = slice / fblk, fb = slice - band * fblk;
  const int b = band * B + bl;
  const size_t row = a.rows ? (size_t)a.rows[b] : (size_t)b;
  const size_t cells = (size_t)nframes * nfreq;
  float rg[EDL_MAXG];
#pragma unroll
  for (int g = 0; g < EDL_MAXG; ++g) rg[g] = g < G ? a.rgain[(size_t)b * G + g] : 0.f;
  const float gs = want_grad ? gscale / a.sum_abs[row] : 0.f;
  float acc = 0.f, dacc[EDL_MAXG] = {0.f, 0.f, 0.f, 0.f};
  const int f = fb * 256 + threadIdx.x;
  if (f < nfreq) {
    // (column walk: cell(m) = c0 + m cs)
    const size_t c0 = edl_cell(0, f, nframes, nfreq, a.tiled);
    const size_t cs = nframes > 1 ? edl_cell(1, f, nframes, nfreq, a.tiled) - c0 : 0;
    const float2* sd = a.Sd + row * cells + c0;
    const float2* st = a.Stau + (size_t)band * G * cells + c0;
    const float* t = a.Tdb + row * cells + c0;
    // ONE descending sweep over the frames: the frame's spectra (direct path, G group spectra) and target are loaded where
    // they are used -- the compiler keeps the loads of the next frames in flight across the dB chain of the current one --
    // S composed in registers, tail energy, dB, |difference|, dL/dE_m; and with the same loads the EDR part of
    //   dL/dgain[b][g] = 2 sum_m gP_m Re(conj(Stau_g[m]) S[m]),  gP_m = sum_{m' <= m} dL/dE_m'
    //                  = 2 sum_m' dL/dE_m' sum_{m >= m'} Re(conj(Stau_g[m]) S[m])        (tail sums, like the energy)
    // (Measured: with the 64 HBM loads issued up front and the group spectra read in two separate passes the launch took
    // 282-310 us instead of 99 and slowed the EDC scans beside it fourfold: twice the cache traffic in 16 KB strides.)
    float gE[EDL_RF];
    float E = 0.f, ct[EDL_MAXG] = {0.f, 0.f, 0.f, 0.f};
    // (Measured, alone on the chip, 235 MB: this form 79 us at 112 registers, 4 waves per SIMD; the sweep in chunks of 8
    // frames whose loads are issued before the chunk's dependent chain: 196 registers, 2 waves, 89 us; chunks of 4 under a
    // 128-register cap: spills, 111 us.)
#pragma unroll
    for (int m = EDL_RF - 1; m >= 0; --m) {
      gE[m] = 0.f;
      if (m < nframes) {
        float2 sv = sd[(size_t)m * cs];
        float2 tg[EDL_MAXG];
#pragma unroll
        for (int g = 0; g < EDL_MAXG; ++g) {
          tg[g] = g < G ? st[(size_t)g * cells + (size_t)m * cs] : make_float2(0.f, 0.f);
          sv.x += rg[g] * tg[g].x;
          sv.y += rg[g] * tg[g].y;
        }
        const float tvm = t[(size_t)m * cs];
        E += sv.x * sv.x + sv.y * sv.y;
        const float lin = fabsf(E) + F32_EPS;
        const float raw = 10.0f * log10f(lin);
        const float d = fmaxf(raw, -200.0f);
        const float diff = tvm - d;
        acc += fabsf(diff);
        const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
        const float dE = (raw > -200.0f) ? TEN_OVER_LN10 / lin : 0.f;
        const float ge = -sg * dE * gs;
        gE[m] = ge;
#pragma unroll
        for (int g = 0; g < EDL_MAXG; ++g) {
          ct[g] += tg[g].x * sv.x + tg[g].y * sv.y;
          dacc[g] += ge * ct[g];
        }
      }
    }
    if (want_grad) {
      float* gp = gP + (size_t)b * cells + c0;
      float run = 0.f;
#pragma unroll
      for (int m = 0; m < EDL_RF; ++m) {
        if (m < nframes) {
          run += gE[m];                                          // dL/d|S_m|^2 = sum_{m' <= m} dL/dE_m'
          gp[(size_t)m * cs] = run;
        }
      }
    }
  }
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) part[(size_t)b * fblk + fb] = acc;
  if (want_grad && dots) {
#pragma unroll
    for (int g = 0; g < EDL_MAXG; ++g) {
      if (g < G) {
        const float v = block_sum(2.0f * dacc[g], s_red);
        if (threadIdx.x == 0) dots[((size_t)b * G + g) * ld_dots + col0 + fb] = v;
      }
    }
  }
}

// The same loss with the gradient spectra summed over the band's receivers IN the launch, without dL/d|S|^2 in memory and
// with the transformed direct paths read once (176 MB per step instead of the 426 MB of k_edr_lin_cols + k_edr_lin_gsum).
// A thread OWNS cells of the band's (frame, frequency) plane -- four consecutive frames of one frequency -- keeps the band's G
// group spectra of its cells (32 registers) and its share of the G gradient spectra (32 accumulators: registers until round 6,
// now the lane's own LDS slots) for the whole launch, and walks the band's receivers: per receiver it loads its 4 cells of the direct-path spectrum and the target,
// composes S, and runs the two scans along the frames (tail energy; prefix sums of dL/dE) inside the thread over its 4 frames
// and across the frame groups inside the WAVE (k_edr_lin_wave below).
// The receivers are added in index order: bitwise reproducible.  ``nsplit`` > 1: the receivers of a band are cut into nsplit
// runs handled by different workgroups (more loads in flight), each writing its own partial planes Gsum[split]; the adjoint
// STFT adds them on load.
// (Round-4 history: the first fused form -- one receiver per wave, the gradient spectra summed through a 64 KB LDS exchange,
// the group spectra staged in another 64 KB -- took 213 us at one workgroup per CU; the second -- the frame scans across
// eight waves through LDS and two barriers per receiver -- 75 us against this form's 59: DESIGN.md section 4.3.)
//   part[b][tile]                           : loss partials (sum |T - EDR| of the tile's cells)
//   dots[(b G + g) ld_dots + col0 + tile]   : partials of the EDR part of dL/dgain[b][g]
#define EDB_Q 4                     // frames per thread
#define EDB_DB_PER_LOG2 3.0102999566398120f      // 10 log10(x) = EDB_DB_PER_LOG2 log2(x)
// k_edr_lin_wave: a WAVE owns 8 frequencies x all 32 frames -- lane =
// (frame group fg = lane >> 3, frequency fl = lane & 7), the thread owns the frames 4 fg .. 4 fg + 3 of its frequency as
// before -- so the two scans along the frames run inside the wave on the VALU: the pair (fg, fg ^ 1) by one DPP row rotation,
// the four rows of sixteen lanes by v_permlane16_swap / v_permlane32_swap (three swaps give every lane the four row totals);
// direct sums only, no total-minus-prefix.  No LDS exchange, no __syncthreads: the waves of a workgroup (4 waves = 32 adjacent
// frequencies, so that the 128-byte lines of a frame row are shared inside the workgroup: with two or one wave per workgroup
// the launch takes 61 / 84 us instead of 50) run independently.  (The form it
// replaced -- eight waves per 64 frequencies, the scans across the waves through two 2 KB LDS exchanges and two barriers per
// receiver -- took 75 us alone against 59: DESIGN.md section 4.3.)  The partial sums are per WAVE:
// gfdn_edr_lin_band_parts(nfreq) = 4 ceil(nfreq / 32) columns per receiver.
#define EDW_F 8                     // frequencies per wave
#define EDW_WG 4                    // waves per workgroup
__device__ __forceinline__ float edw_partner(float v) {     // the value of lane ^ 8 (the other frame group of the pair)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));      // row_ror:8
}
// p = a value that is the same in the 16 lanes' two frame groups ... of every row r: returns (P0, P1, P2, P3) of the four rows
__device__ __forceinline__ void edw_rows(float p, float (&P)[4]) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
  // a[0] = [p0, p0, p2, p2] (even rows over their pair), a[1] = [p1, p1, p3, p3]
  const auto e = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);      // [p0 x4], [p2 x4]
  const auto o = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false);      // [p1 x4], [p3 x4]
  P[0] = __uint_as_float(e[0]); P[2] = __uint_as_float(e[1]);
  P[1] = __uint_as_float(o[0]); P[3] = __uint_as_float(o[1]);
}
// (probe builds only, tools/build_probe_lib.sh ... -DEDW_TIMING: wall-clock stamps per workgroup, 100 MHz)
#ifdef EDW_TIMING
__device__ unsigned long long edw_times[1024 * 4];
#define EDW_STAMP(slot)                                                                                                  \
  do {                                                                                                                   \
    const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                  \
    if (threadIdx.x == 0 && wg_ < 1024) edw_times[wg_ * 4 + (slot)] = __builtin_amdgcn_s_memrealtime();                   \
  } while (0)
extern "C" int gfdn_probe_edr_wave_times(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(edw_times), sizeof(unsigned long long) * n);
}
#else
#define EDW_STAMP(slot) do { } while (0)
#endif
// Round 6: the launch is bound by VECTOR INSTRUCTION ISSUE, not by memory (SQ counters of the round-5 form alone on the chip:
// 340 vector instructions per wave and receiver, the vector unit busy in 92 % of the waves' cycles, 176 MB in 59 us), so the
// receiver loop is written for instruction count:
//   * FULL (all 32 frames exist: the north-star shape): no frame masks -- the round-5 form guarded each of the thread's four
//     cells with a branch in the loads and another in the dB stage;
//   * two receivers per trip of the loop, the prefetched cells alternating between two register sets (no copies);
//   * gscale / sum_abs of the run's receivers computed once (a lane per receiver) and read back by lane index -- it was a full
//     division per receiver and wave;
//   * the rows that enter a thread's two scan offsets selected by 0 / 1 lane constants inside one multiply-add each;
//   * the dot products <Stau_g, dL/dS> accumulated as (x, y) pairs (packed multiply-adds) and their four wave sums taken
//     TOGETHER: two exchange steps leave every lane with ONE of the four values (15 instructions instead of 32), lanes 0..3
//     store them;
//   * the sign of the difference copied as a bit, one compare shared by the -200 dB floor and its gradient, hardware log2 and
//     reciprocal.
// Sums are in a fixed order (bitwise reproducible run to run); they are NOT the round-5 order in the dot products.
typedef float edw2 __attribute__((ext_vector_type(2)));
typedef float edw4 __attribute__((ext_vector_type(4)));
#ifndef EDW_GA_LDS
#define EDW_GA_LDS 1                // the gradient-spectrum accumulators in LDS (0: in registers)
#endif
#ifndef EDW_AHEAD2
#define EDW_AHEAD2 1                // prefetch two receivers ahead, into the register set just used up (0: one ahead)
#endif
// (every multiply-add of the receiver loop is written out: the loop holds its body twice plus a tail copy, and a compiler
// free to contract them differently would make a receiver's numbers depend on which copy it runs through)
__device__ __forceinline__ edw2 edw_fma(float s, edw2 a, edw2 c) { return __builtin_elementwise_fma(edw2{s, s}, a, c); }
__device__ __forceinline__ edw2 edw_fma2(edw2 a, edw2 b, edw2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int CTRL>
__device__ __forceinline__ float edw_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <bool FULL>
__global__ __launch_bounds__(64 * EDW_WG, 4) void k_edr_lin_wave(EdrLin a, int nframes, int nfreq, float gscale,
                                                                 float* __restrict__ part, int ld_part,
                                                                 float* __restrict__ dots, int ld_dots, int col0,
                                                                 float2* __restrict__ Gsum, int nsplit, int nbands) {
  const int G = a.G, B = a.B, band = blockIdx.y, split = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fg = lane >> 3, fl = lane & 7, row = lane >> 4;
  const bool odd = fg & 1;
  const int col = blockIdx.x * EDW_WG + wv;                 // the wave's column of partial sums
  const int f = col * EDW_F + fl;
  const bool live = f < nfreq;
  const int fc = live ? f : nfreq - 1;
  const size_t cells = (size_t)nframes * nfreq;
  unsigned cq[EDB_Q];
  bool mv[EDB_Q];
#pragma unroll
  for (int q = 0; q < EDB_Q; ++q) {
    const int m = EDB_Q * fg + q;
    mv[q] = FULL || m < nframes;
    cq[q] = (unsigned)edl_cell(mv[q] ? m : 0, fc, nframes, nfreq, a.tiled);
  }
  // (a lane beyond the last frequency works on the last column's cells with zero group spectra: its dot products vanish, its
  // loss partial is masked, its gradient planes are not stored)
#if EDW_GA_LDS
  // The thread's share of the G gradient spectra (32 accumulators) lives in LDS as eight 16-byte slots per lane, lane-linear
  // (no bank conflicts), updated by plain read - multiply-add - write (the lane owns its slots): the LDS pipe has nothing
  // else to do in this kernel, and the 32 registers hold the second set of prefetched cells instead.
  __shared__ edw4 s_ga[EDW_WG][EDL_MAXG * EDB_Q / 2][64];
  edw4 (*ga)[64] = s_ga[wv];
#else
  edw2 Ga[EDL_MAXG][EDB_Q];
#endif
  edw2 st[EDL_MAXG][EDB_Q];
#pragma unroll
  for (int g = 0; g < EDL_MAXG; ++g)
#pragma unroll
    for (int q = 0; q < EDB_Q; ++q) {
      const float2 v = (g < G && mv[q] && live) ? (a.Stau + ((size_t)band * G + g) * cells)[cq[q]] : make_float2(0.f, 0.f);
      st[g][q] = edw2{v.x, v.y};
#if EDW_GA_LDS
      if (!(q & 1)) ga[(g * EDB_Q + q) / 2][lane] = edw4{0.f, 0.f, 0.f, 0.f};
#else
      Ga[g][q] = edw2{0.f, 0.f};
#endif
    }
  const int bper = (B + nsplit - 1) / nsplit;
  const int b_lo = split * bper, b_hi = b_lo + bper < B ? b_lo + bper : B;
  // lane constants of the two scans: which row totals lie behind / in front of this thread's frames
  const float m_lt1 = row < 1 ? 1.f : 0.f, m_lt2 = row < 2 ? 1.f : 0.f, m_lt3 = row < 3 ? 1.f : 0.f;
  const float m_gt0 = row > 0 ? 1.f : 0.f, m_gt1 = row > 1 ? 1.f : 0.f, m_gt2 = row > 2 ? 1.f : 0.f;
  const bool bit0 = lane & 1, bit1 = lane & 2;
  // lanes 0 .. 3 end up with the dot products of the groups 0, 2, 1, 3 (see the exchange below)
  const int gl = ((lane & 1) << 1) | ((lane >> 1) & 1);
  edw2 sA[EDB_Q], sB[EDB_Q];
  float tA[EDB_Q], tB[EDB_Q];
  auto fetch = [&](int bl, edw2 (&sn)[EDB_Q], float (&tn)[EDB_Q]) {
    const int b = band * B + bl;
    const size_t rw = a.rows ? (size_t)a.rows[b] : (size_t)b;
    const float2* sdr = a.Sd + rw * cells;
    const float* tdr = a.Tdb + rw * cells;
#pragma unroll
    for (int q = 0; q < EDB_Q; ++q) {
      if (FULL || mv[q]) {
        const float2 v = sdr[cq[q]];
        sn[q] = edw2{v.x, v.y};
        tn[q] = tdr[cq[q]];
      } else {
        sn[q] = edw2{0.f, 0.f};
        tn[q] = 0.f;
      }
    }
  };
  float gs_tab = 0.f;                  // lane l: -(10 / ln 10) gscale / sum_abs of the receiver (bl & ~63) + l of this run
  auto body = [&](int bl, edw2 (&sc)[EDB_Q], float (&tc)[EDB_Q], edw2 (&sn)[EDB_Q], float (&tn)[EDB_Q]) {
#pragma clang fp contract(off)
    const int b = band * B + bl;
    if (((bl - b_lo) & 63) == 0) {
      const int bt = bl + lane < b_hi ? bl + lane : b_hi - 1;
      const size_t rt = a.rows ? (size_t)a.rows[band * B + bt] : (size_t)(band * B + bt);
      gs_tab = -TEN_OVER_LN10 * (gscale / a.sum_abs[rt]);
    }
    const float gsn = readlane_f(gs_tab, (bl - b_lo) & 63);
    float rg[EDL_MAXG];
#pragma unroll
    for (int g = 0; g < EDL_MAXG; ++g) rg[g] = g < G ? a.rgain[(size_t)b * G + g] : 0.f;
    edw2 sv[EDB_Q];
    float pw[EDB_Q];
#pragma unroll
    for (int q = 0; q < EDB_Q; ++q) {
      sv[q] = sc[q];
#pragma unroll
      for (int g = 0; g < EDL_MAXG; ++g) sv[q] = edw_fma(rg[g], st[g][q], sv[q]);
      pw[q] = fmaf(sv[q].x, sv[q].x, sv[q].y * sv[q].y);
    }
#if !EDW_AHEAD2
    if (bl + 1 < b_hi) fetch(bl + 1, sn, tn);
#endif
    const float tot = ((pw[3] + pw[2]) + pw[1]) + pw[0];
    // energy of the frames BEHIND this thread's: the partner group if it is the later one, and the later rows
    float E;
    {
      const float pt = edw_partner(tot);
      float P[4];
      edw_rows(tot + pt, P);
      E = odd ? 0.f : pt;
      E = fmaf(m_lt3, P[3], E);
      E = fmaf(m_lt2, P[2], E);
      E = fmaf(m_lt1, P[1], E);
    }
    float acc = 0.f, ge[EDB_Q];
#pragma unroll
    for (int q = EDB_Q - 1; q >= 0; --q) {
      E += pw[q];
      const float lin = fabsf(E) + F32_EPS;
      const float raw = EDB_DB_PER_LOG2 * __builtin_amdgcn_logf(lin);
      const bool above = raw > -200.0f;
      const float diff = tc[q] - (above ? raw : -200.0f);
      // dL/dE = -sign(diff) (10 / ln 10) / lin x gscale / sum_abs; zero on the floor and where the difference vanishes
      const float mag = __builtin_amdgcn_rcpf(lin) * gsn;
      const float sm = __uint_as_float((__float_as_uint(diff) & 0x80000000u) ^ __float_as_uint(mag));
      if (FULL || mv[q]) {
        acc += fabsf(diff);
        ge[q] = (above && diff != 0.f) ? sm : 0.f;
      } else {
        ge[q] = 0.f;
      }
    }
#if EDW_AHEAD2
    // (this receiver's cells are used up: the cells of the receiver after the next one take their registers -- two sets,
    // more than one receiver of lead)
    if (bl + 2 < b_hi) fetch(bl + 2, sc, tc);
#endif
    const float gtot = ((ge[0] + ge[1]) + ge[2]) + ge[3];
    // dL/d|S_m|^2 = sum_{m' <= m} dL/dE_m': the earlier rows, the partner group if it is the earlier one
    float run;
    {
      const float pt = edw_partner(gtot);
      float P[4];
      edw_rows(gtot + pt, P);
      run = m_gt0 * P[0];
      run = fmaf(m_gt1, P[1], run);
      run = fmaf(m_gt2, P[2], run);
      run += odd ? pt : 0.f;
    }
    edw2 da[EDL_MAXG] = {edw2{0.f, 0.f}, edw2{0.f, 0.f}, edw2{0.f, 0.f}, edw2{0.f, 0.f}};
#if EDW_GA_LDS
#pragma unroll
    for (int q = 0; q < EDB_Q; q += 2) {
      run += ge[q];
      const edw2 dS0 = (2.0f * run) * sv[q];
      run += ge[q + 1];
      const edw2 dS1 = (2.0f * run) * sv[q + 1];
#pragma unroll
      for (int g = 0; g < EDL_MAXG; ++g) {
        da[g] = edw_fma2(st[g][q], dS0, da[g]);
        da[g] = edw_fma2(st[g][q + 1], dS1, da[g]);
        if (g < G) {
          edw4 v = ga[(g * EDB_Q + q) / 2][lane];
          const edw2 lo = edw_fma(rg[g], dS0, edw2{v.x, v.y}), hi = edw_fma(rg[g], dS1, edw2{v.z, v.w});
          ga[(g * EDB_Q + q) / 2][lane] = edw4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    }
#else
#pragma unroll
    for (int q = 0; q < EDB_Q; ++q) {
      run += ge[q];
      const edw2 dS = (2.0f * run) * sv[q];
#pragma unroll
      for (int g = 0; g < EDL_MAXG; ++g) {
        Ga[g][q] = edw_fma(rg[g], dS, Ga[g][q]);
        da[g] = edw_fma2(st[g][q], dS, da[g]);
      }
    }
#endif
    acc = wave_sum_full(live ? acc : 0.f);
    if (lane == 0) part[(size_t)b * ld_part + col] = acc;
    if (dots) {
      // the four wave sums together: after an exchange with lane ^ 1 a lane holds two of the four values (summed over the
      // pair), after one with lane ^ 2 a single one (over the quad); the rest of the wave is summed once
      const float d0 = da[0].x + da[0].y, d1 = da[1].x + da[1].y, d2 = da[2].x + da[2].y, d3 = da[3].x + da[3].y;
      float kA = bit0 ? d2 : d0, kB = bit0 ? d3 : d1;
      kA += edw_dpp<0xB1>(bit0 ? d0 : d2);                  // quad_perm:[1,0,3,2]
      kB += edw_dpp<0xB1>(bit0 ? d1 : d3);
      float k = bit1 ? kB : kA;
      k += edw_dpp<0x4E>(bit1 ? kA : kB);                   // quad_perm:[2,3,0,1]
      k = dpp_pair_sum<0x124>(k);                           // row_ror:4
      k = dpp_pair_sum<0x128>(k);                           // row_ror:8
      k = xor32_sum(xor16_sum(k));
      if (lane < 4 && gl < G) dots[((size_t)b * G + gl) * ld_dots + col0 + col] = k;
    }
  };
  EDW_STAMP(0);
  if (b_lo < b_hi) fetch(b_lo, sA, tA);
#if EDW_AHEAD2
  if (b_lo + 1 < b_hi) fetch(b_lo + 1, sB, tB);
#endif
  int bl = b_lo;
  for (; bl + 1 < b_hi; bl += 2) {
    if (bl == b_lo + 2) EDW_STAMP(1);
    body(bl, sA, tA, sB, tB);
    body(bl + 1, sB, tB, sA, tA);
  }
  if (bl < b_hi) body(bl, sA, tA, sB, tB);
  EDW_STAMP(2);
  if (live) {
    float2* out = Gsum + (size_t)split * nbands * G * cells;
#pragma unroll
    for (int g = 0; g < EDL_MAXG; ++g) {
      if (g < G) {
#pragma unroll
        for (int q = 0; q < EDB_Q; ++q)
          if (FULL || mv[q]) {
#if EDW_GA_LDS
            const edw4 v = ga[(g * EDB_Q + q) / 2][lane];
            (out + ((size_t)band * G + g) * cells)[cq[q]] = (q & 1) ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
#else
            (out + ((size_t)band * G + g) * cells)[cq[q]] = make_float2(Ga[g][q].x, Ga[g][q].y);
#endif
          }
      }
    }
  }
}

// Gsum[band G + g][cell] = sum_{b in band} gain[b][g] 2 gP[b][cell] S[b][cell],  S[b] = Sd[row_b] + sum_g' gain[b][g'] Stau_g'
//                        = sum_b gain[b][g] 2 gP[b] Sd[row_b]  +  sum_g' (sum_b gain[b][g] gain[b][g'] 2 gP[b]) Stau_g'
// -- one thread per (band, cell), the band's receivers in index order
__global__ __launch_bounds__(256) void k_edr_lin_gsum(EdrLin a, const float* __restrict__ gP, int nframes, int nfreq,
                                                      float2* __restrict__ Gsum) {
  __shared__ float s_rg[256];
  __shared__ long long s_row[64];
  const int band = blockIdx.y, G = a.G, B = a.B;
  for (int i = threadIdx.x; i < B * G; i += 256) s_rg[i] = a.rgain[(size_t)band * B * G + i];
  for (int i = threadIdx.x; i < B; i += 256) s_row[i] = a.rows ? a.rows[band * B + i] : (long long)(band * B + i);
  __syncthreads();
  const size_t cells = (size_t)nframes * nfreq;
  const size_t cell = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= cells) return;
  float2 A[EDL_MAXG];
  float W[EDL_MAXG][EDL_MAXG];
#pragma unroll
  for (int g = 0; g < EDL_MAXG; ++g) {
    A[g] = make_float2(0.f, 0.f);
#pragma unroll
    for (int h = 0; h < EDL_MAXG; ++h) W[g][h] = 0.f;
  }
  const int i0 = band * B;
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const float2 sd = a.Sd[(size_t)s_row[b] * cells + cell];
    const float p2 = 2.0f * gP[(size_t)(i0 + b) * cells + cell];
#pragma unroll
    for (int g = 0; g < EDL_MAXG; ++g) {
      if (g < G) {
        const float r = s_rg[b * G + g] * p2;
        A[g].x += r * sd.x;
        A[g].y += r * sd.y;
#pragma unroll
        for (int h = g; h < EDL_MAXG; ++h)
          if (h < G) W[g][h] += r * s_rg[b * G + h];
      }
    }
  }
  float2 st[EDL_MAXG];
#pragma unroll
  for (int g = 0; g < EDL_MAXG; ++g)
    st[g] = g < G ? a.Stau[((size_t)band * G + g) * cells + cell] : make_float2(0.f, 0.f);
#pragma unroll
  for (int g = 0; g < EDL_MAXG; ++g) {
    if (g < G) {
      float2 o = A[g];
#pragma unroll
      for (int h = 0; h < EDL_MAXG; ++h) {
        if (h < G) {
          const float w = h >= g ? W[g][h] : W[h][g];
          o.x += w * st[h].x;
          o.y += w * st[h].y;
        }
      }
      Gsum[((size_t)band * G + g) * cells + cell] = o;
    }
  }
}

// ---- STFT (Hann 4096, hop 2048, periodic window) of pair-interleaved signals to complex one-sided spectra --------------
__device__ __forceinline__ void edl_load_pair(const float2* __restrict__ x2, int T, int m, int i, float2 (&a)[16],
                                              float2& w1) {
  float sn, cs;
  sincospif(2.0f * (float)i / 4096.0f, &sn, &cs);
  w1 = make_float2(cs, -sn);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float sk, ck;
    sincospif((float)k * 0.125f, &sk, &ck);
    const float h = 0.5f - 0.5f * (cs * ck - sn * sk);
    const int t = m * 2048 + i + 256 * k;
    const float2 v = t < T ? x2[t] : make_float2(0.f, 0.f);
    a[k] = make_float2(h * v.x, h * v.y);
  }
}

// S (items, nframes, 2049) complex: S[2p] from the .x signal of pair p, S[2p + 1] from its .y signal
__global__ __launch_bounds__(S4K_T, 3) void k_stft4k_pair_spec(const float2* __restrict__ x2, int ld, int T, int nframes,
                                                            int items, float2* __restrict__ S, int tiled) {
  float2* buf = edl_lds;
  const int p = blockIdx.y, m = blockIdx.x, nf = 2049, i = threadIdx.x;
  const int b1 = 2 * p;
  const bool two = b1 + 1 < items;
  float2 a[16], w1;
  edl_load_pair(x2 + (size_t)p * ld, T, m, i, a, w1);
  fft4096(a, buf, i, w1, 1.0f);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[S4K_PAD(i + 256 * u)] = a[u];
  __syncthreads();
  float2* S1 = S + (size_t)b1 * nframes * nf;
  float2* S2 = S1 + (size_t)nframes * nf;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const float2 zf = a[u], zc = buf[S4K_PAD((4096 - f) & 4095)];
      const size_t c = edl_cell(m, f, nframes, nf, tiled);
      // S_a = (Z_f + conj Z_{W-f}) / 2 ; S_b = (Z_f - conj Z_{W-f}) / (2i)
      S1[c] = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
      if (two) S2[c] = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
    }
  }
}

// (cos, sin)(u pi / 8), u = 0..15: the Hann window of the adjoint's epilogue by angle addition
__constant__ float2 c_edl_hann[16] = {
    {1.0f, 0.0f}, {0.92387953251128674f, 0.38268343236508977f}, {0.70710678118654752f, 0.70710678118654752f},
    {0.38268343236508977f, 0.92387953251128674f}, {0.0f, 1.0f}, {-0.38268343236508977f, 0.92387953251128674f},
    {-0.70710678118654752f, 0.70710678118654752f}, {-0.92387953251128674f, 0.38268343236508977f}, {-1.0f, 0.0f},
    {-0.92387953251128674f, -0.38268343236508977f}, {-0.70710678118654752f, -0.70710678118654752f},
    {-0.38268343236508977f, -0.92387953251128674f}, {0.0f, -1.0f}, {0.38268343236508977f, -0.92387953251128674f},
    {0.70710678118654752f, -0.70710678118654752f}, {0.92387953251128674f, -0.38268343236508977f}};

// Adjoint: gx2 (pairs, ld) = [base2 +] STFT^T(Gs), Gs (items, nframes, 2049) the gradient w.r.t. the one-sided spectra.
// Frames of one parity tile the time axis without overlap: the even launch STORES base + contribution (base alone where no
// even frame reaches), the odd launch adds with a plain read-modify-write -- no atomics, no cleared buffer.
// parity 2: ALL frames in one launch, the odd frames' contributions STORED to a second signal set gx2b (zeros where no odd
// frame reaches) instead of added into gx2 -- no second launch behind the first; the consumer adds the two sets
// (gfdn_lin_gamma_dots: base2 + base2b).
__global__ __launch_bounds__(S4K_T, 3) void k_stft4k_pair_spec_bwd(const float2* __restrict__ Gs, int ld, int T,
                                                                int nframes, int items, const float2* base2,
                                                                float2* gx2, int parity, int tiled, int nsplit,
                                                                float2* gx2b) {
  float2* buf = edl_lds;
  const bool both = parity == 2;
  const int m = both ? (int)blockIdx.x : 2 * (int)blockIdx.x + parity;
  if (both) {
    parity = m & 1;
    if (parity) { gx2 = gx2b; base2 = nullptr; }
  }
  const int p = blockIdx.y, nf = 2049, i = threadIdx.x;
  const int b1 = 2 * p;
  const bool two = b1 + 1 < items;
  const float2* ga = Gs + (size_t)b1 * nframes * nf;
  const float2* gb = ga + (size_t)nframes * nf;
  float sn, cs;
  sincospif(2.0f * (float)i / 4096.0f, &sn, &cs);
  const float2 w1 = make_float2(cs, -sn);
  // U = Ga_sym + i Gb_sym: the pair (f, W - f) is written by exactly one thread
#pragma unroll 3
  for (int u = 0; u < 9; ++u) {
    const int f = i + 256 * u;
    if (u < 8 || i == 0) {
      const int fc = (4096 - f) & 4095;
      const size_t c = edl_cell(m, f, nframes, nf, tiled);
      float2 Ga = ga[c], Gb = two ? gb[c] : make_float2(0.f, 0.f);
      for (int sp = 1; sp < nsplit; ++sp) {                  // partial planes of k_edr_lin_wave, in order
        const size_t o = (size_t)sp * items * nframes * nf + c;
        Ga = cadd(Ga, ga[o]);
        if (two) Gb = cadd(Gb, gb[o]);
      }
      if (f == 0 || f == 2048) {
        buf[S4K_PAD(f)] = make_float2(Ga.x, Gb.x);                                       // real-only bins
      } else {
        buf[S4K_PAD(f)] = make_float2(0.5f * (Ga.x - Gb.y), 0.5f * (Ga.y + Gb.x));       // U_f
        buf[S4K_PAD(fc)] = make_float2(0.5f * (Ga.x + Gb.y), 0.5f * (-Ga.y + Gb.x));     // U_{W-f}
      }
    }
  }
  __syncthreads();
  float2 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = buf[S4K_PAD(i + 256 * k)];
  __syncthreads();
  fft4096(a, buf, i, w1, -1.0f);
  float2* g = gx2 + (size_t)p * ld;
  const float2* bs = base2 ? base2 + (size_t)p * ld : nullptr;
  const bool store = both || !parity;                       // (this launch defines the samples it reaches; else: adds)
  const float2* src = store ? bs : g;
  const int tlim = store ? ld : T;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int j = i + 256 * u, t = m * 2048 + j;
    if (t < tlim) {
      const float2 o = src ? src[t] : make_float2(0.f, 0.f);
      const float hw = t < T ? 0.5f - 0.5f * (w1.x * c_edl_hann[u].x + w1.y * c_edl_hann[u].y) : 0.f;
      g[t] = make_float2(o.x + hw * a[u].x, o.y + (two ? hw * a[u].y : 0.f));
    }
  }
  if (store && m + 2 >= nframes)
    for (int t = (m + 2) * 2048 + i; t < ld; t += S4K_T) g[t] = src ? src[t] : make_float2(0.f, 0.f);
  if (both && m == 1)                                       // (no odd frame reaches the first hop)
    for (int t = i; t < 2048 && t < ld; t += S4K_T) g[t] = make_float2(0.f, 0.f);
}

static int edl_nframes(int T) {
  const int Tp = ((T + 2047) / 2048) * 2048;
  return Tp < 4096 ? 0 : (Tp - 4096) / 2048 + 1;
}

extern "C" int gfdn_stft_pairs_spectrum(const float* x2, int ld, int T, int items, int win, float* S_c64, int tiled,
                                        void* stream) {
  if (!x2 || !S_c64 || items <= 0 || T <= 0 || ld < T) return GFDN_E_BADARG;
  if (win != 4096) return GFDN_E_UNSUPPORTED;
  const int nframes = edl_nframes(T);
  if (nframes <= 0) return GFDN_E_BADARG;
  hipLaunchKernelGGL(k_stft4k_pair_spec, dim3(nframes, (items + 1) / 2), dim3(S4K_T), S4K_LDS * sizeof(float2),
                     (hipStream_t)stream, (const float2*)x2, ld, T, nframes, items, (float2*)S_c64, tiled ? 1 : 0);
  GFDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gfdn_stft_pairs_spectrum_bwd(const float* G_c64, int T, int items, int win, const float* base2, float* gx2,
                                            int ld, int tiled, int nsplit, float* gx2b, void* stream) {
  if (!G_c64 || !gx2 || items <= 0 || T <= 0 || ld < T || base2 == gx2 || nsplit <= 0 || gx2b == gx2) return GFDN_E_BADARG;
  if (win != 4096) return GFDN_E_UNSUPPORTED;
  const int nframes = edl_nframes(T);
  if (nframes <= 0) return GFDN_E_BADARG;
  if (gx2b) {                       // one launch: even frames -> gx2 (+ base2), odd frames -> gx2b
    if (nframes < 2) return GFDN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_stft4k_pair_spec_bwd, dim3(nframes, (items + 1) / 2), dim3(S4K_T), S4K_LDS * sizeof(float2),
                       (hipStream_t)stream, (const float2*)G_c64, ld, T, nframes, items, (const float2*)base2,
                       (float2*)gx2, 2, tiled ? 1 : 0, nsplit, (float2*)gx2b);
    GFDN_LAUNCH_CHECK();
    return 0;
  }
  for (int parity = 0; parity < 2; ++parity) {
    const int nb = (nframes + 1 - parity) / 2;
    if (nb == 0) continue;
    hipLaunchKernelGGL(k_stft4k_pair_spec_bwd, dim3(nb, (items + 1) / 2), dim3(S4K_T), S4K_LDS * sizeof(float2),
                       (hipStream_t)stream, (const float2*)G_c64, ld, T, nframes, items, (const float2*)base2,
                       (float2*)gx2, parity, tiled ? 1 : 0, nsplit, (float2*)nullptr);
    GFDN_LAUNCH_CHECK();
  }
  return 0;
}

// partial-sum columns per item: frequency blocks of 256 (gfdn_edr_lin_loss) or tiles of 64 (gfdn_edr_lin_loss_gsum)
extern "C" int gfdn_edr_lin_parts(int nfreq) { return nfreq > 0 ? (nfreq + 255) / 256 : 0; }
extern "C" int gfdn_edr_lin_band_parts(int nfreq) {
  if (nfreq <= 0) return 0;
  return EDW_WG * ((nfreq + EDW_F * EDW_WG - 1) / (EDW_F * EDW_WG));      // one column per wave of 8 frequencies
}

// EDR loss of nbands x B receivers on composed spectra (see the head of this file).  part (items, gfdn_edr_lin_parts):
// loss partials as gfdn_edr_loss(loss_item = NULL) leaves them; want_grad: gP (items, nframes, nfreq) and the EDR part of
// dL/dgain as partial rows dots[(b G + g) ld_dots + col0 .. + gfdn_edr_lin_parts) (dots may be NULL).
extern "C" int gfdn_edr_lin_loss(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain,
                                 int nbands, int B, int G, const float* T_db, const float* sum_abs, int nframes, int nfreq,
                                 float gscale, int want_grad, float* gP, float* part, float* dots, int ld_dots, int col0,
                                 int tiled, void* stream) {
  if (!Sd_c64 || !Stau_c64 || !rgain || !T_db || !sum_abs || !part || nbands <= 0 || B <= 0 || G <= 0 || nframes <= 0 ||
      nfreq <= 0 || (want_grad && !gP))
    return GFDN_E_BADARG;
  const int fblk = (nfreq + 255) / 256;
  if (G > EDL_MAXG || nframes > EDL_RF || nbands * B > 65535) return GFDN_E_UNSUPPORTED;
  if (dots && (col0 < 0 || ld_dots < col0 + fblk)) return GFDN_E_BADARG;
  EdrLin a{(const float2*)Sd_c64, rows, (const float2*)Stau_c64, rgain, B, G, T_db, sum_abs, tiled ? 1 : 0};
  const int nslices = nbands * fblk;
  hipLaunchKernelGGL(k_edr_lin_cols, dim3(8 * ((nslices + 7) / 8) * B), dim3(256), 0, (hipStream_t)stream, a, nframes, nfreq,
                     gscale, want_grad, gP, part, dots, ld_dots, col0, nslices);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// gfdn_edr_lin_loss(want_grad) and gfdn_edr_lin_gsum as ONE launch (k_edr_lin_wave): part (items, ld_part >=
// gfdn_edr_lin_band_parts) and dots columns [col0, col0 + gfdn_edr_lin_band_parts) as there, Gsum (nsplit, nbands G, nframes,
// nfreq) complex partial planes (their sum over the first index is gfdn_edr_lin_gsum's output; the adjoint STFT takes nsplit);
// dL/d|S|^2 is never written.
extern "C" int gfdn_edr_lin_loss_gsum(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain,
                                      int nbands, int B, int G, const float* T_db, const float* sum_abs, int nframes,
                                      int nfreq, float gscale, float* part, int ld_part, float* dots, int ld_dots, int col0,
                                      float* Gsum_c64, int nsplit, int tiled, void* stream) {
  if (!Sd_c64 || !Stau_c64 || !rgain || !T_db || !sum_abs || !part || !Gsum_c64 || nbands <= 0 || B <= 0 || G <= 0 ||
      nframes <= 0 || nfreq <= 0 || nsplit <= 0 || nsplit > B)
    return GFDN_E_BADARG;
  const int nparts = gfdn_edr_lin_band_parts(nfreq);
  if (G > EDL_MAXG || nframes > EDL_RF || nbands > 65535 || nsplit > 64) return GFDN_E_UNSUPPORTED;
  if (ld_part < nparts || (dots && (col0 < 0 || ld_dots < col0 + nparts))) return GFDN_E_BADARG;
  EdrLin a{(const float2*)Sd_c64, rows, (const float2*)Stau_c64, rgain, B, G, T_db, sum_abs, tiled ? 1 : 0};
  if (nframes == EDL_RF)
    hipLaunchKernelGGL(k_edr_lin_wave<true>, dim3(nparts / EDW_WG, nbands, nsplit), dim3(64 * EDW_WG), 0, (hipStream_t)stream,
                       a, nframes, nfreq, gscale, part, ld_part, dots, ld_dots, col0, (float2*)Gsum_c64, nsplit, nbands);
  else
    hipLaunchKernelGGL(k_edr_lin_wave<false>, dim3(nparts / EDW_WG, nbands, nsplit), dim3(64 * EDW_WG), 0, (hipStream_t)stream,
                       a, nframes, nfreq, gscale, part, ld_part, dots, ld_dots, col0, (float2*)Gsum_c64, nsplit, nbands);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// Gsum (nbands G, nframes, nfreq) complex = sum over the band's receivers of gain[b][g] dL/dS[b] (dL/dS = 2 gP S)
extern "C" int gfdn_edr_lin_gsum(const float* Sd_c64, const long long* rows, const float* Stau_c64, const float* rgain,
                                 int nbands, int B, int G, const float* gP, int nframes, int nfreq, float* Gsum_c64,
                                 void* stream) {
  if (!Sd_c64 || !Stau_c64 || !rgain || !gP || !Gsum_c64 || nbands <= 0 || B <= 0 || G <= 0 || nframes <= 0 || nfreq <= 0)
    return GFDN_E_BADARG;
  if (G > EDL_MAXG || B * G > 256 || B > 64 || nbands > 65535) return GFDN_E_UNSUPPORTED;
  EdrLin a{(const float2*)Sd_c64, rows, (const float2*)Stau_c64, rgain, B, G, nullptr, nullptr, 0};
  const size_t cells = (size_t)nframes * nfreq;
  hipLaunchKernelGGL(k_edr_lin_gsum, dim3((unsigned)((cells + 255) / 256), nbands), dim3(256), 0, (hipStream_t)stream, a, gP,
                     nframes, nfreq, (float2*)Gsum_c64);
  GFDN_LAUNCH_CHECK();
  return 0;
}
