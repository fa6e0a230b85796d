// Fused decay-loss forward of ONE item per workgroup, for gfx950 (pair-interleaved time signals, win = 4096).
//
// What the reference does per item (losses.py:430-495 edr_loss, :501-575 get_stft_torch / get_edr_from_stft,
// :201-238 edc_loss with :187-199 schroeder_backward_integral):
//   S = stft(rir, 4096, hop 2048, periodic Hann, center=False);  EDR[f][m] = 10 log10(sum_{tau >= m} |S[f][tau]|^2 + eps)
//   edr term  = sum_{f,m} |EDR_t - EDR_a| / sum |EDR_t|
//   EDC[t]    = sum_{t' >= t} rir[t']^2 over the window [start, start + len);  edc term = mean |10 log10 EDC_t - ...|
// Both decay curves are SUFFIX sums over time, and both gradients are PREFIX sums of the staged dB-stage terms.  One
// workgroup therefore walks its item from the last frame to the first: the |STFT|^2 of eight frames per round stay in
// LDS, the running tail energies E[f] and the running EDC carry stay in registers, and what crosses HBM is the signal
// (once), the two targets (once), dL/dEDR -> dL/dP (staged and prefix-summed in place by the thread that wrote it) and
// the EDC gradient 2 x cum(dL/dEDC) (planar, per item).  |STFT|^2 itself is never stored; the separate EDR kernel and
// the three EDC scan kernels of the unfused chain (gfdn_stft_power_pairs -> gfdn_edr_loss, gfdn_edc_loss_pairs) do
// not run.  Same arithmetic per element as those kernels (dB, clip at -200, sign, 10/ln10/(E + eps)).
//
// Geometry: 1024 threads = 4 groups of 256; group q transforms frames (base + 2q, base + 2q + 1) of the round as the
// real and imaginary part of ONE register-resident 4096-point FFT (fft4k_dev.h); rounds go base = 8 (R - 1), ..., 0.
// The EDC tile of a round is the 16 384 samples [base * 2048, (base + 8) * 2048) the round's frames begin in.
#include "common.h"
#include "fft4k_dev.h"
#include "scan_dev.h"

#define DK_GROUPS 4
#define DK_THREADS (DK_GROUPS * S4K_T)
#define DK_FPR (2 * DK_GROUPS)             // frames per round
#define DK_TILE (DK_FPR * 2048)            // samples per round = one EDC tile
#define DK_SUB (DK_THREADS * 4)            // one sub-tile: four consecutive samples per thread
#define DK_PLD 2064                        // row pitch of the per-frame |S|^2 rows in LDS (2049 rounded up)
#define DK_NF 2049
#define DB_PER_LOG2 3.0102999566398120f    // 10 log10(2): dB = DB_PER_LOG2 * log2(x) on the hardware log
#define DK_LDS_BYTES ((size_t)DK_GROUPS * S4K_LDS * sizeof(float2) + (64 + 16) * sizeof(float))

struct DecayArgs {
  const float2* x2;          // (ceil(items / 2), ld) pair-interleaved signals
  int ld, T, items, nframes;
  // EDR
  const float* Tdb;          // (rows, nframes, 2049) target EDR in dB
  const float* sum_abs;      // (rows) sum |EDR_t|
  const long long* rows;     // item -> row of the target stores (NULL: identity)
  const float* wf;           // (2049) frequency weights or NULL
  float edr_gs;
  float* gP;                 // (items, nframes, 2049) dL/d|S|^2 out (want_grad)
  float* edr_part;           // (items) sum wf |diff| (not yet divided by sum_abs)
  // EDC
  int start, len;
  const float* Tedc;         // (rows, len) target EDC in dB
  const float* maskw;        // (len) time weights or NULL
  float inv_count, edc_gs;
  float* dxe;                // (items, len) out: d(edc term)/dx over the window (want_grad)
  float* edc_loss;           // (items)
  int want_grad;
};

extern __shared__ float2 dk_lds[];

// Phase stamps of workgroup 0 (diagnostic builds only: -DDK_STAMPS; tools/decay_probe.py --stamps)
#ifdef DK_STAMPS
__device__ long long dk_stamps[64];
#define DK_STAMP(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) dk_stamps[n] = clock64(); } while (0)
extern "C" int gfdn_decay_stamps(long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(dk_stamps), sizeof(long long) * 64);
}
#else
#define DK_STAMP(n)
#endif

__global__ __launch_bounds__(DK_THREADS) void k_decay_item(DecayArgs A) {
  // block -> item: both items of a pair (they share every cache line of x2) on ONE XCD, i.e. block ids of one
  // residue class mod 8
  int b;
  {
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    b = 2 * ((j >> 1) * 8 + xcd) + (j & 1);
    if (b >= A.items) return;
  }
  const int tid = threadIdx.x;
  float* s_scan = (float*)(dk_lds + DK_GROUPS * S4K_LDS);
  float* s_red = s_scan + 64;
  const int nframes = A.nframes, nf = DK_NF, T = A.T;
  const int comp = b & 1;
  const float* xrow = (const float*)(A.x2 + (size_t)(b >> 1) * A.ld);      // float view of the pair's row
  const float* xc = xrow + comp;                                           // this item's samples: xc[2 t]
  const size_t tb = A.rows ? (size_t)A.rows[b] : (size_t)b;
  const float* Trow = A.Tdb + tb * nframes * nf;
  float* gPb = A.gP ? A.gP + (size_t)b * nframes * nf : nullptr;
  const bool grad = A.want_grad && gPb;
  const float inv_norm = 1.0f / A.sum_abs[tb];
  const int R = (nframes + DK_FPR - 1) / DK_FPR;

  // ---- EDC state
  const int ws = A.start, we = A.start + A.len;
  const float* trow = A.Tedc + tb * A.len;
  float* drow = A.dxe ? A.dxe + (size_t)b * A.len : nullptr;
  const bool egrad = A.want_grad && drow;
  const int ti_lo = ws / DK_TILE, ti_hi = (we - 1) / DK_TILE;
  float carry = 0.f, acc_edc = 0.f;

  auto pick4 = [&](const float* p8, float (&o)[4]) {       // samples t .. t + 3 of this item from xrow + 2 t
    const f4u a = *(const f4u*)p8, c = *(const f4u*)(p8 + 4);
    o[0] = comp ? a.y : a.x; o[1] = comp ? a.w : a.z; o[2] = comp ? c.y : c.x; o[3] = comp ? c.w : c.z;
  };

  // suffix scan of x^2 over one tile, from its end: EDC, dB, |diff|, staged dL/dEDC (k_edc_pair_seg_fwd's arithmetic)
  auto edc_fwd_tile = [&](int ti) {
    const int tile_hi = (ti + 1) * DK_TILE;
    float val[4][4], tg[4][4], mk[4][4], loc[4], incl[4], tot[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t_lo = tile_hi - s * DK_SUB - 4 * tid - 4;
      if (t_lo >= ws && t_lo + 4 <= we) {
        float xv[4], t4[4];
        pick4(xrow + 2 * (size_t)t_lo, xv);
        ld4_f(trow + (t_lo - ws), t4);
#pragma unroll
        for (int u = 0; u < 4; ++u) { val[s][u] = xv[3 - u] * xv[3 - u]; tg[s][u] = t4[3 - u]; }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t_lo + 3 - u;
          const bool in = t >= ws && t < we;
          const float v = in ? xc[2 * (size_t)t] : 0.f;
          val[s][u] = v * v;
          tg[s][u] = in ? trow[t - ws] : 0.f;
        }
      }
      {
        float m4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (A.maskw) {
          if (t_lo >= ws && t_lo + 4 <= we) ld4_f(A.maskw + (t_lo - ws), m4);
          else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t_lo + u; m4[u] = (t >= ws && t < we) ? A.maskw[t - ws] : 0.f; }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mk[s][u] = m4[u];
      }
      float run = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) run += val[s][u];
      loc[s] = run;
      incl[s] = run;
    }
    block_scan_multi(incl, tot, s_scan);
    float base = carry;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float excl = base + incl[s] - loc[s];
      const int t_lo = tile_hi - s * DK_SUB - 4 * tid - 4;
      const bool whole = t_lo >= ws && t_lo + 4 <= we;
      float gq[4];
      float run = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        run += val[s][u];
        const float edc = excl + run;
        const int t = t_lo + 3 - u;
        const bool in = t >= ws && t < we;
        float g = 0.f;
        if (in) {
          const float m = mk[s][3 - u];
          const float lin = fabsf(edc) + F32_EPS;
          const float raw = DB_PER_LOG2 * __log2f(lin);
          const float diff = tg[s][u] - fmaxf(raw, -200.0f);
          acc_edc += m * fabsf(diff);
          const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
          const float dE = (raw > -200.0f) ? TEN_OVER_LN10 * __builtin_amdgcn_rcpf(lin) : 0.f;
          g = -sg * dE * m * A.inv_count * A.edc_gs;
        }
        gq[3 - u] = g;
      }
      if (egrad) {
        if (whole) st4_f(drow + (t_lo - ws), gq);
        else {
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int t = t_lo + u; if (t >= ws && t < we) drow[t - ws] = gq[u]; }
        }
      }
      base += tot[s];
    }
    carry = base;
    __syncthreads();          // s_scan is reused by the next scan
  };

  // (window samples beyond the last round's tile: none at n = 65 537 with a window that ends inside the frames)
  for (int ti = ti_hi; ti >= R; --ti) edc_fwd_tile(ti);

  // ---- rounds: eight frames each, last frames first
  float E0 = 0.f, E1 = 0.f, E2 = 0.f, acc_edr = 0.f;
  for (int r = 0; r < R; ++r) {
    // (the thread index is made opaque once per round: every per-thread address below is then recomputed inside the
    // round instead of being hoisted out of the loop and held in registers across the transform -- the kernel has
    // exactly the 128 registers of a 1024-thread workgroup)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int q = tid >> 8, i = tid & 255;
    float2* buf = dk_lds + q * S4K_LDS;
    const int f0 = tid, f1 = tid + 1024;
    const int base = DK_FPR * (R - 1 - r);
    const int ma = base + 2 * q;
    const bool va_ok = ma < nframes, vb_ok = ma + 1 < nframes;
    float2 a[16];
    {
      float sn, cs;
      sincospif(2.0f * (float)i / 4096.0f, &sn, &cs);
      const float2 w1 = make_float2(cs, -sn);
      float u[24];                     // (the two frames of a group overlap by half: 24 distinct samples per thread)
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        const int t = ma * 2048 + i + 256 * k;
        u[k] = t < T ? xc[2 * (size_t)t] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float sk, ck;
        sincospif((float)k * 0.125f, &sk, &ck);                 // compile-time constants after unrolling
        const float h = 0.5f - 0.5f * (cs * ck - sn * sk);      // periodic Hann at j = i + 256 k
        a[k] = make_float2(va_ok ? h * u[k] : 0.f, vb_ok ? h * u[k + 8] : 0.f);
      }
      DK_STAMP(2 + 8 * r);
      fft4096(a, buf, i, w1, 1.0f);
    }
    __syncthreads();
    DK_STAMP(3 + 8 * r);
#pragma unroll
    for (int u_ = 0; u_ < 16; ++u_) buf[S4K_PAD(i + 256 * u_)] = a[u_];
    // the round's target EDR columns: in flight across the Hermitian split (column 2048: one frame per lane of the
    // first eight)
    float tv0[DK_FPR], tv1[DK_FPR];
#pragma unroll
    for (int fr = 0; fr < DK_FPR; ++fr) {
      const int m = base + fr;
      const bool ok = m < nframes;
      tv0[fr] = ok ? Trow[(size_t)m * nf + f0] : 0.f;
      tv1[fr] = ok ? Trow[(size_t)m * nf + f1] : 0.f;
    }
    const float tv2 = (tid < DK_FPR && base + tid < nframes) ? Trow[(size_t)(base + tid) * nf + 2048] : 0.f;
    __syncthreads();
    float pa[9], pb[9];
#pragma unroll
    for (int u_ = 0; u_ < 9; ++u_) {
      const int f = i + 256 * u_;
      pa[u_] = 0.f;
      pb[u_] = 0.f;
      if (u_ < 8 || i == 0) {
        const float2 zf = buf[S4K_PAD(f)], zc = buf[S4K_PAD((4096 - f) & 4095)];
        // S_a = (Z_f + conj Z_{W-f})/2 ; S_b = (Z_f - conj Z_{W-f})/(2i)
        const float2 sa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
        const float2 sb = make_float2(0.5f * (zf.y + zc.y), -0.5f * (zf.x - zc.x));
        pa[u_] = sa.x * sa.x + sa.y * sa.y;
        pb[u_] = sb.x * sb.x + sb.y * sb.y;
      }
    }
    __syncthreads();          // every read of the spectra is done: the buffer becomes the |S|^2 rows of the two frames
    {
      float* Pl = (float*)buf;
#pragma unroll
      for (int u_ = 0; u_ < 9; ++u_) {
        if (u_ < 8 || i == 0) {
          const int f = i + 256 * u_;
          Pl[f] = pa[u_];
          Pl[DK_PLD + f] = pb[u_];
        }
      }
    }
    __syncthreads();
    DK_STAMP(4 + 8 * r);
    // EDR columns: thread tid owns frequencies tid, tid + 1024, frames of the round last first
    auto cell = [&](float E, float tdb, float w, float gs, float* gp) {
      const float lin = fabsf(E) + F32_EPS;
      const float raw = DB_PER_LOG2 * __log2f(lin);
      const float diff = tdb - fmaxf(raw, -200.0f);
      acc_edr += w * fabsf(diff);
      if (grad) {
        const float sg = diff > 0.f ? 1.0f : (diff < 0.f ? -1.0f : 0.0f);
        const float dE = (raw > -200.0f) ? TEN_OVER_LN10 * __builtin_amdgcn_rcpf(lin) : 0.f;
        *gp = -sg * dE * gs;
      }
    };
    auto column = [&](int f, float& E, const float (&tv)[DK_FPR]) {
      const float w = A.wf ? A.wf[f] : 1.0f;
      const float gs = grad ? A.edr_gs * w * inv_norm : 0.f;
#pragma unroll
      for (int fr = DK_FPR - 1; fr >= 0; --fr) {
        const int m = base + fr;
        if (m < nframes) {
          const float* Pl = (const float*)(dk_lds + (fr >> 1) * S4K_LDS) + (fr & 1) * DK_PLD;
          E += Pl[f];
          cell(E, tv[fr], w, gs, gPb + (size_t)m * nf + f);
        }
      }
    };
    column(f0, E0, tv0);
    column(f1, E1, tv1);
    if (tid < 64) {
      // column 2048: lane fr of the first wave takes frame base + fr -- tail energies by a suffix sum over the lanes
      const int fr = tid & (DK_FPR - 1);
      const bool act = tid < DK_FPR && base + fr < nframes;
      const float* Pl = (const float*)(dk_lds + (fr >> 1) * S4K_LDS) + (fr & 1) * DK_PLD;
      float sfx = act ? Pl[2048] : 0.f;
#pragma unroll
      for (int off = 1; off < DK_FPR; off <<= 1) {
        const float o = __shfl_down(sfx, off, 64);
        if (fr + off < DK_FPR) sfx += o;
      }
      if (act) {
        const float w = A.wf ? A.wf[2048] : 1.0f;
        cell(E2 + sfx, tv2, w, grad ? A.edr_gs * w * inv_norm : 0.f, gPb + (size_t)(base + fr) * nf + 2048);
      }
      E2 += __shfl(sfx, 0, 64);
    }
    __syncthreads();          // the |S|^2 rows are dead: the next round's transforms reuse the buffers
    DK_STAMP(5 + 8 * r);
    const int ti = R - 1 - r;
    if (ti >= ti_lo && ti <= ti_hi) edc_fwd_tile(ti);
    DK_STAMP(6 + 8 * r);
  }
  DK_STAMP(40);

  // ---- losses of the item
  acc_edr = block_sum(acc_edr, s_red);
  __syncthreads();
  acc_edc = block_sum(acc_edc, s_red);
  if (tid == 0) {
    A.edr_part[b] = acc_edr;
    A.edc_loss[b] = acc_edc * A.inv_count;
  }
  DK_STAMP(41);
  if (!A.want_grad) return;

  // ---- dL/dEDR -> dL/d|S|^2: E_m = sum_{tau >= m} P_tau  =>  gP_tau = sum_{m <= tau} gE_m.  Every thread prefix-sums the
  // columns it staged itself (same thread, same addresses: program order is enough); sixteen frames of both columns
  // are in flight at a time
  if (grad) {
    const int f0 = tid, f1 = tid + 1024;
    float run0 = 0.f, run1 = 0.f;
    for (int m0 = 0; m0 < nframes; m0 += 16) {
      float v0[16], v1[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool ok = m0 + j < nframes;
        v0[j] = ok ? gPb[(size_t)(m0 + j) * nf + f0] : 0.f;
        v1[j] = ok ? gPb[(size_t)(m0 + j) * nf + f1] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (m0 + j < nframes) {
          run0 += v0[j];
          run1 += v1[j];
          gPb[(size_t)(m0 + j) * nf + f0] = run0;
          gPb[(size_t)(m0 + j) * nf + f1] = run1;
        }
      }
    }
    if (tid < 64) {           // column 2048 (staged by the first eight lanes of this wave): one frame per lane
      float run = 0.f;
      for (int m0 = 0; m0 < nframes; m0 += 64) {
        const int m = m0 + tid;
        float v = m < nframes ? gPb[(size_t)m * nf + 2048] : 0.f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float o = __shfl_up(v, off, 64);
          if (tid >= off) v += o;
        }
        if (m < nframes) gPb[(size_t)m * nf + 2048] = run + v;
        run += __shfl(v, 63, 64);
      }
    }
  }

  DK_STAMP(42);
  // ---- EDC adjoint: prefix sums of the staged dL/dEDC, times 2 x (k_edc_pair_seg_bwd's arithmetic), tiles first to last.
  // The staged terms were written by other threads of this workgroup: the barrier orders them (workgroup-scope fence)
  if (egrad) {
    __syncthreads();
    float cr = 0.f;
    for (int ti = ti_lo; ti <= ti_hi; ++ti) {
      const int tile_lo = ti * DK_TILE;
      float val[4][4], xs[4][4], loc[4], incl[4], tot[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t0 = tile_lo + s * DK_SUB + 4 * tid;
        if (t0 >= ws && t0 + 4 <= we) {
          ld4_f(drow + (t0 - ws), val[s]);
          pick4(xrow + 2 * (size_t)t0, xs[s]);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            const bool in = t >= ws && t < we;
            val[s][u] = in ? drow[t - ws] : 0.f;
            xs[s][u] = in ? xc[2 * (size_t)t] : 0.f;
          }
        }
        float run = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) run += val[s][u];
        loc[s] = run;
        incl[s] = run;
      }
      block_scan_multi(incl, tot, s_scan);
      float base = cr;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float excl = base + incl[s] - loc[s];
        const int t0 = tile_lo + s * DK_SUB + 4 * tid;
        float out[4];
        float run = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          run += val[s][u];
          out[u] = 2.0f * xs[s][u] * (excl + run);
        }
        if (t0 >= ws && t0 + 4 <= we) st4_f(drow + (t0 - ws), out);
        else {
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int t = t0 + u; if (t >= ws && t < we) drow[t - ws] = out[u]; }
        }
        base += tot[s];
      }
      cr = base;
      __syncthreads();
    }
  }
  DK_STAMP(43);
}

extern "C" int gfdn_decay_items_fwd(const float* x2, int ld, int T, int items, int win,
                                    const float* T_edr_db, const float* sum_abs, const long long* target_rows,
                                    const float* wf, float edr_gscale,
                                    int start, int len, const float* T_edc_db, const float* maskw, float inv_count,
                                    float edc_gscale, int want_grad,
                                    float* gP, float* edr_part, float* edc_loss_item, float* dxe, void* stream) {
  if (!x2 || !T_edr_db || !sum_abs || !T_edc_db || !edr_part || !edc_loss_item || items <= 0 || ld < T || start < 0 ||
      len <= 0 || start + len > T)
    return GFDN_E_BADARG;
  if (want_grad && (!gP || !dxe)) return GFDN_E_BADARG;
  if (win != 4096) return GFDN_E_UNSUPPORTED;
  const int nframes = gfdn_stft_nframes(T, win);
  if (nframes <= 0) return GFDN_E_BADARG;
  int rc = ensure_dyn_lds(k_decay_item, DK_LDS_BYTES);
  if (rc) return rc;
  DecayArgs A;
  A.x2 = (const float2*)x2; A.ld = ld; A.T = T; A.items = items; A.nframes = nframes;
  A.Tdb = T_edr_db; A.sum_abs = sum_abs; A.rows = target_rows; A.wf = wf; A.edr_gs = edr_gscale;
  A.gP = gP; A.edr_part = edr_part;
  A.start = start; A.len = len; A.Tedc = T_edc_db; A.maskw = maskw; A.inv_count = inv_count; A.edc_gs = edc_gscale;
  A.dxe = dxe; A.edc_loss = edc_loss_item; A.want_grad = want_grad;
  const int npairs = (items + 1) / 2;
  const int blocks = 16 * ((npairs + 7) / 8);
  hipLaunchKernelGGL(k_decay_item, dim3(blocks), dim3(DK_THREADS), DK_LDS_BYTES, (hipStream_t)stream, A);
  GFDN_LAUNCH_CHECK();
  return 0;
}
