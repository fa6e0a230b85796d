// Float64 transforms for DATASET CONSTANTS of the linear step (round 6), for gfx950.
//
// The reference evaluates irfft(H, n = K) of the decay losses on complex128 spectra (orchidas/DiffGFDN,
// src/diff_gfdn/losses.py:207-213, :442-445; H is complex128 because the complex128 direct path is added, model.py:618-619) and
// the dataset's rfft in float64 (dataloader.py:250, :300-325).  The linear step splits x = irfft((sum_g gain_g T_g + d) filt, n)
// into xd[row] + sum_g gain_g tau_g with xd = irfft(d filt, n) a store built ONCE per dataset (csrc/linear.hip); built with the
// float32 transforms of csrc/fft.hip its samples carry 3e-7 of the row's largest sample as absolute error, which on the last
// tenth of the EDC window is percent-level relative error and reaches dL/dM through the dB stages (DESIGN.md section 2:
// 1.3e-4 of dL/dM's largest entry, 0.5e-4 with an exact store).  Speed is irrelevant here (once per dataset), accuracy is the
// point: plain radix-2 Stockham passes through memory on double2, exact argument reduction for every twiddle
// (sincospi on exactly reduced integer phases), Bluestein's chirp form for the odd length:
//     x[t] = (2 / n) Re( chirp[t] sum_k (c[k] chirp[k]) conj(chirp[t - k]) ),   chirp[j] = e^{i pi j^2 / n},
//     c[0] = Re Y[0] / 2, c[k] = Y[k] (k = 1 .. (n - 1) / 2)
// as a circular convolution of length M = 2^ceil(log2(2 n - 1)).
#include "common.h"

typedef double2 c128;

__device__ __forceinline__ c128 zmul(c128 a, c128 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// e^{i pi j^2 / n}: the phase j^2 mod 2 n is exact in 64-bit integers
__device__ __forceinline__ c128 f64_chirp(long long j, int n) {
  const unsigned long long r = (unsigned long long)(j * j) % (unsigned long long)(2 * (long long)n);
  double s, c;
  sincospi((double)r / (double)n, &s, &c);
  return make_double2(c, s);
}

// one radix-2 Stockham pass of `rows` transforms of length N (Ns = 1, 2, 4, ...: out-of-place, natural order in and out
// after log2 N passes); sign -1: forward
__global__ __launch_bounds__(256) void k_f64_pass(const c128* __restrict__ in, c128* __restrict__ out, int N, int Ns, double sign) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N / 2) return;
  const size_t o = (size_t)blockIdx.y * N;
  const int k = j & (Ns - 1);
  double s, c;
  sincospi(sign * (double)k / (double)Ns, &s, &c);
  const c128 a = in[o + j], b = zmul(make_double2(c, s), in[o + j + N / 2]);
  const int j0 = ((j - k) << 1) + k;
  out[o + j0] = make_double2(a.x + b.x, a.y + b.y);
  out[o + j0 + Ns] = make_double2(a.x - b.x, a.y - b.y);
}

// log2 N passes, ping-pong between a and b; returns the buffer that holds the result
static c128* f64_fft(c128* a, c128* b, int rows, int N, double sign, hipStream_t st) {
  for (int Ns = 1; Ns < N; Ns <<= 1) {
    hipLaunchKernelGGL(k_f64_pass, dim3((N / 2 + 255) / 256, rows), dim3(256), 0, st, a, b, N, Ns, sign);
    c128* t = a; a = b; b = t;
  }
  return a;
}

// ---- rfft of real rows, zero-padded to nfft (a power of two): the first kout bins
__global__ __launch_bounds__(256) void k_f64_real_in(const double* __restrict__ x, int ldx, int len, int N, c128* __restrict__ a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  a[(size_t)blockIdx.y * N + j] = make_double2(j < len ? x[(size_t)blockIdx.y * ldx + j] : 0.0, 0.0);
}
__global__ __launch_bounds__(256) void k_f64_bins_out(const c128* __restrict__ a, int N, int kout, c128* __restrict__ X) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kout) return;
  X[(size_t)blockIdx.y * kout + k] = a[(size_t)blockIdx.y * N + k];
}

extern "C" size_t gfdn_f64_fft_work_bytes(int rows, int M) {
  if (rows <= 0 || M <= 0) return 0;
  return (size_t)2 * rows * M * sizeof(c128);
}
extern "C" int gfdn_irfft_odd_f64_length(int n) {
  if (n < 3 || !(n & 1)) return 0;
  return next_pow2(2 * n - 1);
}

extern "C" int gfdn_rfft_pow2_f64(const double* x, int ldx, int len, int rows, int nfft, double* X_c128, int kout, void* work,
                                  void* stream) {
  if (!x || !X_c128 || !work || rows <= 0 || len <= 0 || ldx < len || nfft < 2 || (nfft & (nfft - 1)) || len > nfft ||
      kout <= 0 || kout > nfft / 2 + 1)
    return GFDN_E_BADARG;
  if (rows > 65535) return GFDN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  c128* a = (c128*)work;
  c128* b = a + (size_t)rows * nfft;
  hipLaunchKernelGGL(k_f64_real_in, dim3((nfft + 255) / 256, rows), dim3(256), 0, st, x, ldx, len, nfft, a);
  c128* r = f64_fft(a, b, rows, nfft, -1.0, st);
  hipLaunchKernelGGL(k_f64_bins_out, dim3((kout + 255) / 256, rows), dim3(256), 0, st, r, nfft, kout, (c128*)X_c128);
  GFDN_LAUNCH_CHECK();
  return 0;
}

// ---- Bluestein's chirp spectrum of length M for the odd length n (once per n)
__global__ __launch_bounds__(256) void k_f64_chirp_seq(int n, int M, c128* __restrict__ b) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const int h = (n - 1) / 2;
  c128 v = make_double2(0.0, 0.0);
  if (j < n) {
    v = f64_chirp(j, n);
    v.y = -v.y;
  } else if (j >= M - h) {
    v = f64_chirp(M - j, n);
    v.y = -v.y;
  }
  b[j] = v;
}

extern "C" int gfdn_irfft_odd_f64_plan(int n, double* bhat_c128, void* work, void* stream) {
  const int M = gfdn_irfft_odd_f64_length(n);
  if (!M || !bhat_c128 || !work) return GFDN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  c128* a = (c128*)work;
  c128* b = a + M;
  hipLaunchKernelGGL(k_f64_chirp_seq, dim3((M + 255) / 256), dim3(256), 0, st, n, M, a);
  c128* r = f64_fft(a, b, 1, M, -1.0, st);
  hipError_t e = hipMemcpyAsync(bhat_c128, r, (size_t)M * sizeof(c128), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return (int)e;
  GFDN_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void k_f64_blu_in(const c128* __restrict__ X, int ldx, const c128* __restrict__ filt, int n,
                                                    int M, c128* __restrict__ a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= M) return;
  const int h = (n - 1) / 2;
  c128 v = make_double2(0.0, 0.0);
  if (k <= h) {
    c128 y = X[(size_t)blockIdx.y * ldx + k];
    if (filt) y = zmul(y, filt[k]);
    if (k == 0) y = make_double2(0.5 * y.x, 0.0);          // (an inverse real transform reads Re Y[0] only)
    v = zmul(y, f64_chirp(k, n));
  }
  a[(size_t)blockIdx.y * M + k] = v;
}
__global__ __launch_bounds__(256) void k_f64_blu_mul(c128* __restrict__ a, const c128* __restrict__ bhat, int M) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= M) return;
  const size_t o = (size_t)blockIdx.y * M + k;
  a[o] = zmul(a[o], bhat[k]);
}
__global__ __launch_bounds__(256) void k_f64_blu_out(const c128* __restrict__ a, int n, int M, float* __restrict__ out32,
                                                     double* __restrict__ out64, int ldo) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const c128 v = zmul(a[(size_t)blockIdx.y * M + t], f64_chirp(t, n));
  const double x = v.x * (2.0 / ((double)n * (double)M));   // (1 / M: the unnormalised inverse transform of length M)
  if (out32) out32[(size_t)blockIdx.y * ldo + t] = (float)x;
  if (out64) out64[(size_t)blockIdx.y * ldo + t] = x;
}

extern "C" int gfdn_irfft_odd_f64(const double* X_c128, int ldx, const double* filt_c128, int rows, int n,
                                  const double* bhat_c128, float* out32, double* out64, int ldo, void* work, void* stream) {
  const int M = gfdn_irfft_odd_f64_length(n);
  if (!M || !X_c128 || !bhat_c128 || !work || (!out32 && !out64) || rows <= 0 || ldx < (n + 1) / 2 || ldo < n)
    return GFDN_E_BADARG;
  if (rows > 65535) return GFDN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  c128* a = (c128*)work;
  c128* b = a + (size_t)rows * M;
  const dim3 gm((M + 255) / 256, rows);
  hipLaunchKernelGGL(k_f64_blu_in, gm, dim3(256), 0, st, (const c128*)X_c128, ldx, (const c128*)filt_c128, n, M, a);
  c128* r = f64_fft(a, b, rows, M, -1.0, st);
  hipLaunchKernelGGL(k_f64_blu_mul, gm, dim3(256), 0, st, r, (const c128*)bhat_c128, M);
  c128* r2 = f64_fft(r, r == a ? b : a, rows, M, 1.0, st);
  hipLaunchKernelGGL(k_f64_blu_out, dim3((n + 255) / 256, rows), dim3(256), 0, st, r2, n, M, out32, out64, ldo);
  GFDN_LAUNCH_CHECK();
  return 0;
}
