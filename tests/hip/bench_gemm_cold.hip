// Decode GEMMs with COLD weights: every launch streams a different copy of W (NCOPY * |W| >> 256 MiB Infinity
// Cache), as in the real step where 379 MB of weights stream once per frame.  -DPTTS_ABLATE={0,1,4}.
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstring>
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
template <int TN, int TM, int WK, int WN, int WM>
static void bench(const char *name, int M, int N, int K, hipStream_t st, float *buf, size_t buf_floats, int gy_split) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * KF * 256, ysz = (size_t)MT * NT * 256;
  int ncopy = (int)((buf_floats - xsz - ysz) / wsz);
  if (ncopy > 64) ncopy = 64;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.X = buf; a.Y = buf + xsz; float *w0 = buf + xsz + ysz;
  a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.XF = KF; a.MT = MT; a.M = M; a.T = 16; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1;
  dim3 grid(cdiv(NT, TN * WN), cdiv(MT, TM * WM));
  const int R = 3 * ncopy;
  for (int i = 0; i < ncopy; ++i) { a.W = w0 + (size_t)i * wsz; gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM, 0, st>>>(a); }
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) { a.W = w0 + (size_t)(i % ncopy) * wsz; gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM, 0, st>>>(a); }
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  printf("ablate=%d cold(%2d copies, %4.0f MB) %-30s grid %4dx%-3d %7.2f us  %6.2f TB/s(W)\n", PTTS_ABLATE, ncopy, ncopy * wsz * 4e-6, name,
         grid.x, grid.y, us, wsz * 4.0 / us * 1e-6);
  (void)gy_split;
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  const size_t nfl = (size_t)3 << 28;  // 3 GiB
  float *buf; hipMalloc(&buf, nfl * 4); hipMemset(buf, 0, nfl * 4);
  bench<1, 1, 8, 1, 1>("qkv B=1   <1,1,8,1,1>", 1, 3072, 1024, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("ff1 B=1   <1,1,8,1,1>", 1, 4096, 1024, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("ff2 B=1   <1,1,8,1,1>", 1, 1024, 4096, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("qkv B=16  <1,1,8,1,1>", 16, 3072, 1024, st, buf, nfl, 0);
  bench<1, 2, 4, 1, 1>("qkv B=64  <1,2,4,1,1>", 64, 3072, 1024, st, buf, nfl, 0);
  bench<1, 4, 4, 1, 1>("qkv B=64  <1,4,4,1,1>", 64, 3072, 1024, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("qkv B=64  <1,1,8,1,1>", 64, 3072, 1024, st, buf, nfl, 0);
  bench<1, 4, 8, 1, 1>("qkv B=64  <1,4,8,1,1>", 64, 3072, 1024, st, buf, nfl, 0);
  bench<1, 4, 4, 1, 1>("ff1 B=64  <1,4,4,1,1>", 64, 4096, 1024, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("ff2 B=64  <1,1,8,1,1>", 64, 1024, 4096, st, buf, nfl, 0);
  bench<1, 4, 8, 1, 1>("ff2 B=64  <1,4,8,1,1>", 64, 1024, 4096, st, buf, nfl, 0);
  bench<1, 1, 8, 1, 1>("out B=64  <1,1,8,1,1>", 64, 1024, 1024, st, buf, nfl, 0);
  return 0;
}
