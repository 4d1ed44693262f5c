// Timing microbenchmark of gemm_kernel on the hot-path shapes (MI355X).  Build variants with
// -DPTTS_ABLATE={0,1,2,4} to see which resource bounds a configuration.
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
template <int TN, int TM, int WK, int WN, int WM>
static double bench(const char *name, int M, int N, int K, int ntaps, hipStream_t st, float *buf) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), CF = K / 16, KF = CF * ntaps;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * CF * 256, ysz = (size_t)MT * NT * 256;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.W = buf; a.X = buf + wsz; a.Y = buf + wsz + 2 * xsz; a.Xdstride = ntaps > 1 ? xsz : 0;
  int *par = (int *)(buf + wsz + 2 * xsz + ysz);
  a.par = ntaps > 1 ? par : nullptr;
  a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.XF = CF; a.MT = MT; a.M = M; a.T = 16; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1; a.halo = ntaps - 1;
  dim3 grid(cdiv(NT, TN * WN), cdiv(MT, TM * WM));
  for (int i = 0; i < 5; ++i) gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM, 0, st>>>(a);
  hipStreamSynchronize(st);
  const int R = 50;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM, 0, st>>>(a);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  double fl = 2.0 * M * N * (double)K * ntaps, by = 4.0 * ((double)N * K * ntaps + (double)M * K + (double)M * N);
  printf("ablate=%d %-34s grid %4dx%-3d %8.2f us  %7.2f TF  %7.1f GB/s(alg)\n", PTTS_ABLATE, name, grid.x, grid.y, us, fl / us * 1e-6, by / us * 1e-3);
  return us;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *buf; CK(hipMalloc(&buf, (size_t)1 << 30)); CK(hipMemset(buf, 0, (size_t)1 << 30));
  // Mimi, batch 64 (rows = 16 * 64 = 1024 and up)
  bench<2, 4, 1, 2, 2>("mimi qkv  1024x1536x512  <2,4,1,2,2>", 1024, 1536, 512, 1, st, buf);
  bench<2, 4, 1, 2, 2>("mimi ff1  1024x2048x512  <2,4,1,2,2>", 1024, 2048, 512, 1, st, buf);
  bench<2, 4, 1, 2, 2>("mimi ff2  1024x512x2048  <2,4,1,2,2>", 1024, 512, 2048, 1, st, buf);
  bench<2, 4, 1, 2, 2>("conv0     1024x512x(7x512)", 1024, 512, 512, 7, st, buf);
  bench<2, 4, 4, 1, 1>("mimi ff2  1024x512x2048  <2,4,4,1,1>", 1024, 512, 2048, 1, st, buf);
  bench<2, 4, 4, 1, 1>("conv0     1024x512x(7x512) <2,4,4,1,1>", 1024, 512, 512, 7, st, buf);
  bench<2, 4, 4, 1, 1>("mimi out  1024x512x512   <2,4,4,1,1>", 1024, 512, 512, 1, st, buf);
  bench<2, 4, 1, 2, 2>("mimi out  1024x512x512   <2,4,1,2,2>", 1024, 512, 512, 1, st, buf);
  bench<1, 4, 8, 1, 1>("mimi ff2  1024x512x2048  <1,4,8,1,1>", 1024, 512, 2048, 1, st, buf);
  bench<1, 4, 8, 1, 1>("conv0     1024x512x(7x512) <1,4,8,1,1>", 1024, 512, 512, 7, st, buf);
  bench<2, 4, 1, 2, 2>("convtr2   6144x640x(2x256)", 6144, 640, 256, 2, st, buf);
  bench<2, 4, 1, 2, 2>("convtr3  30720x256x(2x128)", 30720, 256, 128, 2, st, buf);
  bench<2, 4, 1, 2, 2>("res3b   122880x64x32", 122880, 64, 32, 1, st, buf);
  // FlowLM decode, batch 64
  bench<1, 4, 8, 1, 1>("lm qkv    64x3072x1024 <1,4,8,1,1>", 64, 3072, 1024, 1, st, buf);
  bench<1, 4, 8, 1, 1>("lm ff1    64x4096x1024 <1,4,8,1,1>", 64, 4096, 1024, 1, st, buf);
  bench<1, 1, 8, 1, 1>("lm ff2    64x1024x4096 <1,1,8,1,1>", 64, 1024, 4096, 1, st, buf);
  bench<1, 4, 8, 1, 1>("lm ff2    64x1024x4096 <1,4,8,1,1>", 64, 1024, 4096, 1, st, buf);
  // FlowLM decode, batch 1
  bench<1, 1, 8, 1, 1>("lm qkv     1x3072x1024 <1,1,8,1,1>", 1, 3072, 1024, 1, st, buf);
  bench<1, 1, 8, 1, 1>("lm ff1     1x4096x1024 <1,1,8,1,1>", 1, 4096, 1024, 1, st, buf);
  bench<1, 1, 8, 1, 1>("lm ff2     1x1024x4096 <1,1,8,1,1>", 1, 1024, 4096, 1, st, buf);
  return 0;
}
