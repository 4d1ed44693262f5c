// Why is the codec's FF1 (+LN fold, +GELU) 2x slower in the step than the bare GEMM in sweep_gemm?
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstring>
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
template <int TN, int TM, int WK, int WN, int WM, int PRE>
static void bench(const char *name, int M, int N, int K, int act, bool cold, hipStream_t st, float *buf, size_t nfl) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * KF * 256, ysz = (size_t)MT * NT * 256;
  size_t per = wsz + xsz + ysz + 2 * NT * 16;
  int ncopy = cold ? (int)std::min<size_t>(nfl / per, 96) : 1;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.XF = KF; a.MT = MT; a.M = M; a.T = 16; a.epi = EPI_STORE; a.act = act; a.YF = NT; a.xstride = 1; a.ln_eps = 1e-5f;
  dim3 grid(cdiv(NT, TN * WN), cdiv(MT, TM * WM));
  auto launch = [&](int i) {
    float *b = buf + (size_t)(i % ncopy) * per;
    a.W = b; a.X = b + wsz; a.Y = b + wsz + xsz; a.ln_s = b + wsz + xsz + ysz; a.ln_c = a.ln_s + NT * 16;
    gemm_kernel<TN, TM, WK, WN, WM, PRE><<<grid, 64 * WK * WN * WM, 0, st>>>(a);
  };
  for (int i = 0; i < ncopy; ++i) launch(i);
  hipStreamSynchronize(st);
  const int R = cold ? 2 * ncopy : 40;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch(i);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  printf("%-34s %s pre=%d act=%d grid %4dx%-3d %7.2f us  %6.1f TF\n", name, cold ? "cold" : "warm", PRE, act, grid.x, grid.y, us, 2.0 * M * N * K / us * 1e-6);
}
template <int BMT, int BNT, int PRE>
static void bench_lds(const char *name, int M, int N, int K, int act, bool cold, hipStream_t st, float *buf, size_t nfl) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * KF * 256, ysz = (size_t)MT * NT * 256;
  size_t per = wsz + xsz + ysz + 2 * NT * 16;
  int ncopy = cold ? (int)std::min<size_t>(nfl / per, 96) : 1;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.XF = KF; a.MT = MT; a.M = M; a.T = 16; a.epi = EPI_STORE; a.act = act; a.YF = NT; a.xstride = 1; a.ln_eps = 1e-5f;
  dim3 grid(cdiv(NT, BNT), cdiv(MT, BMT));
  auto launch = [&](int i) {
    float *b = buf + (size_t)(i % ncopy) * per;
    a.W = b; a.X = b + wsz; a.Y = b + wsz + xsz; a.ln_s = b + wsz + xsz + ysz; a.ln_c = a.ln_s + NT * 16;
    gemm_lds_kernel<BMT, BNT, 2, PRE><<<grid, 256, 0, st>>>(a);
  };
  for (int i = 0; i < ncopy; ++i) launch(i);
  hipStreamSynchronize(st);
  const int R = cold ? 2 * ncopy : 40;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch(i);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  printf("%-34s %s pre=%d act=%d grid %4dx%-3d %7.2f us  %6.1f TF\n", name, cold ? "cold" : "warm", PRE, act, grid.x, grid.y, us, 2.0 * M * N * K / us * 1e-6);
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  const size_t nfl = (size_t)3 << 28;
  float *buf; hipMalloc(&buf, nfl * 4); hipMemset(buf, 0, nfl * 4);
  for (int cold = 0; cold < 2; ++cold) {
    bench<2, 4, 1, 2, 2, PRE_NONE>("ff1 <2,4,1,2,2>", 1024, 2048, 512, ACT_NONE, cold, st, buf, nfl);
    bench<2, 4, 1, 2, 2, PRE_NONE>("ff1 <2,4,1,2,2> gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench<2, 4, 1, 2, 2, PRE_LNFOLD>("ff1 <2,4,1,2,2> ln", 1024, 2048, 512, ACT_NONE, cold, st, buf, nfl);
    bench<2, 4, 1, 2, 2, PRE_LNFOLD>("ff1 <2,4,1,2,2> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench<2, 4, 4, 1, 1, PRE_LNFOLD>("ff1 <2,4,4,1,1> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench<2, 2, 4, 1, 1, PRE_LNFOLD>("ff1 <2,2,4,1,1> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench<2, 4, 1, 1, 4, PRE_LNFOLD>("ff1 <2,4,1,1,4> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench_lds<4, 4, PRE_LNFOLD>("ff1 LDS<4,4> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench_lds<4, 8, PRE_LNFOLD>("ff1 LDS<4,8> ln gelu", 1024, 2048, 512, ACT_GELU, cold, st, buf, nfl);
    bench_lds<4, 4, PRE_NONE>("ff1 LDS<4,4>", 1024, 2048, 512, ACT_NONE, cold, st, buf, nfl);
    bench<2, 4, 4, 1, 1, PRE_NONE>("ff2 <2,4,4,1,1>", 1024, 512, 2048, ACT_NONE, cold, st, buf, nfl);
    bench<2, 2, 4, 1, 1, PRE_NONE>("ff2 <2,2,4,1,1>", 1024, 512, 2048, ACT_NONE, cold, st, buf, nfl);
  }
  return 0;
}
