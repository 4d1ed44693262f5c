// Tile-configuration sweep of gemm_kernel over the hot-path GEMM shapes (run on MI355X):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/hip/sweep_gemm tests/hip/sweep_gemm.hip && tests/hip/sweep_gemm
// Prints one line per (shape, config) with the time of a back-to-back launch, best config first.
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include "exp_lds32.h"
#include "exp_ldsp.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static float *g_buf; static hipStream_t g_st;
static int g_lds_target = 0;  // total LDS bytes per workgroup requested (occupancy cap experiment), 0 = none
struct Res { std::string cfg; double us; int gx, gy; };
template <int TN, int TM, int WK, int WN, int WM>
static void run(std::vector<Res> &out, int M, int N, int K, int ntaps, int T) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), CF = K / 16, KF = CF * ntaps;
  if (WK > 1 && KF < 2 * WK) return;           // nothing to split
  if (TM * WM > 2 * MT || TN * WN > 2 * NT) return;  // mostly padding
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * CF * 256, ysz = (size_t)MT * NT * 256;
  if ((wsz + 2 * xsz + ysz + 64) * 4 > ((size_t)3 << 30)) return;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.W = g_buf; a.X = g_buf + wsz; a.Y = g_buf + wsz + 2 * xsz; a.Xdstride = ntaps > 1 ? xsz : 0;
  a.par = ntaps > 1 ? (int *)(g_buf + wsz + 2 * xsz + ysz) : nullptr;
  a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.XF = CF; a.MT = MT; a.M = M; a.T = T; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1; a.halo = ntaps - 1;
  dim3 grid(cdiv(NT, TN * WN), cdiv(MT, TM * WM));
  const int st_lds = WK > 1 ? WK * WN * WM * TN * TM * 1024 : 0;
  const unsigned dyn = g_lds_target > st_lds ? g_lds_target - st_lds : 0;
  auto launch = [&] { gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM, dyn, g_st>>>(a); };
  for (int i = 0; i < 3; ++i) launch();
  hipStreamSynchronize(g_st);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch();
  hipStreamSynchronize(g_st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  char nm[64]; snprintf(nm, sizeof nm, "<%d,%d,%d,%d,%d>", TN, TM, WK, WN, WM);
  out.push_back({nm, us, (int)grid.x, (int)grid.y});
}
template <int BMT, int BNT, int KC>
static void run_lds(std::vector<Res> &out, int M, int N, int K, int ntaps, int T) {
  int MT = cdiv(M, 16), NT = cdiv(N, 16), CF = K / 16, KF = CF * ntaps;
  if (KF % KC || BMT > 2 * MT || BNT > 2 * NT) return;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * CF * 256, ysz = (size_t)MT * NT * 256;
  if ((wsz + 2 * xsz + ysz + 64) * 4 > ((size_t)3 << 30)) return;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.W = g_buf; a.X = g_buf + wsz; a.Y = g_buf + wsz + 2 * xsz; a.Xdstride = ntaps > 1 ? xsz : 0;
  a.par = ntaps > 1 ? (int *)(g_buf + wsz + 2 * xsz + ysz) : nullptr;
  a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.XF = CF; a.MT = MT; a.M = M; a.T = T; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1; a.halo = ntaps - 1;
  dim3 grid(cdiv(NT, BNT), cdiv(MT, BMT));
  const int st_lds = 2 * (BMT + BNT) * KC * 1024;
  const unsigned dyn = g_lds_target > st_lds ? g_lds_target - st_lds : 0;
  auto launch = [&] { gemm_lds_kernel<BMT, BNT, KC, PRE_NONE><<<grid, 256, dyn, g_st>>>(a); };
  for (int i = 0; i < 3; ++i) launch();
  hipStreamSynchronize(g_st);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch();
  hipStreamSynchronize(g_st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  char nm[64]; snprintf(nm, sizeof nm, "LDS<%d,%d,%d>", BMT, BNT, KC);
  out.push_back({nm, us, (int)grid.x, (int)grid.y});
}
template <int BMT, int BNT, int NS>
static void run_ldsp(std::vector<Res> &out, int M, int N, int K, int ntaps, int T) {  // persistent form, 2 workgroups per CU
  int MT = cdiv(M, 16), NT = cdiv(N, 16), CF = K / 16, KF = CF * ntaps;
  if (KF % 2 || BMT > 2 * MT || BNT > 2 * NT) return;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * CF * 256, ysz = (size_t)MT * NT * 256;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.W = g_buf; a.X = g_buf + wsz; a.Y = g_buf + wsz + 2 * xsz; a.Xdstride = ntaps > 1 ? xsz : 0;
  a.par = ntaps > 1 ? (int *)(g_buf + wsz + 2 * xsz + ysz) : nullptr;
  a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.XF = CF; a.MT = MT; a.M = M; a.T = T; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1; a.halo = ntaps - 1;
  const int tx = cdiv(NT, BNT), ty = cdiv(MT, BMT);
  const int G = std::min(tx * ty, 512) & ~7;
  const int st_lds = NS * (BMT + BNT) * 2 * 1024;
  const unsigned dyn = g_lds_target > st_lds ? g_lds_target - st_lds : 0;
  auto launch = [&] { gemm_ldsp_kernel<BMT, BNT, 2, PRE_NONE, NS><<<G, 256, dyn, g_st>>>(a, tx, ty); };
  for (int i = 0; i < 3; ++i) launch();
  hipStreamSynchronize(g_st);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch();
  hipStreamSynchronize(g_st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  char nm[64]; snprintf(nm, sizeof nm, "LDSP<%d,%d,ns%d>", BMT, BNT, NS);
  out.push_back({nm, us, G, 1});
}
template <int WM, int WN, int KC>
static void run_lds32(std::vector<Res> &out, int M, int N, int K, int ntaps, int T) {
  constexpr int BMT = 4 * WM, BNT = 4 * WN;
  int MT = cdiv(M, 16), NT = cdiv(N, 16), CF = K / 16, KF = CF * ntaps;
  if (KF % KC || BMT > 2 * MT || BNT > 2 * NT) return;
  size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * CF * 256, ysz = (size_t)MT * NT * 256;
  if ((wsz + 2 * xsz + ysz + 64) * 4 > ((size_t)3 << 30)) return;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.W = g_buf; a.X = g_buf + wsz; a.Y = g_buf + wsz + 2 * xsz; a.Xdstride = ntaps > 1 ? xsz : 0;
  a.par = ntaps > 1 ? (int *)(g_buf + wsz + 2 * xsz + ysz) : nullptr;
  a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.XF = CF; a.MT = MT; a.M = M; a.T = T; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1; a.halo = ntaps - 1;
  dim3 grid(cdiv(NT, BNT), cdiv(MT, BMT));
  auto launch = [&] { gemm_lds32_kernel<WM, WN, KC><<<grid, 256, 0, g_st>>>(a); };
  for (int i = 0; i < 3; ++i) launch();
  hipStreamSynchronize(g_st);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch();
  hipStreamSynchronize(g_st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  char nm[64]; snprintf(nm, sizeof nm, "L32<%d,%d,%d>", WM, WN, KC);
  out.push_back({nm, us, (int)grid.x, (int)grid.y});
}
static void sweep(const char *name, int M, int N, int K, int ntaps, int T) {
  std::vector<Res> r;
  run_lds32<1, 1, 2>(r, M, N, K, ntaps, T); run_lds32<2, 1, 2>(r, M, N, K, ntaps, T); run_lds32<1, 2, 2>(r, M, N, K, ntaps, T);
  run_lds32<2, 2, 2>(r, M, N, K, ntaps, T); run_lds32<2, 2, 4>(r, M, N, K, ntaps, T); run_lds32<1, 1, 4>(r, M, N, K, ntaps, T);
  run_lds<8, 4, 2>(r, M, N, K, ntaps, T); run_lds<8, 8, 2>(r, M, N, K, ntaps, T); run_lds<4, 4, 2>(r, M, N, K, ntaps, T);
  run_lds<8, 4, 4>(r, M, N, K, ntaps, T); run_lds<4, 4, 4>(r, M, N, K, ntaps, T); run_lds<8, 2, 2>(r, M, N, K, ntaps, T);
  run_lds<4, 2, 2>(r, M, N, K, ntaps, T); run_lds<4, 8, 2>(r, M, N, K, ntaps, T);
  run<1, 1, 8, 1, 1>(r, M, N, K, ntaps, T); run<1, 2, 8, 1, 1>(r, M, N, K, ntaps, T); run<1, 4, 8, 1, 1>(r, M, N, K, ntaps, T);
  run<1, 1, 4, 1, 1>(r, M, N, K, ntaps, T); run<1, 2, 4, 1, 1>(r, M, N, K, ntaps, T); run<1, 4, 4, 1, 1>(r, M, N, K, ntaps, T);
  run<2, 4, 4, 1, 1>(r, M, N, K, ntaps, T); run<2, 4, 2, 1, 2>(r, M, N, K, ntaps, T); run<2, 2, 4, 1, 1>(r, M, N, K, ntaps, T);
  run<2, 4, 1, 2, 2>(r, M, N, K, ntaps, T); run<2, 4, 1, 1, 4>(r, M, N, K, ntaps, T); run<1, 4, 1, 1, 4>(r, M, N, K, ntaps, T);
  run<1, 1, 1, 1, 4>(r, M, N, K, ntaps, T); run<2, 2, 1, 2, 2>(r, M, N, K, ntaps, T); run<1, 4, 1, 2, 2>(r, M, N, K, ntaps, T);
  run<1, 2, 1, 2, 2>(r, M, N, K, ntaps, T); run<2, 4, 2, 2, 1>(r, M, N, K, ntaps, T);
  // round 3: fat register tiles (16 accumulator tiles per wave: twice the MFMAs per operand byte of the 2x4 tile)
  run<4, 4, 1, 2, 2>(r, M, N, K, ntaps, T); run<4, 4, 1, 1, 4>(r, M, N, K, ntaps, T); run<2, 8, 1, 1, 4>(r, M, N, K, ntaps, T);
  run<4, 4, 1, 2, 1>(r, M, N, K, ntaps, T); run<4, 4, 1, 1, 2>(r, M, N, K, ntaps, T); run<2, 8, 1, 2, 2>(r, M, N, K, ntaps, T);
  run<4, 2, 1, 2, 2>(r, M, N, K, ntaps, T); run<2, 4, 1, 2, 4>(r, M, N, K, ntaps, T); run<4, 4, 1, 2, 4>(r, M, N, K, ntaps, T);
  std::sort(r.begin(), r.end(), [](const Res &a, const Res &b) { return a.us < b.us; });
  double fl = 2.0 * M * N * (double)K * ntaps;
  printf("%-28s M=%-6d N=%-5d K=%dx%-4d |", name, M, N, ntaps, K);
  for (size_t i = 0; i < r.size() && i < 7; ++i) printf(" %s %.1fus(%dx%d)", r[i].cfg.c_str(), r[i].us, r[i].gx, r[i].gy);
  printf(" | best %.1f TF\n", fl / r[0].us * 1e-6);
}
int main(int argc, char **argv) {
  hipStreamCreate(&g_st);
  hipMalloc(&g_buf, (size_t)3 << 30); hipMemset(g_buf, 0, (size_t)3 << 30);
  if (argc > 2) g_lds_target = atoi(argv[2]);
  if (argc > 1 && !strcmp(argv[1], "pmc")) {  // a handful of (shape, tile) pairs for a rocprofv3 --pmc pass
    std::vector<Res> r;
    const int R = 1024;
    run_lds<4, 2, 2>(r, 30 * R, 256, 128, 2, 480);   // seanet.convtr3
    run_lds<4, 4, 2>(r, R, 2048, 512, 1, 16);        // mimi.ff1
    run<2, 4, 4, 1, 1>(r, R, 512, 512, 7, 16);       // seanet.conv0
    run_lds<4, 4, 2>(r, 30 * R, 64, 128, 3, 480);    // seanet.res2a
    run_lds<8, 8, 2>(r, 30 * R, 256, 128, 2, 480);   // convtr3 on the large tile
    run_lds<4, 4, 2>(r, 30 * R, 256, 128, 2, 480);   // convtr3 on the tile the pipeline uses
    run_lds<4, 2, 2>(r, R, 1536, 512, 2, 16);        // seanet.convtr1
    run_lds<4, 8, 2>(r, 6 * R, 640, 256, 2, 96);     // seanet.convtr2
    run_ldsp<4, 4, 3>(r, 30 * R, 256, 128, 2, 480);  // convtr3, persistent
    run_ldsp<4, 2, 3>(r, 30 * R, 256, 128, 2, 480);
    run_ldsp<4, 2, 3>(r, R, 1536, 512, 2, 16);       // convtr1, persistent
    run_ldsp<4, 8, 2>(r, 6 * R, 640, 256, 2, 96);    // convtr2, persistent
    run_ldsp<4, 4, 3>(r, R, 2048, 512, 1, 16);       // mimi.ff1, persistent
    for (auto &x : r) printf("%s %.1f us (%dx%d)\n", x.cfg.c_str(), x.us, x.gx, x.gy);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "codec")) {  // only the codec shapes (round-3 fat-tile experiment)
    const int R = 1024;
    sweep("mimi.qkv", R, 1536, 512, 1, 16); sweep("mimi.ff1", R, 2048, 512, 1, 16); sweep("mimi.ff2", R, 512, 2048, 1, 16);
    sweep("seanet.conv0", R, 512, 512, 7, 16); sweep("seanet.convtr1", R, 1536, 512, 2, 16);
    sweep("seanet.res1a", 6 * R, 128, 256, 3, 96); sweep("seanet.convtr2", 6 * R, 640, 256, 2, 96);
    sweep("seanet.res2a", 30 * R, 64, 128, 3, 480); sweep("seanet.convtr3", 30 * R, 256, 128, 2, 480);
    sweep("seanet.res3a", 120 * R, 32, 64, 3, 1920);
    return 0;
  }
  const int Bs[] = {64};
  for (int B : Bs) {
    printf("---- batch %d\n", B);
    sweep("lm.qkv", B, 3072, 1024, 1, 16); sweep("lm.out", B, 1024, 1024, 1, 16);
    sweep("lm.ff1", B, 4096, 1024, 1, 16); sweep("lm.ff2", B, 1024, 4096, 1, 16);
    sweep("flow.512", B, 512, 512, 1, 16); sweep("flow.adaln", B, 10240, 512, 1, 16); sweep("flow.head", B, 528, 1024, 1, 16);
    int R = 16 * B;
    sweep("mimi.qkv", R, 1536, 512, 1, 16); sweep("mimi.out", R, 512, 512, 1, 16);
    sweep("mimi.ff1", R, 2048, 512, 1, 16); sweep("mimi.ff2", R, 512, 2048, 1, 16);
    sweep("seanet.conv0", R, 512, 512, 7, 16); sweep("seanet.convtr1", R, 1536, 512, 2, 16);
    sweep("seanet.res1a", 6 * R, 128, 256, 3, 96); sweep("seanet.res1b", 6 * R, 256, 128, 1, 96);
    sweep("seanet.convtr2", 6 * R, 640, 256, 2, 96); sweep("seanet.res2a", 30 * R, 64, 128, 3, 480);
    sweep("seanet.res2b", 30 * R, 128, 64, 1, 480); sweep("seanet.convtr3", 30 * R, 256, 128, 2, 480);
    sweep("seanet.res3a", 120 * R, 32, 64, 3, 1920); sweep("seanet.res3b", 120 * R, 64, 32, 1, 1920);
    sweep("seanet.last", 120 * R, 1, 64, 3, 1920);
  }
  return 0;
}
