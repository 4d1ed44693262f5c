// Native unit test of the device kernels against plain C++ loops (runs on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tests/hip/test_kernels tests/hip/test_kernels.hip && tests/hip/test_kernels
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include "exp_lds32.h"
#include "exp_ldsp.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
template <typename T> T *dev(const std::vector<T> &h) {
  T *d; CK(hipMalloc(&d, h.size() * sizeof(T) + 256)); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}
template <typename T> T *dzero(size_t n) { T *d; CK(hipMalloc(&d, n * sizeof(T) + 256)); CK(hipMemset(d, 0, n * sizeof(T))); return d; }
static std::vector<float> host(const float *d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }
static double maxerr(const std::vector<float> &a, const std::vector<float> &b) {
  double e = 0; for (size_t i = 0; i < a.size(); ++i) e = std::max(e, (double)std::fabs(a[i] - b[i])); return e;
}
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static int fails = 0;
static void report(const char *name, double err, double tol) {
  printf("%-40s err %.3e %s\n", name, err, err < tol ? "ok" : "FAIL");
  if (!(err < tol)) ++fails;
}

template <int TN, int TM, int WK, int WN, int WM>
static void run_gemm(const GemmArgs &a) {
  dim3 grid(cdiv(a.NT, TN * WN), cdiv(a.MT, TM * WM));
  gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE><<<grid, 64 * WK * WN * WM>>>(a);
  CK(hipDeviceSynchronize());
}

// plain GEMM: Y[M][N] = X[M][K] W[N][K]^T + bias, through pack + to_fm + gemm + from_fm
static void test_gemm(int M, int N, int K, int cfg) {
  std::vector<float> X(M * K), W(N * K), b(N), Y(M * N);
  for (auto &v : X) v = frand();
  for (auto &v : W) v = frand();
  for (auto &v : b) v = frand();
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    double s = b[n]; for (int k = 0; k < K; ++k) s += (double)X[m * K + k] * W[n * K + k]; Y[m * N + n] = (float)s;
  }
  int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  float *dX = dev(X), *dW = dev(W), *db = dev(b);
  float *xfm = dzero<float>((size_t)MT * KF * 256), *wp = dzero<float>((size_t)NT * KF * 256), *bp = dzero<float>(NT * 16);
  float *yfm = dzero<float>((size_t)MT * NT * 256), *dY = dzero<float>((size_t)M * NT * 16);
  to_fm_kernel<<<cdiv((long)MT * KF * 64, 256), 256>>>(dX, xfm, M, K, MT);
  long tot = (long)NT * KF * 256;
  pack_weight_kernel<<<cdiv(tot, 256), 256>>>(dW, wp, N, K, 1, 0, 0, 0, 0, KF, tot, nullptr, K);
  pack_bias_kernel<<<cdiv(NT * 16, 256), 256>>>(db, bp, N, 0, 0, 0, NT * 16);
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.W = wp; a.bias = bp; a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.X = xfm; a.XF = KF; a.MT = MT; a.M = M; a.T = 16;
  a.epi = EPI_STORE; a.Y = yfm; a.YF = NT;
  if (cfg >= 104 && cfg <= 106) {  // persistent LDS GEMM: 8 resident workgroups walk all tiles
    const int bnt = cfg == 104 ? 4 : cfg == 105 ? 2 : 8;
    const int tx = cdiv(a.NT, bnt), ty = cdiv(a.MT, 4);
    const int G = std::min(8, tx * ty);
    if (cfg == 104) gemm_ldsp_kernel<4, 4, 2, PRE_NONE, 3><<<G, 256>>>(a, tx, ty);
    else if (cfg == 105) gemm_ldsp_kernel<4, 2, 2, PRE_NONE, 3><<<G, 256>>>(a, tx, ty);
    else gemm_ldsp_kernel<4, 8, 2, PRE_NONE, 2><<<G, 256>>>(a, tx, ty);
    CK(hipDeviceSynchronize());
  } else if (cfg == 102 || cfg == 103) {  // 32x32x2 MFMA tiles: workgroup = 64 x 64 (cfg 102) or 128 x 128 (cfg 103)
    const int b16 = cfg == 102 ? 4 : 8;
    dim3 grid(cdiv(a.NT, b16), cdiv(a.MT, b16));
    if (cfg == 102) gemm_lds32_kernel<1, 1, 2><<<grid, 256>>>(a); else gemm_lds32_kernel<2, 2, 2><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
  } else if (cfg >= 100) {
    dim3 grid(cdiv(a.NT, cfg == 100 ? 4 : 8), cdiv(a.MT, 8));
    if (cfg == 100) gemm_lds_kernel<8, 4, 2, PRE_NONE><<<grid, 256>>>(a); else gemm_lds_kernel<8, 8, 2, PRE_NONE><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
  } else if (cfg == 0) run_gemm<1, 1, 8, 1, 1>(a);
  else if (cfg == 1) run_gemm<1, 2, 8, 1, 1>(a);
  else if (cfg == 2) run_gemm<1, 4, 8, 1, 1>(a);
  else if (cfg == 3) run_gemm<2, 4, 1, 2, 2>(a);
  else if (cfg == 4) run_gemm<2, 4, 1, 1, 4>(a);
  else run_gemm<1, 4, 1, 1, 4>(a);
  from_fm_kernel<<<cdiv((long)M * NT * 4, 256), 256>>>(yfm, dY, M, NT * 16, NT, 0);
  CK(hipDeviceSynchronize());
  auto got = host(dY, (size_t)M * NT * 16);
  std::vector<float> g2(M * N);
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) g2[m * N + n] = got[(size_t)m * NT * 16 + n];
  char nm[128]; snprintf(nm, sizeof nm, "gemm M=%d N=%d K=%d cfg=%d", M, N, K, cfg);
  report(nm, maxerr(g2, Y), 1e-3 * std::sqrt((double)K / 64));
}

// causal conv with halo from the previous-frame buffer
static void test_conv(int B, int T, int C, int N, int ntaps, int cfg) {
  int M = B * T, halo = ntaps - 1;
  std::vector<float> Xc(M * C), Xp(M * C), W(N * C * ntaps), b(N), Y(M * N);
  for (auto &v : Xc) v = frand(); for (auto &v : Xp) v = frand(); for (auto &v : W) v = frand(); for (auto &v : b) v = frand();
  for (int bb = 0; bb < B; ++bb) for (int t = 0; t < T; ++t) for (int n = 0; n < N; ++n) {
    double s = b[n];
    for (int tap = 0; tap < ntaps; ++tap) {
      int ts = t + tap - halo;
      const float *row = ts >= 0 ? &Xc[(bb * T + ts) * C] : &Xp[(bb * T + T + ts) * C];
      for (int c = 0; c < C; ++c) s += (double)row[c] * W[(n * C + c) * ntaps + tap];
    }
    Y[(bb * T + t) * N + n] = (float)s;
  }
  int MT = M / 16, NT = cdiv(N, 16), CF = C / 16, KF = CF * ntaps;
  float *dXc = dev(Xc), *dXp = dev(Xp), *dW = dev(W), *db = dev(b);
  size_t xs = (size_t)MT * CF * 256;
  float *xfm = dzero<float>(2 * xs), *wp = dzero<float>((size_t)NT * KF * 256), *bp = dzero<float>(NT * 16);
  float *yfm = dzero<float>((size_t)MT * NT * 256), *dY = dzero<float>((size_t)M * NT * 16);
  int *par = dzero<int>(1); int one = 1; CK(hipMemcpy(par, &one, 4, hipMemcpyHostToDevice));  // parity 1: cur = buf 1
  to_fm_kernel<<<cdiv((long)MT * CF * 64, 256), 256>>>(dXc, xfm + xs, M, C, MT);
  to_fm_kernel<<<cdiv((long)MT * CF * 64, 256), 256>>>(dXp, xfm, M, C, MT);
  long tot = (long)NT * KF * 256;
  pack_weight_kernel<<<cdiv(tot, 256), 256>>>(dW, wp, N, C, ntaps, 0, 0, 0, 0, KF, tot, nullptr, C);
  pack_bias_kernel<<<cdiv(NT * 16, 256), 256>>>(db, bp, N, 0, 0, 0, NT * 16);
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.W = wp; a.bias = bp; a.NT = NT; a.KF = KF; a.CF = CF; a.ntaps = ntaps; a.X = xfm; a.Xdstride = xs; a.XF = CF; a.MT = MT; a.M = M; a.T = T; a.par = par; a.xstride = 1; a.halo = ntaps - 1;
  a.epi = EPI_STORE; a.Y = yfm; a.YF = NT;
  if (cfg == 104 || cfg == 105) {
    const int bnt = cfg == 104 ? 4 : 2;
    const int tx = cdiv(a.NT, bnt), ty = cdiv(a.MT, 4);
    const int G = std::min(8, tx * ty);
    if (cfg == 104) gemm_ldsp_kernel<4, 4, 2, PRE_NONE, 3><<<G, 256>>>(a, tx, ty);
    else gemm_ldsp_kernel<4, 2, 2, PRE_NONE, 3><<<G, 256>>>(a, tx, ty);
    CK(hipDeviceSynchronize());
  } else if (cfg == 102) {
    dim3 grid(cdiv(a.NT, 4), cdiv(a.MT, 4));
    gemm_lds32_kernel<1, 1, 2><<<grid, 256>>>(a);
  } else if (cfg >= 100) {
    dim3 grid(cdiv(a.NT, 4), cdiv(a.MT, cfg == 100 ? 8 : 4));
    if (cfg == 100) gemm_lds_kernel<8, 4, 2, PRE_NONE><<<grid, 256>>>(a); else gemm_lds_kernel<4, 4, 2, PRE_NONE><<<grid, 256>>>(a);
    CK(hipDeviceSynchronize());
  } else if (cfg == 0) run_gemm<1, 1, 8, 1, 1>(a);
  else if (cfg == 2) run_gemm<1, 4, 8, 1, 1>(a);
  else if (cfg == 3) run_gemm<2, 4, 1, 2, 2>(a);
  else run_gemm<1, 4, 1, 1, 4>(a);
  from_fm_kernel<<<cdiv((long)M * NT * 4, 256), 256>>>(yfm, dY, M, NT * 16, NT, 0);
  CK(hipDeviceSynchronize());
  auto got = host(dY, (size_t)M * NT * 16);
  std::vector<float> g2(M * N);
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) g2[m * N + n] = got[(size_t)m * NT * 16 + n];
  char nm[128]; snprintf(nm, sizeof nm, "conv B=%d T=%d C=%d N=%d k=%d cfg=%d", B, T, C, N, ntaps, cfg);
  report(nm, maxerr(g2, Y), 2e-3);
}

template <int TN, int TM, int WK, int WN, int WM>
static void run_gemm_ln(const GemmArgs &a) {
  dim3 grid(cdiv(a.NT, TN * WN), cdiv(a.MT, TM * WM));
  gemm_kernel<TN, TM, WK, WN, WM, PRE_LNFOLD><<<grid, 64 * WK * WN * WM>>>(a);
  CK(hipDeviceSynchronize());
}

// LayerNorm folded into the GEMM: Y = LN(X; g, b, eps) W^T + bias
static void test_gemm_lnfold(int M, int N, int K, int cfg) {
  std::vector<float> X(M * K), W(N * K), g(K), b(K), bias(N), Y(M * N);
  for (auto &v : X) v = frand() * 2 + 0.3f; for (auto &v : W) v = frand(); for (auto &v : g) v = 1 + 0.2f * frand();
  for (auto &v : b) v = 0.2f * frand(); for (auto &v : bias) v = frand();
  for (int m = 0; m < M; ++m) {
    double mu = 0, var = 0; for (int k = 0; k < K; ++k) mu += X[m * K + k]; mu /= K;
    for (int k = 0; k < K; ++k) var += (X[m * K + k] - mu) * (X[m * K + k] - mu); var /= K;
    for (int n = 0; n < N; ++n) {
      double s = bias[n];
      for (int k = 0; k < K; ++k) s += ((X[m * K + k] - mu) / std::sqrt(var + 1e-5) * g[k] + b[k]) * W[n * K + k];
      Y[m * N + n] = (float)s;
    }
  }
  int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  float *dX = dev(X), *dW = dev(W), *dg = dev(g), *db = dev(b), *dbias = dev(bias);
  float *xfm = dzero<float>((size_t)MT * KF * 256), *wp = dzero<float>((size_t)NT * KF * 256);
  float *ls = dzero<float>(NT * 16), *lc = dzero<float>(NT * 16);
  float *yfm = dzero<float>((size_t)MT * NT * 256), *dY = dzero<float>((size_t)M * NT * 16);
  to_fm_kernel<<<cdiv((long)MT * KF * 64, 256), 256>>>(dX, xfm, M, K, MT);
  long tot = (long)NT * KF * 256;
  pack_weight_kernel<<<cdiv(tot, 256), 256>>>(dW, wp, N, K, 1, 0, 0, 0, 0, KF, tot, dg, K);
  fold_ln_kernel<<<N, 64>>>(dW, dg, db, dbias, ls, lc, N, K, 0);
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.W = wp; a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.X = xfm; a.XF = KF; a.MT = MT; a.M = M; a.T = 16;
  a.ln_s = ls; a.ln_c = lc; a.ln_eps = 1e-5f; a.epi = EPI_STORE; a.Y = yfm; a.YF = NT;
  if (cfg == 0) run_gemm_ln<1, 1, 8, 1, 1>(a);
  else if (cfg == 2) run_gemm_ln<1, 4, 8, 1, 1>(a);
  else if (cfg == 3) run_gemm_ln<2, 4, 1, 2, 2>(a);
  else run_gemm_ln<2, 4, 4, 1, 1>(a);
  from_fm_kernel<<<cdiv((long)M * NT * 4, 256), 256>>>(yfm, dY, M, NT * 16, NT, 0);
  CK(hipDeviceSynchronize());
  auto got = host(dY, (size_t)M * NT * 16);
  std::vector<float> g2(M * N);
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) g2[m * N + n] = got[(size_t)m * NT * 16 + n];
  char nm[128]; snprintf(nm, sizeof nm, "gemm+LNfold M=%d N=%d K=%d cfg=%d", M, N, K, cfg);
  report(nm, maxerr(g2, Y), 2e-4 * std::sqrt((double)K / 64));
}

static void test_attn(int BH, int Tq, int off, int ctx, int ring, int splits, int nw = 1) {
  const int H = 2; int B = BH / H;
  int T = off + Tq, cap = ring ? ring : cdiv(T, 16) * 16, QB = cdiv(Tq, 16);
  std::vector<float> Q((size_t)BH * Tq * 64), K((size_t)BH * T * 64), V((size_t)BH * T * 64), O((size_t)B * Tq * H * 64);
  for (auto &v : Q) v = frand(); for (auto &v : K) v = frand(); for (auto &v : V) v = frand();
  for (int bh = 0; bh < BH; ++bh) for (int t = 0; t < Tq; ++t) {
    int pq = off + t; std::vector<double> s(T, -1e300); double mx = -1e300;
    for (int k = 0; k < T; ++k) if (k <= pq && (ctx <= 0 || pq - k < ctx)) {
      double d = 0; for (int e = 0; e < 64; ++e) d += (double)Q[((size_t)bh * Tq + t) * 64 + e] * K[((size_t)bh * T + k) * 64 + e];
      s[k] = d * 0.125; mx = std::max(mx, s[k]);
    }
    double l = 0; for (int k = 0; k < T; ++k) if (s[k] > -1e299) l += std::exp(s[k] - mx);
    int b = bh / H, h = bh % H;
    for (int e = 0; e < 64; ++e) {
      double o = 0; for (int k = 0; k < T; ++k) if (s[k] > -1e299) o += std::exp(s[k] - mx) / l * V[((size_t)bh * T + k) * 64 + e];
      O[((size_t)(b * Tq + t) * H + h) * 64 + e] = (float)o;
    }
  }
  // device layouts
  std::vector<float> Qb((size_t)BH * QB * 4 * 256, 0.f), Kc((size_t)BH * cap * 64, 0.f), Vc((size_t)BH * cap * 64, 0.f);
  for (int bh = 0; bh < BH; ++bh) for (int t = 0; t < Tq; ++t) for (int d = 0; d < 64; ++d)
    Qb[((((size_t)bh * QB + (t >> 4)) * 4 + (d >> 4)) * 64 + 16 * ((d & 15) >> 2) + (t & 15)) * 4 + (d & 3)] = Q[((size_t)bh * Tq + t) * 64 + d];
  for (int bh = 0; bh < BH; ++bh) for (int k = 0; k < T; ++k) {
    if (ring && k < T - ring) continue;  // overwritten long ago
    int slot = ring ? k % ring : k;
    for (int d = 0; d < 64; ++d) { Kc[((size_t)bh * cap + slot) * 64 + d] = K[((size_t)bh * T + k) * 64 + d]; Vc[((size_t)bh * cap + slot) * 64 + d] = V[((size_t)bh * T + k) * 64 + d]; }
  }
  std::vector<int> offs(B, off);
  int MT = cdiv(B * Tq, 16), YF = H * 4;
  AttnArgs a; memset(&a, 0, sizeof a);
  a.Q = dev(Qb); a.Kc = dev(Kc); a.Vc = dev(Vc); a.offset = dev(offs); a.H = H; a.Tq = Tq; a.QB = QB; a.cap = cap; a.ring = ring; a.ctx = ctx; a.splits = splits;
  a.part = dzero<float>((size_t)BH * QB * splits * 16 * ATT_PSTRIDE); a.Y = dzero<float>((size_t)MT * YF * 256); a.YF = YF;
  if (nw == 4) attn_kernel<4><<<dim3(BH, QB, splits), 256>>>(a);
  else if (nw == 2) attn_kernel<2><<<dim3(BH, QB, splits), 128>>>(a);
  else attn_kernel<1><<<dim3(BH, QB, splits), 64>>>(a);
  if (splits > 1) attn_combine_kernel<<<dim3(BH, QB), 256>>>(a);
  float *dY = dzero<float>((size_t)B * Tq * H * 64);
  from_fm_kernel<<<cdiv((long)B * Tq * H * 16, 256), 256>>>(a.Y, dY, B * Tq, H * 64, YF, 0);
  CK(hipDeviceSynchronize());
  char nm[128]; snprintf(nm, sizeof nm, "attn BH=%d Tq=%d off=%d ctx=%d ring=%d sp=%d nw=%d", BH, Tq, off, ctx, ring, splits, nw);
  report(nm, maxerr(host(dY, O.size()), O), 2e-5);
}

int main() {
  srand(1);
  test_gemm(3, 48, 64, 0);
  test_gemm(16, 128, 128, 0);
  test_gemm(20, 96, 256, 1);
  test_gemm(50, 80, 1024, 2);
  test_gemm(200, 130, 128, 3);
  test_gemm(100, 32, 64, 4);
  test_gemm(100, 1, 64, 5);
  test_gemm(1, 384, 4096, 0);
  test_gemm(200, 130, 128, 100); test_gemm(300, 200, 512, 101); test_gemm(1024, 64, 64, 100);
  test_gemm(2000, 130, 128, 104); test_gemm(1500, 200, 512, 105); test_gemm(4096, 64, 64, 104); test_gemm(3000, 300, 96, 106); test_gemm(100, 48, 64, 104);
  test_gemm(200, 130, 128, 102); test_gemm(300, 200, 512, 103); test_gemm(1024, 64, 64, 102); test_gemm(40, 48, 32, 102);
  test_conv(2, 16, 32, 48, 7, 0);
  test_conv(3, 32, 64, 32, 3, 2);
  test_conv(5, 48, 32, 70, 2, 3);
  test_conv(4, 96, 16, 1, 3, 5);
  test_conv(5, 48, 32, 70, 2, 100); test_conv(3, 32, 64, 96, 3, 101); test_conv(9, 16, 32, 64, 7, 100);
  test_conv(5, 48, 32, 70, 2, 102); test_conv(9, 16, 32, 64, 7, 102);
  test_conv(40, 48, 32, 70, 2, 104); test_conv(30, 32, 64, 96, 3, 105); test_conv(64, 16, 32, 64, 7, 104);
  test_gemm_lnfold(3, 48, 128, 0); test_gemm_lnfold(50, 96, 1024, 2); test_gemm_lnfold(200, 130, 512, 3);
  test_gemm_lnfold(130, 64, 1024, 7);
  test_attn(4, 1, 0, 0, 0, 1);
  test_attn(4, 1, 37, 0, 0, 1);
  test_attn(4, 1, 200, 0, 0, 5);
  test_attn(2, 10, 5, 0, 0, 1);
  test_attn(2, 39, 20, 0, 0, 2);
  test_attn(6, 16, 0, 40, 64, 1);
  test_attn(6, 16, 64, 40, 64, 2);
  test_attn(2, 16, 1600, 250, 272, 3);
  test_attn(2, 39, 20, 0, 0, 1, 2); test_attn(2, 39, 20, 0, 0, 2, 4); test_attn(6, 16, 64, 40, 64, 1, 4);
  test_attn(2, 16, 1600, 250, 272, 1, 4); test_attn(2, 16, 1600, 250, 272, 2, 2); test_attn(4, 7, 3, 0, 0, 1, 4);
  printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
