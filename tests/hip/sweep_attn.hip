// Decode-attention variants over the FlowLM cache shapes (run on MI355X):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=on -o tests/hip/sweep_attn tests/hip/sweep_attn.hip && tests/hip/sweep_attn
// Six "layers" of K/V (B*H*cap*64 floats each) are cycled so that no launch finds its cache in L2 / MALL, like a
// real step; prints us per launch, the achieved KV rate and the max |difference| against attn_decode_kernel<1>.
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include "exp_attn_v1.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static const int L = 6, H = 16;
static float *dK[L], *dV[L], *dQ, *dY, *dYref;
static int *dOff;
static hipStream_t st;

template <class F>
static double time_us(F launch) {
  for (int i = 0; i < 2 * L; ++i) launch(i % L);
  CK(hipStreamSynchronize(st));
  const int R = 10 * L;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) launch(i % L);
  CK(hipStreamSynchronize(st));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
}
static AttnArgs mk(int l, int cap, float *Y) {
  AttnArgs a; memset(&a, 0, sizeof a);
  a.Q = dQ; a.Kc = dK[l]; a.Vc = dV[l]; a.offset = dOff; a.H = H; a.Tq = 1; a.QB = 1; a.cap = cap; a.ring = 0; a.ctx = 0; a.splits = 1;
  a.Y = Y; a.YF = H * 4; a.h16 = 0;
  return a;
}
static double maxdiff(int B) {
  size_t n = (size_t)((B + 15) / 16) * 16 * H * 64;
  std::vector<float> x(n), y(n);
  CK(hipMemcpy(x.data(), dY, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y.data(), dYref, n * 4, hipMemcpyDeviceToHost));
  double d = 0;
  for (size_t i = 0; i < n; ++i) d = std::max(d, (double)std::fabs(x[i] - y[i]));
  return d;
}
template <int NW, int D>
static void run2(int B, int cap, double bytes) {
  CK(hipMemsetAsync(dY, 0, (size_t)((B + 15) / 16) * 16 * H * 64 * 4, st));
  double us = time_us([&](int l) { attn_decode2_kernel<NW, D><<<dim3(B * H, 1, 1), 64 * NW, 0, st>>>(mk(l, cap, dY)); });
  attn_decode2_kernel<NW, D><<<dim3(B * H, 1, 1), 64 * NW, 0, st>>>(mk(0, cap, dY));
  CK(hipStreamSynchronize(st));
  printf("   v2<NW=%d,D=%d> %6.1f us %5.2f TB/s  maxdiff %.2e\n", NW, D, us, bytes / us * 1e-6, maxdiff(B));
}
template <int NW>
static void run1(int B, int cap, double bytes, bool ref) {
  double us = time_us([&](int l) { attn_decode_kernel<NW><<<dim3(B * H, 1, 1), 64 * NW, 0, st>>>(mk(l, cap, ref ? dYref : dY)); });
  attn_decode_kernel<NW><<<dim3(B * H, 1, 1), 64 * NW, 0, st>>>(mk(0, cap, ref ? dYref : dY));
  CK(hipStreamSynchronize(st));
  printf("   v1<NW=%d>     %6.1f us %5.2f TB/s%s\n", NW, us, bytes / us * 1e-6, ref ? "  (reference)" : "");
}
__global__ void fill_rand(float *p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = counter_normal(seed, (unsigned)(i >> 20), (unsigned)(i & 0xfffff)) * scale;
}
__global__ void set_off(int *p, int n, int base, int ragged) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = base - (ragged ? (i * 7) % 23 : 0);
}
// ---- the codec's attention: 16 queries per (sequence, head) and frame against a ring of 272 slots, window 250
static float *dPart;
template <int NW>
static void run_mimi(int B, int Hm, int ring, int splits, double bytes, bool ref) {
  auto mkm = [&](int l, float *Y) {
    AttnArgs a; memset(&a, 0, sizeof a);
    a.Q = dQ; a.Kc = dK[l]; a.Vc = dV[l]; a.offset = dOff; a.H = Hm; a.Tq = 16; a.QB = 1; a.cap = ring; a.ring = ring; a.ctx = 250; a.splits = splits;
    a.part = dPart; a.Y = Y; a.YF = Hm * 4; a.h16 = 0;
    return a;
  };
  float *Y = ref ? dYref : dY;
  auto go = [&](int l) {
    AttnArgs a = mkm(l, Y);
    attn_kernel<NW><<<dim3(B * Hm, 1, splits), 64 * NW, 0, st>>>(a);
    if (splits > 1) attn_combine_kernel<<<dim3(B * Hm, 1), 256, 0, st>>>(a);
  };
  double us = time_us(go);
  go(0);
  CK(hipStreamSynchronize(st));
  double d = 0;
  if (!ref) {
    size_t n = (size_t)B * 16 * Hm * 64;
    std::vector<float> x(n), y(n);
    CK(hipMemcpy(x.data(), dY, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y.data(), dYref, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) d = std::max(d, (double)std::fabs(x[i] - y[i]));
  }
  printf("   attn<NW=%d> splits=%d %6.1f us %5.2f TB/s  maxdiff %.2e%s\n", NW, splits, us, bytes / us * 1e-6, d, ref ? " (reference, incl. combine)" : "");
}
static void mimi_section() {
  const int Hm = 8, ring = 272;
  CK(hipMalloc(&dPart, (size_t)64 * Hm * 16 * 16 * ATT_PSTRIDE * 4));
  for (int B : {64, 8, 1}) {
    set_off<<<1, 64, 0, st>>>(dOff, 64, 1008, 0);
    const double bytes = (double)B * Hm * 265 * 64 * 4 * 2;
    printf("codec attention B=%d (%.1f MB)\n", B, bytes * 1e-6);
    const int cur = B == 64 ? 2 : (B == 8 ? 8 : 8);
    {
      auto a = [&](int l) {
        AttnArgs x; memset(&x, 0, sizeof x);
        x.Q = dQ; x.Kc = dK[l]; x.Vc = dV[l]; x.offset = dOff; x.H = Hm; x.Tq = 16; x.QB = 1; x.cap = ring; x.ring = ring; x.ctx = 250; x.splits = cur;
        x.part = dPart; x.Y = dYref; x.YF = Hm * 4; x.h16 = 0;
        attn_kernel_v1<<<dim3(B * Hm, 1, cur), 64, 0, st>>>(x);
        attn_combine_kernel<<<dim3(B * Hm, 1), 256, 0, st>>>(x);
      };
      double us = time_us(a);
      a(0);
      CK(hipStreamSynchronize(st));
      printf("   round-1 kernel splits=%d + combine %6.1f us  (reference)\n", cur, us);
    }
    for (int S : {1, 2, 4, 8}) {
      if (S > 1 && B * Hm * S > 4096) continue;
      run_mimi<1>(B, Hm, ring, S, bytes, false);
      run_mimi<2>(B, Hm, ring, S, bytes, false);
      run_mimi<4>(B, Hm, ring, S, bytes, false);
    }
  }
}
int main(int argc, char **argv) {
  CK(hipStreamCreate(&st));
  const int Bmax = 64, cap = 288;
  const size_t per = (size_t)Bmax * H * cap * 64;
  for (int l = 0; l < L; ++l) {
    CK(hipMalloc(&dK[l], per * 4)); CK(hipMalloc(&dV[l], per * 4));
    fill_rand<<<(per + 255) / 256, 256, 0, st>>>(dK[l], per, 11 + l, 1.0f);
    fill_rand<<<(per + 255) / 256, 256, 0, st>>>(dV[l], per, 31 + l, 1.0f);
  }
  const size_t qn = (size_t)Bmax * H * 4 * 256;
  CK(hipMalloc(&dQ, qn * 4)); CK(hipMalloc(&dY, 1024 * 1024 * 4)); CK(hipMalloc(&dYref, 1024 * 1024 * 4)); CK(hipMalloc(&dOff, Bmax * 4));
  fill_rand<<<(qn + 255) / 256, 256, 0, st>>>(dQ, qn, 5, 1.0f);
  CK(hipMemsetAsync(dYref, 0, 64 * 1024 * 4 * 4, st));
  if (argc > 1 && !strcmp(argv[1], "mimi")) { mimi_section(); return 0; }
  for (int B : {64, 32, 8, 1})
    for (int pq : {159, 220, 283})
      for (int ragged : {0, 1}) {
        if (ragged && (B != 64 || pq != 220)) continue;
        set_off<<<1, 64, 0, st>>>(dOff, Bmax, pq, ragged);
        const double bytes = (double)B * H * (pq + 1 - (ragged ? 11 : 0)) * 64 * 4 * 2;
        printf("B=%d keys=%d%s  (%.1f MB)\n", B, pq + 1, ragged ? " ragged" : "", bytes * 1e-6);
        run1<1>(B, cap, bytes, true);
        if (B <= 32) run1<2>(B, cap, bytes, false);
        if (B <= 8) { run1<4>(B, cap, bytes, false); run1<8>(B, cap, bytes, false); }
        if (B >= 32) { run2<1, 3>(B, cap, bytes); run2<1, 4>(B, cap, bytes); run2<1, 6>(B, cap, bytes); run2<1, 8>(B, cap, bytes); }
        run2<2, 3>(B, cap, bytes); run2<2, 4>(B, cap, bytes); run2<2, 6>(B, cap, bytes);
        run2<4, 3>(B, cap, bytes); run2<4, 4>(B, cap, bytes);
        if (B <= 8) { run2<8, 2>(B, cap, bytes); run2<8, 3>(B, cap, bytes); run2<8, 4>(B, cap, bytes); }
      }
  return 0;
}
