// Where does a decode GEMM spend its time?  In-kernel wall-clock stamps (100 MHz) of every wave of gemm_kernel at: entry (0),
// start of the K loop (1), end of the K loop = all MFMAs issued (2), after the K-split barrier (3), end of the epilogue (4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPTTS_STAMP -o tests/hip/stamp_gemm tests/hip/stamp_gemm.hip && tests/hip/stamp_gemm
// Weights are cold (a different copy per launch, caches flushed by reading 512 MB in between), activations warm, as in the step.
#include "../../pocket_tts_amd/csrc/ptts_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
__global__ void flush_kernel(const f32x4 *p, size_t n, float *sink) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s.x + s.y + s.z + s.w == 123.456f) *sink = s.x;
}
template <int TN, int TM, int WK, int WN, int WM, int PRE>
static void run(const char *name, int M, int N, int K, hipStream_t st, float *buf, size_t nfl, float *flush, size_t flush_n) {
  const int MT = cdiv(M, 16), NT = cdiv(N, 16), KF = K / 16;
  const size_t wsz = (size_t)NT * KF * 256, xsz = (size_t)MT * KF * 256, ysz = (size_t)MT * NT * 256;
  GemmArgs a; memset(&a, 0, sizeof a);
  a.X = buf; a.Y = buf + xsz; float *w0 = buf + xsz + ysz;
  const int ncopy = (int)std::min<size_t>(16, (nfl - xsz - ysz - (1 << 22)) / wsz);
  a.NT = NT; a.KF = KF; a.CF = KF; a.ntaps = 1; a.XF = KF; a.MT = MT; a.M = M; a.T = 16; a.epi = EPI_STORE; a.YF = NT; a.xstride = 1;
  a.ln_s = buf; a.ln_c = buf; a.ln_eps = 1e-5f;
  const dim3 grid(cdiv(NT, TN * WN), cdiv(MT, TM * WM)), block(64 * WK * WN * WM);
  const int nwaves = grid.x * grid.y * (block.x / 64);
  unsigned long long *d_st;
  hipMalloc(&d_st, (size_t)nwaves * 64);
  std::vector<unsigned long long> h((size_t)nwaves * 8);
  double acc[6] = {0, 0, 0, 0, 0, 0}, span = 0, skew = 0, evt = 0;
  const int R = 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < R + 1; ++r) {
    a.W = w0 + (size_t)(r % ncopy) * wsz;
    a.stamp = d_st;
    hipMemsetAsync(d_st, 0, (size_t)nwaves * 64, st);
    flush_kernel<<<2048, 256, 0, st>>>((const f32x4 *)flush, flush_n / 4, flush);
    flush_kernel<<<256, 256, 0, st>>>((const f32x4 *)a.X, xsz / 4, flush);  // activations warm (L2 / MALL), as after the producer
    hipEventRecord(e0, st);
    gemm_kernel<TN, TM, WK, WN, WM, PRE><<<grid, block, 0, st>>>(a);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    if (r == 0) continue;  // first launch: code object load
    float ms; hipEventElapsedTime(&ms, e0, e1); evt += ms * 1e3;
    hipMemcpy(h.data(), d_st, (size_t)nwaves * 64, hipMemcpyDeviceToHost);
    unsigned long long t0min = ~0ull, t0max = 0, t4max = 0;
    for (int w = 0; w < nwaves; ++w) { t0min = std::min(t0min, h[w * 8]); t0max = std::max(t0max, h[w * 8]); t4max = std::max(t4max, h[w * 8 + 4]); }
    span += (t4max - t0min) * 0.01; skew += (t0max - t0min) * 0.01;
    double s[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < nwaves; ++w) {
      s[0] += (h[w * 8] - t0min) * 0.01;                       // start delay of the wave
      s[1] += (h[w * 8 + 1] - h[w * 8]) * 0.01;                // prologue (kernarg, addresses, epilogue operand prefetch, LN-mod pre-pass)
      s[2] += (h[w * 8 + 2] - h[w * 8 + 1]) * 0.01;            // K loop
      s[3] += (WK > 1 ? (h[w * 8 + 3] - h[w * 8 + 2]) : 0) * 0.01;  // LDS park + barrier (K split)
      s[4] += (h[w * 8 + 4] - h[w * 8 + (WK > 1 ? 3 : 2)]) * 0.01;  // reduce + epilogue
    }
    for (int i = 0; i < 5; ++i) acc[i] += s[i] / nwaves;
  }
  const double mfma_us = (double)((KF / WK) * 4 * TN * TM) * 32 / 2400.0;  // issue cycles at 2.4 GHz
  printf("%-34s grid %3dx%-2d x%d waves | events %5.1f us | in-kernel span %5.1f (start skew %4.1f) | mean per wave: start +%4.1f, prologue %4.1f, K loop %5.1f "
         "(MFMA issue alone %4.1f), barrier %4.1f, reduce+epilogue %4.1f us\n",
         name, grid.x, grid.y, block.x / 64, evt / R, span / R, skew / R, acc[0] / R, acc[1] / R, acc[2] / R, mfma_us, acc[3] / R, acc[4] / R);
  hipFree(d_st);
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  const size_t nfl = (size_t)3 << 28;  // 3 GiB of zeros: weights (16 cold copies), activations, outputs
  float *buf; hipMalloc(&buf, nfl * 4); hipMemset(buf, 0, nfl * 4);
  const size_t flush_n = (size_t)128 << 20;  // 512 MB read between launches: L2 and Infinity Cache hold none of the weights
  float *flush; hipMalloc(&flush, flush_n * 4); hipMemset(flush, 0, flush_n * 4);
  run<2, 2, 4, 1, 1, PRE_LNFOLD>("qkv B=64 <2,2,4,1,1>+ln", 64, 3072, 1024, st, buf, nfl, flush, flush_n);
  run<1, 4, 4, 1, 1, PRE_LNFOLD>("qkv B=64 <1,4,4,1,1>+ln", 64, 3072, 1024, st, buf, nfl, flush, flush_n);
  run<2, 2, 4, 1, 1, PRE_LNFOLD>("ff1 B=64 <2,2,4,1,1>+ln", 64, 4096, 1024, st, buf, nfl, flush, flush_n);
  run<1, 1, 4, 1, 1, PRE_NONE>("out B=64 <1,1,4,1,1>", 64, 1024, 1024, st, buf, nfl, flush, flush_n);
  run<1, 1, 4, 1, 1, PRE_NONE>("ff2 B=64 <1,1,4,1,1>", 64, 1024, 4096, st, buf, nfl, flush, flush_n);
  run<1, 1, 8, 1, 1, PRE_NONE>("ff2 B=64 <1,1,8,1,1>", 64, 1024, 4096, st, buf, nfl, flush, flush_n);
  run<1, 1, 8, 1, 1, PRE_LNFOLD>("qkv B=1  <1,1,8,1,1>+ln", 1, 3072, 1024, st, buf, nfl, flush, flush_n);
  return 0;
}
