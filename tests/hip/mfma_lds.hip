// What does the MFMA pipe deliver when every MFMA's operands come from fresh ds_read_b128 (as in gemm_lds_kernel's
// k-step) but nothing else happens: no DMA, no barriers?   hipcc --offload-arch=gfx950 -O3 -o tests/hip/mfma_lds tests/hip/mfma_lds.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WMT, int WNT, bool DBUF>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ f32x4 lds[32][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < 32; i += 4) lds[i][lane] = (f32x4){1.f, 2.f, (float)i, (float)lane};
  __syncthreads();
  f32x4 acc[WNT][WMT];
  for (int i = 0; i < WNT; ++i) for (int j = 0; j < WMT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 xa[WMT], wa[WNT], xb[WMT], wb[WNT];
  auto rd = [&](int kc, f32x4 *x, f32x4 *w) {
    for (int j = 0; j < WMT; ++j) x[j] = lds[(kc * 7 + j) & 31][lane];
    for (int i = 0; i < WNT; ++i) w[i] = lds[(kc * 5 + 16 + i) & 31][lane];
  };
  auto mm = [&](const f32x4 *x, const f32x4 *w) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][c], x[j][c], acc[i][j], 0, 0, 0);
  };
  if (DBUF) {
    rd(0, xa, wa);
    for (int it = 0; it < iters; it += 2) {
      rd(it + 1, xb, wb); __builtin_amdgcn_sched_barrier(0);
      mm(xa, wa); __builtin_amdgcn_sched_barrier(0);
      rd(it + 2, xa, wa); __builtin_amdgcn_sched_barrier(0);
      mm(xb, wb); __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int it = 0; it < iters; ++it) { rd(it, xa, wa); mm(xa, wa); }
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < WNT; ++i) for (int j = 0; j < WMT; ++j) s += acc[i][j];
  if (s.x == 123.456f) out[threadIdx.x] = s.x;
}
template <int WMT, int WNT, bool DBUF>
static void run(int wg_per_cu, int iters) {
  float *out; (void)hipMalloc(&out, 4096);
  const int blocks = 256 * wg_per_cu;
  k<WMT, WNT, DBUF><<<blocks, 256>>>(out, 16); (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  k<WMT, WNT, DBUF><<<blocks, 256>>>(out, iters); (void)hipDeviceSynchronize();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  double fl = (double)blocks * 4 * iters * WMT * WNT * 4 * 2048.0;
  printf("wave tile %dx%d dbuf=%d WG/CU=%d: %8.1f us %6.1f TFLOP/s\n", WMT, WNT, (int)DBUF, wg_per_cu, us, fl / us * 1e-6);
  (void)hipFree(out);
}
int main() {
  run<2, 2, false>(1, 4000); run<2, 2, true>(1, 4000); run<2, 2, false>(2, 4000); run<2, 2, true>(2, 4000);
  run<2, 4, false>(1, 2000); run<2, 4, true>(1, 2000); run<2, 4, true>(2, 2000);
  run<4, 4, false>(1, 1000); run<4, 4, true>(1, 1000); run<2, 1, true>(2, 8000); run<2, 1, true>(4, 8000);
  return 0;
}
