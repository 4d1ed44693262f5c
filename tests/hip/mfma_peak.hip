// Practical fp32 MFMA peak on this chip: waves that only issue v_mfma_f32_16x16x4_f32 from registers.
//   hipcc --offload-arch=gfx950 -O3 -o tests/hip/mfma_peak tests/hip/mfma_peak.hip && tests/hip/mfma_peak
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_only(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s.x == 123.456f) out[threadIdx.x] = s.x;
}
template <int NACC>
static void run(int waves_per_simd, int iters) {
  float *out; hipMalloc(&out, 4096);
  const int blocks = 256 * waves_per_simd;  // 4 waves per block = one per SIMD
  mfma_only<NACC><<<blocks, 256>>>(out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  mfma_only<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  double fl = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("NACC=%2d waves/SIMD=%d: %8.1f us  %6.1f TFLOP/s\n", NACC, waves_per_simd, us, fl / us * 1e-6);
  hipFree(out);
}
int main() {
  run<4>(1, 20000); run<8>(1, 10000); run<16>(1, 5000); run<4>(2, 20000); run<8>(2, 10000); run<16>(4, 5000);
  return 0;
}
