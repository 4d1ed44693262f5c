// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the two load kinds the hot path uses:
// plain and non-temporal 16 B/lane streaming reads of a buffer far larger than L2 + Infinity Cache.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tests/hip/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void read_plain(const f32x4 *p, f32x4 *out, long n) {
  f32x4 acc = {0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc.x == 12345.f) out[0] = acc;
}
__global__ void read_nt(const f32x4 *p, f32x4 *out, long n) {
  f32x4 acc = {0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += __builtin_nontemporal_load(p + i);
  if (acc.x == 12345.f) out[0] = acc;
}
__global__ void write_plain(f32x4 *p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (f32x4){1, 2, 3, 4};
}
int main() {
  const long bytes = 1L << 30, n = bytes / 16;
  f32x4 *p, *o;
  hipMalloc(&p, bytes); hipMalloc(&o, 64); hipMemset(p, 0, bytes);
  for (int r = 0; r < 2; ++r) {
    read_plain<<<2048, 256>>>(p, o, n);
    read_nt<<<2048, 256>>>(p, o, n);
    write_plain<<<2048, 256>>>(p, n);
  }
  hipDeviceSynchronize();
  printf("each kernel moves %ld bytes\n", bytes);
  return 0;
}
