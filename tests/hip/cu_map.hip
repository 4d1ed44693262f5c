// Which physical (XCC, SE, CU) does bit i of a hipExtStreamCreateWithCUMask mask enable?  (run on MI355X)
//   hipcc --offload-arch=gfx950 -O2 -o tests/hip/cu_map tests/hip/cu_map.hip && tests/hip/cu_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void where_kernel(uint32_t *out) {
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // keep the CU busy a little so that a multi-CU mask spreads its workgroups
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
}
int main() {
  uint32_t *d; CK(hipMalloc(&d, 4096 * 8));
  std::vector<uint32_t> h(4096 * 2);
  printf("bit : xcc se sh cu\n");
  for (int bit = 0; bit < 256; ++bit) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mask[bit >> 5] = 1u << (bit & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    where_kernel<<<8, 64, 0, s>>>(d);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, 8 * 8, hipMemcpyDeviceToHost));
    std::set<uint32_t> seen;
    for (int i = 0; i < 8; ++i) seen.insert(((h[2 * i + 1] & 15) << 16) | (((h[2 * i] >> 13) & 7) << 8) | (((h[2 * i] >> 12) & 1) << 4) | ((h[2 * i] >> 8) & 15));
    printf("%3d :", bit);
    for (uint32_t v : seen) printf(" xcc%u se%u sh%u cu%u", v >> 16, (v >> 8) & 7, (v >> 4) & 1, v & 15);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
