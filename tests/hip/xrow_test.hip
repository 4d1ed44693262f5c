// Evidence for the inline-asm cross-row exchange in ptts_kernels.h (xrow_swap16 / xrow_swap32).
// gfx950 has v_permlane16_swap_b32 / v_permlane32_swap_b32: rows of 16 lanes (or halves of 32) of two VGPRs are
// exchanged in one VALU instruction.  ROCm 7.2's clang exposes them as __builtin_amdgcn_permlane{16,32}_swap(old, src,
// fi, bc) returning both results as a 2-vector.  This program runs the builtin and the inline-asm form on the same
// inputs and compares both with the lane arithmetic worked out by hand:
//   swap16(a, b): odd rows of a <-> even rows of b     swap32(a, b): upper half of a <-> lower half of b
// Exit code 0 = the inline-asm form (the one the library uses) is right; the builtin's status is printed, and with
// `--require-builtin` a wrong builtin fails too (use that to notice when a later compiler fixes it and the asm can go).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_builtin(const unsigned *in, unsigned *out) {
  const int l = threadIdx.x;
  const unsigned a = in[l], b = in[64 + l];
  const u32x2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const u32x2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r16.x; out[64 + l] = r16.y; out[128 + l] = r32.x; out[192 + l] = r32.y;
}
__global__ void k_asm(const unsigned *in, unsigned *out) {
  const int l = threadIdx.x;
  unsigned a = in[l], b = in[64 + l], c = a, d = b;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  out[l] = a; out[64 + l] = b; out[128 + l] = c; out[192 + l] = d;
}

int main(int argc, char **argv) {
  const bool require_builtin = argc > 1 && !strcmp(argv[1], "--require-builtin");
  std::vector<unsigned> in(128), want(256), got(256);
  for (int l = 0; l < 64; ++l) { in[l] = 1000 + l; in[64 + l] = 2000 + l; }
  for (int l = 0; l < 64; ++l) {
    const int row = l >> 4;
    // swap16: vdst row 1 <-> src row 0, vdst row 3 <-> src row 2
    want[l] = (row & 1) ? in[64 + l - 16] : in[l];
    want[64 + l] = (row & 1) ? in[64 + l] : in[l + 16];
    // swap32: vdst upper half <-> src lower half
    want[128 + l] = l >= 32 ? in[64 + l - 32] : in[l];
    want[192 + l] = l >= 32 ? in[64 + l] : in[l + 32];
  }
  unsigned *d_in, *d_out;
  if (hipMalloc(&d_in, 512) != hipSuccess || hipMalloc(&d_out, 1024) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
  hipMemcpy(d_in, in.data(), 512, hipMemcpyHostToDevice);
  int rc = 0;
  for (int pass = 0; pass < 2; ++pass) {
    hipMemset(d_out, 0, 1024);
    if (pass == 0) k_asm<<<1, 64>>>(d_in, d_out); else k_builtin<<<1, 64>>>(d_in, d_out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(got.data(), d_out, 1024, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) bad[i / 64] += got[i] != want[i];
    const bool second_equals_first = !memcmp(&got[0], &got[64], 256) && !memcmp(&got[128], &got[192], 256);
    printf("%s: swap16 first %d second %d, swap32 first %d second %d mismatching lanes%s\n", pass ? "builtin   " : "inline asm",
           bad[0], bad[1], bad[2], bad[3], second_equals_first ? "  (second result == first result: the miscompile)" : "");
    const bool ok = !(bad[0] | bad[1] | bad[2] | bad[3]);
    if (pass == 0 && !ok) rc = 1;
    if (pass == 1 && !ok && require_builtin) rc = 1;
  }
  return rc;
}
