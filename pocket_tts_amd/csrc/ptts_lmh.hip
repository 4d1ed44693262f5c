// FlowLM Linear layers with bf16 WEIGHTS and bf16-rounded activation operands on v_mfma_f32_16x16x32_bf16
// (PTTS_LM_BF16; SURVEY 8(f).4 "bf16 / int8 per-channel LM weights"; reference hook: quantization.py:91-128, which
// offers CPU dynamic int8 only - no reference counterpart, parity unpinned, judged by SNR + frame counts).
// The kernel is gemm_kernel<.., WF = 2> of ptts_kernels.h: the residual stream, the LayerNorm statistics, the
// accumulation and every epilogue stay fp32; only the two MFMA operands are bf16.  At batch 64 the fp32 decode GEMMs sit
// on the fp32-MFMA side of the ridge (32 flop per weight byte against 19.7); this format halves the weight bytes AND
// takes the matrix pipe out of the picture (16x the fp32 rate).
#include "ptts_ext.h"

static inline int cdiv_(long a, long b) { return (int)((a + b - 1) / b); }

template <int TN, int TM, int WK, int WN, int WM>
static void launch_cfg_b16(hipStream_t st, const GemmArgs &a, int pre, unsigned dyn) {
  dim3 grid(cdiv_(a.NT, TN * WN), cdiv_(a.MT, TM * WM));
  dim3 block(64 * WK * WN * WM);
  if (pre == PRE_LNFOLD) gemm_kernel<TN, TM, WK, WN, WM, PRE_LNFOLD, 2><<<grid, block, dyn, st>>>(a);
  else gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE, 2><<<grid, block, dyn, st>>>(a);
}

void launch_gemm_b16(hipStream_t st, const GemmArgs &a, int pre, int cfg, unsigned dyn) {
  switch (cfg) {
    case 0: launch_cfg_b16<1, 1, 8, 1, 1>(st, a, pre, dyn); break;
    case 1: launch_cfg_b16<1, 2, 4, 1, 1>(st, a, pre, dyn); break;
    case 2: launch_cfg_b16<1, 4, 4, 1, 1>(st, a, pre, dyn); break;
    case 7: launch_cfg_b16<2, 4, 4, 1, 1>(st, a, pre, dyn); break;
    case 10: launch_cfg_b16<2, 2, 4, 1, 1>(st, a, pre, dyn); break;
    case 11: launch_cfg_b16<1, 1, 4, 1, 1>(st, a, pre, dyn); break;
    default: launch_cfg_b16<2, 4, 1, 2, 2>(st, a, pre, dyn); break;  // 3: no K split, any even KF
  }
}

// one workgroup per n-tile: converts the tile's KF fragments pairwise and (optionally) sums the rounded rows
__global__ __launch_bounds__(256) void pack_weight_b16_kernel(const float *src, __bf16 *dst, float *ln_s, int KF) {
  __shared__ float part[256];
  const int nt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;  // q: which fragment pair of a group of 4
  const float *s = src + (size_t)nt * KF * 256;
  __bf16 *d = dst + (size_t)nt * (KF / 2) * 512;
  float acc = 0.f;
  for (int kb = q; kb < KF / 2; kb += 4) {
    const f32x4 lo = *(const f32x4 *)(s + ((size_t)(2 * kb) * 64 + lane) * 4);
    const f32x4 hi = *(const f32x4 *)(s + ((size_t)(2 * kb + 1) * 64 + lane) * 4);
    const bf16x4 l = to_bf16x4(lo), h = to_bf16x4(hi);
    *(bf16x4 *)(d + ((size_t)kb * 64 + lane) * 8) = l;
    *(bf16x4 *)(d + ((size_t)kb * 64 + lane) * 8 + 4) = h;
    const f32x4 lf = from_bf16x4(l), hf = from_bf16x4(h);
    acc += ((lf.x + lf.y) + (lf.z + lf.w)) + ((hf.x + hf.y) + (hf.z + hf.w));
  }
  if (!ln_s) return;
  part[tid] = acc;
  __syncthreads();
  if (tid < 16) {  // row n = tid: lanes tid, tid + 16, tid + 32, tid + 48 of each of the 4 fragment-pair groups, fixed order
    float t = 0.f;
    for (int qq = 0; qq < 4; ++qq)
      for (int g = 0; g < 4; ++g) t += part[qq * 64 + 16 * g + tid];
    ln_s[16 * nt + tid] = t;
  }
}

void pack_weight_b16(hipStream_t st, const float *src, void *dst, float *ln_s, int NT, int KF) {
  pack_weight_b16_kernel<<<NT, 256, 0, st>>>(src, (__bf16 *)dst, ln_s, KF);
}
