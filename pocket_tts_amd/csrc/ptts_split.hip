// Codec GEMMs on ERROR-COMPENSATED bf16 (PTTS_CODEC_SPLIT): every Linear / Conv1d / ConvTranspose1d of the
// Mimi decoder (mimi_transformer.py:39-54, seanet.py:141-180, conv.py:93-163) as  w x ~ hi_w hi_x + hi_w lo_x + lo_w hi_x  with
// hi = bf16(v), lo = bf16(v - hi), three v_mfma_f32_16x16x32_bf16 per 16x16x32 block, fp32 accumulation.  The fp32 codec is
// bound by the fp32-input MFMA (157 TFLOP/s; 34.7 GFLOP per batch-64 frame = 221 us at peak); three bf16 MFMAs cost 48 matrix
// cycles where eight fp32 ones cost 256.  Relative error per product ~2^-16 (the dropped lo*lo term and the rounding of lo):
// 250x the fp32 chain's, 250x below plain bf16's.  Activations, LayerNorm statistics, attention, every epilogue and every
// buffer stay fp32 (the FM layout is untouched), so this is a drop-in for the fp32 codec's GEMM launches; the kernel is
// gemm_kernel<.., WF = 3> of ptts_kernels.h.  Reported BESIDE the fp32 headline, never instead of it.
#include "ptts_ext.h"

static inline int cdivs(long a, long b) { return (int)((a + b - 1) / b); }

template <int TN, int TM, int WK, int WN, int WM>
static void launch_cfg_split(hipStream_t st, const GemmArgs &a, int pre, unsigned dyn) {
  dim3 grid(cdivs(a.NT, TN * WN), cdivs(a.MT, TM * WM));
  dim3 block(64 * WK * WN * WM);
  if (pre == PRE_LNFOLD) gemm_kernel<TN, TM, WK, WN, WM, PRE_LNFOLD, 3><<<grid, block, dyn, st>>>(a);
  else gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE, 3><<<grid, block, dyn, st>>>(a);
}

bool split_cfg(int cfg) { return (cfg >= 1 && cfg <= 7) || cfg == 10 || cfg == 11 || cfg == 13 || cfg == 14; }

void launch_gemm_split(hipStream_t st, const GemmArgs &a, int pre, int cfg, unsigned dyn) {
  switch (cfg) {
    case 1: launch_cfg_split<1, 2, 4, 1, 1>(st, a, pre, dyn); break;
    case 2: launch_cfg_split<1, 4, 4, 1, 1>(st, a, pre, dyn); break;
    case 4: launch_cfg_split<2, 4, 1, 1, 4>(st, a, pre, dyn); break;
    case 5: launch_cfg_split<1, 4, 1, 1, 4>(st, a, pre, dyn); break;
    case 6: launch_cfg_split<1, 1, 1, 1, 4>(st, a, pre, dyn); break;
    case 7: launch_cfg_split<2, 4, 4, 1, 1>(st, a, pre, dyn); break;
    case 10: launch_cfg_split<2, 2, 4, 1, 1>(st, a, pre, dyn); break;
    case 11: launch_cfg_split<1, 1, 4, 1, 1>(st, a, pre, dyn); break;
    case 13: launch_cfg_split<1, 2, 1, 2, 2>(st, a, pre, dyn); break;
    case 14: launch_cfg_split<2, 4, 2, 2, 1>(st, a, pre, dyn); break;
    default: launch_cfg_split<2, 4, 1, 2, 2>(st, a, pre, dyn); break;  // 3
  }
}

__global__ __launch_bounds__(256) void pack_weight_split_kernel(const float *src, __bf16 *hi, __bf16 *lo, int KF) {
  const int nt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const float *s = src + (size_t)nt * KF * 256;
  for (int kb = q; kb < KF / 2; kb += 4) {
    const f32x4 a = *(const f32x4 *)(s + ((size_t)(2 * kb) * 64 + lane) * 4);
    const f32x4 b = *(const f32x4 *)(s + ((size_t)(2 * kb + 1) * 64 + lane) * 4);
    const bf16x4 ah = to_bf16x4(a), bh = to_bf16x4(b);
    const bf16x4 al = to_bf16x4(a - from_bf16x4(ah)), bl = to_bf16x4(b - from_bf16x4(bh));
    const size_t o = (((size_t)nt * (KF / 2) + kb) * 64 + lane) * 8;
    *(bf16x4 *)(hi + o) = ah; *(bf16x4 *)(hi + o + 4) = bh;
    *(bf16x4 *)(lo + o) = al; *(bf16x4 *)(lo + o + 4) = bl;
  }
}
void pack_weight_split(hipStream_t st, const float *src, void *hi, void *lo, int NT, int KF) {
  pack_weight_split_kernel<<<NT, 256, 0, st>>>(src, (__bf16 *)hi, (__bf16 *)lo, KF);
}
