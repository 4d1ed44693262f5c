// Launchers of the kernel families that live in their own translation units (compiled in parallel with ptts.hip, see
// pocket_tts_amd/_lib.py::build).  Plain C++ functions: ptts.hip never instantiates these kernels itself.
#pragma once
#include "ptts_kernels.h"

// ---- ptts_lmh.hip: FlowLM Linear layers with bf16 weights (PTTS_LM_BF16) ------------------------------------------
// register-staged K-split / 2-D tile configurations of gemm_kernel<.., WF = 2>; `cfg` indexes the same table as
// launch_by_cfg in ptts.hip (only the q8_cfg() subset exists), `pre` is PRE_NONE or PRE_LNFOLD
void launch_gemm_b16(hipStream_t st, const GemmArgs &a, int pre, int cfg, unsigned dyn_lds);
// fp32 packed image [NT][KF][64][4] (LayerNorm gain already multiplied in) -> bf16 image [NT][KF/2][64][8]; with
// ln_s != null also the fold vector s[n] = sum_k W'[n][k] of the ROUNDED weights (NT * 16 floats)
void pack_weight_b16(hipStream_t st, const float *src, void *dst, float *ln_s, int NT, int KF);

// ---- ptts_fp8.hip: SEANet decoder convolutions on the fp8 MFMA (PTTS_CODEC_FP8) --------------------------------------
// checkpoint conv weight -> e4m3 image [NT][ntaps * C/32][64][8 bytes] + one fp32 scale per output channel (padded to
// NT * 16); same (mode, cout, stride) conventions as pack_weight_kernel
void pack_weight_f8(hipStream_t st, const float *src, void *dst, float *wscale, int N, int C, int ntaps, int mode, int cout, int stride);
// implicit GEMM on e4m3 operands (GemmArgs::X = e4m3 activations in the "FM8" layout, W = e4m3 weights, wscale = per-channel
// weight scale, xs = activation scale of X, yinv = 1 / scale of Y when Y is e4m3 (yf8 = 1), else Y is bf16 FMH)
void launch_gemm_f8(hipStream_t st, const GemmArgs &a, unsigned dyn_lds);
// max |x| over n bf16 values (calibration of the static activation scales), atomically max-ed into *out (as float bits)
void amax_bf16(hipStream_t st, const void *x, long n, float *out);

// ---- ptts_split.hip: codec GEMMs on error-compensated ("split") bf16 (PTTS_CODEC_SPLIT) ----------------------------------
// every register-staged configuration of gemm_kernel<.., WF = 3> (cfg = index into ptts.hip's table; LDS-staged ones do not exist)
void launch_gemm_split(hipStream_t st, const GemmArgs &a, int pre, int cfg, unsigned dyn_lds);
bool split_cfg(int cfg);
// fp32 packed image [NT][KF][64][4] -> hi = bf16(w) and lo = bf16(w - hi) images, each [NT][KF/2][64][8]
void pack_weight_split(hipStream_t st, const float *src, void *hi, void *lo, int NT, int KF);
