// Reduced-precision codec path (BASELINE.json config #5, second half): the Mimi decoder transformer
// GEMMs and the SEANet decoder convolutions with bf16 weights AND bf16 activations, fp32 accumulation on
// v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate), fp32 epilogue math.  The reference has no counterpart (its only
// reduced-precision mode is CPU dynamic int8 of the FlowLM, quantization.py:60-128): parity is UNPINNED; the path is
// judged by SNR against this build's fp32 path (tests/test_gpu_bf16.py) and is never the headline number.
//
// Layout "FMH" (the FM layout of ptts_kernels.h for 8 bf16 per lane): activation X[M][K], K % 32 == 0, is stored
// so that lane l = (g = l >> 4, m = l & 15) of a wave finds X[16 mt + m][32 kb + 8 g + 0..7] at 16-byte slot
// ((mt * KB + kb) * 64 + l), KB = K / 32: one global_load_dwordx4 per wave = one B operand of the 16x16x32 MFMA.
// Weights are packed [nt][tap * CB + cb][lane][8] with n for m (the A operand).  The 16x16 fp32 accumulator holds,
// in lane (g', m), columns n = 16 nt + 4 g' + r (r = 0..3) of row m: as the NEXT GEMM's operand these are elements
// 4 (g' & 1) + r of slot (g = 2 (nt & 1) + (g' >> 1), m) of fragment kb' = nt / 2, so a producer stores its tile with
// one 8-byte store per lane and a wave instruction still covers 512 contiguous bytes.
#pragma once
#include "ptts_kernels.h"

// (bf16 vector types, to_bf16x4 / from_bf16x4 and fmh_off live in ptts_kernels.h: the attention kernels and the codec
// prologue write FMH outputs too)

// load-time packing (same value(n, c, tap) convention as pack_weight_kernel): dst[nt][tap*CB+cb][lane][j8]
static __global__ void pack_weight_h_kernel(const float *src, __bf16 *dst, int N, int C, int ntaps, int mode, int cout, int stride,
                                     int KBt, long total, const float *colscale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j8 = i & 7, lane = (i >> 3) & 63;
  const long f = i >> 9;
  const int kbt = f % KBt, nt = f / KBt;
  const int CB = C / 32, tap = kbt / CB, cb = kbt - tap * CB;
  const int n = 16 * nt + (lane & 15), c = 32 * cb + 8 * (lane >> 4) + j8;
  float v = 0.f;
  if (n < N) {
    if (mode == 0) {
      v = src[((size_t)n * C + c) * ntaps + tap];
      if (colscale) v *= colscale[c];
    } else {
      const int j = n / cout, nn = n - j * cout, kidx = tap == 1 ? j : j + stride;
      v = src[((size_t)c * cout + nn) * (2 * stride) + kidx];
    }
  }
  dst[i] = (__bf16)v;
}
// LayerNorm-fold vector s[n] = sum_k W'[n][k] from the ROUNDED packed weights (the fold y = rstd (W'x - mean s) + c
// cancels exactly only if s matches what the MFMA multiplies with); one wave per output row, ntaps == 1
static __global__ void fold_s_h_kernel(const __bf16 *Wp, float *s_out, int N, int KB) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const __bf16 *base = Wp + ((size_t)(n >> 4) * KB * 64 + (n & 15)) * 8;
  float s = 0.f;
  for (int q = lane; q < KB * 4; q += 64) {  // (kb, g) slots of row n
    const bf16x8 v = *(const bf16x8 *)(base + ((size_t)(q >> 2) * 64 + (q & 3) * 16) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[j];
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) s_out[n] = s;
}

// Epilogue of one 16x16 tile (fp32 math, bf16 stores); the bf16 twin of gemm_epilogue.  Pointers of GemmArgs that
// address activations (X, Y, Yraw, R) are reinterpreted as __bf16 with strides counted in bf16 elements.
__device__ __forceinline__ void gemm_h_epilogue(const GemmArgs &a, f32x4 acc, int nt, int mt, int lane, int par) {
  const int ml = lane & 15, g = lane >> 4;
  const int m = 16 * mt + ml;
  const int n0 = 16 * nt + 4 * g;
  if (a.bias) acc += *(const f32x4 *)(a.bias + n0);
  switch (a.epi) {
    case EPI_STORE: {
      const size_t o = fmh_off(m, n0, a.YF);
      if (a.Yraw) *(bf16x4 *)((__bf16 *)a.Yraw + par * a.Yrawdstride + o) = to_bf16x4(acc);
      // yf8: the consumer is an fp8 convolution (PTTS_CODEC_FP8): same element order, one byte per element
      if (a.yf8) *(unsigned *)((uint8_t *)a.Y + par * a.Ydstride + o) = to_f8x4(act4(acc, a.act) * a.yinv);
      else *(bf16x4 *)((__bf16 *)a.Y + par * a.Ydstride + o) = to_bf16x4(act4(acc, a.act));
    } break;
    case EPI_RES: {
      const f32x4 rv = from_bf16x4(*(const bf16x4 *)((const __bf16 *)a.R + par * a.Rdstride + fmh_off(m, n0, a.RF)));
      if (a.ls) acc *= *(const f32x4 *)(a.ls + n0);
      *(bf16x4 *)((__bf16 *)a.Y + par * a.Ydstride + fmh_off(m, n0, a.YF)) = to_bf16x4(act4(rv + acc, a.act));
    } break;
    case EPI_QKV: {
      // identical to the fp32 path: q and the KV cache stay fp32 (the attention kernels are shared)
      if (m >= a.M) break;
      const int D = a.H * 64;
      const int which = n0 / D;
      const int hn = n0 - which * D;
      const int h = hn >> 6, d = hn & 63;
      const int b = m / a.Tq, t = m - b * a.Tq;
      const int pos = a.offset[b] + t;
      if (which < 2) {
        const f32x4 cs = *(const f32x4 *)(a.rope + ((size_t)m * 32 + (d >> 1)) * 2);
        f32x4 o;
        o.x = acc.x * cs.x - acc.y * cs.y;
        o.y = acc.x * cs.y + acc.y * cs.x;
        o.z = acc.z * cs.z - acc.w * cs.w;
        o.w = acc.z * cs.w + acc.w * cs.z;
        acc = o;
      }
      const size_t bh = (size_t)b * a.H + h;
      if (which == 0) {
        *(f32x4 *)(a.Q + (((bh * a.QB + (t >> 4)) * 4 + (d >> 4)) * 64 + 16 * g + (t & 15)) * 4) = acc;
      } else {
        const int slot = a.ring ? (pos % a.ring) : pos;
        *(f32x4 *)((which == 1 ? a.Kc : a.Vc) + (bh * a.cap + slot) * 64 + d) = acc;
      }
    } break;
    case EPI_CONVTR: {
      const int j = n0 / a.cout;
      const int n = n0 - j * a.cout;
      const size_t o = fmh_off((size_t)m * a.stride + j, n, a.YF);
      if (a.Yraw) *(bf16x4 *)((__bf16 *)a.Yraw + par * a.Yrawdstride + o) = to_bf16x4(acc);
      *(bf16x4 *)((__bf16 *)a.Y + par * a.Ydstride + o) = to_bf16x4(act4(acc, a.act));
    } break;
    default: break;
  }
}

// Implicit GEMM on bf16 operands, operands straight to VGPRs, register double-buffered k-loop.  A workgroup of
// WN x WM waves owns (WN * TN) x (WM * TM) tiles; a.CF = input channels / 32, a.KF = ntaps * a.CF, a.XF / YF / RF
// are 32-wide block counts.  Streaming causal conv only (halo_mode 0: rows before the sequence start come from the
// previous-frame buffer), stride-1 input.
template <int TN, int TM, int WN, int WM, int PRE>
__global__ __launch_bounds__(64 * WN * WM) void gemm_h_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave % WN, wm = wave / WN;
  int bx, by;
  tile_of_block(a.swz, bx, by);
  const int nt0 = (bx * WN + wn) * TN, mt0 = (by * WM + wm) * TM;
  const int par = a.par ? (*a.par & 1) : 0;
  const __bf16 *Xc = (const __bf16 *)a.X + par * a.Xdstride;
  const __bf16 *Xp = (const __bf16 *)a.X + (par ^ 1) * a.Xdstride;
  const __bf16 *wb[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) wb[i] = (const __bf16 *)a.W + ((size_t)min(nt0 + i, a.NT - 1) * a.KF * 64 + lane) * 8;
  int tin[TM], bT[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int row = 16 * min(mt0 + j, a.MT - 1) + (lane & 15);
    const int t = a.ntaps > 1 ? row % a.T : 0;
    tin[j] = t;
    bT[j] = row - t;
  }
  const __bf16 *xrow[TM];
  auto row_base = [&](int j, int tp) {
    const int ts = tin[j] + tp - a.halo;
    const __bf16 *src = Xc;
    long rr = (long)bT[j] + ts;
    if (ts < 0) { src = Xp; rr += a.T; }
    xrow[j] = src + (((size_t)(rr >> 4) * a.XF) * 64 + (lane & 48) + (rr & 15)) * 8;
  };
  f32x4 acc[TN][TM];
  float sx[TM], sxx[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    sx[j] = sxx[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int tap = 0, cb = 0;
#pragma unroll
  for (int j = 0; j < TM; ++j) row_base(j, 0);
  auto load = [&](int kb, bf16x8 *w, bf16x8 *x) {  // fragment kb = (tap, cb), advanced sequentially
#pragma unroll
    for (int i = 0; i < TN; ++i) w[i] = *(const bf16x8 *)(wb[i] + (size_t)kb * 512);
#pragma unroll
    for (int j = 0; j < TM; ++j) x[j] = *(const bf16x8 *)(xrow[j] + (size_t)cb * 512);
    if (++cb == a.CF) {
      cb = 0;
      ++tap;
#pragma unroll
      for (int j = 0; j < TM; ++j) row_base(j, min(tap, a.ntaps - 1));
    }
  };
  auto compute = [&](const bf16x8 *w, const bf16x8 *x) {
    if constexpr (PRE == PRE_LNFOLD) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float v = (float)x[j][q];
          sx[j] += v;
          sxx[j] += v * v;
        }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], x[j], acc[i][j], 0, 0, 0);
  };
  bf16x8 wA[TN], xA[TM], wB[TN], xB[TM];
  load(0, wA, xA);
  int kb = 0;
  for (; kb + 2 < a.KF; kb += 2) {
    load(kb + 1, wB, xB);
    compute(wA, xA);
    load(kb + 2, wA, xA);
    compute(wB, xB);
  }
  if (kb + 1 < a.KF) {
    load(kb + 1, wB, xB);
    compute(wA, xA);
    compute(wB, xB);
  } else {
    compute(wA, xA);
  }
  float mu[TM], rs[TM];
  if constexpr (PRE == PRE_LNFOLD) {
    const float invK = 1.0f / (float)(a.KF * 32);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      sx[j] += __shfl_xor(sx[j], 16); sx[j] += __shfl_xor(sx[j], 32);
      sxx[j] += __shfl_xor(sxx[j], 16); sxx[j] += __shfl_xor(sxx[j], 32);
      mu[j] = sx[j] * invK;
      rs[j] = 1.0f / sqrtf(fmaxf(sxx[j] * invK - mu[j] * mu[j], 0.f) + a.ln_eps);
    }
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int nt = nt0 + i, mt = mt0 + j;
      if (nt >= a.NT || mt >= a.MT) continue;
      f32x4 v = acc[i][j];
      if constexpr (PRE == PRE_LNFOLD) {
        const int n0 = 16 * nt + 4 * (lane >> 4);
        v = (v - *(const f32x4 *)(a.ln_s + n0) * mu[j]) * rs[j] + *(const f32x4 *)(a.ln_c + n0);
      }
      gemm_h_epilogue(a, v, nt, mt, lane, par);
    }
}

// SEANet's last conv (n_filters -> 1 sample) on a bf16 input: thread = one output sample, weights plain fp32
// [C][ntaps] (the checkpoint tensor), input FMH double-buffered by frame parity
static __global__ __launch_bounds__(256) void pcm_conv_h_kernel(GemmArgs a, const float *wplain, const float *bplain) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.M) return;
  const int par = a.par ? (*a.par & 1) : 0;
  const __bf16 *Xc = (const __bf16 *)a.X + par * a.Xdstride;
  const __bf16 *Xp = (const __bf16 *)a.X + (par ^ 1) * a.Xdstride;
  const int t = (int)(row % a.T);
  const long bT = row - t;
  const int C = a.CF * 32;
  float acc = bplain ? bplain[0] : 0.f;
  for (int tap = 0; tap < a.ntaps; ++tap) {
    const int ts = t + tap - a.halo;
    const __bf16 *src = Xc;
    long rr = bT + ts;
    if (ts < 0) { src = Xp; rr += a.T; }
    for (int c = 0; c < C; c += 8) {
      const bf16x8 v = *(const bf16x8 *)(src + fmh_off((size_t)rr, c, a.XF));
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += (float)v[q] * wplain[(size_t)(c + q) * a.ntaps + tap];
    }
  }
  a.pcm[row] = acc;
  if (a.pcm_i16) a.pcm_i16[row] = (int16_t)(fminf(fmaxf(acc, -1.0f), 1.0f) * 32767.0f);
}
