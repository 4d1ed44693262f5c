// SEANet decoder convolutions on the fp8 MFMA (PTTS_CODEC_FP8; BASELINE.json configs[4] "fp8 MFMA codec convs").
// Reference modules: SEANetDecoder seanet.py:141-180 over StreamingConv1d / StreamingConvTranspose1d conv.py:93-163 and
// SEANetResnetBlock seanet.py:33-41.  The reference never quantises Mimi (docs/quantization.md:67-76): no counterpart,
// parity UNPINNED; judged by SNR against the fp32 codec and frame-count equality (tests/test_gpu_fp8.py).
//
// Arithmetic: OCP e4m3 weights with one fp32 scale per output channel (amax / 448), OCP e4m3 activations with one STATIC
// fp32 scale per tensor (fixed at engine build from a calibration run of the bf16 codec: 2 x amax / 448, saturating), fp32
// accumulation on v_mfma_f32_16x16x32_fp8_fp8, fp32 epilogue (scale, bias, ELU, residual).  The residual block's skip
// input and the input of the last conv stay bf16; the Mimi transformer runs as under PTTS_CODEC_BF16.
//
// Layout "FM8" = the FMH layout of ptts_bf16.h with one byte per element: lane (g = l >> 4, m = l & 15) of a wave finds
// X[16 mt + m][32 kb + 8 g + 0..7] in the 8 bytes at ((mt * KB + kb) * 64 + l) * 8: one global_load_dwordx2 per wave =
// one B operand of the 16x16x32 MFMA; a producer stores the 4 columns a lane holds of a 16x16 tile as ONE dword at the
// byte offset fmh_off(row, n0, KB).  Weights [nt][tap * CB + cb][lane][8].
#include "ptts_ext.h"

static inline int cdiv8(long a, long b) { return (int)((a + b - 1) / b); }

// ---- load-time packing: one workgroup per n-tile --------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_f8_kernel(const float *src, uint8_t *dst, float *wscale, int N, int C, int ntaps,
                                                             int mode, int cout, int stride, int KBt) {
  __shared__ int smax[16];
  const int nt = blockIdx.x, tid = threadIdx.x;
  if (tid < 16) smax[tid] = 0;
  __syncthreads();
  const int CB = C / 32;
  auto value = [&](int n, int c, int tap) -> float {
    if (n >= N) return 0.f;
    if (mode == 0) return src[((size_t)n * C + c) * ntaps + tap];
    const int j = n / cout, nn = n - j * cout, kidx = tap == 1 ? j : j + stride;
    return src[((size_t)c * cout + nn) * (2 * stride) + kidx];
  };
  // quads: (kbt, lane, half) -> 4 consecutive k of one row
  const int nquads = KBt * 64 * 2;
  for (int q = tid; q < nquads; q += 256) {
    const int half = q & 1, lane = (q >> 1) & 63, kbt = q >> 7;
    const int tap = kbt / CB, cb = kbt - tap * CB;
    const int n = 16 * nt + (lane & 15), c0 = 32 * cb + 8 * (lane >> 4) + 4 * half;
    float m = 0.f;
    for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(value(n, c0 + j, tap)));
    atomicMax(&smax[lane & 15], __float_as_int(m));
  }
  __syncthreads();
  if (tid < 16) {
    const float mx = __int_as_float(smax[tid]);
    wscale[16 * nt + tid] = mx > 0.f ? mx / 448.0f : 1.0f;
  }
  __syncthreads();
  for (int q = tid; q < nquads; q += 256) {
    const int half = q & 1, lane = (q >> 1) & 63, kbt = q >> 7;
    const int tap = kbt / CB, cb = kbt - tap * CB;
    const int n = 16 * nt + (lane & 15), c0 = 32 * cb + 8 * (lane >> 4) + 4 * half;
    const float inv = 1.0f / wscale[16 * nt + (lane & 15)];
    const f32x4 v = {value(n, c0, tap) * inv, value(n, c0 + 1, tap) * inv, value(n, c0 + 2, tap) * inv, value(n, c0 + 3, tap) * inv};
    *(unsigned *)(dst + (((size_t)nt * KBt + kbt) * 64 + lane) * 8 + 4 * half) = to_f8x4(v);
  }
}

void pack_weight_f8(hipStream_t st, const float *src, void *dst, float *wscale, int N, int C, int ntaps, int mode, int cout, int stride) {
  const int NT = cdiv8(N, 16), KBt = (C / 32) * ntaps;
  pack_weight_f8_kernel<<<NT, 256, 0, st>>>(src, (uint8_t *)dst, wscale, N, C, ntaps, mode, cout, stride, KBt);
}

__global__ void amax_bf16_kernel(const bf16x8 *x, long n8, float *out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 v = x[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = fabsf((float)v[j]);
      m = f == f ? fmaxf(m, f) : m;
    }
  }
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax((int *)out, __float_as_int(m));
}
void amax_bf16(hipStream_t st, const void *x, long n, float *out) {
  amax_bf16_kernel<<<256, 256, 0, st>>>((const bf16x8 *)x, n / 8, out);
}

// ---- epilogue of one 16x16 tile ---------------------------------------------------------------------------------------
__device__ __forceinline__ void f8_store(const GemmArgs &a, int par, size_t o, f32x4 v) {
  if (a.yf8) *(unsigned *)((uint8_t *)a.Y + par * a.Ydstride + o) = to_f8x4(v * a.yinv);
  else *(bf16x4 *)((__bf16 *)a.Y + par * a.Ydstride + o) = to_bf16x4(v);
}
__device__ __forceinline__ void gemm_f8_epilogue(const GemmArgs &a, f32x4 acc, int nt, int mt, int lane, int par) {
  const int ml = lane & 15, g = lane >> 4;
  const int m = 16 * mt + ml;
  const int n0 = 16 * nt + 4 * g;
  acc = acc * (*(const f32x4 *)(a.wscale + n0) * a.xs);
  if (a.bias) acc += *(const f32x4 *)(a.bias + n0);
  if (a.epi == EPI_STORE) {
    const size_t o = fmh_off(m, n0, a.YF);
    if (a.Yraw) *(bf16x4 *)((__bf16 *)a.Yraw + par * a.Yrawdstride + o) = to_bf16x4(acc);
    f8_store(a, par, o, act4(acc, a.act));
  } else if (a.epi == EPI_RES) {
    const f32x4 rv = from_bf16x4(*(const bf16x4 *)((const __bf16 *)a.R + par * a.Rdstride + fmh_off(m, n0, a.RF)));
    f8_store(a, par, fmh_off(m, n0, a.YF), act4(rv + acc, a.act));
  } else if (a.epi == EPI_CONVTR) {
    const int j = n0 / a.cout;
    const int n = n0 - j * a.cout;
    const size_t o = fmh_off((size_t)m * a.stride + j, n, a.YF);
    if (a.Yraw) *(bf16x4 *)((__bf16 *)a.Yraw + par * a.Yrawdstride + o) = to_bf16x4(acc);
    f8_store(a, par, o, act4(acc, a.act));
  }
}

// Implicit GEMM on e4m3 operands, operands straight to VGPRs (8 bytes per lane and fragment), register double-buffered
// k-loop; same tiling, conv addressing and frame-parity halo rule as gemm_h_kernel (streaming causal conv, stride-1 input)
template <int TN, int TM, int WN, int WM>
__global__ __launch_bounds__(64 * WN * WM) void gemm_f8_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave % WN, wm = wave / WN;
  int bx, by;
  tile_of_block(a.swz, bx, by);
  const int nt0 = (bx * WN + wn) * TN, mt0 = (by * WM + wm) * TM;
  const int par = a.par ? (*a.par & 1) : 0;
  const uint8_t *Xc = (const uint8_t *)a.X + par * a.Xdstride;
  const uint8_t *Xp = (const uint8_t *)a.X + (par ^ 1) * a.Xdstride;
  const uint8_t *wb[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) wb[i] = (const uint8_t *)a.W + ((size_t)min(nt0 + i, a.NT - 1) * a.KF * 64 + lane) * 8;
  int tin[TM], bT[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int row = 16 * min(mt0 + j, a.MT - 1) + (lane & 15);
    const int t = a.ntaps > 1 ? row % a.T : 0;
    tin[j] = t;
    bT[j] = row - t;
  }
  const uint8_t *xrow[TM];
  auto row_base = [&](int j, int tp) {
    const int ts = tin[j] + tp - a.halo;
    const uint8_t *src = Xc;
    long rr = (long)bT[j] + ts;
    if (ts < 0) { src = Xp; rr += a.T; }
    xrow[j] = src + (((size_t)(rr >> 4) * a.XF) * 64 + (lane & 48) + (rr & 15)) * 8;
  };
  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int tap = 0, cb = 0;
#pragma unroll
  for (int j = 0; j < TM; ++j) row_base(j, 0);
  auto load = [&](int kb, long *w, long *x) {  // fragment kb = (tap, cb), advanced sequentially
#pragma unroll
    for (int i = 0; i < TN; ++i) w[i] = *(const long *)(wb[i] + (size_t)kb * 512);
#pragma unroll
    for (int j = 0; j < TM; ++j) x[j] = *(const long *)(xrow[j] + (size_t)cb * 512);
    if (++cb == a.CF) {
      cb = 0;
      ++tap;
#pragma unroll
      for (int j = 0; j < TM; ++j) row_base(j, min(tap, a.ntaps - 1));
    }
  };
  auto compute = [&](const long *w, const long *x) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w[i], x[j], acc[i][j], 0, 0, 0);
  };
  long wA[TN], xA[TM], wB[TN], xB[TM];
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
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int nt = nt0 + i, mt = mt0 + j;
      if (nt >= a.NT || mt >= a.MT) continue;
      gemm_f8_epilogue(a, acc[i][j], nt, mt, lane, par);
    }
}

template <int TN, int TM, int WN, int WM>
static void launch_f8_cfg(hipStream_t st, const GemmArgs &a, unsigned dyn) {
  const dim3 grid(cdiv8(a.NT, TN * WN), cdiv8(a.MT, TM * WM)), block(64 * WN * WM);
  gemm_f8_kernel<TN, TM, WN, WM><<<grid, block, dyn, st>>>(a);
}
// tile choice as for the bf16 codec (launch_gemm_h in ptts.hip): the largest workgroup tile that still yields >= ~2
// workgroups per CU; these kernels are bandwidth / launch bound (the fp8 MFMA runs at the bf16 rate = 16x fp32)
void launch_gemm_f8(hipStream_t st, const GemmArgs &a, unsigned dyn) {
  static const int tiles[4][4] = {{2, 4, 2, 2}, {2, 2, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2}};
  int pick = 3;
  for (int i = 0; i < 4; ++i) {
    const int *t = tiles[i];
    if (t[0] * t[2] > 2 * a.NT && i < 3) continue;  // mostly padding
    if ((long)cdiv8(a.NT, t[0] * t[2]) * cdiv8(a.MT, t[1] * t[3]) >= 512 || i == 3) { pick = i; break; }
  }
  switch (pick) {
    case 0: launch_f8_cfg<2, 4, 2, 2>(st, a, dyn); break;
    case 1: launch_f8_cfg<2, 2, 2, 2>(st, a, dyn); break;
    case 2: launch_f8_cfg<1, 2, 2, 2>(st, a, dyn); break;
    default: launch_f8_cfg<1, 1, 2, 2>(st, a, dyn); break;
  }
}
