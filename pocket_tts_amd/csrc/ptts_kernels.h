// Device kernels of the Pocket-TTS decode hot path for gfx950 (MI355X, CDNA4).
//
// Data layout ("FM" = MFMA-fragment-major).  Every activation matrix X[M][K] (rows = batch rows or
// (sequence, time) positions, K = channels, both padded to multiples of 16) is stored so that the
// 1 KiB a wave loads with one `global_load_dwordx4` is exactly one operand fragment of
// v_mfma_f32_16x16x4_f32:
//     element (m, k)  ->  float index  (((m/16)*F + k/16)*64 + 16*((k%16)/4) + m%16)*4 + k%4,   F = K/16
// i.e. lane l of the wave holds X[16*mt + (l&15)][16*kf + 4*(l>>4) + j], j = 0..3, and MFMA step j
// consumes element j of every lane.  Weights W[N][K] are packed once at load time into the same
// order with n in place of m, so the weight stream is a sequence of fully coalesced 1 KiB loads that go
// straight to VGPRs (no LDS round trip for an operand that is read exactly once).  The 16x16 fp32
// accumulator of an (n-tile, m-tile) pair is, lane for lane, fragment kf = n-tile of the next GEMM's
// input, so producers store it with one 1 KiB coalesced store and no shuffles.
//
// All arithmetic is fp32 (the reference runs fp32: english.yaml:9,27); the fp32-input MFMA is an exact
// k-ordered fmaf chain, so results differ from the reference only by summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// bf16 activations of the reduced-precision codec path (layout "FMH", see ptts_bf16.h)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 from_bf16x4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
// element offset (in bf16) of the 4 consecutive columns n0 .. n0 + 3 (n0 % 4 == 0) of row `row` in an FMH buffer with KB
// 32-column blocks per row tile: lane (g, m) of fragment (row tile, block) holds 8 consecutive columns in 16 bytes
__device__ __forceinline__ size_t fmh_off(size_t row, int n0, int KB) {
  return (((row >> 4) * KB + (n0 >> 5)) * 64 + (((n0 & 31) >> 3) * 16 + (row & 15))) * 8 + (n0 & 7);
}

// four floats -> four OCP e4m3 bytes (gfx950 v_cvt_pk_fp8_f32; saturating by an explicit clamp to +-448), one dword
__device__ __forceinline__ unsigned to_f8x4(f32x4 v) {
  v.x = fminf(fmaxf(v.x, -448.f), 448.f); v.y = fminf(fmaxf(v.y, -448.f), 448.f);
  v.z = fminf(fmaxf(v.z, -448.f), 448.f); v.w = fminf(fmaxf(v.w, -448.f), 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(v.x, v.y, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(v.z, v.w, r, true);
  return (unsigned)r;
}

// Ablation switches for tests/hip/bench_gemm.hip only (timing builds; results are wrong when set):
// 1 = no X loads, 2 = no W loads, 4 = no MFMA; gemm_lds_kernel: 8 = no DMA wait, 16 = no stage barrier, 32 = no DMA after the prologue;
// gemm_kernel: 64 = no PRE_LNMOD statistics pass; 128 = GEMM and attention kernels return at once.
// Always 0 in the library.
#ifndef PTTS_ABLATE
#define PTTS_ABLATE 0
#endif
// In-kernel time stamps for tests/hip/stamp_gemm.hip only (-DPTTS_STAMP): lane 0 of every wave writes the 100 MHz wall clock
// at five points of gemm_kernel into a buffer passed in GemmArgs::stamp.  Never defined in the library.
#ifdef PTTS_STAMP
#define STAMP(i)                                                                                                               \
  do {                                                                                                                         \
    if (a.stamp && (threadIdx.x & 63) == 0)                                                                                    \
      a.stamp[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define STAMP(i)
#endif

enum { PRE_NONE = 0, PRE_ELU = 1, PRE_ADDSILU = 2, PRE_LNFOLD = 3, PRE_LNMOD = 4 };
enum { EPI_STORE = 0, EPI_RES, EPI_GATE, EPI_QKV, EPI_HEAD, EPI_LATENT, EPI_CONVTR, EPI_PCM };
enum { ACT_NONE = 0, ACT_GELU, ACT_SILU, ACT_ELU };

#define NEG_BIG (-1e30f)

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float elu_f(float x) { return x > 0.0f ? x : expm1f(x); }
// ELU applied by a CONSUMER on its operand fragments (PRE_ELU: the producer stored the raw value once): evaluated once per
// tap and per wave that reads the fragment, so it has to be cheap: exp on the transcendental unit, absolute error <= 1.2e-7
// (the outputs are in (-1, 0]); expm1f's relative accuracy near 0 buys nothing against the 2e-4 parity tolerance
__device__ __forceinline__ float elu_fast_f(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
// ELU in producer epilogues: exp(x) - 1 has an ABSOLUTE error of ~1e-7 near 0, far inside the 2e-4 parity bound
__device__ __forceinline__ float elu_fast(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }

__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
  if (act == ACT_GELU) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
  else if (act == ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
  else if (act == ACT_ELU) { v.x = elu_fast(v.x); v.y = elu_fast(v.y); v.z = elu_fast(v.z); v.w = elu_fast(v.w); }
  return v;
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM: Y[m][n] = epilogue( sum_tap sum_c pre(X[row(m) + tap - halo][c]) * W[n][c][tap] )
// One kernel serves every Linear, every causal Conv1d (ntaps = kernel) and every ConvTranspose1d
// (kernel 2s, stride s == causal conv with 2 taps and s*Cout outputs) of the hot path.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
#ifdef PTTS_STAMP
  unsigned long long *stamp;  // tests/hip/stamp_gemm.hip only
#endif
  // weights, packed [NT][KF][64][4]; bias padded to NT*16
  const float *W, *bias;
  // int8 weight-only variant (Q8 kernels): biased bytes (q + 128) packed [NT][KF/4][64][16] so that one 16-byte load
  // per lane carries the lane's operands of FOUR consecutive k-fragments; y[n] = wscale[n] * sum_k q[n][k] x[k]
  const uint8_t *Wq;
  const float *wscale;
  // weight image format: 0 = fp32 (W), 1 = int8 (Wq, above), 2 = bf16 (Wq reinterpreted as __bf16, packed [NT][KF/2][64][8]:
  // lane (g, n) holds W[n][32 kb + 4 g + 0..3] and W[n][32 kb + 16 + 4 g + 0..3], i.e. the lane's operands of the two fp32
  // k-fragments 2 kb and 2 kb + 1, so the activation operand is the two fp32 fragments converted in registers, no shuffle),
  // 3 = split bf16 (error-compensated: Wq = the bf16 image of hi = bf16(w), W reinterpreted = the bf16 image of
  // lo = bf16(w - hi), both in the format-2 layout; the kernel splits the fp32 activation the same way in registers and
  // accumulates hi*hi + hi*lo + lo*hi in fp32 on v_mfma_f32_16x16x32_bf16: ~2^-16 relative per product instead of bf16's 2^-8)
  int wfmt;
  int swz;  // XCD-aware workgroup -> tile mapping (tile_of_block)
  int krot;  // Linear layers, K-split tiles: workgroup bx starts its K loop at chunk bx % nchunks, so the column blocks that
             // re-read the same activation rows do not request the same L2 lines at the same time
  // Row statistics hand-over (flow MLP): a producer (EPI_STORE / EPI_GATE) also writes, per output tile, each row's
  // (sum, sum of squares) over the tile's 16 columns to stat_out[((mt * NT + nt) * 16 + row) * 2]; the PRE_LNMOD
  // consumer then adds stat_nt partials per row instead of re-reading the whole row (fixed order, deterministic).
  float *stat_out;
  const float *stat_in;
  int stat_nt;
  const float *ln_g;  // Q8 + PRE_LNFOLD: the LayerNorm gain stays out of the quantised matrix and scales x on load
  int NT, KF, CF, ntaps;
  // input FM view; double-buffered by frame parity when Xdstride != 0
  const float *X;
  long Xdstride;
  int XF, MT, M, T;  // M = valid rows, T = rows per sequence (multiple of 16 when ntaps > 1)
  // conv addressing: input row of (output row t, tap) = t * xstride + tap - halo.  Rows before the start
  // of the sequence come from the previous-frame buffer (halo_mode 0, streaming decode), are zero (1,
  // whole-signal "constant" padding) or repeat the first row (2, "replicate"): reference conv.py:84-115
  int xstride, halo, halo_mode;
  const float *zeros;  // >= 16 B of zeros (halo_mode 1)
  const int *par;    // device frame counter (parity = *par & 1) or null
  const float *prevec;  // PRE_ADDSILU: per-k vector
  // PRE_LNFOLD: LayerNorm folded into this GEMM.  W is packed with the LN gain multiplied in (W' = W diag(g)),
  // ln_s[n] = sum_k W'[n][k], ln_c[n] = sum_k W[n][k] beta[k] (+ bias); the kernel accumulates sum(x), sum(x^2)
  // per row from the X fragments it loads anyway and finishes  y = rstd (W'x - mean ln_s) + ln_c.
  const float *ln_s, *ln_c;
  float ln_eps;
  // PRE_LNMOD: AdaLN-modulated LayerNorm applied to the operand on load (flow MLP, reference mlp.py:107-109,
  // 127-129): x' = (LN(x) * lnm_w + lnm_b) * (1 + scale[m][k]) + shift[m][k]; row statistics come from a
  // pre-pass over the (short) rows.  lnm_w == null: no affine (final layer).
  const float *lnm_w, *lnm_b, *mod_shift, *mod_scale;
  int modF;
  int epi, act;
  // output FM view
  float *Y;
  long Ydstride;
  int YF;
  float *Yraw;  // optional second output holding the value BEFORE `act` (same view shape as Y)
  long Yrawdstride;
  const float *R;  // residual
  long Rdstride;
  int RF;
  const float *G;  // gate (EPI_GATE)
  int GF;
  const float *ls;  // layer scale [N] or null
  // EPI_QKV
  float *Q, *Kc, *Vc;
  const int *offset;
  const float *rope;  // [M][32][2] (cos, sin) of pos * freq, built once per step by rope_table_kernel
  int H, Tq, QB, cap, ring;
  // EPI_HEAD
  float *eos_logit, *eos_logit2;  // state copy, caller's copy (device or pinned host)
  uint8_t *is_eos, *is_eos2;
  float eos_thr;
  int head_nt;  // n-tile holding the EOS row
  // end-of-step bookkeeping done by the EOS row's threads (one per sequence, so exactly once): offset[m] += 1 for active rows
  // (increment_steps, stateful_module.py:19-26) and, by row 0, the step counter.  Nothing after the head GEMM reads a position,
  // and the flow cluster / the next step's prologue only need the counter to have moved once per step.
  int *tail_offset, *tail_ctr;
  const int *tail_active;
  // EPI_LATENT
  float *lat;  // plain [M][ldim], updated in place
  float *lat_out1, *lat_out2;  // optional extra copies of the updated latent (state's next input, caller's buffer)
  float inv_steps;
  int ldim;
  // EPI_CONVTR
  int cout, stride;
  // gemm_lds_kernel<.., NT2 > 0>: the 1x1 conv that follows this conv inside a SEANet residual block, fused into
  // the same launch (Y / R / YF / RF / Ydstride then describe ITS output and skip input): packed [NT2][NT][64][4]
  const float *W2, *bias2;
  int act2;
  // fp8 codec convolutions (ptts_fp8.hip): xs = scale of the e4m3 activation operand (real = stored * xs); yf8 != 0: the
  // output Y is e4m3 in the FM8 layout, stored as value * yinv (yinv = 1 / scale of that tensor); else Y is bf16 FMH
  float xs, yinv;
  int yf8;
  // EPI_PCM
  float *pcm;
  int16_t *pcm_i16;  // optional 16-bit copy: (clamp(x, -1, 1) * 32767) truncated, as data/audio.py:79
  // fused residual block of the LAST stage + SEANet's last conv (k taps, n_filters -> 1 sample): the block's output tile
  // never leaves the CU.  pcm_w = the last conv's packed weights (NT = 1), pcm_part[row] = the taps whose rows lie in this
  // workgroup's 64 rows, pcm_carry[parity][tile][2] = what rows 0 and 1 of the NEXT tile still miss (pcm_fix_kernel adds them)
  const float *pcm_w;
  float *pcm_part, *pcm_carry;
  long pcm_cstride;  // floats between the two frame parities of pcm_carry
};

// Workgroup -> tile mapping.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (ids b and
// b + 8 share an L2).  With `swz` the linear id is permuted so that each XCD owns one CONTIGUOUS run of tiles in
// (row block, column block) order: the column blocks that re-read the same X rows then sit behind one L2 instead of
// up to 8.  Used when the activations are the larger operand (codec convs); decode GEMMs keep the identity
// mapping, under which the row blocks sharing a weight tile already share an XCD (grid.x is a multiple of 8).
__device__ __forceinline__ void tile_of_block(int swz, int &bx, int &by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  if (!swz) return;
  const int gx = gridDim.x, per = (gx * gridDim.y) >> 3;
  int bid = by * gx + bx;
  if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  by = bid / gx;
  bx = bid - by * gx;
}

template <int PRE>
__device__ __forceinline__ f32x4 pre4(f32x4 x, const float *prevec, int kf, int lane) {
  if (PRE == PRE_ELU) {
    x.x = elu_fast_f(x.x); x.y = elu_fast_f(x.y); x.z = elu_fast_f(x.z); x.w = elu_fast_f(x.w);
  } else if (PRE == PRE_ADDSILU) {
    f32x4 t = *(const f32x4 *)(prevec + 16 * kf + 4 * (lane >> 4));
    x.x = silu_f(x.x + t.x); x.y = silu_f(x.y + t.y); x.z = silu_f(x.z + t.z); x.w = silu_f(x.w + t.w);
  }
  return x;
}

// (sum, sum of squares) of each row over the 16 columns of one output tile: lanes l, l^16, l^32, l^48 hold the row
__device__ __forceinline__ void tile_row_stats(const GemmArgs &a, f32x4 v, int nt, int mt, int lane) {
  float s1 = (v.x + v.y) + (v.z + v.w);
  float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
  if (lane < 16) {
    float *p = a.stat_out + (((size_t)mt * a.NT + nt) * 16 + lane) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

// Epilogue operands a wave can fetch at kernel START for the tiles it will finish (K-split path): bias, residual
// and gate tiles, LN-fold vectors, int8 scales.  Loaded after the reduction they would add one dependent L2 round
// trip (~0.6 us) to every decode GEMM.
struct EpiPre {
  f32x4 bias, r, g, ln_s, ln_c, scale;
};

__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x4 acc, int nt, int mt, int lane, int par,
                                              const EpiPre *pf = nullptr) {
  const int ml = lane & 15, g = lane >> 4;
  const int m = 16 * mt + ml;
  const int n0 = 16 * nt + 4 * g;
  if (a.bias) {
    f32x4 b = pf ? pf->bias : *(const f32x4 *)(a.bias + n0);
    acc += b;
  }
  switch (a.epi) {
    case EPI_STORE: {
      if (a.Yraw) *(f32x4 *)(a.Yraw + par * a.Yrawdstride + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = acc;
      acc = act4(acc, a.act);
      float *y = a.Y + par * a.Ydstride;
      *(f32x4 *)(y + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = acc;
      if (a.stat_out) tile_row_stats(a, acc, nt, mt, lane);
    } break;
    case EPI_RES: {
      const float *r = a.R + par * a.Rdstride;
      f32x4 rv = pf ? pf->r : *(const f32x4 *)(r + (((size_t)mt * a.RF + nt) * 64 + lane) * 4);
      if (a.ls) acc *= *(const f32x4 *)(a.ls + n0);
      float *y = a.Y + par * a.Ydstride;
      *(f32x4 *)(y + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = act4(rv + acc, a.act);
    } break;
    case EPI_GATE: {
      f32x4 rv = pf ? pf->r : *(const f32x4 *)(a.R + (((size_t)mt * a.RF + nt) * 64 + lane) * 4);
      f32x4 gv = pf ? pf->g : *(const f32x4 *)(a.G + (((size_t)mt * a.GF + nt) * 64 + lane) * 4);
      const f32x4 out = rv + gv * acc;
      *(f32x4 *)(a.Y + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = out;
      if (a.stat_out) tile_row_stats(a, out, nt, mt, lane);
    } break;
    case EPI_QKV: {
      // packed in_proj rows: [q | k | v] x [H][64]  (reference transformer.py:138-143)
      if (m >= a.M) break;
      const int D = a.H * 64;
      const int which = n0 / D;
      const int hn = n0 - which * D;
      const int h = hn >> 6, d = hn & 63;
      const int b = m / a.Tq, t = m - b * a.Tq;
      const int pos = a.offset[b] + t;
      if (which < 2) {
        // interleaved-pair RoPE in fp32 (reference rope.py:28-58)
        const f32x4 cs = *(const f32x4 *)(a.rope + ((size_t)m * 32 + (d >> 1)) * 2);  // cos0 sin0 cos1 sin1
        const float c0 = cs.x, s0 = cs.y, c1 = cs.z, s1 = cs.w;
        f32x4 o;
        o.x = acc.x * c0 - acc.y * s0;
        o.y = acc.x * s0 + acc.y * c0;
        o.z = acc.z * c1 - acc.w * s1;
        o.w = acc.z * s1 + acc.w * c1;
        acc = o;
      }
      const size_t bh = (size_t)b * a.H + h;
      if (which == 0) {
        float *q = a.Q + (((bh * a.QB + (t >> 4)) * 4 + (d >> 4)) * 64 + 16 * g + (t & 15)) * 4;
        *(f32x4 *)q = acc;
      } else {
        const int slot = a.ring ? (pos % a.ring) : pos;
        float *c = (which == 1 ? a.Kc : a.Vc) + (bh * a.cap + slot) * 64 + d;
        *(f32x4 *)c = acc;
      }
    } break;
    case EPI_HEAD: {
      if (nt < a.head_nt) {
        // with a single LSD step the AdaLN input silu(t_emb + cond_embed(c)) (mlp.py:107,127,210) is formed here, once,
        // instead of on every operand load of the 640-tile modulation GEMM
        if (a.act == ACT_SILU) acc = act4(acc + *(const f32x4 *)(a.prevec + n0), ACT_SILU);
        *(f32x4 *)(a.Y + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = acc;
      } else if (g == 0 && m < a.M) {
        const uint8_t fl = acc.x > a.eos_thr ? 1 : 0;
        if (a.eos_logit) a.eos_logit[m] = acc.x;
        if (a.eos_logit2) a.eos_logit2[m] = acc.x;
        if (a.is_eos) a.is_eos[m] = fl;
        if (a.is_eos2) a.is_eos2[m] = fl;
        if (a.tail_offset && (!a.tail_active || a.tail_active[m])) a.tail_offset[m] += 1;
        if (a.tail_ctr && m == 0) *a.tail_ctr += 1;
      }
    } break;
    case EPI_LATENT: {
      // Euler update of lsd_decode: current += flow_dir / num_steps (reference flow_lm.py:39)
      f32x4 v;
      if (m < a.M) {
        float *p = a.lat + (size_t)m * a.ldim + n0;
        v = *(f32x4 *)p + acc * a.inv_steps;
        *(f32x4 *)p = v;
        if (a.lat_out1) *(f32x4 *)(a.lat_out1 + (size_t)m * a.ldim + n0) = v;
        if (a.lat_out2) *(f32x4 *)(a.lat_out2 + (size_t)m * a.ldim + n0) = v;
      } else {
        v = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      *(f32x4 *)(a.Y + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = v;
    } break;
    case EPI_CONVTR: {
      // out[(m*s + j)][n] with n' = j*cout + n (reference conv.py:151-163: overlap-add == 2-tap causal conv)
      const int j = n0 / a.cout;
      const int n = n0 - j * a.cout;
      const size_t mo = (size_t)m * a.stride + j;
      const size_t idx = (((mo >> 4) * a.YF + (n >> 4)) * 64 + 16 * g + (mo & 15)) * 4;
      if (a.Yraw) *(f32x4 *)(a.Yraw + par * a.Yrawdstride + idx) = acc;
      float *y = a.Y + par * a.Ydstride;
      *(f32x4 *)(y + idx) = act4(acc, a.act);
    } break;
    case EPI_PCM: {
      if (g == 0 && m < a.M) {
        a.pcm[m] = acc.x;
        if (a.pcm_i16) a.pcm_i16[m] = (int16_t)(fminf(fmaxf(acc.x, -1.0f), 1.0f) * 32767.0f);
      }
    } break;
  }
}

// TN x TM 16x16 tiles per wave; WK waves split K (LDS-reduced), WN x WM waves tile N x M.
template <int TN, int TM, int WK, int WN, int WM, int PRE, int WF = 0>
__global__ __launch_bounds__(64 * WK * WN * WM) void gemm_kernel(GemmArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;  // ablation 128 (timing only): empty kernels = launch + boundary cost
  STAMP(0);
  constexpr bool Q8 = WF == 1;   // int8 weights, fp32 activations, fp32 MFMA
  constexpr bool SPL = WF == 3;  // split bf16 (hi + lo images, three bf16 MFMAs per block): see GemmArgs::wfmt
  constexpr bool B16 = WF == 2 || SPL;  // bf16 weights, activations rounded to bf16 in registers, v_mfma_f32_16x16x32_bf16
  constexpr int NW = WK * WN * WM;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wk = wave % WK;
  const int wn = (wave / WK) % WN;
  const int wm = wave / (WK * WN);
  int bx, by;
  tile_of_block(a.swz, bx, by);
  const int nt0 = (bx * WN + wn) * TN;
  const int mt0 = (by * WM + wm) * TM;
  const int par = a.par ? (*a.par & 1) : 0;
  const float *Xc = a.X + par * a.Xdstride;
  const float *Xp = a.X + (par ^ 1) * a.Xdstride;
  const int k0 = (a.KF * wk) / WK, k1 = (a.KF * (wk + 1)) / WK;

  float sx[TM], sxx[TM];  // PRE_LNFOLD row statistics (this lane's row of each m-tile, its k-groups only)
#pragma unroll
  for (int j = 0; j < TM; ++j) sx[j] = sxx[j] = 0.f;
  // a lone tile per wave would be one dependent MFMA chain: give it two accumulators (summed at the end)
  constexpr int NACC = (TN * TM == 1) ? 2 : 1;
  f32x4 acc[TN][TM][NACC];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float *wb[TN];
  const uint8_t *wqb[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    int nt = nt0 + i < a.NT ? nt0 + i : a.NT - 1;
    wb[i] = SPL ? (const float *)((const uint8_t *)a.W + (size_t)nt * a.KF * 512 + lane * 16) : a.W + (size_t)nt * a.KF * 256 + lane * 4;
    wqb[i] = Q8 ? a.Wq + (size_t)nt * a.KF * 256 + lane * 16 : B16 ? a.Wq + (size_t)nt * a.KF * 512 + lane * 16 : nullptr;
  }
  int mtc[TM], tin[TM], bT[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    int mt = mt0 + j < a.MT ? mt0 + j : a.MT - 1;
    mtc[j] = mt;
    int row = 16 * mt + (lane & 15);
    int t = a.ntaps > 1 ? row % a.T : 0;
    tin[j] = t;
    bT[j] = row - t;
  }
  float lmu[TM], lrs[TM];  // PRE_LNMOD row statistics
  if constexpr (PRE == PRE_LNMOD && (PTTS_ABLATE & 64)) {  // ablation 64 (timing only): no statistics pass
#pragma unroll
    for (int j = 0; j < TM; ++j) { lmu[j] = 0.f; lrs[j] = 1.f; }
  } else if constexpr (PRE == PRE_LNMOD) {
    // pre-pass over the (short, L2-resident) rows: 8 independent loads in flight per batch; or, when the producer
    // handed over per-tile partials (stat_in), stat_nt / 4 eight-byte loads per lane
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float s1 = 0.f, s2 = 0.f;
      if (a.stat_in) {
        const float *sp = a.stat_in + (((size_t)mtc[j] * a.stat_nt) * 16 + (lane & 15)) * 2;
        for (int t = lane >> 4; t < a.stat_nt; t += 4) {  // lane group g adds tiles g, g+4, ..
          s1 += sp[(size_t)t * 32];
          s2 += sp[(size_t)t * 32 + 1];
        }
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const float invK = 1.0f / (float)(a.KF * 16);
        lmu[j] = s1 * invK;
        lrs[j] = 1.0f / sqrtf(fmaxf(s2 * invK - lmu[j] * lmu[j], 0.f) + a.ln_eps);
        continue;
      }
      const float *xr = Xc + ((size_t)mtc[j] * a.XF * 64 + lane) * 4;
      int kf = 0;
      for (; kf + 8 <= a.KF; kf += 8) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *(const f32x4 *)(xr + (size_t)(kf + q) * 256);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          s1 += (v[q].x + v[q].y) + (v[q].z + v[q].w);
          s2 += (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
        }
      }
      for (; kf < a.KF; ++kf) {
        const f32x4 v = *(const f32x4 *)(xr + (size_t)kf * 256);
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      const float invK = 1.0f / (float)(a.KF * 16);
      lmu[j] = s1 * invK;
      lrs[j] = 1.0f / sqrtf(fmaxf(s2 * invK - lmu[j] * lmu[j], 0.f) + a.ln_eps);
    }
  }
  // K-split path: tile e of this wave (tile index wk + e * WK in i-major order) is finished by this wave; fetch its
  // epilogue operands now (see EpiPre)
  constexpr int EPT = (WK > 1) ? (TN * TM + WK - 1) / WK : 1;
  EpiPre epf[EPT];
  if constexpr (WK > 1) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int tl = wk + e * WK;
      const int i = tl / TM, j = tl - i * TM;
      const int nt = min(nt0 + i, a.NT - 1), mt = min(mt0 + j, a.MT - 1);
      const int n0 = 16 * nt + 4 * (lane >> 4);
      if (tl < TN * TM) {
        if (a.bias) epf[e].bias = *(const f32x4 *)(a.bias + n0);
        if constexpr (Q8) epf[e].scale = *(const f32x4 *)(a.wscale + n0);
        if constexpr (PRE == PRE_LNFOLD) {
          epf[e].ln_s = *(const f32x4 *)(a.ln_s + n0);
          epf[e].ln_c = *(const f32x4 *)(a.ln_c + n0);
        }
        if (a.epi == EPI_RES) epf[e].r = *(const f32x4 *)(a.R + par * a.Rdstride + (((size_t)mt * a.RF + nt) * 64 + lane) * 4);
        if (a.epi == EPI_GATE) {
          epf[e].r = *(const f32x4 *)(a.R + (((size_t)mt * a.RF + nt) * 64 + lane) * 4);
          epf[e].g = *(const f32x4 *)(a.G + (((size_t)mt * a.GF + nt) * 64 + lane) * 4);
        }
      }
    }
  }
  int tap = 0, cf = k0;
  if (a.ntaps > 1) {
    tap = k0 / a.CF;
    cf = k0 - tap * a.CF;
  }
  const int halo = a.halo;
  // Incremental operand addressing (as in gemm_lds_kernel): fragment (row tile j, tap, cf) sits at
  // xrow[j] + cf * 1 KiB; the per-lane 64-bit row arithmetic is redone only when the tap changes.
  const float *xrow[TM];
  bool xz[TM];
  auto row_base = [&](int j, int tp) {
    if (a.ntaps == 1) {
      xrow[j] = Xc + (((size_t)mtc[j] * a.XF) * 64 + lane) * 4;
      xz[j] = false;
      return;
    }
    const int ts = tin[j] * a.xstride + tp - halo;
    const float *src = Xc;
    long rr = (long)bT[j] * a.xstride + ts;
    if (ts < 0) {
      if (a.halo_mode == 0) { src = Xp; rr += (long)a.T * a.xstride; }
      else if (a.halo_mode == 2) rr = (long)bT[j] * a.xstride;
    }
    xrow[j] = src + (((size_t)(rr >> 4) * a.XF) * 64 + (lane & 48) + (rr & 15)) * 4;
    xz[j] = ts < 0 && a.halo_mode == 1;
  };
#pragma unroll
  for (int j = 0; j < TM; ++j) row_base(j, tap);
  // Software pipeline over chunks of U k-fragments: the 1 KiB operand loads of chunk c+1 are issued before the
  // MFMAs of chunk c, into a second register set, so the matrix pipe works while the next operands fly.
  // decode (K-split) tiles: as many fragments per round as the register file allows, so that a wave needs
  // few serialized HBM round trips for its cold weight stream
  // K-split (decode) tiles hold PTTS_KSPLIT_DIV times fewer fragments per round than the register file would allow.
  // Measured (tools/ab.sh lib, batch 64 pipelined): DIV 1 (253 VGPRs for the 2x2 tile) 0.935-0.939 ms per step, DIV 2 (173)
  // 0.905-0.910, DIV 4 (128) 0.920.  A FlowLM GEMM is ~1 wave per SIMD that mostly waits for its weights: at 253
  // registers it pins HALF of every SIMD's register file while resident and evicts the codec stream's waves; with half
  // the fragments in flight it is barely slower alone (batch 1: 0.358 -> 0.355 ms per step) and the codec keeps its occupancy.
  // The optimum is sharp and the same for every tile: the single-tile (1x1) configurations alone at 4 or at 1: 0.875 / 0.878
  // against 0.846 (PTTS_KSPLIT_DIV1, tools/ab.sh lib).
#ifndef PTTS_KSPLIT_DIV
#define PTTS_KSPLIT_DIV 2
#endif
  constexpr int U00 = (TN * TM == 1) ? 8 : (TN * TM == 2 && WK > 1) ? 8 : (TN * TM <= 4 && WK > 1) ? 4 : 2;
#ifndef PTTS_KSPLIT_DIV1
#define PTTS_KSPLIT_DIV1 PTTS_KSPLIT_DIV  // divisor for the single-tile (1x1) K-split configurations
#endif
  constexpr int KDIV = (TN * TM == 1) ? PTTS_KSPLIT_DIV1 : PTTS_KSPLIT_DIV;
  constexpr int U0 = (WK > 1) ? (U00 / KDIV >= 1 ? U00 / KDIV : 1) : U00;
  constexpr int U = (Q8 && U0 < 4) ? 4 : (B16 && U0 < 2) ? 2 : U0;  // int8 / bf16 weights arrive four / two k-fragments per load
  // (split bf16: a chunk of U fragments parks U / 2 hi blocks in slots 0 .. U/2 - 1 and U / 2 lo blocks behind them)
  auto load_chunk = [&](auto uc, int kf, f32x4 (*w)[TN], f32x4 (*x)[TM]) {
    constexpr int UU = decltype(uc)::value;
    if (a.ntaps == 1) cf = kf;  // a Linear's fragments are addressed by k alone (chunks may come in rotated order)
#pragma unroll
    for (int u = 0; u < UU; ++u) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        // weights are read exactly once by the K-split (decode) configuration: stream them non-temporally
        if constexpr (PTTS_ABLATE & 2) w[u][i] = (f32x4){1.f, 2.f, 3.f, (float)lane};
        else if constexpr (Q8) {
          // raw bytes of fragments kf+u .. kf+u+3, parked in slot u/4 until compute_chunk converts them
          if (u % 4 == 0) {
            const i32x4 raw = __builtin_nontemporal_load((const i32x4 *)(wqb[i] + (size_t)(kf + u) * 256));
            w[u / 4][i] = __builtin_bit_cast(f32x4, raw);
          }
        } else if constexpr (B16) {
          // eight bf16 of fragments kf+u, kf+u+1 (kf + u even), parked in slot u/2
          if (u % 2 == 0) {
            if constexpr (WK > 1) {
              w[u / 2][i] = __builtin_bit_cast(f32x4, __builtin_nontemporal_load((const i32x4 *)(wqb[i] + (size_t)((kf + u) >> 1) * 1024)));
              if constexpr (SPL)
                w[UU / 2 + u / 2][i] = __builtin_bit_cast(f32x4, __builtin_nontemporal_load((const i32x4 *)((const uint8_t *)wb[i] + (size_t)((kf + u) >> 1) * 1024)));
            } else {  // 2-D tilings re-read a weight tile from several row blocks: plain (cached) loads
              w[u / 2][i] = *(const f32x4 *)(wqb[i] + (size_t)((kf + u) >> 1) * 1024);
              if constexpr (SPL) w[UU / 2 + u / 2][i] = *(const f32x4 *)((const uint8_t *)wb[i] + (size_t)((kf + u) >> 1) * 1024);
            }
          }
        } else if constexpr (WK > 1) w[u][i] = __builtin_nontemporal_load((const f32x4 *)(wb[i] + (size_t)(kf + u) * 256));
        else w[u][i] = *(const f32x4 *)(wb[i] + (size_t)(kf + u) * 256);
      }
      if constexpr (PTTS_ABLATE & 1) {
#pragma unroll
        for (int j = 0; j < TM; ++j) x[u][j] = (f32x4){1.f, (float)lane, 3.f, (float)kf};
      } else {
        // ONE load site per fragment, behind a pointer select: loads inside divergent-looking branches make the
        // s_waitcnt insertion merge the paths conservatively and serialise the prefetch (see attn_kernel)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const float *ptr = xz[j] ? a.zeros : xrow[j] + (size_t)cf * 256;
          x[u][j] = *(const f32x4 *)ptr;
        }
        if (++cf == a.CF) {  // next tap: the rows move by one (a Linear has CF == KF: never taken before the end)
          cf = 0;
          ++tap;
#pragma unroll
          for (int j = 0; j < TM; ++j) row_base(j, tap);
        }
      }
    }
  };
  auto compute_chunk = [&](auto uc, int kf, f32x4 (*w)[TN], f32x4 (*x)[TM]) {
    constexpr int UU = decltype(uc)::value;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
#pragma unroll
      for (int j = 0; j < TM; ++j) x[u][j] = pre4<PRE>(x[u][j], a.prevec, kf + u, lane);
      if constexpr (PRE == PRE_LNMOD) {
        const int k = 16 * (kf + u) + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          f32x4 v = (x[u][j] - lmu[j]) * lrs[j];
          if (a.lnm_w) v = v * *(const f32x4 *)(a.lnm_w + k) + *(const f32x4 *)(a.lnm_b + k);
          const size_t mi = (((size_t)mtc[j] * a.modF + kf + u) * 64 + lane) * 4;
          x[u][j] = v * (1.0f + *(const f32x4 *)(a.mod_scale + mi)) + *(const f32x4 *)(a.mod_shift + mi);
        }
      }
      if constexpr (PRE == PRE_LNFOLD) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const f32x4 v = x[u][j];
          sx[j] += (v.x + v.y) + (v.z + v.w);
          sxx[j] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if constexpr (Q8) {
          const f32x4 g4 = *(const f32x4 *)(a.ln_g + 16 * (kf + u) + 4 * (lane >> 4));
#pragma unroll
          for (int j = 0; j < TM; ++j) x[u][j] *= g4;
        }
      }
      if constexpr (B16) {
        // one 16x16x32 MFMA per (tile pair, two k-fragments): issued at the odd fragment, when both halves of the
        // activation operand have gone through the operand pre-processing above
        if (u % 2 == 1) {
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const bf16x4 lo = to_bf16x4(x[u - 1][j]), hi = to_bf16x4(x[u][j]);
            const bf16x8 xb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            if constexpr (SPL) {
              // residual halves of the activation: x - bf16(x) is exact in fp32 and rounds to bf16 with 2^-17 relative error
              const bf16x4 rl = to_bf16x4(x[u - 1][j] - from_bf16x4(lo)), rh = to_bf16x4(x[u][j] - from_bf16x4(hi));
              const bf16x8 xr = __builtin_shufflevector(rl, rh, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
              for (int i = 0; i < TN; ++i) {
                const bf16x8 wh = __builtin_bit_cast(bf16x8, w[u / 2][i]), wl = __builtin_bit_cast(bf16x8, w[UU / 2 + u / 2][i]);
                // small terms first, the hi * hi term last
                acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xb, acc[i][j][0], 0, 0, 0);
                acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xr, acc[i][j][0], 0, 0, 0);
                acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xb, acc[i][j][0], 0, 0, 0);
              }
              continue;
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
              acc[i][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u / 2][i]), xb, acc[i][j][0], 0, 0, 0);
          }
        }
        continue;
      }
      f32x4 wv[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if constexpr (Q8) {
          const unsigned word = (unsigned)__builtin_bit_cast(i32x4, w[u / 4][i])[u % 4];
          wv[i] = (f32x4){(float)(word & 0xffu), (float)((word >> 8) & 0xffu), (float)((word >> 16) & 0xffu), (float)(word >> 24)} - 128.0f;
        } else {
          wv[i] = w[u][i];
        }
      }
      // k-step major: back-to-back MFMAs hit DIFFERENT accumulators, so none waits out the 40-cycle
      // dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (issue interval 32 cycles)
#pragma unroll
      for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if constexpr (PTTS_ABLATE & 4) acc[i][j][cidx % NACC][cidx] += wv[i][cidx] + x[u][j][cidx];
            else
              acc[i][j][cidx % NACC] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][cidx], x[u][j][cidx], acc[i][j][cidx % NACC], 0, 0, 0);
          }
    }
  };
  const std::integral_constant<int, U> cU{};
  const std::integral_constant<int, 1> c1{};
  const int nfull = (k1 - k0) / U;
  STAMP(1);
  // chunk c of this wave covers k-fragments kfc(c) .. kfc(c) + U - 1; rotated by the column block for Linear layers
  const int rot = (a.krot && a.ntaps == 1 && nfull > 1) ? (int)(bx % nfull) : 0;
  auto kfc = [&](int c) { int i = c + rot; if (i >= nfull) i -= nfull; return k0 + i * U; };
  if (nfull > 0) {
    // Two register sets: chunk c+1 is in flight while chunk c feeds the MFMAs.  The steady-state loop issues
    // its prefetches unconditionally (the tail is peeled), so the wait counts stay exact.  A decode wave that
    // owns several chunks of a long K (FFN2: 4) no longer pays one full HBM round trip per chunk.
    f32x4 wA[U][TN], xA[U][TM], wB[U][TN], xB[U][TM];
    load_chunk(cU, kfc(0), wA, xA);
    int c = 0;
    for (; c + 2 < nfull; c += 2) {
      load_chunk(cU, kfc(c + 1), wB, xB);
      compute_chunk(cU, kfc(c), wA, xA);
      load_chunk(cU, kfc(c + 2), wA, xA);
      compute_chunk(cU, kfc(c + 1), wB, xB);
    }
    if (c + 1 < nfull) {
      load_chunk(cU, kfc(c + 1), wB, xB);
      compute_chunk(cU, kfc(c), wA, xA);
      compute_chunk(cU, kfc(c + 1), wB, xB);
    } else {
      compute_chunk(cU, kfc(c), wA, xA);
    }
  }
  int kf = k0 + nfull * U;
  if constexpr (Q8) {
    const std::integral_constant<int, 4> c4{};  // host guarantees (k1 - k0) % 4 == 0
    for (; kf < k1; kf += 4) {
      f32x4 w4[4][TN], x4[4][TM];
      load_chunk(c4, kf, w4, x4);
      compute_chunk(c4, kf, w4, x4);
    }
  }
  if constexpr (B16) {
    const std::integral_constant<int, 2> c2{};  // host guarantees (k1 - k0) % 2 == 0
    for (; kf < k1; kf += 2) {
      f32x4 w2[2][TN], x2[2][TM];
      load_chunk(c2, kf, w2, x2);
      compute_chunk(c2, kf, w2, x2);
    }
  }
  for (; kf < k1; ++kf) {
    f32x4 w1[1][TN], x1[1][TM];
    load_chunk(c1, kf, w1, x1);
    compute_chunk(c1, kf, w1, x1);
  }

  STAMP(2);
  f32x4 accs[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      accs[i][j] = acc[i][j][0];
      if constexpr (NACC == 2) accs[i][j] += acc[i][j][1];
    }
  // PRE_LNFOLD: finish the row statistics.  Lanes l, l^16, l^32, l^48 hold the four k-groups of one row.
  float mu[TM], rs[TM];
  if constexpr (PRE == PRE_LNFOLD) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      sx[j] += __shfl_xor(sx[j], 16); sx[j] += __shfl_xor(sx[j], 32);
      sxx[j] += __shfl_xor(sxx[j], 16); sxx[j] += __shfl_xor(sxx[j], 32);
    }
  }
  auto ln_fix = [&](f32x4 v, int nt, int j, const EpiPre *pf) {
    if constexpr (Q8) v *= pf ? pf->scale : *(const f32x4 *)(a.wscale + 16 * nt + 4 * (lane >> 4));
    if constexpr (PRE == PRE_LNFOLD) {
      const int n0 = 16 * nt + 4 * (lane >> 4);
      const f32x4 s4 = pf ? pf->ln_s : *(const f32x4 *)(a.ln_s + n0), c4 = pf ? pf->ln_c : *(const f32x4 *)(a.ln_c + n0);
      return (v - s4 * mu[j]) * rs[j] + c4;
    } else {
      return v;
    }
  };
  if constexpr (WK > 1) {
    // every wave parks its partial tiles in LDS; tile i*TM+j is then summed (fixed order) and finished by
    // wave (i*TM+j) % WK, so the epilogues of a workgroup run on several SIMDs at once
    __shared__ f32x4 red[WK * WN * WM * TN * TM * 64];
    __shared__ float red_st[(PRE == PRE_LNFOLD) ? WK * WN * WM * TM * 32 : 1];
    const int grp = wave / WK;  // (wn, wm) group
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) red[(((wk * WN * WM + grp) * TN + i) * TM + j) * 64 + lane] = accs[i][j];
    if constexpr (PRE == PRE_LNFOLD) {
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          red_st[(((wk * WN * WM + grp) * TM + j) * 16 + lane) * 2 + 0] = sx[j];
          red_st[(((wk * WN * WM + grp) * TM + j) * 16 + lane) * 2 + 1] = sxx[j];
        }
      }
    }
    __syncthreads();
    STAMP(3);
    if constexpr (PRE == PRE_LNFOLD) {
      const float invK = 1.0f / (float)(a.KF * 16);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float t0 = 0.f, t1 = 0.f;
        for (int s2 = 0; s2 < WK; ++s2) {
          t0 += red_st[(((s2 * WN * WM + grp) * TM + j) * 16 + (lane & 15)) * 2 + 0];
          t1 += red_st[(((s2 * WN * WM + grp) * TM + j) * 16 + (lane & 15)) * 2 + 1];
        }
        mu[j] = t0 * invK;
        rs[j] = 1.0f / sqrtf(fmaxf(t1 * invK - mu[j] * mu[j], 0.f) + a.ln_eps);
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if ((i * TM + j) % WK != wk) continue;
        f32x4 sum = red[(((0 * WN * WM + grp) * TN + i) * TM + j) * 64 + lane];
        for (int s2 = 1; s2 < WK; ++s2) sum += red[(((s2 * WN * WM + grp) * TN + i) * TM + j) * 64 + lane];
        const EpiPre *pf = &epf[(i * TM + j) / WK];
        if (nt0 + i < a.NT && mt0 + j < a.MT) gemm_epilogue(a, ln_fix(sum, nt0 + i, j, pf), nt0 + i, mt0 + j, lane, par, pf);
      }
  } else {
    if constexpr (PRE == PRE_LNFOLD) {
      const float invK = 1.0f / (float)(a.KF * 16);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        mu[j] = sx[j] * invK;
        rs[j] = 1.0f / sqrtf(fmaxf(sxx[j] * invK - mu[j] * mu[j], 0.f) + a.ln_eps);
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        if (nt0 + i < a.NT && mt0 + j < a.MT) gemm_epilogue(a, ln_fix(accs[i][j], nt0 + i, j, nullptr), nt0 + i, mt0 + j, lane, par);
  }
  STAMP(4);
  (void)NW;
}

// ---------------------------------------------------------------------------------------------
// Single-output-channel causal conv (SEANet's last conv: n_filters -> 1 sample, reference seanet.py:168-172) on
// the vector ALU.  As a GEMM it would use one of 16 MFMA rows; here a wave owns 16 output rows, lane (row, g)
// dots its quarter of the channels for every tap and the four quarters meet by two xor-shuffles.  Memory bound:
// every input row is read once from HBM, the other taps re-read its lines through L1 (plain loads on purpose).  Same GemmArgs (taps, halo rule, frame-parity double buffer, packed weights with NT = 1).
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void pcm_conv_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = blockIdx.x * 4 + wave;
  if (mt >= a.MT) return;
  const int par = a.par ? (*a.par & 1) : 0;
  const float *Xc = a.X + par * a.Xdstride;
  const float *Xp = a.X + (par ^ 1) * a.Xdstride;
  const int row = 16 * mt + (lane & 15);
  const int t = a.ntaps > 1 ? row % a.T : 0;
  const int bT = row - t;
  // CF == 4 channel fragments, at most 4 taps (host-checked): all 32 loads of a lane are issued before the first
  // FMA, branch-free (a missing tap reads the zero line), so a wave pays one memory round trip
  f32x4 x[4][4], w[4][4];
#pragma unroll
  for (int tap = 0; tap < 4; ++tap) {
    const bool have = tap < a.ntaps;
    const int tp = have ? tap : a.ntaps - 1;
    const int ts = t * a.xstride + tp - a.halo;
    const float *src = Xc;
    long rr = (long)bT * a.xstride + ts;
    if (ts < 0) {
      if (a.halo_mode == 0) { src = Xp; rr += (long)a.T * a.xstride; }
      else if (a.halo_mode == 2) rr = (long)bT * a.xstride;
    }
    const float *xr = src + (((size_t)(rr >> 4) * a.XF) * 64 + (lane & 48) + (rr & 15)) * 4;
    const bool zero = !have || (ts < 0 && a.halo_mode == 1);
    // weights of output channel 0: packed element (n = 0, k) sits in lane 16 * ((k % 16) / 4) of fragment k / 16
    const float *wr = a.W + ((size_t)tp * 4 * 64 + (lane & 48)) * 4;
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) {
      x[tap][cf] = *(const f32x4 *)(zero ? a.zeros : xr + (size_t)cf * 256);  // plain: the taps re-read lines through L1
      w[tap][cf] = *(const f32x4 *)(wr + (size_t)cf * 256);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int tap = 0; tap < 4; ++tap)
#pragma unroll
    for (int cf = 0; cf < 4; ++cf)
      acc += (x[tap][cf].x * w[tap][cf].x + x[tap][cf].y * w[tap][cf].y) + (x[tap][cf].z * w[tap][cf].z + x[tap][cf].w * w[tap][cf].w);
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if (lane < 16 && row < a.M) {
    if (a.bias) acc += a.bias[0];
    a.pcm[row] = acc;
    if (a.pcm_i16) a.pcm_i16[row] = (int16_t)(fminf(fmaxf(acc, -1.0f), 1.0f) * 32767.0f);
  }
}

// Final PCM of the fused (residual block + last conv) path: row u of a 64-row tile adds what the PREVIOUS tile of the same
// sequence left for it (rows 0 and 1 only; the first tile of a frame takes the last tile of the previous frame from the
// other parity, zero on a sequence's first frame) and the bias, and writes fp32 (+ int16) PCM - straight into the caller's
// (pinned host) buffer like pcm_conv_kernel.
static __global__ void pcm_fix_kernel(const float *part, const float *carry, long cstride, const int *parp, const float *bias,
                                      float *pcm, int16_t *pcm_i16, int M, int tps) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  const int par = parp ? (*parp & 1) : 0;
  const int u = row & 63, tile = row >> 6;
  float v = part[row] + (bias ? bias[0] : 0.f);
  if (u < 2) {
    const bool first = tile % tps == 0;
    const float *c = carry + (first ? (par ^ 1) : par) * cstride + (size_t)(first ? tile + tps - 1 : tile - 1) * 2;
    v += c[u];
  }
  pcm[row] = v;
  if (pcm_i16) pcm_i16[row] = (int16_t)(fminf(fmaxf(v, -1.0f), 1.0f) * 32767.0f);
}

// ---------------------------------------------------------------------------------------------
// LDS-staged variant for the large-M GEMMs of the codec (rows = sequences x time).  A workgroup of 4 waves
// (2 x 2) owns BMT x BNT 16x16 tiles.  Per stage of KC k-fragments every operand fragment is copied ONCE per
// workgroup by `global_load_lds_dwordx4` (1 KiB, wave-linear = exactly the FM / packed fragment image, no VGPRs),
// then each wave reads its fragments with conflict-free ds_read_b128 (all reads of a stage issued before its
// MFMAs).  The DMA of stage s+1 is in flight while stage s feeds the MFMAs; one barrier per stage.
// ---------------------------------------------------------------------------------------------
#define GLDS16(gptr, lptr)                                                                      \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),      \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NT2 > 0 (fused SEANet residual block, reference seanet.py:12-49: x + conv1x1(elu(conv_k3(elu(x))))): the workgroup owns
// ALL BNT = NT column tiles of the k3 conv for its 16 * BMT rows, so h = act(conv + bias) never leaves the CU: the
// accumulators are written to LDS as they are (the MFMA C layout of a 16x16 tile IS the FM operand fragment of the
// next GEMM), and after one barrier every wave runs the 1x1 conv for its share of the NT2 output column tiles
// (K = 16 * BNT, weights straight from L2 into registers, fetched at kernel start) and finishes with the residual
// epilogue.  Saves the launch, the HBM write + read of h and the second kernel's ramp.
template <int BMT, int BNT, int KC, int PRE, int NS = 2, int NT2 = 0>
__global__ __launch_bounds__(256) void gemm_lds_kernel(GemmArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  static_assert(BMT % 4 == 0 && BNT % 2 == 0 && (BNT * KC) % 4 == 0, "tile shape");
  static_assert(NT2 == 0 || ((PRE == PRE_NONE || PRE == PRE_ELU) && BMT * BNT <= NS * (BMT + BNT) * KC && NT2 % 4 == 0), "fused tail");
  static_assert(NS >= 2 && NS <= 4, "stage count");
  constexpr int WMT = BMT / 2, WNT = BNT / 2;  // tiles per wave
  constexpr int NX = BMT * KC, NFRAG = (BMT + BNT) * KC;
  constexpr int XPW = BMT / 4;  // distinct m-tiles a loader wave touches (fragment f -> wave f % 4)
  constexpr int IPS = NFRAG / 4;  // DMA instructions per stage and wave
  // NS-deep ring: NS - 1 stages are in flight while one feeds the MFMAs.  Measured (tests/hip/bench_lds_ring.hip):
  // 3 or 4 stages are never faster than 2 on the codec shapes and cost occupancy, so 2 is the default.
  __shared__ f32x4 lds[NS][NFRAG][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by;
  tile_of_block(a.swz, bx, by);
  const int mt0 = by * BMT, nt0 = bx * BNT;
  const int par = a.par ? (*a.par & 1) : 0;
  const float *Xc = a.X + par * a.Xdstride;
  const float *Xp = a.X + (par ^ 1) * a.Xdstride;
  const int halo = a.halo;

  // loader bookkeeping: this wave copies X fragments of m-tiles {wave, wave+4, ..} and W fragments f % 4 == wave
  int l_mt[XPW], l_t[XPW], l_bT[XPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    int mt = mt0 + wave + 4 * q;
    mt = mt < a.MT ? mt : a.MT - 1;
    l_mt[q] = mt;
    int row = 16 * mt + (lane & 15);
    int t = a.ntaps > 1 ? row % a.T : 0;
    l_t[q] = t;
    l_bT[q] = row - t;
  }
  // Incremental addressing.  The DMA source of fragment (row tile, tap, cf) is  row_base(tap) + cf * 1 KiB: the
  // per-lane 64-bit row arithmetic (tile / halo / previous-frame selection) is redone only when the tap changes,
  // i.e. once per CF k-fragments, and the tap / cf position is a running scalar state instead of a division per
  // fragment.  Measured with the DMA removed (-DPTTS_ABLATE=32): the old per-fragment arithmetic + the fetch cost
  // 20 % (Linear) to 60 % (conv k7) of these kernels.  issue() must be called once per stage, in k order.
  const float *xp[XPW];
  bool xz[XPW];
  int s_cf = 0, s_tap = 0;
  auto row_base = [&](int q, int tap) {
    if (a.ntaps == 1) {
      xp[q] = Xc + (((size_t)l_mt[q] * a.XF) * 64 + lane) * 4;
      xz[q] = false;
      return;
    }
    const int ts = l_t[q] * a.xstride + tap - halo;
    const float *base = Xc;
    long rr = (long)l_bT[q] * a.xstride + ts;
    if (ts < 0) {
      if (a.halo_mode == 0) { base = Xp; rr += (long)a.T * a.xstride; }
      else if (a.halo_mode == 2) rr = (long)l_bT[q] * a.xstride;
    }
    xp[q] = base + (((size_t)(rr >> 4) * a.XF) * 64 + (lane & 48) + (rr & 15)) * 4;
    xz[q] = ts < 0 && a.halo_mode == 1;  // whole-signal zero padding: every cf reads the zero line
  };
#pragma unroll
  for (int q = 0; q < XPW; ++q) row_base(q, 0);
  // W fragments f = wave + 4 i of a stage (index NX + f = NX + kc * BNT + n): base pointers at k-fragment 0
  constexpr int WPW = BNT * KC / 4;
  const float *wp[WPW];
  int wslot[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int f = wave + 4 * i, kc = f / BNT, n = f - kc * BNT;
    const int nt = nt0 + n < a.NT ? nt0 + n : a.NT - 1;
    wp[i] = a.W + ((size_t)nt * a.KF + kc) * 256 + lane * 4;
    wslot[i] = NX + f;
  }
  auto issue = [&](int kf0, int buf) {
    // X fragments: index kc * BMT + m, m = wave + 4q
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int q = 0; q < XPW; ++q) {
        const float *src = xz[q] ? a.zeros : xp[q] + (size_t)s_cf * 256;
        GLDS16(src, &lds[buf][kc * BMT + wave + 4 * q][0]);
      }
      if (++s_cf == a.CF) {  // next tap: one row further (a.CF == a.KF for a Linear: never taken)
        s_cf = 0;
        ++s_tap;
#pragma unroll
        for (int q = 0; q < XPW; ++q) row_base(q, s_tap);
      }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) GLDS16(wp[i] + (size_t)kf0 * 256, &lds[buf][wslot[i]][0]);
  };

  // fused tail: wave w finishes column tiles w, w + 4, .. of the second layer; its weights are requested now
  constexpr int T2 = NT2 / 4;
  f32x4 w2[T2 > 0 ? T2 : 1][BNT];
  if constexpr (NT2 > 0) {
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
      for (int k = 0; k < BNT; ++k)
        w2[t][k] = *(const f32x4 *)(a.W2 + (((size_t)(wave + 4 * t) * BNT + k) * 64 + lane) * 4);
  }

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sx[WMT], sxx[WMT];
#pragma unroll
  for (int j = 0; j < WMT; ++j) sx[j] = sxx[j] = 0.f;

  const int nst = a.KF / KC;  // host guarantees KF % KC == 0
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < nst) issue(p * KC, p);
  for (int s = 0; s < nst; ++s) {
    const int cur = s % NS;
    // this wave's DMA of stage s has landed once at most `ahead` later stages' instructions are outstanding
    const int ahead = min(NS - 2, nst - 1 - s);
    if constexpr (!(PTTS_ABLATE & 8)) {  // ablation 8 (timing only): no DMA wait
      if (ahead >= 2) wait_vmcnt<2 * IPS>();
      else if (ahead == 1) wait_vmcnt<IPS>();
      else wait_vmcnt<0>();
    }
    if constexpr (!(PTTS_ABLATE & 16)) __syncthreads();  // ... everyone's has, and the buffer of stage s-1 is free again (ablation 16: no barrier)
    if constexpr (!(PTTS_ABLATE & 32))  // ablation 32 (timing only): no operand DMA after the prologue
      if (s + NS - 1 < nst) issue((s + NS - 1) * KC, (s + NS - 1) % NS);
    // LDS reads are register double-buffered: the fragments of k-step kc+1 are read while the MFMAs of kc run
    // (with one wave per SIMD nothing else would cover the ~128-cycle ds_read latency)
    f32x4 xa[WMT], wa[WNT], xb[WMT], wb[WNT];
    auto rd = [&](int kc, f32x4 *x, f32x4 *w) {
#pragma unroll
      for (int j = 0; j < WMT; ++j) x[j] = lds[cur][kc * BMT + wm * WMT + j][lane];
#pragma unroll
      for (int i = 0; i < WNT; ++i) w[i] = lds[cur][NX + kc * BNT + wn * WNT + i][lane];
    };
    auto mm = [&](f32x4 *x, const f32x4 *w) {
      if constexpr (PRE == PRE_ELU) {  // the producer stored the raw value once (it is also the block's skip input)
#pragma unroll
        for (int j = 0; j < WMT; ++j) x[j] = pre4<PRE_ELU>(x[j], nullptr, 0, lane);
      }
      if constexpr (PRE == PRE_LNFOLD) {
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          sx[j] += (x[j].x + x[j].y) + (x[j].z + x[j].w);
          sxx[j] += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
        }
      }
#pragma unroll
      for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
          for (int j = 0; j < WMT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][cidx], x[j][cidx], acc[i][j], 0, 0, 0);
    };
    rd(0, xa, wa);
#pragma unroll
    for (int kc = 0; kc < KC; kc += 2) {
      // sched_barrier: keep the reads ahead of the MFMAs they overlap (the scheduler otherwise sinks them to
      // save registers and the pipe idles behind s_waitcnt lgkmcnt(0))
      if (kc + 1 < KC) rd(kc + 1, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      mm(xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 2 < KC) rd(kc + 2, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 1 < KC) mm(xb, wb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float mu[WMT], rs[WMT];
  if constexpr (PRE == PRE_LNFOLD) {
    const float invK = 1.0f / (float)(a.KF * 16);
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      sx[j] += __shfl_xor(sx[j], 16); sx[j] += __shfl_xor(sx[j], 32);
      sxx[j] += __shfl_xor(sxx[j], 16); sxx[j] += __shfl_xor(sxx[j], 32);
      mu[j] = sx[j] * invK;
      rs[j] = 1.0f / sqrtf(fmaxf(sxx[j] * invK - mu[j] * mu[j], 0.f) + a.ln_eps);
    }
  }
  if constexpr (NT2 > 0) {
    // h tile -> LDS, fragment (row tile m, k-fragment n) at hfrag[m * BNT + n]: the stage buffers are free once every
    // wave has left the K loop
    f32x4(*hfrag)[64] = &lds[0][0];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const int n = wn * WNT + i;
        f32x4 v = acc[i][j];
        if (a.bias) v += *(const f32x4 *)(a.bias + 16 * n + 4 * (lane >> 4));
        hfrag[(wm * WMT + j) * BNT + n][lane] = act4(v, a.act);
      }
    __syncthreads();
    GemmArgs a2 = a;
    a2.bias = a.bias2; a2.act = a.act2; a2.epi = EPI_RES; a2.ls = nullptr;
#pragma unroll
    for (int m = 0; m < BMT; ++m) {
      const int mt = mt0 + m;
      if (mt >= a.MT) break;
      f32x4 x[BNT];
#pragma unroll
      for (int k = 0; k < BNT; ++k) x[k] = hfrag[m * BNT + k][lane];
#pragma unroll
      for (int t = 0; t < T2; ++t) {
        f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < BNT; ++k)
#pragma unroll
          for (int cidx = 0; cidx < 4; ++cidx)
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[t][k][cidx], x[k][cidx], c2, 0, 0, 0);
        if constexpr (NT2 == 4) {
          if (a.pcm_w) {
            // == gemm_epilogue(EPI_RES) without the store: z = act2(skip + conv + bias) stays in registers, and its three
            // last-conv taps are dotted over this lane's 4 channels, then over the 16 channels of the wave's column tile
            const int nt = wave, g = lane >> 4;
            const int n0 = 16 * nt + 4 * g;
            f32x4 v = c2 + *(const f32x4 *)(a2.bias + n0);
            v = act4(*(const f32x4 *)(a.R + par * a.Rdstride + (((size_t)mt * a.RF + nt) * 64 + lane) * 4) + v, a2.act);
            if (a.Y) *(f32x4 *)(a.Y + par * a.Ydstride + (((size_t)mt * a.YF + nt) * 64 + lane) * 4) = v;  // debug taps only
            float *pp = (float *)&lds[0][BMT * BNT][0];  // [wave][m][tap][16 rows], behind the h tile
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const f32x4 wd = *(const f32x4 *)(a.pcm_w + ((size_t)(d * 4 + nt) * 64 + 16 * g) * 4);
              float p = (v.x * wd.x + v.y * wd.y) + (v.z * wd.z + v.w * wd.w);
              p += __shfl_xor(p, 16);
              p += __shfl_xor(p, 32);
              if (lane < 16) pp[((wave * BMT + m) * 3 + d) * 16 + lane] = p;
            }
            continue;
          }
        }
        gemm_epilogue(a2, c2, wave + 4 * t, mt, lane, par);
      }
    }
    if constexpr (NT2 == 4) {
      if (a.pcm_w) {
        __syncthreads();
        if (wave != 0) return;
        // lane = row of the workgroup's 64; S_d = tap d's dot product over all 64 channels of that row;
        // pcm[u] = bias + S_2[u] + S_1[u - 1] + S_0[u - 2] (causal, reference conv.py:84-91 with k = 3)
        const float *pp = (const float *)&lds[0][BMT * BNT][0];
        const int m = lane >> 4, ml = lane & 15;
        float sd[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          sd[d] = (pp[((0 * BMT + m) * 3 + d) * 16 + ml] + pp[((1 * BMT + m) * 3 + d) * 16 + ml]) +
                  (pp[((2 * BMT + m) * 3 + d) * 16 + ml] + pp[((3 * BMT + m) * 3 + d) * 16 + ml]);
        const float s1u = __shfl_up(sd[1], 1), s0u = __shfl_up(sd[0], 2), s0v = __shfl_up(sd[0], 1);
        const size_t row = (size_t)16 * mt0 + lane;
        if (16 * mt0 + lane < a.M) a.pcm_part[row] = sd[2] + (lane >= 1 ? s1u : 0.f) + (lane >= 2 ? s0u : 0.f);
        if (lane == 63) {  // what the next tile's rows 0 and 1 miss
          float *c = a.pcm_carry + par * a.pcm_cstride + (size_t)(mt0 / BMT) * 2;
          c[0] = sd[1] + s0v;
          c[1] = sd[0];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int nt = nt0 + wn * WNT + i, mt = mt0 + wm * WMT + j;
      if (nt >= a.NT || mt >= a.MT) continue;
      f32x4 v = acc[i][j];
      if constexpr (PRE == PRE_LNFOLD) {
        const int n0 = 16 * nt + 4 * (lane >> 4);
        v = (v - *(const f32x4 *)(a.ln_s + n0) * mu[j]) * rs[j] + *(const f32x4 *)(a.ln_c + n0);
      }
      gemm_epilogue(a, v, nt, mt, lane, par);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight packing (load time).  mode 0: Linear / Conv1d weight [N][C][ntaps] -> value(n, c, tap);
// mode 1: ConvTranspose1d weight [C][cout][2s] viewed as a 2-tap conv with n' = j*cout + n:
//         tap 1 (current input row) uses kernel index j, tap 0 (previous row) uses j + s.
// dst[nt][tap*CF + cf][lane][j4] = W[n = 16nt + (lane&15)][c = 16cf + 4(lane>>4) + j4][tap]
// ---------------------------------------------------------------------------------------------
static __global__ void pack_weight_kernel(const float *src, float *dst, int N, int C, int ntaps, int mode, int cout,
                                   int stride, int nt_off, int KF, long total, const float *colscale, int Creal) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j4 = i & 3;
  int lane = (i >> 2) & 63;
  long f = i >> 8;
  int kf = f % KF;
  int nt = f / KF;
  int CF = C / 16;
  int tap = kf / CF, cf = kf - tap * CF;
  int n = 16 * nt + (lane & 15);
  int c = 16 * cf + 4 * (lane >> 4) + j4;
  float v = 0.f;
  if (n < N) {
    if (mode == 0) {
      v = c < Creal ? src[((size_t)n * Creal + c) * ntaps + tap] : 0.f;
      if (colscale) v *= colscale[c];
    } else {
      int j = n / cout, nn = n - j * cout;
      int kidx = tap == 1 ? j : j + stride;
      v = src[((size_t)c * cout + nn) * (2 * stride) + kidx];
    }
  }
  dst[((size_t)(nt + nt_off) * KF) * 256 + (size_t)kf * 256 + lane * 4 + j4] = v;
}

// LayerNorm folding constants of one Linear [N][K]: s[n] = sum_k W[n][k] g[k], c[n] = sum_k W[n][k] b[k] + bias[n].
// One wave per output row; load time only.
static __global__ void fold_ln_kernel(const float *W, const float *g, const float *b, const float *bias, float *s_out,
                               float *c_out, int N, int K, int n_off) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float s = 0.f, c = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(size_t)n * K + k];
    s += w * g[k];
    c += w * b[k];
  }
  for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
  if (lane == 0) {
    s_out[n_off + n] = s;
    c_out[n_off + n] = c + (bias ? bias[n] : 0.f);
  }
}

// int8 weight-only quantisation of one packed Linear matrix (load time).  One workgroup per 16-row n-tile:
// per output row, scale = max|w| / 127 (symmetric, per channel), q = rint(w / scale) in [-127, 127], stored as the
// biased byte q + 128 in the Q8 layout dst[nt][kf/4][lane][4 * (kf % 4) + j4].  For a matrix that follows a
// LayerNorm (gam/bet non-null) the fold vectors are built from the DEQUANTISED weights:
//   ln_s[n] = sum_k Wdq[n][k] gam[k],  ln_c[n] = sum_k Wdq[n][k] bet[k] + bias[n]
// so the result equals LayerNorm followed by a Linear whose weight is Wdq, which is what the oracle computes.
// (The reference quantises the same Linear layers with torch.ao / torchao dynamic int8, quantization.py:60-128;
// this build keeps activations in fp32: weight-only, per channel.)
static __global__ __launch_bounds__(256) void quantize_packed_kernel(const float *src, uint8_t *dst, float *wscale, int KF,
                                                              const float *gam, const float *bet, const float *bias,
                                                              int N, float *ln_s, float *ln_c) {
  __shared__ int smax[16];
  __shared__ float ps[256], pc[256];
  const int nt = blockIdx.x, tid = threadIdx.x;
  const int lane = tid >> 2, j4 = tid & 3, row = lane & 15;
  if (tid < 16) smax[tid] = 0;
  __syncthreads();
  const float *s = src + (size_t)nt * KF * 256;
  float m = 0.f;
  for (int kf = 0; kf < KF; ++kf) m = fmaxf(m, fabsf(s[(size_t)kf * 256 + tid]));
  atomicMax(&smax[row], __float_as_int(m));  // non-negative floats order like their bit patterns
  __syncthreads();
  const float mx = __int_as_float(smax[row]);
  const float scale = mx > 0.f ? mx / 127.0f : 1.0f;
  float as = 0.f, ac = 0.f;
  uint8_t *d = dst + (size_t)nt * KF * 256 + lane * 16 + j4;
  for (int kf = 0; kf < KF; ++kf) {
    int q = (int)rintf(s[(size_t)kf * 256 + tid] / scale);
    q = q > 127 ? 127 : q < -127 ? -127 : q;
    d[(size_t)(kf >> 2) * 1024 + 4 * (kf & 3)] = (uint8_t)(q + 128);
    if (gam) {
      const int k = 16 * kf + 4 * (lane >> 4) + j4;
      as += (float)q * gam[k];
      ac += (float)q * bet[k];
    }
  }
  ps[tid] = as;
  pc[tid] = ac;
  __syncthreads();
  if (tid < 16) {
    const float sc = __int_as_float(smax[tid]) > 0.f ? __int_as_float(smax[tid]) / 127.0f : 1.0f;
    wscale[16 * nt + tid] = sc;
    if (gam) {
      // the 16 holders of row `tid`: lanes tid, tid+16, tid+32, tid+48, four components each; fixed order
      float ts = 0.f, tc = 0.f;
      for (int g = 0; g < 4; ++g)
        for (int c = 0; c < 4; ++c) {
          ts += ps[(16 * g + tid) * 4 + c];
          tc += pc[(16 * g + tid) * 4 + c];
        }
      const int n = 16 * nt + tid;
      ln_s[n] = sc * ts;
      ln_c[n] = sc * tc + ((bias && n < N) ? bias[n] : 0.f);
    }
  }
}

static __global__ void pack_bias_kernel(const float *src, float *dst, int N, int mode, int cout, int n_off, int Npad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  float v = 0.f;
  if (i < N && src) v = mode == 0 ? src[i] : src[i % cout];
  dst[n_off + i] = v;
}

// ---------------------------------------------------------------------------------------------
// Layout conversion: row-major [M][K] <-> FM.  One thread per float4.
// ---------------------------------------------------------------------------------------------
static __global__ void to_fm_kernel(const float *src, float *dst, int M, int K, int MT) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int KF = K / 16;
  long total = (long)MT * KF * 64;
  if (i >= total) return;
  int lane = i & 63;
  long f = i >> 6;
  int kf = f % KF;
  int mt = f / KF;
  int m = 16 * mt + (lane & 15);
  int k = 16 * kf + 4 * (lane >> 4);
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m < M) v = *(const f32x4 *)(src + (size_t)m * K + k);
  *(f32x4 *)(dst + i * 4) = v;
}

static __global__ void from_fm_kernel(const float *src, float *dst, int M, int K, int F, int f_off) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int K4 = K / 4;
  if (i >= (long)M * K4) return;
  int m = i / K4;
  int k = (i - (long)m * K4) * 4;
  int kf = k >> 4;
  const float *p = src + (((size_t)(m >> 4) * F + kf + f_off) * 64 + 16 * ((k & 15) >> 2) + (m & 15)) * 4;
  *(f32x4 *)(dst + (size_t)m * K + k) = *(const f32x4 *)p;
}

// FlowLM step input: latent (external or the previous output; NaN = BOS -> bos_emb, reference
// flow_lm.py:121) to FM; LSD start point: noise (or zeros) to the plain `lat` buffer and its FM copy.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// N(0,1) from a counter (Box-Muller on two 24-bit uniforms of a splitmix64 hash): perf-run noise source
__device__ __forceinline__ float counter_normal(unsigned long long seed, unsigned ctr, unsigned idx) {
  unsigned long long z = mix64(seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)ctr << 32) | idx));
  float u1 = ((float)((z >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);
  float u2 = (float)((z >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// cos / sin of (offset[b] + t) * freq[i] for every row m = b*Tq + t, i < 32 (reference rope.py:28-50):
// computed once per step instead of once per (layer, n-tile) in the QKV epilogue
struct RopeArgs {
  const int *offset;
  const float *freq;
  float *tab;
  int M, Tq;
};
__device__ __forceinline__ void rope_table_entry(const RopeArgs &r, int i) {
  if (i >= r.M * 32) return;
  int m = i >> 5, f = i & 31;
  int b = m / r.Tq, t = m - b * r.Tq;
  float sn, cs;
  sincosf(r.freq[f] * (float)(r.offset[b] + t), &sn, &cs);
  r.tab[2 * i] = cs;
  r.tab[2 * i + 1] = sn;
}

// Step prologue of the FlowLM: BOS substitution + noise + FM conversion of the input latent (blocks < nb_prep) and the
// step's RoPE table (the remaining blocks): one launch instead of two.
static __global__ void prep_lm_kernel(const float *lat_in, const float *bos, const float *noise, float *x_fm, float *lat,
                               float *lat_fm, int B, int ldim, int MT, float rng_std, unsigned long long rng_seed,
                               const int *rng_ctr, int nb_prep, RopeArgs rope) {
  if ((int)blockIdx.x >= nb_prep) {
    rope_table_entry(rope, (blockIdx.x - nb_prep) * blockDim.x + threadIdx.x);
    return;
  }
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int KF = ldim / 16;
  if (i >= MT * KF * 64) return;
  int lane = i & 63;
  int f = i >> 6;
  int kf = f % KF, mt = f / KF;
  int m = 16 * mt + (lane & 15);
  int k = 16 * kf + 4 * (lane >> 4);
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f}, z = v;
  if (m < B) {
    v = *(const f32x4 *)(lat_in + (size_t)m * ldim + k);
    f32x4 b = *(const f32x4 *)(bos + k);
    v.x = v.x != v.x ? b.x : v.x;
    v.y = v.y != v.y ? b.y : v.y;
    v.z = v.z != v.z ? b.z : v.z;
    v.w = v.w != v.w ? b.w : v.w;
    if (noise) {
      z = *(const f32x4 *)(noise + (size_t)m * ldim + k);
    } else if (rng_std > 0.f) {
      const unsigned ctr = (unsigned)*rng_ctr, base = (unsigned)(m * ldim + k);
      z.x = rng_std * counter_normal(rng_seed, ctr, base + 0);
      z.y = rng_std * counter_normal(rng_seed, ctr, base + 1);
      z.z = rng_std * counter_normal(rng_seed, ctr, base + 2);
      z.w = rng_std * counter_normal(rng_seed, ctr, base + 3);
    }
  }
  *(f32x4 *)(x_fm + (size_t)i * 4) = v;
  *(f32x4 *)(lat_fm + (size_t)i * 4) = z;
  if (m < B) *(f32x4 *)(lat + (size_t)m * ldim + k) = z;
}

// Step prologue of the FlowLM + input_linear in ONE launch (round 3): the K = ldim (32) GEMM  x = input_linear(where(isnan(z), bos, z))
// (reference flow_lm.py:121-122) is one wave per 16x16 output tile reading the plain latent rows directly (BOS substitution
// on load), beside the noise / LSD start point writes and the step's RoPE table of prep_lm_kernel.  As its own K-split GEMM
// launch the tiny product cost ~10 us of a step (8 waves splitting two k-fragments) plus a dependent launch.
static __global__ __launch_bounds__(256) void prep_in_kernel(const float *lat_in, const float *bos, const float *noise, const float *w_in,
                                                             float *x_out, float *lat, float *lat_fm, int B, int ldim, int MT, int DF,
                                                             float rng_std, unsigned long long rng_seed, const int *rng_ctr,
                                                             int nb_gemm, int nb_prep, RopeArgs rope) {
  const int LF = ldim / 16;
  if ((int)blockIdx.x < nb_gemm) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= DF * MT) return;
    const int nt = tile % DF, mt = tile / DF;
    const int m = 16 * mt + (lane & 15), g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kf = 0; kf < LF; ++kf) {
      const int k = 16 * kf + 4 * g;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < B) {
        v = *(const f32x4 *)(lat_in + (size_t)m * ldim + k);
        const f32x4 b = *(const f32x4 *)(bos + k);
        v.x = v.x != v.x ? b.x : v.x; v.y = v.y != v.y ? b.y : v.y; v.z = v.z != v.z ? b.z : v.z; v.w = v.w != v.w ? b.w : v.w;
      }
      const f32x4 w = *(const f32x4 *)(w_in + (((size_t)nt * LF + kf) * 64 + lane) * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c], v[c], acc, 0, 0, 0);
    }
    *(f32x4 *)(x_out + (((size_t)mt * DF + nt) * 64 + lane) * 4) = acc;
    return;
  }
  if ((int)blockIdx.x >= nb_gemm + nb_prep) {
    rope_table_entry(rope, (blockIdx.x - nb_gemm - nb_prep) * blockDim.x + threadIdx.x);
    return;
  }
  const int i = (blockIdx.x - nb_gemm) * blockDim.x + threadIdx.x;
  if (i >= MT * LF * 64) return;
  const int lane = i & 63, f = i >> 6;
  const int kf = f % LF, mt = f / LF;
  const int m = 16 * mt + (lane & 15), k = 16 * kf + 4 * (lane >> 4);
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (m < B) {
    if (noise) {
      z = *(const f32x4 *)(noise + (size_t)m * ldim + k);
    } else if (rng_std > 0.f) {
      const unsigned ctr = (unsigned)*rng_ctr, base = (unsigned)(m * ldim + k);
      z.x = rng_std * counter_normal(rng_seed, ctr, base + 0);
      z.y = rng_std * counter_normal(rng_seed, ctr, base + 1);
      z.z = rng_std * counter_normal(rng_seed, ctr, base + 2);
      z.w = rng_std * counter_normal(rng_seed, ctr, base + 3);
    }
  }
  *(f32x4 *)(lat_fm + (size_t)i * 4) = z;
  if (m < B) *(f32x4 *)(lat + (size_t)m * ldim + k) = z;
}

// Codec frame prologue in one launch (blocks < nb_main; the frame's RoPE table rides along in the remaining blocks):
//   z = latent * emb_std + emb_mean                                  (reference tts_model.py:449)
//   zq = quantizer.output_proj(z), a 1x1 conv ldim -> C, no bias     (dummy_quantizer.py:17-18)
//   depthwise ConvTranspose1d k = 2s, stride s on that one input step (resample.py:40-51, conv.py:151-163):
//     out[b, t, c] = zq[b, c] w[c, t] + zq_prev[b, c] w[c, s + t]
// zq is double-buffered by frame parity (FM layout, one row per sequence), so `partial` is never materialised.
// The three tiny launches this replaces cost more in launch latency than in work.
static __global__ void mimi_prologue_kernel(const float *lat, const float *std, const float *mean, const float *wq,
                                     const float *wup, float *zq, long zdstride, const int *par_p, float *out, int B,
                                     int ldim, int C, int s, int nb_main, RopeArgs rope, int h16) {
  if ((int)blockIdx.x >= nb_main) {
    rope_table_entry(rope, (blockIdx.x - nb_main) * blockDim.x + threadIdx.x);
    return;
  }
  // thread = (sequence b, 4 channels c, output step t), t fastest: the s threads of one (b, c) recompute the same
  // 4 x ldim dot products from broadcast loads (cheaper than a second phase) and write s adjacent 16-byte outputs
  const int C4 = C / 4, CF = C / 16;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C4 * s) return;
  const int t = (int)(i % s);
  const int c = (int)((i / s) % C4) * 4, b = (int)(i / ((long)s * C4));
  const int par = *par_p & 1;
  f32x4 zc = {0.f, 0.f, 0.f, 0.f};
  const float *lr = lat + (size_t)b * ldim;
  for (int k = 0; k < ldim; k += 4) {
    const f32x4 z = *(const f32x4 *)(lr + k) * *(const f32x4 *)(std + k) + *(const f32x4 *)(mean + k);
    const f32x4 w0 = *(const f32x4 *)(wq + (size_t)(c + 0) * ldim + k), w1 = *(const f32x4 *)(wq + (size_t)(c + 1) * ldim + k);
    const f32x4 w2 = *(const f32x4 *)(wq + (size_t)(c + 2) * ldim + k), w3 = *(const f32x4 *)(wq + (size_t)(c + 3) * ldim + k);
    zc.x += (z.x * w0.x + z.y * w0.y) + (z.z * w0.z + z.w * w0.w);
    zc.y += (z.x * w1.x + z.y * w1.y) + (z.z * w1.z + z.w * w1.w);
    zc.z += (z.x * w2.x + z.y * w2.y) + (z.z * w2.z + z.w * w2.w);
    zc.w += (z.x * w3.x + z.y * w3.y) + (z.z * w3.z + z.w * w3.w);
  }
  const size_t zi = (((size_t)(b >> 4) * CF + (c >> 4)) * 64 + 16 * ((c & 15) >> 2) + (b & 15)) * 4;
  const f32x4 zp = *(const f32x4 *)(zq + (par ^ 1) * zdstride + zi);
  if (t == 0) *(f32x4 *)(zq + par * zdstride + zi) = zc;  // next frame's zq_prev
  const int k2 = 2 * s;
  f32x4 o;
  o.x = zc.x * wup[(c + 0) * k2 + t] + zp.x * wup[(c + 0) * k2 + s + t];
  o.y = zc.y * wup[(c + 1) * k2 + t] + zp.y * wup[(c + 1) * k2 + s + t];
  o.z = zc.z * wup[(c + 2) * k2 + t] + zp.z * wup[(c + 2) * k2 + s + t];
  o.w = zc.w * wup[(c + 3) * k2 + t] + zp.w * wup[(c + 3) * k2 + s + t];
  const long m = (long)b * s + t;
  if (h16) *(bf16x4 *)((__bf16 *)out + fmh_off((size_t)m, c, C / 32)) = to_bf16x4(o);
  else *(f32x4 *)(out + (((size_t)(m >> 4) * CF + (c >> 4)) * 64 + 16 * ((c & 15) >> 2) + (m & 15)) * 4) = o;
}

static __global__ void rope_table_kernel(const int *offset, const float *freq, float *tab, int M, int Tq) {
  rope_table_entry(RopeArgs{offset, freq, tab, M, Tq}, blockIdx.x * blockDim.x + threadIdx.x);
}

// mono audio [T] -> FM rows of 16 channels (channel 0 = sample, the rest zero) for the encoder's first conv
static __global__ void audio_to_fm_kernel(const float *audio, float *x_fm, int n_valid, int n_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 slot per (row tile, lane)
  if (i >= (n_rows / 16) * 64) return;
  int lane = i & 63, mt = i >> 6;
  int m = 16 * mt + (lane & 15);
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if ((lane >> 4) == 0 && m < n_valid) v.x = audio[m];
  *(f32x4 *)(x_fm + (size_t)i * 4) = v;
}

static __global__ void add_int_kernel(int *p, int n, int inc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += inc;
}
// end-of-step bookkeeping in one launch: offsets of all rows += inc (increment_steps, reference
// stateful_module.py:19-26) and one scalar counter (noise counter / frame parity) += 1
// `active` (optional): parked rows of a continuously batched state keep their offset
static __global__ void step_tail_kernel(int *offset, int n, int inc, int *counter, const int *active) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (!active || active[i])) offset[i] += inc;
  if (i == 0 && counter) *counter += 1;
}
// zero row `row` of an FM buffer with F k-fragments per row tile (a joining utterance's codec carries)
static __global__ void zero_row_fm_kernel(float *buf, int F, int row) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // (kf, group, j4)
  if (i >= F * 16) return;
  const int kf = i >> 4, g = (i >> 2) & 3, j4 = i & 3;
  buf[(((size_t)(row >> 4) * F + kf) * 64 + 16 * g + (row & 15)) * 4 + j4] = 0.f;
}
// A sequence whose first `len` positions (a multiple of 16) are the keys / values of ANOTHER state's cache: the voice prefix
// that every utterance cloned from one voice state shares (reference: the per-chunk deepcopy of the voice state,
// tts_model.py:637-638, gives every generation its own copy of the same bytes).  `kv` = base of the owner's cache
// [L][2][1][H][cap][64]; the sequence's own cache holds positions >= len at their absolute slots.
struct KvPrefix {
  const float *kv;
  int cap, len;
};
static __global__ void set_prefix_kernel(KvPrefix *p, int n, const float *kv, int cap, int len) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = KvPrefix{kv, cap, len};
}
// LUTConditioner._get_condition (reference text.py:74-76): out[i] = table[tokens[i]], one float4 per thread; an id outside
// the table yields a zero row (the host checks the ids before the call)
static __global__ void embed_gather_kernel(const float *table, int n_bins, int D, const long long *tokens, long n, float *out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = D / 4;
  if (i >= n * d4) return;
  const long r = i / d4;
  const int c = (int)(i - r * d4);
  const long long t = tokens[r];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (t >= 0 && t < n_bins) v = *(const f32x4 *)(table + (size_t)t * D + 4 * c);
  *(f32x4 *)(out + (size_t)r * D + 4 * c) = v;
}
static __global__ void set_int_kernel(int *p, int n, int v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
static __global__ void fill_kernel(float *p, long n, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// The flow MLP's variance-based "RMSNorm" on one row + average of the two time embeddings
// (reference mlp.py:20-25,70,204-206).  Load-time constant folding, one wave.
static __global__ void tcomb_kernel(const float *h0_fm, const float *h1_fm, const float *alpha0, const float *alpha1,
                             float *out, int fd) {
  // row 0 of an FM tensor with F = fd/16: element k at ((k/16)*64 + 16*((k%16)/4))*4 + k%4
  const int lane = threadIdx.x;
  auto at = [&](const float *p, int k) { return p[((size_t)(k >> 4) * 64 + 16 * ((k & 15) >> 2)) * 4 + (k & 3)]; };
  float r[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int e = 0; e < 2; ++e) {
    const float *p = e ? h1_fm : h0_fm;
    float s = 0.f;
    for (int k = lane; k < fd; k += 64) s += at(p, k);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    float mean = s / fd, q = 0.f;
    for (int k = lane; k < fd; k += 64) { float d = at(p, k) - mean; q += d * d; }
    for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
    r[e][0] = 1.0f / sqrtf(1e-5f + q / (fd - 1));  // unbiased variance
  }
  for (int k = lane; k < fd; k += 64)
    out[k] = (at(h0_fm, k) * (alpha0[k] * r[0][0]) + at(h1_fm, k) * (alpha1[k] * r[1][0])) * 0.5f;
}

// cat(cos(t f), sin(t f)) for one scalar t -> FM row 0 (reference mlp.py:79-81)
static __global__ void timestep_embed_kernel(const float *freqs, float t, float *e_fm, int half) {
  int k = threadIdx.x + blockIdx.x * blockDim.x;
  if (k >= 2 * half) return;
  float arg = t * freqs[k % half];
  float v = k < half ? cosf(arg) : sinf(arg);
  e_fm[((size_t)(k >> 4) * 64 + 16 * ((k & 15) >> 2)) * 4 + (k & 3)] = v;
}

// ---------------------------------------------------------------------------------------------
// Streaming attention on the KV cache, all on fp32 MFMA, no LDS.  One wave per (sequence, head,
// 16-query block, key split).  S^T = K Q^T puts the query on the lane (softmax reductions are two
// xor-shuffles), and its accumulator is directly the A operand of O = P V.
// Causal + optional sliding-window mask (reference transformer.py:22-29); cache is linear (FlowLM)
// or a ring of `ring` slots (Mimi decoder, context 250).
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const float *Q, *Kc, *Vc;
  const KvPrefix *pre = nullptr;  // per-sequence shared prefixes (linear caches only) or null
  int layer = 0;                  // layer index into a prefix owner's cache
  const int *offset;
  int H, Tq, QB, cap, ring, ctx, splits;
  float *part;  // [BH*QB][splits][16][64 + 2 (pad to 80)]
  float *Y;
  int YF;
  int h16;  // output as bf16 FMH (YF = 32-column blocks) instead of fp32 FM
  int nseq = 0;  // sequences in the launch (attn_cascade_kernel: its groups of R may overhang)
};
#define ATT_PSTRIDE 80
// key tiles [0, ptl) of (sequence b, head h) come from Kp / Vp (the prefix owner's cache), the rest from the sequence's own
__device__ __forceinline__ int attn_prefix(const AttnArgs &a, int b, int h, const float *&Kp, const float *&Vp) {
  if (!a.pre) return 0;
  const KvPrefix p = a.pre[b];
  if (p.len <= 0) return 0;
  Kp = p.kv + ((size_t)(2 * a.layer) * a.H + h) * p.cap * 64;
  Vp = p.kv + ((size_t)(2 * a.layer + 1) * a.H + h) * p.cap * 64;
  return p.len >> 4;
}

// Cross-row all-reduce over the four 16-lane rows of a wave on the vector ALU (gfx950 v_permlane16_swap /
// v_permlane32_swap): lanes c, c + 16, c + 32, c + 48 end up with the max / sum of their four values.  The ds_bpermute
// path of __shfl_xor costs an LDS round trip (> 100 cycles) per step, in the middle of every tile's dependent chain.
// (inline asm: ROCm 7.2's clang returns the FIRST result of __builtin_amdgcn_permlane{16,32}_swap in both elements
// of its result vector - tests/hip/xrow_test.hip, run by tests/test_gpu_parity_r3.py - so the builtin cannot be used; the s_nop covers the VALU-write ->
// permlane-swap hazard the compiler would otherwise pad)
struct xrow_pair { float a, b; };
__device__ __forceinline__ xrow_pair xrow_swap16(float x) {
  xrow_pair r = {x, x};
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
  return r;
}
__device__ __forceinline__ xrow_pair xrow_swap32(float x) {
  xrow_pair r = {x, x};
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
  return r;
}
__device__ __forceinline__ float xrow_max(float x) {
  xrow_pair r = xrow_swap16(x);
  x = fmaxf(r.a, r.b);
  r = xrow_swap32(x);
  return fmaxf(r.a, r.b);
}
__device__ __forceinline__ float xrow_sum(float x) {
  xrow_pair r = xrow_swap16(x);
  x = r.a + r.b;
  r = xrow_swap32(x);
  return r.a + r.b;
}

// Output of one (query c, 16-wide d group g) lane: o[j][rr] = O[query c][d = 16 g + 4 rr + j]
__device__ __forceinline__ void attn_store_out(const AttnArgs &a, int b, int h, int qb, int nq, int lane, const f32x4 o[4],
                                               float linv) {
  const int c = lane & 15, g = lane >> 4;
  if (c >= nq) return;
  const size_t m = (size_t)b * a.Tq + 16 * qb + c;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    f32x4 v;
    v.x = o[0][rr] * linv; v.y = o[1][rr] * linv; v.z = o[2][rr] * linv; v.w = o[3][rr] * linv;
    // column n = h*64 + 16g + 4rr + j -> fragment 4h + g, k-group rr
    if (a.h16) *(bf16x4 *)((__bf16 *)a.Y + fmh_off(m, h * 64 + 16 * g + 4 * rr, a.YF)) = to_bf16x4(v);
    else *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + g) * 64 + 16 * rr + (m & 15)) * 4) = v;
  }
}

// NW waves of a workgroup share one (sequence, head, query block, key split): each streams a contiguous run of the
// split's key tiles and the partial (o, m, l) meet in LDS in fixed wave order (as in attn_decode_kernel), so a codec
// frame reaches ~2048 waves without partial buffers and without the combine launch.
//   S^T = K Q^T  : lane (c = query, g) holds s[r] = score(key 4g + r, query c)
//   O^T = V^T P^T: A = V[key 4g + r'][4 i + j] (the lane's own 16-byte V load), B = p[r'] -> lane (c, g) accumulates
//                  O[query c][16 g + 4 rr + j]: the online-softmax rescale and the final 1/l are per-LANE scalars,
//                  and the only cross-lane traffic of a tile is one max and one sum over the four rows (xrow_*).
template <int NW, int D = 3>
__global__ __launch_bounds__(64 * NW) void attn_kernel(AttnArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  const int bh = blockIdx.x, qb = blockIdx.y, sp = blockIdx.z;
  const int b = bh / a.H, h = bh - b * a.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int off = a.offset[b];
  const int q0 = off + 16 * qb;
  const int nq = min(16, a.Tq - 16 * qb);
  const int klo = a.ctx > 0 ? max(0, q0 - a.ctx + 1) : 0;
  const int khi = q0 + nq;
  const int tile_lo = klo >> 4, tile_hi = (khi + 15) >> 4;
  const int per = (tile_hi - tile_lo + a.splits - 1) / a.splits;
  const int gs = tile_lo + sp * per, ge = min(tile_hi, gs + per);  // the workgroup's tiles
  const int perw = (max(ge - gs, 0) + NW - 1) / NW;
  const int ts = gs + wave * perw;
  const int te = min(ge, ts + perw);                                 // this wave's tiles

  f32x4 qf[4];
#pragma unroll
  for (int df = 0; df < 4; ++df)
    qf[df] = *(const f32x4 *)(a.Q + ((((size_t)bh * a.QB + qb) * 4 + df) * 64 + lane) * 4) * 0.125f;  // 1/sqrt(64)

  f32x4 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const float *Kown = a.Kc + (size_t)bh * a.cap * 64;
  const float *Vown = a.Vc + (size_t)bh * a.cap * 64;
  const float *Kpre = Kown, *Vpre = Vown;
  const int ptl = attn_prefix(a, b, h, Kpre, Vpre);  // shared-prefix tiles: a pointer select, ONE load site
  const int pq = q0 + c;

  auto load_tile = [&](int tile, f32x4 *kk, f32x4 *vv) {
    const int p0 = tile * 16;
    const int slot0 = a.ring ? (p0 % a.ring) : p0;
    const float *Kb = tile < ptl ? Kpre : Kown, *Vb = tile < ptl ? Vpre : Vown;
#pragma unroll
    // each (sequence, head) streams its keys and values once per launch: non-temporal loads (see attn_decode_kernel)
    for (int df = 0; df < 4; ++df)
      kk[df] = __builtin_nontemporal_load((const f32x4 *)(Kb + (size_t)(slot0 + c) * 64 + 16 * df + 4 * g));
#pragma unroll
    for (int r = 0; r < 4; ++r)
      vv[r] = __builtin_nontemporal_load((const f32x4 *)(Vb + (size_t)(slot0 + 4 * g + r) * 64 + 4 * c));
  };
  auto process = [&](int tile, const f32x4 *kf4, const f32x4 *vf4) {
    const int p0 = tile * 16;
    // four independent accumulators (one per 16-wide slice of d), issued round-robin: no MFMA waits for
    // the 40-cycle dependent-accumulator latency
    f32x4 sp4[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) sp4[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
      for (int df = 0; df < 4; ++df)
        sp4[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf4[df][cidx], qf[df][cidx], sp4[df], 0, 0, 0);
    const f32x4 s = (sp4[0] + sp4[1]) + (sp4[2] + sp4[3]);
    // s[r] = score(key p0 + 4g + r, query c)
    bool ok[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pk = p0 + 4 * g + r;
      ok[r] = (c < nq) && (pk <= pq) && (a.ctx <= 0 || pq - pk < a.ctx);
      mx = ok[r] ? fmaxf(mx, s[r]) : mx;
    }
    mx = xrow_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    f32x4 p;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = ok[r] ? expf(s[r] - m_new) : 0.f;
      ps += p[r];
    }
    ps = xrow_sum(ps);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] *= alpha;
    // O^T[d = 4i + j][query] += sum_key V[key][4i + j] P[query][key]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].x, p[r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].y, p[r], o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].z, p[r], o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].w, p[r], o[3], 0, 0, 0);
    }
  };
  // D rotating register tiles: the K/V of the next D - 1 key tiles (8 KB each) are in flight while a tile's scores,
  // softmax and P.V run (D = 3 alone; D = 2 costs 32 registers less per wave, which matters beside the other stream).
  // The prefetches are UNCONDITIONAL (tile index clamped to the last tile, whose lines are then L1/L2 hits): a
  // load behind a branch makes the compiler's s_waitcnt insertion merge the two paths conservatively and wait for
  // the newest loads as well, which silently serialises the whole pipeline (seen in the ISA: vmcnt(5)..vmcnt(0)
  // in front of the first MFMAs of a tile).
  f32x4 kt[D][4], vt[D][4];
  if (ts < te) {
    const int tl = te - 1;
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_tile(min(ts + j, tl), kt[j], vt[j]);
    int tile = ts;
    // whole groups of D tiles: one back-edge, no exits from inside the body (every extra control-flow join
    // makes the wait counts more conservative)
    for (; tile + D <= te; tile += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load_tile(min(tile + j + D - 1, tl), kt[(j + D - 1) % D], vt[(j + D - 1) % D]);
        process(tile + j, kt[j], vt[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
      if (tile + j < te) process(tile + j, kt[j], vt[j]);
  }
  if constexpr (NW > 1) {
    // a wave without tiles has m = NEG_BIG, l = 0, o = 0 and weight exp(NEG_BIG - M) = 0
    __shared__ f32x4 so[NW][4][64];
    __shared__ float sm[NW][16], sl[NW][16];
#pragma unroll
    for (int j = 0; j < 4; ++j) so[wave][j][lane] = o[j];
    if (g == 0) { sm[wave][c] = m_run; sl[wave][c] = l_run; }
    __syncthreads();
    if (wave != 0) return;
    float M = sm[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w][c]);
    float L = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float e = expf(sm[w][c] - M);
      L += sl[w][c] * e;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += so[w][j][lane] * e;
    }
    m_run = M;
    l_run = L;
  }

  if (a.splits == 1) {
    attn_store_out(a, b, h, qb, nq, lane, o, 1.0f / l_run);
  } else {
    float *pp = a.part + (((size_t)bh * a.QB + qb) * a.splits + sp) * 16 * ATT_PSTRIDE + c * ATT_PSTRIDE;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      f32x4 v;
      v.x = o[0][rr]; v.y = o[1][rr]; v.z = o[2][rr]; v.w = o[3][rr];
      *(f32x4 *)(pp + 16 * g + 4 * rr) = v;
    }
    if (g == 0) {
      pp[64] = m_run;
      pp[65] = l_run;
    }
  }
}

// Decode-step attention (ONE query per sequence): the same tiles, loads, masks, online softmax, split partials and
// output layout as attn_kernel, but on the vector ALU with wavefront reductions.  A single query would use one of
// the 16 MFMA columns; here a 16-key tile costs ~32 FMAs and a dozen cross-lane moves per lane, so the kernel is
// bound by the KV stream (reference transformer.py:135-158 with T = 1).
//   scores : lane (c = key, g) holds K[key c][16 df + 4g ..+3] (df = 0..3), dot with q, xor-reduce over g
//   values : lane (c, g) holds V[key 4g + r][4c ..+3] (r = 0..3), o[4c..] += p[key] * V, xor-reduce over g at the end
// NW waves of a workgroup share one (sequence, head, key split): each streams a contiguous run of key tiles and the
// partial results meet in LDS, so a CU has NW times the bytes in flight without extra partial buffers or launches.
template <int NW, bool NT = true>
__global__ __launch_bounds__(64 * NW) void attn_decode_kernel(AttnArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  const int bh = blockIdx.x, sp = blockIdx.z;
  const int b = bh / a.H, h = bh - b * a.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int pq = a.offset[b];  // position of the query; keys klo .. pq
  const int klo = a.ctx > 0 ? max(0, pq - a.ctx + 1) : 0;
  const int tile_lo = klo >> 4, tile_hi = (pq + 16) >> 4;
  const int per = (tile_hi - tile_lo + a.splits - 1) / a.splits;
  const int gs = tile_lo + sp * per, ge = min(tile_hi, gs + per);  // the workgroup's tiles
  const int perw = (max(ge - gs, 0) + NW - 1) / NW;
  const int ts = gs + wave * perw;
  const int te = min(ge, ts + perw);                                 // this wave's tiles

  f32x4 qv[4];  // q[16 df + 4g + j] * 1/sqrt(64): query row 0 of the block sits in lanes 16 g
#pragma unroll
  for (int df = 0; df < 4; ++df)
    qv[df] = *(const f32x4 *)(a.Q + ((((size_t)bh * a.QB) * 4 + df) * 64 + 16 * g) * 4) * 0.125f;
  const float *Kown = a.Kc + (size_t)bh * a.cap * 64;
  const float *Vown = a.Vc + (size_t)bh * a.cap * 64;
  const float *Kpre = Kown, *Vpre = Vown;
  const int ptl = attn_prefix(a, b, h, Kpre, Vpre);
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  auto load_tile = [&](int tile, f32x4 *kk, f32x4 *vv) {
    const int p0 = tile * 16;
    const int slot0 = a.ring ? (p0 % a.ring) : p0;
    const float *Kb = tile < ptl ? Kpre : Kown, *Vb = tile < ptl ? Vpre : Vown;
#pragma unroll
    // the cache is streamed once per step: non-temporal loads (profiles/r01_fetch_size_calibration.csv: a 1 GiB
    // stream reads at 7.4 TB/s with nt loads, 4.8 TB/s with plain ones)
    for (int df = 0; df < 4; ++df) {
      const f32x4 *p = (const f32x4 *)(Kb + (size_t)(slot0 + c) * 64 + 16 * df + 4 * g);
      kk[df] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 *p = (const f32x4 *)(Vb + (size_t)(slot0 + 4 * g + r) * 64 + 4 * c);
      vv[r] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  };
  auto process = [&](int tile, const f32x4 *kk, const f32x4 *vv) {
    float s = 0.f;
#pragma unroll
    for (int df = 0; df < 4; ++df)
      s += (kk[df].x * qv[df].x + kk[df].y * qv[df].y) + (kk[df].z * qv[df].z + kk[df].w * qv[df].w);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);  // score of key tile*16 + c, in all four lanes of column c
    const int pk = tile * 16 + c;
    const bool ok = (pk <= pq) && (a.ctx <= 0 || pq - pk < a.ctx);
    float mx = ok ? s : NEG_BIG;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    const float p = ok ? expf(s - m_new) : 0.f;
    float ps = p;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) ps += __shfl_xor(ps, d);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    o *= alpha;
#pragma unroll
    for (int r = 0; r < 4; ++r) o += vv[r] * __shfl(p, 4 * g + r);  // p of key 4g + r lives in lane 4g + r
  };
  f32x4 k0[4], v0[4], k1[4], v1[4], k2[4], v2[4];
  if (ts < te) {  // same branch-free 2-deep prefetch as attn_kernel
    const int tl = te - 1;
    load_tile(ts, k0, v0);
    load_tile(min(ts + 1, tl), k1, v1);
    int tile = ts;
    for (; tile + 3 <= te; tile += 3) {
      load_tile(min(tile + 2, tl), k2, v2);
      process(tile, k0, v0);
      load_tile(min(tile + 3, tl), k0, v0);
      process(tile + 1, k1, v1);
      load_tile(min(tile + 4, tl), k1, v1);
      process(tile + 2, k2, v2);
    }
    if (tile < te) process(tile, k0, v0);
    if (tile + 1 < te) process(tile + 1, k1, v1);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] += __shfl_xor(o[j], 16);
    o[j] += __shfl_xor(o[j], 32);
  }
  if constexpr (NW > 1) {
    // merge the waves' (o, m, l) in fixed order; a wave without tiles contributes exp(NEG_BIG - M) = 0
    __shared__ f32x4 so[NW][16];
    __shared__ float sm[NW], sl[NW];
    if (g == 0) so[wave][c] = o;
    if (lane == 0) { sm[wave] = m_run; sl[wave] = l_run; }
    __syncthreads();
    if (wave != 0) return;
    float M = sm[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w]);
    float L = 0.f;
    f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float e = expf(sm[w] - M);
      L += sl[w] * e;
      O += so[w][c] * e;
    }
    o = O; m_run = M; l_run = L;
  }
  if (g != 0) return;
  if (a.splits == 1) {
    const size_t m = (size_t)b * a.Tq;
    *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + (c >> 2)) * 64 + 16 * (c & 3) + (m & 15)) * 4) = o * (1.0f / l_run);
  } else {
    float *pp = a.part + (((size_t)bh * a.QB) * a.splits + sp) * 16 * ATT_PSTRIDE;  // row 0 of the block
    *(f32x4 *)(pp + 4 * c) = o;
    if (c == 0) {
      pp[64] = m_run;
      pp[65] = l_run;
    }
  }
}

// Decode-step attention, second layout (round 2).  The first kernel above spends its time in the memory system's
// latency, not its bandwidth: two 8 KB tiles in flight per wave, and per tile a chain of ~14 cross-row lane exchanges
// (LDS crossbar) that nothing hides at one wave per SIMD.  Here
//   * every load instruction reads 1 KB CONTIGUOUS (lane l: 16 B at l * 16 of a quarter tile): lane (rw = l / 16,
//     cc = l % 16) of load i holds key 4 i + rw, dims 4 cc .. 4 cc + 3 - of K and of V alike;
//   * a score is a sum over the 16 lanes of a DPP row: four v_add_f32 row_ror, no LDS crossbar;
//   * each row rw keeps its OWN online-softmax state (m, l, o) over the keys = rw mod 4; the four states are merged
//     once, after the stream, so the loop has no cross-row traffic at all;
//   * D register tiles rotate (D - 1 tiles = (D - 1) * 8 KB in flight per wave; one wave per SIMD leaves 512 VGPRs).
// Same tiles, masks, key split over NW waves, LDS merge, partial buffers and output layout as attn_decode_kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float x) {  // sum over the 16 lanes of a DPP row, in every lane of it
  x += dpp_mov_f<0x128>(x);  // row_ror:8
  x += dpp_mov_f<0x124>(x);  // row_ror:4
  x += dpp_mov_f<0x122>(x);  // row_ror:2
  x += dpp_mov_f<0x121>(x);  // row_ror:1
  return x;
}
struct KvTile { f32x4 k[4], v[4]; };

template <int NW, int D>
__global__ __launch_bounds__(64 * NW) void attn_decode2_kernel(AttnArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  const int bh = blockIdx.x, sp = blockIdx.z;
  const int b = bh / a.H, h = bh - b * a.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cc = lane & 15, rw = lane >> 4;
  const int pq = a.offset[b];  // position of the query; keys klo .. pq
  const int klo = a.ctx > 0 ? max(0, pq - a.ctx + 1) : 0;
  const int tile_lo = klo >> 4, tile_hi = (pq + 16) >> 4;
  const int per = (tile_hi - tile_lo + a.splits - 1) / a.splits;
  const int gs = tile_lo + sp * per, ge = min(tile_hi, gs + per);  // the workgroup's tiles
  const int perw = (max(ge - gs, 0) + NW - 1) / NW;
  const int ts = gs + wave * perw;
  const int te = min(ge, ts + perw);                                 // this wave's tiles

  // q[4 cc .. 4 cc + 3] / sqrt(64): row 0 of the query block, fragment cc / 4, k-group cc % 4
  const f32x4 q = *(const f32x4 *)(a.Q + ((((size_t)bh * a.QB) * 4 + (cc >> 2)) * 64 + 16 * (cc & 3)) * 4) * 0.125f;
  const float *Kown = a.Kc + (size_t)bh * a.cap * 64;
  const float *Vown = a.Vc + (size_t)bh * a.cap * 64;
  const float *Kpre = Kown, *Vpre = Vown;
  // tiles of a shared voice prefix come from its owner's cache: 64 utterances of one voice then stream those keys from L2
  // (all sequences of a head sit on one XCD: bh % 8 == h % 8 for 16 heads) instead of 64 private copies from HBM
  const int ptl = attn_prefix(a, b, h, Kpre, Vpre);
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  auto load_tile = [&](int tile, KvTile &t) {
    const int p0 = tile * 16;
    const size_t base = (size_t)(a.ring ? (p0 % a.ring) : p0) * 64 + lane * 4;
    const float *Kb = tile < ptl ? Kpre : Kown, *Vb = tile < ptl ? Vpre : Vown;  // uniform select, one load site
#pragma unroll
    for (int i = 0; i < 4; ++i) t.k[i] = __builtin_nontemporal_load((const f32x4 *)(Kb + base + 256 * i));
#pragma unroll
    for (int i = 0; i < 4; ++i) t.v[i] = __builtin_nontemporal_load((const f32x4 *)(Vb + base + 256 * i));
  };
  auto process = [&](int tile, const KvTile &t) {
    float s[4];
    bool ok[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = row16_sum((t.k[i].x * q.x + t.k[i].y * q.y) + (t.k[i].z * q.z + t.k[i].w * q.w));
      const int pk = tile * 16 + 4 * i + rw;
      ok[i] = (pk <= pq) && (a.ctx <= 0 || pq - pk < a.ctx);
      mx = ok[i] ? fmaxf(mx, s[i]) : mx;
    }
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    o *= alpha;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p = ok[i] ? expf(s[i] - m_new) : 0.f;
      ps += p;
      o += t.v[i] * p;
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
  };
  KvTile t[D];
  if (ts < te) {
    const int tl = te - 1;
    // unconditional, clamped prefetches and one back-edge (see attn_kernel: a load behind a branch serialises the
    // compiler's wait counts)
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_tile(min(ts + j, tl), t[j]);
    int tile = ts;
    for (; tile + D <= te; tile += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load_tile(min(tile + j + D - 1, tl), t[(j + D - 1) % D]);
        process(tile + j, t[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
      if (tile + j < te) process(tile + j, t[j]);
  }
  // merge the four row states (a row without a valid key has m = NEG_BIG and weight exp(NEG_BIG - M) = 0)
  float M = fmaxf(m_run, __shfl_xor(m_run, 16));
  M = fmaxf(M, __shfl_xor(M, 32));
  {
    const float e = expf(m_run - M);
    l_run *= e;
    o *= e;
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] += __shfl_xor(o[j], 16);
      o[j] += __shfl_xor(o[j], 32);
    }
    m_run = M;
  }
  if constexpr (NW > 1) {
    __shared__ f32x4 so[NW][16];
    __shared__ float sm[NW], sl[NW];
    if (rw == 0) so[wave][cc] = o;
    if (lane == 0) { sm[wave] = m_run; sl[wave] = l_run; }
    __syncthreads();
    if (wave != 0) return;
    float MM = sm[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) MM = fmaxf(MM, sm[w]);
    float L = 0.f;
    f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float e = expf(sm[w] - MM);
      L += sl[w] * e;
      O += so[w][cc] * e;
    }
    o = O; m_run = MM; l_run = L;
  }
  if (rw != 0) return;
  if (a.splits == 1) {
    const size_t m = (size_t)b * a.Tq;
    *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + (cc >> 2)) * 64 + 16 * (cc & 3) + (m & 15)) * 4) = o * (1.0f / l_run);
  } else {
    float *pp = a.part + (((size_t)bh * a.QB) * a.splits + sp) * 16 * ATT_PSTRIDE;  // row 0 of the block
    *(f32x4 *)(pp + 4 * cc) = o;
    if (cc == 0) {
      pp[64] = m_run;
      pp[65] = l_run;
    }
  }
}

// Decode-step attention for sequences that share a prefix ("cascade"): B utterances cloned from one voice state attend the
// SAME first `len` keys (KvPrefix), so the scores against those keys are a matrix product Q[rows] x K_prefix^T, not B
// separate matrix-vector products.  One workgroup = R consecutive sequences of one head:
//   * waves 0 .. R-1   ("suffix"): attn_decode2_kernel's row-state stream over the sequence's PRIVATE keys [len, pos];
//   * waves R .. R+PW-1 ("prefix"): attn_kernel's MFMA tiles over a 1/PW share of the prefix keys with the R queries in
//     the B operand's columns (R of 16 columns used: the tile costs 32 MFMAs per 16 keys whatever R is), so the prefix is
//     fetched once per R sequences, with plain (cacheable) loads - the other workgroups of the head read it from L2;
//   * the PW partial (m, l, o) of a row meet its suffix state in LDS; the suffix wave writes the output.  No partial
//     buffers in memory, no combine launch.
// A sequence whose prefix differs from that of the group's first sequence (other voice, no prefix, parked row) is
// handled entirely by its suffix wave, prefix tiles through the pointer select of the kernels above.  Linear caches,
// no window (FlowLM).  Summation order differs from attn_decode2_kernel's: results equal to fp32 rounding, not bitwise.
template <int R, int PW, int D, int NS = 1>
__global__ __launch_bounds__(64 * (R * NS + PW)) void attn_cascade_kernel(AttnArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  const int bg = blockIdx.x / a.H, h = blockIdx.x - bg * a.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b0 = bg * R;
  const KvPrefix p0 = a.pre[b0];
  const int ptl0 = p0.len >> 4;  // the group's shared tiles (0: nothing shared)
  constexpr int NP = PW + NS - 1;  // partials a sequence's first suffix wave merges: PW prefix shares, NS - 1 suffix shares
  __shared__ f32x4 so[NP][R][16];
  __shared__ float sm[NP][R], sl[NP][R];

  if (wave >= R * NS) {
    // ---- prefix wave: 16-key MFMA tiles, queries of the R sequences in columns c % R
    const int pw = wave - R * NS, c = lane & 15, g = lane >> 4;
    const int perw = (ptl0 + PW - 1) / PW;
    const int ts = pw * perw, te = min(ptl0, ts + perw);
    const int bq = min(b0 + (c & (R - 1)), a.nseq - 1);
    f32x4 qf[4];
#pragma unroll
    for (int df = 0; df < 4; ++df)  // row 0 of sequence bq's query block: Q[16 df + 4 g + j]
      qf[df] = *(const f32x4 *)(a.Q + (((((size_t)bq * a.H + h) * a.QB) * 4 + df) * 64 + 16 * g) * 4) * 0.125f;
    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_BIG, l_run = 0.f;
    const float *Kp = p0.kv + ((size_t)(2 * a.layer) * a.H + h) * p0.cap * 64;
    const float *Vp = p0.kv + ((size_t)(2 * a.layer + 1) * a.H + h) * p0.cap * 64;
    auto load_tile = [&](int tile, f32x4 *kk, f32x4 *vv) {
      const int p = tile * 16;
#pragma unroll
      for (int df = 0; df < 4; ++df) kk[df] = *(const f32x4 *)(Kp + (size_t)(p + c) * 64 + 16 * df + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = *(const f32x4 *)(Vp + (size_t)(p + 4 * g + r) * 64 + 4 * c);
    };
    auto process = [&](const f32x4 *kf4, const f32x4 *vf4) {  // every key of a prefix tile is visible to every query
      f32x4 sp4[4];
#pragma unroll
      for (int df = 0; df < 4; ++df) sp4[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
        for (int df = 0; df < 4; ++df)
          sp4[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf4[df][cidx], qf[df][cidx], sp4[df], 0, 0, 0);
      const f32x4 s = (sp4[0] + sp4[1]) + (sp4[2] + sp4[3]);  // s[r] = score(key 4 g + r, query c)
      const float mx = xrow_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = expf(m_run - m_new);
      f32x4 p;
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = expf(s[r] - m_new);
        ps += p[r];
      }
      ps = xrow_sum(ps);
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].x, p[r], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].y, p[r], o[1], 0, 0, 0);
        o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].z, p[r], o[2], 0, 0, 0);
        o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4[r].w, p[r], o[3], 0, 0, 0);
      }
    };
    f32x4 kt[D][4], vt[D][4];
    if (ts < te) {
      const int tl = te - 1;
#pragma unroll
      for (int j = 0; j < D - 1; ++j) load_tile(min(ts + j, tl), kt[j], vt[j]);
      int tile = ts;
      for (; tile + D <= te; tile += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          load_tile(min(tile + j + D - 1, tl), kt[(j + D - 1) % D], vt[(j + D - 1) % D]);
          process(kt[j], vt[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < D - 1; ++j)
        if (tile + j < te) process(kt[j], vt[j]);
    }
    if (c < R) {  // lane (c, g) holds O[query c][16 g + 4 rr + j] in o[j][rr]
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) so[pw][c][4 * g + rr] = (f32x4){o[0][rr], o[1][rr], o[2][rr], o[3][rr]};
      if (g == 0) { sm[pw][c] = m_run; sl[pw][c] = l_run; }
    }
    __syncthreads();
    return;
  }

  // ---- suffix wave: one sequence, its private keys (or all of them when it does not share the group's prefix)
  const int cc = lane & 15, rw = lane >> 4;
  const int row = wave / NS, part = wave - row * NS;  // NS waves split a sequence's private tiles
  const int b = b0 + row;
  const bool live = b < a.nseq;
  const int bs = live ? b : a.nseq - 1;
  const KvPrefix pb = a.pre[bs];
  const bool casc = live && ptl0 > 0 && pb.kv == p0.kv && pb.len == p0.len && pb.cap == p0.cap;
  const int pq = a.offset[bs];
  const int tile_hi = (pq + 16) >> 4;
  const int bh = bs * a.H + h;
  const f32x4 q = *(const f32x4 *)(a.Q + ((((size_t)bh * a.QB) * 4 + (cc >> 2)) * 64 + 16 * (cc & 3)) * 4) * 0.125f;
  const float *Kown = a.Kc + (size_t)bh * a.cap * 64;
  const float *Vown = a.Vc + (size_t)bh * a.cap * 64;
  const float *Kpre = Kown, *Vpre = Vown;
  const int ptl = attn_prefix(a, bs, h, Kpre, Vpre);
  const int gs = casc ? ptl0 : 0, ge = live ? tile_hi : 0;
  const int perw = (max(ge - gs, 0) + NS - 1) / NS;
  const int ts = gs + part * perw, te = min(ge, ts + perw);
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  auto load_tile = [&](int tile, KvTile &t) {
    const size_t base = (size_t)tile * 16 * 64 + lane * 4;
    const float *Kb = tile < ptl ? Kpre : Kown, *Vb = tile < ptl ? Vpre : Vown;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.k[i] = __builtin_nontemporal_load((const f32x4 *)(Kb + base + 256 * i));
#pragma unroll
    for (int i = 0; i < 4; ++i) t.v[i] = __builtin_nontemporal_load((const f32x4 *)(Vb + base + 256 * i));
  };
  auto process = [&](int tile, const KvTile &t) {
    float s[4];
    bool ok[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = row16_sum((t.k[i].x * q.x + t.k[i].y * q.y) + (t.k[i].z * q.z + t.k[i].w * q.w));
      ok[i] = tile * 16 + 4 * i + rw <= pq;
      mx = ok[i] ? fmaxf(mx, s[i]) : mx;
    }
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    o *= alpha;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p = ok[i] ? expf(s[i] - m_new) : 0.f;
      ps += p;
      o += t.v[i] * p;
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
  };
  KvTile t[D];
  if (ts < te) {
    const int tl = te - 1;
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_tile(min(ts + j, tl), t[j]);
    int tile = ts;
    for (; tile + D <= te; tile += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load_tile(min(tile + j + D - 1, tl), t[(j + D - 1) % D]);
        process(tile + j, t[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
      if (tile + j < te) process(tile + j, t[j]);
  }
  float M = fmaxf(m_run, __shfl_xor(m_run, 16));
  M = fmaxf(M, __shfl_xor(M, 32));
  {
    const float e = expf(m_run - M);
    l_run *= e;
    o *= e;
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] += __shfl_xor(o[j], 16);
      o[j] += __shfl_xor(o[j], 32);
    }
    m_run = M;
  }
  if (NS > 1 && part > 0 && rw == 0) {
    so[PW + part - 1][row][cc] = o;
    if (cc == 0) { sm[PW + part - 1][row] = m_run; sl[PW + part - 1][row] = l_run; }
  }
  __syncthreads();
  if (rw != 0 || !live || part > 0) return;
  {  // fixed order: own state, then the prefix partials 0 .. PW-1 (shared prefix only), then the other suffix shares
    const int w0 = casc ? 0 : PW;
    float MM = m_run;
#pragma unroll
    for (int w = 0; w < NP; ++w) MM = w >= w0 ? fmaxf(MM, sm[w][row]) : MM;
    const float e0 = expf(m_run - MM);
    float L = l_run * e0;
    f32x4 O = o * e0;
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const float e = w >= w0 ? expf(sm[w][row] - MM) : 0.f;
      L += (w >= w0 ? sl[w][row] : 0.f) * e;
      O += (w >= w0 ? so[w][row][cc] : (f32x4){0.f, 0.f, 0.f, 0.f}) * e;
    }
    o = O; l_run = L;
  }
  const size_t m = (size_t)b * a.Tq;
  *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + (cc >> 2)) * 64 + 16 * (cc & 3) + (m & 15)) * 4) = o * (1.0f / l_run);
}

// merges the key splits: thread (query, 4-wide d group)
static __global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs a) {
  const int bh = blockIdx.x, qb = blockIdx.y;
  const int b = bh / a.H, h = bh - b * a.H;
  const int qi = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int nq = min(16, a.Tq - 16 * qb);
  if (qi >= nq) return;
  const float *pp = a.part + (((size_t)bh * a.QB + qb) * a.splits) * 16 * ATT_PSTRIDE + qi * ATT_PSTRIDE;
  float M = NEG_BIG;
  for (int s = 0; s < a.splits; ++s) M = fmaxf(M, pp[(size_t)s * 16 * ATT_PSTRIDE + 64]);
  float L = 0.f;
  f32x4 O = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.splits; ++s) {
    const float *ps = pp + (size_t)s * 16 * ATT_PSTRIDE;
    const float w = expf(ps[64] - M);
    L += ps[65] * w;
    O += *(const f32x4 *)(ps + 4 * c) * w;
  }
  O = O * (1.0f / L);
  const size_t m = (size_t)b * a.Tq + 16 * qb + qi;
  if (a.h16) *(bf16x4 *)((__bf16 *)a.Y + fmh_off(m, h * 64 + 4 * c, a.YF)) = to_bf16x4(O);
  else *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + (c >> 2)) * 64 + 16 * (c & 3) + (m & 15)) * 4) = O;
}

// ---------------------------------------------------------------------------------------------
// KV cache import / export between the reference layout f32[2, B, T, H, 64] (transformer.py:32-36)
// and the internal K[b][h][slot][64], V[b][h][slot][64].
// ---------------------------------------------------------------------------------------------
static __global__ void kv_import_kernel(const float *src, float *Kc, float *Vc, int B, int srcB, int T, int H, int cap) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [2][B][T][H][16 float4]
  long total = 2L * B * T * H * 16;
  if (i >= total) return;
  int d4 = i & 15;
  long r = i >> 4;
  int h = r % H; r /= H;
  int t = r % T; r /= T;
  int b = r % B; r /= B;
  int which = (int)r;
  int sb = srcB == 1 ? 0 : b;
  f32x4 v = *(const f32x4 *)(src + ((((size_t)which * srcB + sb) * T + t) * H + h) * 64 + d4 * 4);
  float *dst = which ? Vc : Kc;
  *(f32x4 *)(dst + (((size_t)b * H + h) * cap + t) * 64 + d4 * 4) = v;
}

static __global__ void kv_export_kernel(float *dst, const float *Kc, const float *Vc, int B, int T, int H, int cap,
                                 const KvPrefix *pre = nullptr, int layer = 0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = 2L * B * T * H * 16;
  if (i >= total) return;
  int d4 = i & 15;
  long r = i >> 4;
  int h = r % H; r /= H;
  int t = r % T; r /= T;
  int b = r % B; r /= B;
  int which = (int)r;
  const float *src = which ? Vc : Kc;
  f32x4 v;
  if (pre && t < pre[b].len) v = *(const f32x4 *)(pre[b].kv + ((((size_t)(2 * layer + which)) * H + h) * pre[b].cap + t) * 64 + d4 * 4);
  else v = *(const f32x4 *)(src + (((size_t)b * H + h) * cap + t) * 64 + d4 * 4);
  *(f32x4 *)(dst + ((((size_t)which * B + b) * T + t) * H + h) * 64 + d4 * 4) = v;
}

// dst rows <- src rows (src batch 1 broadcasts) of a [L][2][B][H][cap][64] cache block
// one batch-1 state -> row `row` of a batch state: [planes][1][H][src_cap][64] -> [planes][B][H][dst_cap][64], T positions
static __global__ void kv_copy_row_kernel(float *dst, const float *src, int planes, int H, int T, int src_cap, int dst_cap,
                                   int B, int row, int srcB = 1, int src_row = 0, int t0 = 0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (plane, head, t, 16 float4), positions t0 .. t0 + T - 1
  if (i >= (long)planes * H * T * 16) return;
  const int v = i & 15;
  long r = i >> 4;
  const int t = t0 + r % T; r /= T;
  const int h = r % H;
  const int pl = r / H;
  const f32x4 x = *(const f32x4 *)(src + ((((size_t)pl * srcB + src_row) * H + h) * src_cap + t) * 64 + v * 4);
  *(f32x4 *)(dst + ((((size_t)pl * B + row) * H + h) * dst_cap + t) * 64 + v * 4) = x;
}

// dst rows <- src rows (src batch 1 broadcasts) of a [L][2][B][H][cap][64] cache block: the first T positions (a clone needs
// the written part of the cache, not its whole capacity), any two capacities
static __global__ void kv_copy_t_kernel(float *dst, const float *src, int planes, int B, int srcB, int H, int T, int src_cap, int dst_cap,
                                        int t0 = 0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 units over [planes][B][H][T][16], positions t0 .. t0 + T - 1
  if (i >= (long)planes * B * H * T * 16) return;
  const int v = i & 15;
  long r = i >> 4;
  const int t = t0 + r % T; r /= T;
  const int h = r % H; r /= H;
  const int b = r % B;
  const int pl = r / B;
  const int sb = srcB == 1 ? 0 : b;
  *(f32x4 *)(dst + ((((size_t)pl * B + b) * H + h) * dst_cap + t) * 64 + v * 4) =
      *(const f32x4 *)(src + ((((size_t)pl * srcB + sb) * H + h) * src_cap + t) * 64 + v * 4);
}
