// All transformer layers of one FlowLM decode step as ONE launch (reference mimi_transformer.py:39-54 x L with
// transformer.py:135-158, rope.py:28-58 and the linear KV cache of transformer.py:9-19, for one query per sequence).
//
// Same scheme as the flow MLP (ptts_flow.h): a CLUSTER of DF = d_model / 16 workgroups owns one row tile (16
// sequences) through every layer; rows are independent, so clusters never talk to each other.  A layer is five
// phases separated by in-cluster hand-offs (write-through stores + flag, polled by the consumers):
//   A  norm1 (folded) + in_proj      workgroup (head h, quarter q) computes the 16 columns q of head h of Q, K and V;
//                                    RoPE; K / V rows go straight to the cache (sc1), Q to the exchange slot
//   B  attention                     the same workgroup runs the one-query attention of head h for 4 of the 16
//                                    sequences (2 waves per sequence split the keys, merged in LDS)
//   C  out_proj + residual           workgroup j owns column tile j; the residual tile lives in its registers
//   D  norm2 (folded) + linear1 + GELU  workgroup j owns FF/D column tiles
//   E  linear2 + residual            column tile j; the result is the next layer's input
// Per phase a workgroup streams only its own weight tiles (64-256 KB), prefetching the first fragments of the next
// phase while the hand-off completes; activations travel as 1 KiB MFMA operand fragments (FM layout).  Compared with
// five launches per layer this removes ~3.3 us of kernel boundary + ramp per GEMM and keeps ~256 workgroups resident
// and mostly waiting on memory, which leaves the matrix pipes to the codec stream running beside it.
//
// Workgroup = 8 worker waves + 1 coordinator wave (poll, reduce the K-split partials, epilogue, publish).
#pragma once
#include "ptts_flow.h"

struct LmLayerP {  // device-resident table, one entry per layer (built at engine creation)
  const float *wqkv, *qkv_s, *qkv_c;  // packed [3 DF][DF][64][4] with the norm1 gain folded in; fold vectors
  const float *wout;                  // [DF][DF]
  const float *wff1, *ff1_s, *ff1_c;  // [FFF][DF], norm2 folded
  const float *wff2;                  // [DF][FFF]
};

struct LmArgs {
  int MT, M, NG, NCL, DF, FFF, H, L, cap;
  const LmLayerP *layers;
  float *kv;          // [L][2][B][H][cap][64]
  long kv_plane;      // floats per (layer, K|V) plane
  const int *offset;  // position of the query row of every sequence
  const float *freq;  // RoPE frequencies [32]
  float *x;           // residual stream, FM [MT][DF]: input of layer 0, output of the last layer
  float *exch;        // [NG][L][(4 DF + FFF)][256]
  u64 *flags;         // [NG][L][5][DF]
  const int *ctr;
  int *err;
  float ln_eps;
};

#define LM_KPW 8  // k-fragments per worker and chunk: 8 workers x 8 = 64 fragments = one 1024-wide row per chunk

template <int DUMMY>
__global__ __launch_bounds__(FLOW_THREADS) void lm_cluster_kernel(LmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: k ranges and buffer offsets stay scalar
  const int DF = a.DF, FFF = a.FFF, H = a.H;
  const int cl = blockIdx.x / a.DF, j = blockIdx.x % a.DF;  // resident cluster, workgroup in the cluster (same j = same XCD)
  const int hd = j >> 2, qd = j & 3;                          // phase A / B role: head, 16-column quarter of the head
  const bool coord = wave == FLOW_WORKERS;
  const int NFF = FFF / DF;  // linear1 column tiles per workgroup (host: <= 4)
  __shared__ f32x4 red[FLOW_WORKERS][4][64];
  __shared__ float st[FLOW_WORKERS][16][2];
  __shared__ f32x4 so[FLOW_WORKERS][16];
  __shared__ float sm[FLOW_WORKERS], sl[FLOW_WORKERS];
  const u64 ebase = ((u64)(unsigned)(*a.ctr) + 1ull) << 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int SLOT_Q = 0, SLOT_AO = DF, SLOT_X1 = 2 * DF, SLOT_H = 3 * DF, SLOT_X2 = 3 * DF + FFF, LSTRIDE = 4 * DF + FFF;
  bool dead = false;
  const int kf0 = wave * LM_KPW;  // this worker's k-fragments inside a 64-fragment chunk

  // ---- worker helpers.  Every load of a phase is issued before the first MFMA that needs it, and the first weight
  // fragments of the NEXT phase are requested before the hand-off is polled (they do not depend on it).
  const unsigned lane16 = (unsigned)lane * 16u;
  auto ldw = [&](const float *wt, int KF, int kbase, f32x4 *w) {  // 8 weight fragments kbase + kf0 .. of column tile wt
#ifdef LM_DEBUG_SAMEW  // timing experiment: every weight tile is the same (cache-resident) 64 KB: results invalid
    wt = a.layers[0].wqkv;
#endif
    const __amdgpu_buffer_rsrc_t rw = flow_rsrc(wt);
#pragma unroll
    for (int u = 0; u < LM_KPW; ++u) {
      const int kf = kbase + kf0 + u;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)lane16, min(kf, KF - 1) * 1024, 0));
      w[u] = kf < KF ? v : zero4;
    }
  };
  auto mm = [&](f32x4 &acc, const f32x4 *w, const f32x4 *x) {
#pragma unroll
    for (int u = 0; u < LM_KPW; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][q], x[u][q], acc, 0, 0, 0);
  };
  auto row_stats = [&](const f32x4 *x) {  // partial (sum, sum of squares) of this worker's k range, rows on lanes 0..15
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < LM_KPW; ++u) {
      s1 += (x[u].x + x[u].y) + (x[u].z + x[u].w);
      s2 += (x[u].x * x[u].x + x[u].y * x[u].y) + (x[u].z * x[u].z + x[u].w * x[u].w);
    }
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    if (lane < 16) { st[wave][lane][0] = s1; st[wave][lane][1] = s2; }
  };

  // Coordinator and workers run two separate copies of the phase loop (same number of workgroup barriers per phase:
  // A 2, B 3, C 2, D 2, E 2, plus one per row group), so that neither role's registers are kept alive through the
  // other role's code.
  if (coord) {
    for (int grp = cl; grp < a.NG; grp += a.NCL) {
      const int mt = grp;
      float *ex = a.exch + (size_t)grp * a.L * LSTRIDE * 256;
      u64 *fl = a.flags + (size_t)grp * a.L * 5 * DF;
      const __amdgpu_buffer_rsrc_t rs = flow_rsrc(ex);
      __syncthreads();
      f32x4 xres = *(const f32x4 *)(a.x + (((size_t)mt * DF + j) * 64 + lane) * 4);  // residual stream tile j
      // reduce the partials of tile t (fixed order), optional LayerNorm-fold finish
      auto reduce_tile = [&](int t, const float *ln_s, const float *ln_c, int nt_global, int KF) {
        f32x4 s = red[0][t][lane];
#pragma unroll
        for (int q = 1; q < FLOW_WORKERS; ++q) s += red[q][t][lane];
        if (ln_s) {
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int q = 0; q < FLOW_WORKERS; ++q) { t0 += st[q][lane & 15][0]; t1 += st[q][lane & 15][1]; }
          const float invK = 1.0f / (float)(KF * 16);
          const float mu = t0 * invK;
          const float rsd = 1.0f / sqrtf(fmaxf(t1 * invK - mu * mu, 0.f) + a.ln_eps);
          const int n0 = 16 * nt_global + 4 * (lane >> 4);
          s = (s - *(const f32x4 *)(ln_s + n0) * mu) * rsd + *(const f32x4 *)(ln_c + n0);
        }
        return s;
      };
      auto publish = [&](int l, int phase) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(fl + ((size_t)l * 5 + phase) * DF + j, ebase | (u64)(l * 5 + phase + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto wait_phase = [&](int l, int phase) {  // all DF producers of (layer, phase)
        if (!dead && !flow_wait(fl + ((size_t)l * 5 + phase) * DF, DF, ebase | (u64)(l * 5 + phase + 1), lane, a.err)) dead = true;
      };
      for (int l = 0; l < a.L; ++l) {
        const LmLayerP P = a.layers[l];
        const unsigned lbase = (unsigned)l * LSTRIDE;
        float *Kc = a.kv + (size_t)(2 * l) * a.kv_plane, *Vc = Kc + a.kv_plane;
        // ---- A
        if (l > 0) wait_phase(l - 1, 4);
        __syncthreads();
        __syncthreads();
        {
          const int m = 16 * mt + (lane & 15), g = lane >> 4;
          const int pos = a.offset[min(m, a.M - 1)];
          const int d = 16 * qd + 4 * g;  // column inside the head
          f32x4 cs;
          {
            const float f0 = a.freq[d >> 1], f1 = a.freq[(d >> 1) + 1];
            float s0, c0, s1, c1;
            sincosf(f0 * (float)pos, &s0, &c0);
            sincosf(f1 * (float)pos, &s1, &c1);
            cs = (f32x4){c0, s0, c1, s1};
          }
          auto rope = [&](f32x4 v) {  // interleaved pairs (reference rope.py:28-58)
            f32x4 o;
            o.x = v.x * cs.x - v.y * cs.y;
            o.y = v.x * cs.y + v.y * cs.x;
            o.z = v.z * cs.z - v.w * cs.w;
            o.w = v.z * cs.w + v.w * cs.z;
            return o;
          };
          const f32x4 qv = rope(reduce_tile(0, P.qkv_s, P.qkv_c, j, DF));
          const f32x4 kv = rope(reduce_tile(1, P.qkv_s, P.qkv_c, DF + j, DF));
          const f32x4 vv = reduce_tile(2, P.qkv_s, P.qkv_c, 2 * DF + j, DF);
          flow_st_sc1(rs, ((lbase + SLOT_Q + j) * 64u + lane) * 16u, qv);
          if (m < a.M) {
            const size_t o = (((size_t)m * H + hd) * a.cap + pos) * 64 + d;
            const __amdgpu_buffer_rsrc_t rk = flow_rsrc(Kc), rv = flow_rsrc(Vc);
            flow_st_sc1(rk, (unsigned)(o * 4), kv);
            flow_st_sc1(rv, (unsigned)(o * 4), vv);
          }
          publish(l, 0);
        }
        // ---- B
        wait_phase(l, 0);
        __syncthreads();
        __syncthreads();
        __syncthreads();
        publish(l, 1);
        // ---- C
        wait_phase(l, 1);
        __syncthreads();
        __syncthreads();
        xres = xres + reduce_tile(0, nullptr, nullptr, j, DF);
        flow_st_sc1(rs, ((lbase + SLOT_X1 + j) * 64u + lane) * 16u, xres);
        publish(l, 2);
        // ---- D
        wait_phase(l, 2);
        __syncthreads();
        __syncthreads();
        for (int t = 0; t < NFF; ++t) {
          const f32x4 v = act4(reduce_tile(t, P.ff1_s, P.ff1_c, NFF * j + t, DF), ACT_GELU);
          flow_st_sc1(rs, ((lbase + SLOT_H + NFF * j + t) * 64u + lane) * 16u, v);
        }
        publish(l, 3);
        // ---- E
        wait_phase(l, 3);
        __syncthreads();
        __syncthreads();
        xres = xres + reduce_tile(0, nullptr, nullptr, j, FFF);
        if (l + 1 < a.L) {
          flow_st_sc1(rs, ((lbase + SLOT_X2 + j) * 64u + lane) * 16u, xres);
          publish(l, 4);
        } else {
          *(f32x4 *)(a.x + (((size_t)mt * DF + j) * 64 + lane) * 4) = xres;  // read by the next launch (head GEMM)
        }
      }
    }
    return;
  }

  // ======================================= workers =======================================
  for (int grp = cl; grp < a.NG; grp += a.NCL) {
    const int mt = grp;
    float *ex = a.exch + (size_t)grp * a.L * LSTRIDE * 256;
    const __amdgpu_buffer_rsrc_t rs = flow_rsrc(ex);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rx = flow_rsrc(a.x + (size_t)mt * DF * 256);
    auto ldx = [&](unsigned slot_base, bool from_x, int KF, int kbase, f32x4 *x) {  // 8 operand fragments
#pragma unroll
      for (int u = 0; u < LM_KPW; ++u) {
        const int kf = kbase + kf0 + u, kc = min(kf, KF - 1);
        f32x4 v;
        if (from_x) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lane16, kc * 1024, 0));
        else v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane16, (int)(slot_base + (unsigned)kc) * 1024, 16));
        x[u] = kf < KF ? v : zero4;
      }
    };
    f32x4 wA[LM_KPW], wB[LM_KPW];  // weight fragments: loaded ahead of the phase that uses them
    ldw(a.layers[0].wqkv + (size_t)j * DF * 256, DF, 0, wA);

    for (int l = 0; l < a.L; ++l) {
      const LmLayerP P = a.layers[l];
      const unsigned lbase = (unsigned)l * LSTRIDE;
      float *Kc = a.kv + (size_t)(2 * l) * a.kv_plane, *Vc = Kc + a.kv_plane;
      // ================= A: norm1 + in_proj (tile 0 of Q is in wA)
      __syncthreads();
      {
        f32x4 x[LM_KPW], a0 = zero4, a1 = zero4, a2 = zero4;
        ldx(lbase - LSTRIDE + SLOT_X2, l == 0, DF, 0, x);
        ldw(P.wqkv + (size_t)(DF + j) * DF * 256, DF, 0, wB);
        mm(a0, wA, x);
        row_stats(x);
        ldw(P.wqkv + (size_t)(2 * DF + j) * DF * 256, DF, 0, wA);
        mm(a1, wB, x);
        mm(a2, wA, x);
        red[wave][0][lane] = a0; red[wave][1][lane] = a1; red[wave][2][lane] = a2;
      }
      __syncthreads();
      // ================= B: attention of head hd for sequences 4 qd .. 4 qd + 3 of this row tile
      {
        const int sq = wave >> 1, half = wave & 1;
        const int ml = 4 * qd + sq, m = 16 * mt + ml;
        const int c = lane & 15, g = lane >> 4;
        const bool live = m < a.M;
        const int pq = a.offset[min(m, a.M - 1)];
        const int tile_hi = (pq + 16) >> 4;
        const int per = (tile_hi + 1) >> 1;
        const int ts = half * per, te = min(tile_hi, ts + per), tl = te - 1;
        const __amdgpu_buffer_rsrc_t rk = flow_rsrc(Kc), rv = flow_rsrc(Vc);
        const unsigned kvb = (unsigned)((((size_t)min(m, a.M - 1) * H + hd) * a.cap) * 256);  // wave-uniform
        const unsigned kvo = (unsigned)c * 256u + (unsigned)g * 16u, vvo = (unsigned)g * 1024u + (unsigned)c * 16u;
        auto load_tile = [&](int tile, f32x4 *kk, f32x4 *vv) {
          const int p0 = (int)(kvb + (unsigned)tile * 4096u);
#pragma unroll
          for (int df = 0; df < 4; ++df)
            kk[df] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(kvo + 64u * df), p0, 16));
#pragma unroll
          for (int r = 0; r < 4; ++r)
            vv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(vvo + 256u * r), p0, 16));
        };
        f32x4 k0[4], v0[4], k1[4], v1[4];
        __syncthreads();
        f32x4 o = zero4;
        float m_run = NEG_BIG, l_run = 0.f;
        if (live && ts < te) {
          f32x4 qv[4];
#pragma unroll
          for (int df = 0; df < 4; ++df)
            qv[df] = flow_ld_sc1(rs, ((lbase + SLOT_Q + 4 * hd + df) * 64u + 16 * g + ml) * 16u) * 0.125f;  // 1/sqrt(64)
          auto process = [&](int tile, const f32x4 *kk, const f32x4 *vv) {
            float s = 0.f;
#pragma unroll
            for (int df = 0; df < 4; ++df)
              s += (kk[df].x * qv[df].x + kk[df].y * qv[df].y) + (kk[df].z * qv[df].z + kk[df].w * qv[df].w);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const bool ok = tile * 16 + c <= pq;
            float mx = ok ? s : NEG_BIG;
#pragma unroll
            for (int dd = 1; dd < 16; dd <<= 1) mx = fmaxf(mx, __shfl_xor(mx, dd));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);
            const float p = ok ? expf(s - m_new) : 0.f;
            float ps = p;
#pragma unroll
            for (int dd = 1; dd < 16; dd <<= 1) ps += __shfl_xor(ps, dd);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            o *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) o += vv[r] * __shfl(p, 4 * g + r);
          };
          load_tile(ts, k0, v0);
          load_tile(min(ts + 1, tl), k1, v1);
          int tile = ts;
          for (; tile + 2 <= te; tile += 2) {  // two register tiles: the reload of one flies during the math of the other
            process(tile, k0, v0);
            load_tile(min(tile + 2, tl), k0, v0);
            process(tile + 1, k1, v1);
            load_tile(min(tile + 3, tl), k1, v1);
          }
          if (tile < te) process(tile, k0, v0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] += __shfl_xor(o[q], 16);
            o[q] += __shfl_xor(o[q], 32);
          }
        }
        if (g == 0) so[wave][c] = o;
        if (lane == 0) { sm[wave] = m_run; sl[wave] = l_run; }
      }
      __syncthreads();  // M: the two halves of every sequence are in LDS
      ldw(P.wout + (size_t)j * DF * 256, DF, 0, wA);  // phase C's weights fly during the hand-off
      if ((wave & 1) == 0 && (lane >> 4) == 0) {
        const int sq = wave >> 1, ml = 4 * qd + sq, c = lane & 15;
        const float M0 = fmaxf(sm[wave], sm[wave + 1]);
        const float e0 = expf(sm[wave] - M0), e1 = expf(sm[wave + 1] - M0);
        const float L0 = sl[wave] * e0 + sl[wave + 1] * e1;
        f32x4 O = so[wave][c] * e0 + so[wave + 1][c] * e1;
        O = L0 > 0.f ? O * (1.0f / L0) : zero4;
        // column n = hd*64 + 4c + q of row ml -> fragment 4 hd + c/4, k-group c%4
        flow_st_sc1(rs, ((lbase + SLOT_AO + 4 * hd + (c >> 2)) * 64u + 16 * (c & 3) + ml) * 16u, O);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the barrier
      __syncthreads();
      // ================= C: out_proj + residual
      __syncthreads();
      {
        f32x4 x[LM_KPW], a0 = zero4;
        ldx(lbase + SLOT_AO, false, DF, 0, x);
        ldw(P.wff1 + (size_t)(NFF * j) * DF * 256, DF, 0, wB);  // phase D, tile 0
        mm(a0, wA, x);
        red[wave][0][lane] = a0;
      }
      __syncthreads();
      // ================= D: norm2 + linear1 + GELU (tile 0 is in wB)
      __syncthreads();
      {
        f32x4 x[LM_KPW], a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
        ldx(lbase + SLOT_X1, false, DF, 0, x);
        const float *w1 = P.wff1 + (size_t)(NFF * j) * DF * 256;
        if (NFF > 1) ldw(w1 + (size_t)DF * 256, DF, 0, wA);
        mm(a0, wB, x);
        row_stats(x);
        if (NFF > 2) ldw(w1 + (size_t)2 * DF * 256, DF, 0, wB);
        if (NFF > 1) mm(a1, wA, x);
        if (NFF > 3) ldw(w1 + (size_t)3 * DF * 256, DF, 0, wA);
        if (NFF > 2) mm(a2, wB, x);
        if (NFF > 3) mm(a3, wA, x);
        red[wave][0][lane] = a0; red[wave][1][lane] = a1; red[wave][2][lane] = a2; red[wave][3][lane] = a3;
        ldw(P.wff2 + (size_t)j * FFF * 256, FFF, 0, wA);  // phase E, chunk 0
      }
      __syncthreads();
      // ================= E: linear2 + residual (K = FFF: chunks of 64 fragments, double-buffered)
      __syncthreads();
      {
        const float *w2 = P.wff2 + (size_t)j * FFF * 256;
        const int nch = (FFF + 8 * LM_KPW - 1) / (8 * LM_KPW);
        f32x4 xA[LM_KPW], xB[LM_KPW], a0 = zero4;
        ldx(lbase + SLOT_H, false, FFF, 0, xA);
        int ch = 0;
        for (; ch + 2 <= nch; ch += 2) {
          ldx(lbase + SLOT_H, false, FFF, (ch + 1) * 8 * LM_KPW, xB);
          ldw(w2, FFF, (ch + 1) * 8 * LM_KPW, wB);
          mm(a0, wA, xA);
          ldx(lbase + SLOT_H, false, FFF, min(ch + 2, nch - 1) * 8 * LM_KPW, xA);
          ldw(w2, FFF, min(ch + 2, nch - 1) * 8 * LM_KPW, wA);
          mm(a0, wB, xB);
        }
        if (ch < nch) mm(a0, wA, xA);
        red[wave][0][lane] = a0;
        if (l + 1 < a.L) ldw(a.layers[l + 1].wqkv + (size_t)j * DF * 256, DF, 0, wA);  // next layer, phase A, tile 0
      }
      __syncthreads();
    }
  }
}
