// The flow MLP of one FlowLM step as ONE launch (reference mlp.py:188-215 inside the LSD loop of flow_lm.py:19-40).
//
// The chain  input_proj -> depth x [AdaLN-LN -> Linear -> SiLU -> Linear -> gate + residual] -> final layer -> Euler
// is 2 * depth + 2 dependent GEMMs on 512-wide rows: 1 MB of weights and a few hundred kFLOP per row tile each, i.e.
// nothing but dependent-launch latency when every GEMM is its own kernel (16 launches, 193 us per step at batch 64).
// Here a CLUSTER of FDF = flow_dim / 16 workgroups owns RT row tiles (16 rows each) for the whole chain; workgroup j
// owns output columns 16 j .. 16 j + 15 of every layer.  Rows are independent, so clusters never talk to each other
// and a batch is covered by ceil(MT / RT) clusters (batch 64: 4 clusters x 32 workgroups = 128 CUs, the rest of the
// chip stays free for the codec stream).  Between two layers the cluster exchanges the layer output (RT x 512 floats)
// through L2: every workgroup publishes its 16 columns as ONE 1 KiB fragment per row tile - lane for lane the MFMA
// operand fragment kf = j of the next layer (FM layout, see ptts_kernels.h) - with write-through (`sc1`) stores, drains
// them and raises a flag; consumers poll the flags of their cluster and read the fragments with `sc1` loads
// (MI355X_MICROARCH.md, "Valid forms": every payload byte stored sc1 and drained before the flag, every load of it a
// buffer_load sc1 behind the matched poll and a workgroup barrier).  Every (phase, producer) has its own slot and flag,
// so nothing is overwritten inside a launch; the flag value carries the state's step counter, so a replayed graph
// needs no memset node and stale flags of earlier steps never match.
//
// Workgroup = 1 coordinator wave + 8 worker waves.  Workers split K eight ways: each loads its k-fragments of the
// operand (and of the AdaLN shift / scale), the LayerNorm statistics meet in LDS, the partial accumulators too.  The
// coordinator polls, sums the partials, runs the epilogue, publishes and holds the residual stream tile in registers;
// it issues no weight loads, so its polls and drains never queue behind an HBM miss.  Workers fetch the NEXT phase's
// weight fragments right after handing over their partials: the 32 KB a workgroup needs per layer are in flight
// while the hand-off completes.
#pragma once
#include "ptts_kernels.h"

#define FLOW_MAX_DEPTH 12
#define FLOW_WORKERS 8
#define FLOW_THREADS (64 * (FLOW_WORKERS + 1))
#define FLOW_SPIN_LIMIT 400000u  // ~0.1-0.3 s per phase before a workgroup gives up (never reached when all blocks run)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct FlowArgs {
  int MT, M, NG, NCL, FDF, LF, AF, depth, steps, ldim;  // NG row groups of RT tiles, served by NCL resident clusters
  const float *w_in, *b_in;  // input_proj, packed [FDF][LF][64][4]
  const float *w_l0[FLOW_MAX_DEPTH], *b_l0[FLOW_MAX_DEPTH], *w_l2[FLOW_MAX_DEPTH], *b_l2[FLOW_MAX_DEPTH];
  const float *ln_w[FLOW_MAX_DEPTH], *ln_b[FLOW_MAX_DEPTH];
  const float *w_fin, *b_fin;  // final linear, packed [LF][FDF][64][4]
  const float *mod;            // AdaLN modulations of every LSD step, FM [steps][MT][AF]
  long mod_step;               // floats per LSD step in `mod`
  const float *latfm;          // FM [MT][LF]: start point of the LSD integration (noise), written by prep_lm_kernel
  float *lat, *lat_out1, *lat_out2;  // plain [M][ldim] outputs (state copy, next step's input, caller's buffer)
  float inv_steps;
  float *exch;         // [NG][steps * (2 depth + 2)][RT][FDF][256]
  u64 *flags;          // [NG][steps * (2 depth + 2)][FDF]
  const int *ctr;      // the state's step counter (incremented by step_tail_kernel after this launch)
  int *err;            // set to 1 when a poll gave up
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t flow_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ f32x4 flow_ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));  // aux 16 = sc1
}
__device__ __forceinline__ void flow_st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 16);
}

// one wave polls the flags of `np` producers (np <= 64) until all carry `epoch`; bounded
__device__ __forceinline__ bool flow_wait(const u64 *f, int np, u64 epoch, int lane, int *err) {
  for (unsigned spins = 0;; ++spins) {
    const u64 v = lane < np ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
    if (__all(v == epoch)) return true;
    if (spins > FLOW_SPIN_LIMIT) {
      if (lane == 0) atomicExch(err, 1);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// RT = row tiles per cluster (the host uses 1: one cluster per 16 rows keeps the worker waves within the 168
// registers a 9-wave workgroup leaves per lane), KPW = k-fragments per worker wave (>= ceil(max(FDF, LF) / 8))
template <int RT, int KPW>
__global__ __launch_bounds__(FLOW_THREADS) void flow_cluster_kernel(FlowArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int FDF = a.FDF, LF = a.LF;
  // workgroups with the same column tile j (they stream the SAME weight tiles) sit 8 apart in blockIdx, i.e. on one XCD
  // under round-robin placement (speed only): the clusters' weight re-reads are then L2 hits instead of fabric reads
  const int cl = blockIdx.x / a.FDF, j = blockIdx.x % a.FDF;  // resident cluster, column tile
  const bool coord = wave == FLOW_WORKERS;
  const int PPS = 2 * a.depth + 2, NPH = a.steps * PPS;
  __shared__ f32x4 red[FLOW_WORKERS][RT][64];
  __shared__ float st[FLOW_WORKERS][RT][16][2];
  const u64 ebase = ((u64)(unsigned)(*a.ctr) + 1ull) << 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;  // coordinator only: a poll gave up; finish without waiting (outputs are garbage, *err is set)
  // a resident cluster serves row groups cl, cl + NCL, ..: at most NCL * FDF workgroups wait for each other, however
  // large the batch is (a 9-wave workgroup of this kernel fills a CU: residency is one workgroup per CU)
  for (int grp = cl; grp < a.NG; grp += a.NCL) {
  const int mt0 = grp * RT;
  float *ex = a.exch + (size_t)grp * NPH * RT * FDF * 256;
  u64 *fl = a.flags + (size_t)grp * NPH * FDF;
  const __amdgpu_buffer_rsrc_t rs = flow_rsrc(ex);
  __syncthreads();  // LDS of the previous group is free

  int mtc[RT];  // clamped row tiles (a partially filled last cluster computes duplicates and stores nothing for them)
#pragma unroll
  for (int t = 0; t < RT; ++t) mtc[t] = min(mt0 + t, a.MT - 1);

  // phase p of the launch: kind 0 = input_proj, 1 = block linear 0 (AdaLN-LN on load, SiLU), 2 = block linear 2
  // (gate + residual), 3 = final layer (no-affine LN + modulation on load, Euler update)
  auto kind_of = [&](int ph) { return ph == 0 ? 0 : ph == PPS - 1 ? 3 : 1 + ((ph - 1) & 1); };
  auto wtile = [&](int ph, int &KF, bool &use) -> const float * {
    const int k = kind_of(ph), r = (ph - 1) >> 1;
    use = true;
    if (k == 0) { KF = LF; return a.w_in + (size_t)j * LF * 256; }
    KF = FDF;
    if (k == 1) return a.w_l0[r] + (size_t)j * FDF * 256;
    if (k == 2) return a.w_l2[r] + (size_t)j * FDF * 256;
    use = j < LF;
    return a.w_fin + (size_t)(use ? j : 0) * FDF * 256;
  };
  auto bias_of = [&](int ph) -> const float * {
    const int k = kind_of(ph), r = (ph - 1) >> 1;
    return (k == 0 ? a.b_in : k == 1 ? a.b_l0[r] : k == 2 ? a.b_l2[r] : a.b_fin) + (k == 3 && j >= LF ? 0 : 16 * j) + 4 * (lane >> 4);
  };

  // ---- worker state: weight fragments of the current phase
  f32x4 w[KPW];
  auto load_w = [&](int ph) {
    int KF;
    bool use;
    const float *Wt = wtile(ph, KF, use);
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
      const int kf = wave * KPW + u;
      // plain loads: the clusters of a batch read the same tiles, L2 / Infinity Cache serve the re-reads
      const f32x4 v = *(const f32x4 *)(Wt + (size_t)min(kf, KF - 1) * 256 + lane * 4);
      w[u] = (use && kf < KF) ? v : zero4;
    }
  };
  // ---- coordinator state: residual stream tile, latent tile, epilogue operands of the current and the next phase
  f32x4 xres[RT], latv[RT], e_bias = zero4, e_gate[RT], n_bias = zero4, n_gate[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) xres[t] = latv[t] = e_gate[t] = n_gate[t] = zero4;
  auto load_epi = [&](int p, f32x4 &bias, f32x4 *gate) {  // bias (+ gate tiles) of phase p
    const int i = p / PPS, ph = p - i * PPS;
    bias = *(const f32x4 *)bias_of(ph);
    if (kind_of(ph) == 2) {
      const int r = (ph - 1) >> 1;
#pragma unroll
      for (int t = 0; t < RT; ++t)
        gate[t] = *(const f32x4 *)(a.mod + (size_t)i * a.mod_step + (((size_t)mtc[t] * a.AF + r * 3 * FDF + 2 * FDF + j) * 64 + lane) * 4);
    }
  };
  if (coord) {
    load_epi(0, e_bias, e_gate);
    if (j < LF) {
#pragma unroll
      for (int t = 0; t < RT; ++t) latv[t] = *(const f32x4 *)(a.latfm + (((size_t)mtc[t] * LF + j) * 64 + lane) * 4);
    }
  } else {
    load_w(0);
  }

  for (int p = 0; p < NPH; ++p) {
    const int i = p / PPS, ph = p - i * PPS;
    const int kind = kind_of(ph);
    const int KF = kind == 0 ? LF : FDF;
    const bool lnmod = kind == 1 || kind == 3;
    // ---- A: the inputs of phase p are published (slot p - 1)
    if (coord && p > 0 && !dead) {
      const int np = ph == 0 ? LF : FDF;
      if (!flow_wait(fl + (size_t)(p - 1) * FDF, np, ebase | (u64)p, lane, a.err)) dead = true;
    }
    __syncthreads();
    f32x4 x[RT][KPW], ma[RT][KPW], mb[RT][KPW];  // operand fragments; modulation as x' = LN0(x) * ma + mb
    const int r = (ph - 1) >> 1;
    if (!coord) {
      if (p == 0) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int u = 0; u < KPW; ++u) {
            const int kf = wave * KPW + u;
            const f32x4 v = *(const f32x4 *)(a.latfm + (((size_t)mtc[t] * LF + min(kf, KF - 1)) * 64 + lane) * 4);
            x[t][u] = kf < KF ? v : zero4;
          }
      } else {
        const unsigned sbase = (unsigned)(p - 1) * RT * FDF * 1024u;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int u = 0; u < KPW; ++u) {
            const int kf = wave * KPW + u;
            const f32x4 v = flow_ld_sc1(rs, sbase + ((unsigned)(t * FDF + min(kf, KF - 1)) * 64u + lane) * 16u);
            x[t][u] = kf < KF ? v : zero4;
          }
      }
      if (lnmod) {
        // AdaLN-modulated LayerNorm on load (reference mlp.py:107-109, 127-129): statistics over the whole row
        const float *mbase = a.mod + (size_t)i * a.mod_step;
        const int sh0 = kind == 1 ? r * 3 * FDF : a.depth * 3 * FDF;
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
          const int kf = min(wave * KPW + u, KF - 1);
          f32x4 lw = {1.f, 1.f, 1.f, 1.f}, lb = zero4;
          if (kind == 1) {
            lw = *(const f32x4 *)(a.ln_w[r] + 16 * kf + 4 * (lane >> 4));
            lb = *(const f32x4 *)(a.ln_b[r] + 16 * kf + 4 * (lane >> 4));
          }
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const float *mp = mbase + (((size_t)mtc[t] * a.AF + sh0 + kf) * 64 + lane) * 4;
            const f32x4 shv = *(const f32x4 *)mp, scv = 1.0f + *(const f32x4 *)(mp + (size_t)FDF * 256);
            // (n * w + b) * (1 + scale) + shift  ==  n * [w (1 + scale)] + [b (1 + scale) + shift]
            ma[t][u] = lw * scv;
            mb[t][u] = lb * scv + shv;
          }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int u = 0; u < KPW; ++u) {
            const f32x4 v = x[t][u];
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
          s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
          if (lane < 16) { st[wave][t][lane][0] = s1; st[wave][t][lane][1] = s2; }
        }
      }
    } else if (p + 1 < NPH) {
      load_epi(p + 1, n_bias, n_gate);  // lands while the workers compute
    }
    if (lnmod) __syncthreads();  // B: row statistics of all workers are in LDS
    if (!coord) {
      if (lnmod) {
        const float invK = 1.0f / (float)(KF * 16);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int q = 0; q < FLOW_WORKERS; ++q) { t0 += st[q][t][lane & 15][0]; t1 += st[q][t][lane & 15][1]; }
          const float mu = t0 * invK;
          const float rsd = 1.0f / sqrtf(fmaxf(t1 * invK - mu * mu, 0.f) + 1e-6f);  // flow-MLP LayerNorm eps (mlp.py:95)
#pragma unroll
          for (int u = 0; u < KPW; ++u) {
            const f32x4 v = ((x[t][u] - mu) * rsd) * ma[t][u] + mb[t][u];
            x[t][u] = (wave * KPW + u) < KF ? v : zero4;
          }
        }
      }
      f32x4 acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = zero4;
#pragma unroll
      for (int u = 0; u < KPW; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][c], x[t][u][c], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) red[wave][t][lane] = acc[t];
    }
    __syncthreads();  // C: partial accumulators are in LDS
    if (!coord) {
      if (p + 1 < NPH) load_w((p + 1) % PPS);  // in flight while the hand-off completes
      continue;
    }
    // ---- coordinator: reduce (fixed order), epilogue, publish
    const bool last_step = i == a.steps - 1;
    const bool publish = p + 1 < NPH && (kind != 3 || j < LF);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      f32x4 s = red[0][t][lane];
#pragma unroll
      for (int q = 1; q < FLOW_WORKERS; ++q) s += red[q][t][lane];
      s += e_bias;
      f32x4 out;
      if (kind == 0) { xres[t] = s; out = s; }
      else if (kind == 1) out = act4(s, ACT_SILU);
      else if (kind == 2) { xres[t] = xres[t] + e_gate[t] * s; out = xres[t]; }
      else {
        // Euler update of lsd_decode: current += flow_dir / num_steps (reference flow_lm.py:39)
        latv[t] = latv[t] + s * a.inv_steps;
        out = latv[t];
        const int m = 16 * (mt0 + t) + (lane & 15);
        if (last_step && j < LF && mt0 + t < a.MT && m < a.M) {
          const size_t o = (size_t)m * a.ldim + 16 * j + 4 * (lane >> 4);
          *(f32x4 *)(a.lat + o) = out;
          if (a.lat_out1) *(f32x4 *)(a.lat_out1 + o) = out;
          if (a.lat_out2) *(f32x4 *)(a.lat_out2 + o) = out;
        }
      }
      if (publish) flow_st_sc1(rs, ((unsigned)((p * RT + t) * FDF + j) * 64u + lane) * 16u, out);
    }
    if (publish) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left the CU
      if (lane == 0) __hip_atomic_store(fl + (size_t)p * FDF + j, ebase | (u64)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    e_bias = n_bias;
#pragma unroll
    for (int t = 0; t < RT; ++t) e_gate[t] = n_gate[t];
  }
  }  // row groups
}
