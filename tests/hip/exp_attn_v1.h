// The round-1 MFMA attention kernel (16 queries per wave; softmax statistics and accumulator rescale through
// ds_bpermute shuffles), kept test-only as the A/B baseline of tests/hip/sweep_attn.hip.
#pragma once

__device__ __forceinline__ void attn_store_out_v1(const AttnArgs &a, int b, int h, int qb, int nq, int lane, f32x4 o[4],
                                               float linv_for_row[4]) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = 4 * g + r;
    if (qi < nq) {
      const size_t m = (size_t)b * a.Tq + 16 * qb + qi;
      f32x4 v;
      v.x = o[0][r] * linv_for_row[r];
      v.y = o[1][r] * linv_for_row[r];
      v.z = o[2][r] * linv_for_row[r];
      v.w = o[3][r] * linv_for_row[r];
      // column n = h*64 + 4c + j -> fragment 4h + c/4, k-group c%4
      if (a.h16) *(bf16x4 *)((__bf16 *)a.Y + fmh_off(m, h * 64 + 4 * c, a.YF)) = to_bf16x4(v);
      else *(f32x4 *)(a.Y + (((m >> 4) * a.YF + 4 * h + (c >> 2)) * 64 + 16 * (c & 3) + (m & 15)) * 4) = v;
    }
  }
}

__global__ __launch_bounds__(64) void attn_kernel_v1(AttnArgs a) {
  if constexpr (PTTS_ABLATE & 128) return;
  const int bh = blockIdx.x, qb = blockIdx.y, sp = blockIdx.z;
  const int b = bh / a.H, h = bh - b * a.H;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const int off = a.offset[b];
  const int q0 = off + 16 * qb;
  const int nq = min(16, a.Tq - 16 * qb);
  const int klo = a.ctx > 0 ? max(0, q0 - a.ctx + 1) : 0;
  const int khi = q0 + nq;
  const int tile_lo = klo >> 4, tile_hi = (khi + 15) >> 4;
  const int per = (tile_hi - tile_lo + a.splits - 1) / a.splits;
  const int ts = tile_lo + sp * per;
  const int te = min(tile_hi, ts + per);

  f32x4 qf[4];
#pragma unroll
  for (int df = 0; df < 4; ++df)
    qf[df] = *(const f32x4 *)(a.Q + ((((size_t)bh * a.QB + qb) * 4 + df) * 64 + lane) * 4) * 0.125f;  // 1/sqrt(64)

  f32x4 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const float *Kb = a.Kc + (size_t)bh * a.cap * 64;
  const float *Vb = a.Vc + (size_t)bh * a.cap * 64;
  const int pq = q0 + c;

  auto load_tile = [&](int tile, f32x4 *kk, f32x4 *vv) {
    const int p0 = tile * 16;
    const int slot0 = a.ring ? (p0 % a.ring) : p0;
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
    f32x4 sp[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) sp[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
      for (int df = 0; df < 4; ++df)
        sp[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf4[df][cidx], qf[df][cidx], sp[df], 0, 0, 0);
    const f32x4 s = (sp[0] + sp[1]) + (sp[2] + sp[3]);
    // s[r] = score(key p0 + 4g + r, query c)
    bool ok[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pk = p0 + 4 * g + r;
      ok[r] = (c < nq) && (pk <= pq) && (a.ctx <= 0 || pq - pk < a.ctx);
      mx = ok[r] ? fmaxf(mx, s[r]) : mx;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    f32x4 p;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = ok[r] ? expf(s[r] - m_new) : 0.f;
      ps += p[r];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // rescale the accumulator rows (query 4g + r lives in lane 4g + r of the score layout)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, 4 * g + r);
      o[0][r] *= ar; o[1][r] *= ar; o[2][r] *= ar; o[3][r] *= ar;
    }
    // O[query][d = 4c' + j] += sum_key P[query][key] V[key][4c' + j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vf4[r].x, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vf4[r].y, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vf4[r].z, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vf4[r].w, o[3], 0, 0, 0);
    }
  };
  // Three rotating register tiles: the K/V of the next TWO key tiles (16 KB per wave) are in flight while a tile's
  // scores, softmax and P.V run.  A decode step streams the whole cache once with ~4 waves per CU, so bytes in
  // flight per wave are what sets the achieved HBM rate.
  // The prefetches are UNCONDITIONAL (tile index clamped to the last tile, whose lines are then L1/L2 hits): a
  // load behind a branch makes the compiler's s_waitcnt insertion merge the two paths conservatively and wait for
  // the newest loads as well, which silently serialises the whole pipeline (seen in the ISA: vmcnt(5)..vmcnt(0)
  // in front of the first MFMAs of a tile).
  f32x4 k0[4], v0[4], k1[4], v1[4], k2[4], v2[4];
  if (ts < te) {
    const int tl = te - 1;
    load_tile(ts, k0, v0);
    load_tile(min(ts + 1, tl), k1, v1);
    int tile = ts;
    // whole groups of three tiles: one back-edge, no exits from inside the body (every extra control-flow join
    // makes the wait counts more conservative)
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

  if (a.splits == 1) {
    float linv[4];
    const float li = 1.0f / l_run;
#pragma unroll
    for (int r = 0; r < 4; ++r) linv[r] = __shfl(li, 4 * g + r);
    attn_store_out_v1(a, b, h, qb, nq, lane, o, linv);
  } else {
    float *pp = a.part + (((size_t)bh * a.QB + qb) * a.splits + sp) * 16 * ATT_PSTRIDE;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v;
      v.x = o[0][r]; v.y = o[1][r]; v.z = o[2][r]; v.w = o[3][r];
      *(f32x4 *)(pp + (4 * g + r) * ATT_PSTRIDE + 4 * c) = v;
    }
    if (g == 0) {
      pp[c * ATT_PSTRIDE + 64] = m_run;
      pp[c * ATT_PSTRIDE + 65] = l_run;
    }
  }
}

