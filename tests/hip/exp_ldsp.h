// Experiment kept for the record (round 2): the persistent form of gemm_lds_kernel (resident workgroups walk the tiles,
// operand stages prefetched across tile boundaries).  Correct (test_kernels.hip cfg 104-106), but NOT faster than one
// workgroup per tile: alone, under the two-per-CU cap, convtr3 51.2 vs 50.1 us, convtr1 45.2 vs 43.7, convtr2 45.6 vs
// 44.1, mimi.ff1 25.8 vs 24.9 (profiles/r02_sweep_persist.txt); in the two-stream pipeline 0.867 vs 0.842 ms per step
// (profiles/r02_pipeline_experiments.txt).  A workgroup's fixed costs (dispatch, cold first stage, epilogue) are therefore
// NOT what separates these GEMMs from the LDS-fed MFMA rate; the operand DMA inside the K loop is (16-23 %, tools/abl_cap.sh).
// Included after ptts_kernels.h by the two harnesses only.
#pragma once

// ---------------------------------------------------------------------------------------------
// Persistent form of gemm_lds_kernel for launches that run under the codec's occupancy cap (two workgroups per CU).
// With few workgroups per CU nothing hides a workgroup's fixed costs - dispatch, the cold first operand stage (~2 us),
// the epilogue - and K is short on this path (a 64x64 tile of convtr3 holds 3.7 us of MFMA work in a ~12 us life).
// Here gridDim.x <= 2 x CUs workgroups stay resident and walk the tiles t = blockIdx.x, + gridDim.x, ..: the operand DMA
// runs NS - 1 stages ahead of the MFMAs ACROSS tile boundaries, so the first stages of tile t + 1 are in flight while
// tile t finishes its last stages and its epilogue.  Same tiles, fragments, LDS image and arithmetic order per tile as
// gemm_lds_kernel (results are bit-identical); the first stage after an epilogue waits for vmcnt(0) because the
// epilogue's stores sit between the prefetched stages and the wait.
// ---------------------------------------------------------------------------------------------
template <int BMT, int BNT, int KC, int PRE, int NS>
__global__ __launch_bounds__(256) void gemm_ldsp_kernel(GemmArgs a, int tiles_x, int tiles_y) {
  if constexpr (PTTS_ABLATE & 128) return;
  static_assert(BMT % 4 == 0 && BNT % 2 == 0 && (BNT * KC) % 4 == 0, "tile shape");
  static_assert(NS >= 2 && NS <= 3, "stage count");
  constexpr int WMT = BMT / 2, WNT = BNT / 2;
  constexpr int NX = BMT * KC, NFRAG = (BMT + BNT) * KC;
  constexpr int XPW = BMT / 4, WPW = BNT * KC / 4, IPS = NFRAG / 4;
  __shared__ f32x4 lds[NS][NFRAG][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_x * tiles_y, G = gridDim.x;
  const int par = a.par ? (*a.par & 1) : 0;
  const float *Xc = a.X + par * a.Xdstride;
  const float *Xp = a.X + (par ^ 1) * a.Xdstride;
  const int halo = a.halo;
  const int nst = a.KF / KC;  // host guarantees KF % KC == 0 and nst >= NS - 1

  // tile id -> (column block, row block), XCD-aware as tile_of_block (tile t runs on XCD t % 8: G is a multiple of 8)
  auto tile_xy = [&](int t, int &bx, int &by) {
    if (a.swz) {
      const int per = ntiles >> 3;
      if (t < per * 8) t = (t & 7) * per + (t >> 3);
    }
    by = t / tiles_x;
    bx = t - by * tiles_x;
  };

  // ---- issue side: the tile whose operand stages are being requested (runs ahead of the compute side)
  int i_tile = blockIdx.x, i_stage = 0;
  int l_mt[XPW], l_t[XPW], l_bT[XPW];
  const float *xp[XPW];
  bool xz[XPW];
  int s_cf = 0, s_tap = 0;
  const float *wp[WPW];
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
    xz[q] = ts < 0 && a.halo_mode == 1;
  };
  auto setup_issue = [&](int t) {
    int bx, by;
    tile_xy(t, bx, by);
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
      int mt = by * BMT + wave + 4 * q;
      mt = mt < a.MT ? mt : a.MT - 1;
      l_mt[q] = mt;
      const int row = 16 * mt + (lane & 15);
      const int tt = a.ntaps > 1 ? row % a.T : 0;
      l_t[q] = tt;
      l_bT[q] = row - tt;
    }
    s_cf = 0; s_tap = 0; i_stage = 0;
#pragma unroll
    for (int q = 0; q < XPW; ++q) row_base(q, 0);
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int f = wave + 4 * i, kc = f / BNT, n = f - kc * BNT;
      const int nt = bx * BNT + n < a.NT ? bx * BNT + n : a.NT - 1;
      wp[i] = a.W + ((size_t)nt * a.KF + kc) * 256 + lane * 4;
    }
  };
  auto issue = [&](int buf) {  // the next stage of the issue-side tile; moves on to the workgroup's next tile after its last
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int q = 0; q < XPW; ++q) {
        const float *src = xz[q] ? a.zeros : xp[q] + (size_t)s_cf * 256;
        GLDS16(src, &lds[buf][kc * BMT + wave + 4 * q][0]);
      }
      if (++s_cf == a.CF) {
        s_cf = 0;
        ++s_tap;
#pragma unroll
        for (int q = 0; q < XPW; ++q) row_base(q, s_tap);
      }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) GLDS16(wp[i] + (size_t)(i_stage * KC) * 256, &lds[buf][NX + wave + 4 * i][0]);
    if (++i_stage == nst) {
      i_tile += G;
      if (i_tile < ntiles) setup_issue(i_tile);
    }
  };

  setup_issue(i_tile);
  int gi = 0, gc = 0;  // stages issued / consumed so far by this workgroup (ring positions)
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (i_tile < ntiles) { issue(gi % NS); ++gi; }

  for (int c_tile = blockIdx.x; c_tile < ntiles; c_tile += G) {
    int bx, by;
    tile_xy(c_tile, bx, by);
    const int mt0 = by * BMT, nt0 = bx * BNT;
    f32x4 acc[WNT][WMT];
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sx[WMT], sxx[WMT];
#pragma unroll
    for (int j = 0; j < WMT; ++j) sx[j] = sxx[j] = 0.f;

    for (int s = 0; s < nst; ++s, ++gc) {
      const int cur = gc % NS;
      // stage gc has landed once at most the stages issued after it are outstanding; after an epilogue its stores are
      // in the queue too, so the first stage of a later tile waits for everything
      const int ahead = gi - gc - 1;
      if (s == 0 && c_tile != (int)blockIdx.x) wait_vmcnt<0>();
      else if (ahead >= 1 && NS == 3) wait_vmcnt<IPS>();
      else wait_vmcnt<0>();
      __syncthreads();
      if (i_tile < ntiles) { issue(gi % NS); ++gi; }
      f32x4 xa[WMT], wa[WNT], xb[WMT], wb[WNT];
      auto rd = [&](int kc, f32x4 *x, f32x4 *w) {
#pragma unroll
        for (int j = 0; j < WMT; ++j) x[j] = lds[cur][kc * BMT + wm * WMT + j][lane];
#pragma unroll
        for (int i = 0; i < WNT; ++i) w[i] = lds[cur][NX + kc * BNT + wn * WNT + i][lane];
      };
      auto mm = [&](const f32x4 *x, const f32x4 *w) {
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
}

