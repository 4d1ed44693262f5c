// Experiment kept for the record (round 2): the LDS-staged GEMM on v_mfma_f32_32x32x2_f32.  Correct (test_kernels.hip
// cfg 102/103) and within +-5 % of gemm_lds_kernel on every codec shape at batch 64 (profiles/r02_sweep_lds32.txt), so it
// is NOT part of libptts: halving the LDS operand reads per flop does not help, these GEMMs are bound by launch + ramp +
// tail, not by LDS bandwidth.  Included after ptts_kernels.h by the two harnesses only.
#pragma once

// LDS-staged GEMM on v_mfma_f32_32x32x2_f32 (round 2).  Same operand DMA, stages and LDS image as gemm_lds_kernel (the
// 1 KiB FM fragments), but a wave owns WM x WN tiles of 32 x 32: half the MFMA instructions per flop (64-cycle issue
// instead of 32), so the address arithmetic, DMA issue and LDS reads of a stage hide behind fewer, longer matrix ops.
// A 32-row operand tile is two FM row tiles; lane l reads rows l & 31 and, per 8-wide k block, the k group l >> 5
// (k assignment inside an MFMA is free as long as both operands agree).  The 32 x 32 accumulator holds, per lane
// (column m = l & 31, half h = l >> 5), registers 4g' .. 4g'+3 = rows n = 8g' + 4h + (0..3): four FM pieces that go
// through the common epilogue.  Plain operands only (no LayerNorm fold).
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int KC>
__global__ __launch_bounds__(256) void gemm_lds32_kernel(GemmArgs a) {
  constexpr int BMT = 4 * WM, BNT = 4 * WN;  // 16-row / 16-column tiles per workgroup (2 x 2 waves, WM x WN tiles of 32)
  constexpr int NX = BMT * KC, NFRAG = (BMT + BNT) * KC;
  constexpr int XPW = BMT / 4, WPW = BNT * KC / 4, IPS = NFRAG / 4;
  __shared__ f32x4 lds[2][NFRAG][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by;
  tile_of_block(a.swz, bx, by);
  const int mt0 = by * BMT, nt0 = bx * BNT;
  const int par = a.par ? (*a.par & 1) : 0;
  const float *Xc = a.X + par * a.Xdstride;
  const float *Xp = a.X + (par ^ 1) * a.Xdstride;
  const int halo = a.halo;
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
    xz[q] = ts < 0 && a.halo_mode == 1;
  };
#pragma unroll
  for (int q = 0; q < XPW; ++q) row_base(q, 0);
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
    for (int i = 0; i < WPW; ++i) GLDS16(wp[i] + (size_t)kf0 * 256, &lds[buf][wslot[i]][0]);
  };
  f32x16 acc[WN][WM];
#pragma unroll
  for (int i = 0; i < WN; ++i)
#pragma unroll
    for (int j = 0; j < WM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // LDS slot of this lane inside a 16-row fragment, per 8-wide k block q: 16 * (2q + (lane >> 5)) + (lane & 15)
  const int sub = (lane & 31) >> 4, r16 = lane & 15, h = lane >> 5;
  const int nst = a.KF / KC;
  issue(0, 0);
  for (int s = 0; s < nst; ++s) {
    const int cur = s & 1;
    wait_vmcnt<0>();
    __syncthreads();
    if (s + 1 < nst) issue((s + 1) * KC, cur ^ 1);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 x[WM], w[WN];
        const int slot = 16 * (2 * q + h) + r16;
#pragma unroll
        for (int j = 0; j < WM; ++j) x[j] = lds[cur][kc * BMT + 2 * (wm * WM + j) + sub][slot];
#pragma unroll
        for (int i = 0; i < WN; ++i) w[i] = lds[cur][NX + kc * BNT + 2 * (wn * WN + i) + sub][slot];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < WN; ++i)
#pragma unroll
            for (int j = 0; j < WM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i][c], x[j][c], acc[i][j], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < WN; ++i)
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const int mt = mt0 + 2 * (wm * WM + j) + sub;
      if (mt >= a.MT) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nt = nt0 + 2 * (wn * WN + i) + (g >> 1);
        if (nt >= a.NT) continue;
        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        gemm_epilogue(a, v, nt, mt, 16 * (2 * (g & 1) + h) + r16, par);
      }
    }
}
