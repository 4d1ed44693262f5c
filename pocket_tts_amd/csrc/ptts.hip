// Host side of libptts: engine construction (weight packing), FlowLM / Mimi step orchestration,
// hipGraph capture and the C ABI declared in include/ptts.h.
#include "ptts_kernels.h"
#include "ptts_flow.h"
#include "ptts_bf16.h"
#include "ptts_lm.h"
#include "ptts_ext.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptts.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess)                                                                              \
      return fail(-2, std::string(#x) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
  } while (0)
#define CHK(x)            \
  do {                    \
    int r_ = (x);         \
    if (r_ < 0) return r_; \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// Optional per-launch profiler: every kernel launch is bracketed by two HIP events on the stream it is
// launched on and tagged with its call site, its kernel name and its ALGORITHMIC bytes / flops.
// Off by default (and always off during graph capture); bench.py switches it on for a few eager steps.
struct ProfRec { std::string site, kernel; double bytes, flops; hipEvent_t a, b; };
struct Profiler { bool on = false; std::vector<ProfRec> recs; };
// Per-call context: every entry point binds its engine's profiler / tuner / zero line in the CALLING thread
// (bind_engine), so two engines driven from two threads never see each other's state.  One engine is driven by one
// thread at a time (entry points take the engine mutex; see the threading contract in include/ptts.h).
static thread_local Profiler *g_prof = nullptr;
static thread_local const char *g_site = "";
struct ProfScope {
  hipStream_t st; Profiler *pr; size_t idx;
  ProfScope(hipStream_t st_, const std::string &kernel, double bytes, double flops) : st(st_), pr(g_prof && g_prof->on ? g_prof : nullptr), idx(0) {
    if (!pr) return;
    ProfRec r{g_site, kernel, bytes, flops, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { pr = nullptr; return; }
    (void)hipEventRecord(r.a, st);
    idx = pr->recs.size();
    pr->recs.push_back(r);
  }
  ~ProfScope() { if (pr) (void)hipEventRecord(pr->recs[idx].b, st); }
};
#define SITE(x) g_site = (x)

// ------------------------------------------------------------------------------------------------
struct Lin {  // one packed weight matrix
  float *w = nullptr, *bias = nullptr;
  int N = 0, NT = 0, C = 0, CF = 0, ntaps = 1, KF = 0;
  int cout = 0, stride = 0;  // transposed-conv view
  float *ln_s = nullptr, *ln_c = nullptr;  // LayerNorm folded into this matrix (PRE_LNFOLD)
  // int8 weight-only variant (PTTS_QUANT_*): wq replaces w; ln_g = the LayerNorm gain applied to x on load
  uint8_t *wq = nullptr;
  float *wscale = nullptr, *ln_g = nullptr;
  // bf16 weight variant of a FlowLM Linear (PTTS_LM_BF16): replaces w; packed [NT][KF/2][64][8] (GemmArgs::wfmt == 2)
  void *wb16 = nullptr;
  // split-bf16 twin of a codec matrix (PTTS_CODEC_SPLIT): hi = bf16(w), lo = bf16(w - hi) images beside the fp32 one
  void *wsh = nullptr, *wsl = nullptr;
  // bf16 twin for the reduced-precision codec path (PTTS_CODEC_BF16): packed [NT][ntaps * C/32][64][8], ln_s from the rounded image
  __bf16 *wh = nullptr;
  float *ln_s_h = nullptr;
  // e4m3 twin of a SEANet conv for the fp8 path (PTTS_CODEC_FP8): packed [NT][ntaps * C/32][64][8 bytes] + per-channel scale
  void *wf8 = nullptr;
  float *wscale8 = nullptr;
  size_t bytes() const { return wq ? (size_t)NT * KF * 256 + (size_t)NT * 64 : wb16 ? (size_t)NT * KF * 512 : (size_t)NT * KF * 1024; }
};

struct TrLayer {
  float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ls1 = nullptr, *ls2 = nullptr;
  Lin qkv, out, ff1, ff2;
};

struct ptts_engine {
  ptts_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void *> allocs;
  std::map<void *, size_t> alloc_bytes;  // engine-owned allocations and their sizes (packed-engine files)
  size_t n_build_allocs = 0;             // allocs[0 .. n) were made by build_engine, in a deterministic order
  const ptts_tensor *blob_dummy = nullptr;  // ptts_create_from_file: every checkpoint lookup resolves to this zero tensor
  int blob_has_encoder = 0;
  std::map<std::string, const ptts_tensor *> tmap;
  // FlowLM
  float *bos = nullptr, *freq_lm = nullptr;
  Lin in_linear;
  std::vector<TrLayer> lm;
  float *outnorm_w, *outnorm_b;
  Lin head, adaln, input_proj, fin;
  struct Res { float *ln_w, *ln_b; Lin l0, l2; };
  std::vector<Res> res;
  Lin te_l0[2], te_l2[2];
  float *te_freqs[2], *te_alpha[2];
  std::map<int, float *> tcomb;  // lsd_steps -> [S][flow_dim]
  float *te_scratch = nullptr;
  // Mimi
  float *emb_std, *emb_mean, *up_w, *freq_mimi;
  float *quant_w = nullptr;  // quantizer.output_proj weight [C][ldim], plain (mimi_prologue_kernel)
  std::vector<TrLayer> mm;
  Lin conv0, convtr[3], res_a[3], res_b[3], conv_last;
  float *conv_last_w = nullptr, *conv_last_b = nullptr;  // plain checkpoint tensors (bf16 path's last conv)
  bool codec_bf16 = false;
  int64_t mimi_bytes_h = 0;
  // fp8 SEANet convolutions: static activation scales (device copy lives in an engine allocation, so packed-engine files
  // carry it; f8s is its host mirror).  Index: 0 = conv0 output, 1 + 3 i = convtr_i output (ELU'd), 2 + 3 i = hidden
  // activation of residual block i, 3 + 3 i = output of block i (i < 2: the next transposed conv's input)
  bool codec_split = false;  // PTTS_CODEC_SPLIT: the fp32 codec's GEMM launches use the split-bf16 images
  bool codec_fp8 = false;
  float *d_f8s = nullptr;
  float f8s[16] = {};
  int64_t mimi_bytes_f8 = 0;
  int ring = 0;
  // voice-prompt encode path (SEANet encoder, encoder transformer, downsample, speaker projection)
  bool has_encoder = false;
  Lin enc_conv0, enc_res_a[3], enc_res_b[3], enc_down[3], enc_final, enc_downsample, speaker_proj;
  std::vector<TrLayer> enc_tr;
  float *zeros = nullptr;
  int64_t lm_bytes = 0, mimi_bytes = 0;
  struct Tuner *tuner = nullptr;
  Profiler prof;
  int opt_flow_cluster = 1;
  int opt_lm_cluster = 0;  // measured slower than five launches per layer: kept as an experiment
  LmLayerP *lm_table = nullptr;  // device table of the FlowLM layers for lm_cluster_kernel (null: not eligible)
  int opt_k_rotate = 0;
  long fuse_res_min_rows = 0;
  int opt_codec_lds_target = 56 * 1024;  // see lds_pad()
  int opt_fuse_pcm = 1;      // SEANet's last conv inside the last stage's fused residual block (its output tile never leaves the CU)
  int opt_debug_taps = 0;    // materialise buffers that fused kernels keep on chip (ptts_debug_read of every stage)
  int opt_single_store = 1;  // SEANet transposed convs store their raw output once; the next conv applies ELU on its operand read
  int opt_fuse_res = 1;  // SEANet residual blocks of stages 2 and 3 as one launch each (gemm_lds_kernel<.., NT2>)
  int opt_flow_max_cus = 128;  // resident workgroups of the single-launch flow MLP (<= the CUs its stream may use)
  int opt_share_prefix = 1;    // clones of a batch-1 state share its keys / values (KvPrefix) instead of copying them
  int opt_cascade = 423;       // decode attention of such clones: 10 R + PW of attn_cascade_kernel (0: every sequence on its own)
  int n_cus = 256;             // hipDeviceProp_t::multiProcessorCount of `device` (cooperative grids never exceed it)
  std::recursive_mutex mu;  // entry points that enqueue work or touch tuner / profiler / LSD tables hold it
  int quant_flags = 0;
};
#define ENGINE_LOCK(e) std::lock_guard<std::recursive_mutex> lock_((e)->mu)

struct Scratch {
  float *x = nullptr, *h = nullptr, *ao = nullptr, *ff = nullptr, *q = nullptr, *part = nullptr, *rope = nullptr;
  int MT = 0, QB = 0, splits_cap = 0;
};

struct ptts_lm_state {
  ptts_engine *e;
  int B, cap, MT;
  float *kv = nullptr;  // [L][2][B][H][cap][64]
  int *offset = nullptr;
  std::vector<int> h_off;
  Scratch dec, pre;
  // flow head scratch (FM) + io
  float *xlat, *c, *ce, *mod, *latfm, *fx, *fh, *f1;
  float *fstat = nullptr;  // per-tile row statistics of fx (GemmArgs::stat_out / stat_in)
  // single-launch flow MLP (flow_cluster_kernel): exchange slots, flags, error word; sized for `flow_steps` LSD steps
  float *fexch = nullptr;
  unsigned long long *fflags = nullptr;
  int *ferr = nullptr;
  int flow_steps = 0, flow_rt = 1, flow_ng = 1;
  // shared prefixes (KvPrefix): device table read by the attention kernels, its host mirror, and who owns each row's prefix.
  // An owner with borrowers is not freed by ptts_lm_state_destroy until the last borrower lets go (`zombie`).
  KvPrefix *d_pre = nullptr;
  std::vector<KvPrefix> h_pre;
  std::vector<ptts_lm_state *> pre_owner;
  int casc_mode = -1;   // decode attention: -1 = cascade kernel iff rows borrow prefixes now (eager steps), 0 / 1 forced (captures)
  int n_pre = 0;        // rows of this state that have a prefix
  int borrowers = 0;    // rows of OTHER states whose prefix is this state's cache
  bool zombie = false;
  int n_graphs = 0;   // captured graphs that hold this state's buffer pointers (ensure_flow must not re-allocate under them)
  int coop_wgs = 0;   // workgroups of the cooperative launch of the last enqueued step (0: per-layer launches)
  // single-launch transformer stack (lm_cluster_kernel): exchange slots + flags, allocated on first use
  float *lexch = nullptr;
  unsigned long long *lflags = nullptr;
  float *lat, *lat_prev;  // plain [B][ldim]
  float *eos_logit;
  uint8_t *is_eos;
  // device noise source for perf runs (d_noise == NULL and rng_std > 0): N(0, rng_std^2), counter-based
  float rng_std = 0.f;
  unsigned long long rng_seed = 0;
  int *rng_ctr = nullptr;
  // continuous batching: parked rows (active[b] == 0) keep computing but do not advance their position
  int *active = nullptr;
  std::vector<int> h_active;
  size_t kv_plane() const { return (size_t)B * e->cfg.num_heads * cap * 64; }
  float *K(int l) { return kv + (size_t)(2 * l) * kv_plane(); }
  float *V(int l) { return kv + (size_t)(2 * l + 1) * kv_plane(); }
};

struct ptts_mimi_state {
  ptts_engine *e;
  int B, MTb, MT16;
  int *frame = nullptr, *offset = nullptr;
  int h_frame = 0;
  float *kv = nullptr;  // [ML][2][B][H][ring][64]
  float *zl, *zq, *u0, *u, *h, *ao, *ff, *q, *part, *tr_out, *rope;
  long zq_stride, tr_stride;
  int splits;
  float *a0;
  long a0_stride;
  float *cbuf[3], *craw[3], *rbuf[3], *sbuf[3];  // cbuf/sbuf/rbuf hold ELU'd values, craw the raw skip input
  long c_stride[3], s_stride[3];
  int rows[4];  // rows per sequence at each SEANet stage
  float *pcm_dbg;
  int16_t *pcm_i16 = nullptr;
  // fused last stage ("fuse_pcm"): per-row partial PCM + what each 64-row tile leaves for the first two rows of the next
  float *pcm_part = nullptr, *pcm_carry = nullptr;
  long pcm_cstride = 0;
  size_t kv_plane() const { return (size_t)B * e->cfg.m_heads * e->ring * 64; }
  float *K(int l) { return kv + (size_t)(2 * l) * kv_plane(); }
  float *V(int l) { return kv + (size_t)(2 * l + 1) * kv_plane(); }
};

struct ptts_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // second capture of an LM step for states whose rows share a prefix (attn_cascade_kernel instead of the per-row decode
  // attention): ptts_graph_launch picks by the state's borrow count, which the host knows
  hipGraph_t graph_c = nullptr;
  hipGraphExec_t exec_c = nullptr;
  ptts_lm_state *lm = nullptr;
  ptts_mimi_state *mimi = nullptr;
  hipStream_t cap_stream = nullptr;
  int coop_wgs = 0;  // workgroups of the largest cooperative launch inside (0: none): the launch stream needs that many CUs
  hipStream_t cu_checked = nullptr;  // the last launch stream whose CU mask was found large enough (checked once per stream)
};

// ------------------------------------------------------------------------------------------------
// allocation helpers
// Zero-filled device allocation.  The fill is queued on `st` (the engine stream, which is
// non-blocking and therefore NOT ordered against the null stream a plain hipMemset would use).
// The zero fill is queued on the stream of the innermost AllocScope of the CALLING thread (thread-local: a
// concurrent call on another engine / thread can no longer redirect it).
static thread_local hipStream_t t_alloc_stream = nullptr;
struct AllocScope {
  hipStream_t prev;
  explicit AllocScope(hipStream_t st) : prev(t_alloc_stream) { t_alloc_stream = st; }
  ~AllocScope() { t_alloc_stream = prev; }
};
static int dalloc(ptts_engine *e, void **p, size_t bytes) {
  hipStream_t st = t_alloc_stream ? t_alloc_stream : (e ? e->stream : nullptr);
  if (!st) return fail(-1, "internal: allocation outside an AllocScope");
  if (bytes == 0) bytes = 256;
  *p = nullptr;
  const hipError_t err = hipMalloc(p, bytes);
  if (err != hipSuccess) {
    (void)hipGetLastError();  // the error is reported here; do not leave it sticky for the next hipGetLastError()
    *p = nullptr;
    return fail(-2, "hipMalloc of " + std::to_string(bytes) + " bytes: " + hipGetErrorString(err));
  }
  HIPCHK(hipMemsetAsync(*p, 0, bytes, st));
  if (e) { e->allocs.push_back(*p); e->alloc_bytes[*p] = bytes; }
  return 0;
}
template <typename T>
static int dallocT(ptts_engine *e, T **p, size_t n) {
  return dalloc(e, (void **)p, n * sizeof(T));
}


static const ptts_tensor *find_tensor(ptts_engine *e, const std::string &name, int64_t numel, int *err) {
  if (e->blob_dummy) return e->blob_dummy;  // loading a packed engine: the packing kernels run on zeros, the real images follow
  auto it = e->tmap.find(name);
  if (it == e->tmap.end()) {
    *err = fail(-3, "missing tensor: " + name);
    return nullptr;
  }
  if (numel >= 0 && it->second->numel != numel) {
    *err = fail(-3, "tensor " + name + ": expected " + std::to_string(numel) + " elements, got " +
                        std::to_string(it->second->numel));
    return nullptr;
  }
  return it->second;
}

// copies a small fp32 vector (norm gains, biases, ...) into engine-owned memory, padded
static int copy_vec(ptts_engine *e, const std::string &name, int64_t n, float **out, int64_t pad_to = 0) {
  int err = 0;
  const ptts_tensor *t = find_tensor(e, name, n, &err);
  if (!t) return err;
  CHK(dallocT(e, out, std::max<int64_t>(pad_to, n)));
  HIPCHK(hipMemcpyAsync(*out, t->d_data, n * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

struct PackPart { std::string w, b; int N; };

// packs one or several [N_i][C][ntaps] matrices (stacked along N) into one Lin
static int pack_lin(ptts_engine *e, Lin *L, const std::vector<PackPart> &parts, int C, int ntaps, int mode = 0,
                    int cout = 0, int stride = 0, const std::string &ln_w = "", const std::string &ln_b = "",
                    int creal = 0, int wfmt = 0) {
  const bool q8 = wfmt == 1, b16 = wfmt == 2;
  if (C % 16) return fail(-4, "channel count must be a multiple of 16: " + parts[0].w);
  if (b16 && (parts.size() != 1 || mode != 0 || ntaps != 1 || (C / 16) % 2))
    return fail(-4, "bf16 LM weights need a single Linear matrix with in_features % 32 == 0: " + parts[0].w);
  if (q8 && (parts.size() != 1 || mode != 0 || ntaps != 1 || (C / 16) % 4))
    return fail(-4, "int8 weights need a single Linear matrix with in_features % 64 == 0: " + parts[0].w);
  int ntot = 0;
  for (auto &p : parts) ntot += cdiv(p.N, 16);
  L->NT = ntot;
  L->C = C;
  L->CF = C / 16;
  L->ntaps = ntaps;
  L->KF = L->CF * ntaps;
  L->cout = cout;
  L->stride = stride;
  L->N = 0;
  CHK(dallocT(e, &L->w, (size_t)L->NT * L->KF * 256));
  bool any_bias = false;
  for (auto &p : parts) any_bias |= !p.b.empty();
  const float *gam = nullptr, *bet = nullptr;
  if (!ln_w.empty()) {
    // fold the preceding LayerNorm: gain into the weights, mean / bias terms into two per-row vectors
    int err = 0;
    const ptts_tensor *tg = find_tensor(e, ln_w, C, &err), *tb = find_tensor(e, ln_b, C, &err);
    if (!tg || !tb) return err;
    gam = tg->d_data;
    bet = tb->d_data;
    CHK(dallocT(e, &L->ln_s, (size_t)L->NT * 16));
    CHK(dallocT(e, &L->ln_c, (size_t)L->NT * 16));
    any_bias = false;  // the bias is absorbed into ln_c
  }
  if (any_bias) CHK(dallocT(e, &L->bias, (size_t)L->NT * 16));
  int nt_off = 0;
  for (auto &p : parts) {
    int err = 0;
    const ptts_tensor *t = find_tensor(e, p.w, (int64_t)(mode == 0 ? p.N : (p.N / stride)) * (creal > 0 ? creal : C) * (mode == 0 ? ntaps : 2 * stride), &err);
    if (!t) return err;
    int nt = cdiv(p.N, 16);
    long total = (long)nt * L->KF * 256;
    pack_weight_kernel<<<cdiv(total, 256), 256, 0, e->stream>>>(t->d_data, L->w, p.N, C, ntaps, mode, cout, stride,
                                                                nt_off, L->KF, total, q8 ? nullptr : gam, creal > 0 ? creal : C);
    if (gam && !q8) {
      const float *bsrc = nullptr;
      if (!p.b.empty()) {
        const ptts_tensor *tb2 = find_tensor(e, p.b, p.N, &err);
        if (!tb2) return err;
        bsrc = tb2->d_data;
      }
      fold_ln_kernel<<<p.N, 64, 0, e->stream>>>(t->d_data, gam, bet, bsrc, L->ln_s, L->ln_c, p.N, C, nt_off * 16);
    }
    if (any_bias) {
      const float *bsrc = nullptr;
      if (!p.b.empty()) {
        const ptts_tensor *tb = find_tensor(e, p.b, mode == 0 ? p.N : cout, &err);
        if (!tb) return err;
        bsrc = tb->d_data;
      }
      pack_bias_kernel<<<cdiv(nt * 16, 256), 256, 0, e->stream>>>(bsrc, L->bias, p.N, mode, cout, nt_off * 16, nt * 16);
    }
    nt_off += nt;
    L->N += p.N;
  }
  if (q8) {
    int err = 0;
    const float *bsrc = nullptr;
    if (gam && !parts[0].b.empty()) {
      const ptts_tensor *tb2 = find_tensor(e, parts[0].b, parts[0].N, &err);
      if (!tb2) return err;
      bsrc = tb2->d_data;
    }
    CHK(dalloc(e, (void **)&L->wq, (size_t)L->NT * L->KF * 256));
    CHK(dallocT(e, &L->wscale, (size_t)L->NT * 16));
    if (gam) CHK(copy_vec(e, ln_w, C, &L->ln_g));
    quantize_packed_kernel<<<L->NT, 256, 0, e->stream>>>(L->w, L->wq, L->wscale, L->KF, gam, bet, bsrc, L->N, L->ln_s, L->ln_c);
    HIPCHK(hipGetLastError());
    // the fp32 image is only the quantiser's input
    HIPCHK(hipStreamSynchronize(e->stream));
    e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void *)L->w), e->allocs.end());
    e->alloc_bytes.erase((void *)L->w);
    HIPCHK(hipFree(L->w));
    L->w = nullptr;
  }
  if (b16) {
    // bf16 image from the fp32 one (LayerNorm gain already multiplied in); the fold vector s comes from the ROUNDED rows
    CHK(dalloc(e, &L->wb16, (size_t)L->NT * L->KF * 512));
    pack_weight_b16(e->stream, L->w, L->wb16, gam ? L->ln_s : nullptr, L->NT, L->KF);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void *)L->w), e->allocs.end());
    e->alloc_bytes.erase((void *)L->w);
    HIPCHK(hipFree(L->w));
    L->w = nullptr;
  }
  HIPCHK(hipGetLastError());
  return 0;
}


// bf16 twin of a packed matrix for the reduced-precision codec path (single-part matrices only)
static int pack_lin_h(ptts_engine *e, Lin *L, const std::string &wname, int N, int C, int ntaps, int mode = 0, int cout = 0,
                      int stride = 0, const std::string &ln_w = "") {
  if (C % 32) return fail(-4, "bf16 codec path: channel count must be a multiple of 32: " + wname);
  int err = 0;
  const ptts_tensor *t = find_tensor(e, wname, -1, &err);
  if (!t) return err;
  const float *gam = nullptr;
  if (!ln_w.empty()) {
    const ptts_tensor *tg = find_tensor(e, ln_w, C, &err);
    if (!tg) return err;
    gam = tg->d_data;
  }
  const int KBt = (C / 32) * ntaps;
  const long total = (long)L->NT * KBt * 512;
  CHK(dalloc(e, (void **)&L->wh, (size_t)total * 2));
  pack_weight_h_kernel<<<cdiv(total, 256), 256, 0, e->stream>>>(t->d_data, L->wh, N, C, ntaps, mode, cout, stride, KBt, total, gam);
  if (gam) {
    CHK(dallocT(e, &L->ln_s_h, (size_t)L->NT * 16));
    fold_s_h_kernel<<<N, 64, 0, e->stream>>>(L->wh, L->ln_s_h, N, KBt);
  }
  HIPCHK(hipGetLastError());
  e->mimi_bytes_h += total * 2;
  return 0;
}

static int pack_tr_layer(ptts_engine *e, TrLayer *T, const std::string &p, int d, int ff, bool ls, int quant = 0) {
  const int qa = (quant & PTTS_LM_BF16) ? 2 : (quant & PTTS_QUANT_ATTENTION) ? 1 : 0;
  const int qf = (quant & PTTS_LM_BF16) ? 2 : (quant & PTTS_QUANT_FFN) ? 1 : 0;
  CHK(copy_vec(e, p + ".norm1.weight", d, &T->ln1_w));
  CHK(copy_vec(e, p + ".norm1.bias", d, &T->ln1_b));
  CHK(copy_vec(e, p + ".norm2.weight", d, &T->ln2_w));
  CHK(copy_vec(e, p + ".norm2.bias", d, &T->ln2_b));
  if (ls) {
    CHK(copy_vec(e, p + ".layer_scale_1.scale", d, &T->ls1));
    CHK(copy_vec(e, p + ".layer_scale_2.scale", d, &T->ls2));
  }
  CHK(pack_lin(e, &T->qkv, {{p + ".self_attn.in_proj.weight", "", 3 * d}}, d, 1, 0, 0, 0, p + ".norm1.weight", p + ".norm1.bias", 0, qa));
  CHK(pack_lin(e, &T->out, {{p + ".self_attn.out_proj.weight", "", d}}, d, 1, 0, 0, 0, "", "", 0, qa));
  CHK(pack_lin(e, &T->ff1, {{p + ".linear1.weight", "", ff}}, d, 1, 0, 0, 0, p + ".norm2.weight", p + ".norm2.bias", 0, qf));
  CHK(pack_lin(e, &T->ff2, {{p + ".linear2.weight", "", d}}, ff, 1, 0, 0, 0, "", "", 0, qf));
  return 0;
}

static int make_freq(ptts_engine *e, float **out, float max_period) {
  // freqs = exp(ds * (-ln(max_period) * 2 / D)) in fp32, D = 64 (reference rope.py:28-29)
  float h[32];
  const float coef = (float)(-std::log((double)max_period) * 2.0 / 64.0);
  for (int i = 0; i < 32; ++i) h[i] = std::exp((float)i * coef);
  CHK(dallocT(e, out, 32));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(*out, h, sizeof(h), hipMemcpyHostToDevice));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GEMM dispatch
// int8-weight variants exist for these tiles and for plain / LN-folded operands only
template <int TN, int TM, int WK, int WN, int WM>
static void launch_cfg_q8(hipStream_t st, const GemmArgs &a, int pre) {
  dim3 grid(cdiv(a.NT, TN * WN), cdiv(a.MT, TM * WM));
  dim3 block(64 * WK * WN * WM);
  if (pre == PRE_LNFOLD) gemm_kernel<TN, TM, WK, WN, WM, PRE_LNFOLD, true><<<grid, block, 0, st>>>(a);
  else gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE, true><<<grid, block, 0, st>>>(a);
}

// Occupancy cap of the codec's GEMM launches (engine option "codec_lds_target", bytes <= 64 KB; 0 = off): they request
// dynamic LDS up to this total per workgroup, which limits their workgroups per CU to 160 KB / target and leaves wave slots
// and registers for the FlowLM stream's kernels, whose dependent chain is what the pipelined step waits for.  Measured at
// batch 64 (tools/ab.sh env PTTS_CODEC_LDS_TARGET, final kernels): 0 -> 0.893 ms per step, 36 KB (4 per CU) -> 0.882,
// 44 KB (3) -> 0.864, 56 KB (2) -> 0.851; the codec graph alone 0.548 -> 0.554 (44 KB) -> 0.589 ms (56 KB).
static thread_local int g_lds_target = 0;
static unsigned lds_pad(int static_bytes) { return g_lds_target > static_bytes ? (unsigned)(g_lds_target - static_bytes) : 0u; }
#define LDS_LAUNCH(kernel, grid, block, dyn, st, arg) (kernel)<<<(grid), (block), (dyn), (st)>>>(arg)

template <int TN, int TM, int WK, int WN, int WM>
static void launch_cfg(hipStream_t st, const GemmArgs &a, int pre) {
  dim3 grid(cdiv(a.NT, TN * WN), cdiv(a.MT, TM * WM));
  dim3 block(64 * WK * WN * WM);
  const unsigned dyn = lds_pad(WK > 1 ? WK * WN * WM * TN * TM * 1024 : 0);
  switch (pre) {
    case PRE_NONE: LDS_LAUNCH((gemm_kernel<TN, TM, WK, WN, WM, PRE_NONE>), grid, block, dyn, st, a); break;
    case PRE_LNFOLD: LDS_LAUNCH((gemm_kernel<TN, TM, WK, WN, WM, PRE_LNFOLD>), grid, block, dyn, st, a); break;
    case PRE_LNMOD: LDS_LAUNCH((gemm_kernel<TN, TM, WK, WN, WM, PRE_LNMOD>), grid, block, dyn, st, a); break;
    case PRE_ELU: LDS_LAUNCH((gemm_kernel<TN, TM, WK, WN, WM, PRE_ELU>), grid, block, dyn, st, a); break;
    default: LDS_LAUNCH((gemm_kernel<TN, TM, WK, WN, WM, PRE_ADDSILU>), grid, block, dyn, st, a); break;
  }
}

template <int BMT, int BNT>
static void launch_lds(hipStream_t st, const GemmArgs &a, int pre) {
  dim3 grid(cdiv(a.NT, BNT), cdiv(a.MT, BMT));
  // Under the occupancy cap the padded LDS is free, so the capped launches run a THREE-stage ring (two stages in flight
  // with two workgroups per CU): 0.862 -> 0.857 ms per pipelined step; four k-fragments per stage instead: 0.909
  // (tools/ab.sh env PTTS_LDS_VARIANT 0 / 1 / 2).  Uncapped launches keep two stages (never slower, round 1).
  static const int variant = [] { const char *v = getenv("PTTS_LDS_VARIANT"); return v ? atoi(v) : 1; }();
  if constexpr (BMT == 4 && BNT <= 4) {
    if (variant == 1 && g_lds_target) {
      const unsigned dyn3 = lds_pad(3 * (BMT + BNT) * 2 * 1024);
      if (pre == PRE_LNFOLD) gemm_lds_kernel<BMT, BNT, 2, PRE_LNFOLD, 3><<<grid, 256, dyn3, st>>>(a);
      else if (pre == PRE_ELU) gemm_lds_kernel<BMT, BNT, 2, PRE_ELU, 3><<<grid, 256, dyn3, st>>>(a);
      else gemm_lds_kernel<BMT, BNT, 2, PRE_NONE, 3><<<grid, 256, dyn3, st>>>(a);
      return;
    }
    if (variant == 2 && g_lds_target && a.KF % 4 == 0 && pre != PRE_ELU) {
      const unsigned dyn4 = lds_pad(2 * (BMT + BNT) * 4 * 1024);
      if (pre == PRE_LNFOLD) gemm_lds_kernel<BMT, BNT, 4, PRE_LNFOLD><<<grid, 256, dyn4, st>>>(a);
      else gemm_lds_kernel<BMT, BNT, 4, PRE_NONE><<<grid, 256, dyn4, st>>>(a);
      return;
    }
  }
  const unsigned dyn = lds_pad(2 * (BMT + BNT) * 2 * 1024);
  if (pre == PRE_LNFOLD) LDS_LAUNCH((gemm_lds_kernel<BMT, BNT, 2, PRE_LNFOLD>), grid, dim3(256), dyn, st, a);
  else if (pre == PRE_ELU) LDS_LAUNCH((gemm_lds_kernel<BMT, BNT, 2, PRE_ELU>), grid, dim3(256), dyn, st, a);
  else LDS_LAUNCH((gemm_lds_kernel<BMT, BNT, 2, PRE_NONE>), grid, dim3(256), dyn, st, a);
}

// Tile selection.  K-split configs (TM row tiles per wave, 4 waves split K, LDS-reduced) give NT x ceil(MT/TM)
// workgroups and stream each weight fragment ceil(MT/TM) times (L2 / Infinity Cache absorb the re-reads);
// the 2-D tiled configs amortise operand loads over 2x4 tiles per wave but need a large grid to fill 256 CUs.
// Pick the K-split row-tile count TM that still yields >= ~256 workgroups, and use 2-D tiles only when
// their grid is large.
static int pick_cfg(const GemmArgs &a) {
  const long tiled = (long)cdiv(a.NT, 4) * cdiv(a.MT, 8);
  // many rows (codec convs at batch >= 16): LDS-staged kernel, each operand fragment DMA'd once per workgroup
  // (tests/hip/sweep_gemm.hip: 5-12 % faster than the register-staged tiles on these shapes)
  if (a.MT >= 256 && a.NT >= 4 && a.KF % 2 == 0 && a.epi != EPI_QKV && !a.mod_scale) return a.NT >= 8 ? 8 : 9;
  if (a.MT > 4 && tiled >= 192) return a.NT >= 4 ? 3 : a.NT >= 2 ? 4 : 5;
  // few output tiles but a long K (Mimi FFN2 / conv k7 / out_proj at moderate batch): 2x4 tiles per wave,
  // the 4 waves of a workgroup split K -> 4x the workgroups of the 2-D tiling at the same operand reuse
  if (a.MT >= 8 && a.NT >= 2 && a.KF >= 32 && (long)cdiv(a.NT, 2) * cdiv(a.MT, 4) >= 128) return 7;
  if (a.MT > 64) {  // K-split would re-stream weights too often
    if (tiled < 192) return 6;  // small problem, many rows: one tile per wave for the largest grid
    return a.NT >= 4 ? 3 : a.NT >= 2 ? 4 : 5;
  }
  int tm = a.MT >= 4 ? 4 : a.MT >= 2 ? 2 : 1;
  while (tm > 1 && (long)a.NT * cdiv(a.MT, tm) < 256) tm >>= 1;
  return tm == 4 ? 2 : tm == 2 ? 1 : 0;
}
static constexpr int kNumCfg = 18;  // 16, 17 appended in round 2 (older cache files stay valid)
static const char *const kCfgName[kNumCfg] = {
    "gemm<1,1,8,1,1>", "gemm<1,2,4,1,1>", "gemm<1,4,4,1,1>", "gemm<2,4,1,2,2>", "gemm<2,4,1,1,4>", "gemm<1,4,1,1,4>",
    "gemm<1,1,1,1,4>", "gemm<2,4,4,1,1>", "gemm_lds<4,8,2>", "gemm_lds<4,4,2>", "gemm<2,2,4,1,1>", "gemm<1,1,4,1,1>",
    "gemm_lds<4,2,2>", "gemm<1,2,1,2,2>", "gemm<2,4,2,2,1>", "gemm_lds<8,8,2>", "gemm_lds<8,4,2>", "gemm_lds<8,2,2>"};
// {TN, TM, WK, WN, WM} of the register-staged configs, {BNT, BMT, 0, 0, 0} of the LDS-staged ones
static const int kCfgShape[kNumCfg][5] = {{1, 1, 8, 1, 1}, {1, 2, 4, 1, 1}, {1, 4, 4, 1, 1}, {2, 4, 1, 2, 2}, {2, 4, 1, 1, 4},
                                          {1, 4, 1, 1, 4}, {1, 1, 1, 1, 4}, {2, 4, 4, 1, 1}, {8, 4, 0, 0, 0}, {4, 4, 0, 0, 0},
                                          {2, 2, 4, 1, 1}, {1, 1, 4, 1, 1}, {2, 4, 0, 0, 0}, {1, 2, 1, 2, 2}, {2, 4, 2, 2, 1},
                                          {8, 8, 0, 0, 0}, {4, 8, 0, 0, 0}, {2, 8, 0, 0, 0}};

static bool q8_cfg(int cfg) { return cfg == 0 || cfg == 1 || cfg == 2 || cfg == 3 || cfg == 7 || cfg == 10 || cfg == 11; }

static bool cfg_valid(int cfg, const GemmArgs &a, int pre) {
  const int *s = kCfgShape[cfg];
  if (a.wfmt == 3) {  // split bf16: every register-staged configuration, whole pairs of k-fragments per wave
    if (!split_cfg(cfg) || (pre != PRE_NONE && pre != PRE_LNFOLD)) return false;
    if (a.KF % (2 * s[2])) return false;
  } else if (a.wfmt) {  // whole groups of four (int8) / two (bf16) k-fragments per wave
    if (!q8_cfg(cfg) || (pre != PRE_NONE && pre != PRE_LNFOLD)) return false;
    // bf16 weights: the 8-wave single-tile configuration is excluded - its bf16 instantiation produced NaNs on the GPU
    // (gpurun_out/r3 debug run, every shape) while the 4-wave K-split and the 2-D tilings are exact; not understood yet
    if (a.wfmt == 2 && cfg == 0) return false;
    if (a.KF % ((a.wfmt == 1 ? 4 : 2) * s[2])) return false;
  }
  if (s[2] == 0) {  // LDS-staged: two k-fragments per stage, plain or LN-folded operand only
    if (a.KF % 2 || (pre != PRE_NONE && pre != PRE_LNFOLD && pre != PRE_ELU)) return false;
    return a.MT >= s[1] && 2 * a.NT >= s[0];
  }
  const int tn = s[0] * s[3], tm = s[1] * s[4];
  if (tm > 1 && tm > 2 * a.MT) return false;  // mostly padding
  if (tn > 1 && tn > 2 * a.NT) return false;
  if (s[2] > 1 && a.KF < 2) return false;  // nothing to split
  return true;
}

// XCD-aware mapping (tile_of_block) when the activations outweigh the weights and several column blocks re-read them
// (PMC: conv k7 with 7.3 MB of weights and 2 MB of activations fetched 112 MB when it was swizzled by rows)
static int swz_for(int cfg, const GemmArgs &a) {
  static const int swz_env = [] { const char *v = getenv("PTTS_SWZ"); return v ? atoi(v) : -1; }();
  const int *sh = kCfgShape[cfg];
  const int gx = sh[2] == 0 ? cdiv(a.NT, sh[0]) : cdiv(a.NT, sh[0] * sh[3]);
  if (gx <= 1) return 0;
  // swizzle when the activations (M x C) outweigh the weights (N x K, K = taps x C)
  return swz_env >= 0 ? swz_env : ((double)a.M * a.CF > (double)a.NT * 16 * a.KF && a.MT >= 64);
}

static void launch_by_cfg(hipStream_t st, const GemmArgs &a_in, int pre, int cfg) {
  GemmArgs a = a_in;
  a.swz = swz_for(cfg, a);
  if (a.wfmt == 2) {
    launch_gemm_b16(st, a, pre, cfg, 0);
    return;
  }
  if (a.wfmt == 3) {
    const int *sh = kCfgShape[cfg];
    launch_gemm_split(st, a, pre, cfg, lds_pad(sh[2] > 1 ? sh[2] * sh[3] * sh[4] * sh[0] * sh[1] * 1024 : 0));
    return;
  }
  if (a.wfmt == 1) {
    switch (cfg) {
      case 0: launch_cfg_q8<1, 1, 8, 1, 1>(st, a, pre); break;
      case 1: launch_cfg_q8<1, 2, 4, 1, 1>(st, a, pre); break;
      case 2: launch_cfg_q8<1, 4, 4, 1, 1>(st, a, pre); break;
      case 7: launch_cfg_q8<2, 4, 4, 1, 1>(st, a, pre); break;
      case 10: launch_cfg_q8<2, 2, 4, 1, 1>(st, a, pre); break;
      case 11: launch_cfg_q8<1, 1, 4, 1, 1>(st, a, pre); break;
      default: launch_cfg_q8<2, 4, 1, 2, 2>(st, a, pre); break;  // 3: no K split, any KF % 4 == 0
    }
    return;
  }
  switch (cfg) {
    case 0: launch_cfg<1, 1, 8, 1, 1>(st, a, pre); break;  // 8 waves: most bytes in flight per CU for cold weights
    case 1: launch_cfg<1, 2, 4, 1, 1>(st, a, pre); break;
    case 2: launch_cfg<1, 4, 4, 1, 1>(st, a, pre); break;
    case 3: launch_cfg<2, 4, 1, 2, 2>(st, a, pre); break;
    case 4: launch_cfg<2, 4, 1, 1, 4>(st, a, pre); break;
    case 5: launch_cfg<1, 4, 1, 1, 4>(st, a, pre); break;
    case 6: launch_cfg<1, 1, 1, 1, 4>(st, a, pre); break;
    case 7: launch_cfg<2, 4, 4, 1, 1>(st, a, pre); break;
    case 8: launch_lds<4, 8>(st, a, pre); break;
    case 9: launch_lds<4, 4>(st, a, pre); break;
    case 10: launch_cfg<2, 2, 4, 1, 1>(st, a, pre); break;
    case 11: launch_cfg<1, 1, 4, 1, 1>(st, a, pre); break;
    case 12: launch_lds<4, 2>(st, a, pre); break;
    case 13: launch_cfg<1, 2, 1, 2, 2>(st, a, pre); break;
    case 14: launch_cfg<2, 4, 2, 2, 1>(st, a, pre); break;
    case 16: launch_lds<8, 4>(st, a, pre); break;
    case 17: launch_lds<8, 2>(st, a, pre); break;
    default: launch_lds<8, 8>(st, a, pre); break;
  }
}

// Per-engine table of measured tile choices.  `ptts_tune` runs one FlowLM step and one codec frame of a given
// batch on scratch states with `active` set: every GEMM shape met for the first time is timed with every valid
// configuration (caches flushed before each timed launch, as in the real step where ~1 GB streams between two
// uses of a weight) and the fastest is remembered.  Shapes never tuned fall back to pick_cfg.
typedef std::array<int, 13> TuneKey;
static constexpr int kTuneVersion = 2;  // bump when the key or the configuration list changes (cache files carry it)
struct Tuner {
  std::map<TuneKey, int> table;
  bool active = false;
  void *flush = nullptr;
  size_t flush_bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::string log;
};
static thread_local bool g_use_split = false;  // set around the codec's enqueue by engines built with PTTS_CODEC_SPLIT
static thread_local Tuner *g_tuner = nullptr;
static thread_local const float *g_zeros = nullptr;  // both set by the entry points from the engine
static thread_local int g_krot = 1;

static TuneKey tune_key(const GemmArgs &a, int pre) {
  return TuneKey{a.NT, a.KF, a.CF, a.ntaps, a.MT, a.epi, pre, a.act, a.xstride, a.halo_mode, a.Yraw ? 1 : 0, a.R ? 1 : 0, a.wfmt};
}

// Evicts L2 and the Infinity Cache by READING a large buffer (a write flush would leave dirty lines whose
// write-back then competes with the timed kernel).
__global__ void flush_read_kernel(const f32x4 *p, size_t n, float *sink) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s.x + s.y + s.z + s.w == 123.456f) *sink = s.x;  // never true for a zero buffer; keeps the loads alive
}

static int tune_one(hipStream_t st, const GemmArgs &a, int pre, Tuner &t) {
  // tiles are timed WITHOUT the codec's occupancy cap: under it the LDS-staged tiles look slower alone and the search drifts
  // to the register-heavy K-split tiles, which cost the pipelined step 6 % (tools/ab.sh cache)
  struct NoCap { int keep; NoCap() : keep(g_lds_target) { g_lds_target = 0; } ~NoCap() { g_lds_target = keep; } } nocap;
  int best = pick_cfg(a);
  float best_ms = 1e30f, heur_ms = 0.f;
  const int heur = best;
  // PTTS_TUNE_EXCLUDE="9,12": experiment knob, drops configurations from the search
  static const unsigned excl = [] {
    unsigned m = 0;
    if (const char *v = getenv("PTTS_TUNE_EXCLUDE"))
      for (const char *p = v; *p;) { m |= 1u << (atoi(p) & 31); while (*p && *p != ',') ++p; if (*p) ++p; }
    return m;
  }();
  static const bool verbose = getenv("PTTS_TUNE_VERBOSE") != nullptr;  // log every configuration's time
  std::string all;
  for (int cfg = 0; cfg < kNumCfg; ++cfg) {
    if (!cfg_valid(cfg, a, pre) || ((excl >> cfg) & 1)) continue;
    float ms_min = 1e30f;
    for (int r = 0; r < 4; ++r) {
      if (t.flush) flush_read_kernel<<<4096, 256, 0, st>>>((const f32x4 *)t.flush, t.flush_bytes / 16, (float *)t.flush);
      (void)hipEventRecord(t.e0, st);
      launch_by_cfg(st, a, pre, cfg);
      (void)hipEventRecord(t.e1, st);
      if (hipEventSynchronize(t.e1) != hipSuccess) return heur;
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, t.e0, t.e1);
      ms_min = std::min(ms_min, ms);
    }
    if (cfg == heur) heur_ms = ms_min;
    if (ms_min < best_ms) { best_ms = ms_min; best = cfg; }
    if (verbose) { char b[64]; snprintf(b, sizeof b, " %d:%.1f", cfg, ms_min * 1e3); all += b; }
  }
  if (best_ms > 1e29f) return heur;
  char line[256];
  snprintf(line, sizeof line, "%s NT=%d KF=%d taps=%d MT=%d epi=%d pre=%d: %s %.1f us (heuristic %s %.1f us)\n", g_site, a.NT,
           a.KF, a.ntaps, a.MT, a.epi, pre, kCfgName[best], best_ms * 1e3, kCfgName[heur], heur_ms * 1e3);
  t.log += line;
  if (verbose) t.log += "   all (cfg:us)" + all + "\n";
  return best;
}

static void launch_gemm(hipStream_t st, const GemmArgs &a_in, int pre) {
  GemmArgs a = a_in;
  a.zeros = g_zeros;
  a.krot = g_krot;
  // algorithmic traffic: weights once + input rows once (x taps re-read from cache, not counted) + output
  const double K = (double)a.KF * 16, N = (double)a.NT * 16, M = (double)a.M;
  double bytes = 4.0 * (N * K + M * (double)a.CF * 16 + M * N);
  if (a.epi == EPI_RES || a.epi == EPI_GATE) bytes += 4.0 * M * N;
  if (a.epi == EPI_PCM && a.NT == 1 && pre == PRE_NONE && !a.Wq && a.CF == 4 && a.ntaps <= 4) {  // one output channel: vector-ALU kernel
    ProfScope ps(st, "pcm_conv", 4.0 * (M * (double)a.CF * 16 + M), 2.0 * M * K);
    pcm_conv_kernel<<<cdiv(a.MT, 4), 256, 0, st>>>(a);
    return;
  }
  int cfg = -1;
  if (g_tuner) {
    const TuneKey key = tune_key(a, pre);
    auto it = g_tuner->table.find(key);
    if (it != g_tuner->table.end()) cfg = it->second;
    else if (g_tuner->active) cfg = g_tuner->table[key] = tune_one(st, a, pre, *g_tuner);
  }
  {
    // PTTS_FORCE_CFG="NT:MT:cfg[,NT:MT:cfg...]": experiment knob, pins the configuration of one GEMM shape
    static const std::vector<std::array<int, 3>> forced = [] {
      std::vector<std::array<int, 3>> v;
      if (const char *e = getenv("PTTS_FORCE_CFG")) {
        std::array<int, 3> t;
        const char *p = e;
        while (sscanf(p, "%d:%d:%d", &t[0], &t[1], &t[2]) == 3) {
          v.push_back(t);
          while (*p && *p != ',') ++p;
          if (!*p) break;
          ++p;
        }
      }
      return v;
    }();
    for (auto &f : forced)
      if (f[0] == a.NT && f[1] == a.MT && f[2] >= 0 && f[2] < kNumCfg && cfg_valid(f[2], a, pre)) cfg = f[2];
  }
  if (cfg < 0 || !cfg_valid(cfg, a, pre)) cfg = pick_cfg(a);
  if (a.wfmt == 3 && !cfg_valid(cfg, a, pre)) {  // the heuristic may name an LDS-staged tile: nearest register-staged one
    cfg = a.MT >= 8 ? 3 : 13;
    for (int c : {3, 13, 4, 6, 11}) if (cfg_valid(c, a, pre)) { cfg = c; break; }
  }
  if (a.wfmt && a.wfmt != 3 && !cfg_valid(cfg, a, pre)) cfg = (a.wfmt == 2 && !cfg_valid(3, a, pre) && cfg_valid(11, a, pre)) ? 11 : 3;
  if (a.wfmt == 1) bytes -= 3.0 * N * K;  // one byte per weight
  if (a.wfmt == 2) bytes -= 2.0 * N * K;  // two
  // label = configuration + operand variant + "@<work-items>" (what rocprofv3 reports as Grid_Size), so that the
  // launches of one label are GEMMs of one grid, i.e. of one (NT, MT) shape class
  const int *sh = kCfgShape[cfg];
  const long wgs = sh[2] == 0 ? (long)cdiv(a.NT, sh[0]) * cdiv(a.MT, sh[1])
                              : (long)cdiv(a.NT, sh[0] * sh[3]) * cdiv(a.MT, sh[1] * sh[4]);
  const long threads = wgs * (sh[2] == 0 ? 256 : 64 * sh[2] * sh[3] * sh[4]);
  ProfScope ps(st, std::string(kCfgName[cfg]) + (pre == PRE_NONE ? "" : pre == PRE_LNFOLD ? "+ln" : pre == PRE_LNMOD ? "+lnmod" : pre == PRE_ELU ? "+elu" : "+addsilu") +
  (a.wfmt == 1 ? "+q8" : a.wfmt == 2 ? "+b16" : a.wfmt == 3 ? "+split" : "") + "@" + std::to_string(threads), bytes, 2.0 * M * N * K);
  launch_by_cfg(st, a, pre, cfg);
}

// k3 conv + ELU + 1x1 conv + skip of a SEANet residual block in one launch (a = the k3 conv's arguments with Y / R
// already describing the block's output and skip input)
static bool resblock_fusable(const Lin &A, const Lin &Bl, int MT) {
  return !A.wq && !Bl.wq && A.bias && Bl.bias && Bl.ntaps == 1 && Bl.KF == A.NT && A.KF % 2 == 0 && MT >= 4 &&
         ((A.NT == 2 && Bl.NT == 4) || (A.NT == 4 && Bl.NT == 8));
}
static void launch_resblock(hipStream_t st, GemmArgs a, const Lin &Bl, int pre = PRE_NONE) {
  a.zeros = g_zeros;
  a.W2 = Bl.w; a.bias2 = Bl.bias;
  const double M = a.M, K = a.KF * 16.0, N = a.NT * 16.0, N2 = Bl.NT * 16.0;
  ProfScope ps(st, std::string(a.NT == 2 ? "resblock<2,4>" : "resblock<4,8>") + (pre == PRE_ELU ? "+elu" : "") + "@" + std::to_string((long)cdiv(a.MT, 4) * 256),
               4.0 * (N * K + N2 * N + M * a.CF * 16.0 + (pre == PRE_ELU ? 1.0 : 2.0) * M * N2), 2.0 * M * N * K + 2.0 * M * N2 * N);
  dim3 grid(1, cdiv(a.MT, 4));
  if (pre == PRE_ELU) {
    if (a.NT == 2) LDS_LAUNCH((gemm_lds_kernel<4, 2, 2, PRE_ELU, 2, 4>), grid, dim3(256), lds_pad(24 * 1024), st, a);
    else LDS_LAUNCH((gemm_lds_kernel<4, 4, 2, PRE_ELU, 2, 8>), grid, dim3(256), lds_pad(32 * 1024), st, a);
    return;
  }
  if (a.NT == 2) LDS_LAUNCH((gemm_lds_kernel<4, 2, 2, PRE_NONE, 2, 4>), grid, dim3(256), lds_pad(24 * 1024), st, a);
  else LDS_LAUNCH((gemm_lds_kernel<4, 4, 2, PRE_NONE, 2, 8>), grid, dim3(256), lds_pad(32 * 1024), st, a);
}

static void bind_engine(ptts_engine *e) {
  g_zeros = e->zeros;
  g_tuner = e->tuner;
  g_prof = &e->prof;
  g_krot = e->opt_k_rotate;
}

static GemmArgs mk_gemm(const Lin &L, const float *X, int XF, int MT, int M) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.W = L.w;
  a.Wq = L.wq ? L.wq : (const uint8_t *)L.wb16;
  a.wfmt = L.wq ? 1 : L.wb16 ? 2 : 0;
  if (g_use_split && L.wsh) {  // codec launches of a PTTS_CODEC_SPLIT engine: hi image in Wq, lo image in W
    a.Wq = (const uint8_t *)L.wsh;
    a.W = (const float *)L.wsl;
    a.wfmt = 3;
  }
  a.wscale = L.wscale;
  a.ln_g = L.ln_g;
  a.bias = L.bias;
  a.ln_s = L.ln_s;
  a.ln_c = L.ln_c;
  a.ln_eps = 1e-5f;  // nn.LayerNorm(eps=1e-5): mimi_transformer.py:26-27, flow_lm.py:89
  a.NT = L.NT;
  a.KF = L.KF;
  a.CF = L.CF;
  a.ntaps = L.ntaps;
  a.X = X;
  a.XF = XF;
  a.MT = MT;
  a.M = M;
  a.T = 16;
  a.xstride = 1;
  a.halo = L.ntaps - 1;  // streaming causal conv: kernel - 1 rows of left context (stride 1)
  a.halo_mode = 0;
  a.epi = EPI_STORE;
  a.act = ACT_NONE;
  return a;
}

// waves per (sequence, head) in the decode attention: enough to reach the wave target, at most 8 (one workgroup)
static int decode_attn_waves(int BH) {
  static const int forced = [] { const char *v = getenv("PTTS_ATTN_NW"); return v ? atoi(v) : 0; }();
  if (forced) return forced;
  int nw = 1;
  while (nw < 8 && BH * nw * 2 <= 1024) nw *= 2;
  return nw;
}
static int attn_wave_target() {
  static int t = [] { const char *v = getenv("PTTS_ATTN_WAVES"); return v ? atoi(v) : 1024; }();
  return t;
}
// waves per workgroup of attn_kernel (they split the workgroup's key tiles and merge in LDS, no combine launch).
// Only for small launches: on the codec frame at batch 64 (512 (sequence, head) pairs, 17 key tiles) 4 waves x 1 split
// is faster alone (18.2 us against 19.8 us + the combine launch, tests/hip/sweep_attn.hip) but SLOWER in the two-stream
// pipeline (0.958 vs 0.947 ms per step, 2 waves 0.963 vs 0.955; tools/ab.sh env PTTS_ATTN_KERNEL_NW): the extra resident waves delay
// the FlowLM stream's kernels.  At batch 8 / 1 it saves 1.5 / 1.1 us per layer.
static int attn_nw(int base) {
  static const int forced = [] { const char *v = getenv("PTTS_ATTN_KERNEL_NW"); return v ? atoi(v) : 0; }();  // A/B knob
  if (forced) return forced;
  return base <= 128 ? 4 : (base <= 256 ? 2 : 1);
}
static int attn_splits(int base, int max_tiles) {
  // `base` = (sequence, head, query block) triples.  Keys are split over workgroups only until ~1024 waves exist
  // (measured at batch 64: 1024 -> 1.139 ms/step, 4096 -> 1.168, 8192 -> 1.211; more splits only add combine launches);
  // a wave never gets less than ~1 key tile.  PTTS_ATTN_WAVES overrides the target for experiments.
  const int nw = attn_nw(base);
  const int tiles = cdiv(max_tiles, nw);
  int s = std::max(1, cdiv(attn_wave_target(), std::max(1, base * nw)));
  return std::max(1, std::min(s, tiles));
}
// attn_cascade_kernel<R, PW, D, NS> shapes selectable with "prefix_cascade" / PTTS_CASCADE besides the default <4, 2, 3, 1>
// (the parity tests run these; all shapes measured are in profiles/r03_experiments.txt: more waves per workgroup or a fourth
// register tile lose beside the codec stream)
#define CASC_SHAPES \
  CASC(442, 4, 4, 2, 2) CASC(222, 2, 2, 2, 2) CASC(42, 4, 2, 2, 1) CASC(44, 4, 4, 2, 1) CASC(84, 8, 4, 2, 1) CASC(22, 2, 2, 2, 1)
static void launch_attn(hipStream_t st, const AttnArgs &at, int BH) {
  const dim3 grid(BH, at.QB, at.splits);
  const int nw = attn_nw(BH * at.QB);
  // large launches (one wave per workgroup) keep two register tiles instead of three: 32 registers less per wave, 0.854 ->
  // 0.850 ms per pipelined step at batch 64 (tools/ab.sh env PTTS_ATTN_DEPTH)
  static const int depth = [] { const char *v = getenv("PTTS_ATTN_DEPTH"); return v ? atoi(v) : 2; }();  // A/B knob
  if (nw == 4) attn_kernel<4><<<grid, 256, 0, st>>>(at);
  else if (nw == 2) attn_kernel<2><<<grid, 128, 0, st>>>(at);
  else if (depth == 2) attn_kernel<1, 2><<<grid, 64, 0, st>>>(at);
  else attn_kernel<1><<<grid, 64, 0, st>>>(at);
}

// One pre-LN transformer layer on M rows (reference mimi_transformer.py:39-54, transformer.py:135-158)
struct TrCtx {
  int D, H, FF, MT, M, Tq, QB, cap, ring, ctx, splits;
  float *x_in;       // residual stream input (FM, F = D/16)
  float *x;          // residual stream after attention (may equal x_in)
  float *x_out;      // residual stream output of the layer (dbl-buffered if out_ds != 0)
  long out_ds;
  const int *par;
  float *h, *ao, *ff, *q, *part;
  float *Kc, *Vc;
  const int *offset;
  const KvPrefix *pre = nullptr;  // shared voice prefixes of the sequences (FlowLM states) or null
  int layer = 0;
  int cascade = 0;                // decode steps: attn_cascade_kernel tile shape (10 R + PW) or 0
  const float *rope;  // [M][32][2]
  double kv_keys;  // sum over sequences of the keys attended (profiling only)
  const char *tag;
};

static void run_tr_layer(hipStream_t st, const TrLayer &T, const TrCtx &c) {
  const int DF = c.D / 16;
  const std::string tg(c.tag);
  std::string s1 = tg + ".ln1", s2 = tg + ".qkv", s3 = tg + ".attn", s4 = tg + ".out", s5 = tg + ".ln2", s6 = tg + ".ff1", s7 = tg + ".ff2";
  // norm1 is folded into the QKV projection, norm2 into linear1 (PRE_LNFOLD): no LayerNorm launches
  SITE(s2.c_str());
  GemmArgs a = mk_gemm(T.qkv, c.x_in, DF, c.MT, c.M);
  a.epi = EPI_QKV;
  a.Q = c.q; a.Kc = c.Kc; a.Vc = c.Vc; a.offset = c.offset; a.rope = c.rope;
  a.H = c.H; a.Tq = c.Tq; a.QB = c.QB; a.cap = c.cap; a.ring = c.ring;
  launch_gemm(st, a, PRE_LNFOLD);
  AttnArgs at;
  at.Q = c.q; at.Kc = c.Kc; at.Vc = c.Vc; at.offset = c.offset; at.H = c.H; at.Tq = c.Tq; at.QB = c.QB;
  at.cap = c.cap; at.ring = c.ring; at.ctx = c.ctx; at.splits = c.splits; at.part = c.part; at.Y = c.ao; at.YF = DF; at.h16 = 0;
  at.pre = c.pre; at.layer = c.layer;
  const int BH = (c.M / c.Tq) * c.H;
  SITE(s3.c_str());
  {
    // K and V rows of every attended key once per head + q in + o out
    const int nseq = BH / c.H;
    at.nseq = nseq;
    const bool casc = c.Tq == 1 && c.cascade && at.pre && c.splits == 1 && !c.ring && c.ctx <= 0 && nseq >= 16;
    int casc_r = 4, casc_threads = 64 * 6;  // the default shape; others are A/B knobs
    switch (c.cascade) {
#define CASC(code, R, PW, D, NS) case code: casc_r = R; casc_threads = 64 * (R * NS + PW); break;
      CASC_SHAPES
#undef CASC
    }
    // bytes: SURVEY 8d's per-sequence figure (every sequence reads all of its keys), also for the cascade kernel, which
    // fetches a shared prefix once per R sequences - bench.py reports the unique bytes next to it
    ProfScope ps(st, casc ? "attn_cascade@" + std::to_string((long)cdiv(nseq, casc_r) * c.H * casc_threads)
                          : std::string(c.Tq == 1 ? "attn_decode" : "attn") + "@" + std::to_string((long)BH * c.QB * c.splits * 64 * (c.Tq == 1 ? decode_attn_waves(BH) : attn_nw(BH * c.QB))),
                 c.kv_keys * c.H * 64 * 4 * 2 + 8.0 * c.M * c.D, 4.0 * c.kv_keys * c.H * 64 * std::min(c.Tq, 16));
    if (c.Tq == 1) {
      // one query: vector ALU + wave reductions.  The keys of a (sequence, head) are split over the nw waves of ONE
      // workgroup and merged in LDS, so small batches reach ~1024 waves without partial buffers or a combine launch
      const int nw = decode_attn_waves(BH);
      if (casc) {
        // sequences cloned from one voice: prefix keys as MFMA tiles shared by R sequences, private keys per sequence,
        // merged in LDS (attn_cascade_kernel).  Tile shape: tools/ab.sh env PTTS_CASCADE
        switch (c.cascade) {
#define CASC(code, R, PW, D, NS) case code: attn_cascade_kernel<R, PW, D, NS><<<cdiv(nseq, R) * c.H, 64 * (R * NS + PW), 0, st>>>(at); break;
          CASC_SHAPES
          default: attn_cascade_kernel<4, 2, 3, 1><<<cdiv(nseq, 4) * c.H, 64 * 6, 0, st>>>(at); break;
#undef CASC
        }
      } else
      // the row-state kernel (no cross-row traffic in its loop).  Small batches: three register tiles (6.0 vs 7.2 us per
      // layer at batch 1, 221 keys).  >= 1024 (sequence, head) pairs: TWO register tiles - alone it streams at the rate of
      // the first kernel (attn_decode_kernel, 188 VGPRs, still selectable with PTTS_ATTN_V=1), but at ~110 registers per wave
      // it leaves the codec stream its occupancy: 0.904 -> 0.877 ms per pipelined step at batch 64 (tools/ab.sh env
      // PTTS_ATTN_V; three tiles: 0.881)
      if (nw >= 8) attn_decode2_kernel<8, 3><<<dim3(BH, 1, c.splits), 512, 0, st>>>(at);
      else if (nw == 4) attn_decode2_kernel<4, 3><<<dim3(BH, 1, c.splits), 256, 0, st>>>(at);
      else if (nw == 2) attn_decode2_kernel<2, 3><<<dim3(BH, 1, c.splits), 128, 0, st>>>(at);
      else {
        static const int v = [] { const char *e = getenv("PTTS_ATTN_V"); return e ? atoi(e) : 2; }();  // A/B knob
        if (v == 1) attn_decode_kernel<1><<<dim3(BH, 1, c.splits), 64, 0, st>>>(at);
        else if (v == 3) attn_decode2_kernel<1, 3><<<dim3(BH, 1, c.splits), 64, 0, st>>>(at);
        else attn_decode2_kernel<1, 2><<<dim3(BH, 1, c.splits), 64, 0, st>>>(at);
      }
    }
    else launch_attn(st, at, BH);
  }
  if (c.splits > 1) {
    ProfScope ps(st, "attn_combine", (double)BH * c.QB * c.splits * 16 * ATT_PSTRIDE * 4, 0);
    attn_combine_kernel<<<dim3(BH, c.QB), 256, 0, st>>>(at);
  }
  SITE(s4.c_str());
  a = mk_gemm(T.out, c.ao, DF, c.MT, c.M);
  a.epi = EPI_RES; a.R = c.x_in; a.RF = DF; a.Y = c.x; a.YF = DF; a.ls = T.ls1;
  launch_gemm(st, a, PRE_NONE);
  SITE(s6.c_str());
  a = mk_gemm(T.ff1, c.x, DF, c.MT, c.M);
  a.epi = EPI_STORE; a.act = ACT_GELU; a.Y = c.ff; a.YF = c.FF / 16;
  launch_gemm(st, a, PRE_LNFOLD);
  SITE(s7.c_str());
  a = mk_gemm(T.ff2, c.ff, c.FF / 16, c.MT, c.M);
  a.epi = EPI_RES; a.R = c.x; a.RF = DF; a.Y = c.x_out; a.YF = DF; a.Ydstride = c.out_ds; a.par = c.par; a.ls = T.ls2;
  launch_gemm(st, a, PRE_NONE);
  SITE("");
}

// ------------------------------------------------------------------------------------------------
extern "C" int ptts_abi_version(void) { return PTTS_ABI_VERSION; }
extern "C" const char *ptts_last_error(void) { return g_err.c_str(); }

static int seanet_check(const ptts_config &c) {
  // an untrusted config (a packed-engine file) reaches this function: no division before the divisors are checked
  if (c.d_model < 64 || c.num_heads < 1 || c.num_layers < 1 || c.m_dim < 64 || c.m_heads < 1 || c.m_layers < 1 || c.ldim < 16 ||
      c.flow_dim < 16 || c.flow_depth < 1 || c.ff_dim < 16 || c.m_ff < 16 || c.n_filters < 1 || c.compress < 1 ||
      c.kernel_size < 1 || c.res_kernel_size < 1 || c.last_kernel_size < 1 || c.ratios[0] < 1 || c.ratios[1] < 1 || c.ratios[2] < 1)
    return fail(-4, "model dimensions must be positive");
  if (c.d_model % 64 || c.d_model / c.num_heads != 64 || c.d_model % c.num_heads) return fail(-4, "FlowLM head dim must be 64");
  if (c.m_dim % c.m_heads || c.m_dim / c.m_heads != 64) return fail(-4, "Mimi head dim must be 64");
  if (c.d_model > 1024 || c.m_dim > 1024 || c.flow_dim > 1024) return fail(-4, "LayerNorm width > 1024 unsupported");
  if (c.ldim % 16 || c.flow_dim % 16 || c.ff_dim % 16 || c.m_ff % 16) return fail(-4, "dims must be multiples of 16");
  if (c.upsample_stride != 16) return fail(-4, "upsample stride must be 16 (one row tile per frame)");
  if ((c.n_filters / c.compress) % 16) return fail(-4, "n_filters/compress must be a multiple of 16");
  return 0;
}

static int build_engine(ptts_engine *e, const ptts_tensor *tensors, int32_t n);
static int build_fp8_codec(ptts_engine *e);
static int sync_host_tables(ptts_engine *e);

// device properties + the option defaults from the environment: shared by ptts_create_ex and ptts_create_from_file
static int init_engine_options(ptts_engine *e, int device) {
  e->device = device;
  e->tuner = new Tuner();
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  e->n_cus = std::max(1, prop.multiProcessorCount);
  if (const char *v = getenv("PTTS_FLOW_CLUSTER")) e->opt_flow_cluster = atoi(v) != 0;
  if (const char *v = getenv("PTTS_LM_CLUSTER")) e->opt_lm_cluster = atoi(v) != 0;
  if (const char *v = getenv("PTTS_K_ROTATE")) e->opt_k_rotate = atoi(v) != 0;
  if (const char *v = getenv("PTTS_FUSE_RES")) e->opt_fuse_res = atoi(v) != 0;
  if (const char *v = getenv("PTTS_SHARE_PREFIX")) e->opt_share_prefix = atoi(v) != 0;
  if (const char *v = getenv("PTTS_SINGLE_STORE")) e->opt_single_store = atoi(v) != 0;
  if (const char *v = getenv("PTTS_FUSE_PCM")) e->opt_fuse_pcm = atoi(v) != 0;
  if (const char *v = getenv("PTTS_CASCADE")) e->opt_cascade = atoi(v);
  if (const char *v = getenv("PTTS_FLOW_MAX_CUS")) e->opt_flow_max_cus = std::max(8, atoi(v));
  e->opt_flow_max_cus = std::min(e->opt_flow_max_cus, e->n_cus);  // a cooperative grid never exceeds the device
  if (const char *v = getenv("PTTS_CODEC_LDS_TARGET")) e->opt_codec_lds_target = std::max(0, std::min(atoi(v), 64 * 1024));
  return 0;
}

extern "C" int ptts_create(const ptts_config *cfg, const ptts_tensor *tensors, int32_t n, int32_t device,
                           ptts_engine **out) {
  return ptts_create_ex(cfg, tensors, n, device, 0, out);
}

extern "C" int ptts_create_ex(const ptts_config *cfg, const ptts_tensor *tensors, int32_t n, int32_t device,
                              int32_t quant_flags, ptts_engine **out) {
  if (!cfg || !tensors || !out) return fail(-1, "null argument");
  if (quant_flags & ~(PTTS_QUANT_ATTENTION | PTTS_QUANT_FFN | PTTS_CODEC_BF16 | PTTS_CODEC_FP8 | PTTS_LM_BF16 | PTTS_CODEC_SPLIT)) return fail(-1, "unknown quantisation group");
  if ((quant_flags & PTTS_CODEC_BF16) && (quant_flags & PTTS_CODEC_FP8)) return fail(-1, "PTTS_CODEC_BF16 and PTTS_CODEC_FP8 are exclusive");
  if ((quant_flags & PTTS_CODEC_SPLIT) && (quant_flags & (PTTS_CODEC_BF16 | PTTS_CODEC_FP8))) return fail(-1, "PTTS_CODEC_SPLIT and the bf16 / fp8 codec are exclusive");
  if ((quant_flags & PTTS_LM_BF16) && (quant_flags & (PTTS_QUANT_ATTENTION | PTTS_QUANT_FFN))) return fail(-1, "PTTS_LM_BF16 and the int8 groups are exclusive");
  CHK(seanet_check(*cfg));
  HIPCHK(hipSetDevice(device));
  ptts_engine *e = new ptts_engine();
  e->cfg = *cfg;
  e->quant_flags = quant_flags;
  int rc = init_engine_options(e, device);
  if (rc == 0) rc = build_engine(e, tensors, n);
  if (rc < 0) {  // missing / ill-shaped tensor, HIP error: release what was built so far
    const std::string msg = g_err;
    ptts_destroy(e);
    return fail(rc, msg);
  }
  *out = e;
  return 0;
}

static int build_engine(ptts_engine *e, const ptts_tensor *tensors, int32_t n) {
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  AllocScope alloc_scope(e->stream);
  HIPCHK(hipEventCreate(&e->ev0));
  HIPCHK(hipEventCreate(&e->ev1));
  for (int i = 0; i < n; ++i) e->tmap[tensors[i].name] = &tensors[i];
  const ptts_config &c = e->cfg;
  const int D = c.d_model, FD = c.flow_dim;
  std::string p = "flow_lm.";
  CHK(copy_vec(e, p + "bos_emb", c.ldim, &e->bos));
  CHK(make_freq(e, &e->freq_lm, c.max_period));
  CHK(pack_lin(e, &e->in_linear, {{p + "input_linear.weight", "", D}}, c.ldim, 1));
  e->lm.resize(c.num_layers);
  for (int l = 0; l < c.num_layers; ++l)
    CHK(pack_tr_layer(e, &e->lm[l], p + "transformer.layers." + std::to_string(l), D, c.ff_dim, false, e->quant_flags));
  CHK(copy_vec(e, p + "out_norm.weight", D, &e->outnorm_w));
  CHK(copy_vec(e, p + "out_norm.bias", D, &e->outnorm_b));
  std::string f = p + "flow_net.";
  // head = [cond_embed ; out_eos]: both read the out_norm output (flow_lm.py:129, mlp.py:209)
  CHK(pack_lin(e, &e->head, {{f + "cond_embed.weight", f + "cond_embed.bias", FD}, {p + "out_eos.weight", p + "out_eos.bias", 1}}, D, 1,
               0, 0, 0, p + "out_norm.weight", p + "out_norm.bias"));
  {
    std::vector<PackPart> parts;
    for (int i = 0; i < c.flow_depth; ++i) {
      std::string r = f + "res_blocks." + std::to_string(i) + ".adaLN_modulation.1.";
      parts.push_back({r + "weight", r + "bias", 3 * FD});
    }
    parts.push_back({f + "final_layer.adaLN_modulation.1.weight", f + "final_layer.adaLN_modulation.1.bias", 2 * FD});
    CHK(pack_lin(e, &e->adaln, parts, FD, 1));
  }
  CHK(pack_lin(e, &e->input_proj, {{f + "input_proj.weight", f + "input_proj.bias", FD}}, c.ldim, 1));
  e->res.resize(c.flow_depth);
  for (int i = 0; i < c.flow_depth; ++i) {
    std::string r = f + "res_blocks." + std::to_string(i) + ".";
    CHK(copy_vec(e, r + "in_ln.weight", FD, &e->res[i].ln_w));
    CHK(copy_vec(e, r + "in_ln.bias", FD, &e->res[i].ln_b));
    CHK(pack_lin(e, &e->res[i].l0, {{r + "mlp.0.weight", r + "mlp.0.bias", FD}}, FD, 1));
    CHK(pack_lin(e, &e->res[i].l2, {{r + "mlp.2.weight", r + "mlp.2.bias", FD}}, FD, 1));
  }
  CHK(pack_lin(e, &e->fin, {{f + "final_layer.linear.weight", f + "final_layer.linear.bias", c.ldim}}, FD, 1));
  for (int i = 0; i < 2; ++i) {
    std::string t = f + "time_embed." + std::to_string(i) + ".";
    CHK(copy_vec(e, t + "freqs", 128, &e->te_freqs[i]));
    CHK(copy_vec(e, t + "mlp.3.alpha", FD, &e->te_alpha[i]));
    CHK(pack_lin(e, &e->te_l0[i], {{t + "mlp.0.weight", t + "mlp.0.bias", FD}}, 256, 1));
    CHK(pack_lin(e, &e->te_l2[i], {{t + "mlp.2.weight", t + "mlp.2.bias", FD}}, FD, 1));
  }
  CHK(dallocT(e, &e->te_scratch, (size_t)4 * 1024 * 16));
  e->lm_bytes = e->in_linear.bytes() + e->head.bytes() + e->adaln.bytes() + e->input_proj.bytes() + e->fin.bytes();
  for (auto &L : e->lm) e->lm_bytes += L.qkv.bytes() + L.out.bytes() + L.ff1.bytes() + L.ff2.bytes();
  for (auto &R : e->res) e->lm_bytes += R.l0.bytes() + R.l2.bytes();
  {
    // layer table of the single-launch transformer stack: fp32 weights, no LayerScale, head dim 64 (4 column tiles per
    // head), linear1 at most 4 column tiles per workgroup
    const int DF = D / 16, FFF = c.ff_dim / 16;
    bool ok = DF <= 64 && DF % 4 == 0 && c.num_heads * 4 == DF && FFF % DF == 0 && FFF / DF <= 4 && c.num_layers <= 64;
    for (auto &L : e->lm) ok = ok && L.qkv.w && L.out.w && L.ff1.w && L.ff2.w && !L.ls1 && !L.ls2 && L.qkv.ln_s && L.ff1.ln_s;
    if (ok) {
      std::vector<LmLayerP> h(c.num_layers);
      for (int l = 0; l < c.num_layers; ++l) {
        const TrLayer &T = e->lm[l];
        h[l] = LmLayerP{T.qkv.w, T.qkv.ln_s, T.qkv.ln_c, T.out.w, T.ff1.w, T.ff1.ln_s, T.ff1.ln_c, T.ff2.w};
      }
      CHK(dalloc(e, (void **)&e->lm_table, h.size() * sizeof(LmLayerP)));
      HIPCHK(hipStreamSynchronize(e->stream));
      HIPCHK(hipMemcpy(e->lm_table, h.data(), h.size() * sizeof(LmLayerP), hipMemcpyHostToDevice));
    }
  }

  // ---- Mimi decode side
  CHK(copy_vec(e, p + "emb_std", c.ldim, &e->emb_std));
  CHK(copy_vec(e, p + "emb_mean", c.ldim, &e->emb_mean));
  const int C = c.m_dim;
  CHK(copy_vec(e, "mimi.quantizer.output_proj.weight", (int64_t)C * c.ldim, &e->quant_w));
  CHK(copy_vec(e, "mimi.upsample.convtr.convtr.weight", (int64_t)C * 2 * c.upsample_stride, &e->up_w));
  CHK(make_freq(e, &e->freq_mimi, c.m_max_period));
  e->mm.resize(c.m_layers);
  for (int l = 0; l < c.m_layers; ++l)
    CHK(pack_tr_layer(e, &e->mm[l], "mimi.decoder_transformer.transformer.layers." + std::to_string(l), C, c.m_ff, true));
  e->ring = c.m_context > 0 ? (cdiv(c.m_context - 1, 16) + 1) * 16 : 0;
  if (e->ring == 0) return fail(-4, "Mimi decoder transformer needs a finite context");
  int mult = 8, idx = 1;
  const int nf = c.n_filters;
  CHK(pack_lin(e, &e->conv0, {{"mimi.decoder.model.0.conv.weight", "mimi.decoder.model.0.conv.bias", mult * nf}}, C, c.kernel_size));
  if (C != mult * nf && false) return fail(-4, "unexpected seanet dims");
  for (int i = 0; i < 3; ++i) {
    const int cin = mult * nf, cout = cin / 2, s = c.ratios[i], hid = cout / c.compress;
    std::string m = "mimi.decoder.model." + std::to_string(idx + 1);
    CHK(pack_lin(e, &e->convtr[i], {{m + ".convtr.weight", m + ".convtr.bias", s * cout}}, cin, 2, 1, cout, s));
    std::string r = "mimi.decoder.model." + std::to_string(idx + 2);
    CHK(pack_lin(e, &e->res_a[i], {{r + ".block.1.conv.weight", r + ".block.1.conv.bias", hid}}, cout, c.res_kernel_size));
    CHK(pack_lin(e, &e->res_b[i], {{r + ".block.3.conv.weight", r + ".block.3.conv.bias", cout}}, hid, 1));
    idx += 3;
    mult /= 2;
  }
  {
    std::string m = "mimi.decoder.model." + std::to_string(idx + 1);
    CHK(pack_lin(e, &e->conv_last, {{m + ".conv.weight", m + ".conv.bias", 1}}, nf, c.last_kernel_size));
  }
  CHK(dallocT(e, &e->zeros, 64));
  if (e->quant_flags & (PTTS_CODEC_BF16 | PTTS_CODEC_FP8)) {
    // bf16 images of every codec GEMM (reference modules: mimi_transformer.py:12-54, seanet.py:141-180, conv.py:93-163)
    for (int l = 0; l < c.m_layers; ++l) {
      const std::string q = "mimi.decoder_transformer.transformer.layers." + std::to_string(l);
      TrLayer &T = e->mm[l];
      CHK(pack_lin_h(e, &T.qkv, q + ".self_attn.in_proj.weight", 3 * C, C, 1, 0, 0, 0, q + ".norm1.weight"));
      CHK(pack_lin_h(e, &T.out, q + ".self_attn.out_proj.weight", C, C, 1));
      CHK(pack_lin_h(e, &T.ff1, q + ".linear1.weight", c.m_ff, C, 1, 0, 0, 0, q + ".norm2.weight"));
      CHK(pack_lin_h(e, &T.ff2, q + ".linear2.weight", C, c.m_ff, 1));
    }
    CHK(pack_lin_h(e, &e->conv0, "mimi.decoder.model.0.conv.weight", 8 * nf, C, c.kernel_size));
    int m2 = 8, i2 = 1;
    for (int i = 0; i < 3; ++i) {
      const int cin = m2 * nf, cout = cin / 2, sr = c.ratios[i], hid = cout / c.compress;
      CHK(pack_lin_h(e, &e->convtr[i], "mimi.decoder.model." + std::to_string(i2 + 1) + ".convtr.weight", sr * cout, cin, 2, 1, cout, sr));
      const std::string r = "mimi.decoder.model." + std::to_string(i2 + 2);
      CHK(pack_lin_h(e, &e->res_a[i], r + ".block.1.conv.weight", hid, cout, c.res_kernel_size));
      CHK(pack_lin_h(e, &e->res_b[i], r + ".block.3.conv.weight", cout, hid, 1));
      i2 += 3;
      m2 /= 2;
    }
    const std::string lc = "mimi.decoder.model." + std::to_string(i2 + 1);
    CHK(copy_vec(e, lc + ".conv.weight", (int64_t)nf * c.last_kernel_size, &e->conv_last_w));
    CHK(copy_vec(e, lc + ".conv.bias", 1, &e->conv_last_b, 4));
    e->codec_bf16 = true;
    e->mimi_bytes_h += (int64_t)C * c.ldim * 4;
  }
  if (e->quant_flags & PTTS_CODEC_FP8) CHK(build_fp8_codec(e));
  if (e->quant_flags & PTTS_CODEC_SPLIT) {
    std::vector<Lin *> ls = {&e->conv0};
    for (auto &T : e->mm) { ls.push_back(&T.qkv); ls.push_back(&T.out); ls.push_back(&T.ff1); ls.push_back(&T.ff2); }
    for (int i = 0; i < 3; ++i) { ls.push_back(&e->convtr[i]); ls.push_back(&e->res_a[i]); ls.push_back(&e->res_b[i]); }
    for (Lin *L : ls) {
      if (L->KF % 2) return fail(-4, "split-bf16 codec: every codec matrix needs an even number of 16-wide k-fragments");
      CHK(dalloc(e, &L->wsh, (size_t)L->NT * L->KF * 512));
      CHK(dalloc(e, &L->wsl, (size_t)L->NT * L->KF * 512));
      pack_weight_split(e->stream, L->w, L->wsh, L->wsl, L->NT, L->KF);
    }
    HIPCHK(hipGetLastError());
    e->codec_split = true;
  }
  if (e->blob_dummy ? e->blob_has_encoder != 0
                    : (e->tmap.count("mimi.encoder.model.0.conv.weight") && e->tmap.count("flow_lm.speaker_proj_weight"))) {
    // reference mimi.py:96-119, seanet.py:63-104, resample.py:7-29, tts_model.py:379-388
    int em = 1, eidx = 1;
    CHK(pack_lin(e, &e->enc_conv0, {{"mimi.encoder.model.0.conv.weight", "mimi.encoder.model.0.conv.bias", nf}}, 16,
                 c.kernel_size, 0, 0, 0, "", "", 1));  // mono input padded to 16 channels
    for (int i = 0; i < 3; ++i) {
      const int dim = em * nf, r = c.ratios[2 - i], hid = dim / c.compress;
      std::string rb = "mimi.encoder.model." + std::to_string(eidx);
      CHK(pack_lin(e, &e->enc_res_a[i], {{rb + ".block.1.conv.weight", rb + ".block.1.conv.bias", hid}}, dim, c.res_kernel_size));
      CHK(pack_lin(e, &e->enc_res_b[i], {{rb + ".block.3.conv.weight", rb + ".block.3.conv.bias", dim}}, hid, 1));
      std::string dn = "mimi.encoder.model." + std::to_string(eidx + 2);
      CHK(pack_lin(e, &e->enc_down[i], {{dn + ".conv.weight", dn + ".conv.bias", 2 * dim}}, dim, 2 * r));
      e->enc_down[i].stride = r;
      eidx += 3;
      em *= 2;
    }
    std::string fn = "mimi.encoder.model." + std::to_string(eidx + 1);
    CHK(pack_lin(e, &e->enc_final, {{fn + ".conv.weight", fn + ".conv.bias", C}}, em * nf, c.last_kernel_size));
    e->enc_tr.resize(c.m_layers);
    for (int l = 0; l < c.m_layers; ++l)
      CHK(pack_tr_layer(e, &e->enc_tr[l], "mimi.encoder_transformer.transformer.layers." + std::to_string(l), C, c.m_ff, true));
    CHK(pack_lin(e, &e->enc_downsample, {{"mimi.downsample.conv.conv.weight", "", c.ldim}}, C, 2 * c.upsample_stride));
    e->enc_downsample.stride = c.upsample_stride;
    CHK(pack_lin(e, &e->speaker_proj, {{"flow_lm.speaker_proj_weight", "", D}}, c.ldim, 1));
    e->has_encoder = true;
  }
  e->mimi_bytes = (int64_t)C * c.ldim * 4 + e->conv0.bytes() + e->conv_last.bytes();
  for (auto &L : e->mm) e->mimi_bytes += L.qkv.bytes() + L.out.bytes() + L.ff1.bytes() + L.ff2.bytes();
  for (int i = 0; i < 3; ++i) e->mimi_bytes += e->convtr[i].bytes() + e->res_a[i].bytes() + e->res_b[i].bytes();
  HIPCHK(hipStreamSynchronize(e->stream));
  e->tmap.clear();
  e->n_build_allocs = e->allocs.size();
  return sync_host_tables(e);
}

// PTTS_CODEC_FP8: e4m3 images of the SEANet convolutions that run on the fp8 MFMA, and the static activation scales from a
// calibration run of the bf16 codec (8 sequences x 6 frames of N(0, 1) latents; scale = 2 x amax / 448, saturating).
__global__ void calib_latent_kernel(float *lat, int n, unsigned frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lat[i] = counter_normal(0x5EEDull, frame, (unsigned)i);
}
static int mimi_enqueue_h(hipStream_t st, ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm);
static int build_fp8_codec(ptts_engine *e) {
  const ptts_config &c = e->cfg;
  const int nf = c.n_filters;
  hipStream_t st = e->stream;
  int mult = 8, idx = 1;
  e->mimi_bytes_f8 = e->mimi_bytes_h;
  for (int i = 0; i < 3; ++i) {
    const int cin = mult * nf, cout = cin / 2, sr = c.ratios[i], hid = cout / c.compress;
    if (cin % 32 || cout % 32 || hid % 32) return fail(-4, "fp8 codec path: channel counts must be multiples of 32");
    struct Job { Lin *L; std::string name; int N, C, ntaps, mode, cout, stride; };
    const std::string r = "mimi.decoder.model." + std::to_string(idx + 2);
    const Job jobs[3] = {{&e->convtr[i], "mimi.decoder.model." + std::to_string(idx + 1) + ".convtr.weight", sr * cout, cin, 2, 1, cout, sr},
                         {&e->res_a[i], r + ".block.1.conv.weight", hid, cout, c.res_kernel_size, 0, 0, 0},
                         {&e->res_b[i], r + ".block.3.conv.weight", cout, hid, 1, 0, 0, 0}};
    for (const Job &j : jobs) {
      int err = 0;
      const ptts_tensor *t = find_tensor(e, j.name, -1, &err);
      if (!t) return err;
      const size_t bytes = (size_t)j.L->NT * (j.C / 32) * j.ntaps * 512;
      CHK(dalloc(e, &j.L->wf8, bytes));
      CHK(dallocT(e, &j.L->wscale8, (size_t)j.L->NT * 16));
      pack_weight_f8(st, t->d_data, j.L->wf8, j.L->wscale8, j.N, j.C, j.ntaps, j.mode, j.cout, j.stride);
      HIPCHK(hipGetLastError());
      e->mimi_bytes_f8 += (int64_t)bytes - 2 * (int64_t)bytes;  // e4m3 instead of bf16 for this matrix
    }
    idx += 3;
    mult /= 2;
  }
  CHK(dallocT(e, &e->d_f8s, 16));
  // ---- calibration on the bf16 path
  const int B = 8, frames = 6;
  ptts_mimi_state *ms = nullptr;
  CHK(ptts_mimi_state_create(e, B, &ms));
  float *lat = nullptr, *amax = nullptr;
  int rc = 0;
  if (hipMalloc((void **)&lat, (size_t)B * c.ldim * 4) != hipSuccess || hipMalloc((void **)&amax, 64) != hipSuccess) rc = fail(-2, "hipMalloc (fp8 calibration)");
  if (rc == 0) {
    (void)hipMemsetAsync(amax, 0, 64, st);
    for (int f = 0; f < frames && rc == 0; ++f) {
      calib_latent_kernel<<<cdiv(B * c.ldim, 256), 256, 0, st>>>(lat, B * c.ldim, (unsigned)f);
      rc = mimi_enqueue_h(st, e, ms, lat, nullptr);
      ms->h_frame += 1;
      // both parity halves hold bf16 data in their first half (see mimi_enqueue_h); scan them after every frame
      auto scan = [&](const float *buf, long stride, int slot) {
        for (int par = 0; par < 2; ++par) amax_bf16(st, (const char *)buf + (size_t)par * stride * 4, stride, amax + slot);
      };
      scan(ms->a0, ms->a0_stride, 0);
      for (int i = 0; i < 3; ++i) {
        scan(ms->cbuf[i], ms->c_stride[i], 1 + 3 * i);
        amax_bf16(st, ms->rbuf[i], (long)B * (ms->rows[i + 1] / 16) * 256 * ((8 >> i) * nf / 2 / c.compress / 16), amax + 2 + 3 * i);
        if (i < 2) scan(ms->sbuf[i], ms->s_stride[i], 3 + 3 * i);
      }
    }
  }
  float h[16] = {};
  if (rc == 0 && (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h, amax, 64, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail(-2, "fp8 calibration failed");
  if (rc == 0) {
    for (float &v : h) v = v > 0.f ? 2.0f * v / 448.0f : 1.0f;
    if (hipMemcpy(e->d_f8s, h, 64, hipMemcpyHostToDevice) != hipSuccess) rc = fail(-2, "hipMemcpy (fp8 scales)");
  }
  if (lat) (void)hipFree(lat);
  if (amax) (void)hipFree(amax);
  ptts_mimi_state_destroy(ms);
  CHK(rc);
  e->codec_fp8 = true;
  return 0;
}

// host mirrors of small device tables (after build_engine and again after a packed file has replaced the device images)
static int sync_host_tables(ptts_engine *e) {
  if (e->codec_fp8) {
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->f8s, e->d_f8s, sizeof(e->f8s), hipMemcpyDeviceToHost));
    for (float &v : e->f8s)
      if (!(v > 0.f)) v = 1.0f;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Packed-engine files (SURVEY 8(f).4 "offline packer"): everything build_engine leaves on the device - weights in MFMA
// fragment order, int8 / bf16 images, LayerNorm-fold vectors, small tables - as one file, so a deployment neither
// needs the fp32 checkpoint on the device nor re-packs / re-quantises at every start (reference load-time hook:
// quantization.py:60-88).  Layout: header, sizes, then the allocations in build order, each padded to 256 bytes.
struct PackHeader {
  char magic[8];
  int32_t abi, tune_version;
  ptts_config cfg;
  int32_t quant_flags, has_encoder;
  int64_t n_allocs;
};
static const char kPackMagic[8] = {'P', 'T', 'T', 'S', 'P', 'K', '1', 0};

extern "C" int ptts_engine_save(ptts_engine *e, const char *path) {
  if (!e || !path) return fail(-1, "null argument");
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  FILE *f = fopen(path, "wb");
  if (!f) return fail(-1, std::string("cannot open ") + path);
  PackHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, kPackMagic, 8);
  h.abi = PTTS_ABI_VERSION; h.tune_version = kTuneVersion; h.cfg = e->cfg; h.quant_flags = e->quant_flags;
  h.has_encoder = e->has_encoder ? 1 : 0; h.n_allocs = (int64_t)e->n_build_allocs;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  std::vector<int64_t> sizes;
  for (size_t i = 0; i < e->n_build_allocs; ++i) sizes.push_back((int64_t)e->alloc_bytes[e->allocs[i]]);
  ok = ok && fwrite(sizes.data(), 8, sizes.size(), f) == sizes.size();
  std::vector<char> buf;
  for (size_t i = 0; i < e->n_build_allocs && ok; ++i) {
    const size_t n = (size_t)sizes[i], padded = (n + 255) / 256 * 256;
    buf.assign(padded, 0);
    if (hipMemcpy(buf.data(), e->allocs[i], n, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
    ok = fwrite(buf.data(), 1, padded, f) == padded;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : fail(-2, std::string("writing ") + path + " failed");
}

extern "C" int ptts_create_from_file(const char *path, int32_t device, ptts_engine **out) {
  if (!path || !out) return fail(-1, "null argument");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(-1, std::string("cannot open ") + path);
  PackHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kPackMagic, 8) != 0 || h.abi != PTTS_ABI_VERSION || h.n_allocs < 1 ||
      h.n_allocs > 100000) {
    fclose(f);
    return fail(-3, std::string(path) + " is not a packed engine of this library version");
  }
  std::vector<int64_t> sizes((size_t)h.n_allocs);
  if (fread(sizes.data(), 8, sizes.size(), f) != sizes.size()) { fclose(f); return fail(-3, "truncated packed engine (sizes)"); }
  if (h.quant_flags & ~(PTTS_QUANT_ATTENTION | PTTS_QUANT_FFN | PTTS_CODEC_BF16 | PTTS_CODEC_FP8 | PTTS_LM_BF16 | PTTS_CODEC_SPLIT)) {
    fclose(f);
    return fail(-3, "packed engine carries unknown weight-format flags");
  }
  if (seanet_check(h.cfg) < 0) { fclose(f); return -4; }
  if (hipSetDevice(device) != hipSuccess) { fclose(f); return fail(-2, "hipSetDevice"); }
  // the largest checkpoint tensor any packing kernel reads (a zero stand-in: the images it produces are overwritten)
  const ptts_config &c = h.cfg;
  const int64_t big = std::max<int64_t>({(int64_t)c.ff_dim * c.d_model, (int64_t)3 * c.d_model * c.d_model, (int64_t)c.m_ff * c.m_dim,
                                         (int64_t)c.m_dim * c.m_dim * c.kernel_size, (int64_t)16 * c.n_filters * c.n_filters * 16,
                                         (int64_t)3 * c.flow_dim * c.flow_dim, (int64_t)c.m_dim * c.ldim * 2 * c.upsample_stride}) * 2;
  float *zeros = nullptr;
  if (hipMalloc((void **)&zeros, (size_t)big * 4) != hipSuccess || hipMemset(zeros, 0, (size_t)big * 4) != hipSuccess) {
    (void)hipGetLastError();
    fclose(f);
    return fail(-2, "out of device memory");
  }
  ptts_tensor dummy{"<packed>", zeros, -1};
  ptts_engine *e = new ptts_engine();
  e->cfg = h.cfg;
  e->quant_flags = h.quant_flags;
  e->blob_dummy = &dummy;
  e->blob_has_encoder = h.has_encoder;
  int rc = init_engine_options(e, device);
  if (rc == 0) rc = build_engine(e, nullptr, 0);
  e->blob_dummy = nullptr;
  std::string msg = g_err;
  if (rc == 0 && (int64_t)e->n_build_allocs != h.n_allocs) { rc = -3; msg = "packed engine does not match this build's layout"; }
  std::vector<char> buf;
  for (size_t i = 0; rc == 0 && i < e->n_build_allocs; ++i) {
    const size_t n = (size_t)sizes[i], padded = (n + 255) / 256 * 256;
    if (e->alloc_bytes[e->allocs[i]] != n) { rc = -3; msg = "packed engine does not match this build's layout (allocation size)"; break; }
    buf.resize(padded);
    if (fread(buf.data(), 1, padded, f) != padded) { rc = -3; msg = "truncated packed engine"; break; }
    if (hipMemcpy(e->allocs[i], buf.data(), n, hipMemcpyHostToDevice) != hipSuccess) { rc = -2; msg = "hipMemcpy"; break; }
  }
  fclose(f);
  (void)hipFree(zeros);
  if (rc == 0 && sync_host_tables(e) < 0) { rc = -2; msg = g_err; }
  if (rc < 0) {
    ptts_destroy(e);
    return fail(rc, msg);
  }
  *out = e;
  return 0;
}

extern "C" void ptts_destroy(ptts_engine *e) {
  if (!e) return;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (void *p : e->allocs) hipFree(p);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->stream) hipStreamDestroy(e->stream);
  if (g_tuner == e->tuner) g_tuner = nullptr;
  if (g_prof == &e->prof) g_prof = nullptr;
  for (auto &r : e->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  delete e->tuner;
  delete e;
}

static hipStream_t S(ptts_engine *e, void *stream) { return stream ? (hipStream_t)stream : e->stream; }

// time-embedding constant for a given LSD schedule (reference mlp.py:203-206): computed once on device
static int prepare_lsd(ptts_engine *e, int steps) {
  if (e->tcomb.count(steps)) return 0;
  bind_engine(e);
  if (steps < 1 || steps > 64) return fail(-1, "lsd_decode_steps out of range");
  const int FD = e->cfg.flow_dim;
  float *tab;
  AllocScope alloc_scope(e->stream);
  CHK(dallocT(e, &tab, (size_t)steps * FD));
  hipStream_t st = e->stream;
  float *efm = e->te_scratch, *h = efm + 16 * 1024, *h0 = h + 16 * 1024, *h1 = h0 + 16 * 1024;
  for (int i = 0; i < steps; ++i) {
    const float tv[2] = {(float)((double)i / steps), (float)((double)(i + 1) / steps)};
    float *ho[2] = {h0, h1};
    for (int k = 0; k < 2; ++k) {
      timestep_embed_kernel<<<1, 256, 0, st>>>(e->te_freqs[k], tv[k], efm, 128);
      GemmArgs a = mk_gemm(e->te_l0[k], efm, 16, 1, 1);
      a.act = ACT_SILU; a.Y = h; a.YF = FD / 16;
      launch_gemm(st, a, PRE_NONE);
      a = mk_gemm(e->te_l2[k], h, FD / 16, 1, 1);
      a.Y = ho[k]; a.YF = FD / 16;
      launch_gemm(st, a, PRE_NONE);
    }
    tcomb_kernel<<<1, 64, 0, st>>>(h0, h1, e->te_alpha[0], e->te_alpha[1], tab + (size_t)i * FD, FD);
  }
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  e->tcomb[steps] = tab;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// FlowLM state
static int alloc_scratch(ptts_engine *e, Scratch *s, int B, int Tq, int D, int H, int FF, int cap) {
  const int M = B * Tq;
  s->MT = cdiv(M, 16);
  s->QB = cdiv(Tq, 16);
  const size_t rowt = (size_t)s->MT * 256;
  CHK(dallocT(nullptr, &s->x, rowt * (D / 16)));
  CHK(dallocT(nullptr, &s->h, rowt * (D / 16)));
  CHK(dallocT(nullptr, &s->ao, rowt * (D / 16)));
  CHK(dallocT(nullptr, &s->ff, rowt * (FF / 16)));
  CHK(dallocT(nullptr, &s->q, (size_t)B * H * s->QB * 4 * 256));
  CHK(dallocT(nullptr, &s->rope, (size_t)s->MT * 16 * 64));
  s->splits_cap = attn_splits(B * H * s->QB, cdiv(cap, 16));
  CHK(dallocT(nullptr, &s->part, (size_t)B * H * s->QB * s->splits_cap * 16 * ATT_PSTRIDE));
  return 0;
}
static void free_scratch(Scratch *s) {
  hipFree(s->x); hipFree(s->h); hipFree(s->ao); hipFree(s->ff); hipFree(s->q); hipFree(s->part); hipFree(s->rope);
  *s = Scratch();
}

static int build_lm_state(ptts_engine *e, ptts_lm_state *s);
static int ensure_flow(ptts_engine *e, ptts_lm_state *s, int steps, hipStream_t st);

extern "C" int ptts_lm_state_create(ptts_engine *e, int32_t B, int32_t t_cap, ptts_lm_state **out) {
  if (!e || !out || B < 1 || t_cap < 1) return fail(-1, "bad argument");
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  ptts_lm_state *s = new ptts_lm_state();
  s->e = e;
  s->B = B;
  s->cap = cdiv(t_cap, 16) * 16;
  s->MT = cdiv(B, 16);
  const int rc = build_lm_state(e, s);
  if (rc < 0) {  // typically out of device memory for the KV cache: free the partial state
    const std::string msg = g_err;
    ptts_lm_state_destroy(s);
    return fail(rc, msg);
  }
  *out = s;
  return 0;
}

static int build_lm_state(ptts_engine *e, ptts_lm_state *s) {
  const ptts_config &c = e->cfg;
  const int B = s->B;
  AllocScope alloc_scope(e->stream);
  CHK(dallocT(nullptr, &s->kv, (size_t)c.num_layers * 2 * s->kv_plane()));
  CHK(dallocT(nullptr, &s->offset, B));
  s->h_off.assign(B, 0);
  CHK(dalloc(nullptr, (void **)&s->d_pre, (size_t)B * sizeof(KvPrefix)));
  s->h_pre.assign(B, KvPrefix{nullptr, 0, 0});
  s->pre_owner.assign(B, nullptr);
  CHK(alloc_scratch(e, &s->dec, B, 1, c.d_model, c.num_heads, c.ff_dim, s->cap));
  const size_t rt = (size_t)s->MT * 256;
  const int FD = c.flow_dim;
  CHK(dallocT(nullptr, &s->xlat, rt * (c.ldim / 16)));
  CHK(dallocT(nullptr, &s->latfm, rt * (c.ldim / 16)));
  CHK(dallocT(nullptr, &s->c, rt * (c.d_model / 16)));
  CHK(dallocT(nullptr, &s->ce, rt * (FD / 16)));
  CHK(dallocT(nullptr, &s->ferr, 1));
  CHK(dallocT(nullptr, &s->fx, rt * (FD / 16)));
  CHK(dallocT(nullptr, &s->fh, rt * (FD / 16)));
  CHK(dallocT(nullptr, &s->f1, rt * (FD / 16)));
  CHK(dallocT(nullptr, &s->fstat, (size_t)s->MT * (FD / 16) * 32));
  CHK(dallocT(nullptr, &s->lat, (size_t)B * c.ldim));
  CHK(dallocT(nullptr, &s->lat_prev, (size_t)B * c.ldim));
  CHK(dallocT(nullptr, &s->eos_logit, B));
  CHK(dallocT(nullptr, &s->is_eos, B));
  CHK(dallocT(nullptr, &s->rng_ctr, 1));
  CHK(dallocT(nullptr, &s->active, B));
  s->h_active.assign(B, 1);
  set_int_kernel<<<cdiv(B, 256), 256, 0, e->stream>>>(s->active, B, 1);
  fill_kernel<<<cdiv(B * c.ldim, 256), 256, 0, e->stream>>>(s->lat_prev, (long)B * c.ldim, NAN);
  HIPCHK(hipStreamSynchronize(e->stream));
  CHK(ensure_flow(e, s, 1, e->stream));  // AdaLN modulation buffer + flow-cluster exchange slots for lsd_decode_steps = 1
  return 0;
}

// ---- shared prefixes: bookkeeping (host side; the device table is updated by set_prefix_kernel on the caller's stream)
static void lm_state_free(ptts_lm_state *s) {
  hipFree(s->kv); hipFree(s->offset); hipFree(s->d_pre);
  free_scratch(&s->dec);
  if (s->pre.x) free_scratch(&s->pre);
  hipFree(s->xlat); hipFree(s->latfm); hipFree(s->c); hipFree(s->ce); hipFree(s->mod); hipFree(s->fx);
  hipFree(s->fh); hipFree(s->f1); hipFree(s->lat); hipFree(s->lat_prev); hipFree(s->eos_logit); hipFree(s->is_eos); hipFree(s->rng_ctr); hipFree(s->active); hipFree(s->fstat);
  hipFree(s->fexch); hipFree(s->fflags); hipFree(s->ferr); hipFree(s->lexch); hipFree(s->lflags);
  delete s;
}
static void prefix_release(ptts_lm_state *s, int row) {  // the row stops reading its owner's cache
  ptts_lm_state *o = s->pre_owner[row];
  if (!o) return;
  s->pre_owner[row] = nullptr;
  s->h_pre[row] = KvPrefix{nullptr, 0, 0};
  s->n_pre -= 1;
  o->borrowers -= 1;
  if (o->zombie && o->borrowers == 0) {  // its destroy was deferred until now; the GPU may still be reading it
    hipDeviceSynchronize();
    lm_state_free(o);
  }
}
static void prefix_borrow(ptts_lm_state *s, int row, ptts_lm_state *owner, int len) {
  if (s->pre_owner[row] == owner && s->h_pre[row].len == len) return;
  owner->borrowers += 1;  // before the release: re-borrowing from the same (zombie) owner must not free it in between
  prefix_release(s, row);
  s->pre_owner[row] = owner;
  s->h_pre[row] = KvPrefix{owner->kv, owner->cap, len};
  s->n_pre += 1;
}
// a state whose cache other states read as their prefix must not have it rewritten under them
static int refuse_if_lent(const ptts_lm_state *s, const char *what) {
  if (s->borrowers > 0) return fail(-1, std::string(what) + ": this state's cache is the shared prefix of " + std::to_string(s->borrowers) +
                                           " sequence(s) cloned from it; destroy or re-clone them first");
  return 0;
}

extern "C" void ptts_lm_state_destroy(ptts_lm_state *s) {
  if (!s) return;
  ENGINE_LOCK(s->e);  // the borrow counts of OTHER states (prefix owners) change here
  hipSetDevice(s->e->device);
  hipDeviceSynchronize();
  for (int b = 0; b < (int)s->pre_owner.size(); ++b) prefix_release(s, b);  // (empty after a failed creation)
  if (s->borrowers > 0) {  // clones still read this cache: the memory goes when the last of them lets go
    s->zombie = true;
    return;
  }
  lm_state_free(s);
}

extern "C" int ptts_lm_state_reset(ptts_lm_state *s, void *stream) {
  hipStream_t st = S(s->e, stream);
  ENGINE_LOCK(s->e);
  CHK(refuse_if_lent(s, "reset"));
  for (int b = 0; b < s->B; ++b) prefix_release(s, b);
  set_prefix_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->d_pre, s->B, nullptr, 0, 0);
  set_int_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->offset, s->B, 0);
  fill_kernel<<<cdiv(s->B * s->e->cfg.ldim, 256), 256, 0, st>>>(s->lat_prev, (long)s->B * s->e->cfg.ldim, NAN);
  std::fill(s->h_off.begin(), s->h_off.end(), 0);
  std::fill(s->h_active.begin(), s->h_active.end(), 1);
  set_int_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->active, s->B, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int ptts_lm_state_import(ptts_lm_state *s, int32_t layer, const float *d_cache, int32_t src_batch,
                                    int32_t t, void *stream) {
  const ptts_config &c = s->e->cfg;
  if (layer < 0 || layer >= c.num_layers) return fail(-1, "layer out of range");
  if (t > s->cap) return fail(-5, "import: t exceeds cache capacity");
  if (src_batch != 1 && src_batch != s->B) return fail(-1, "import: src_batch must be 1 or B");
  hipStream_t st = S(s->e, stream);
  ENGINE_LOCK(s->e);
  CHK(refuse_if_lent(s, "import"));
  if (s->n_pre > 0) {  // the imported rows are complete copies
    for (int b = 0; b < s->B; ++b) prefix_release(s, b);
    set_prefix_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->d_pre, s->B, nullptr, 0, 0);
  }
  if (t > 0) {
    long total = 2L * s->B * t * c.num_heads * 16;
    kv_import_kernel<<<cdiv(total, 256), 256, 0, st>>>(d_cache, s->K(layer), s->V(layer), s->B, src_batch, t,
                                                        c.num_heads, s->cap);
  }
  set_int_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->offset, s->B, t);
  std::fill(s->h_off.begin(), s->h_off.end(), t);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int ptts_lm_state_export(ptts_lm_state *s, int32_t layer, float *d_cache, int32_t t, void *stream) {
  const ptts_config &c = s->e->cfg;
  if (layer < 0 || layer >= c.num_layers) return fail(-1, "layer out of range");
  if (t > s->cap) return fail(-5, "export: t exceeds cache capacity");
  hipStream_t st = S(s->e, stream);
  long total = 2L * s->B * t * c.num_heads * 16;
  if (total) kv_export_kernel<<<cdiv(total, 256), 256, 0, st>>>(d_cache, s->K(layer), s->V(layer), s->B, t, c.num_heads, s->cap,
                                                                 s->n_pre > 0 ? s->d_pre : nullptr, layer);
  HIPCHK(hipGetLastError());
  return 0;
}

// KV rows of (src, src_row) -> (dst, row).  With "share_prefix" the destination row BORROWS the leading positions instead
// of copying them: from src itself when src is a one-sequence state (a cached voice state: every clone of it reads the same
// [0, T & ~15) keys, so the clones' attention fetches them once through L2 instead of once per row), or from the state
// src's row already borrows from.  Only the private positions behind the prefix are copied.  Without the option a
// borrowed prefix is materialised into dst.  Host table only; the caller updates dst->d_pre.
static int kv_clone_row(ptts_lm_state *dst, int row, const ptts_lm_state *src, int src_row, int T, hipStream_t st) {
  const ptts_config &c = dst->e->cfg;
  const int planes = c.num_layers * 2, H = c.num_heads;
  ptts_lm_state *owner = src->pre_owner[src_row];
  int plen = owner ? src->h_pre[src_row].len : 0;
  if (plen > T) plen = 0, owner = nullptr;  // cannot happen (offsets only grow past a prefix); a full copy is always right
  bool share = dst->e->opt_share_prefix && !dst->e->opt_lm_cluster && dst != src;  // (the single-launch stack reads own caches only)
  if (share && !owner && src->B == 1 && !src->zombie && T >= 16) owner = const_cast<ptts_lm_state *>(src), plen = T & ~15;
  if (!owner || owner == dst) share = false;
  int t0 = 0;
  if (share) {
    prefix_borrow(dst, row, owner, plen);
    t0 = owner == src ? plen : src->h_pre[src_row].len;
  } else {
    prefix_release(dst, row);
    if (src->pre_owner[src_row]) {  // materialise the part src only borrows
      const ptts_lm_state *o = src->pre_owner[src_row];
      t0 = src->h_pre[src_row].len;
      const long total = (long)planes * H * t0 * 16;
      kv_copy_row_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst->kv, o->kv, planes, H, t0, o->cap, dst->cap, dst->B, row, o->B, 0, 0);
    }
  }
  if (T > t0) {
    const long total = (long)planes * H * (T - t0) * 16;
    kv_copy_row_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst->kv, src->kv, planes, H, T - t0, src->cap, dst->cap, dst->B, row, src->B, src_row, t0);
  }
  return 0;
}
static int upload_prefixes(ptts_lm_state *s, hipStream_t st) {
  bool uniform = true;
  for (int b = 1; b < s->B; ++b) uniform &= s->h_pre[b].kv == s->h_pre[0].kv && s->h_pre[b].len == s->h_pre[0].len && s->h_pre[b].cap == s->h_pre[0].cap;
  if (uniform) {
    set_prefix_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->d_pre, s->B, s->h_pre[0].kv, s->h_pre[0].cap, s->h_pre[0].len);
  } else {
    HIPCHK(hipMemcpyAsync(s->d_pre, s->h_pre.data(), s->B * sizeof(KvPrefix), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // pageable host memory
  }
  return 0;
}

extern "C" int ptts_lm_state_copy(ptts_lm_state *dst, const ptts_lm_state *src, void *stream) {
  if (!dst || !src) return fail(-1, "null state");
  if (dst->e != src->e) return fail(-1, "states belong to different engines");
  if (src->B != 1 && src->B != dst->B) return fail(-1, "copy: src batch must be 1 or equal");
  if (dst == src) return 0;
  const ptts_config &c = dst->e->cfg;
  hipStream_t st = S(dst->e, stream);
  ENGINE_LOCK(dst->e);
  CHK(refuse_if_lent(dst, "copy"));
  const int T = *std::max_element(src->h_off.begin(), src->h_off.end());
  if (T > dst->cap) return fail(-5, "copy: destination capacity too small");
  const bool had_pre = dst->n_pre > 0;
  const bool plain = src->n_pre == 0 && !(dst->e->opt_share_prefix && !dst->e->opt_lm_cluster && src->B == 1 && T >= 16);
  if (plain) {
    // full copies: ONE kernel over the T written positions of every row (not the whole capacity: a batch-64 clone of 126
    // positions in a 284-position cache moved 1.8 GB instead of 0.8 GB), whatever the two capacities are
    for (int b = 0; b < dst->B; ++b) prefix_release(dst, b);
    if (T > 0) {
      const long total = (long)c.num_layers * 2 * dst->B * c.num_heads * T * 16;
      kv_copy_t_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst->kv, src->kv, c.num_layers * 2, dst->B, src->B, c.num_heads, T, src->cap, dst->cap);
    }
  } else if (src->B == 1 && !src->pre_owner[0] && !src->zombie && dst->e->opt_share_prefix && !dst->e->opt_lm_cluster && T >= 16) {
    // the common case, a plain voice state cloned into every row: all rows borrow its first T & ~15 positions and ONE
    // kernel copies the (< 16) positions behind them
    ptts_lm_state *owner = const_cast<ptts_lm_state *>(src);
    const int plen = T & ~15;
    for (int b = 0; b < dst->B; ++b) prefix_borrow(dst, b, owner, plen);
    if (T > plen) {
      const long total = (long)c.num_layers * 2 * dst->B * c.num_heads * (T - plen) * 16;
      kv_copy_t_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst->kv, src->kv, c.num_layers * 2, dst->B, 1, c.num_heads, T - plen, src->cap, dst->cap, plen);
    }
  } else {
    // row by row: one copy kernel per destination row for the positions it does not share
    for (int b = 0; b < dst->B; ++b) {
      const int sb = src->B == 1 ? 0 : b;
      CHK(kv_clone_row(dst, b, src, sb, src->h_off[sb], st));
    }
  }
  if (had_pre || dst->n_pre > 0) CHK(upload_prefixes(dst, st));
  for (int b = 0; b < dst->B; ++b) dst->h_off[b] = src->h_off[src->B == 1 ? 0 : b];
  if (*std::min_element(dst->h_off.begin(), dst->h_off.end()) == T) {
    set_int_kernel<<<cdiv(dst->B, 256), 256, 0, st>>>(dst->offset, dst->B, T);
  } else {
    HIPCHK(hipMemcpyAsync(dst->offset, dst->h_off.data(), dst->B * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // h_off.data() is pageable host memory
  }
  // a cloned state starts a new generation: next input is BOS (NaN), reference tts_model.py:748-753
  fill_kernel<<<cdiv(dst->B * c.ldim, 256), 256, 0, st>>>(dst->lat_prev, (long)dst->B * c.ldim, NAN);
  HIPCHK(hipGetLastError());
  return 0;
}

// Row `row` of dst <- the single sequence of src (B = 1): KV rows, offset and a BOS pending input.  Lets a batch
// be assembled from utterances prefilled one by one with different prompt lengths (per-row offsets).
extern "C" int ptts_lm_state_copy_row(ptts_lm_state *dst, int32_t row, const ptts_lm_state *src, void *stream) {
  if (src && src->B != 1) return fail(-1, "copy_row: src must have batch 1 (use ptts_lm_state_copy_row_from)");
  return ptts_lm_state_copy_row_from(dst, row, src, 0, stream);
}
extern "C" int ptts_lm_state_copy_row_from(ptts_lm_state *dst, int32_t row, const ptts_lm_state *src, int32_t src_row, void *stream) {
  if (!dst || !src) return fail(-1, "null state");
  if (dst->e != src->e) return fail(-1, "states belong to different engines");
  if (src_row < 0 || src_row >= src->B || row < 0 || row >= dst->B) return fail(-1, "copy_row: row out of range");
  if (dst == src) return fail(-1, "copy_row: source and destination are the same state");
  const ptts_config &c = dst->e->cfg;
  hipStream_t st = S(dst->e, stream);
  HIPCHK(hipSetDevice(dst->e->device));
  ENGINE_LOCK(dst->e);
  CHK(refuse_if_lent(dst, "copy_row"));
  const int T = src->h_off[src_row];
  if (T > dst->cap) return fail(-5, "copy_row: destination capacity too small");
  const bool had_pre = dst->pre_owner[row] != nullptr;
  CHK(kv_clone_row(dst, row, src, src_row, T, st));
  if (had_pre || dst->pre_owner[row]) {
    const KvPrefix &p = dst->h_pre[row];
    set_prefix_kernel<<<1, 64, 0, st>>>(dst->d_pre + row, 1, p.kv, p.cap, p.len);
  }
  dst->h_off[row] = T;
  dst->h_active[row] = 1;
  set_int_kernel<<<1, 64, 0, st>>>(dst->active + row, 1, 1);
  set_int_kernel<<<1, 64, 0, st>>>(dst->offset + row, 1, T);
  fill_kernel<<<cdiv(c.ldim, 256), 256, 0, st>>>(dst->lat_prev + (size_t)row * c.ldim, (long)c.ldim, NAN);
  HIPCHK(hipGetLastError());
  return 0;
}

// Continuous batching: park / unpark one row.  A parked row still flows through the batched kernels (its
// outputs are ignored by the caller) but its position is pinned to 0, so it never exhausts the KV capacity
// and its attention reads one key.
extern "C" int ptts_lm_state_set_row_active(ptts_lm_state *s, int32_t row, int32_t active, void *stream) {
  if (!s || row < 0 || row >= s->B) return fail(-1, "set_row_active: row out of range");
  hipStream_t st = S(s->e, stream);
  ENGINE_LOCK(s->e);
  s->h_active[row] = active ? 1 : 0;
  set_int_kernel<<<1, 64, 0, st>>>(s->active + row, 1, active ? 1 : 0);
  if (!active) {
    s->h_off[row] = 0;
    set_int_kernel<<<1, 64, 0, st>>>(s->offset + row, 1, 0);
    if (s->pre_owner[row]) {  // a parked row reads key 0 of its own cache only
      prefix_release(s, row);
      set_prefix_kernel<<<1, 64, 0, st>>>(s->d_pre + row, 1, nullptr, 0, 0);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int ptts_lm_state_offsets(ptts_lm_state *s, int32_t *h, void *stream) {
  hipStream_t st = S(s->e, stream);
  HIPCHK(hipMemcpyAsync(h, s->offset, s->B * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" const float *ptts_lm_latent_ptr(ptts_lm_state *s) { return s->lat_prev; }

extern "C" int ptts_lm_set_noise(ptts_lm_state *s, float temp, uint64_t seed) {
  if (temp < 0) return fail(-1, "temperature must be >= 0");
  s->rng_std = std::sqrt(temp);  // std = temp ** 0.5 (reference flow_lm.py:132)
  s->rng_seed = seed;
  return 0;
}

extern "C" int ptts_profile_start(ptts_engine *e) {
  if (!e) return fail(-1, "null engine");
  ENGINE_LOCK(e);
  for (auto &r : e->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  e->prof.recs.clear();
  e->prof.on = true;
  return 0;
}

// Stops profiling and writes one line per (site, kernel): "site kernel count total_ms bytes flops\n"
extern "C" int64_t ptts_profile_stop(ptts_engine *e, char *h_out, int64_t capacity) {
  if (!e) return fail(-1, "null engine");
  ENGINE_LOCK(e);
  e->prof.on = false;
  if (hipDeviceSynchronize() != hipSuccess) return fail(-2, "sync failed");
  std::map<std::pair<std::string, std::string>, std::array<double, 4>> agg;
  std::vector<std::pair<std::string, std::string>> order;
  for (auto &r : e->prof.recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    auto key = std::make_pair(r.site, r.kernel);
    if (!agg.count(key)) { agg[key] = {0, 0, 0, 0}; order.push_back(key); }
    auto &v = agg[key];
    v[0] += 1; v[1] += ms; v[2] += r.bytes; v[3] += r.flops;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  e->prof.recs.clear();
  std::string out;
  char line[512];
  for (auto &k : order) {
    auto &v = agg[k];
    snprintf(line, sizeof line, "%s %s %.0f %.6f %.0f %.0f\n", k.first.empty() ? "-" : k.first.c_str(), k.second.c_str(), v[0], v[1], v[2], v[3]);
    out += line;
  }
  if ((int64_t)out.size() + 1 > capacity) return fail(-1, "profile buffer too small");
  memcpy(h_out, out.c_str(), out.size() + 1);
  return (int64_t)out.size();
}


// ------------------------------------------------------------------------------------------------
// Single-launch flow MLP (ptts_flow.h).  Geometry per state: RT row tiles per cluster, NG clusters of FDF workgroups.
static bool flow_cluster_ok(const ptts_engine *e, const ptts_lm_state *s) {
  const ptts_config &c = e->cfg;
  const int FDF = c.flow_dim / 16, LF = c.ldim / 16;
  if (!e->opt_flow_cluster || c.flow_depth > FLOW_MAX_DEPTH || FDF > 64 || LF > FDF) return false;
  const int kpw = cdiv(std::max(FDF, LF), FLOW_WORKERS);
  if (kpw != 1 && kpw != 2 && kpw != 4) return false;  // flow_dim <= 512
  if (!e->input_proj.bias || !e->fin.bias) return false;
  for (auto &r : e->res) if (!r.l0.bias || !r.l2.bias) return false;
  // every workgroup of the launch must be resident at once (they wait for each other) and a 9-wave workgroup fills a
  // CU: the grid is capped at opt_flow_max_cus workgroups (default 128: half the chip, so a codec kernel always has CUs left) (larger batches loop over row groups inside the kernel)
  return FDF <= std::min(e->opt_flow_max_cus, e->n_cus);
}
// (re)allocates the per-state buffers whose size depends on the number of LSD steps; never called during capture
static int ensure_flow(ptts_engine *e, ptts_lm_state *s, int steps, hipStream_t st) {
  if (steps <= s->flow_steps) return 0;
  if (s->n_graphs > 0)
    return fail(-1, "lsd_decode_steps grows beyond what this state's captured graphs were built for: destroy the graphs first "
                    "(they hold the old AdaLN / exchange buffers)");
  const ptts_config &c = e->cfg;
  const int FDF = c.flow_dim / 16;
  // row tiles per cluster.  1 (default): one cluster per 16 rows, the clusters' weight re-reads stay in L2 / Infinity Cache.
  // 2 (PTTS_FLOW_RT=2, A/B knob): half the clusters, half the weight fetches and CUs, twice the exchange payload per phase.
  static const int rt_env = [] { const char *v = getenv("PTTS_FLOW_RT"); return v ? atoi(v) : 1; }();
  s->flow_rt = (rt_env == 2 && s->MT >= 2) ? 2 : 1;
  s->flow_ng = cdiv(s->MT, s->flow_rt);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipStreamSynchronize(e->stream));
  AllocScope alloc_scope(st);
  hipFree(s->mod); hipFree(s->fexch); hipFree(s->fflags);
  s->mod = nullptr; s->fexch = nullptr; s->fflags = nullptr;
  const size_t rt = (size_t)s->MT * 256;
  const int nph = steps * (2 * c.flow_depth + 2);
  CHK(dallocT(nullptr, &s->mod, (size_t)steps * rt * e->adaln.NT));
  CHK(dallocT(nullptr, &s->fexch, (size_t)s->flow_ng * nph * s->flow_rt * FDF * 256));
  CHK(dallocT(nullptr, &s->fflags, (size_t)s->flow_ng * nph * FDF));
  HIPCHK(hipStreamSynchronize(st));
  s->flow_steps = steps;
  return 0;
}

template <int RT>
static void launch_flow_rt(hipStream_t st, const FlowArgs &fa, int kpw) {
  const dim3 grid(fa.NCL * fa.FDF), block(FLOW_THREADS);
  switch (kpw) {
    case 1: flow_cluster_kernel<RT, 1><<<grid, block, 0, st>>>(fa); break;
    case 2: flow_cluster_kernel<RT, 2><<<grid, block, 0, st>>>(fa); break;
    default: flow_cluster_kernel<RT, 4><<<grid, block, 0, st>>>(fa); break;
  }
}

static void launch_flow_cluster(hipStream_t st, ptts_engine *e, ptts_lm_state *s, int lsd_steps, float *d_latent_out) {
  const ptts_config &c = e->cfg;
  const int FDF = c.flow_dim / 16, LF = c.ldim / 16;
  FlowArgs fa;
  memset(&fa, 0, sizeof fa);
  fa.MT = s->MT; fa.M = s->B; fa.NG = s->flow_ng; fa.NCL = std::max(1, std::min(s->flow_ng, std::min(e->opt_flow_max_cus, e->n_cus) / FDF)); fa.FDF = FDF; fa.LF = LF; fa.AF = e->adaln.NT;
  fa.depth = c.flow_depth; fa.steps = lsd_steps; fa.ldim = c.ldim;
  fa.w_in = e->input_proj.w; fa.b_in = e->input_proj.bias;
  for (int r = 0; r < c.flow_depth; ++r) {
    fa.w_l0[r] = e->res[r].l0.w; fa.b_l0[r] = e->res[r].l0.bias;
    fa.w_l2[r] = e->res[r].l2.w; fa.b_l2[r] = e->res[r].l2.bias;
    fa.ln_w[r] = e->res[r].ln_w; fa.ln_b[r] = e->res[r].ln_b;
  }
  fa.w_fin = e->fin.w; fa.b_fin = e->fin.bias;
  fa.mod = s->mod; fa.mod_step = (long)s->MT * 256 * e->adaln.NT;
  fa.latfm = s->latfm;
  fa.lat = s->lat; fa.lat_out1 = s->lat_prev; fa.lat_out2 = d_latent_out;
  fa.inv_steps = 1.0f / (float)lsd_steps;
  fa.exch = s->fexch; fa.flags = s->fflags; fa.ctr = s->rng_ctr; fa.err = s->ferr;
  // weights of the chain once per cluster-set (L2 / Infinity Cache absorb the clusters' re-reads) + modulations + io
  const double wbytes = 4.0 * 256 * ((double)FDF * LF + 2.0 * c.flow_depth * FDF * FDF + (double)LF * FDF);
  const double flops = 2.0 * s->B * 256.0 * ((double)FDF * LF + 2.0 * c.flow_depth * FDF * FDF + (double)LF * FDF) * lsd_steps;
  ProfScope ps(st, "flow_cluster@" + std::to_string((long)fa.NCL * FDF * FLOW_THREADS),
               lsd_steps * (wbytes + 4.0 * s->B * 16.0 * e->adaln.NT), flops);
  const int kpw = cdiv(std::max(FDF, LF), FLOW_WORKERS);
  s->coop_wgs = std::max(s->coop_wgs, fa.NCL * FDF);
  if (s->flow_rt == 2) launch_flow_rt<2>(st, fa, kpw);
  else launch_flow_rt<1>(st, fa, kpw);
}


// ------------------------------------------------------------------------------------------------
// Single-launch transformer stack (ptts_lm.h)
static constexpr int kLmMaxWGs = 256;  // resident workgroups of one launch (one per CU)
static bool lm_cluster_ok(const ptts_engine *e, const ptts_lm_state *s) {
  if (!e->opt_lm_cluster || !e->lm_table) return false;
  if (e->cfg.d_model / 16 > std::min(kLmMaxWGs, e->n_cus)) return false;  // one cluster must be resident at once
  if (s->n_pre > 0) return false;  // the single-launch stack reads its keys from the state's own cache only
  return (double)s->kv_plane() * 4.0 < 2.0e9;  // 32-bit buffer offsets into a K / V plane
}
static int ensure_lm_cluster(ptts_engine *e, ptts_lm_state *s, hipStream_t st) {
  if (s->lexch || !e->lm_table) return 0;
  const ptts_config &c = e->cfg;
  const int DF = c.d_model / 16, FFF = c.ff_dim / 16;
  AllocScope alloc_scope(st);
  CHK(dallocT(nullptr, &s->lexch, (size_t)s->MT * c.num_layers * (4 * DF + FFF) * 256));
  CHK(dallocT(nullptr, &s->lflags, (size_t)s->MT * c.num_layers * 5 * DF));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
static void launch_lm_cluster(hipStream_t st, ptts_engine *e, ptts_lm_state *s, Scratch &sc) {
  const ptts_config &c = e->cfg;
  LmArgs la;
  memset(&la, 0, sizeof la);
  la.MT = s->MT; la.M = s->B; la.NG = s->MT; la.DF = c.d_model / 16; la.FFF = c.ff_dim / 16; la.H = c.num_heads;
  la.NCL = std::max(1, std::min(la.NG, std::min(kLmMaxWGs, e->n_cus) / la.DF));
  s->coop_wgs = std::max(s->coop_wgs, la.NCL * la.DF);
  la.L = c.num_layers; la.cap = s->cap;
  la.layers = e->lm_table; la.kv = s->kv; la.kv_plane = (long)s->kv_plane();
  la.offset = s->offset; la.freq = e->freq_lm; la.x = sc.x; la.exch = s->lexch; la.flags = s->lflags;
  la.ctr = s->rng_ctr; la.err = s->ferr; la.ln_eps = 1e-5f;
  double keys = 0;
  for (int b = 0; b < s->B; ++b) keys += s->h_off[b] + 1;
  const double wbytes = 4.0 * 256.0 * c.num_layers * ((double)4 * la.DF * la.DF + 2.0 * la.DF * la.FFF);
  ProfScope ps(st, "lm_cluster@" + std::to_string((long)la.NCL * la.DF * FLOW_THREADS),
               wbytes + c.num_layers * keys * c.num_heads * 64 * 4 * 2,
               c.num_layers * (2.0 * s->B * 256.0 * ((double)4 * la.DF * la.DF + 2.0 * la.DF * la.FFF) + 4.0 * keys * c.num_heads * 64));
  lm_cluster_kernel<0><<<dim3(la.NCL * la.DF), dim3(FLOW_THREADS), 0, st>>>(la);
}

// Cooperative kernels (flow / transformer clusters) need all their workgroups resident: a launch of W workgroups (one per
// CU, see flow_cluster_ok) may share the GPU with other cooperative launches only while the SUM of their grids fits the
// device - beyond that two half-resident grids could wait for each other.  Steps that contain such a launch therefore take
// one of N = floor(CUs / W) per-device SLOTS: step k (in launch order, all states and streams of the device) waits on the GPU
// for step k - N and records its slot's event when it ends.  N = 2 for the flow cluster's default 128 workgroups, so the
// FlowLM steps of two states queued on two streams overlap (round 2 chained them all, lm || lm = 2.16x one alone); the
// 256-workgroup transformer cluster gets N = 1, i.e. the old strict chain.  GPU-side order only, the host never waits.
static constexpr int kCoopMaxSlots = 4;
struct CoopSlots { hipEvent_t ev[kCoopMaxSlots] = {}; bool used[kCoopMaxSlots] = {}; unsigned long long next = 0; };
static std::mutex g_coop_mu;
static CoopSlots g_coop[64];
struct CoopGuard {
  CoopSlots *cs = nullptr;
  hipStream_t st;
  int slot = 0, rc = 0;
  // `wgs` = workgroups of the step's cooperative launch (0: none, the guard does nothing); `n_cus` = CUs of the device
  CoopGuard(int device, hipStream_t stream, int wgs, int n_cus) : st(stream) {
    if (wgs <= 0) return;
    g_coop_mu.lock();
    cs = &g_coop[device & 63];
    // the events belong to the device of the engine, whatever device the calling thread had current (a scheduler thread
    // of an engine on cuda:N > 0): the callers have run hipSetDevice(e->device)
    const int n = std::max(1, std::min(kCoopMaxSlots, n_cus / wgs));
    slot = (int)(cs->next++ % (unsigned)n);
    if (!cs->ev[slot] && hipEventCreateWithFlags(&cs->ev[slot], hipEventDisableTiming) != hipSuccess) { rc = -2; cs->ev[slot] = nullptr; return; }
    // a launch that needs more than its share (n shrank: e.g. the transformer cluster after flow-only steps) waits for ALL slots
    for (int i = 0; i < kCoopMaxSlots; ++i)
      if (cs->ev[i] && cs->used[i] && (i == slot || i >= n))
        if (hipStreamWaitEvent(st, cs->ev[i], 0) != hipSuccess) rc = -2;
  }
  int finish() {  // call after the step is enqueued; returns < 0 when the chain could not be established
    if (!cs) return 0;
    if (cs->ev[slot]) {
      if (hipEventRecord(cs->ev[slot], st) != hipSuccess) rc = -2;
      else cs->used[slot] = true;
    }
    CoopSlots *c = cs;
    cs = nullptr;
    (void)c;
    g_coop_mu.unlock();
    return rc;
  }
  ~CoopGuard() { if (cs) { cs = nullptr; g_coop_mu.unlock(); } }
};
// CUs a stream may use (hipExtStreamCreateWithCUMask streams: the bits of its mask; else the whole device)
static int stream_cu_count(hipStream_t st, int n_cus) {
  uint32_t mask[16] = {};
  if (!st || hipExtStreamGetCUMask(st, 16, mask) != hipSuccess) { (void)hipGetLastError(); return n_cus; }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  return bits > 0 ? std::min(bits, n_cus) : n_cus;
}

static void lm_layers(hipStream_t st, ptts_engine *e, ptts_lm_state *s, Scratch &sc, int M, int Tq, bool rope_done = false) {
  const ptts_config &c = e->cfg;
  bind_engine(e);
  if (!rope_done) {  // decode steps build the table in their prologue kernel
    SITE("lm.rope");
    ProfScope ps(st, "rope_table", 256.0 * M, 0);
    rope_table_kernel<<<cdiv(M * 32, 256), 256, 0, st>>>(s->offset, e->freq_lm, sc.rope, M, Tq);
  }
  for (int l = 0; l < c.num_layers; ++l) {
    TrCtx t;
    t.D = c.d_model; t.H = c.num_heads; t.FF = c.ff_dim; t.MT = sc.MT; t.M = M; t.Tq = Tq; t.QB = sc.QB;
    t.cap = s->cap; t.ring = 0; t.ctx = 0;
    // decode steps split the keys over the waves of a workgroup (decode_attn_waves): no key splits, no combine launch
    t.splits = Tq == 1 ? 1 : std::min(sc.splits_cap, attn_splits(s->B * c.num_heads * sc.QB, cdiv(s->cap, 16)));
    t.x_in = sc.x; t.x = sc.x; t.x_out = sc.x; t.out_ds = 0; t.par = nullptr;
    t.h = sc.h; t.ao = sc.ao; t.ff = sc.ff; t.q = sc.q; t.part = sc.part;
    t.Kc = s->K(l); t.Vc = s->V(l); t.offset = s->offset; t.rope = sc.rope;
    t.pre = s->d_pre; t.layer = l; t.cascade = e->opt_share_prefix && (s->casc_mode < 0 ? s->n_pre > 0 : s->casc_mode > 0) ? e->opt_cascade : 0;
    t.kv_keys = 0;
    for (int b = 0; b < s->B; ++b) t.kv_keys += s->h_off[b] + Tq;
    t.tag = "lm";
    run_tr_layer(st, e->lm[l], t);
  }
}

extern "C" int ptts_lm_prefill(ptts_engine *e, ptts_lm_state *s, const float *d_emb, int32_t T, void *stream) {
  ENGINE_LOCK(e);
  if (T < 1) return 0;
  HIPCHK(hipSetDevice(e->device));
  const ptts_config &c = e->cfg;
  for (int b = 0; b < s->B; ++b)
    if (s->h_off[b] + T > s->cap) return fail(-5, "prefill: KV cache capacity exceeded");
  hipStream_t st = S(e, stream);
  const int M = s->B * T;
  if (!s->pre.x || s->pre.MT < cdiv(M, 16) || s->pre.QB != cdiv(T, 16)) {
    HIPCHK(hipStreamSynchronize(st));
    if (s->pre.x) free_scratch(&s->pre);
    AllocScope alloc_scope(st);
    CHK(alloc_scratch(e, &s->pre, s->B, T, c.d_model, c.num_heads, c.ff_dim, s->cap));
  }
  Scratch &sc = s->pre;
  const int MT = cdiv(M, 16);
  sc.MT = MT;
  long n4 = (long)MT * (c.d_model / 16) * 64;
  to_fm_kernel<<<cdiv(n4, 256), 256, 0, st>>>(d_emb, sc.x, M, c.d_model, MT);
  lm_layers(st, e, s, sc, M, T);
  add_int_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(s->offset, s->B, T);
  for (auto &o : s->h_off) o += T;
  HIPCHK(hipGetLastError());
  return 0;
}

// workgroups of the largest cooperative launch a decode step of this state contains (0: per-layer launches only)
static int coop_wgs_of_step(const ptts_engine *e, const ptts_lm_state *s, int lsd_steps) {
  int w = 0;
  const ptts_config &c = e->cfg;
  if (flow_cluster_ok(e, s) && lsd_steps <= s->flow_steps) {
    const int FDF = c.flow_dim / 16;
    w = std::max(1, std::min(s->flow_ng, std::min(e->opt_flow_max_cus, e->n_cus) / FDF)) * FDF;
  }
  if (lm_cluster_ok(e, s) && s->lexch) {
    const int DF = c.d_model / 16;
    w = std::max(w, std::max(1, std::min(s->MT, std::min(kLmMaxWGs, e->n_cus) / DF)) * DF);
  }
  return w;
}

static int lm_step_enqueue(hipStream_t st, ptts_engine *e, ptts_lm_state *s, const float *d_latent_in,
                           const float *d_noise, int lsd_steps, float eos_thr, float *d_latent_out,
                           float *d_eos_logit, uint8_t *d_is_eos) {
  const ptts_config &c = e->cfg;
  bind_engine(e);
  const int B = s->B, MT = s->MT, D = c.d_model, FD = c.flow_dim, DF = D / 16, FDF = FD / 16, LF = c.ldim / 16;
  Scratch &sc = s->dec;
  s->coop_wgs = 0;
  const float *tcomb = e->tcomb[lsd_steps];
  SITE("lm.prep");  // BOS substitution + input_linear + noise / LSD start point + RoPE table: one launch
  if (e->in_linear.w && !e->in_linear.bias && e->in_linear.KF == LF) {
    ProfScope ps(st, "prep_in", 16.0 * B * c.ldim + 4.0 * (D * c.ldim + B * D), 2.0 * B * D * c.ldim);
    const int nb_gemm = cdiv(DF * MT, 4), nb_prep = cdiv(MT * LF * 64, 256);
    prep_in_kernel<<<nb_gemm + nb_prep + cdiv(B * 32, 256), 256, 0, st>>>(
        d_latent_in ? d_latent_in : s->lat_prev, e->bos, d_noise, e->in_linear.w, sc.x, s->lat, s->latfm, B, c.ldim, MT, DF, s->rng_std,
        s->rng_seed, s->rng_ctr, nb_gemm, nb_prep, RopeArgs{s->offset, e->freq_lm, sc.rope, B, 1});
  } else {
    {
      ProfScope ps(st, "prep_lm", 16.0 * B * c.ldim, 0);
      const int nb_prep = cdiv(MT * LF * 64, 256);
      prep_lm_kernel<<<nb_prep + cdiv(B * 32, 256), 256, 0, st>>>(d_latent_in ? d_latent_in : s->lat_prev, e->bos, d_noise,
                                                                   s->xlat, s->lat, s->latfm, B, c.ldim, MT, s->rng_std,
                                                                   s->rng_seed, s->rng_ctr, nb_prep,
                                                                   RopeArgs{s->offset, e->freq_lm, sc.rope, B, 1});
    }
    SITE("lm.in_linear");
    GemmArgs a0 = mk_gemm(e->in_linear, s->xlat, LF, MT, B);
    a0.Y = sc.x; a0.YF = DF;
    launch_gemm(st, a0, PRE_NONE);
  }
  GemmArgs a;
  if (lm_cluster_ok(e, s) && s->lexch) {
    SITE("lm.cluster");
    launch_lm_cluster(st, e, s, sc);
  } else {
    lm_layers(st, e, s, sc, B, 1, true);
  }
  SITE("flow.head");  // out_norm is folded into [cond_embed ; out_eos]
  a = mk_gemm(e->head, sc.x, DF, MT, B);
  a.epi = EPI_HEAD; a.Y = s->ce; a.YF = FDF; a.head_nt = FDF; a.eos_thr = eos_thr;
  a.eos_logit = s->eos_logit; a.is_eos = s->is_eos;
  a.eos_logit2 = d_eos_logit; a.is_eos2 = d_is_eos;  // caller's buffers are written by the epilogue itself
  a.tail_offset = s->offset; a.tail_ctr = s->rng_ctr; a.tail_active = s->active;  // the step's bookkeeping rides along (was a launch)
  const bool silu_in_head = lsd_steps == 1;  // s->ce then holds silu(t_emb + cond) and the modulation GEMM loads it as is
  if (silu_in_head) { a.act = ACT_SILU; a.prevec = tcomb; }
  launch_gemm(st, a, PRE_LNFOLD);
  const int AF = e->adaln.NT;
  if (flow_cluster_ok(e, s) && lsd_steps <= s->flow_steps) {
    // the AdaLN modulations of every LSD step (they depend on the step's (s, t) only through t_emb), then the whole
    // chain + Euler loop in ONE launch
    for (int i = 0; i < lsd_steps; ++i) {
      SITE("flow.adaln");
      a = mk_gemm(e->adaln, s->ce, FDF, MT, B);
      a.prevec = tcomb + (size_t)i * FD; a.Y = s->mod + (size_t)i * MT * 256 * AF; a.YF = AF;
      launch_gemm(st, a, silu_in_head ? PRE_NONE : PRE_ADDSILU);
    }
    SITE("flow.cluster");
    launch_flow_cluster(st, e, s, lsd_steps, d_latent_out);
  } else
  for (int i = 0; i < lsd_steps; ++i) {
    // all AdaLN modulations of the step in one GEMM on silu(t_emb + cond)  (mlp.py:107,127,210)
    SITE("flow.adaln");
    a = mk_gemm(e->adaln, s->ce, FDF, MT, B);
    a.prevec = tcomb + (size_t)i * FD; a.Y = s->mod; a.YF = AF;
    launch_gemm(st, a, silu_in_head ? PRE_NONE : PRE_ADDSILU);
    SITE("flow.input_proj");
    a = mk_gemm(e->input_proj, s->latfm, LF, MT, B);
    a.Y = s->fx; a.YF = FDF;
    a.stat_out = s->fstat;  // row statistics of x for the first block's LayerNorm
    launch_gemm(st, a, PRE_NONE);
    for (int r = 0; r < c.flow_depth; ++r) {
      const float *shift = s->mod + (size_t)(r * 3 * FDF) * 256;
      const float *scale = shift + (size_t)FDF * 256;
      const float *gate = scale + (size_t)FDF * 256;
      SITE("flow.res.l0");  // in_ln + AdaLN modulate are applied to the operand on load (PRE_LNMOD)
      a = mk_gemm(e->res[r].l0, s->fx, FDF, MT, B);
      a.act = ACT_SILU; a.Y = s->f1; a.YF = FDF;
      a.lnm_w = e->res[r].ln_w; a.lnm_b = e->res[r].ln_b; a.mod_shift = shift; a.mod_scale = scale; a.modF = AF;
      a.ln_eps = 1e-6f;  // flow-MLP LayerNorm eps (mlp.py:95)
      a.stat_in = s->fstat; a.stat_nt = FDF;
      launch_gemm(st, a, PRE_LNMOD);
      SITE("flow.res.l2");
      a = mk_gemm(e->res[r].l2, s->f1, FDF, MT, B);
      a.epi = EPI_GATE; a.R = s->fx; a.RF = FDF; a.G = gate; a.GF = AF; a.Y = s->fx; a.YF = FDF;
      a.stat_out = s->fstat;  // ... of the updated x for the next block / the final layer
      launch_gemm(st, a, PRE_NONE);
    }
    const float *shift = s->mod + (size_t)(c.flow_depth * 3 * FDF) * 256;
    const float *scale = shift + (size_t)FDF * 256;
    SITE("flow.final");  // norm_final (no affine) + modulate on load
    a = mk_gemm(e->fin, s->fx, FDF, MT, B);
    a.lnm_w = nullptr; a.lnm_b = nullptr; a.mod_shift = shift; a.mod_scale = scale; a.modF = AF; a.ln_eps = 1e-6f;
    a.stat_in = s->fstat; a.stat_nt = FDF;
    a.epi = EPI_LATENT; a.lat = s->lat; a.ldim = c.ldim; a.inv_steps = 1.0f / (float)lsd_steps; a.Y = s->latfm; a.YF = LF;
    if (i == lsd_steps - 1) { a.lat_out1 = s->lat_prev; a.lat_out2 = d_latent_out; }  // next step's input + caller's copy
    launch_gemm(st, a, PRE_LNMOD);
  }
  SITE("");
  return 0;
}

extern "C" int ptts_lm_decode_step(ptts_engine *e, ptts_lm_state *s, const float *d_latent_in, const float *d_noise,
                                   int32_t lsd_steps, float eos_threshold, float *d_latent_out, float *d_eos_logit,
                                   uint8_t *d_is_eos, void *stream) {
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  CHK(prepare_lsd(e, lsd_steps));
  CHK(ensure_flow(e, s, lsd_steps, S(e, stream)));
  CHK(ensure_lm_cluster(e, s, S(e, stream)));
  for (int b = 0; b < s->B; ++b)
    if (s->h_off[b] + 1 > s->cap) return fail(-5, "decode: KV cache capacity exceeded");
  {
    // grid of the cooperative launch this step will contain (the enqueue records the exact value in s->coop_wgs)
    const int wgs = coop_wgs_of_step(e, s, lsd_steps);
    if (wgs > stream_cu_count(S(e, stream), e->n_cus))
      return fail(-1, "decode: the stream's CU mask is smaller than the cooperative flow launch (lower the flow_max_cus option)");
    CoopGuard guard(e->device, S(e, stream), wgs, e->n_cus);
    CHK(lm_step_enqueue(S(e, stream), e, s, d_latent_in, d_noise, lsd_steps, eos_threshold, d_latent_out, d_eos_logit, d_is_eos));
    if (guard.finish() < 0) return fail(-2, "decode: could not chain the cooperative launch behind its predecessor");
  }
  for (int b = 0; b < s->B; ++b) s->h_off[b] += s->h_active[b];
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Mimi
static int build_mimi_state(ptts_engine *e, ptts_mimi_state *s);

extern "C" int ptts_mimi_state_create(ptts_engine *e, int32_t B, ptts_mimi_state **out) {
  if (!e || !out || B < 1) return fail(-1, "bad argument");
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  ptts_mimi_state *s = new ptts_mimi_state();
  s->e = e;
  s->B = B;
  const int rc = build_mimi_state(e, s);
  if (rc < 0) {
    const std::string msg = g_err;
    ptts_mimi_state_destroy(s);
    return fail(rc, msg);
  }
  *out = s;
  return 0;
}

static int build_mimi_state(ptts_engine *e, ptts_mimi_state *s) {
  const ptts_config &c = e->cfg;
  const int B = s->B;
  AllocScope alloc_scope(e->stream);
  s->MTb = cdiv(B, 16);
  s->MT16 = B;  // one 16-row tile per sequence
  const int C = c.m_dim, CF = C / 16;
  CHK(dallocT(nullptr, &s->frame, 1));
  CHK(dallocT(nullptr, &s->offset, B));
  CHK(dallocT(nullptr, &s->kv, (size_t)c.m_layers * 2 * s->kv_plane()));
  CHK(dallocT(nullptr, &s->zl, (size_t)s->MTb * 256 * (c.ldim / 16)));
  s->zq_stride = (long)s->MTb * 256 * CF;
  CHK(dallocT(nullptr, &s->zq, (size_t)2 * s->zq_stride));
  const size_t r16 = (size_t)s->MT16 * 256;
  CHK(dallocT(nullptr, &s->u0, r16 * CF));
  CHK(dallocT(nullptr, &s->u, r16 * CF));
  CHK(dallocT(nullptr, &s->h, r16 * CF));
  CHK(dallocT(nullptr, &s->ao, r16 * CF));
  CHK(dallocT(nullptr, &s->ff, r16 * (c.m_ff / 16)));
  CHK(dallocT(nullptr, &s->q, (size_t)B * c.m_heads * 4 * 256));
  CHK(dallocT(nullptr, &s->rope, (size_t)B * 16 * 64));
  s->splits = attn_splits(B * c.m_heads, e->ring / 16);
  CHK(dallocT(nullptr, &s->part, (size_t)B * c.m_heads * s->splits * 16 * ATT_PSTRIDE));
  s->tr_stride = (long)r16 * CF;
  CHK(dallocT(nullptr, &s->tr_out, (size_t)2 * s->tr_stride));
  int mult = 8, rows = 16;
  s->rows[0] = rows;
  s->a0_stride = (long)r16 * (mult * c.n_filters / 16);
  CHK(dallocT(nullptr, &s->a0, (size_t)2 * s->a0_stride));
  for (int i = 0; i < 3; ++i) {
    const int cout = mult * c.n_filters / 2, hid = cout / c.compress;
    rows *= c.ratios[i];
    s->rows[i + 1] = rows;
    const size_t rt = (size_t)B * (rows / 16) * 256;
    s->c_stride[i] = (long)rt * (cout / 16);
    s->s_stride[i] = s->c_stride[i];
    CHK(dallocT(nullptr, &s->cbuf[i], (size_t)2 * s->c_stride[i]));
    CHK(dallocT(nullptr, &s->craw[i], (size_t)s->c_stride[i]));
    CHK(dallocT(nullptr, &s->sbuf[i], (size_t)2 * s->s_stride[i]));
    CHK(dallocT(nullptr, &s->rbuf[i], rt * (hid / 16)));
    mult /= 2;
  }
  CHK(dallocT(nullptr, &s->pcm_dbg, (size_t)B * rows));
  CHK(dallocT(nullptr, &s->pcm_part, (size_t)B * rows));
  s->pcm_cstride = (long)cdiv(B * rows, 64) * 2;
  CHK(dallocT(nullptr, &s->pcm_carry, (size_t)2 * s->pcm_cstride));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" void ptts_mimi_state_destroy(ptts_mimi_state *s) {
  if (!s) return;
  hipSetDevice(s->e->device);
  hipDeviceSynchronize();
  hipFree(s->frame); hipFree(s->offset); hipFree(s->kv); hipFree(s->zl); hipFree(s->zq); hipFree(s->u0);
  hipFree(s->u); hipFree(s->h); hipFree(s->ao); hipFree(s->ff); hipFree(s->q); hipFree(s->part);
  hipFree(s->tr_out); hipFree(s->a0); hipFree(s->pcm_dbg); hipFree(s->pcm_part); hipFree(s->pcm_carry); hipFree(s->rope);
  for (int i = 0; i < 3; ++i) { hipFree(s->cbuf[i]); hipFree(s->craw[i]); hipFree(s->rbuf[i]); hipFree(s->sbuf[i]); }
  delete s;
}

extern "C" int ptts_mimi_state_reset(ptts_mimi_state *s, void *stream) {
  hipStream_t st = S(s->e, stream);
  // zero-initialised carries == `previous` / `partial` zeros of init_state (conv.py:84-91,145-149)
  HIPCHK(hipMemsetAsync(s->frame, 0, sizeof(int), st));
  HIPCHK(hipMemsetAsync(s->offset, 0, s->B * sizeof(int), st));
  HIPCHK(hipMemsetAsync(s->zq, 0, (size_t)2 * s->zq_stride * 4, st));
  HIPCHK(hipMemsetAsync(s->tr_out, 0, (size_t)2 * s->tr_stride * 4, st));
  HIPCHK(hipMemsetAsync(s->a0, 0, (size_t)2 * s->a0_stride * 4, st));
  for (int i = 0; i < 3; ++i) {
    HIPCHK(hipMemsetAsync(s->cbuf[i], 0, (size_t)2 * s->c_stride[i] * 4, st));
    HIPCHK(hipMemsetAsync(s->sbuf[i], 0, (size_t)2 * s->s_stride[i] * 4, st));
  }
  HIPCHK(hipMemsetAsync(s->kv, 0, (size_t)s->e->cfg.m_layers * 2 * s->kv_plane() * 4, st));
  if (s->pcm_carry) HIPCHK(hipMemsetAsync(s->pcm_carry, 0, (size_t)2 * s->pcm_cstride * 4, st));  // the fused last conv's tile carries
  s->h_frame = 0;
  return 0;
}

// Continuous batching: zero the streaming carries of ONE sequence (a new utterance joins in `row`): the
// previous-frame buffers its first frame will read as halo, and its position in the decoder transformer.
extern "C" int ptts_mimi_state_reset_row(ptts_mimi_state *s, int32_t row, void *stream) {
  if (!s || row < 0 || row >= s->B) return fail(-1, "reset_row: row out of range");
  hipStream_t st = S(s->e, stream);
  const int CF = s->e->cfg.m_dim / 16;
  // bytes per element of each double-buffered conv input: fp32 codec 4; bf16 codec 2; fp8 codec: 1 for the fp8 conv inputs
  // (a0, cbuf, sbuf[0..1]), 2 for the transformer output and the last conv's input.  A sequence owns whole 16-row tiles, i.e.
  // one contiguous block of every buffer, at element offset row * (elements per sequence) inside either parity half.
  const bool h16 = s->e->codec_bf16, f8 = s->e->codec_fp8;
  const int e_tr = h16 ? 2 : 4, e_a0 = f8 ? 1 : h16 ? 2 : 4;
  set_int_kernel<<<1, 64, 0, st>>>(s->offset + row, 1, 0);
  auto zero_block = [&](float *buf, long stride_floats, int par, int esz) -> hipError_t {
    const size_t per_seq = (size_t)stride_floats / s->B;  // elements
    return hipMemsetAsync((char *)buf + (size_t)par * stride_floats * 4 + (size_t)row * per_seq * esz, 0, per_seq * esz, st);
  };
  for (int par = 0; par < 2; ++par) {
    zero_row_fm_kernel<<<cdiv(CF * 16, 256), 256, 0, st>>>(s->zq + par * s->zq_stride, CF, row);
    HIPCHK(zero_block(s->tr_out, s->tr_stride, par, e_tr));
    HIPCHK(zero_block(s->a0, s->a0_stride, par, e_a0));
    for (int i = 0; i < 3; ++i) {
      HIPCHK(zero_block(s->cbuf[i], s->c_stride[i], par, e_a0));
      HIPCHK(zero_block(s->sbuf[i], s->s_stride[i], par, (f8 && i == 2) ? 2 : e_a0));
    }
    if (s->pcm_carry && s->rows[3] % 64 == 0) HIPCHK(zero_block(s->pcm_carry, s->pcm_cstride, par, 4));  // the row's tile carries
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// Optional 16-bit PCM output of the codec's last kernel (device or pinned host memory, [B, frame_samples]);
// applies to the decodes / graph captures issued after the call.  NULL switches it off.
extern "C" int ptts_mimi_set_pcm_i16(ptts_mimi_state *s, int16_t *d_pcm_i16) {
  if (!s) return fail(-1, "null state");
  s->pcm_i16 = d_pcm_i16;
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Reduced-precision codec (ptts_bf16.h): tile choice is static (the kernels are bandwidth / launch bound: bf16 MFMA
// runs at 16x the fp32 rate) - the largest workgroup tile that still yields >= ~2 workgroups per CU.
template <int TN, int TM, int WN, int WM>
static void launch_h_cfg(hipStream_t st, const GemmArgs &a, int pre) {
  const dim3 grid(cdiv(a.NT, TN * WN), cdiv(a.MT, TM * WM)), block(64 * WN * WM);
  const unsigned dyn = lds_pad(0);  // occupancy cap of the codec stream (the kernel itself uses no LDS)
  if (pre == PRE_LNFOLD) gemm_h_kernel<TN, TM, WN, WM, PRE_LNFOLD><<<grid, block, dyn, st>>>(a);
  else gemm_h_kernel<TN, TM, WN, WM, PRE_NONE><<<grid, block, dyn, st>>>(a);
}
static void launch_gemm_h(hipStream_t st, const GemmArgs &a_in, int pre, const Lin &L) {
  GemmArgs a = a_in;
  a.W = (const float *)L.wh;
  a.CF = L.C / 32;
  a.KF = a.CF * L.ntaps;
  if (pre == PRE_LNFOLD) a.ln_s = L.ln_s_h;
  a.swz = 0;
  const double K = (double)a.KF * 32, N = (double)a.NT * 16, M = (double)a.M;
  double bytes = 2.0 * (N * K + M * (double)a.CF * 32 + M * N * (a.Yraw ? 2 : 1)) + (a.epi == EPI_RES ? 2.0 * M * N : 0.0);
  if (a.epi == EPI_QKV) bytes += 2.0 * M * N;  // q / k / v leave as fp32
  static const int tiles[4][4] = {{2, 4, 2, 2}, {2, 2, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2}};
  int pick = 3;
  for (int i = 0; i < 4; ++i) {
    const int *t = tiles[i];
    if (t[0] * t[2] > 2 * a.NT && i < 3) continue;  // mostly padding
    if ((long)cdiv(a.NT, t[0] * t[2]) * cdiv(a.MT, t[1] * t[3]) >= 512 || i == 3) { pick = i; break; }
  }
  static const char *const names[4] = {"gemm_h<2,4,2,2>", "gemm_h<2,2,2,2>", "gemm_h<1,2,2,2>", "gemm_h<1,1,2,2>"};
  const int *t = tiles[pick];
  ProfScope ps(st, std::string(names[pick]) + (pre == PRE_LNFOLD ? "+ln" : "") + "@" +
                       std::to_string((long)cdiv(a.NT, t[0] * t[2]) * cdiv(a.MT, t[1] * t[3]) * 256), bytes, 2.0 * M * N * K);
  switch (pick) {
    case 0: launch_h_cfg<2, 4, 2, 2>(st, a, pre); break;
    case 1: launch_h_cfg<2, 2, 2, 2>(st, a, pre); break;
    case 2: launch_h_cfg<1, 2, 2, 2>(st, a, pre); break;
    default: launch_h_cfg<1, 1, 2, 2>(st, a, pre); break;
  }
}

// the codec frame with bf16 activations: same dataflow, buffers and carries as mimi_enqueue (the fp32 buffers are
// reused, half filled); block counts are per 32 channels
static int mimi_enqueue_h(hipStream_t st, ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm) {
  const ptts_config &c = e->cfg;
  bind_engine(e);
  struct LdsScope { LdsScope(int t) { g_lds_target = t; } ~LdsScope() { g_lds_target = 0; } } lds_scope(e->opt_codec_lds_target);
  const int B = s->B, C = c.m_dim, CB = C / 32, st16 = c.upsample_stride;
  SITE("mimi.prologue");
  {
    ProfScope ps(st, "mimi_prologue", 4.0 * (B * c.ldim + (double)C * c.ldim) + B * C * (8.0 + 2.0 * st16), 2.0 * B * C * (c.ldim + 2.0 * st16));
    const int nb_main = cdiv((long)B * (C / 4) * st16, 256);
    mimi_prologue_kernel<<<nb_main + cdiv(B * st16 * 32, 256), 256, 0, st>>>(
        d_latent, e->emb_std, e->emb_mean, e->quant_w, e->up_w, s->zq, s->zq_stride, s->frame, s->u0, B, c.ldim, C, st16,
        nb_main, RopeArgs{s->offset, e->freq_mimi, s->rope, B * st16, st16}, 1);
  }
  const int M16 = B * st16, MT16 = s->MT16;
  GemmArgs a;
  for (int l = 0; l < c.m_layers; ++l) {
    const TrLayer &T = e->mm[l];
    float *x_in = l == 0 ? s->u0 : s->u;
    const bool last = l == c.m_layers - 1;
    SITE("mimi.qkv");
    a = mk_gemm(T.qkv, x_in, CB, MT16, M16);
    a.epi = EPI_QKV;
    a.Q = s->q; a.Kc = s->K(l); a.Vc = s->V(l); a.offset = s->offset; a.rope = s->rope;
    a.H = c.m_heads; a.Tq = st16; a.QB = 1; a.cap = e->ring; a.ring = e->ring;
    launch_gemm_h(st, a, PRE_LNFOLD, T.qkv);
    AttnArgs at;
    at.Q = s->q; at.Kc = s->K(l); at.Vc = s->V(l); at.offset = s->offset; at.H = c.m_heads; at.Tq = st16; at.QB = 1;
    at.cap = e->ring; at.ring = e->ring; at.ctx = c.m_context; at.splits = s->splits; at.part = s->part; at.Y = s->ao; at.YF = CB;
    at.h16 = 1;
    const int BH = B * c.m_heads;
    SITE("mimi.attn");
    {
      const double keys = (double)B * std::min(e->ring, (s->h_frame + 1) * st16);
      ProfScope ps(st, "attn@" + std::to_string((long)BH * s->splits * 64 * attn_nw(BH)), keys * c.m_heads * 64 * 4 * 2 + 6.0 * M16 * C, 4.0 * keys * c.m_heads * 64 * 16);
      launch_attn(st, at, BH);
    }
    if (s->splits > 1) {
      ProfScope ps(st, "attn_combine", (double)BH * s->splits * 16 * ATT_PSTRIDE * 4, 0);
      attn_combine_kernel<<<dim3(BH, 1), 256, 0, st>>>(at);
    }
    SITE("mimi.out");
    a = mk_gemm(T.out, s->ao, CB, MT16, M16);
    a.epi = EPI_RES; a.R = x_in; a.RF = CB; a.Y = s->u; a.YF = CB; a.ls = T.ls1;
    launch_gemm_h(st, a, PRE_NONE, T.out);
    SITE("mimi.ff1");
    a = mk_gemm(T.ff1, s->u, CB, MT16, M16);
    a.epi = EPI_STORE; a.act = ACT_GELU; a.Y = s->ff; a.YF = c.m_ff / 32;
    launch_gemm_h(st, a, PRE_LNFOLD, T.ff1);
    SITE("mimi.ff2");
    a = mk_gemm(T.ff2, s->ff, c.m_ff / 32, MT16, M16);
    a.epi = EPI_RES; a.R = s->u; a.RF = CB; a.ls = T.ls2;
    a.Y = last ? s->tr_out : s->u; a.YF = CB; a.Ydstride = last ? 2 * s->tr_stride : 0; a.par = last ? s->frame : nullptr;
    launch_gemm_h(st, a, PRE_NONE, T.ff2);
  }
  // strides of the frame-parity double buffers: the float buffers hold bf16 (or, for the inputs of the fp8 convolutions,
  // e4m3 bytes), so a parity step of `stride` floats is 2 * stride bf16 elements / 4 * stride bytes
  const bool f8 = e->codec_fp8;
  const float *f8s = e->f8s;
  // fp8 conv: operand images, scales and output format (sc_in / sc_out index e->f8s; sc_out < 0: bf16 output)
  auto f8_conv = [&](GemmArgs &g, const Lin &L, int sc_in, int sc_out) {
    g.W = (const float *)L.wf8;
    g.wscale = L.wscale8;
    g.CF = L.C / 32;
    g.KF = g.CF * L.ntaps;
    g.swz = 0;
    g.xs = f8s[sc_in];
    g.yf8 = sc_out >= 0;
    g.yinv = sc_out >= 0 ? 1.0f / f8s[sc_out] : 1.0f;
    const double K = (double)g.KF * 32, N = (double)g.NT * 16, M = (double)g.M;
    ProfScope ps(st, "gemm_f8@" + std::to_string((long)g.NT * g.MT), N * K + M * g.CF * 32.0 + M * N * (g.yf8 ? 1 : 2) + (g.Yraw ? 2.0 * M * N : 0.0) +
                         (g.epi == EPI_RES ? 2.0 * M * N : 0.0), 2.0 * M * N * K);
    launch_gemm_f8(st, g, lds_pad(0));
  };
  int mult = 8;
  SITE("seanet.conv0");
  a = mk_gemm(e->conv0, s->tr_out, CB, MT16, M16);
  a.Xdstride = 2 * s->tr_stride; a.T = s->rows[0]; a.par = s->frame;
  a.Y = s->a0; a.Ydstride = (f8 ? 4 : 2) * s->a0_stride; a.YF = mult * c.n_filters / 32; a.act = ACT_ELU;
  if (f8) { a.yf8 = 1; a.yinv = 1.0f / f8s[0]; }  // bf16 in (the transformer's output), e4m3 out
  launch_gemm_h(st, a, PRE_NONE, e->conv0);
  const float *xin = s->a0;
  long xds = (f8 ? 4 : 2) * s->a0_stride;
  for (int i = 0; i < 3; ++i) {
    const int cin = mult * c.n_filters, cout = cin / 2, hid = cout / c.compress;
    const int Tin = s->rows[i], Tout = s->rows[i + 1];
    const int MTin = B * Tin / 16, MTout = B * Tout / 16;
    static const char *sn[3][3] = {{"seanet.convtr1", "seanet.res1a", "seanet.res1b"},
                                   {"seanet.convtr2", "seanet.res2a", "seanet.res2b"},
                                   {"seanet.convtr3", "seanet.res3a", "seanet.res3b"}};
    const int es = f8 ? 4 : 2;  // elements per float of stride for the buffers that follow the codec's operand format
    const bool s_bf16 = !f8 || i == 2;  // the last block's output feeds the (bf16-input) last conv
    SITE(sn[i][0]);
    a = mk_gemm(e->convtr[i], xin, cin / 32, MTin, B * Tin);
    a.Xdstride = xds; a.T = Tin; a.par = s->frame;
    a.epi = EPI_CONVTR; a.cout = cout; a.stride = c.ratios[i];
    a.Y = s->cbuf[i]; a.Ydstride = es * s->c_stride[i]; a.YF = cout / 32; a.act = ACT_ELU;
    a.Yraw = s->craw[i]; a.Yrawdstride = 0;
    if (f8) f8_conv(a, e->convtr[i], i == 0 ? 0 : 3 * i, 1 + 3 * i);
    else launch_gemm_h(st, a, PRE_NONE, e->convtr[i]);
    SITE(sn[i][1]);
    a = mk_gemm(e->res_a[i], s->cbuf[i], cout / 32, MTout, B * Tout);
    a.Xdstride = es * s->c_stride[i]; a.T = Tout; a.par = s->frame;
    a.Y = s->rbuf[i]; a.YF = hid / 32; a.act = ACT_ELU;
    if (f8) f8_conv(a, e->res_a[i], 1 + 3 * i, 2 + 3 * i);
    else launch_gemm_h(st, a, PRE_NONE, e->res_a[i]);
    SITE(sn[i][2]);
    a = mk_gemm(e->res_b[i], s->rbuf[i], hid / 32, MTout, B * Tout);
    a.T = Tout; a.par = s->frame;
    a.epi = EPI_RES; a.R = s->craw[i]; a.Rdstride = 0; a.RF = cout / 32; a.act = ACT_ELU;
    a.Y = s->sbuf[i]; a.Ydstride = (s_bf16 ? 2 : 4) * s->s_stride[i]; a.YF = cout / 32;
    if (f8) f8_conv(a, e->res_b[i], 2 + 3 * i, s_bf16 ? -1 : 3 + 3 * i);
    else launch_gemm_h(st, a, PRE_NONE, e->res_b[i]);
    xin = s->sbuf[i];
    xds = (s_bf16 ? 2 : 4) * s->s_stride[i];
    mult /= 2;
  }
  const int Tl = s->rows[3];
  SITE("seanet.conv_last");
  a = mk_gemm(e->conv_last, xin, c.n_filters / 32, B * Tl / 16, B * Tl);
  a.CF = c.n_filters / 32;
  a.Xdstride = xds; a.T = Tl; a.par = s->frame;
  a.epi = EPI_PCM; a.pcm = d_pcm ? d_pcm : s->pcm_dbg; a.pcm_i16 = s->pcm_i16;
  {
    ProfScope ps(st, "pcm_conv_h", 2.0 * a.M * c.n_filters + 4.0 * a.M, 2.0 * a.M * c.n_filters * c.last_kernel_size);
    pcm_conv_h_kernel<<<cdiv(a.M, 256), 256, 0, st>>>(a, e->conv_last_w, e->conv_last_b);
  }
  SITE("mimi.tail");
  {
    ProfScope ps(st, "step_tail", 8.0 * B, 0);
    step_tail_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->offset, B, st16, s->frame, nullptr);
  }
  SITE("");
  return 0;
}

static int mimi_enqueue(hipStream_t st, ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm) {
  if (e->codec_bf16) return mimi_enqueue_h(st, e, s, d_latent, d_pcm);
  const ptts_config &c = e->cfg;
  bind_engine(e);
  struct SplitScope { SplitScope(bool on) { g_use_split = on; } ~SplitScope() { g_use_split = false; } } split_scope(e->codec_split);
  struct LdsScope { LdsScope(int t) { g_lds_target = t; } ~LdsScope() { g_lds_target = 0; } } lds_scope(e->opt_codec_lds_target);
  const int B = s->B, C = c.m_dim, CF = C / 16, LF = c.ldim / 16, st16 = c.upsample_stride;
  SITE("mimi.prologue");  // de-normalise + quantizer 1x1 conv + depthwise x16 upsample + RoPE table: one launch
  {
    ProfScope ps(st, "mimi_prologue", 4.0 * (B * c.ldim + (double)C * c.ldim + B * C * (2.0 + st16)), 2.0 * B * C * (c.ldim + 2.0 * st16));
    const int nb_main = cdiv((long)B * (C / 4) * st16, 256);
    mimi_prologue_kernel<<<nb_main + cdiv(B * st16 * 32, 256), 256, 0, st>>>(
        d_latent, e->emb_std, e->emb_mean, e->quant_w, e->up_w, s->zq, s->zq_stride, s->frame, s->u0, B, c.ldim, C, st16,
        nb_main, RopeArgs{s->offset, e->freq_mimi, s->rope, B * st16, st16}, 0);
  }
  GemmArgs a;
  const int M16 = B * st16;

  for (int l = 0; l < c.m_layers; ++l) {
    TrCtx t;
    t.D = C; t.H = c.m_heads; t.FF = c.m_ff; t.MT = s->MT16; t.M = M16; t.Tq = st16; t.QB = 1;
    t.cap = e->ring; t.ring = e->ring; t.ctx = c.m_context; t.splits = s->splits;
    t.x_in = l == 0 ? s->u0 : s->u; t.x = s->u;
    const bool last = l == c.m_layers - 1;
    t.x_out = last ? s->tr_out : s->u; t.out_ds = last ? s->tr_stride : 0; t.par = last ? s->frame : nullptr;
    t.h = s->h; t.ao = s->ao; t.ff = s->ff; t.q = s->q; t.part = s->part;
    t.Kc = s->K(l); t.Vc = s->V(l); t.offset = s->offset; t.rope = s->rope;
    t.kv_keys = (double)B * std::min(e->ring, (s->h_frame + 1) * st16);
    t.tag = "mimi";
    run_tr_layer(st, e->mm[l], t);
  }
  // SEANet decoder (seanet.py:141-180) as implicit GEMMs over (sequence, time) rows
  int mult = 8;
  bool fused_pcm = false;
  SITE("seanet.conv0");
  a = mk_gemm(e->conv0, s->tr_out, CF, s->MT16, M16);
  a.Xdstride = s->tr_stride; a.T = s->rows[0]; a.par = s->frame;
  a.Y = s->a0; a.Ydstride = s->a0_stride; a.YF = mult * c.n_filters / 16;
  a.act = ACT_ELU;  // every SEANet conv input is ELU(previous output): apply it once, in the producer
  launch_gemm(st, a, PRE_NONE);
  const float *xin = s->a0;
  long xds = s->a0_stride;
  for (int i = 0; i < 3; ++i) {
    const int cin = mult * c.n_filters, cout = cin / 2, hid = cout / c.compress;
    const int Tin = s->rows[i], Tout = s->rows[i + 1];
    const int MTin = B * Tin / 16, MTout = B * Tout / 16;
    static const char *sn[3][3] = {{"seanet.convtr1", "seanet.res1a", "seanet.res1b"},
                                   {"seanet.convtr2", "seanet.res2a", "seanet.res2b"},
                                   {"seanet.convtr3", "seanet.res3a", "seanet.res3b"}};
    SITE(sn[i][0]);
    a = mk_gemm(e->convtr[i], xin, cin / 16, MTin, B * Tin);
    a.Xdstride = xds; a.T = Tin; a.par = s->frame;
    a.epi = EPI_CONVTR; a.cout = cout; a.stride = c.ratios[i];
    a.Y = s->cbuf[i]; a.Ydstride = s->c_stride[i]; a.YF = cout / 16; a.act = ACT_ELU;
    a.Yraw = s->craw[i]; a.Yrawdstride = 0;  // raw value = the resnet block's skip input
    // "single_store": the transposed conv stores its RAW output once (the block's skip input); the k3 conv that follows
    // applies ELU to its operand fragments as it reads them (PRE_ELU).  Saves one write and one read of every stage output
    // (107 MB per 64-sequence frame): +1.x % pipelined; fp32 weights only
    const bool single = e->opt_single_store && !g_use_split && !e->res_a[i].wq && !e->res_a[i].wb16;
    const int pre_a = single ? PRE_ELU : PRE_NONE;
    if (single) { a.act = ACT_NONE; a.Yraw = nullptr; }
    launch_gemm(st, a, PRE_NONE);
    if (e->opt_fuse_res && resblock_fusable(e->res_a[i], e->res_b[i], MTout) && (long)MTout * 16 >= e->fuse_res_min_rows) {
      SITE(sn[i][1]);
      a = mk_gemm(e->res_a[i], s->cbuf[i], cout / 16, MTout, B * Tout);
      a.Xdstride = s->c_stride[i]; a.T = Tout; a.par = s->frame; a.act = ACT_ELU;
      a.R = s->craw[i]; a.Rdstride = 0; a.RF = cout / 16; a.act2 = ACT_ELU;
      if (single) { a.R = s->cbuf[i]; a.Rdstride = s->c_stride[i]; }
      a.Y = s->sbuf[i]; a.Ydstride = s->s_stride[i]; a.YF = cout / 16;
      // a split-bf16 engine keeps the FUSED fp32 residual blocks: the unfused split pair is slower (stage 3: 34 + 35 us
      // against 43 us fused, stage 2: 25 + 21 against 32; gpurun_out r3 profile), the block is bound by its activations
      a.W = e->res_a[i].w; a.Wq = nullptr; a.wfmt = 0;
      // last stage: SEANet's final conv (k = 3, n_filters -> 1 sample) rides in the block's epilogue: the block's output
      // (31 MB per 64-sequence frame, written and read back by a separate conv before) never leaves the CU
      const Lin &CL = e->conv_last;
      fused_pcm = e->opt_fuse_pcm && i == 2 && e->res_a[i].NT == 2 && Tout % 64 == 0 && MTout % 4 == 0 && cout == 64 &&
                  CL.ntaps == 3 && CL.NT == 1 && !CL.wq && !CL.wb16 && !g_use_split && s->pcm_carry;
      if (fused_pcm) {
        a.pcm_w = CL.w; a.pcm_part = s->pcm_part; a.pcm_carry = s->pcm_carry; a.pcm_cstride = s->pcm_cstride;
        if (!e->opt_debug_taps) a.Y = nullptr;
      }
      launch_resblock(st, a, e->res_b[i], pre_a);
      xin = s->sbuf[i];
      xds = s->s_stride[i];
      mult /= 2;
      continue;
    }
    SITE(sn[i][1]);
    a = mk_gemm(e->res_a[i], s->cbuf[i], cout / 16, MTout, B * Tout);
    a.Xdstride = s->c_stride[i]; a.T = Tout; a.par = s->frame;
    a.Y = s->rbuf[i]; a.YF = hid / 16; a.act = ACT_ELU;
    launch_gemm(st, a, pre_a);
    SITE(sn[i][2]);
    a = mk_gemm(e->res_b[i], s->rbuf[i], hid / 16, MTout, B * Tout);
    a.T = Tout; a.par = s->frame;
    a.epi = EPI_RES; a.R = s->craw[i]; a.Rdstride = 0; a.RF = cout / 16; a.act = ACT_ELU;
    if (single) { a.R = s->cbuf[i]; a.Rdstride = s->c_stride[i]; }
    a.Y = s->sbuf[i]; a.Ydstride = s->s_stride[i]; a.YF = cout / 16;
    launch_gemm(st, a, PRE_NONE);
    xin = s->sbuf[i];
    xds = s->s_stride[i];
    mult /= 2;
  }
  const int Tl = s->rows[3];
  SITE("seanet.conv_last");
  if (fused_pcm) {
    ProfScope ps(st, "pcm_fix", 4.0 * (2.0 * B * Tl + B * Tl / 32.0), 0);
    pcm_fix_kernel<<<cdiv(B * Tl, 256), 256, 0, st>>>(s->pcm_part, s->pcm_carry, s->pcm_cstride, s->frame, e->conv_last.bias,
                                                      d_pcm ? d_pcm : s->pcm_dbg, s->pcm_i16, B * Tl, Tl / 64);
  } else {
    a = mk_gemm(e->conv_last, xin, c.n_filters / 16, B * Tl / 16, B * Tl);
    a.Xdstride = xds; a.T = Tl; a.par = s->frame;
    a.epi = EPI_PCM; a.pcm = d_pcm ? d_pcm : s->pcm_dbg; a.pcm_i16 = s->pcm_i16;
    launch_gemm(st, a, PRE_NONE);
  }
  SITE("mimi.tail");
  {
    ProfScope ps(st, "step_tail", 8.0 * B, 0);
    step_tail_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->offset, B, st16, s->frame, nullptr);
  }
  SITE("");
  return 0;
}

extern "C" int ptts_mimi_decode(ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm, void *stream) {
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  if (!d_latent) return fail(-1, "null latent");
  CHK(mimi_enqueue(S(e, stream), e, s, d_latent, d_pcm));
  s->h_frame += 1;
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Tile autotuning for one batch size (see Tuner).  Runs on scratch states; the caller's states are untouched.
extern "C" int ptts_tune(ptts_engine *e, int32_t B, void *stream) { return ptts_tune_streams(e, B, stream, stream); }
// The FlowLM step is tuned on `lm_stream`, the codec frame on `codec_stream`: with CU-masked streams
// (ptts_stream_create_masked) each stage gets the tiles that are fastest on ITS share of the chip.
extern "C" int ptts_tune_streams(ptts_engine *e, int32_t B, void *lm_stream, void *codec_stream) {
  if (!e || B < 1) return fail(-1, "bad argument");
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  hipStream_t st = S(e, lm_stream), st2 = S(e, codec_stream);
  Tuner &t = *e->tuner;
  ptts_lm_state *ls = nullptr;
  ptts_mimi_state *ms = nullptr;
  int rc = ptts_lm_state_create(e, B, 32, &ls);
  if (rc == 0) rc = ptts_mimi_state_create(e, B, &ms);
  t.flush_bytes = (size_t)320 << 20;  // > Infinity Cache (256 MiB)
  if (rc == 0) {
    if (hipMalloc(&t.flush, t.flush_bytes) != hipSuccess) { t.flush = nullptr; (void)hipGetLastError(); }  // tune warm if memory is short
    else (void)hipMemsetAsync(t.flush, 0, t.flush_bytes, st);
  }
  if (rc == 0 && (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess)) rc = fail(-2, "hipEventCreate");
  if (rc == 0) {
    HIPCHK(hipStreamSynchronize(e->stream));  // state zero-fills
    const bool prof = e->prof.on;
    e->prof.on = false;
    t.active = true;
    rc = ptts_lm_decode_step(e, ls, nullptr, nullptr, 1, 1e30f, nullptr, nullptr, nullptr, st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == 0) rc = fail(-2, "tune: stream error");
    if (rc == 0) rc = ptts_mimi_decode(e, ms, ls->lat, nullptr, st2);
    t.active = false;
    e->prof.on = prof;
    if (hipStreamSynchronize(st2) != hipSuccess && rc == 0) rc = fail(-2, "tune: stream error");
  }
  if (t.e0) hipEventDestroy(t.e0);
  if (t.e1) hipEventDestroy(t.e1);
  t.e0 = t.e1 = nullptr;
  if (t.flush) hipFree(t.flush);
  t.flush = nullptr;
  if (ms) ptts_mimi_state_destroy(ms);
  if (ls) ptts_lm_state_destroy(ls);
  return rc;
}

// Same for the prefill GEMM shapes of `batch` sequences x `t` positions (first-chunk path: text prefill of a chunk).
extern "C" int ptts_tune_prefill(ptts_engine *e, int32_t B, int32_t T, void *stream) {
  if (!e || B < 1 || T < 1) return fail(-1, "bad argument");
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  hipStream_t st = S(e, stream);
  Tuner &t = *e->tuner;
  ptts_lm_state *ls = nullptr;
  float *emb = nullptr;
  int rc = ptts_lm_state_create(e, B, T + 16, &ls);
  t.flush_bytes = (size_t)320 << 20;
  if (rc == 0 && hipMalloc((void **)&emb, (size_t)B * T * e->cfg.d_model * 4) != hipSuccess) { (void)hipGetLastError(); rc = fail(-2, "hipMalloc"); }
  if (rc == 0) {
    (void)hipMemsetAsync(emb, 0, (size_t)B * T * e->cfg.d_model * 4, st);
    if (hipMalloc(&t.flush, t.flush_bytes) != hipSuccess) { t.flush = nullptr; (void)hipGetLastError(); }
    else (void)hipMemsetAsync(t.flush, 0, t.flush_bytes, st);
  }
  if (rc == 0 && (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess)) rc = fail(-2, "hipEventCreate");
  if (rc == 0) {
    HIPCHK(hipStreamSynchronize(e->stream));
    const bool prof = e->prof.on;
    e->prof.on = false;
    t.active = true;
    rc = ptts_lm_prefill(e, ls, emb, T, st);
    t.active = false;
    e->prof.on = prof;
    if (hipStreamSynchronize(st) != hipSuccess && rc == 0) rc = fail(-2, "tune_prefill: stream error");
  }
  if (t.e0) hipEventDestroy(t.e0);
  if (t.e1) hipEventDestroy(t.e1);
  t.e0 = t.e1 = nullptr;
  if (t.flush) hipFree(t.flush);
  t.flush = nullptr;
  if (emb) hipFree(emb);
  if (ls) ptts_lm_state_destroy(ls);
  return rc;
}
extern "C" int ptts_tune_version(void) { return kTuneVersion; }
extern "C" const char *ptts_tune_log(ptts_engine *e) { return e ? e->tuner->log.c_str() : ""; }

// The tuned table as text, one line per shape: the 13 key integers (tune_key) then the configuration index.
extern "C" int64_t ptts_tune_export(ptts_engine *e, char *h_out, int64_t capacity) {
  if (!e) return fail(-1, "null engine");
  ENGINE_LOCK(e);
  std::string out;
  char line[256];
  for (auto &kv : e->tuner->table) {
    int n = 0;
    for (int v : kv.first) n += snprintf(line + n, sizeof line - n, "%d ", v);
    snprintf(line + n, sizeof line - n, "%d\n", kv.second);
    out += line;
  }
  if ((int64_t)out.size() + 1 > capacity) return fail(-1, "tune_export: buffer too small");
  memcpy(h_out, out.c_str(), out.size() + 1);
  return (int64_t)out.size();
}

extern "C" int ptts_tune_import(ptts_engine *e, const char *text) {
  if (!e || !text) return fail(-1, "null argument");
  ENGINE_LOCK(e);
  const char *p = text;
  int n_ok = 0;
  while (*p) {
    TuneKey k;
    int cfg = -1, consumed = 0, ok = 1;
    for (int i = 0; i < 13 && ok; ++i) {
      if (sscanf(p, "%d%n", &k[i], &consumed) != 1) ok = 0;
      else p += consumed;
    }
    if (ok && sscanf(p, "%d%n", &cfg, &consumed) == 1) {
      p += consumed;
      if (cfg >= 0 && cfg < kNumCfg) { e->tuner->table[k] = cfg; ++n_ok; }
    } else {
      ok = 0;
    }
    while (*p && *p != '\n') ++p;
    if (*p) ++p;
    if (!ok && !*p) break;
  }
  return n_ok;
}

extern "C" void ptts_tune_clear(ptts_engine *e) {
  if (e) { e->tuner->table.clear(); e->tuner->log.clear(); }
}

// ------------------------------------------------------------------------------------------------
// Voice-prompt encode path (one-off per voice): MimiModel.encode_to_latent + speaker projection
// (reference mimi.py:96-119, tts_model.py:379-388).  Whole-signal causal convs = the same implicit GEMMs with
// zero / replicate left padding and an input stride.
extern "C" int ptts_encode_voice(ptts_engine *e, const float *d_audio, int64_t n_samples, float *d_latent_out,
                                 float *d_cond_out, int32_t *h_frames, void *stream) {
  ENGINE_LOCK(e);
  if (!e->has_encoder) return fail(-3, "the checkpoint passed to ptts_create has no Mimi encoder tensors");
  if (n_samples < 1) return fail(-1, "empty audio");
  HIPCHK(hipSetDevice(e->device));
  const ptts_config &c = e->cfg;
  hipStream_t st = S(e, stream);
  bind_engine(e);
  AllocScope alloc_scope(st);
  const int hop = c.ratios[0] * c.ratios[1] * c.ratios[2];
  const long fs = (long)hop * c.upsample_stride;  // 1920
  const long R0 = (n_samples + fs - 1) / fs * fs;  // pad_for_conv1d(x, frame_size, frame_size)
  const int Tf = (int)(R0 / fs);
  const int C = c.m_dim, nf = c.n_filters;
  std::vector<void *> tmp;
  auto talloc = [&](float **p, size_t n) { int r = dallocT(nullptr, p, n); if (r == 0) tmp.push_back(*p); return r; };
  float *x0, *cur_r, *cur_e;
  CHK(talloc(&x0, (size_t)R0 * 16));
  CHK(talloc(&cur_r, (size_t)R0 * nf));
  CHK(talloc(&cur_e, (size_t)R0 * nf));
  audio_to_fm_kernel<<<cdiv(R0 / 16 * 64, 256), 256, 0, st>>>(d_audio, x0, (int)n_samples, (int)R0);
  long rows = R0;
  auto conv = [&](const Lin &L, const float *X, int XF, long out_rows, int xstride, int halo, int mode) {
    GemmArgs a = mk_gemm(L, X, XF, (int)cdiv(out_rows, 16), (int)out_rows);
    a.T = cdiv(out_rows, 16) * 16;  // one sequence: t == row
    a.xstride = xstride; a.halo = halo; a.halo_mode = mode;
    return a;
  };
  SITE("enc.conv0");
  GemmArgs a = conv(e->enc_conv0, x0, 1, rows, 1, c.kernel_size - 1, 1);
  a.Y = cur_e; a.YF = nf / 16; a.act = ACT_ELU; a.Yraw = cur_r;
  launch_gemm(st, a, PRE_NONE);
  int dim = nf;
  for (int i = 0; i < 3; ++i) {
    const int r = c.ratios[2 - i], hid = dim / c.compress;
    float *rb, *se, *nr, *ne;
    CHK(talloc(&rb, (size_t)rows * hid));
    CHK(talloc(&se, (size_t)rows * dim));
    SITE("enc.res_a");
    a = conv(e->enc_res_a[i], cur_e, dim / 16, rows, 1, c.res_kernel_size - 1, 1);
    a.Y = rb; a.YF = hid / 16; a.act = ACT_ELU;
    launch_gemm(st, a, PRE_NONE);
    SITE("enc.res_b");
    a = conv(e->enc_res_b[i], rb, hid / 16, rows, 1, 0, 1);
    a.epi = EPI_RES; a.R = cur_r; a.RF = dim / 16; a.Y = se; a.YF = dim / 16; a.act = ACT_ELU;
    launch_gemm(st, a, PRE_NONE);
    const long nrows = rows / r;
    CHK(talloc(&nr, (size_t)nrows * 2 * dim));
    CHK(talloc(&ne, (size_t)nrows * 2 * dim));
    SITE("enc.down");
    a = conv(e->enc_down[i], se, dim / 16, nrows, r, r, 1);  // kernel 2r, stride r: left context r
    a.Y = ne; a.YF = 2 * dim / 16; a.act = ACT_ELU; a.Yraw = nr;
    launch_gemm(st, a, PRE_NONE);
    cur_r = nr; cur_e = ne; rows = nrows; dim *= 2;
  }
  // final conv -> encoder-transformer input (rows = R0 / hop, a multiple of 16)
  const int M2 = (int)rows, MT2 = M2 / 16;
  const long ds_rows_in = (long)cdiv(Tf, 16) * 16 * c.upsample_stride;  // rows the downsample may touch
  float *tx;
  CHK(talloc(&tx, (size_t)std::max<long>(M2, ds_rows_in) * C));
  SITE("enc.final");
  a = conv(e->enc_final, cur_e, dim / 16, M2, 1, c.last_kernel_size - 1, 1);
  a.Y = tx; a.YF = C / 16;
  launch_gemm(st, a, PRE_NONE);
  // encoder transformer, whole sequence, RoPE offset 0, window = context (transformer.py:63-75)
  {
    const int H = c.m_heads, QB = MT2, cap = MT2 * 16;
    float *ao, *ff, *q, *part, *rope, *kv;
    int *off;
    const int splits = attn_splits(H * QB, std::min(MT2, cdiv(c.m_context, 16) + 2));
    CHK(talloc(&ao, (size_t)M2 * C));
    CHK(talloc(&ff, (size_t)M2 * c.m_ff));
    CHK(talloc(&q, (size_t)H * QB * 4 * 256));
    CHK(talloc(&part, (size_t)H * QB * splits * 16 * ATT_PSTRIDE));
    CHK(talloc(&rope, (size_t)M2 * 64));
    CHK(talloc(&kv, (size_t)2 * H * cap * 64));
    CHK(talloc((float **)&off, 4));
    rope_table_kernel<<<cdiv(M2 * 32, 256), 256, 0, st>>>(off, e->freq_mimi, rope, M2, M2);
    for (int l = 0; l < c.m_layers; ++l) {
      TrCtx t;
      t.D = C; t.H = H; t.FF = c.m_ff; t.MT = MT2; t.M = M2; t.Tq = M2; t.QB = QB;
      t.cap = cap; t.ring = 0; t.ctx = c.m_context; t.splits = splits;
      t.x_in = tx; t.x = tx; t.x_out = tx; t.out_ds = 0; t.par = nullptr;
      t.h = nullptr; t.ao = ao; t.ff = ff; t.q = q; t.part = part;
      t.Kc = kv; t.Vc = kv + (size_t)H * cap * 64; t.offset = off; t.rope = rope;
      t.kv_keys = (double)M2 * std::min(M2, c.m_context) / 16.0;
      t.tag = "enc";
      run_tr_layer(st, e->enc_tr[l], t);
    }
  }
  // ConvDownsample1d: kernel 2s, stride s, replicate padding, no bias (resample.py:7-29)
  float *lat, *cond;
  const int MT3 = cdiv(Tf, 16);
  CHK(talloc(&lat, (size_t)MT3 * 16 * c.ldim));
  CHK(talloc(&cond, (size_t)MT3 * 16 * c.d_model));
  SITE("enc.downsample");
  a = conv(e->enc_downsample, tx, C / 16, Tf, c.upsample_stride, c.upsample_stride, 2);
  a.Y = lat; a.YF = c.ldim / 16;
  launch_gemm(st, a, PRE_NONE);
  SITE("enc.speaker_proj");
  a = mk_gemm(e->speaker_proj, lat, c.ldim / 16, MT3, Tf);
  a.Y = cond; a.YF = c.d_model / 16;
  launch_gemm(st, a, PRE_NONE);
  SITE("");
  if (d_latent_out) from_fm_kernel<<<cdiv((long)Tf * c.ldim / 4, 256), 256, 0, st>>>(lat, d_latent_out, Tf, c.ldim, c.ldim / 16, 0);
  if (d_cond_out) from_fm_kernel<<<cdiv((long)Tf * c.d_model / 4, 256), 256, 0, st>>>(cond, d_cond_out, Tf, c.d_model, c.d_model / 16, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  for (void *p : tmp) hipFree(p);
  if (h_frames) *h_frames = Tf;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// hipGraph capture
template <typename F>
static int capture_into(ptts_graph *g, hipGraph_t *graph, hipGraphExec_t *exec, F &&body) {
  if (!g->cap_stream) HIPCHK(hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal));
  int r = body(g->cap_stream);
  hipError_t er = hipStreamEndCapture(g->cap_stream, graph);
  if (r < 0) return r;
  HIPCHK(er);
  HIPCHK(hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0));
  return 0;
}
// a graph with an FlowLM step inside is captured twice when its state may come to share prefixes: rows on their own / cascade
template <class F>
static int capture(ptts_engine *e, ptts_graph *g, F &&body) {
  ptts_lm_state *s = g->lm;
  const bool two = s && e->opt_share_prefix && e->opt_cascade && !e->opt_lm_cluster && s->B >= 16;
  if (s) s->casc_mode = 0;
  int r = capture_into(g, &g->graph, &g->exec, body);
  if (r >= 0 && two) {
    s->casc_mode = 1;
    r = capture_into(g, &g->graph_c, &g->exec_c, body);
  }
  if (s) s->casc_mode = -1;
  return r;
}

extern "C" int ptts_graph_capture_lm_step(ptts_engine *e, ptts_lm_state *s, const float *d_noise, int32_t lsd_steps,
                                          float eos_threshold, float *d_latent_out, float *d_eos_logit,
                                          uint8_t *d_is_eos, ptts_graph **out) {
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  CHK(prepare_lsd(e, lsd_steps));
  CHK(ensure_flow(e, s, lsd_steps, e->stream));
  CHK(ensure_lm_cluster(e, s, e->stream));
  ptts_graph *g = new ptts_graph();
  g->lm = s;
  const int rc = capture(e, g, [&](hipStream_t st) {
    return lm_step_enqueue(st, e, s, nullptr, d_noise, lsd_steps, eos_threshold, d_latent_out, d_eos_logit, d_is_eos);
  });
  if (rc < 0) { g->lm = nullptr; ptts_graph_destroy(g); return rc; }
  g->coop_wgs = s->coop_wgs;
  s->n_graphs += 1;
  *out = g;
  return 0;
}

extern "C" int ptts_graph_capture_mimi(ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm,
                                       ptts_graph **out) {
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  ptts_graph *g = new ptts_graph();
  g->mimi = s;
  const int rc = capture(e, g, [&](hipStream_t st) { return mimi_enqueue(st, e, s, d_latent, d_pcm); });
  if (rc < 0) { g->mimi = nullptr; ptts_graph_destroy(g); return rc; }
  *out = g;
  return 0;
}

// One graph, two parallel branches: FlowLM step (writes d_latent_out) || Mimi decode of the PREVIOUS frame
// (reads d_mimi_latent_in, a different buffer).  Fork/join inside the capture, so a replay needs no
// cross-stream events: consecutive launches on one stream give step t+1 || frame t.
extern "C" int ptts_graph_capture_pipelined(ptts_engine *e, ptts_lm_state *s, ptts_mimi_state *m, const float *d_noise,
                                            int32_t lsd_steps, float eos_threshold, float *d_latent_out,
                                            float *d_eos_logit, uint8_t *d_is_eos, const float *d_mimi_latent_in,
                                            float *d_pcm, ptts_graph **out) {
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  CHK(prepare_lsd(e, lsd_steps));
  CHK(ensure_flow(e, s, lsd_steps, e->stream));
  CHK(ensure_lm_cluster(e, s, e->stream));
  ptts_graph *g = new ptts_graph();
  g->lm = s;
  g->mimi = m;
  hipStream_t side;
  hipEvent_t fork, join;
  HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  int rc = capture(e, g, [&](hipStream_t st) {
    if (hipEventRecord(fork, st) != hipSuccess) return fail(-2, "fork record");
    if (hipStreamWaitEvent(side, fork, 0) != hipSuccess) return fail(-2, "fork wait");
    int r = lm_step_enqueue(st, e, s, nullptr, d_noise, lsd_steps, eos_threshold, d_latent_out, d_eos_logit, d_is_eos);
    if (r < 0) return r;
    r = mimi_enqueue(side, e, m, d_mimi_latent_in, d_pcm);
    if (r < 0) return r;
    if (hipEventRecord(join, side) != hipSuccess) return fail(-2, "join record");
    if (hipStreamWaitEvent(st, join, 0) != hipSuccess) return fail(-2, "join wait");
    return 0;
  });
  hipEventDestroy(fork);
  hipEventDestroy(join);
  hipStreamDestroy(side);
  if (rc < 0) { g->lm = nullptr; g->mimi = nullptr; ptts_graph_destroy(g); return rc; }
  g->coop_wgs = s->coop_wgs;
  s->n_graphs += 1;
  *out = g;
  return 0;
}

extern "C" int ptts_graph_launch(ptts_graph *g, void *stream) {
  ptts_engine *e = g->lm ? g->lm->e : g->mimi->e;
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));  // the calling thread may have another device current (scheduler threads of cuda:N > 0)
  if (g->lm) {
    for (int b = 0; b < g->lm->B; ++b)
      if (g->lm->h_off[b] + 1 > g->lm->cap) return fail(-5, "decode: KV cache capacity exceeded");
  }
  {
    if (g->coop_wgs > 0 && g->cu_checked != S(e, stream)) {
      if (g->coop_wgs > stream_cu_count(S(e, stream), e->n_cus))
        return fail(-1, "graph launch: the stream's CU mask is smaller than the cooperative launch captured in this graph");
      g->cu_checked = S(e, stream);
    }
    CoopGuard guard(e->device, S(e, stream), g->coop_wgs, e->n_cus);
    HIPCHK(hipGraphLaunch(g->exec_c && g->lm && g->lm->n_pre > 0 ? g->exec_c : g->exec, S(e, stream)));
    if (guard.finish() < 0) return fail(-2, "graph launch: could not chain the cooperative launch behind its predecessor");
  }
  if (g->lm) for (int b = 0; b < g->lm->B; ++b) g->lm->h_off[b] += g->lm->h_active[b];
  if (g->mimi) g->mimi->h_frame += 1;
  return 0;
}

extern "C" void ptts_graph_destroy(ptts_graph *g) {
  if (!g) return;
  if (g->lm && g->exec && g->lm->n_graphs > 0) g->lm->n_graphs -= 1;
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  if (g->exec_c) hipGraphExecDestroy(g->exec_c);
  if (g->graph_c) hipGraphDestroy(g->graph_c);
  if (g->cap_stream) hipStreamDestroy(g->cap_stream);
  delete g;
}

// ------------------------------------------------------------------------------------------------
extern "C" int ptts_set_option(ptts_engine *e, const char *key, int32_t value) {
  if (!e || !key) return fail(-1, "null argument");
  ENGINE_LOCK(e);
  const std::string k(key);
  if (k == "flow_cluster") e->opt_flow_cluster = value != 0;
  else if (k == "lm_cluster") e->opt_lm_cluster = value != 0;
  else if (k == "k_rotate") e->opt_k_rotate = value != 0;
  else if (k == "fuse_res") e->opt_fuse_res = value != 0;
  else if (k == "share_prefix") e->opt_share_prefix = value != 0;
  else if (k == "single_store") e->opt_single_store = value != 0;
  else if (k == "fuse_pcm") e->opt_fuse_pcm = value != 0;
  else if (k == "debug_taps") e->opt_debug_taps = value != 0;
  else if (k == "prefix_cascade") e->opt_cascade = value == 1 ? 423 : value;
  else if (k == "codec_lds_target") {
    if (value < 0 || value > 64 * 1024) return fail(-1, "codec_lds_target must be in [0, 65536]");
    e->opt_codec_lds_target = value;
  }
  else if (k == "flow_max_cus") {
    if (value < 8 || value > e->n_cus) return fail(-1, "flow_max_cus must be in [8, CUs of the device = " + std::to_string(e->n_cus) + "]");
    e->opt_flow_max_cus = value;
  }
  else return fail(-1, "unknown option " + k);
  return 0;
}
extern "C" int ptts_lm_state_error(ptts_lm_state *s, void *stream) {
  if (!s) return fail(-1, "null state");
  int h = 0;
  hipStream_t st = S(s->e, stream);
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipMemcpyAsync(&h, s->ferr, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemsetAsync(s->ferr, 0, sizeof(int), st));  // read AND clear: one transient timeout is reported once
  HIPCHK(hipStreamSynchronize(st));
  return h != 0;
}
extern "C" int ptts_debug_set_error(ptts_lm_state *s, int32_t value, void *stream) {
  if (!s) return fail(-1, "null state");
  hipStream_t st = S(s->e, stream);
  HIPCHK(hipSetDevice(s->e->device));
  HIPCHK(hipMemsetAsync(s->ferr, value ? 1 : 0, 1, st));  // little endian: byte 0 = the word's low byte
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
// A stream whose kernels run on a subset of the CUs (hipExtStreamCreateWithCUMask): CUs [cu_lo, cu_hi) of EVERY XCD.
// Measured on MI355X / ROCm 7.2 (tools/cu_mask_probe.py): mask bit i addresses CU i / 8 of XCD i % 8, a kernel runs
// at the speed of its most-masked XCD (workgroups are dealt round-robin over the XCDs), and the mask applies to graph
// launches on the stream as well.  Lets the codec stream leave a slice of every XCD to the latency-bound FlowLM stream.
extern "C" int ptts_stream_create_masked(ptts_engine *e, int32_t cu_lo, int32_t cu_hi, void **out) {
  if (!e || !out) return fail(-1, "bad argument");
  HIPCHK(hipSetDevice(e->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, e->device));
  const int ncu = prop.multiProcessorCount, nxcd = 8, per = ncu / nxcd, words = (ncu + 31) / 32;
  if (cu_lo < 0 || cu_hi > per || cu_lo >= cu_hi) return fail(-1, "CU range must lie inside [0, CUs per XCD)");
  std::vector<uint32_t> mask(words, 0u);
  for (int i = 0; i < ncu; ++i) {
    const int c = i / nxcd;
    if (c >= cu_lo && c < cu_hi) mask[i >> 5] |= 1u << (i & 31);
  }
  hipStream_t st = nullptr;
  HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
  *out = (void *)st;
  return 0;
}
// Do two streams run concurrently?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4,
// assigned round-robin at creation): two streams that share a queue execute strictly one after the other, and a
// pipeline whose FlowLM and codec streams collide loses its whole overlap (measured: 0.88 -> 1.15 ms per step, in about
// one of four stream pairs).  Two 200 us single-workgroup spin kernels released together: 1 = overlapped, 0 = serialised.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
extern "C" int ptts_streams_overlap(ptts_engine *e, void *stream_a, void *stream_b) {
  if (!e) return fail(-1, "null engine");
  HIPCHK(hipSetDevice(e->device));
  hipStream_t a = S(e, stream_a), b = S(e, stream_b);
  if (a == b) return 0;
  struct Events {  // destroyed on every exit path
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
  } ev;
  for (hipEvent_t &x : ev.e) HIPCHK(hipEventCreate(&x));
  hipEvent_t e0 = ev.e[0], e1 = ev.e[1], e2 = ev.e[2];
  HIPCHK(hipStreamSynchronize(a)); HIPCHK(hipStreamSynchronize(b));
  float worst = 0.f;
  for (int rep = 0; rep < 2; ++rep) {  // first round also warms the kernel up
    HIPCHK(hipEventRecord(e0, a));
    HIPCHK(hipStreamWaitEvent(b, e0, 0));
    spin_kernel<<<1, 64, 0, a>>>(20000);
    spin_kernel<<<1, 64, 0, b>>>(20000);
    HIPCHK(hipEventRecord(e1, a));
    HIPCHK(hipEventRecord(e2, b));
    HIPCHK(hipStreamSynchronize(a)); HIPCHK(hipStreamSynchronize(b));
    float t1 = 0.f, t2 = 0.f;
    HIPCHK(hipEventElapsedTime(&t1, e0, e1));
    HIPCHK(hipEventElapsedTime(&t2, e0, e2));
    worst = std::max(t1, t2);
  }
  return worst < 0.33f ? 1 : 0;  // 0.2 ms each: ~0.21 overlapped, ~0.41 serialised
}
extern "C" int ptts_stream_destroy(void *stream) {
  if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
  return 0;
}
extern "C" int ptts_sync(ptts_engine *e, void *stream) {
  HIPCHK(hipStreamSynchronize(S(e, stream)));
  return 0;
}
extern "C" void *ptts_engine_stream(ptts_engine *e) { return (void *)e->stream; }
// Text-embedding gather of the prefill (reference text.py:74-76, tts_model.py:722-725): d_out[i, :] = d_table[d_tokens[i], :].
// The table stays the caller's tensor (checkpoint "flow_lm.conditioner.embed.weight", f32[n_bins, d_model]).  Asynchronous;
// ids outside [0, n_bins) give zero rows (callers validate ids on the host, where the tokenizer produced them).
extern "C" int ptts_embed_tokens(ptts_engine *e, const float *d_table, int32_t n_bins, const int64_t *d_tokens, int64_t n,
                                 float *d_out, void *stream) {
  if (!e || !d_table || !d_tokens || !d_out || n < 0 || n_bins < 1) return fail(-1, "embed_tokens: bad argument");
  if (n == 0) return 0;
  ENGINE_LOCK(e);
  HIPCHK(hipSetDevice(e->device));
  const int D = e->cfg.d_model;
  embed_gather_kernel<<<cdiv(n * (D / 4), 256), 256, 0, S(e, stream)>>>(d_table, n_bins, D, (const long long *)d_tokens, n, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int ptts_copy_to_host_async(ptts_engine *e, void *h_dst, const void *d_src, int64_t bytes, void *stream) {
  HIPCHK(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, S(e, stream)));
  return 0;
}
extern "C" int ptts_timer_start(ptts_engine *e, void *stream) {
  HIPCHK(hipEventRecord(e->ev0, S(e, stream)));
  return 0;
}
extern "C" int ptts_timer_stop_ms(ptts_engine *e, void *stream, float *ms) {
  HIPCHK(hipEventRecord(e->ev1, S(e, stream)));
  HIPCHK(hipEventSynchronize(e->ev1));
  HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
  return 0;
}
extern "C" int64_t ptts_lm_weight_bytes(ptts_engine *e) { return e->lm_bytes; }
extern "C" int64_t ptts_mimi_weight_bytes(ptts_engine *e) { return e->codec_fp8 ? e->mimi_bytes_f8 : e->codec_bf16 ? e->mimi_bytes_h : e->mimi_bytes; }

extern "C" int64_t ptts_debug_read(ptts_engine *e, void *state, int32_t is_mimi, const char *name, float *d_out,
                                   int64_t capacity, int32_t *rows, int32_t *cols, void *stream) {
  hipStream_t st = S(e, stream);
  const ptts_config &c = e->cfg;
  const float *src = nullptr;
  int M = 0, K = 0, F = 0;
  std::string n(name);
  if (!is_mimi) {
    ptts_lm_state *s = (ptts_lm_state *)state;
    if (n == "x") { src = s->dec.x; M = s->B; K = c.d_model; }
    else if (n == "ce") { src = s->ce; M = s->B; K = c.flow_dim; }
    else if (n == "fx") { src = s->fx; M = s->B; K = c.flow_dim; }
    else if (n == "prefill_x") { src = s->pre.x; M = s->pre.MT * 16; K = c.d_model; }
    else return fail(-1, "unknown buffer " + n);
    F = K / 16;
  } else {
    ptts_mimi_state *s = (ptts_mimi_state *)state;
    const int par = (s->h_frame - 1) & 1;  // parity of the frame decoded last
    const int B = s->B;
    if (n == "upsample") { src = s->u0; M = B * 16; K = c.m_dim; }
    else if (n == "dec_tr") { src = s->tr_out + par * s->tr_stride; M = B * 16; K = c.m_dim; }
    else if (n == "seanet0") { src = s->a0 + par * s->a0_stride; M = B * 16; K = 8 * c.n_filters; }
    else if (n == "tr_attn") { src = s->ao; M = B * 16; K = c.m_dim; }       // last layer's attention output
    else if (n == "tr_resid") { src = s->u; M = B * 16; K = c.m_dim; }       // last layer's stream after attention
    else if (n == "tr_ff") { src = s->ff; M = B * 16; K = c.m_ff; }          // last layer's GELU(linear1)
    else if (n == "seanet11") {
      M = B; K = s->rows[3];
      if ((int64_t)M * K > capacity) return fail(-1, "capacity");
      if (hipMemcpyAsync(d_out, s->pcm_dbg, (size_t)M * K * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(-2, "memcpy");
      *rows = M; *cols = K;
      return (int64_t)M * K;
    } else {
      int idx = atoi(n.c_str() + 6);
      if (n.compare(0, 6, "seanet") != 0 || idx < 2 || idx > 9) return fail(-1, "unknown buffer " + n);
      int stage = (idx - 2) / 3;
      bool is_res = (idx - 2) % 3 == 1;
      if ((idx - 2) % 3 == 2) return fail(-1, "unknown buffer " + n);
      int mult = 8 >> stage;
      K = mult * c.n_filters / 2;
      M = B * s->rows[stage + 1];
      const ptts_engine *en = s->e;
      if (is_res && stage == 2 && en->opt_fuse_pcm && !en->opt_debug_taps && !en->codec_split && en->res_a[2].NT == 2 && s->rows[3] % 64 == 0)
        return fail(-1, "debug_read: " + n + " stays on chip with \"fuse_pcm\"; set the engine option \"debug_taps\" before decoding");
      const bool single = en->opt_single_store && !en->codec_split && !en->res_a[stage].wq && !en->res_a[stage].wb16;
      src = is_res ? s->sbuf[stage] + par * s->s_stride[stage] : single ? s->cbuf[stage] + par * s->c_stride[stage] : s->craw[stage];
    }
    F = K / 16;
  }
  if ((int64_t)M * K > capacity) return fail(-1, "debug_read: capacity too small");
  long n4 = (long)M * (K / 4);
  from_fm_kernel<<<cdiv(n4, 256), 256, 0, st>>>(src, d_out, M, K, F, 0);
  *rows = M;
  *cols = K;
  return (int64_t)M * K;
}
