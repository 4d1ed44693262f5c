/*
 * ptts.h -- C ABI of the MI355X (gfx950) Pocket-TTS decode hot path.
 *
 * The reference (kyutai-labs/pocket-tts) is pure Python and has no FFI; the seam this
 * library replaces is the set of Python call sites listed per entry point below
 * (paths relative to the reference checkout).  All pointers named `d_*` are DEVICE
 * pointers to fp32 data (e.g. torch `tensor.data_ptr()` on a ROCm device); `h_*` are host
 * pointers.  `stream` is a `hipStream_t` passed as `void*` (NULL = the engine's own
 * stream).  Every function returns 0 on success or a negative error code;
 * `ptts_last_error()` returns the message (per calling thread).  No exceptions cross the ABI.
 *
 * Threading contract.  The library keeps NO process-global mutable state: profiler, tile table, LSD tables and the
 * allocation stream live in the engine or in the calling thread.  Engines are independent: any number of engines
 * (one per GPU, or several on one GPU) may be driven from different threads at the same time.  Entry points that
 * enqueue work on an engine serialise on that engine's mutex, so two threads MAY share one engine (e.g. a batching
 * scheduler thread and a request thread); device-side ordering between their calls is the caller's business
 * (streams / events).  A state or graph handle is not re-entrant: one thread per ptts_lm_state / ptts_mimi_state /
 * ptts_graph at a time.  Decode steps contain cooperative kernels (the single-launch flow MLP: all its workgroups must
 * be resident together).  Such a launch never exceeds the device's CU count or the `flow_max_cus` option, a launch on a
 * CU-masked stream with fewer CUs than its grid is refused (-1), and the library admits only as many of these steps at a
 * time per DEVICE as their grids fit the chip together (two for the default 128-workgroup grid): step k waits ON THE GPU
 * for step k - 2 through a small per-device ring of events (no host wait), so steps of different states queued on
 * different streams are safe and overlap pairwise.  The one process-wide object is that per-device ring (with its mutex).
 */
#ifndef PTTS_H_
#define PTTS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTTS_ABI_VERSION 1

typedef struct ptts_engine ptts_engine;         /* weights + kernels for one GPU         */
typedef struct ptts_lm_state ptts_lm_state;     /* FlowLM KV caches of B sequences       */
typedef struct ptts_mimi_state ptts_mimi_state; /* Mimi streaming state of B sequences   */
typedef struct ptts_graph ptts_graph;           /* a captured hipGraph of one step       */

/* Model dimensions: pocket_tts/config/english.yaml:7-61 (schema utils/config.py:15-118). */
typedef struct ptts_config {
  int32_t d_model, num_heads, num_layers, ff_dim, ldim; /* flow_lm.transformer, quantizer.dimension */
  int32_t flow_dim, flow_depth;                         /* flow_lm.flow                              */
  float max_period;                                     /* flow_lm.transformer.max_period            */
  int32_t m_dim, m_heads, m_layers, m_ff, m_context;    /* mimi.transformer (d_model == seanet dim)  */
  float m_max_period;
  int32_t n_filters, ratios[3], kernel_size, res_kernel_size, last_kernel_size, compress;
  int32_t upsample_stride;                              /* encoder_frame_rate / frame_rate = 16      */
} ptts_config;

/* One checkpoint tensor, name as in TTSModel.state_dict() (tts_model.py:206-210). */
typedef struct ptts_tensor {
  const char *name;
  const float *d_data;
  int64_t numel;
} ptts_tensor;

int ptts_abi_version(void);
const char *ptts_last_error(void);

/* Builds the engine: packs the checkpoint tensors it needs into MFMA-fragment order on
 * `device`.  Replaces TTSModel._from_pydantic_config_with_weights (tts_model.py:129-230)
 * for the decode-side modules.  The caller may free the source tensors afterwards. */
int ptts_create(const ptts_config *cfg, const ptts_tensor *tensors, int32_t n_tensors, int32_t device,
                ptts_engine **out);
/* Same, with int8 weights for layer groups of the FlowLM transformer: replaces
 * quantization.apply_dynamic_int8(flow_lm, groups) (quantization.py:60-128; load_model(quantize=True) uses
 * {"attention", "ffn"}, tts_model.py:312-315).  The reference quantises activations dynamically as well (torch.ao /
 * torchao CPU kernels); here the weights are int8 per output channel (symmetric) and activations and accumulation
 * stay fp32, so the result is closer to the fp32 model than the reference's int8 path.  flags = 0 == ptts_create. */
#define PTTS_QUANT_ATTENTION 1 /* self_attn.in_proj, self_attn.out_proj */
#define PTTS_QUANT_FFN 2       /* linear1, linear2 */
/* Reduced-precision CODEC (BASELINE.json config #5, second half; no reference counterpart, parity unpinned): the Mimi
 * decoder-transformer GEMMs and the SEANet decoder convolutions run with bf16 weights and bf16 activations, fp32
 * accumulation (v_mfma_f32_16x16x32_bf16) and fp32 epilogue math; attention, the KV ring and the FlowLM stay fp32, so
 * EOS decisions / frame counts are those of the fp32 model.  Quality: tests/test_gpu_bf16.py (SNR vs the fp32 path). */
#define PTTS_CODEC_BF16 4
/* fp8 CODEC CONVOLUTIONS (BASELINE.json config #5: "fp8 MFMA codec convs"; no reference counterpart - the reference never
 * quantises Mimi, docs/quantization.md:67-76 - parity unpinned, judged by SNR and frame-count equality): the SEANet decoder
 * convolutions (seanet.py:141-180, conv.py:93-163) run on v_mfma_f32_16x16x32_fp8_fp8 with OCP e4m3 weights (one fp32
 * scale per output channel) and e4m3 activations (one fp32 scale per tensor, fixed at load time from a calibration
 * frame), fp32 accumulation and epilogue math; the Mimi decoder transformer runs as under PTTS_CODEC_BF16 (bf16), the
 * FlowLM stays fp32.  Exclusive with PTTS_CODEC_BF16.  Quality: tests/test_gpu_fp8.py. */
#define PTTS_CODEC_FP8 8
/* bf16 WEIGHTS for the FlowLM transformer's Linear layers (SURVEY 8(f).4 "bf16 / int8 per-channel LM weights"; same
 * hook as the int8 groups: quantization.py:91-128): weights rounded to bf16 once at load, activations and accumulation
 * fp32 (the bf16 image is widened in registers and fed to the fp32 MFMA), half the weight bytes of a decode step.
 * Exclusive with the int8 groups.  Quality: tests/test_gpu_quant.py. */
#define PTTS_LM_BF16 16
/* ERROR-COMPENSATED bf16 for the codec's GEMMs (an experiment reported beside the fp32 headline): every Linear / conv of the
 * Mimi decoder as hi*hi + hi*lo + lo*hi of bf16 halves (hi = bf16(v), lo = bf16(v - hi)) on the bf16 MFMA with fp32
 * accumulation; activations, buffers, attention and epilogues stay fp32.  ~2^-16 relative error per product; passes the fp32
 * codec's parity tests at their tolerance (tests/test_gpu_split.py).  Exclusive with the bf16 / fp8 codec. */
#define PTTS_CODEC_SPLIT 32
int ptts_create_ex(const ptts_config *cfg, const ptts_tensor *tensors, int32_t n_tensors, int32_t device,
                   int32_t quant_flags, ptts_engine **out);
/* Packed-engine files ("offline packer"): ptts_engine_save writes everything ptts_create[_ex] built on
 * the device (MFMA-fragment-ordered weights, int8 / bf16 images, LayerNorm-fold vectors) to one file;
 * ptts_create_from_file rebuilds the engine from it without the checkpoint and without packing or quantising again
 * (the reference re-quantises at every load: quantization.py:60-88, tts_model.py:312-313).  The file is tied to this
 * library's ABI version and layout; a mismatch is reported (-3), never loaded. */
int ptts_engine_save(ptts_engine *e, const char *path);
int ptts_create_from_file(const char *path, int32_t device, ptts_engine **out);
void ptts_destroy(ptts_engine *e);

/* ---- FlowLM state: init_states(flow_lm, B, T) (stateful_module.py:7-16, transformer.py:46-57) */
int ptts_lm_state_create(ptts_engine *e, int32_t batch, int32_t t_cap, ptts_lm_state **out);
void ptts_lm_state_destroy(ptts_lm_state *s);
/* zero offsets (fresh init_states) */
int ptts_lm_state_reset(ptts_lm_state *s, void *stream);
/* Import / export one layer in the reference layout cache f32[2, src_batch, t, H, 64] (device) with
 * `t` valid positions (transformer.py:32-36; voice files tts_model.py:1047-1072).  On import,
 * src_batch == 1 broadcasts to every row of the state, and every row's offset is set to t. */
int ptts_lm_state_import(ptts_lm_state *s, int32_t layer, const float *d_cache, int32_t src_batch,
                         int32_t t, void *stream);
int ptts_lm_state_export(ptts_lm_state *s, int32_t layer, float *d_cache, int32_t t, void *stream);
/* dst <- src (replaces copy.deepcopy(model_state), tts_model.py:637-638); src batch 1 broadcasts.
 * The clone starts a new generation: its pending input latent is BOS (tts_model.py:748-753).
 * With "share_prefix" (default) a clone of a one-sequence state borrows its leading keys instead of copying them: see
 * ptts_set_option.  These calls, destroy, reset, import and set_row_active serialise on the engine's mutex. */
int ptts_lm_state_copy(ptts_lm_state *dst, const ptts_lm_state *src, void *stream);
/* row `row` of dst <- the single sequence of src (batch 1): assembles a batch from utterances prefilled one by
 * one with different prompt lengths; every kernel reads per-row offsets, so rows need not be in sync (the
 * reference requires equal offsets across the batch: transformer.py:12-13). */
int ptts_lm_state_copy_row(ptts_lm_state *dst, int32_t row, const ptts_lm_state *src, void *stream);
/* the same from row `src_row` of a batch state: a group of utterances with the same voice and text length is prefilled
 * as ONE batch (one GEMM pass instead of one per utterance) and its rows are then dealt to their slots */
int ptts_lm_state_copy_row_from(ptts_lm_state *dst, int32_t row, const ptts_lm_state *src, int32_t src_row, void *stream);
/* offsets of all rows to host (transformer.py:14 `.item()`; synchronises the stream) */
int ptts_lm_state_offsets(ptts_lm_state *s, int32_t *h_offsets, void *stream);

/* Text or voice conditioning d_emb f32[B, t, d_model] run through every layer; only the KV cache
 * is kept.  Replaces _run_flow_lm_and_increment_step(text_tokens=.. | audio_conditioning=..)
 * (tts_model.py:317-346, call sites :723, :899). */
int ptts_lm_prefill(ptts_engine *e, ptts_lm_state *s, const float *d_emb, int32_t t, void *stream);
/* The text-embedding gather in front of a text prefill (LUTConditioner._get_condition, conditioners/text.py:74-76):
 * d_out f32[n, d_model] <- d_table[d_tokens[i]] for int64 ids; d_table = the checkpoint tensor
 * "flow_lm.conditioner.embed.weight" f32[n_bins, d_model], which stays the caller's.  Asynchronous on `stream`; an id outside
 * [0, n_bins) gives a zero row (validate ids on the host, where the tokenizer made them). */
int ptts_embed_tokens(ptts_engine *e, const float *d_table, int32_t n_bins, const int64_t *d_tokens, int64_t n,
                      float *d_out, void *stream);

/* One autoregressive step = _run_flow_lm_and_increment_step(backbone_input_latents=..)
 * (tts_model.py:758-760 -> flow_lm.py:96-139).
 *   d_latent_in  f32[B, ldim]  rows of NaN mean BOS; NULL = previous step's output (or BOS after reset)
 *   d_noise      f32[B, ldim]  starting point of the LSD flow (flow_lm.py:131-137); NULL = zeros (temp 0)
 *   d_latent_out f32[B, ldim], d_eos_logit f32[B], d_is_eos u8[B] (logit > eos_threshold); any may be NULL */
int ptts_lm_decode_step(ptts_engine *e, ptts_lm_state *s, const float *d_latent_in, const float *d_noise,
                        int32_t lsd_steps, float eos_threshold, float *d_latent_out, float *d_eos_logit,
                        uint8_t *d_is_eos, void *stream);
/* Perf-run noise source: when d_noise is NULL and temp > 0 the step draws N(0, temp) itself from a
 * counter-based device generator (the reference draws from torch's global CPU generator,
 * flow_lm.py:131-135, which cannot be reproduced on device; parity runs pass d_noise or temp 0). */
int ptts_lm_set_noise(ptts_lm_state *s, float temp, uint64_t seed);
/* device pointer of the state's own copy of the latest latent f32[B, ldim] */
const float *ptts_lm_latent_ptr(ptts_lm_state *s);

/* ---- Mimi streaming decode: init_states(mimi, B, ..) + _decode_audio_worker body
 *      (tts_model.py:444-455 -> mimi.py:89-94) */
int ptts_mimi_state_create(ptts_engine *e, int32_t batch, ptts_mimi_state **out);
void ptts_mimi_state_destroy(ptts_mimi_state *s);
int ptts_mimi_state_reset(ptts_mimi_state *s, void *stream);
/* ---- continuous batching (the reference serves one request at a time, main.py:80-181) ----
 * A batch state is a set of slots.  An utterance JOINS slot `row` with ptts_lm_state_copy_row (its prefilled
 * batch-1 state) + ptts_mimi_state_reset_row (zero conv carries == init_states for that sequence,
 * stateful_module.py:7-16) and LEAVES with ptts_lm_state_set_row_active(row, 0): a parked row keeps flowing
 * through the batched kernels but stays at position 0.  Captured graphs stay valid across joins / leaves. */
int ptts_mimi_state_reset_row(ptts_mimi_state *s, int32_t row, void *stream);
int ptts_lm_state_set_row_active(ptts_lm_state *s, int32_t row, int32_t active, void *stream);
/* 16-bit PCM straight from the codec's last kernel: (clamp(x, -1, 1) * 32767) truncated, the conversion of
 * StreamingWAVWriter.write_pcm_data (data/audio.py:79).  i16[B, frame_samples], device or pinned host. */
int ptts_mimi_set_pcm_i16(ptts_mimi_state *s, int16_t *d_pcm_i16);
/* d_latent f32[B, ldim] (normalised FlowLM output) -> d_pcm f32[B, frame_samples]; includes the
 * emb_std/emb_mean de-normalisation, the quantizer 1x1 conv and increment_steps(mimi, 16). */
int ptts_mimi_decode(ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm, void *stream);

/* ---- Voice-prompt encode path (one-off per voice): MimiModel.encode_to_latent + F.linear(speaker_proj_weight)
 *      (mimi.py:96-119, tts_model.py:379-388).  d_audio f32[n_samples] mono at the model rate; the signal is
 *      zero-padded to a whole number of frames.  Outputs (either may be NULL): d_latent_out f32[frames, ldim],
 *      d_cond_out f32[frames, d_model] (feed it to ptts_lm_prefill after bos_before_voice).  Synchronises. */
int ptts_encode_voice(ptts_engine *e, const float *d_audio, int64_t n_samples, float *d_latent_out,
                      float *d_cond_out, int32_t *h_frames, void *stream);

/* ---- hipGraph capture of one step (north star: "each decode step hipGraph-captured").
 * The captured step uses the same argument pointers on every launch. */
int ptts_graph_capture_lm_step(ptts_engine *e, ptts_lm_state *s, const float *d_noise, int32_t lsd_steps,
                               float eos_threshold, float *d_latent_out, float *d_eos_logit,
                               uint8_t *d_is_eos, ptts_graph **out);
int ptts_graph_capture_mimi(ptts_engine *e, ptts_mimi_state *s, const float *d_latent, float *d_pcm,
                            ptts_graph **out);
/* One graph with two parallel branches: the FlowLM step (as ptts_graph_capture_lm_step) and the Mimi decode of
 * the PREVIOUS frame (reads d_mimi_latent_in, which must differ from d_latent_out).  Replaying such graphs
 * back to back on one stream overlaps step t+1 with frame t without cross-stream events (the reference
 * pipelines the same two stages with two threads: tts_model.py:651-658). */
int ptts_graph_capture_pipelined(ptts_engine *e, ptts_lm_state *s, ptts_mimi_state *m, const float *d_noise,
                                 int32_t lsd_steps, float eos_threshold, float *d_latent_out, float *d_eos_logit,
                                 uint8_t *d_is_eos, const float *d_mimi_latent_in, float *d_pcm, ptts_graph **out);
int ptts_graph_launch(ptts_graph *g, void *stream);
void ptts_graph_destroy(ptts_graph *g);

/* ---- tile autotuning (no reference counterpart: torch picks its CPU kernels at dispatch time).
 * Runs one FlowLM step + one codec frame of `batch` rows on scratch states and, for every GEMM shape on the
 * path, times each valid tile configuration (cache flushed before every timed launch) and remembers the
 * fastest for this engine.  Call it before capturing graphs for that batch; shapes never tuned use the static
 * heuristic.  Results are numerically equivalent up to fp32 summation order.  Synchronises.
 * ptts_tune_log: one text line per tuned shape (valid until the next tune/clear). */
int ptts_tune(ptts_engine *e, int32_t batch, void *stream);
/* the same with the FlowLM step tuned on `lm_stream` and the codec frame on `codec_stream`: with CU-masked streams
 * (ptts_stream_create_masked) each stage gets the tiles that are fastest on its own share of the chip */
int ptts_tune_streams(ptts_engine *e, int32_t batch, void *lm_stream, void *codec_stream);
/* the same for the prefill GEMM shapes of `batch` sequences x `t` positions (text prefill of a chunk: first-chunk path) */
int ptts_tune_prefill(ptts_engine *e, int32_t batch, int32_t t, void *stream);
const char *ptts_tune_log(ptts_engine *e);
void ptts_tune_clear(ptts_engine *e);
/* The tuned table as text (one line per GEMM shape) so that a deployment tunes once: export after ptts_tune,
 * import (returns the number of entries accepted) before capturing graphs in a later process. */
int64_t ptts_tune_export(ptts_engine *e, char *h_out, int64_t capacity);
/* version of the table format + configuration list: a cache file written by another version must not be imported */
int ptts_tune_version(void);
int ptts_tune_import(ptts_engine *e, const char *text);

/* ---- utilities */
/* Engine options (experiments / A-B tests; defaults come from the environment variable in brackets):
 *   "flow_cluster"  [PTTS_FLOW_CLUSTER, 1]  1 = the flow MLP of a decode step runs as ONE cooperative launch
 *                                           (ptts_flow.h), 0 = one GEMM launch per layer
 *   "lm_cluster"    [PTTS_LM_CLUSTER, 0]    1 = all transformer layers of a decode step run as ONE cooperative launch
 *                                           (ptts_lm.h; fp32 weights only; correct, but measured SLOWER than the default:
 *                                           DESIGN.md section 3), 0 = five launches per layer
 *   "k_rotate"      [PTTS_K_ROTATE, 0]      1 = K-split GEMM workgroups start their K loop at a column-block dependent
 *                                           chunk (spreads the re-reads of the shared activation rows over L2 channels)
 *   "fuse_res"      [PTTS_FUSE_RES, 1]      1 = a SEANet residual block (k3 conv, ELU, 1x1 conv, skip) of decoder stages 2 and 3
 *                                           is ONE launch, the hidden activation staying in LDS; 0 = two launches
 *   "codec_lds_target" [PTTS_CODEC_LDS_TARGET, 57344]  the codec's GEMM launches pad their LDS request to this many bytes
 *                                           per workgroup (0 = off): fewer codec workgroups per CU, so the FlowLM stream's
 *                                           short dependent kernels find free wave slots and registers (+5 % pipelined throughput at batch 64)
 *   "flow_max_cus"  [PTTS_FLOW_MAX_CUS, 128] resident workgroups of the cooperative flow launch (8..CUs of the device): at most the number
 *                                           of CUs its stream may use (a CU-masked FlowLM stream needs it lowered; fewer is
 *                                           slower in the shared pipeline too: 128 -> 0.852, 64 -> 0.904, 32 -> 1.010 ms per step)
 *   "single_store"  [PTTS_SINGLE_STORE, 1]  1 = a SEANet transposed conv stores its RAW output once (it is the residual block's skip
 *                                           input) and the k3 conv that follows applies ELU to its operand fragments as it reads
 *                                           them; 0 = the producer stores raw + ELU'd copies (rounds 1-2).  fp32 codec only;
 *                                           -107 MB of HBM traffic per 64-sequence frame, +1.4 % pipelined throughput
 *   "fuse_pcm"      [PTTS_FUSE_PCM, 1]      1 = SEANet's last conv (n_filters -> 1 sample) runs in the epilogue of the last stage's fused
 *                                           residual block (per 64-row tile: partial sums + a carry for the next tile's first two
 *                                           samples; a small kernel adds carries and bias and writes the PCM); 0 = a separate
 *                                           conv over the stored block output.  fp32 codec, en100m-shaped last stage only
 *   "debug_taps"    [-, 0]                  1 = buffers that fused kernels keep on chip are also stored, so that ptts_debug_read can
 *                                           return every stage (parity tests); reading such a buffer without it fails (-1)
 *   "share_prefix"  [PTTS_SHARE_PREFIX, 1]  1 = ptts_lm_state_copy / _copy_row(_from) from a ONE-sequence state (a voice state) do
 *                                           not copy its first T & ~15 keys / values: the clone's rows BORROW them (the
 *                                           attention kernels read those key tiles from the owner's cache), so the
 *                                           utterances of one voice fetch them through L2 once instead of once per row, and a
 *                                           clone copies < 16 positions.  Bitwise the same results as full copies.  While
 *                                           lent, the owner refuses ptts_lm_state_reset / _import / being a copy
 *                                           destination (-1); it may be prefilled further (appends only) and destroyed (its
 *                                           memory is released when the last borrower is destroyed, re-cloned or reset).
 *                                           Export materialises the borrowed positions.
 *   "prefix_cascade" [PTTS_CASCADE, 1]      decode steps of >= 16 sequences: 1 = the scores against a prefix shared by 4
 *                                           neighbouring rows are MFMA tiles computed once for the 4 (attn_cascade_kernel),
 *                                           their private keys per row, merged in LDS; 0 = every row on its own.  Equal to
 *                                           fp32 summation order, not bitwise; a row's bits then depend on whether its 4-row
 *                                           group shares a prefix (a row next to other voices takes the per-row path)
 * Applies to steps enqueued / graphs captured after the call.  Returns -1 for an unknown key. */
int ptts_set_option(ptts_engine *e, const char *key, int32_t value);
/* 1 after a cooperative kernel of this state gave up waiting for a peer workgroup since the last call (the outputs of
 * the steps in between are then invalid); READS AND CLEARS the word, synchronises the stream.  Never 1 unless the GPU was
 * oversubscribed beyond the contract above. */
int ptts_lm_state_error(ptts_lm_state *s, void *stream);
/* Test hook: sets (value != 0) or clears the word ptts_lm_state_error reports. */
int ptts_debug_set_error(ptts_lm_state *s, int32_t value, void *stream);
/* A HIP stream restricted to CUs [cu_lo, cu_hi) of every XCD (hipExtStreamCreateWithCUMask; 32 CUs per XCD on
 * MI355X): work queued on it - eager launches and graph launches alike - leaves the other CUs to the other streams.
 * Destroy with ptts_stream_destroy after the work queued on it has finished. */
int ptts_stream_create_masked(ptts_engine *e, int32_t cu_lo, int32_t cu_hi, void **out_stream);
int ptts_stream_destroy(void *stream);
/* 1 if work queued on the two streams runs concurrently, 0 if the runtime put them on the same hardware queue (HIP
 * multiplexes streams onto GPU_MAX_HW_QUEUES queues, default 4, round-robin at creation; such a pair executes strictly
 * in turn).  Callers that pipeline the FlowLM step and the codec frame on two streams check the pair once and pick
 * another stream on 0.  Synchronises both streams; costs ~0.5 ms. */
int ptts_streams_overlap(ptts_engine *e, void *stream_a, void *stream_b);
int ptts_sync(ptts_engine *e, void *stream);
void *ptts_engine_stream(ptts_engine *e);
/* asynchronous device -> pinned-host copy on `stream` (PCM chunks, EOS flags) */
int ptts_copy_to_host_async(ptts_engine *e, void *h_dst, const void *d_src, int64_t bytes, void *stream);
/* HIP-event timing on `stream` (bench.py measures on the stream the kernels run on) */
int ptts_timer_start(ptts_engine *e, void *stream);
int ptts_timer_stop_ms(ptts_engine *e, void *stream, float *h_ms);
/* Test hook: copies an internal activation buffer, converted to row-major f32[rows, cols], to
 * d_out (capacity in floats).  Names: see DESIGN.md.  Returns rows*cols or <0. */
int64_t ptts_debug_read(ptts_engine *e, void *state, int32_t is_mimi, const char *name, float *d_out,
                        int64_t capacity, int32_t *rows, int32_t *cols, void *stream);
/* Per-launch profiler (HIP events around every kernel launch on its own stream, tagged with call site,
 * kernel and algorithmic bytes / flops).  stop() writes text lines "site kernel count total_ms bytes flops"
 * to h_out and returns the length.  Never active inside a captured graph. */
int ptts_profile_start(ptts_engine *e);
int64_t ptts_profile_stop(ptts_engine *e, char *h_out, int64_t capacity);
/* Packed weight bytes streamed by one LM decode step / one Mimi frame (roofline accounting) */
int64_t ptts_lm_weight_bytes(ptts_engine *e);
int64_t ptts_mimi_weight_bytes(ptts_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* PTTS_H_ */
