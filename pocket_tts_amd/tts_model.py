"""Drop-in `TTSModel` surface over the MI355X engine.

Keeps the reference's public API (`pocket_tts/__init__.py:6-19`, `tts_model.py:232-242,477-552,788-790,
1047-1052`): `TTSModel.load_model`, `.device`, `.sample_rate`, `get_state_for_audio_prompt`,
`generate_audio`, `generate_audio_stream`, and `export_model_state`.  Voice states keep the reference
format (dict module name -> {"cache": f32[2,1,T,H,64], "offset": i64[1]}, safetensors keys
"<module>/<cache|offset>").  The two hot loops of the reference (`_autoregressive_generation`
tts_model.py:744-779 and `_decode_audio_worker` :433-474) are replaced by hipGraph launches of the
C-ABI engine; no worker threads are needed because the GPU queue provides the pipelining.
"""

from __future__ import annotations

import logging
import math
import threading
import time
from pathlib import Path

import numpy as np
import safetensors
import safetensors.torch
import torch

from .config import CONFIGS_DIR, Config, load_config
from .engine import Engine
from .text import estimate_max_gen_len, prepare_text_prompt, split_into_best_sentences
from .weights import generate_state_dict

logger = logging.getLogger(__name__)

# Like the reference at import (tts_model.py:49).  Here it matters for another reason: the host loops of the batched paths
# copy small PCM frames with CPU tensor ops; with ATen's intra-op pool sized to the machine (256 hardware threads on an
# MI355X host, of which a container may own 16) every such op wakes an oversubscribed OpenMP team whose spinning workers
# then starve the thread that feeds the GPU (measured: 7 ms per 491 KB frame copy, 5 ms per event wait; 10x the step time).
torch.set_num_threads(1)

DEFAULT_LANGUAGE = "english"
DEFAULT_TEMPERATURE = 0.7
DEFAULT_LSD_DECODE_STEPS = 1
DEFAULT_NOISE_CLAMP = None
DEFAULT_EOS_THRESHOLD = -4.0
MAX_TOKEN_PER_CHUNK = 50

PREDEFINED_VOICES = (
    "cosette marius javert alba jean anna vera fantine charles paul eponine azelma george mary jane michael "
    "eve bill_boerst peter_yearsley stuart_bell caro_davy giovanni lola juergen rafael estelle"
).split()


class SentencePieceTokenizer:
    """Host-side tokenizer (reference conditioners/text.py:13-35); stays on the CPU."""

    def __init__(self, n_bins: int, path: str):
        import sentencepiece

        if str(path).startswith(("hf://", "http://", "https://")):
            raise FileNotFoundError(
                f"tokenizer_path {path} needs a download; this build runs offline: point "
                "flow_lm.lookup_table.tokenizer_path at a local sentencepiece model"
            )
        self.sp = sentencepiece.SentencePieceProcessor(str(path))
        if n_bins != self.sp.vocab_size():
            raise ValueError(f"sentencepiece tokenizer has vocab size={self.sp.vocab_size()} but nbins={n_bins}")

    def encode(self, text: str) -> list:
        return self.sp.encode(text, out_type=int)


def _load_weights(cfg: Config, seed: int = 0) -> dict:
    path = cfg.weights_path
    if path is None:
        logger.warning("No weights_path specified for FlowLM or TTSModel, model is uninitialized! "
                       "(deterministic synthetic weights, seed %d)", seed)
        return generate_state_dict(cfg, seed)
    for p in (path, cfg.weights_path_without_voice_cloning):
        if p and not str(p).startswith(("hf://", "http://", "https://")) and Path(p).exists():
            return safetensors.torch.load_file(str(p))
    raise FileNotFoundError(
        f"weights_path {path} needs a download; this build runs offline: set weights_path to a local "
        ".safetensors file (same keys as the reference checkpoint) or to null for synthetic weights"
    )


class TTSModel:
    def __init__(self, engine: Engine, config: Config, tokenizer, temp, lsd_decode_steps, noise_clamp,
                 eos_threshold, origin: Path | None = None):
        self.engine = engine
        self.config = config
        self.tokenizer = tokenizer
        self.temp = temp
        self.lsd_decode_steps = lsd_decode_steps
        self.noise_clamp = noise_clamp
        self.eos_threshold = eos_threshold
        self.origin = origin
        self.has_voice_cloning = engine.has_voice_encoder
        self._ctx_cache: dict = {}
        self._voice_cache: dict = {}  # id(model_state) -> (signature, LMState, tensors): device-resident voice states
        self._voice_lock = threading.Lock()  # the cache is shared with ContinuousBatcher's scheduler thread
        self.pad_with_spaces_for_short_inputs = config.pad_with_spaces_for_short_inputs
        self.model_recommended_frames_after_eos = config.model_recommended_frames_after_eos
        self.remove_semicolons = config.remove_semicolons

    # ---- reference properties (tts_model.py:92-98)
    @property
    def device(self) -> torch.device:
        return self.engine.device

    @property
    def sample_rate(self) -> int:
        return self.config.mimi.sample_rate

    @classmethod
    def load_model(cls, language: str | None = None, config: str | Path | None = None,
                   temp: float | int = DEFAULT_TEMPERATURE, lsd_decode_steps: int = DEFAULT_LSD_DECODE_STEPS,
                   noise_clamp: float | int | None = DEFAULT_NOISE_CLAMP, eos_threshold: float = DEFAULT_EOS_THRESHOLD,
                   quantize: bool = False, device: str = "cuda:0", tokenizer=None, codec_bf16: bool = False,
                   codec_fp8: bool = False, lm_bf16: bool = False):
        """Same arguments and errors as the reference (tts_model.py:232-315) plus `device`, `tokenizer` and this build's
        reduced-precision formats (no reference counterpart): `codec_bf16` (bf16 Mimi decoder, fp32 accumulation),
        `codec_fp8` (SEANet convolutions on the fp8 MFMA, transformer bf16), `lm_bf16` (bf16 weights / operands for the
        FlowLM Linear layers; exclusive with `quantize`)."""
        if config is not None and language is not None:
            raise ValueError("Cannot specify both config and language, please choose one or the other.")
        if config is None and language is None:
            language = DEFAULT_LANGUAGE
        if language is not None:
            if language == "french":
                raise ValueError("For technical reasons, only a larger 24-layer model is available for French. "
                                 "Please use the 'french_24l' language instead.")
            config = CONFIGS_DIR / f"{language}.yaml"
        config = Path(config)
        if config.suffix not in (".yaml", ".yml"):
            raise ValueError("Config should be a path to a YAML file ending with .yaml")
        cfg = load_config(config)
        weights = _load_weights(cfg)
        if tokenizer is None:
            tp = str(cfg.flow_lm.lookup_table.tokenizer_path)
            if not tp.startswith(("hf://", "http://", "https://")) and not Path(tp).exists() and (config.parent / tp).exists():
                tp = str(config.parent / tp)  # relative to the YAML file
            tokenizer = SentencePieceTokenizer(cfg.flow_lm.lookup_table.n_bins, tp)
        # quantize=True: int8 weights for the reference's RECOMMENDED_CONFIG groups (quantization.py:21,
        # tts_model.py:312-315); weight-only and per output channel here (see include/ptts.h)
        groups = (({"attention", "ffn"} if quantize else set()) | ({"codec_bf16"} if codec_bf16 else set())
                  | ({"codec_fp8"} if codec_fp8 else set()) | ({"lm_bf16"} if lm_bf16 else set()))
        engine = Engine(cfg, weights, device, quantize_groups=groups or None)
        return cls(engine, cfg, tokenizer, temp, lsd_decode_steps, noise_clamp, eos_threshold, origin=config)

    # ---- voice state ------------------------------------------------------------------------
    def get_state_for_audio_prompt(self, audio_conditioning, truncate: bool = False) -> dict:
        """Voice state from a `.safetensors` file (tts_model.py:846-851,1055-1072), a WAV file or an audio
        tensor [1, samples] at the model sample rate (tts_model.py:874-899): the waveform goes through the
        Mimi encoder + speaker projection on the GPU (`Engine.encode_voice`) and is prefilled."""
        if isinstance(audio_conditioning, (str, Path)) and str(audio_conditioning).endswith(".safetensors"):
            if str(audio_conditioning).startswith(("hf://", "http://", "https://")):
                raise FileNotFoundError(f"{audio_conditioning} needs a download; this build runs offline")
            return _import_model_state(audio_conditioning, self.device)
        if isinstance(audio_conditioning, str) and audio_conditioning in PREDEFINED_VOICES:
            if self.origin is None or not self.origin.is_relative_to(CONFIGS_DIR):
                raise ValueError("Cannot use predefined voices when the model is not loaded from a config "
                                 f"associated with a language.Here the origin is {self.origin}")
            raise FileNotFoundError(f"predefined voice '{audio_conditioning}' needs a download; this build runs "
                                    "offline: pass a local .safetensors voice state instead")
        if not self.has_voice_cloning:
            raise ValueError("We could not load the weights for the model with voice cloning, but you're trying to "
                             "use voice cloning: the checkpoint has no Mimi encoder tensors.")
        if isinstance(audio_conditioning, (str, Path)):
            audio, sr = _audio_read(audio_conditioning)
            if truncate:
                audio = audio[..., : int(30 * sr)]  # first 30 seconds (tts_model.py:880-884)
            audio_conditioning = _convert_audio(audio, sr, self.config.mimi.sample_rate)
        # audio tensor [channels=1, samples] at the model rate -> conditioning (tts_model.py:889-890,379-388)
        _, cond = self.engine.encode_voice(audio_conditioning)
        return self.get_state_for_conditioning(cond[None])

    def get_state_for_conditioning(self, conditioning: torch.Tensor) -> dict:
        """Voice state from pre-computed speaker conditioning f32[1, T, d_model] (the output of the
        reference's `_encode_audio`, tts_model.py:379-388).  Prepends `bos_before_voice` like
        tts_model.py:893-894 and prefills the FlowLM."""
        eng = self.engine
        prompt = conditioning.to(self.device, torch.float32)
        if self.config.flow_lm.insert_bos_before_voice:
            prompt = torch.cat([eng.bos_before_voice, prompt], dim=1)
        T = prompt.shape[1]
        st = eng.new_lm_state(1, T)
        eng.lm_prefill(st, prompt)
        state = _export_lm_state(eng, st, T)
        st.close()
        return state

    # ---- generation ---------------------------------------------------------------------------
    @torch.no_grad()
    def generate_audio(self, model_state: dict, text_to_generate: str, max_tokens: int = MAX_TOKEN_PER_CHUNK,
                       frames_after_eos: int | None = None, copy_state: bool = True) -> torch.Tensor:
        chunks = list(self.generate_audio_stream(model_state, text_to_generate, max_tokens, frames_after_eos, copy_state))
        return torch.cat(chunks, dim=0)

    @torch.no_grad()
    def generate_audio_stream(self, model_state: dict, text_to_generate: str, max_tokens: int = MAX_TOKEN_PER_CHUNK,
                              frames_after_eos: int | None = None, copy_state: bool = True):
        """Yields fp32 CPU tensors of `frame_samples` (1920) samples (reference tts_model.py:545-631)."""
        if frames_after_eos is None:
            frames_after_eos = self.model_recommended_frames_after_eos
        chunks = split_into_best_sentences(self.tokenizer.encode, self.tokenizer.sp, text_to_generate, max_tokens,
                                           self.pad_with_spaces_for_short_inputs, self.remove_semicolons)
        for chunk in chunks:
            _, guess = prepare_text_prompt(chunk, self.pad_with_spaces_for_short_inputs, self.remove_semicolons)
            guess += 2
            effective = frames_after_eos if frames_after_eos is not None else guess
            yield from self._generate_audio_stream_short_text(model_state, chunk, effective, copy_state)

    @torch.no_grad()
    def generate_audio_batch(self, model_states, texts, frames_after_eos: int | None = None) -> list:
        """Generate several utterances concurrently on one GPU (not in the reference, which is batch 1 and
        requires equal cache offsets across a batch: tts_model.py:491-492, transformer.py:12-13).

        `model_states`: one voice state or a list (one per text); `texts`: list of short texts (each must
        fit one chunk of `MAX_TOKEN_PER_CHUNK` tokens; split longer texts first).  Utterances that share a voice
        state and a token count are prefilled together as one batch (one GEMM pass per group, the voice's KV cloned
        from its device-resident copy), every utterance then owns one row of a batch state, and all rows are decoded
        in lock-step with per-row positions on the two-stream step pipeline (FlowLM step t+1 overlaps codec frame t).
        The host never waits for the GPU inside the loop: the EOS flags and the PCM of a step land in pinned memory
        and are read a few steps later, so a row runs at most `StepPipeline.nb` steps past its end (those frames are
        dropped); per-row EOS bookkeeping follows tts_model.py:756-768.  With temp == 0 each waveform equals the
        single-utterance result; with temp > 0 rows draw independent noise (the reference's sequential use of the
        global generator cannot be reproduced across a batch).  Returns a list of fp32 CPU tensors."""
        from .batching import eos_bookkeeping_rows
        from .engine import StepPipeline

        eng = self.engine
        B = len(texts)
        if not isinstance(model_states, (list, tuple)):
            model_states = [model_states] * B
        if len(model_states) != B or B == 0:
            raise ValueError("need one voice state per text")
        # rows of one voice next to each other: they share the voice's keys (KvPrefix), and the decode attention scores a
        # shared prefix once per 4 neighbouring rows (attn_cascade_kernel)
        first: dict = {}
        for i, m in enumerate(model_states):
            first.setdefault(id(m), i)
        order = sorted(range(B), key=lambda i: first[id(model_states[i])])
        if order != list(range(B)):
            got = self.generate_audio_batch([model_states[i] for i in order], [texts[i] for i in order], frames_after_eos)
            res = [None] * B
            for pos, i in enumerate(order):
                res[i] = got[pos]
            return res
        toks, gens, faes, t0s = [], [], [], []
        for text, ms_ in zip(texts, model_states):
            # the reference tokenises the chunk text as split_into_best_sentences returns it (stripped)
            chunk = split_into_best_sentences(self.tokenizer.encode, self.tokenizer.sp, text, 10 ** 9,
                                              self.pad_with_spaces_for_short_inputs, self.remove_semicolons)
            if len(chunk) != 1:
                raise ValueError("generate_audio_batch takes single-chunk texts")
            ids = self.tokenizer.encode(chunk[0])
            if len(ids) > MAX_TOKEN_PER_CHUNK:
                raise ValueError(f"text has {len(ids)} tokens; split it into chunks of <= {MAX_TOKEN_PER_CHUNK}")
            _, guess = prepare_text_prompt(chunk[0], self.pad_with_spaces_for_short_inputs, self.remove_semicolons)
            fae = frames_after_eos if frames_after_eos is not None else (
                self.model_recommended_frames_after_eos if self.model_recommended_frames_after_eos is not None else guess + 2)
            toks.append(torch.tensor(ids, dtype=torch.long)[None, :])
            gens.append(estimate_max_gen_len(len(ids), self.config.mimi.frame_rate))
            faes.append(fae)
        # one device-resident copy per distinct voice state (no per-utterance import, no host sync on a hit)
        voices = {}
        for ms_ in model_states:
            if id(ms_) not in voices:
                voices[id(ms_)] = self._voice_acquire(ms_)
        try:
            t0s = [voices[id(ms_)][1] for ms_ in model_states]
            steps_max = max(gens)
            use_noise = self.temp > 0
            need = max(t + tk.shape[1] for t, tk in zip(t0s, toks)) + steps_max + StepPipeline.NB_EVENTS + 2
            cap = -(-need // 256) * 256
            key = ("batch", B, cap, self.lsd_decode_steps, float(self.eos_threshold), use_noise)
            ctx = self._ctx_cache.pop(key, None)
            if ctx is None:
                self._drop_batch_contexts(keep=1)
                batch, ms = eng.new_lm_state(B, cap), eng.new_mimi_state(B)
                if use_noise:
                    batch.set_noise(self.temp, int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
                pipe = StepPipeline(eng, batch, ms, None, self.lsd_decode_steps, float(self.eos_threshold), mode="events")
                ctx = dict(st=batch, ms=ms, pipe=pipe)
            batch, ms, pipe = ctx["st"], ctx["ms"], ctx["pipe"]
            nb = pipe.nb
            # group prefill: rows with the same voice and token count share one batched pass (lengths differ across groups)
            groups: dict = {}
            for b in range(B):
                groups.setdefault((id(model_states[b]), toks[b].shape[1]), []).append(b)
            tmp = []
            for (vid, Tt), rows in groups.items():
                voice_st, t_voice = voices[vid]
                grp = eng.new_lm_state(len(rows), t_voice + Tt)
                tmp.append(grp)
                grp.copy_from(voice_st)
                eng.lm_prefill(grp, eng.embed_text(torch.cat([toks[b] for b in rows], dim=0)))
                for i, b in enumerate(rows):
                    batch.copy_row_from(b, grp, i)
            pipe.restart()  # zero codec carries, on the codec stream
            gens_a, faes_a = np.asarray(gens), np.asarray(faes)
            eos_step = np.full(B, -1, np.int64)
            n_emit = np.full(B, -1, np.int64)  # frames to keep for row b (decided when its loop would break)
            out = np.empty((B, steps_max, eng.frame_samples), np.float32)
            flags_np = [f_.numpy() for f_ in pipe.flag]   # views of the pinned buffers the kernels write
            pcm_np = [p_.numpy() for p_ in pipe.pcm]
            t = collected = 0

            def collect(f):
                pipe.done_event(f).synchronize()  # codec frame f done => FlowLM step f done too
                eos_bookkeeping_rows(f, gens_a, faes_a, eos_step, n_emit, flags_np[f % nb] != 0)
                if f < steps_max:
                    out[:, f] = pcm_np[f % nb]    # plain memcpy per row (numpy: no intra-op thread team)

            while True:
                decided = bool((n_emit >= 0).all())
                if t - collected >= nb or (collected < t and pipe.done_event(collected).query()):
                    collect(collected)  # before its pinned buffers are reused (mandatory), or as soon as it is ready
                    collected += 1
                    continue
                if t >= steps_max or (decided and t >= int(n_emit.max())):
                    if collected == t:
                        break
                    collect(collected)
                    collected += 1
                    continue
                pipe.step()
                t += 1
            pipe.sync()
            for g in tmp:
                g.close()
            if batch.error():  # a cooperative kernel gave up waiting for a peer: the audio of this batch is invalid
                raise RuntimeError("libptts: a cooperative FlowLM kernel timed out; the audio of this batch is invalid")
            self._ctx_cache[key] = ctx
        finally:
            for v in voices.values():
                self._voice_release(v)
        res = []
        for b in range(B):
            n = int(n_emit[b]) if n_emit[b] >= 0 else gens[b]
            if eos_step[b] < 0:
                logger.warning("Maximum generation length reached without EOS, this very often indicates an error.")
            # a view of this call's own frame buffer (allocated per call, so nothing rewrites it): no second copy of the audio
            res.append(torch.from_numpy(out[b, :n].reshape(-1)) if n > 0 else torch.zeros(0))
        return res

    def _drop_batch_contexts(self, keep: int = 0):
        """cached batch contexts (state + graphs of `generate_audio_batch`) beyond the `keep` most recent are released"""
        keys = [k for k in self._ctx_cache if k and k[0] == "batch"]
        for k in keys[: max(0, len(keys) - keep)]:
            c = self._ctx_cache.pop(k)
            c["pipe"].close()
            c["st"].close()
            c["ms"].close()

    class _Voice:
        """one device-resident voice: engine-layout KV of a voice-state dict + what identifies the dict's contents"""

        __slots__ = ("sig", "st", "keep", "t_voice", "refs", "evicted")

        def __init__(self, sig, st, keep, t_voice):
            self.sig, self.st, self.keep, self.t_voice, self.refs, self.evicted = sig, st, keep, t_voice, 0, False

        def __iter__(self):  # `voice_st, t_voice = entry`
            return iter((self.st, self.t_voice))

        def __getitem__(self, i):
            return (self.st, self.t_voice)[i]

    def _voice_acquire(self, model_state: dict, t_voice: int | None = None) -> "_Voice":
        """Device-resident engine-layout copy of a voice state dict, built on first use and reused while the dict's
        tensors are unchanged (same storage, same in-place version); at most 8 voices stay resident.  On a hit nothing
        touches the device (the reference reads `offset` with `.item()` on every call, tts_model.py:412-414: a device
        sync when the state lives on the GPU).  The entry is REFERENCE-COUNTED: an eviction (more than 8 voices, or the
        dict's tensors changed) by another thread - the cache is shared with ContinuousBatcher's scheduler thread - only
        marks it; its `LMState` is destroyed by the last `_voice_release` (the clone kernels of a generation
        that was still using the handle raced with its hipFree)."""
        with self._voice_lock:
            eng = self.engine
            sig = tuple((model_state[_layer_key(i)]["cache"].data_ptr(), model_state[_layer_key(i)]["cache"]._version,
                         model_state[_layer_key(i)]["offset"].data_ptr(), model_state[_layer_key(i)]["offset"]._version)
                        for i in range(eng.L))
            hit = self._voice_cache.get(id(model_state))
            if hit is not None and hit.sig == sig:
                hit.refs += 1
                return hit
            if hit is not None:
                self._voice_evict(id(model_state))
            while len(self._voice_cache) >= 8:
                self._voice_evict(next(iter(self._voice_cache)))
            if t_voice is None:
                t_voice = _state_current_end(model_state)
            vst = eng.new_lm_state(1, max(t_voice, 1))
            _import_lm_state(eng, vst, model_state, t_voice)
            # the entry keeps the source tensors alive, so a recycled id() / data_ptr() can never alias a cached voice
            keep = [model_state[_layer_key(i)][k] for i in range(eng.L) for k in ("cache", "offset")]
            ent = self._Voice(sig, vst, keep, t_voice)
            ent.refs = 1
            self._voice_cache[id(model_state)] = ent
            return ent

    def _voice_evict(self, key):
        """caller holds the lock"""
        ent = self._voice_cache.pop(key)
        ent.evicted = True
        if ent.refs == 0:
            ent.st.close()

    def _voice_release(self, ent: "_Voice"):
        with self._voice_lock:
            ent.refs -= 1
            if ent.evicted and ent.refs == 0:
                ent.st.close()

    def _draw_noise(self, out: torch.Tensor):
        """Same draws as the reference CPU path (flow_lm.py:131-137): torch's global CPU generator."""
        std = self.temp ** 0.5
        if self.noise_clamp is None:
            torch.nn.init.normal_(out, mean=0.0, std=std)
        else:
            torch.nn.init.trunc_normal_(out, mean=0.0, std=std, a=-self.noise_clamp, b=self.noise_clamp)

    def _generate_audio_stream_short_text(self, model_state: dict, text: str, frames_after_eos: int, copy_state: bool):
        eng = self.engine
        tokens = torch.tensor(self.tokenizer.encode(text), dtype=torch.long)[None, :]
        Tt = tokens.shape[1]
        max_gen_len = estimate_max_gen_len(Tt, self.config.mimi.frame_rate)
        voice = self._voice_acquire(model_state)  # no device sync when the voice is already resident
        voice_st, t_voice = voice
        use_noise = self.temp > 0
        t_start = time.monotonic()
        # states, scratch and captured graphs are reused across chunks and calls (capacity rounded up)
        cap = -(-(t_voice + Tt + max_gen_len) // 256) * 256
        key = (cap, self.lsd_decode_steps, float(self.eos_threshold), use_noise)
        ctx = self._ctx_cache.pop(key, None)
        if ctx is None:
            from .engine import StepPipeline

            st = eng.new_lm_state(1, cap)
            ms = eng.new_mimi_state(1)
            noise_dev = torch.zeros(1, eng.ldim, device=self.device) if use_noise else None
            pipe = StepPipeline(eng, st, ms, noise_dev, self.lsd_decode_steps, float(self.eos_threshold), mode="hostsync")
            ctx = dict(st=st, ms=ms, noise_dev=noise_dev, pipe=pipe, noise_host=torch.zeros(1, eng.ldim).pin_memory())
        st, ms, noise_dev, pipe = ctx["st"], ctx["ms"], ctx["noise_dev"], ctx["pipe"]
        noise_host = ctx["noise_host"]
        # per-chunk clone of the voice state (replaces deepcopy + _expand_kv_cache, tts_model.py:637-638,390-421): the
        # voice's KV lives on the device in the engine's layout, one row-copy kernel clones it, nothing synchronises.
        # Clone + prefill are queued BEFORE the codec reset: the GPU starts on them while the host issues the rest.
        pipe.flush()
        st.copy_from(voice_st)
        eng.lm_prefill(st, eng.embed_text(tokens))            # text prefill (tts_model.py:722-725)
        pipe.restart()
        if use_noise:
            # the reference's text prefill runs the whole forward, including one (discarded) noise draw
            # (tts_model.py:722-725 -> flow_lm.py:131-137): consume it to stay on the same generator stream
            self._draw_noise(torch.empty(1, eng.ldim))
        total = 0
        try:
            eos_step = None
            emitted = 0    # frames handed to the codec
            yielded = 0    # frames handed to the caller

            def pop(frame):
                pipe.done_event(frame).synchronize()
                return pipe.pcm_of(frame)[0].clone()

            for step in range(max_gen_len):
                if yielded == 0 and emitted == 1:
                    # first chunk: hand it over before the next FlowLM step is queued beside its codec frame (the two
                    # would share the chip and the chunk would arrive later); from here on step t+1 overlaps frame t
                    chunk = pop(0)
                    yielded = 1
                    total += chunk.shape[0]
                    yield chunk
                if use_noise:
                    self._draw_noise(noise_host)
                    with torch.cuda.stream(eng.stream):
                        noise_dev.copy_(noise_host, non_blocking=True)
                pipe.lm_step_async()
                # hand finished frames to the consumer while the GPU runs this step
                while yielded < emitted:
                    chunk = pop(yielded)
                    yielded += 1
                    total += chunk.shape[0]
                    yield chunk
                # the EOS decision is a host decision, as in the reference (tts_model.py:761)
                if bool(pipe.wait_flags(step)[0].item()) and eos_step is None:
                    eos_step = step
                if eos_step is not None and step >= eos_step + frames_after_eos:
                    break  # the break-step latent is not decoded (tts_model.py:763-764)
                pipe.decode_async(step)  # codec frame `step` overlaps FlowLM step `step + 1`
                emitted += 1
            else:
                logger.warning("Maximum generation length reached without EOS, this very often indicates an error.")
            while yielded < emitted:
                chunk = pop(yielded)
                yielded += 1
                total += chunk.shape[0]
                yield chunk
            if not copy_state:
                # the reference mutates the caller's state in place (tts_model.py:637-638)
                pipe.sync()
                n = int(st.offsets()[0])
                model_state.update(_export_lm_state(eng, st, n))
        finally:
            pipe.sync()
            self._voice_release(voice)
            self._ctx_cache[key] = ctx
        if st.error():  # a cooperative kernel gave up waiting for a peer (GPU oversubscribed beyond the library's contract)
            raise RuntimeError("libptts: a cooperative FlowLM kernel timed out; the audio of this chunk is invalid")
        dur_ms = int(total * 1000 / self.config.mimi.sample_rate)
        gen_ms = max(1, int((time.monotonic() - t_start) * 1000))
        logger.info("Generated: %d ms of audio in %d ms so %.2fx faster than real-time", dur_ms, gen_ms, dur_ms / gen_ms)


def _audio_read(path):
    """WAV via the standard library (reference data/audio.py:23-36): int16 -> float32 / 32768, channel mean."""
    import wave

    path = Path(path)
    if path.suffix.lower() != ".wav":
        raise ImportError("only WAV files are supported for audio prompts in this build")
    with wave.open(str(path), "rb") as w:
        sr, nch = w.getframerate(), w.getnchannels()
        x = np.frombuffer(w.readframes(-1), dtype=np.int16).astype(np.float32) / 32768.0
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return torch.from_numpy(x).unsqueeze(0), sr


def _convert_audio(wav: torch.Tensor, from_rate: int, to_rate: int) -> torch.Tensor:
    """polyphase resampling like the reference (data/audio_utils.py:8-28)"""
    if from_rate != to_rate:
        from scipy.signal import resample_poly

        g = math.gcd(int(from_rate), int(to_rate))
        wav = torch.from_numpy(resample_poly(wav.cpu().numpy(), int(to_rate) // g, int(from_rate) // g, axis=-1)).to(wav.dtype)
    return wav


# ---- model-state helpers (reference format) ---------------------------------------------------
def _layer_key(i: int) -> str:
    return f"transformer.layers.{i}.self_attn"


def _state_current_end(model_state: dict) -> int:
    for ms in model_state.values():
        off = ms.get("offset")
        if off is not None:
            return int(off.view(-1)[0].item())
    raise ValueError("Could not find offset in model state")


def _import_lm_state(eng: Engine, st, model_state: dict, t: int):
    for i in range(eng.L):
        st.import_layer(i, model_state[_layer_key(i)]["cache"], t)


def _export_lm_state(eng: Engine, st, t: int) -> dict:
    out = {}
    for i in range(eng.L):
        out[_layer_key(i)] = dict(cache=st.export_layer(i, t),
                                  offset=torch.full((1,), t, dtype=torch.long, device=eng.device))
    return out


def export_model_state(model_state: dict, dest: str | Path):
    """safetensors keys "<module>/<key>" (reference tts_model.py:1047-1052)."""
    flat = {}
    for module_name, module_state in model_state.items():
        for key, value in module_state.items():
            flat[f"{module_name}/{key}"] = value.detach().cpu().contiguous()
    safetensors.torch.save_file(flat, str(dest))


def _import_model_state(source: str | Path, device) -> dict:
    """Reads a voice-state file, including the legacy `current_end` key whose shape[0] is the offset
    (reference tts_model.py:1055-1072)."""
    result: dict = {}
    with safetensors.safe_open(str(source), framework="pt") as f:
        for key in f.keys():
            module_name, tensor_key = key.split("/")
            result.setdefault(module_name, {})
            if tensor_key == "current_end":
                n = f.get_tensor(key).shape[0]
                result[module_name]["offset"] = torch.full((1,), n, dtype=torch.long, device=device)
            else:
                result[module_name][tensor_key] = f.get_tensor(key).to(device)
    return result
