"""Python host wrapper over the C ABI (include/ptts.h).  PyTorch-ROCm is used only for device
memory and streams; all arithmetic of the hot path runs in libptts.so.

Mirrors the reference's internal seam:
  * `Engine.lm_prefill / lm_decode_step`  <->  `TTSModel._run_flow_lm_and_increment_step`
    (reference tts_model.py:317-346)
  * `Engine.mimi_decode`                  <->  the body of `_decode_audio_worker`
    (reference tts_model.py:444-455) = de-normalise + quantizer + `MimiModel.decode_from_latent`
  * `LMState` / `MimiState`               <->  `init_states(...)` dicts (reference stateful_module.py:7-16)
"""

from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib
from .config import Config
from .weights import mimi_encode_spec, state_dict_spec


def _ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def make_ptts_config(cfg: Config) -> _lib.PttsConfig:
    t, m, sn = cfg.flow_lm.transformer, cfg.mimi.transformer, cfg.mimi.seanet
    if len(sn.ratios) != 3:
        raise ValueError("SEANet decoder with 3 upsampling stages expected")
    pc = _lib.PttsConfig()
    pc.d_model, pc.num_heads, pc.num_layers = t.d_model, t.num_heads, t.num_layers
    pc.ff_dim = t.d_model * t.hidden_scale
    pc.ldim = cfg.mimi.quantizer.dimension
    pc.flow_dim, pc.flow_depth = cfg.flow_lm.flow.dim, cfg.flow_lm.flow.depth
    pc.max_period = float(t.max_period)
    pc.m_dim, pc.m_heads, pc.m_layers, pc.m_ff = m.d_model, m.num_heads, m.num_layers, m.dim_feedforward
    pc.m_context = m.context
    pc.m_max_period = float(m.max_period)
    pc.n_filters = sn.n_filters
    pc.ratios = (C.c_int32 * 3)(*[int(r) for r in sn.ratios])
    pc.kernel_size, pc.res_kernel_size, pc.last_kernel_size = sn.kernel_size, sn.residual_kernel_size, sn.last_kernel_size
    pc.compress = sn.compress
    pc.upsample_stride = cfg.upsample_stride
    return pc


class LMState:
    """FlowLM KV caches of `batch` sequences with capacity `t_cap` positions."""

    def __init__(self, engine: "Engine", batch: int, t_cap: int):
        self.engine, self.batch, self.t_cap = engine, batch, t_cap
        h = C.c_void_p()
        _lib.check(engine.lib.ptts_lm_state_create(engine.handle, batch, t_cap, C.byref(h)))
        self.handle = h
        engine._states.add(self)

    def close(self):
        if self.handle is not None:
            self.engine.lib.ptts_lm_state_destroy(self.handle)
            self.handle = None

    def reset(self):
        _lib.check(self.engine.lib.ptts_lm_state_reset(self.handle, self.engine._sp))

    def set_noise(self, temp: float, seed: int = 0):
        """device-side N(0, temp) noise for steps called with noise=None (perf runs)"""
        _lib.check(self.engine.lib.ptts_lm_set_noise(self.handle, float(temp), int(seed)))

    def error(self) -> bool:
        """True after a cooperative kernel of this state gave up waiting for a peer workgroup"""
        r = self.engine.lib.ptts_lm_state_error(self.handle, self.engine._sp)
        _lib.check(min(int(r), 0))
        return bool(r)

    def offsets(self) -> np.ndarray:
        out = (C.c_int32 * self.batch)()
        _lib.check(self.engine.lib.ptts_lm_state_offsets(self.handle, out, self.engine._sp))
        return np.array(out[:], dtype=np.int64)

    def import_layer(self, layer: int, cache: torch.Tensor, t: int):
        """cache: f32[2, Bsrc, >=t, H, 64] in the reference layout (reference transformer.py:32-36)."""
        e = self.engine
        if cache.dim() != 5 or cache.shape[0] != 2 or cache.shape[1] not in (1, self.batch) or cache.shape[2] < t \
                or cache.shape[3] != e.H or cache.shape[4] != 64:
            raise ValueError(f"voice-state cache of layer {layer}: expected [2, 1|{self.batch}, >={t}, {e.H}, 64], "
                             f"got {tuple(cache.shape)}")
        cache = cache[:, :, :t].to(self.engine.device, torch.float32).contiguous()
        self.engine._pre()
        _lib.check(self.engine.lib.ptts_lm_state_import(self.handle, layer, _ptr(cache), cache.shape[1], t, self.engine._sp))
        self.engine.sync()

    def export_layer(self, layer: int, t: int) -> torch.Tensor:
        e = self.engine
        out = torch.empty((2, self.batch, t, e.H, 64), dtype=torch.float32, device=e.device)
        _lib.check(e.lib.ptts_lm_state_export(self.handle, layer, _ptr(out), t, e._sp))
        e.sync()
        return out

    def copy_row_from(self, row: int, src: "LMState", src_row: int = 0):
        """row `row` <- sequence `src_row` of `src`; rows may have different lengths"""
        _lib.check(self.engine.lib.ptts_lm_state_copy_row_from(self.handle, row, src.handle, src_row, self.engine._sp))

    def copy_from(self, src: "LMState"):
        _lib.check(self.engine.lib.ptts_lm_state_copy(self.handle, src.handle, self.engine._sp))

    def set_row_active(self, row: int, active: bool):
        """continuous batching: a parked row stays at position 0 (copy_row_from re-activates it)"""
        _lib.check(self.engine.lib.ptts_lm_state_set_row_active(self.handle, row, int(bool(active)), self.engine._sp))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MimiState:
    def __init__(self, engine: "Engine", batch: int):
        self.engine, self.batch = engine, batch
        h = C.c_void_p()
        _lib.check(engine.lib.ptts_mimi_state_create(engine.handle, batch, C.byref(h)))
        self.handle = h
        engine._states.add(self)

    def close(self):
        if self.handle is not None:
            self.engine.lib.ptts_mimi_state_destroy(self.handle)
            self.handle = None

    def reset(self, stream: torch.cuda.Stream | None = None):
        sp = self.engine._sp if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.engine.lib.ptts_mimi_state_reset(self.handle, sp))

    def reset_row(self, row: int, stream: torch.cuda.Stream | None = None):
        """zero the streaming carries of one sequence (a new utterance joins in `row`)"""
        sp = self.engine._sp if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.engine.lib.ptts_mimi_state_reset_row(self.handle, row, sp))

    def set_pcm_i16(self, buf: torch.Tensor | None):
        """int16 copy of the PCM written by the decodes / graph captures issued after this call (None: off)"""
        _lib.check(self.engine.lib.ptts_mimi_set_pcm_i16(self.handle, _ptr(buf) if buf is not None else None))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """Weights of one model on one GPU + entry points of the hot path."""

    # PTTS_QUANT_* / PTTS_CODEC_BF16 / PTTS_CODEC_FP8 / PTTS_LM_BF16 (include/ptts.h)
    QUANT_GROUPS = {"attention": 1, "ffn": 2, "codec_bf16": 4, "codec_fp8": 8, "lm_bf16": 16, "codec_split": 32}

    def __init__(self, cfg: Config, weights: dict, device: str | torch.device = "cuda:0",
                 quantize_groups: set | frozenset | None = None, _packed: str | None = None):
        """`quantize_groups`: subset of {"attention", "ffn"} (the keys of the reference's
        quantization.apply_dynamic_int8): those Linear layers of the FlowLM transformer get int8 weights; plus
        "codec_bf16": the Mimi decoder runs with bf16 weights / activations and fp32 accumulation; "codec_fp8": its SEANet
        convolutions run on the fp8 MFMA (e4m3 weights + activations, the transformer as under codec_bf16); "lm_bf16": bf16
        weights and bf16-rounded activation operands for the FlowLM transformer's Linear layers (no reference counterpart
        for any of the three; BASELINE.json config #5 / SURVEY 8(f).4)."""
        self.lib = _lib.load()
        self.handle = None
        self._states = weakref.WeakSet()
        self._tuned = set()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("pocket_tts_amd runs on a ROCm GPU only (device must be cuda:N)")
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        if _packed is not None:
            # a packed engine: the device images come from the file, `weights` holds only what stays on the Python side
            self.has_voice_encoder = bool(weights.get("_has_voice_encoder", True))
            self.quantize_groups = frozenset(quantize_groups or ())
            _lib.check(self.lib.ptts_create_from_file(str(_packed).encode(), self.device.index or 0, C.byref(h)))
        else:
            spec = state_dict_spec(cfg)
            optional = set(mimi_encode_spec(cfg))  # checkpoints without voice cloning may lack the encoder
            if any(n not in weights for n in optional):
                spec = {k: v for k, v in spec.items() if k not in optional}
            self.has_voice_encoder = all(n in weights for n in optional)
            keep, arr = [], (_lib.PttsTensor * len(spec))()
            for i, (name, shape) in enumerate(spec.items()):
                if name not in weights:
                    raise KeyError(f"checkpoint is missing tensor {name}")
                w = weights[name]
                if isinstance(w, np.ndarray):
                    w = torch.from_numpy(w)
                if tuple(w.shape) != tuple(shape):
                    raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(w.shape)}")
                w = w.to(self.device, torch.float32).contiguous()
                keep.append(w)
                arr[i].name = name.encode()
                arr[i].d_data = w.data_ptr()
                arr[i].numel = w.numel()
            torch.cuda.synchronize(self.device)
            pc = make_ptts_config(cfg)
            flags = 0
            for g in quantize_groups or ():
                if g not in self.QUANT_GROUPS:
                    raise ValueError(f"unknown quantization group {g!r} (this build supports {sorted(self.QUANT_GROUPS)})")
                flags |= self.QUANT_GROUPS[g]
            self.quantize_groups = frozenset(quantize_groups or ())
            _lib.check(self.lib.ptts_create_ex(C.byref(pc), arr, len(spec), self.device.index or 0, flags, C.byref(h)))
            del keep
        self.handle = h
        # All work is queued on a torch-owned stream passed through the ABI's `stream` argument, so torch's
        # caching allocator (record_stream) and the library agree on one stream whose lifetime torch manages.
        self.stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PTTS_PRIO_LM", "0")))
        self._sp = C.c_void_p(self.stream.cuda_stream)
        t = cfg.flow_lm.transformer
        self.D, self.H, self.L = t.d_model, t.num_heads, t.num_layers
        self.ldim = cfg.mimi.quantizer.dimension
        self.frame_samples = cfg.frame_samples
        # embedding table + voice-path parameters stay as torch tensors (gather / prefill inputs)
        self.embed = torch.as_tensor(weights["flow_lm.conditioner.embed.weight"]).to(self.device, torch.float32)
        self.bos_before_voice = None
        if "flow_lm.bos_before_voice" in weights:
            self.bos_before_voice = torch.as_tensor(weights["flow_lm.bos_before_voice"]).to(self.device, torch.float32)

    # ---- packed-engine files (offline packer): device images + the few tensors the Python side keeps + the config
    def save_packed(self, path):
        """Writes `<path>` (device images, C ABI `ptts_engine_save`), `<path>.aux.safetensors` (embedding table,
        bos_before_voice) and `<path>.yaml` (model config + weight format).  Load with `Engine.from_packed`."""
        import safetensors.torch
        import yaml

        from .config import config_to_dict

        _lib.check(self.lib.ptts_engine_save(self.handle, str(path).encode()))
        aux = {"flow_lm.conditioner.embed.weight": self.embed.cpu().contiguous()}
        if self.bos_before_voice is not None:
            aux["flow_lm.bos_before_voice"] = self.bos_before_voice.cpu().contiguous()
        safetensors.torch.save_file(aux, str(path) + ".aux.safetensors")
        with open(str(path) + ".yaml", "w") as f:
            yaml.safe_dump(dict(config=config_to_dict(self.cfg), quantize_groups=sorted(self.quantize_groups),
                                has_voice_encoder=bool(self.has_voice_encoder)), f)

    @classmethod
    def from_packed(cls, path, device: str | torch.device = "cuda:0") -> "Engine":
        """Engine from the files `save_packed` wrote: no checkpoint, no packing, no quantisation pass."""
        import safetensors.torch
        import yaml

        from .config import config_from_dict

        meta = yaml.safe_load(open(str(path) + ".yaml"))
        aux = safetensors.torch.load_file(str(path) + ".aux.safetensors")
        aux["_has_voice_encoder"] = meta["has_voice_encoder"]
        return cls(config_from_dict(meta["config"]), aux, device, quantize_groups=set(meta["quantize_groups"]), _packed=str(path))

    # ---- stream ordering between torch's current stream and the engine stream
    def _pre(self):
        """engine stream waits for work already queued on torch's current stream (inputs)"""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _post(self):
        """torch's current stream waits for the engine stream (outputs)"""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    # ---- utilities
    def set_option(self, key: str, value: int):
        """engine options of include/ptts.h (`flow_cluster`, ...); applies to later steps / captures"""
        _lib.check(self.lib.ptts_set_option(self.handle, key.encode(), int(value)))

    def sync(self):
        _lib.check(self.lib.ptts_sync(self.handle, self._sp))

    @property
    def stream_ptr(self):
        return self.stream.cuda_stream

    def timer_start(self):
        _lib.check(self.lib.ptts_timer_start(self.handle, self._sp))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        _lib.check(self.lib.ptts_timer_stop_ms(self.handle, self._sp, C.byref(ms)))
        return ms.value

    def tune(self, batch: int, force: bool = False) -> str:
        """Measure the tile configuration of every GEMM on the step path for this batch size (once per
        engine and batch; `PTTS_NO_TUNE=1` keeps the static heuristic).  Returns the tuner's log.

        `PTTS_TUNE_CACHE=<file>`: tile choices measured by an earlier process are imported first (a deployment,
        or a profiling run whose counters would perturb the timings, reuses them).  The tuner then still runs:
        it only measures GEMM shapes ABSENT from the table (another model, batch or weight format), so a cache
        never silently leaves shapes on the heuristic.  Newly measured shapes are appended to
        `PTTS_TUNE_CACHE_OUT` (default: the cache file itself).  Files carry the table-format version."""
        if os.environ.get("PTTS_NO_TUNE") == "1" or (batch in self._tuned and not force):
            return ""
        ver = int(self.lib.ptts_tune_version())
        head = f"# ptts-tune-version {ver}\n"
        cache = os.environ.get("PTTS_TUNE_CACHE")
        out = os.environ.get("PTTS_TUNE_CACHE_OUT", cache)
        if cache and os.path.exists(cache) and not force:
            text = open(cache).read()
            if text.startswith(head):
                self.lib.ptts_tune_import(self.handle, text.encode())
        before = self._tune_table()
        self._pre()
        _lib.check(self.lib.ptts_tune(self.handle, int(batch), self._sp))
        if batch == 1:
            # streaming path: the text prefill of a chunk sits on the first-chunk latency; its GEMM shapes depend on
            # ceil(tokens / 16) only (chunks hold <= 50 tokens + padding)
            for t in (16, 32, 48, 64):
                _lib.check(self.lib.ptts_tune_prefill(self.handle, 1, t, self._sp))
        elif os.environ.get("PTTS_TUNE_PREFILL_TOKENS"):
            # batched prefill (a group of requests with one token count): GEMMs of batch x ceil(tokens / 16) row tiles
            for t in os.environ["PTTS_TUNE_PREFILL_TOKENS"].split(","):
                _lib.check(self.lib.ptts_tune_prefill(self.handle, int(batch), int(t), self._sp))
        self._tuned.add(batch)
        after = self._tune_table()
        new = [ln for ln in after if ln not in before]
        if out and new:
            t = self.cfg.flow_lm.transformer
            fresh = not os.path.exists(out) or not open(out).read().startswith(head)
            with open(out, "w" if fresh else "a") as f:
                if fresh:
                    f.write(head)
                f.write(f"# batch {int(batch)} d_model {t.d_model} layers {t.num_layers} "
                        f"quant {'+'.join(sorted(self.quantize_groups)) or 'none'}\n" + "\n".join(new) + "\n")
        return (self.lib.ptts_tune_log(self.handle) or b"").decode()

    def streams_overlap(self, a, b) -> bool:
        """do two torch streams execute concurrently (False: the runtime mapped them to one hardware queue)"""
        return int(_lib.check(self.lib.ptts_streams_overlap(self.handle, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream)))) == 1

    def concurrent_stream(self, other, tries: int = 8):
        """a stream that runs CONCURRENTLY with `other`.  HIP assigns streams round-robin to a few hardware queues
        (GPU_MAX_HW_QUEUES, default 4); a FlowLM / codec stream pair that shares a queue serialises the pipeline
        (0.88 -> 1.15 ms per step at batch 64), so the pair is checked and another stream taken if it collides."""
        prio = int(os.environ.get("PTTS_PRIO_CODEC", "0"))
        held = []  # rejected streams stay alive until the choice is made, so the round-robin moves on
        for _ in range(tries):
            s = torch.cuda.Stream(device=self.device, priority=prio)
            if self.streams_overlap(other, s):
                return s
            held.append(s)
        return held[-1]

    def _tune_table(self) -> list:
        buf = C.create_string_buffer(1 << 20)
        n = self.lib.ptts_tune_export(self.handle, buf, len(buf))
        _lib.check(int(n))
        return [ln for ln in buf.value.decode().splitlines() if ln.strip()]

    def profile_start(self):
        _lib.check(self.lib.ptts_profile_start(self.handle))

    def profile_stop(self) -> list:
        """-> [{site, kernel, count, total_ms, bytes, flops}] in launch order"""
        buf = C.create_string_buffer(1 << 20)
        n = self.lib.ptts_profile_stop(self.handle, buf, len(buf))
        _lib.check(int(n))
        rows = []
        for line in buf.value.decode().splitlines():
            site, kernel, cnt, ms, by, fl = line.split()
            rows.append(dict(site=site, kernel=kernel, count=int(float(cnt)), total_ms=float(ms),
                             bytes=float(by), flops=float(fl)))
        return rows

    def lm_weight_bytes(self) -> int:
        return self.lib.ptts_lm_weight_bytes(self.handle)

    def mimi_weight_bytes(self) -> int:
        return self.lib.ptts_mimi_weight_bytes(self.handle)

    def new_lm_state(self, batch: int, t_cap: int) -> LMState:
        return LMState(self, batch, t_cap)

    def new_mimi_state(self, batch: int) -> MimiState:
        return MimiState(self, batch)

    # ---- FlowLM
    def embed_text(self, tokens: torch.Tensor) -> torch.Tensor:
        """`LUTConditioner._get_condition` gather (reference text.py:74-76); prefill input only.  tokens: int64 [B, T]
        (host or device); ids are validated where they are cheap to read, the gather itself is `ptts_embed_tokens`."""
        tokens = tokens.to(torch.int64)
        if tokens.device.type == "cpu" and tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= self.embed.shape[0]):
            raise IndexError("index out of range in self")  # torch.nn.Embedding's message (reference LUT conditioner)
        tok = tokens.to(self.device).contiguous()
        out = torch.empty((*tok.shape, self.D), dtype=torch.float32, device=self.device)
        self._pre()
        _lib.check(self.lib.ptts_embed_tokens(self.handle, _ptr(self.embed), int(self.embed.shape[0]), _ptr(tok), tok.numel(),
                                              _ptr(out), self._sp))
        tok.record_stream(self.stream)
        out.record_stream(self.stream)
        self._post()
        return out

    def lm_prefill(self, state: LMState, emb: torch.Tensor):
        """emb f32[B, T, D]: text embeddings or voice conditioning (reference tts_model.py:722-725,899)."""
        emb = emb.to(self.device, torch.float32).contiguous()
        if emb.dim() != 3 or emb.shape[0] != state.batch or emb.shape[2] != self.D:
            raise ValueError(f"prefill expects [B={state.batch}, T, {self.D}], got {tuple(emb.shape)}")
        self._pre()
        _lib.check(self.lib.ptts_lm_prefill(self.handle, state.handle, _ptr(emb), emb.shape[1], self._sp))
        emb.record_stream(self.stream)
        self._post()

    def lm_decode_step(self, state: LMState, latent_in=None, noise=None, lsd_steps: int = 1,
                       eos_threshold: float = -4.0, out_latent=None, out_logit=None, out_eos=None):
        """One autoregressive step (reference tts_model.py:758-760, flow_lm.py:96-139).  Asynchronous on
        the engine stream; outputs are device tensors (allocated if not given)."""
        B = state.batch
        dev = self.device
        out_latent = torch.empty((B, self.ldim), dtype=torch.float32, device=dev) if out_latent is None else out_latent
        out_logit = torch.empty((B,), dtype=torch.float32, device=dev) if out_logit is None else out_logit
        out_eos = torch.empty((B,), dtype=torch.uint8, device=dev) if out_eos is None else out_eos
        self._pre()
        _lib.check(self.lib.ptts_lm_decode_step(
            self.handle, state.handle, _ptr(latent_in), _ptr(noise), lsd_steps, eos_threshold,
            _ptr(out_latent), _ptr(out_logit), _ptr(out_eos), self._sp))
        for t in (latent_in, noise, out_latent, out_logit, out_eos):
            if t is not None:
                t.record_stream(self.stream)
        self._post()
        return out_latent, out_logit, out_eos

    def lm_latent(self, state: LMState) -> int:
        return self.lib.ptts_lm_latent_ptr(state.handle)

    def capture_lm_step(self, state: LMState, noise, lsd_steps, eos_threshold, out_latent, out_logit, out_eos):
        g = C.c_void_p()
        _lib.check(self.lib.ptts_graph_capture_lm_step(
            self.handle, state.handle, _ptr(noise), lsd_steps, eos_threshold, _ptr(out_latent), _ptr(out_logit),
            _ptr(out_eos), C.byref(g)))
        return g

    def capture_mimi(self, state: MimiState, latent, pcm: torch.Tensor):
        """latent: device tensor f32[B, ldim] or a raw device pointer (e.g. `lm_latent(state)`)."""
        g = C.c_void_p()
        lp = C.c_void_p(latent) if isinstance(latent, int) else _ptr(latent)
        _lib.check(self.lib.ptts_graph_capture_mimi(self.handle, state.handle, lp, _ptr(pcm), C.byref(g)))
        return g

    def graph_launch(self, g, stream: torch.cuda.Stream | None = None):
        sp = self._sp if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.lib.ptts_graph_launch(g, sp))

    def copy_to_host_async(self, host: torch.Tensor, dev: torch.Tensor, stream: torch.cuda.Stream | None = None):
        """device -> pinned host, truly asynchronous (torch's non_blocking copy_ blocks the host when the
        destination is a view of a pinned tensor on this stack)"""
        assert host.is_contiguous() and dev.is_contiguous() and host.numel() == dev.numel()
        sp = self._sp if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.lib.ptts_copy_to_host_async(self.handle, C.c_void_p(host.data_ptr()), _ptr(dev),
                                                    host.numel() * host.element_size(), sp))

    def graph_destroy(self, g):
        self.lib.ptts_graph_destroy(g)

    # ---- voice-prompt encode path
    def encode_voice(self, audio: torch.Tensor):
        """audio f32[n_samples] (mono, model sample rate) -> (latent [frames, ldim], conditioning [frames, D]):
        `MimiModel.encode_to_latent` + speaker projection (reference mimi.py:96-119, tts_model.py:379-388)."""
        audio = audio.reshape(-1).to(self.device, torch.float32).contiguous()
        n = audio.numel()
        frames = -(-n // self.frame_samples)
        lat = torch.empty((frames, self.ldim), dtype=torch.float32, device=self.device)
        cond = torch.empty((frames, self.D), dtype=torch.float32, device=self.device)
        self._pre()
        nf = C.c_int32()
        _lib.check(self.lib.ptts_encode_voice(self.handle, _ptr(audio), n, _ptr(lat), _ptr(cond), C.byref(nf), self._sp))
        assert nf.value == frames
        return lat, cond

    # ---- Mimi
    def mimi_decode(self, state: MimiState, latent: torch.Tensor, out_pcm=None) -> torch.Tensor:
        """latent f32[B, ldim] (normalised FlowLM output) -> pcm f32[B, frame_samples]."""
        B = state.batch
        if out_pcm is None:
            out_pcm = torch.empty((B, self.frame_samples), dtype=torch.float32, device=self.device)
        self._pre()
        _lib.check(self.lib.ptts_mimi_decode(self.handle, state.handle, _ptr(latent), _ptr(out_pcm), self._sp))
        latent.record_stream(self.stream)
        out_pcm.record_stream(self.stream)
        self._post()
        return out_pcm

    def debug_read(self, state, name: str) -> torch.Tensor:
        is_mimi = isinstance(state, MimiState)
        cap = 64 * 1024 * 1024
        buf = torch.empty((cap,), dtype=torch.float32, device=self.device)
        self._pre()
        r, c = C.c_int32(), C.c_int32()
        n = self.lib.ptts_debug_read(self.handle, state.handle, int(is_mimi), name.encode(), _ptr(buf), cap,
                                     C.byref(r), C.byref(c), self._sp)
        _lib.check(int(n))
        self.sync()
        return buf[: r.value * c.value].view(r.value, c.value).clone()

    def close(self):
        """Destroys every state created from this engine, then the engine (order matters: states
        point into the engine)."""
        if self.handle is not None:
            for st in list(self._states):
                st.close()
            self.lib.ptts_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StepPipeline:
    """hipGraph-per-step driver in which the FlowLM step of frame t+1 overlaps the codec decode of frame t
    (the reference pipelines the same two stages with two CPU threads and a queue:
    tts_model.py:651-658,741-742).  Latents ping-pong between two buffers; `pcm` buffers are pinned host
    memory, so the codec's last kernel writes the samples straight over PCIe.  Two modes:

    * "events" (throughput, default for batch > 8): FlowLM graphs on stream 1, codec graphs on stream 2,
      ordered by two events per step (codec t after step t; step t+2 after codec t); the host never waits.
    * "fork": one graph per step with two parallel branches {FlowLM step t} || {Mimi decode of frame t-1}.
      Measured slower than "events" on ROCm 7.2 (the branches of a replayed graph run back to back).
    * "hostsync" (latency, small batch): the host waits for FlowLM step t-1 (it needs its EOS flag anyway,
      like the reference's `.item()` at tts_model.py:761), then launches the codec graph of frame t-1 on a
      second stream while step t is already running on the first.  No cross-stream event wait sits on the
      critical path (on this stack such waits around graph launches cost ~80 us per step, and the branches
      of a forked graph do not run concurrently).
    """

    NB_EVENTS = int(os.environ.get("PTTS_PIPE_NB", "4"))  # output-buffer ring depth of the "events" mode

    def __init__(self, eng: Engine, lm_state: LMState, mimi_state: MimiState, noise=None, lsd_steps: int = 1,
                 eos_threshold: float = -4.0, mode: str | None = None, pcm_i16: bool = False,
                 lm_stream: torch.cuda.Stream | None = None):
        self.eng, self.st, self.ms = eng, lm_state, mimi_state
        B, dev = lm_state.batch, eng.device
        self.mode = mode or ("hostsync" if B <= 8 else "events")
        # ring of output buffers (latent -> codec input, EOS flags, PCM).  Throughput mode keeps 4 so that the FlowLM
        # stream may run up to 3 steps ahead of the codec stream (with 2 the two streams move in lock-step and every
        # hiccup of one stalls the other); the latency modes need only 2.  A host loop over the "events" mode must have
        # read flag[t % nb] / pcm[t % nb] of step t - nb before it calls step() for step t.
        self.nb = nb = self.NB_EVENTS if self.mode == "events" else 2
        self.lat = [torch.zeros(B, eng.ldim, device=dev) for _ in range(nb)]
        self.logit = [torch.empty(B, device=dev) for _ in range(nb)]
        self.flag = [torch.zeros(B, dtype=torch.uint8).pin_memory() for _ in range(nb)]  # EOS flags land on the host
        self.pcm = [torch.zeros(B, eng.frame_samples).pin_memory() for _ in range(nb)]
        self.ev = [torch.cuda.Event() for _ in range(nb)]    # codec frame (f % nb) complete -> pcm_of(f) valid
        self.ev_lm = [torch.cuda.Event() for _ in range(nb)]  # FlowLM step (t % nb) complete -> flag valid
        self.s1 = lm_stream or eng.stream  # FlowLM stream (several pipelines of one engine may use their own)
        self.s2 = eng.concurrent_stream(self.s1)
        eng.sync()
        torch.cuda.synchronize(dev)
        eng.tune(B)  # before capture: graphs freeze the tile choices
        lib, H = eng.lib, eng.handle
        self.g_first = [eng.capture_lm_step(lm_state, noise, lsd_steps, eos_threshold, self.lat[p], self.logit[p],
                                            self.flag[p]) for p in range(nb)]
        # optional 16-bit PCM beside the fp32 one (the WAV sample format, data/audio.py:79), also pinned
        self.pcm16 = [torch.zeros(B, eng.frame_samples, dtype=torch.int16).pin_memory() for _ in range(nb)] if pcm_i16 else None
        self.g_last = []
        for p in range(nb):
            mimi_state.set_pcm_i16(self.pcm16[p] if pcm_i16 else None)
            self.g_last.append(eng.capture_mimi(mimi_state, self.lat[p], self.pcm[p]))
        mimi_state.set_pcm_i16(None)
        self.g_both = []
        if self.mode == "fork":
            for p in range(nb):
                g = C.c_void_p()
                _lib.check(lib.ptts_graph_capture_pipelined(
                    H, lm_state.handle, mimi_state.handle, _ptr(noise), lsd_steps, eos_threshold, _ptr(self.lat[p]),
                    _ptr(self.logit[p]), _ptr(self.flag[p]), _ptr(self.lat[p ^ 1]), _ptr(self.pcm[p ^ 1]), C.byref(g)))
                self.g_both.append(g)
        for p in range(nb):
            self.ev[p].record(self.s2)
        self.t = 0          # FlowLM steps launched for the current utterances
        self.decoded = 0    # codec frames launched

    def restart(self):
        """new utterances: flush the pending frame, codec state back to zero carries"""
        self.flush()
        if self.mode != "fork":
            # zero carries on the CODEC stream: ordered behind the frames already queued there and ahead of the next
            # utterance's first frame, off the FlowLM stream's critical path (clone + prefill + first step)
            self.ms.reset(self.s2)
        else:
            self.ms.reset()
        self.t = 0
        self.decoded = 0

    def _decode_pending(self):
        """hostsync mode: wait for FlowLM step t-1 on the host, then start its codec frame on stream 2"""
        f = self.decoded
        q = f % self.nb
        self.ev_lm[q].synchronize()
        self.eng.graph_launch(self.g_last[q], self.s2)
        self.ev[q].record(self.s2)
        self.decoded += 1
        return f

    def step(self):
        """Launch FlowLM step t and the decode of frame t-1.  Returns the index of the frame whose PCM is
        complete when `done_event(frame)` fires (None for the first step).  After the call,
        `flag[(t-1) % nb]` (hostsync mode) holds the EOS flags of step t-1."""
        eng, p = self.eng, self.t % self.nb
        done = None
        if self.mode == "fork":
            if self.decoded < self.t:
                eng.graph_launch(self.g_both[p])
                self.ev[p ^ 1].record(eng.stream)
                self.decoded += 1
                done = self.decoded - 1
            else:
                eng.graph_launch(self.g_first[p])
        elif self.mode == "events":
            self.s1.wait_event(self.ev[p])             # codec frame t-nb done: lat[p] / pcm[p] are free
            eng.graph_launch(self.g_first[p], self.s1)  # FlowLM step t -> lat[p]
            self.ev_lm[p].record(self.s1)
            self.s2.wait_event(self.ev_lm[p])
            eng.graph_launch(self.g_last[p], self.s2)  # codec frame t -> pcm[p] (overlaps FlowLM step t+1)
            self.ev[p].record(self.s2)
            self.decoded += 1
            done = self.t
        else:
            self.s1.wait_event(self.ev[p])  # frame t-2 decoded: lat[p] / pcm[p] may be reused (long done)
            eng.graph_launch(self.g_first[p], self.s1)
            self.ev_lm[p].record(self.s1)
            if self.decoded < self.t:
                done = self._decode_pending()
        self.t += 1
        return done

    # ---- building blocks for a host-driven loop with an EOS decision per step (TTSModel) ----------
    def lm_step_async(self) -> int:
        """launch FlowLM step t on stream 1; returns t"""
        p = self.t % self.nb
        self.s1.wait_event(self.ev[p])  # frame t-2 decoded: lat[p] may be overwritten
        self.eng.graph_launch(self.g_first[p], self.s1)
        self.ev_lm[p].record(self.s1)
        self.t += 1
        return self.t - 1

    def wait_flags(self, step: int) -> torch.Tensor:
        """host waits for FlowLM step `step`; returns its EOS flags u8[B] (pinned host memory)"""
        self.ev_lm[step % self.nb].synchronize()
        return self.flag[step % self.nb]

    def decode_async(self, frame: int):
        """launch the codec decode of `frame` on stream 2 (call after wait_flags(frame))"""
        q = frame % self.nb
        self.eng.graph_launch(self.g_last[q], self.s2)
        self.ev[q].record(self.s2)
        self.decoded = frame + 1

    def flush(self):
        """decode the last pending frame (no FlowLM step rides along)"""
        if self.decoded >= self.t:
            return None
        if self.mode == "hostsync":
            return self._decode_pending()
        p = (self.t - 1) % self.nb
        self.eng.graph_launch(self.g_last[p])
        self.ev[p].record(self.eng.stream)
        self.decoded += 1
        return self.decoded - 1

    def pcm_of(self, frame: int) -> torch.Tensor:
        """host tensor [B, frame_samples] of `frame` (valid after `done_event(frame).synchronize()`)"""
        return self.pcm[frame % self.nb]

    def done_event(self, frame: int):
        """event that fires when the codec decode of `frame` (PCM in `pcm_of(frame)`) is complete"""
        return self.ev[frame % self.nb]

    def pcm16_of(self, frame: int) -> torch.Tensor:
        return self.pcm16[frame % self.nb]

    def sync(self):
        self.s1.synchronize()
        self.eng.stream.synchronize()
        self.s2.synchronize()

    def close(self):
        self.sync()
        for g in self.g_first + self.g_last + self.g_both:
            self.eng.graph_destroy(g)
