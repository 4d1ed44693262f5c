"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-numpy (float32) CPU restatement of the Pocket-TTS decode hot path:
the FlowLM autoregressive latent step and the Mimi/SEANet codec decode.  Every
function cites the reference file:line (relative to /root/reference) it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module, and only as the checker.  The product
(`pocket_tts_amd/`) never imports it and fails loudly when its HIP library is missing.

Pinning: this restatement is checked against golden vectors produced by running the
reference's own modules in the build container on the same synthetic weights
(`tests/golden/gen_golden.py` -> `tests/golden/*.npz`, test `tests/test_oracle_golden.py`).
Parity status: PINNED by those fixtures (fp32 tolerance stated in the test).
"""

from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


# --------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------
def gelu(x):
    """Exact-erf GELU, `F.gelu` default (reference `mimi_transformer.py:42`)."""
    return (x * F32(0.5) * (F32(1.0) + _erf(x * F32(1.0 / math.sqrt(2.0))))).astype(F32)


def silu(x):
    """`nn.SiLU` (reference `mlp.py:67,100,104,121`)."""
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def elu(x):
    """`nn.ELU(alpha=1.0)` (reference `seanet.py:26,156,170`)."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, F32(0.0)))).astype(F32)


def layer_norm(x, w, b, eps):
    """`nn.LayerNorm` / flow-MLP `LayerNorm`: biased variance
    (reference `mimi_transformer.py:26-27`, `mlp.py:49-55`)."""
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    y = xc / np.sqrt(var + F32(eps))
    if w is not None:
        y = y * w + b
    return y.astype(F32)


def rms_norm_var(x, alpha, eps):
    """The flow MLP's "RMSNorm": y = x * alpha * rsqrt(eps + var_unbiased(x)); the mean is
    removed inside the variance only (reference `mlp.py:20-25`)."""
    var = x.var(axis=-1, keepdims=True, ddof=1, dtype=F32) + F32(eps)
    return (x * (alpha / np.sqrt(var))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32)


# --------------------------------------------------------------------------
# RoPE + streaming attention
# --------------------------------------------------------------------------
def apply_rope(q, k, offset, max_period=10000.0):
    """Interleaved-pair rotary embedding in fp32 (reference `rope.py:7-58`).
    q, k: [B, T, H, D]; offset: scalar position of the first row."""
    B, T, H, D = q.shape
    ds = np.arange(D // 2, dtype=F32)
    freqs = np.exp(ds * F32(-math.log(max_period) * 2 / D)).astype(F32)
    ts = (np.arange(T, dtype=F32) + F32(offset)).reshape(-1, 1, 1)
    ang = (freqs * ts).astype(F32)  # [T,1,D/2]
    rotr, roti = np.cos(ang).astype(F32), np.sin(ang).astype(F32)

    def rot(x):
        x = x.reshape(B, T, x.shape[2], D // 2, 2)
        xr, xi = x[..., 0], x[..., 1]
        o = np.stack([xr * rotr - xi * roti, xr * roti + xi * rotr], axis=-1)
        return o.reshape(B, T, -1, D).astype(F32)

    return rot(q), rot(k)


def streaming_attention(x, state, wqkv, wo, num_heads, context, max_period):
    """`StreamingMultiheadAttention.forward` with the linear KV cache
    (reference `transformer.py:135-158`, cache `transformer.py:9-19,39-84`,
    mask `transformer.py:22-29`).  state = {"cache": [2,B,Tcap,H,D], "offset": int}."""
    B, T, C = x.shape
    proj = linear(x, wqkv)
    return linear(attention_core(proj, state, num_heads, context, max_period), wo)


def attention_core(proj, state, num_heads, context, max_period):
    """the part of `StreamingMultiheadAttention.forward` between the packed in_proj and out_proj
    (reference `transformer.py:138-157`): proj [B, T, 3C] -> attention output [B, T, C]"""
    B, T, C3 = proj.shape
    C = C3 // 3
    D = C // num_heads
    proj = proj.reshape(B, T, 3, num_heads, D)
    q, k, v = proj[:, :, 0], proj[:, :, 1], proj[:, :, 2]
    off = int(state["offset"])
    q, k = apply_rope(q, k, off, max_period)
    cache = state["cache"]
    if off + T > cache.shape[2]:
        raise ValueError("KV cache capacity exceeded")
    cache[0, :, off : off + T] = k
    cache[1, :, off : off + T] = v
    K = cache[0, :, : off + T].transpose(0, 2, 1, 3)  # [B,H,Tk,D]
    V = cache[1, :, : off + T].transpose(0, 2, 1, 3)
    Q = q.transpose(0, 2, 1, 3)
    pos_q = off + np.arange(T)
    pos_k = np.arange(off + T)
    delta = pos_q[:, None] - pos_k[None, :]
    mask = delta >= 0
    if context is not None:
        mask &= delta < context
    s = (Q @ K.transpose(0, 1, 3, 2)) * F32(1.0 / math.sqrt(D))
    s = np.where(mask[None, None], s, F32(-np.inf)).astype(F32)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s).astype(F32)
    p = p / p.sum(axis=-1, keepdims=True, dtype=F32)
    # masked-out keys may hold NaN (cache is NaN-initialised, `transformer.py:52-57`)
    Vz = np.where(mask.any(axis=0)[None, None, :, None], V, F32(0.0))
    return (p @ Vz).astype(F32).transpose(0, 2, 1, 3).reshape(B, T, C)


def transformer_layer(x, state, W, p, num_heads, context, max_period, taps=None):
    """Pre-LN block with optional LayerScale (reference `mimi_transformer.py:39-54`)."""
    h = layer_norm(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-5)
    a = streaming_attention(
        h, state, W[p + ".self_attn.in_proj.weight"], W[p + ".self_attn.out_proj.weight"],
        num_heads, context, max_period,
    )
    ls1 = W.get(p + ".layer_scale_1.scale")
    x = x + (a if ls1 is None else ls1 * a)
    if taps is not None:
        taps[p + ":attn_res"] = x.copy()
    h = layer_norm(x, W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-5)
    f = linear(gelu(linear(h, W[p + ".linear1.weight"])), W[p + ".linear2.weight"])
    ls2 = W.get(p + ".layer_scale_2.scale")
    x = (x + (f if ls2 is None else ls2 * f)).astype(F32)
    if taps is not None:
        taps[p + ":out"] = x.copy()
    return x


# --------------------------------------------------------------------------
# FlowLM
# --------------------------------------------------------------------------
class FlowLM:
    """`FlowLMModel` inference path (reference `flow_lm.py:96-157`) driven the way
    `TTSModel._run_flow_lm_and_increment_step` does (reference `tts_model.py:317-367`)."""

    def __init__(self, cfg, W):
        self.cfg = cfg
        self.W = W
        t = cfg.flow_lm.transformer
        self.D, self.H, self.L = t.d_model, t.num_heads, t.num_layers
        self.max_period = float(t.max_period)
        self.ldim = cfg.mimi.quantizer.dimension
        self.fd, self.depth = cfg.flow_lm.flow.dim, cfg.flow_lm.flow.depth

    # `init_states` (reference `stateful_module.py:7-16`, `transformer.py:46-57`)
    def init_state(self, B, T):
        return [
            dict(cache=np.full((2, B, T, self.H, self.D // self.H), np.nan, F32), offset=0)
            for _ in range(self.L)
        ]

    def embed_text(self, tokens):
        """`LUTConditioner._get_condition` (reference `text.py:74-76`)."""
        return self.W["flow_lm.conditioner.embed.weight"][tokens].astype(F32)

    def backbone(self, state, text_emb, seq, taps=None):
        """`FlowLMModel.backbone` + BOS substitution + input_linear
        (reference `flow_lm.py:121-122,141-157`); increments offsets like
        `increment_steps` (reference `stateful_module.py:19-26`)."""
        W = self.W
        seq = np.where(np.isnan(seq), W["flow_lm.bos_emb"], seq).astype(F32)
        x = linear(seq, W["flow_lm.input_linear.weight"])
        x = np.concatenate([text_emb.astype(F32), x], axis=1)
        T = x.shape[1]
        for i in range(self.L):
            x = transformer_layer(
                x, state[i], W, f"flow_lm.transformer.layers.{i}", self.H, None,
                self.max_period, taps,
            )
        for st in state:
            st["offset"] += T
        return layer_norm(x, W["flow_lm.out_norm.weight"], W["flow_lm.out_norm.bias"], 1e-5)

    def prefill(self, state, emb):
        """Text or voice conditioning enters as `text_embeddings` with an empty latent
        sequence (reference `tts_model.py:722-725,899`); outputs are discarded."""
        B = emb.shape[0]
        self.backbone(state, emb, np.zeros((B, 0, self.ldim), F32))

    # ---- flow head ---------------------------------------------------------
    def time_embed(self, i, t):
        """`TimestepEmbedder.forward` (reference `mlp.py:79-83`); t: [B,1]."""
        W, p = self.W, f"flow_lm.flow_net.time_embed.{i}."
        args = (t * W[p + "freqs"]).astype(F32)
        e = np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(F32)
        h = silu(linear(e, W[p + "mlp.0.weight"], W[p + "mlp.0.bias"]))
        h = linear(h, W[p + "mlp.2.weight"], W[p + "mlp.2.bias"])
        return rms_norm_var(h, W[p + "mlp.3.alpha"], 1e-5)

    def flow_net(self, c, s, t, x, taps=None):
        """`SimpleMLPAdaLN.forward` (reference `mlp.py:188-215`; ResBlock `mlp.py:107-111`,
        FinalLayer `mlp.py:127-131`)."""
        W, p = self.W, "flow_lm.flow_net."
        x = linear(x, W[p + "input_proj.weight"], W[p + "input_proj.bias"])
        t_comb = ((self.time_embed(0, s) + self.time_embed(1, t)) / F32(2)).astype(F32)
        y = t_comb + linear(c, W[p + "cond_embed.weight"], W[p + "cond_embed.bias"])
        sy = silu(y)
        for i in range(self.depth):
            r = f"{p}res_blocks.{i}."
            mod = linear(sy, W[r + "adaLN_modulation.1.weight"], W[r + "adaLN_modulation.1.bias"])
            shift, scale, gate = np.split(mod, 3, axis=-1)
            h = layer_norm(x, W[r + "in_ln.weight"], W[r + "in_ln.bias"], 1e-6)
            h = h * (F32(1) + scale) + shift
            h = silu(linear(h, W[r + "mlp.0.weight"], W[r + "mlp.0.bias"]))
            h = linear(h, W[r + "mlp.2.weight"], W[r + "mlp.2.bias"])
            x = (x + gate * h).astype(F32)
            if taps is not None:
                taps[f"flow_res{i}"] = x.copy()
        r = p + "final_layer."
        mod = linear(sy, W[r + "adaLN_modulation.1.weight"], W[r + "adaLN_modulation.1.bias"])
        shift, scale = np.split(mod, 2, axis=-1)
        h = layer_norm(x, None, None, 1e-6) * (F32(1) + scale) + shift
        return linear(h.astype(F32), W[r + "linear.weight"], W[r + "linear.bias"])

    def decode_step(self, state, latent_in, noise=None, lsd_steps=1, eos_threshold=-4.0, taps=None):
        """One autoregressive step: `_sample_next_latent` on a [B,1,ldim] input
        (reference `flow_lm.py:96-139`, driver `tts_model.py:756-760`).
        latent_in: [B, ldim] (NaN rows = BOS).  noise: [B, ldim] or None (= temp 0, zeros).
        Returns (next_latent [B,ldim], eos_logit [B], is_eos [B] bool)."""
        B = latent_in.shape[0]
        out = self.backbone(
            state, np.zeros((B, 0, self.D), F32), latent_in.reshape(B, 1, self.ldim), taps
        )
        c = out[:, -1]
        if taps is not None:
            taps["cond"] = c.copy()
        W = self.W
        eos_logit = linear(c, W["flow_lm.out_eos.weight"], W["flow_lm.out_eos.bias"])[:, 0]
        is_eos = eos_logit > F32(eos_threshold)
        cur = np.zeros((B, self.ldim), F32) if noise is None else noise.astype(F32).copy()
        # `lsd_decode` (reference `flow_lm.py:19-40`)
        for i in range(lsd_steps):
            s = np.full((B, 1), i / lsd_steps, F32)
            t = np.full((B, 1), (i + 1) / lsd_steps, F32)
            cur = (cur + self.flow_net(c, s, t, cur, taps) / F32(lsd_steps)).astype(F32)
        return cur, eos_logit, is_eos


# --------------------------------------------------------------------------
# streaming convolutions (channel-first [B, C, T], like the reference)
# --------------------------------------------------------------------------
def conv1d(x, w, b):
    """`nn.Conv1d`, stride 1, no padding: w [O, C, K]."""
    K = w.shape[2]
    To = x.shape[2] - K + 1
    cols = np.stack([x[:, :, k : k + To] for k in range(K)], axis=2)  # [B,C,K,To]
    y = np.einsum("ock,bckt->bot", w, cols, optimize=True)
    if b is not None:
        y = y + b[None, :, None]
    return y.astype(F32)


def conv_transpose1d(x, w, b, stride):
    """`nn.ConvTranspose1d`, groups=1: w [C, O, K] -> length (T-1)*stride + K."""
    B, C, T = x.shape
    O, K = w.shape[1], w.shape[2]
    y = np.zeros((B, O, (T - 1) * stride + K), F32)
    contrib = np.einsum("bct,cok->botk", x, w, optimize=True).astype(F32)
    for t in range(T):
        y[:, :, t * stride : t * stride + K] += contrib[:, :, t]
    if b is not None:
        y = y + b[None, :, None]
    return y.astype(F32)


def streaming_conv1d(x, w, b, st):
    """`StreamingConv1d.forward`, pad_mode "constant", stride 1
    (reference `conv.py:93-115`); st = {"previous": [B,C,K-1]}."""
    TP = st["previous"].shape[-1]
    if TP:
        x = np.concatenate([st["previous"], x], axis=-1)
    y = conv1d(x, w, b)
    if TP:
        st["previous"] = x[..., -TP:].copy()
    return y


def streaming_conv_transpose1d(x, w, b, stride, st):
    """`StreamingConvTranspose1d.forward` (reference `conv.py:151-163`);
    st = {"partial": [B,O,K-stride]} stored without bias."""
    y = conv_transpose1d(x, w, b, stride)
    PT = st["partial"].shape[-1]
    if PT > 0:
        y[..., :PT] += st["partial"]
        tail = y[..., -PT:].copy()
        if b is not None:
            tail -= b[None, :, None]
        st["partial"] = tail
        y = y[..., :-PT]
    return y


# --------------------------------------------------------------------------
# Mimi decode
# --------------------------------------------------------------------------
class MimiDecoder:
    """`_decode_audio_worker` body + `MimiModel.decode_from_latent`
    (reference `tts_model.py:449-455`, `mimi.py:89-94`)."""

    def __init__(self, cfg, W):
        from pocket_tts_amd.weights import seanet_decoder_layers  # inventory only, no compute

        self.cfg, self.W = cfg, W
        self.layers = seanet_decoder_layers(cfg)
        self.stride = cfg.upsample_stride
        tr = cfg.mimi.transformer
        self.tr = tr

    def init_state(self, B, max_frames):
        """`init_states(mimi, B, seq_len)` for the decode-side modules
        (reference `conv.py:84-91,145-149`, `transformer.py:46-57`)."""
        W, st = self.W, {}
        C = self.cfg.mimi.seanet.dimension
        st["upsample"] = dict(partial=np.zeros((B, C, self.stride), F32))
        tr = self.tr
        Dh = tr.d_model // tr.num_heads
        st["attn"] = [
            dict(cache=np.full((2, B, max_frames * self.stride, tr.num_heads, Dh), np.nan, F32), offset=0)
            for _ in range(tr.num_layers)
        ]
        for idx, kind, cin, cout, k, stride in self.layers:
            if kind == "conv":
                st[idx] = dict(previous=np.zeros((B, cin, k - 1), F32))
            elif kind == "convtr":
                st[idx] = dict(partial=np.zeros((B, cout, k - stride), F32))
            else:
                st[idx] = dict(previous=np.zeros((B, cin, k - 1), F32))
        return st

    def decode(self, st, latent, taps=None):
        """latent: [B, ldim] (normalised FlowLM output) -> pcm [B, frame_samples]."""
        W = self.W
        B = latent.shape[0]
        x = (latent * W["flow_lm.emb_std"] + W["flow_lm.emb_mean"]).astype(F32)  # tts_model.py:449
        x = linear(x, W["mimi.quantizer.output_proj.weight"][:, :, 0])  # dummy_quantizer.py:17-18
        x = x[:, :, None]  # [B, C, 1]
        # ConvTrUpsample1d: depthwise convtr k=2*stride (resample.py:40-51)
        wu = W["mimi.upsample.convtr.convtr.weight"][:, 0, :]  # [C, 2s]
        y = (x * wu[None]).astype(F32)  # T_in = 1 -> [B, C, 2s]
        s = self.stride
        y[..., :s] += st["upsample"]["partial"]
        st["upsample"]["partial"] = y[..., s:].copy()
        x = y[..., :s]
        if taps is not None:
            taps["upsample"] = x.copy()
        # ProjectedTransformer (mimi_transformer.py:140-150): [B,C,T] -> [B,T,C] -> layers
        h = x.transpose(0, 2, 1)
        tr = self.tr
        for i in range(tr.num_layers):
            h = transformer_layer(
                h, st["attn"][i], W, f"mimi.decoder_transformer.transformer.layers.{i}",
                tr.num_heads, tr.context, float(tr.max_period), None,
            )
        for a in st["attn"]:
            a["offset"] += s  # increment_steps(mimi, state, 16) tts_model.py:455
        x = h.transpose(0, 2, 1).astype(F32)
        if taps is not None:
            taps["dec_tr"] = x.copy()
        # SEANetDecoder (seanet.py:141-180)
        for n, (idx, kind, cin, cout, k, stride) in enumerate(self.layers):
            p = f"mimi.decoder.model.{idx}"
            if kind == "conv":
                if n > 0:
                    x = elu(x)
                x = streaming_conv1d(x, W[p + ".conv.weight"], W[p + ".conv.bias"], st[idx])
            elif kind == "convtr":
                x = streaming_conv_transpose1d(
                    elu(x), W[p + ".convtr.weight"], W[p + ".convtr.bias"], stride, st[idx]
                )
            else:  # SEANetResnetBlock (seanet.py:33-41)
                v = streaming_conv1d(
                    elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"], st[idx]
                )
                v = conv1d(elu(v), W[p + ".block.3.conv.weight"], W[p + ".block.3.conv.bias"])
                x = (x + v).astype(F32)
            if taps is not None:
                taps[f"seanet{idx}"] = x.copy()
        return x[:, 0, :]


# --------------------------------------------------------------------------
# voice-prompt encode path (one-off per voice)
# --------------------------------------------------------------------------
def conv1d_strided(x, w, b, stride):
    """`nn.Conv1d` with stride, no padding: w [O, C, K]."""
    K = w.shape[2]
    To = (x.shape[2] - K) // stride + 1
    cols = np.stack([x[:, :, k : k + (To - 1) * stride + 1 : stride] for k in range(K)], axis=2)
    y = np.einsum("ock,bckt->bot", w, cols, optimize=True)
    if b is not None:
        y = y + b[None, :, None]
    return y.astype(F32)


def causal_conv_full(x, w, b, stride=1, replicate=False):
    """`StreamingConv1d.forward` with `model_state=None` on a whole signal: left context of
    kernel - stride samples, zeros ("constant") or the first sample ("replicate")
    (reference `conv.py:84-115`, `resample.py:18-26`)."""
    TP = w.shape[2] - stride
    if TP:
        pad = np.repeat(x[..., :1], TP, axis=-1) if replicate else np.zeros(x.shape[:2] + (TP,), F32)
        x = np.concatenate([pad, x], axis=-1)
    return conv1d_strided(x, w, b, stride)


def full_attention_layer_stack(h, W, prefix, num_layers, num_heads, context, max_period):
    """`ProjectedTransformer` with `model_state=None`: whole sequence at once, RoPE offset 0, causal +
    sliding-window mask (reference `transformer.py:63-75,135-158`, `mimi_transformer.py:140-150`)."""
    B, T, C = h.shape
    for i in range(num_layers):
        st = dict(cache=np.full((2, B, T, num_heads, C // num_heads), np.nan, F32), offset=0)
        h = transformer_layer(h, st, W, f"{prefix}.{i}", num_heads, context, max_period, None)
    return h


class VoiceEncoder:
    """`MimiModel.encode_to_latent` + `TTSModel._encode_audio`
    (reference `mimi.py:96-119`, `tts_model.py:379-388`)."""

    def __init__(self, cfg, W):
        from pocket_tts_amd.weights import seanet_encoder_layers  # inventory only

        self.cfg, self.W = cfg, W
        self.layers = seanet_encoder_layers(cfg)

    def encode_to_latent(self, audio, taps=None):
        """audio [B, 1, T] -> latent [B, inner_dim, ceil(T / frame_samples)]"""
        W, cfg = self.W, self.cfg
        fs = cfg.frame_samples
        T = audio.shape[-1]
        pad = (-T) % fs  # pad_for_conv1d(x, frame_size, frame_size): zeros at the end (conv.py:22-33)
        x = np.concatenate([audio.astype(F32), np.zeros(audio.shape[:2] + (pad,), F32)], axis=-1)
        for n, (idx, kind, cin, cout, k, stride) in enumerate(self.layers):
            p = f"mimi.encoder.model.{idx}"
            if kind == "res":
                v = causal_conv_full(elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"])
                v = causal_conv_full(elu(v), W[p + ".block.3.conv.weight"], W[p + ".block.3.conv.bias"])
                x = (x + v).astype(F32)
            else:
                if n > 0:
                    x = elu(x)
                x = causal_conv_full(x, W[p + ".conv.weight"], W[p + ".conv.bias"], stride)
            if taps is not None:
                taps[f"enc{idx}"] = x.copy()
        tr = cfg.mimi.transformer
        h = full_attention_layer_stack(x.transpose(0, 2, 1), W, "mimi.encoder_transformer.transformer.layers",
                                       tr.num_layers, tr.num_heads, tr.context, float(tr.max_period))
        x = h.transpose(0, 2, 1).astype(F32)
        if taps is not None:
            taps["enc_tr"] = x.copy()
        # ConvDownsample1d: kernel 2*stride, stride, replicate padding, no bias (resample.py:7-29)
        return causal_conv_full(x, W["mimi.downsample.conv.conv.weight"], None, cfg.upsample_stride, replicate=True)

    def conditioning(self, audio):
        """-> [B, frames, d_model] = latents^T @ speaker_proj_weight^T (tts_model.py:386-388)"""
        lat = self.encode_to_latent(audio)
        return linear(lat.transpose(0, 2, 1), self.W["flow_lm.speaker_proj_weight"])


# --------------------------------------------------------------------------
# the two hot loops (reference `tts_model.py:744-779` and `:433-474`)
# --------------------------------------------------------------------------
def autoregressive_generation(lm, state, max_gen_len, frames_after_eos, noise=None,
                              lsd_steps=1, eos_threshold=-4.0, B=1):
    """Hot loop 1.  Returns (latents [n, B, ldim], eos_logits [steps, B], eos_step).
    Batch rows share one EOS decision taken on row 0, as the reference's `.item()` does
    for B=1 (`tts_model.py:761`)."""
    x = np.full((B, lm.ldim), np.nan, F32)
    lat, logits, eos_step = [], [], None
    for step in range(max_gen_len):
        nz = None if noise is None else noise[step]
        x, logit, is_eos = lm.decode_step(state, x, nz, lsd_steps, eos_threshold)
        logits.append(logit)
        if bool(is_eos[0]) and eos_step is None:
            eos_step = step
        if eos_step is not None and step >= eos_step + frames_after_eos:
            break
        lat.append(x.copy())
    return np.stack(lat) if lat else np.zeros((0, B, lm.ldim), F32), np.stack(logits), eos_step


# --------------------------------------------------------------------------------------------------
# int8 weight path (BASELINE config #5).  The reference quantises the FlowLM transformer's Linear layers with
# torch.ao / torchao *dynamic* int8 (quantization.py:60-128: groups "attention" = self_attn.in_proj/out_proj,
# "ffn" = linear1/linear2; load_model(quantize=True) uses both, tts_model.py:312-315).  Those CPU kernels
# (FBGEMM / QNNPACK / torchao) are third-party code outside /root/reference and also quantise the activations;
# the build's GPU path is weight-only: int8 per output channel, symmetric, fp32 activations and accumulation.
# This function restates THAT scheme (parity unpinned against the reference's int8 arithmetic; the GPU tests
# report SNR against the fp32 model next to it, the reference's own quality metric,
# scripts/evaluate_quantization.py:215-228).
QUANT_GROUP_SUFFIXES = {
    "attention": ("self_attn.in_proj.weight", "self_attn.out_proj.weight"),
    "ffn": ("linear1.weight", "linear2.weight"),
}


def quantize_dequantize_int8(w):
    """per-row symmetric int8: scale = max|w| / 127, q = rint(w / scale) clipped to [-127, 127]; returns q * scale"""
    w = np.asarray(w, np.float32)
    mx = np.abs(w).max(axis=1, keepdims=True).astype(np.float32)
    scale = np.where(mx > 0, mx / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(w / scale), -127, 127).astype(np.float32)
    return (q * scale).astype(np.float32)


def quantized_weights(W, groups=("attention", "ffn")):
    """copy of the weight dict with the FlowLM transformer's Linear weights of `groups` replaced by their
    int8-dequantised values; everything else (flow net, Mimi, embeddings, norms) stays fp32"""
    out = dict(W)
    for g in groups:
        for name in W:
            if name.startswith("flow_lm.transformer.layers.") and name.endswith(QUANT_GROUP_SUFFIXES[g]):
                out[name] = quantize_dequantize_int8(W[name])
    return out


# --------------------------------------------------------------------------------------------------
# bf16 codec (BASELINE config #5, second half): a ROUNDING MODEL of the build's reduced-precision Mimi decoder, so that
# the HIP bf16 path is checked against the oracle and not only against the build's own fp32 path.
# The reference has no bf16 / quantised Mimi at all (docs/quantization.md:67-76), so there is nothing to pin this to:
# parity of the bf16 path with the REFERENCE stays "unpinned".  What this model states is WHERE the build rounds:
#   * every GEMM / conv weight to bf16 once (a LayerNorm that precedes a Linear has its gain multiplied in first and
#     the fold vector s[n] is summed from the ROUNDED matrix; the constant c[n] = W beta + bias comes from the fp32 one);
#   * every activation buffer the codec writes to HBM to bf16 at the producer (after bias / activation / residual):
#     upsample output, residual stream after out_proj and after linear2, GELU(linear1), attention output, every SEANet
#     conv output (the ELU'd copy and the raw skip copy are rounded separately);
#   * q, k, v, the KV ring, the softmax, all accumulation and all epilogue arithmetic stay fp32; the last conv
#     (n_filters -> 1 sample) uses the fp32 checkpoint weights on the bf16 activation.
# Same algorithm as MimiDecoder.decode otherwise (reference `tts_model.py:449-455`, `mimi.py:89-94`,
# `mimi_transformer.py:39-54`, `seanet.py:141-180`, `conv.py:93-163`).
def bf16_round(x):
    """round-to-nearest-even to bfloat16, returned as float32 (v_cvt_pk_bf16_f32)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def elu_fast(x):
    """the build's epilogue ELU: x > 0 ? x : exp(x) - 1 in fp32 (absolute error ~1e-7 near 0, see ptts_kernels.h)"""
    return np.where(x > 0, x, np.exp(np.minimum(x, F32(0.0))) - F32(1.0)).astype(F32)


def lnfold_linear_bf16(x, w, g, beta, bias, eps, stats_from=None):
    """LayerNorm folded into the following Linear, bf16 weights (ptts_bf16.h / gemm_kernel<.., WF = 2> PRE_LNFOLD): x holds
    bf16 values; `stats_from`: the fp32 tensor the row statistics are taken from when the operand is rounded on load"""
    wr = bf16_round(w * g[None, :])
    s = wr.sum(axis=1, dtype=F32)
    c = (w @ beta + (bias if bias is not None else F32(0))).astype(F32)
    K = x.shape[-1]
    xs = x if stats_from is None else stats_from
    mu = xs.sum(axis=-1, dtype=F32) / F32(K)
    var = np.maximum((xs * xs).sum(axis=-1, dtype=F32) / F32(K) - mu * mu, F32(0))
    rs = F32(1.0) / np.sqrt(var + F32(eps))
    return (((x @ wr.T) - mu[..., None] * s) * rs[..., None] + c).astype(F32)


class MimiDecoderBF16(MimiDecoder):
    def decode(self, st, latent, taps=None):
        W = self.W
        x = (latent * W["flow_lm.emb_std"] + W["flow_lm.emb_mean"]).astype(F32)
        x = linear(x, W["mimi.quantizer.output_proj.weight"][:, :, 0])[:, :, None]   # fp32 (kept per frame in fp32)
        wu = W["mimi.upsample.convtr.convtr.weight"][:, 0, :]
        y = (x * wu[None]).astype(F32)
        s = self.stride
        y[..., :s] += st["upsample"]["partial"]
        st["upsample"]["partial"] = y[..., s:].copy()
        h = bf16_round(y[..., :s]).transpose(0, 2, 1)                                # [B, T, C], bf16 values
        tr = self.tr
        for i in range(tr.num_layers):
            p = f"mimi.decoder_transformer.transformer.layers.{i}"
            proj = lnfold_linear_bf16(h, W[p + ".self_attn.in_proj.weight"], W[p + ".norm1.weight"], W[p + ".norm1.bias"], None, 1e-5)
            ao = bf16_round(attention_core(proj, st["attn"][i], tr.num_heads, tr.context, float(tr.max_period)))
            a = linear(ao, bf16_round(W[p + ".self_attn.out_proj.weight"]))
            h = bf16_round(h + W[p + ".layer_scale_1.scale"] * a)
            f = lnfold_linear_bf16(h, W[p + ".linear1.weight"], W[p + ".norm2.weight"], W[p + ".norm2.bias"], None, 1e-5)
            f = linear(bf16_round(gelu(f)), bf16_round(W[p + ".linear2.weight"]))
            h = bf16_round(h + W[p + ".layer_scale_2.scale"] * f)
        for a_ in st["attn"]:
            a_["offset"] += s
        x = h.transpose(0, 2, 1).astype(F32)
        # SEANet: `x` is always the bf16 buffer the next conv reads (ELU already applied by its producer), `raw` the
        # bf16 copy of the un-activated value a residual block adds back
        raw = None
        last = len(self.layers) - 1
        for n, (idx, kind, cin, cout, k, stride) in enumerate(self.layers):
            p = f"mimi.decoder.model.{idx}"
            if kind == "conv" and n == last:
                w = W[p + ".conv.weight"]                                         # fp32 weights on the bf16 activation
                x = streaming_conv1d(x, w, W[p + ".conv.bias"], st[idx])
            elif kind == "conv":
                v = streaming_conv1d(x, bf16_round(W[p + ".conv.weight"]), W[p + ".conv.bias"], st[idx])
                x = bf16_round(elu_fast(v))
            elif kind == "convtr":
                v = streaming_conv_transpose1d(x, bf16_round(W[p + ".convtr.weight"]), W[p + ".convtr.bias"], stride, st[idx])
                raw, x = bf16_round(v), bf16_round(elu_fast(v))
            else:
                v = streaming_conv1d(x, bf16_round(W[p + ".block.1.conv.weight"]), W[p + ".block.1.conv.bias"], st[idx])
                v = conv1d(bf16_round(elu_fast(v)), bf16_round(W[p + ".block.3.conv.weight"]), W[p + ".block.3.conv.bias"])
                x = bf16_round(elu_fast(raw + v))
            if taps is not None:
                taps[f"seanet{idx}"] = x.copy()
        return x[:, 0, :]


# --------------------------------------------------------------------------------------------------
# bf16 LM weights (PTTS_LM_BF16): rounding model of the build's FlowLM with bf16 Linear weights.  The four
# Linear layers of every transformer layer (in_proj, out_proj, linear1, linear2: the reference's quantisation groups,
# quantization.py:21,91-128) multiply bf16 weights with activation operands rounded to bf16 ON LOAD; the residual stream,
# the LayerNorm statistics (taken from the fp32 values), q / k / v, the KV cache, attention, GELU, accumulation and the
# whole flow head stay fp32.  No reference counterpart (parity with the reference unpinned); pins the HIP path to this
# stated arithmetic.
class FlowLMBF16(FlowLM):
    def backbone(self, state, text_emb, seq, taps=None):
        W = self.W
        seq = np.where(np.isnan(seq), W["flow_lm.bos_emb"], seq).astype(F32)
        x = linear(seq, W["flow_lm.input_linear.weight"])
        x = np.concatenate([text_emb.astype(F32), x], axis=1)
        T = x.shape[1]
        for i in range(self.L):
            p = f"flow_lm.transformer.layers.{i}"
            proj = lnfold_linear_bf16(bf16_round(x), W[p + ".self_attn.in_proj.weight"], W[p + ".norm1.weight"],
                                      W[p + ".norm1.bias"], None, 1e-5, stats_from=x)
            ao = attention_core(proj, state[i], self.H, None, self.max_period)
            x = (x + linear(bf16_round(ao), bf16_round(W[p + ".self_attn.out_proj.weight"]))).astype(F32)
            f = lnfold_linear_bf16(bf16_round(x), W[p + ".linear1.weight"], W[p + ".norm2.weight"], W[p + ".norm2.bias"],
                                   None, 1e-5, stats_from=x)
            x = (x + linear(bf16_round(gelu(f)), bf16_round(W[p + ".linear2.weight"]))).astype(F32)
        for st in state:
            st["offset"] += T
        return layer_norm(x, W["flow_lm.out_norm.weight"], W["flow_lm.out_norm.bias"], 1e-5)
