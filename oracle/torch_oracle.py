"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The same CPU restatement of the Pocket-TTS decode hot path as `oracle/np_oracle.py`, driven through STOCK
PyTorch CPU operators (`F.linear`, `F.layer_norm`, `F.scaled_dot_product_attention`, `F.conv1d`,
`F.conv_transpose1d`, ...) - the operators the reference itself runs on ("all arithmetic is
delegated to PyTorch CPU ATen kernels").  It exists for `bench.py`'s `cpu_baseline` leg ("the build's
CPU restatement driven through stock PyTorch CPU ops"), so that the CPU figure beside the GPU one is not a numpy
strawman, and for tests.  Each function cites the reference file:line (relative to /root/reference) it follows.

Only `tests/` and `bench.py`'s `cpu_baseline` may import this module.  Pinning: `tests/test_oracle_golden.py` checks it
against the same reference-generated golden vectors as the numpy oracle (fp32 tolerance stated there).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _t(W, name):
    v = W[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


class _Weights:
    """checkpoint dict -> torch CPU tensors (converted once)"""

    def __init__(self, W):
        self.W = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)).float() for k, v in W.items()}

    def __getitem__(self, k):
        return self.W[k]

    def get(self, k):
        return self.W.get(k)


def apply_rope(q, k, offset, max_period):
    """Interleaved-pair rotary embedding in fp32 (reference `rope.py:7-58`).  q, k: [B, T, H, D]."""
    B, T, H, D = q.shape
    ds = torch.arange(D // 2, dtype=torch.float32)
    freqs = torch.exp(ds * (-math.log(max_period) * 2 / D))
    ts = (torch.arange(T, dtype=torch.float32) + float(offset)).view(-1, 1, 1)
    rotr, roti = torch.cos(freqs * ts), torch.sin(freqs * ts)  # [T, 1, D/2]

    def rot(x):
        x = x.view(B, T, H, D // 2, 2)
        xr, xi = x[..., 0], x[..., 1]
        return torch.stack([xr * rotr - xi * roti, xr * roti + xi * rotr], dim=-1).view(B, T, H, D)

    return rot(q), rot(k)


def streaming_attention(x, state, wqkv, wo, num_heads, context, max_period):
    """`StreamingMultiheadAttention.forward` on the linear KV cache (reference `transformer.py:135-158`, cache
    `:9-19,39-84`, mask `:22-29`).  state = {"cache": [2, B, Tcap, H, D], "offset": int}."""
    B, T, C = x.shape
    D = C // num_heads
    proj = F.linear(x, wqkv).view(B, T, 3, num_heads, D)
    q, k, v = proj[:, :, 0], proj[:, :, 1], proj[:, :, 2]
    off = int(state["offset"])
    q, k = apply_rope(q, k, off, max_period)
    cache = state["cache"]
    if off + T > cache.shape[2]:
        raise ValueError("KV cache capacity exceeded")
    cache[0, :, off:off + T] = k
    cache[1, :, off:off + T] = v
    K = cache[0, :, :off + T].transpose(1, 2)
    V = cache[1, :, :off + T].transpose(1, 2)
    pos_q = off + torch.arange(T)
    pos_k = torch.arange(off + T)
    delta = pos_q[:, None] - pos_k[None, :]
    mask = delta >= 0
    if context is not None:
        mask &= delta < context
    o = F.scaled_dot_product_attention(q.transpose(1, 2), K, V, attn_mask=mask)
    return F.linear(o.transpose(1, 2).reshape(B, T, C), wo)


def transformer_layer(x, state, W, p, num_heads, context, max_period):
    """Pre-LN block with optional LayerScale (reference `mimi_transformer.py:39-54`)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-5)
    a = streaming_attention(h, state, W[p + ".self_attn.in_proj.weight"], W[p + ".self_attn.out_proj.weight"],
                            num_heads, context, max_period)
    ls1 = W.get(p + ".layer_scale_1.scale")
    x = x + (a if ls1 is None else ls1 * a)
    h = F.layer_norm(x, (C,), W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-5)
    f = F.linear(F.gelu(F.linear(h, W[p + ".linear1.weight"])), W[p + ".linear2.weight"])
    ls2 = W.get(p + ".layer_scale_2.scale")
    return x + (f if ls2 is None else ls2 * f)


class FlowLM:
    """`FlowLMModel` inference path (reference `flow_lm.py:96-157`) as `_run_flow_lm_and_increment_step` drives it
    (reference `tts_model.py:317-367`)."""

    def __init__(self, cfg, W):
        self.cfg, self.W = cfg, _Weights(W)
        t = cfg.flow_lm.transformer
        self.D, self.H, self.L = t.d_model, t.num_heads, t.num_layers
        self.max_period = float(t.max_period)
        self.ldim = cfg.mimi.quantizer.dimension
        self.fd, self.depth = cfg.flow_lm.flow.dim, cfg.flow_lm.flow.depth
        self._tcomb = {}

    def init_state(self, B, T):
        """`init_states` (reference `stateful_module.py:7-16`, `transformer.py:46-57`)"""
        return [dict(cache=torch.full((2, B, T, self.H, self.D // self.H), float("nan")), offset=0) for _ in range(self.L)]

    def embed_text(self, tokens):
        return self.W["flow_lm.conditioner.embed.weight"][torch.as_tensor(tokens)]

    def backbone(self, state, text_emb, seq):
        """BOS substitution + input_linear + layers + out_norm (reference `flow_lm.py:121-122,141-157`)"""
        W = self.W
        seq = torch.where(torch.isnan(seq), W["flow_lm.bos_emb"], seq)
        x = torch.cat([text_emb, F.linear(seq, W["flow_lm.input_linear.weight"])], dim=1)
        T = x.shape[1]
        for i in range(self.L):
            x = transformer_layer(x, state[i], W, f"flow_lm.transformer.layers.{i}", self.H, None, self.max_period)
        for st in state:
            st["offset"] += T
        return F.layer_norm(x, (self.D,), W["flow_lm.out_norm.weight"], W["flow_lm.out_norm.bias"], 1e-5)

    def prefill(self, state, emb):
        emb = torch.as_tensor(emb, dtype=torch.float32)
        self.backbone(state, emb, torch.zeros(emb.shape[0], 0, self.ldim))

    def time_embed(self, i, t):
        """`TimestepEmbedder.forward` incl. the variance-based "RMSNorm" (reference `mlp.py:20-25,79-83`)"""
        W, p = self.W, f"flow_lm.flow_net.time_embed.{i}."
        args = t * W[p + "freqs"]
        e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = F.linear(F.silu(F.linear(e, W[p + "mlp.0.weight"], W[p + "mlp.0.bias"])), W[p + "mlp.2.weight"], W[p + "mlp.2.bias"])
        var = h.var(dim=-1, keepdim=True, unbiased=True) + 1e-5
        return h * (W[p + "mlp.3.alpha"] * torch.rsqrt(var))

    def flow_net(self, c, s, t, x):
        """`SimpleMLPAdaLN.forward` (reference `mlp.py:188-215`)"""
        W, p = self.W, "flow_lm.flow_net."
        x = F.linear(x, W[p + "input_proj.weight"], W[p + "input_proj.bias"])
        t_comb = (self.time_embed(0, s) + self.time_embed(1, t)) / 2
        sy = F.silu(t_comb + F.linear(c, W[p + "cond_embed.weight"], W[p + "cond_embed.bias"]))
        for i in range(self.depth):
            r = f"{p}res_blocks.{i}."
            shift, scale, gate = F.linear(sy, W[r + "adaLN_modulation.1.weight"], W[r + "adaLN_modulation.1.bias"]).chunk(3, dim=-1)
            h = F.layer_norm(x, (self.fd,), W[r + "in_ln.weight"], W[r + "in_ln.bias"], 1e-6) * (1 + scale) + shift
            h = F.linear(F.silu(F.linear(h, W[r + "mlp.0.weight"], W[r + "mlp.0.bias"])), W[r + "mlp.2.weight"], W[r + "mlp.2.bias"])
            x = x + gate * h
        r = p + "final_layer."
        shift, scale = F.linear(sy, W[r + "adaLN_modulation.1.weight"], W[r + "adaLN_modulation.1.bias"]).chunk(2, dim=-1)
        h = F.layer_norm(x, (self.fd,), None, None, 1e-6) * (1 + scale) + shift
        return F.linear(h, W[r + "linear.weight"], W[r + "linear.bias"])

    @torch.no_grad()
    def decode_step(self, state, latent_in, noise=None, lsd_steps=1, eos_threshold=-4.0):
        """`_sample_next_latent` on a [B, 1, ldim] input (reference `flow_lm.py:96-139`, `lsd_decode` :19-40)"""
        latent_in = torch.as_tensor(latent_in, dtype=torch.float32)
        B = latent_in.shape[0]
        c = self.backbone(state, torch.zeros(B, 0, self.D), latent_in.view(B, 1, self.ldim))[:, -1]
        W = self.W
        eos_logit = F.linear(c, W["flow_lm.out_eos.weight"], W["flow_lm.out_eos.bias"])[:, 0]
        cur = torch.zeros(B, self.ldim) if noise is None else torch.as_tensor(noise, dtype=torch.float32).clone()
        for i in range(lsd_steps):
            s = torch.full((B, 1), i / lsd_steps)
            t = torch.full((B, 1), (i + 1) / lsd_steps)
            cur = cur + self.flow_net(c, s, t, cur) / lsd_steps
        return cur, eos_logit, eos_logit > eos_threshold


class MimiDecoder:
    """`_decode_audio_worker` body + `MimiModel.decode_from_latent` (reference `tts_model.py:449-455`, `mimi.py:89-94`)
    over the streaming convolutions of `conv.py:93-163` and the SEANet decoder of `seanet.py:141-180`."""

    def __init__(self, cfg, W):
        from pocket_tts_amd.weights import seanet_decoder_layers  # inventory only, no compute

        self.cfg, self.W = cfg, _Weights(W)
        self.layers = seanet_decoder_layers(cfg)
        self.stride = cfg.upsample_stride
        self.tr = cfg.mimi.transformer

    def init_state(self, B, max_frames):
        st = {}
        C = self.cfg.mimi.seanet.dimension
        st["upsample"] = dict(partial=torch.zeros(B, C, self.stride))
        tr = self.tr
        Dh = tr.d_model // tr.num_heads
        st["attn"] = [dict(cache=torch.full((2, B, max_frames * self.stride, tr.num_heads, Dh), float("nan")), offset=0)
                      for _ in range(tr.num_layers)]
        for idx, kind, cin, cout, k, stride in self.layers:
            if kind == "convtr":
                st[idx] = dict(partial=torch.zeros(B, cout, k - stride))
            else:
                st[idx] = dict(previous=torch.zeros(B, cin, k - 1))
        return st

    @staticmethod
    def _sconv(x, w, b, st):
        """`StreamingConv1d.forward`, pad_mode "constant", stride 1 (reference `conv.py:93-115`)"""
        TP = st["previous"].shape[-1]
        if TP:
            x = torch.cat([st["previous"], x], dim=-1)
            st["previous"] = x[..., -TP:].clone()
        return F.conv1d(x, w, b)

    @staticmethod
    def _sconvtr(x, w, b, stride, st):
        """`StreamingConvTranspose1d.forward`: overlap-add with `partial` stored without bias (reference `conv.py:151-163`)"""
        y = F.conv_transpose1d(x, w, b, stride=stride)
        PT = st["partial"].shape[-1]
        if PT > 0:
            y[..., :PT] += st["partial"]
            tail = y[..., -PT:].clone()
            if b is not None:
                tail -= b[None, :, None]
            st["partial"] = tail
            y = y[..., :-PT]
        return y

    @torch.no_grad()
    def decode(self, st, latent):
        """latent [B, ldim] (normalised FlowLM output) -> pcm [B, frame_samples]"""
        W = self.W
        latent = torch.as_tensor(latent, dtype=torch.float32)
        x = latent * W["flow_lm.emb_std"] + W["flow_lm.emb_mean"]                    # tts_model.py:449
        x = F.conv1d(x[:, :, None], W["mimi.quantizer.output_proj.weight"])           # dummy_quantizer.py:17-18
        s = self.stride
        y = F.conv_transpose1d(x, W["mimi.upsample.convtr.convtr.weight"], None, stride=s, groups=x.shape[1])  # resample.py:40-51
        y[..., :s] += st["upsample"]["partial"]
        st["upsample"]["partial"] = y[..., s:].clone()
        h = y[..., :s].transpose(1, 2)                                                # mimi_transformer.py:140-150
        tr = self.tr
        for i in range(tr.num_layers):
            h = transformer_layer(h, st["attn"][i], W, f"mimi.decoder_transformer.transformer.layers.{i}",
                                  tr.num_heads, tr.context, float(tr.max_period))
        for a in st["attn"]:
            a["offset"] += s                                                          # increment_steps(mimi, state, 16)
        x = h.transpose(1, 2)
        for n, (idx, kind, cin, cout, k, stride) in enumerate(self.layers):            # seanet.py:141-180
            p = f"mimi.decoder.model.{idx}"
            if kind == "conv":
                x = self._sconv(F.elu(x) if n > 0 else x, W[p + ".conv.weight"], W[p + ".conv.bias"], st[idx])
            elif kind == "convtr":
                x = self._sconvtr(F.elu(x), W[p + ".convtr.weight"], W[p + ".convtr.bias"], stride, st[idx])
            else:                                                                     # SEANetResnetBlock seanet.py:33-41
                v = self._sconv(F.elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"], st[idx])
                x = x + F.conv1d(F.elu(v), W[p + ".block.3.conv.weight"], W[p + ".block.3.conv.bias"])
        return x[:, 0, :]
