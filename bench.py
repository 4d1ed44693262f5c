#!/usr/bin/env python3
"""Benchmark of the Pocket-TTS decode hot path on MI355X (metric: BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset headline|config4|b1|int8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one 80 ms frame of every utterance of the batch: one FlowLM autoregressive step plus one Mimi codec
decode for `--batch` (default 64) concurrent fixed-length utterances per GPU (BASELINE.json configs[2]; SURVEY.md
section 8d).  An *utterance job* is: voice-state clone + text prefill (32 tokens) + 125 steps (10 s of audio).  The
timed region consists of WHOLE utterance jobs, whatever `--steps K` is: at least 3 of them, at least ceil(K / 125), and
enough for >= 1 s of wall time (one untimed job calibrates the count), so the per-utterance costs (clone, prefill,
pipeline fill and drain) are always amortised over 125 steps each and a 20-step request does not turn into a 20 ms
measurement.  `steps` in the JSON line is the number of steps actually timed, `ms_per_step` = timed wall / steps, and
`value` is the MEDIAN per-utterance rate (GPU-timeline events at the utterance boundaries, max over ranks per
utterance); `value_whole_region` is total audio / total wall.  Inputs (weights, voice KV, token ids) are resident in HBM;
every PCM chunk lands in pinned host memory inside the timed region.  A cooperative-kernel timeout inside the timed
region (`ptts_lm_state_error`) makes the run exit non-zero without a JSON line.

`api_batch` in the line (N = 1 only) is the throughput a USER of the drop-in surface gets for 64 concurrent requests:
`TTSModel.generate_audio_batch` and `ContinuousBatcher` (64 slots), per-row EOS bookkeeping on, every PCM chunk handed to
the caller, tokenisation, voice clone and prefill inside the timed calls.

`--gpus N` without a launcher starts N fresh worker processes itself (before the parent touches any GPU), one engine
per GPU, rendezvous on 127.0.0.1; under `torch.distributed.run` the ranks it is given are used.  Utterances are
independent: each rank runs the same work on its own GPU, RCCL carries only the barriers and a few scalars, scaling is
"weak".  Presets: `config4` = 24-layer model, 32 utterances per GPU (BASELINE.json configs[3], meant for --gpus 8);
`b1` = batch 1; `int8` / `bf16codec` / `config5` / `fp8codec` / `config5fp8` / `lmbf16` = configs[4] and the other reduced-precision
formats (int8 LM weights, bf16 codec, fp8 codec convs, bf16 LM weights): separate lines, never the headline.

Weights are synthetic (seed 0), data synthetic; fp32 end to end like the reference.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FRAME_S = 0.08
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MIMI_MAC_PER_FRAME = 271e6  # Mimi decode, MACs per sequence and frame (same codec in every config)
PRESETS = {
    "headline": {},
    "config4": dict(config="24l", batch=32),
    "b1": dict(batch=1),
    "int8": dict(quantize=True),
    "bf16codec": dict(codec_bf16=True),
    "config5": dict(quantize=True, codec_bf16=True),
    "fp8codec": dict(codec_fp8=True),
    "config5fp8": dict(quantize=True, codec_fp8=True),   # BASELINE.json configs[4] as worded: int8 LM weights + fp8 codec convs
    "lmbf16": dict(lm_bf16=True),
    "split": dict(codec_split=True),   # error-compensated bf16x3 codec GEMMs: beside the headline
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=125)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--preset", choices=sorted(PRESETS), default="headline")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 64)")
    ap.add_argument("--config", default=None, choices=["en100m", "24l", "tiny"])
    ap.add_argument("--voice-len", type=int, default=126)
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--frames", type=int, default=125, help="frames per utterance (10 s)")
    ap.add_argument("--temp", type=float, default=0.7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--quantize", action="store_true", default=None,
                    help="BASELINE config #5: int8 weights for the FlowLM attention + FFN layers (not the headline)")
    ap.add_argument("--codec-bf16", action="store_true", default=None,
                    help="BASELINE config #5, second half: bf16 Mimi decoder, fp32 accumulate (not the headline)")
    ap.add_argument("--codec-fp8", action="store_true", default=None,
                    help="BASELINE config #5: SEANet convolutions on the fp8 MFMA, transformer bf16 (not the headline)")
    ap.add_argument("--codec-split", action="store_true", default=None,
                    help="codec GEMMs as hi*hi + hi*lo + lo*hi of bf16 halves, fp32 accumulate (experiment, not the headline)")
    ap.add_argument("--lm-bf16", action="store_true", default=None,
                    help="bf16 weights + operands for the FlowLM Linear layers, fp32 accumulate (not the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work budget of the cpu_baseline leg")
    ap.add_argument("--latency-trials", type=int, default=200)
    ap.add_argument("--no-api", action="store_true", help="skip the API-level batch throughput (api_batch)")
    ap.add_argument("--voices", choices=("distinct", "one"), default="distinct",
                    help="distinct: every utterance of the batch has its own voice state (SURVEY 8d: conditioning f32[B,126,1024]; "
                         "the headline); one: all clone ONE resident voice state, whose keys they then share (`one_voice` object)")
    ap.add_argument("--quick", action="store_true", help="A/B runs: only the timed region (no cpu baseline, latency, API, kernel profile)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="lower bound of the timed region's wall time")
    ap.add_argument("--min-utterances", type=int, default=3)
    args = ap.parse_args(argv)
    for k, v in PRESETS[args.preset].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.quick:
        args.no_cpu_baseline = args.no_latency = args.no_api = args.no_profile = True
    args.batch = 64 if args.batch is None else args.batch
    args.config = args.config or "en100m"
    args.quantize = bool(args.quantize)
    args.codec_bf16 = bool(args.codec_bf16)
    args.codec_fp8 = bool(args.codec_fp8)
    args.lm_bf16 = bool(args.lm_bf16)
    args.codec_split = bool(args.codec_split)
    return args


# --------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: the parent spawns the ranks (it has not touched a GPU: no re-exec of a GPU process)
def spawn_ranks(n: int) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0:
            print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
            rc = rc or c or 1
    return rc


class Job:
    """Fixed-length synthetic utterance batch on one GPU, hipGraph per step."""

    def __init__(self, eng, B, args, seed):
        import torch

        from pocket_tts_amd.engine import StepPipeline

        self.eng, self.B, self.args = eng, B, args
        cfg = eng.cfg
        dev = eng.device
        g = torch.Generator().manual_seed(1 + seed)
        nv = B if getattr(args, "voices", "one") == "distinct" else 1
        voice = (torch.randn(nv, args.voice_len, eng.D, generator=g) * 0.1).to(dev)
        g2 = torch.Generator().manual_seed(2 + seed)
        self.tokens = torch.randint(0, cfg.flow_lm.lookup_table.n_bins, (B, args.text_len), generator=g2).to(dev)
        cap = args.voice_len + args.text_len + args.frames + 1
        # voice states are computed once and cloned by every utterance (predefined voices are pre-baked KV files in the
        # reference: tts_model.py:853-869).  distinct: one per row of the batch; one: a single state every row clones - the
        # clones then borrow its first voice_len & ~15 keys instead of copying them (KvPrefix)
        self.voice = eng.new_lm_state(nv, cap)
        eng.lm_prefill(self.voice, voice)
        self.st = eng.new_lm_state(B, cap)
        self.ms = eng.new_mimi_state(B)
        self.st.set_noise(args.temp, 1234 + seed)
        # EOS stop disabled (threshold +inf) so every utterance has exactly `frames` frames
        self.pipe = StepPipeline(eng, self.st, self.ms, None, 1, float("inf"))
        self.frame = args.frames  # forces a (re)start on the first step
        self.contexts = []        # FlowLM context (keys attended) of every step since reset_log()

    def start_utterances(self):
        eng = self.eng
        self.st.copy_from(self.voice)            # per-chunk state clone (tts_model.py:637-638)
        emb = eng.embed_text(self.tokens)        # LUT gather (text.py:74-76)
        eng.lm_prefill(self.st, emb)             # text prefill (tts_model.py:722-725)
        self.pipe.restart()
        self.frame = 0

    def step(self):
        if self.frame >= self.args.frames:
            self.start_utterances()
        self.pipe.step()
        self.contexts.append(self.args.voice_len + self.args.text_len + self.frame + 1)
        self.frame += 1
        if self.frame >= self.args.frames:
            self.pipe.flush()  # the utterances' last frame has no FlowLM step to ride along with

    def run_utterance(self):
        """one whole utterance job: clone + prefill + `frames` steps (the last frame's codec decode included)"""
        self.frame = self.args.frames
        for _ in range(self.args.frames):
            self.step()

    def sync(self):
        self.pipe.sync()


class _CharTokenizer:
    """Stand-in tokenizer for the synthetic model (no sentencepiece model of vocabulary n_bins exists offline): one id
    per character behind a leading marker token, invertible, so the reference's sentence splitting works unchanged."""

    def __init__(self, n_bins):
        self.n_bins, self.sp = n_bins, self

    def encode(self, text):
        return [self.n_bins - 1] + [ord(c) % (self.n_bins - 1) for c in text]

    def decode(self, ids):
        return "".join(chr(i) for i in ids if i != self.n_bins - 1)


def first_chunk_latency(eng, args, job1):
    """B=1 streaming path (BASELINE.json configs[1]) measured THROUGH the drop-in API: time from the
    `TTSModel.generate_audio_stream()` call (voice state resident) to the first [1920] chunk on the host = state clone +
    text prefill + FlowLM step + Mimi frame + D2H.  Also: the same at engine level, and the steady-state step time."""
    import numpy as np

    from pocket_tts_amd.tts_model import TTSModel, _export_lm_state

    cfg = eng.cfg
    model = TTSModel(eng, cfg, _CharTokenizer(cfg.flow_lm.lookup_table.n_bins), args.temp, 1, None, float("inf"))
    voice_state = _export_lm_state(eng, job1.voice, args.voice_len)  # reference-format dict, as a caller holds it
    text = "The quick brown fox jumps over."  # 31 characters + marker = 32 tokens
    assert len(model.tokenizer.encode(text)) == args.text_len or args.text_len != 32
    api_ms = []
    for t in range(args.latency_trials + 5):
        eng.sync()
        t0 = time.perf_counter()
        gen = model.generate_audio_stream(voice_state, text)
        chunk = next(gen)
        dt = (time.perf_counter() - t0) * 1e3
        gen.close()
        assert chunk.shape[0] == eng.frame_samples
        if t >= 5:
            api_ms.append(dt)
    eng_ms = []
    for t in range(args.latency_trials // 4 + 5):
        job1.sync()
        eng.sync()
        t0 = time.perf_counter()
        job1.start_utterances()
        job1.pipe.step()
        f = job1.pipe.flush()
        job1.pipe.done_event(f).synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if t >= 5:
            eng_ms.append(dt)
    # steady-state per-step time of the B=1 pipeline
    job1.start_utterances()
    for _ in range(20):
        job1.step()
    job1.sync()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        job1.step()
    job1.sync()
    per_step_ms = (time.perf_counter() - t0) * 1e3 / n
    return dict(first_chunk_ms_p50=float(np.percentile(api_ms, 50)), first_chunk_ms_p99=float(np.percentile(api_ms, 99)),
                measured_through="TTSModel.generate_audio_stream()", trials=args.latency_trials,
                engine_level_first_chunk_ms_p50=float(np.percentile(eng_ms, 50)),
                b1_ms_per_step=per_step_ms, b1_xrt=FRAME_S * 1e3 / per_step_ms)


def one_voice_throughput(eng, args, n_utt, seed, table_id, profile):
    """The same utterance jobs with every row cloned from ONE resident voice state (64 requests against one voice, as
    `api_batch` and a server with a handful of predefined voices run them): the clones borrow the voice's first
    voice_len & ~15 keys (KvPrefix) and the decode attention scores them as MFMA tiles shared by 4 rows
    (attn_cascade_kernel).  Same arithmetic to fp32 rounding (tests/test_gpu_prefix.py, test_gpu_parity_r3.py); reported
    BESIDE the headline, whose rows have distinct voices as SURVEY 8d specifies."""
    import numpy as np
    import torch

    a = argparse.Namespace(**vars(args))
    a.voices = "one"
    job = Job(eng, args.batch, a, seed=seed)
    job.run_utterance()
    job.sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_utt + 1)]
    eng.sync()
    torch.cuda.synchronize()
    job.contexts = []
    t0 = time.perf_counter()
    marks[0].record(job.pipe.s2)
    for u in range(n_utt):
        job.run_utterance()
        marks[u + 1].record(job.pipe.s2)
    job.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if job.st.error():
        raise RuntimeError("a cooperative FlowLM kernel timed out in the one-voice leg")
    utt_ms = [marks[u].elapsed_time(marks[u + 1]) for u in range(n_utt)]
    rates = [args.batch * args.frames * FRAME_S / (m * 1e-3) for m in utt_ms]
    pre = args.voice_len & ~15
    out = dict(value=float(np.median(rates)), unit="audio-seconds/sec", ms_per_step=wall * 1e3 / (n_utt * args.frames),
               utterances_timed=n_utt, utterance_ms=[round(m, 3) for m in utt_ms], shared_prefix_keys=pre,
               options=dict(share_prefix=os.environ.get("PTTS_SHARE_PREFIX", "1"), prefix_cascade=os.environ.get("PTTS_CASCADE", "1")),
               note="all rows clone ONE voice state and share its keys; the headline `value` is measured with a distinct voice per row")
    if profile:
        rows, per_kernel, nst, pctx = kernel_profile(eng, job)
        for k, v in per_kernel.items():
            if k.startswith(("attn_cascade", "attn_decode")):
                t_ = eng.cfg.flow_lm.transformer
                avg_s = v["total_ms"] / v["count"] * 1e-3
                uniq = 2 * 4 * 64 * t_.num_heads * (pre + args.batch * (pctx - pre)) + 8.0 * args.batch * t_.num_heads * 64
                # `achieved` on SURVEY 8d's per-sequence figure (what the reference's algorithm reads), `achieved_unique` with the
                # shared keys counted once (what has to come from HBM); traffic = the committed PMC pass for this label or null
                out["attention"] = dict(kernel=k, context_keys=pctx, avg_us=avg_s * 1e6, launches=v["count"],
                                        algorithmic_bytes_per_launch=v["bytes"] / v["count"],
                                        achieved_GBs=v["bytes"] / v["count"] / avg_s / 1e9,
                                        frac=v["bytes"] / v["count"] / avg_s / 1e9 / HBM_PEAK_GBS,
                                        unique_bytes_per_launch=uniq, achieved_unique_GBs=uniq / avg_s / 1e9,
                                        frac_unique=uniq / avg_s / 1e9 / HBM_PEAK_GBS,
                                        traffic=pmc_traffic(k, table_id, PMC_FILE_ONE),
                                        rocprof_pipelined_avg_us=rocprof_avg_us("attn_cascade_kernel", STATS_FILE_ONE),
                                        rocprof_file="profiles/" + STATS_FILE_ONE)
        out["kernel_sum_ms_per_step"] = sum(r["total_ms"] for r in rows) / nst
    return out, job


def api_batch_throughput(eng, args, voice_lm_state):
    """BASELINE config #3 through the API a user calls: 64 requests of 32 tokens against one voice,
    EOS stop disabled by the threshold (every request runs its `estimate_max_gen_len` = 159 frames, the reference's own
    bound, tts_model.py:907-910) but the per-row EOS bookkeeping of tts_model.py:756-768 runs on every step; tokenising,
    voice clone, grouped prefill, the step pipeline and handing every PCM chunk to the caller are inside the timed calls.
    Median of 3 calls after one warm call each."""
    import logging

    import numpy as np

    from pocket_tts_amd.batching import ContinuousBatcher
    from pocket_tts_amd.text import estimate_max_gen_len
    from pocket_tts_amd.tts_model import TTSModel, _export_lm_state

    logging.getLogger("pocket_tts_amd").setLevel(logging.ERROR)  # "max length without EOS" x 64 per call is expected here
    cfg, B = eng.cfg, args.batch
    model = TTSModel(eng, cfg, _CharTokenizer(cfg.flow_lm.lookup_table.n_bins), args.temp, 1, None, float("inf"))
    voice_state = _export_lm_state(eng, voice_lm_state, args.voice_len)
    texts = [f"The quick brown fox jumps {i:04d}." for i in range(B)]  # 31 characters + marker = 32 tokens each
    assert all(len(model.tokenizer.encode(t)) == 32 for t in texts)
    frames = estimate_max_gen_len(32, cfg.mimi.frame_rate)
    audio = B * frames * FRAME_S
    out = dict(requests=B, tokens_per_request=32, frames_per_request=frames,
               note="EOS threshold +inf: fixed-length requests, bookkeeping on; PCM handed to the caller")
    ts = []
    for i in range(4):
        eng.sync()
        t0 = time.perf_counter()
        wavs = model.generate_audio_batch(voice_state, texts)
        dt = time.perf_counter() - t0
        assert len(wavs) == B and all(w.shape[0] == frames * eng.frame_samples for w in wavs)
        if i:
            ts.append(dt)
    out["generate_audio_batch_xrt"] = audio / float(np.median(ts))
    out["generate_audio_batch_ms"] = float(np.median(ts)) * 1e3
    model._drop_batch_contexts()
    cb = ContinuousBatcher(model, slots=B, capacity=512)
    try:
        ts = []
        for i in range(4):
            eng.sync()
            t0 = time.perf_counter()
            reqs = [cb.submit(voice_state, t) for t in texts]
            cb.run_until_idle()
            n = 0
            for r in reqs:
                for chunk in r:  # a server would write each chunk to its client's socket here
                    n += chunk.shape[0]
            dt = time.perf_counter() - t0
            assert n == B * frames * eng.frame_samples, (n, B * frames * eng.frame_samples)
            if i:
                ts.append(dt)
        out["continuous_batcher_xrt"] = audio / float(np.median(ts))
        out["continuous_batcher_ms"] = float(np.median(ts)) * 1e3
    finally:
        cb.close()
    return out


def kernel_profile(eng, job, nsteps=6):
    """A few eager (non-graph) steps with per-launch HIP events on the launch stream."""
    import torch

    job.start_utterances()
    for _ in range(40):  # mid-utterance context
        job.step()
    eng.sync()
    ctx = job.contexts[-1] + 1
    eng.profile_start()
    job.pipe.flush()
    job.sync()
    P = job.pipe
    pcm_dev = torch.empty(job.B, eng.frame_samples, device=eng.device)
    for _ in range(nsteps):
        eng.lm_decode_step(job.st, None, None, 1, float("inf"), P.lat[0], P.logit[0], None)
        eng.mimi_decode(job.ms, P.lat[0], pcm_dev)
    rows = eng.profile_stop()
    job.frame = job.args.frames
    per_kernel = {}
    for r in rows:
        k = per_kernel.setdefault(r["kernel"], dict(count=0, total_ms=0.0, bytes=0.0, flops=0.0))
        for f in ("count", "total_ms", "bytes", "flops"):
            k[f] += r[f]
    return rows, per_kernel, nsteps, ctx


def tune_table_id(eng) -> str:
    """identifies the tile table the kernels of this run were chosen from (stamped into PMC files and bench lines)"""
    return hashlib.sha256("\n".join(sorted(eng._tune_table())).encode()).hexdigest()[:12]


PMC_FILE = "r03_pmc_traffic.json"     # tools/profile_round.sh -> profiles/ (FETCH_SIZE / WRITE_SIZE passes of this round's build)
STATS_FILE = "r03_b64_kernel_stats.csv"  # rocprofv3 --kernel-trace --stats of the headline command
PMC_FILE_ONE = "r03_onevoice_pmc_traffic.json"     # the same passes with --voices one (rows share the voice's keys)
STATS_FILE_ONE = "r03_onevoice_kernel_stats.csv"


def pmc_traffic(name, table_id, pmc_file=None):
    """HBM bytes per launch of `name` from the committed PMC passes (profiles/r03_pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 calibration).  PMC counters cannot be
    read from inside this process, so the file is a replay; it is only used when it was measured on the SAME tile
    table as this run (its `tune_table_id` stamp), else None."""
    try:
        with open(os.path.join(REPO, "profiles", pmc_file or PMC_FILE)) as f:
            d = json.load(f)
        if d.get("tune_table_id") != table_id:
            return None
        return d["kernels"][name]["traffic_bytes_per_launch"]
    except Exception:
        return None


def rocprof_avg_us(kernel_substr, stats_file=None):
    """average duration of a kernel over the WHOLE pipelined bench run from the committed rocprofv3 --stats summary (the
    cross-check SURVEY 8d asks for: both streams running, contexts 159-283), or None"""
    import csv

    try:
        with open(os.path.join(REPO, "profiles", stats_file or STATS_FILE)) as f:
            for r in csv.DictReader(f):
                if kernel_substr in r["Name"]:
                    return float(r["AverageNs"]) / 1e3
    except Exception:
        pass
    return None


def roofline_of(rows, table_id):
    """The dominant kernel = the (launch site, kernel) pair with the largest total time: one site is one operand shape,
    so bytes / flops per launch are those of ONE GEMM or attention shape (a kernel label alone can cover several shapes,
    e.g. out_proj and linear2 of the FlowLM share a tile configuration and a grid)."""
    per_site = {}
    for r in rows:
        k = per_site.setdefault((r["site"], r["kernel"]), dict(count=0, total_ms=0.0, bytes=0.0, flops=0.0))
        for f in ("count", "total_ms", "bytes", "flops"):
            k[f] += r[f]
    (site, name), k = max(per_site.items(), key=lambda kv: kv[1]["total_ms"])
    avg_s = k["total_ms"] / k["count"] * 1e-3
    gbs = k["bytes"] / k["count"] / avg_s / 1e9
    tfs = k["flops"] / k["count"] / avg_s / 1e12
    common = dict(kernel=name, site=site, traffic=pmc_traffic(name, table_id), avg_us=avg_s * 1e6, launches=k["count"],
                  algorithmic_bytes_per_launch=k["bytes"] / k["count"], flops_per_launch=k["flops"] / k["count"],
                  timing="HIP events around each launch on its own stream (eager steps at mid-utterance context)")
    if name.startswith(("attn_decode", "attn_cascade")):
        # cross-check against the committed rocprofv3 summary: its average is over the whole PIPELINED run (codec stream
        # co-running, contexts up to 283 keys), so it is slower than the isolated eager launch timed here
        common["rocprof_pipelined_avg_us"] = rocprof_avg_us("attn_cascade_kernel" if name.startswith("attn_cascade") else "attn_decode2_kernel")
        common["rocprof_file"] = "profiles/" + STATS_FILE
    if tfs / MFMA_F32_PEAK_TF > gbs / HBM_PEAK_GBS:
        return dict(bound="mfma", achieved=tfs, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=tfs / MFMA_F32_PEAK_TF, **common)
    return dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, **common)


def cpu_baseline(args, cfg, W):
    """The CPU restatement of the path on stock PyTorch CPU operators (oracle/torch_oracle.py, pinned to the reference's
    golden vectors by tests/test_torch_oracle.py), timed on this box's host cores for a bounded sample of the same
    workload: (ii) the benchmark's batch on all host cores -> `value`; (i) the reference's own operating
    point: batch 1, torch.set_num_threads(1), FlowLM and Mimi pipelined on two threads (tts_model.py:49,651-658)."""
    import queue
    import threading

    import torch

    from oracle import torch_oracle as O

    B = args.batch
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    g = torch.Generator().manual_seed(1)
    ncores = os.cpu_count() or 1
    torch.set_num_threads(min(ncores, 64))  # ATen's intra-op pool; more threads than ~64 only add contention
    used = torch.get_num_threads()

    def prefilled(b, nsteps):
        st = lm.init_state(b, args.voice_len + args.text_len + nsteps + 1)
        lm.prefill(st, (torch.randn(1, args.voice_len, lm.D, generator=g) * 0.1).expand(b, -1, -1))
        lm.prefill(st, lm.embed_text(torch.randint(0, cfg.flow_lm.lookup_table.n_bins, (b, args.text_len), generator=g)))
        return st

    # (ii) batch B, all cores: one warm step, then as many steps as fit the budget
    nmax = 64
    st, ms = prefilled(B, nmax), dec.init_state(B, nmax)
    x = torch.full((B, lm.ldim), float("nan"))
    budget = args.cpu_seconds * 0.6
    n, t0 = 0, None
    for i in range(nmax):
        if i == 1:
            t0 = time.perf_counter()
        noise = torch.randn(B, lm.ldim, generator=g) * args.temp ** 0.5
        x, _, _ = lm.decode_step(st, x, noise, 1, float("inf"))
        dec.decode(ms, x)
        if i >= 1:
            n += 1
            if time.perf_counter() - t0 > budget:
                break
    dt = time.perf_counter() - t0
    out = dict(value=B * n * FRAME_S / dt, unit="audio-seconds/sec", cores=used, kind="port",
               sample=f"{n} decode steps (FlowLM + Mimi) of batch {B} after voice+text prefill, torch {torch.__version__} CPU "
                      f"operators with {used} intra-op threads on a {ncores}-thread host, {dt:.1f} s wall")
    # (i) batch 1, one intra-op thread per stage, two pipeline threads like the reference
    torch.set_num_threads(1)
    st1, ms1 = prefilled(1, nmax), dec.init_state(1, nmax)
    q: queue.Queue = queue.Queue()
    done = []

    def codec():
        torch.set_num_threads(1)
        while True:
            z = q.get()
            if z is None:
                return
            dec.decode(ms1, z)
            done.append(time.perf_counter())

    th = threading.Thread(target=codec)
    th.start()
    x1 = torch.full((1, lm.ldim), float("nan"))
    budget1 = args.cpu_seconds * 0.4
    t1 = time.perf_counter()
    n1 = 0
    for i in range(nmax):
        x1, _, _ = lm.decode_step(st1, x1, torch.randn(1, lm.ldim, generator=g) * args.temp ** 0.5, 1, float("inf"))
        q.put(x1)
        n1 += 1
        if time.perf_counter() - t1 > budget1:
            break
    q.put(None)
    th.join()
    d1 = done[-1] - t1
    out["batch1_reference_setting"] = dict(
        value=n1 * FRAME_S / d1, unit="audio-seconds/sec", cores=2,
        sample=f"{n1} frames of batch 1, torch.set_num_threads(1), FlowLM thread -> queue -> Mimi thread "
               f"(reference tts_model.py:49,651-658), {d1:.1f} s wall")
    torch.set_num_threads(used)
    return out


def dryrun(args):
    """`PTTS_BENCH_DRYRUN=1`: the multi-rank plumbing of this file WITHOUT a GPU (tests/test_host_cpu.py, world_size 2
    over gloo): rendezvous, barriers, whole-job aggregation (sum of units over ranks / max wall over ranks), per-rank
    values, ONE JSON line from rank 0.  Every rank pretends a wall time of 0.1 s x (rank + 1)."""
    import torch

    from pocket_tts_amd import parallel

    rank, local, world = parallel.env_ranks()
    dist = parallel.init_distributed("gloo")
    if dist is not None:
        dist.barrier()
    wall_rank = 0.1 * (rank + 1)
    audio_rank = args.batch * args.steps * FRAME_S
    rate, wall = parallel.job_throughput(audio_rank, wall_rank, dist)
    per_rank = [audio_rank / wall_rank]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.float64)
        t[rank] = per_rank[0]
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.tolist()]
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dryrun", "value": rate, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": wall * 1e3 / args.steps, "per_rank_xrt": per_rank, "scaling": "weak"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if os.environ.get("PTTS_BENCH_DRYRUN"):
        return dryrun(args)
    import numpy as np
    import torch

    # Tile choices come from the table measured on an MI355X and committed with the profiles (so kernel names in this
    # run, in profiles/*_kernel_stats.csv and in profiles/*_pmc_traffic.json mean the same configurations); shapes
    # missing from it are tuned live and written to an UNTRACKED file.  PTTS_TUNE_CACHE= (empty) tunes everything live.
    os.environ.setdefault("PTTS_TUNE_CACHE", os.path.join(REPO, "profiles", "tune_cache_mi355x.txt"))
    if not os.environ["PTTS_TUNE_CACHE"]:
        del os.environ["PTTS_TUNE_CACHE"]
    scratch = os.path.join(REPO, "gpurun_out")
    os.makedirs(scratch, exist_ok=True)
    from pocket_tts_amd import parallel

    rank, local, world = parallel.env_ranks()
    os.environ.setdefault("PTTS_TUNE_CACHE_OUT", os.path.join(scratch, f"tune_additions_rank{rank}.txt"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using the launcher's world size", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    dist = parallel.init_distributed("nccl", dev)  # RCCL; only barriers + scalar reductions

    from pocket_tts_amd.config import named_config
    from pocket_tts_amd.engine import Engine
    from pocket_tts_amd.weights import generate_state_dict

    cfg = named_config(args.config)
    W = generate_state_dict(cfg, 0)
    groups = (({"attention", "ffn"} if args.quantize else set()) | ({"codec_bf16"} if args.codec_bf16 else set())
              | ({"codec_fp8"} if args.codec_fp8 else set()) | ({"lm_bf16"} if args.lm_bf16 else set())
              | ({"codec_split"} if args.codec_split else set()))
    eng = Engine(cfg, W, dev, quantize_groups=groups or None)
    job = Job(eng, args.batch, args, seed=rank)

    def barrier():
        job.sync()
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.step()
    job.pipe.flush()
    # one untimed whole utterance: warms the prefill path and tells how many utterances make >= --min-seconds
    barrier()
    tc = time.perf_counter()
    job.run_utterance()
    job.sync()
    t_utt = time.perf_counter() - tc
    n_utt = max(args.min_utterances, -(-args.steps // args.frames), int(np.ceil(args.min_seconds / max(t_utt, 1e-6))))
    if dist is not None:  # every rank times the same number of utterances
        t = torch.tensor([n_utt], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_utt = int(t.item())
    steps_timed = n_utt * args.frames
    job.contexts = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_utt + 1)]
    barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    marks[0].record(job.pipe.s2)
    for u in range(n_utt):
        job.run_utterance()
        marks[u + 1].record(job.pipe.s2)  # behind the utterance's last codec frame (the codec stream)
    job.sync()
    ev_ms = eng.timer_stop_ms()
    barrier()
    wall_rank = time.perf_counter() - t0
    err = torch.tensor([1 if job.st.error() else 0], dtype=torch.int32, device=dev)
    if dist is not None:
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
    if int(err.item()):
        print("bench.py: a cooperative FlowLM kernel timed out inside the timed region (ptts_lm_state_error): the "
              "measurement is invalid", file=sys.stderr)
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)
    utt_ms = torch.tensor([marks[u].elapsed_time(marks[u + 1]) for u in range(n_utt)], dtype=torch.float64, device=dev)
    audio_rank = args.batch * steps_timed * FRAME_S
    rate, wall = parallel.job_throughput(audio_rank, wall_rank, dist, dev)
    per_rank = [audio_rank / wall_rank]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = per_rank[0]
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.tolist()]
        dist.all_reduce(utt_ms, op=dist.ReduceOp.MAX)  # an utterance of the job ends when its slowest rank ends
    utt_ms = [float(v) for v in utt_ms.tolist()]
    utt_rates = [world * args.batch * args.frames * FRAME_S / (m * 1e-3) for m in utt_ms]

    if rank == 0:
        L = cfg.flow_lm.transformer.num_layers
        ctx = float(np.mean(job.contexts))
        restarts = sum(1 for c in job.contexts if c == args.voice_len + args.text_len + 1)
        out = {
            "metric": "audio-seconds/sec (xRT), 100M en model, whole job over all GPUs",
            "value": float(np.median(utt_rates)),
            "unit": "audio-seconds/sec",
            "n_gpus": world,
            "steps": steps_timed,
            "steps_requested": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / steps_timed,
            "value_whole_region": rate,
            "utterances_timed": n_utt,
            "utterance_ms": [round(m, 3) for m in utt_ms],
            "value_definition": "median over the timed utterances of (GPUs x batch x 10 s) / utterance time; utterance "
                                "time from HIP events behind each utterance's last codec frame (codec stream), max over ranks",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": " + ".join(([("int8 weights (FlowLM attention+ffn), f32 activations/accumulate")] if args.quantize else [])
                                + (["bf16 FlowLM Linear weights + operands, f32 accumulate"] if args.lm_bf16 else [])
                                + (["f32 FlowLM + codec GEMMs on split bf16 (hi*hi + hi*lo + lo*hi, f32 accumulate, f32 buffers)"] if args.codec_split else [])
                                + (["bf16 codec (weights+activations), f32 accumulate"] if args.codec_bf16 else [])
                                + (["fp8 e4m3 SEANet convs + bf16 Mimi transformer, f32 accumulate"] if args.codec_fp8 else [])) or "f32",
            "data": "synthetic (seeded weights, voice KV, token ids; fixed-length utterances, EOS stop disabled)",
            "config": {
                "workload": f"{args.config}: batch {args.batch} concurrent utterances/GPU, "
                            f"{'a distinct voice per utterance' if args.voices == 'distinct' else 'one voice for all utterances'}, voice KV {args.voice_len} + "
                            f"text {args.text_len} tokens, {args.frames} frames (10 s) each, temp {args.temp}, "
                            f"lsd_decode_steps 1; per utterance: state clone + text prefill + FlowLM step + Mimi "
                            f"decode per frame, hipGraph per FlowLM step and per codec frame on two streams (step t+1 "
                            f"overlaps frame t), PCM written straight into pinned host memory",
                "preset": args.preset,
                "batch_per_gpu": args.batch,
                "parallelism": f"replicas x{world} (no collective on the data path)",
                "timed_region": f"{n_utt} whole utterances = {steps_timed} steps: {restarts} clone+prefill inside, "
                                f"mean FlowLM context {ctx:.1f} keys, {wall:.3f} s wall",
            },
            "xrt_per_gpu": rate / world,
            "per_rank_xrt": per_rank,
            "stream_event_ms_per_step": ev_ms / steps_timed,
            "tune_table_id": tune_table_id(eng),
        }
        # whole-step bounds at the contexts actually timed (SURVEY 8d bytes_step / flops formulas)
        bytes_step = (eng.lm_weight_bytes() + eng.mimi_weight_bytes()
                      + args.batch * (8 * L * ctx * 1024 + 2.18e6 + 68e3))
        t_, f_ = cfg.flow_lm.transformer, cfg.flow_lm.flow   # one MAC per weight and row, whatever the weight format
        d_, ff_, ld_ = t_.d_model, t_.d_model * t_.hidden_scale, cfg.mimi.quantizer.dimension
        lm_mac = (L * (4 * d_ * d_ + 2 * d_ * ff_) + ld_ * d_ + d_ * (f_.dim + 1) + f_.dim * (3 * f_.dim * f_.depth + 2 * f_.dim)
                  + ld_ * f_.dim + f_.depth * 2 * f_.dim * f_.dim + f_.dim * ld_)
        flops_step = 2.0 * args.batch * (lm_mac + 2 * L * ctx * 1024 + MIMI_MAC_PER_FRAME)
        step_s = wall / steps_timed
        out["step_roofline"] = {
            "algorithmic_bytes_per_step": bytes_step, "hbm_bound_us_at_8TBs": bytes_step / 8e12 * 1e6,
            "hbm_frac": bytes_step / 8e12 / step_s,
            "algorithmic_flops_per_step": flops_step, "mfma_f32_bound_us": flops_step / (MFMA_F32_PEAK_TF * 1e12) * 1e6,
            "mfma_frac": flops_step / (MFMA_F32_PEAK_TF * 1e12) / step_s,
            "mean_context": ctx,
        }
        if args.voices == "one" and os.environ.get("PTTS_SHARE_PREFIX", "1") != "0":  # the clones borrow the voice's first keys
            pre = args.voice_len & ~15
            out["step_roofline"]["shared_prefix_keys"] = pre
            out["step_roofline"]["unique_bytes_per_step"] = bytes_step - (args.batch - 1) * 8 * L * pre * 1024
        if not args.no_profile:
            rows, per_kernel, nst, pctx = kernel_profile(eng, job)
            tid = out["tune_table_id"]
            out["roofline"] = roofline_of(rows, tid)
            out["roofline"]["context_keys"] = pctx
            tot = sum(r["total_ms"] for r in rows)
            out["kernel_ms_per_step"] = {k: round(v["total_ms"] / nst, 4) for k, v in
                                         sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_ms"])}
            out["kernel_sum_ms_per_step"] = tot / nst
            # per label: [algorithmic MB per launch, HBM-side MB per launch from the committed PMC passes or null]
            out["kernel_mb_per_launch"] = {k: [round(v["bytes"] / v["count"] / 1e6, 2),
                                               (lambda t: None if t is None else round(t / 1e6, 2))(pmc_traffic(k, tid))]
                                           for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_ms"])}
            sites = {}
            for r in rows:
                k = sites.setdefault(r["site"] + " " + r["kernel"], [0, 0.0])
                k[0] += r["count"]
                k[1] += r["total_ms"]
            out["site_us_per_launch"] = {k: [round(v[1] / v[0] * 1e3, 2), round(v[0] / nst, 2)]
                                         for k, v in sorted(sites.items(), key=lambda kv: -kv[1][1])}
        if not args.no_latency and world == 1:
            a1 = argparse.Namespace(**vars(args))
            job1 = Job(eng, 1, a1, seed=7)
            out["latency_b1"] = first_chunk_latency(eng, a1, job1)
            job1 = None
        job_one = None
        if world == 1 and args.voices == "distinct" and args.batch >= 16 and not args.quick:
            try:  # a secondary leg must not cost the headline line
                out["one_voice"], job_one = one_voice_throughput(eng, args, n_utt, rank, out["tune_table_id"], not args.no_profile)
            except Exception as e1:  # noqa: BLE001
                out["one_voice"] = {"error": f"{type(e1).__name__}: {e1}"}
        if not args.no_api and world == 1 and args.preset in ("headline", "b1") and args.config != "24l" and args.batch > 1:
            if job_one is None:
                a1v = argparse.Namespace(**vars(args))
                a1v.voices = "one"
                job_one = Job(eng, args.batch, a1v, seed=rank)
            out["api_batch"] = api_batch_throughput(eng, args, job_one.voice)
            out["api_batch"]["engine_level_xrt"] = out["value"]
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, cfg, W)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    job = None
    eng.close()


if __name__ == "__main__":
    main()
