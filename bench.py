#!/usr/bin/env python3
"""Benchmark of the Pocket-TTS decode hot path on MI355X (metric: BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one 80 ms frame of every utterance of the batch: one FlowLM autoregressive step plus one
Mimi codec decode for `--batch` (default 64) concurrent fixed-length utterances per GPU
(BASELINE.json configs[2]).  A *job* is one batch of 10 s utterances: voice-state
clone + text prefill (32 tokens) + 125 steps; all of it sits inside the timed region, with inputs
(weights, voice KV, token ids) resident in HBM.  Each rank runs the same work on its own GPU
(utterances are independent: no collective on the data path), so scaling is "weak".

Weights are synthetic (seed 0), data synthetic; fp32 end to end like the reference.
One JSON line is printed by rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FRAME_S = 0.08
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=125)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--config", default="en100m", choices=["en100m", "24l", "tiny"])
    ap.add_argument("--voice-len", type=int, default=126)
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--frames", type=int, default=125, help="frames per utterance (10 s)")
    ap.add_argument("--temp", type=float, default=0.7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--quantize", action="store_true",
                    help="BASELINE config #5: int8 weights for the FlowLM attention + FFN layers (not the headline: "
                         "the default run is the fp32 path whose parity is pinned)")
    ap.add_argument("--cpu-steps", type=int, default=8)
    return ap.parse_args()


class Job:
    """Fixed-length synthetic utterance batch on one GPU, hipGraph per step."""

    def __init__(self, eng, B, args, seed):
        self.eng, self.B, self.args = eng, B, args
        cfg = eng.cfg
        dev = eng.device
        g = torch.Generator().manual_seed(1 + seed)
        voice = (torch.randn(1, args.voice_len, eng.D, generator=g) * 0.1).to(dev)
        g2 = torch.Generator().manual_seed(2 + seed)
        self.tokens = torch.randint(0, cfg.flow_lm.lookup_table.n_bins, (B, args.text_len), generator=g2).to(dev)
        cap = args.voice_len + args.text_len + args.frames + 1
        # voice state is computed once and reused by every utterance (predefined voices are pre-baked KV
        # files in the reference: tts_model.py:853-869)
        self.voice = eng.new_lm_state(1, cap)
        eng.lm_prefill(self.voice, voice)
        self.st = eng.new_lm_state(B, cap)
        self.ms = eng.new_mimi_state(B)
        self.st.set_noise(args.temp, 1234 + seed)
        # EOS stop disabled (threshold +inf) so every utterance has exactly `frames` frames
        from pocket_tts_amd.engine import StepPipeline

        self.pipe = StepPipeline(eng, self.st, self.ms, None, 1, float("inf"))
        self.frame = args.frames  # forces a (re)start on the first step

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
        self.frame += 1
        if self.frame >= self.args.frames:
            self.pipe.flush()  # the utterances' last frame has no FlowLM step to ride along with

    def sync(self):
        self.pipe.sync()


def first_chunk_latency(eng, args, trials=200):  # 200 trials: SURVEY 8(d)
    """B=1 streaming path (BASELINE.json configs[1]): time from request (voice state resident) to the
    first 80 ms PCM chunk on the host = state clone + text prefill + 1 LM step + 1 Mimi frame + D2H."""
    B = 1
    a = argparse.Namespace(**vars(args))
    job = Job(eng, B, a, seed=7)
    lat_ms = []
    for t in range(trials + 5):
        job.sync()
        eng.sync()
        t0 = time.perf_counter()
        job.start_utterances()
        job.pipe.step()
        job.pipe.flush()
        job.sync()
        dt = (time.perf_counter() - t0) * 1e3
        if t >= 5:
            lat_ms.append(dt)
    # steady-state per-step time of the B=1 pipeline (one full utterance)
    job.start_utterances()
    for _ in range(20):
        job.step()
    job.sync()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        job.step()
    job.sync()
    per_step_ms = (time.perf_counter() - t0) * 1e3 / n
    return dict(first_chunk_ms_p50=float(np.percentile(lat_ms, 50)), first_chunk_ms_p99=float(np.percentile(lat_ms, 99)),
                b1_ms_per_step=per_step_ms, b1_xrt=FRAME_S * 1e3 / per_step_ms, trials=trials)


def kernel_profile(eng, job, nsteps=6):
    """A few eager (non-graph) steps with per-launch HIP events on the launch stream."""
    job.start_utterances()
    for _ in range(40):  # mid-utterance context
        job.step()
    eng.sync()
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
    return rows, per_kernel, nsteps


def pmc_traffic(name):
    """HBM bytes per launch of `name` from the committed PMC passes (profiles/r01_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 calibration); None if that
    kernel was not profiled.  PMC counters cannot be read from inside this process."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["kernels"][name]["traffic_bytes_per_launch"]
    except Exception:
        return None


def roofline_of(per_kernel):
    name, k = max(per_kernel.items(), key=lambda kv: kv[1]["total_ms"])
    avg_s = k["total_ms"] / k["count"] * 1e-3
    gbs = k["bytes"] / k["count"] / avg_s / 1e9
    tfs = k["flops"] / k["count"] / avg_s / 1e12
    if tfs / MFMA_F32_PEAK_TF > gbs / HBM_PEAK_GBS:
        return dict(kernel=name, bound="mfma", achieved=tfs, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                    frac=tfs / MFMA_F32_PEAK_TF, traffic=pmc_traffic(name), avg_us=avg_s * 1e6, launches=k["count"],
                    algorithmic_bytes_per_launch=k["bytes"] / k["count"], flops_per_launch=k["flops"] / k["count"])
    return dict(kernel=name, bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                traffic=pmc_traffic(name), avg_us=avg_s * 1e6, launches=k["count"],
                algorithmic_bytes_per_launch=k["bytes"] / k["count"], flops_per_launch=k["flops"] / k["count"])


def cpu_baseline(args, cfg, W, nsteps):
    """The numpy oracle (a port of the reference algorithm, see oracle/np_oracle.py) timed on the
    host cores for a bounded sample of the same workload."""
    from oracle import np_oracle as O

    B = args.batch
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    rng = np.random.default_rng(1)
    st = lm.init_state(B, args.voice_len + args.text_len + nsteps + 1)
    lm.prefill(st, np.repeat((rng.standard_normal((1, args.voice_len, lm.D)) * 0.1).astype(np.float32), B, 0))
    lm.prefill(st, lm.embed_text(rng.integers(0, cfg.flow_lm.lookup_table.n_bins, (B, args.text_len))))
    ms = dec.init_state(B, nsteps)
    x = np.full((B, lm.ldim), np.nan, np.float32)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        noise = (rng.standard_normal((B, lm.ldim)) * args.temp ** 0.5).astype(np.float32)
        x, _, _ = lm.decode_step(st, x, noise, 1, float("inf"))
        dec.decode(ms, x)
    dt = time.perf_counter() - t0
    out = dict(value=B * nsteps * FRAME_S / dt, unit="audio-seconds/sec", cores=os.cpu_count(), kind="port",
               sample=f"{nsteps} decode steps (LM + Mimi) of batch {B} after voice+text prefill, numpy/BLAS oracle, "
                      f"{dt:.1f} s wall")
    # the reference's own operating point (SURVEY 8d setting (i)): batch 1, one BLAS thread (torch.set_num_threads(1),
    # tts_model.py:49); LM step and codec frame timed back to back, i.e. without the reference's two-thread overlap
    try:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1):
            st1 = lm.init_state(1, args.voice_len + args.text_len + 9)
            lm.prefill(st1, (rng.standard_normal((1, args.voice_len + args.text_len, lm.D)) * 0.1).astype(np.float32))
            ms1 = dec.init_state(1, 8)
            x1 = np.full((1, lm.ldim), np.nan, np.float32)
            t1 = time.perf_counter()
            for _ in range(8):
                x1, _, _ = lm.decode_step(st1, x1, None, 1, float("inf"))
                dec.decode(ms1, x1)
            d1 = time.perf_counter() - t1
        out["batch1_one_thread"] = dict(value=8 * FRAME_S / d1, unit="audio-seconds/sec", cores=1,
                                        sample=f"8 decode steps (LM + Mimi) of batch 1, one BLAS thread, {d1:.2f} s wall")
    except Exception as e:  # threadpoolctl missing: report only the all-cores figure
        out["batch1_one_thread"] = dict(error=str(e))
    return out


def main():
    args = parse()
    # Tile choices are pinned to the table measured on an MI355X and committed with the profiles, so that the kernel
    # names in this run, in profiles/*_kernel_stats.csv and in profiles/*_pmc_traffic.json refer to the same
    # configurations (the autotuner's near-ties otherwise flip between runs).  Shapes missing from the table are
    # tuned live and appended; PTTS_TUNE_CACHE= (empty) tunes everything live.
    os.environ.setdefault("PTTS_TUNE_CACHE", os.path.join(REPO, "profiles", "tune_cache_mi355x.txt"))
    if not os.environ["PTTS_TUNE_CACHE"]:
        del os.environ["PTTS_TUNE_CACHE"]
    from pocket_tts_amd import parallel

    rank, local, world = parallel.env_ranks()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    dist = parallel.init_distributed("nccl", dev)  # RCCL; only barriers + scalar reductions

    from pocket_tts_amd.config import named_config
    from pocket_tts_amd.engine import Engine
    from pocket_tts_amd.weights import generate_state_dict

    cfg = named_config(args.config)
    W = generate_state_dict(cfg, 0)
    eng = Engine(cfg, W, dev, quantize_groups={"attention", "ffn"} if args.quantize else None)
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
    barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.sync()
    ev_ms = eng.timer_stop_ms()
    barrier()
    wall = time.perf_counter() - t0
    audio_total, wall = parallel.job_throughput(args.batch * args.steps * FRAME_S, wall, dist, dev)
    audio_total *= wall  # job_throughput returns units/s; keep the totals explicit below

    out = None
    if rank == 0:
        audio_s = audio_total
        out = {
            "metric": "audio-seconds/sec (xRT), 100M en model, whole job over all GPUs",
            "value": audio_s / wall,
            "unit": "audio-seconds/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 weights (FlowLM attention+ffn), f32 activations/accumulate" if args.quantize else "f32",
            "data": "synthetic (seeded weights, voice KV, token ids; fixed-length utterances, EOS stop disabled)",
            "config": {
                "workload": f"{args.config}: batch {args.batch} concurrent utterances/GPU, voice KV {args.voice_len} + "
                            f"text {args.text_len} tokens, {args.frames} frames (10 s) each, temp {args.temp}, "
                            f"lsd_decode_steps 1; per utterance: state clone + text prefill + FlowLM step + Mimi "
                            f"decode per frame, hipGraph per FlowLM step and per codec frame on two streams (step t+1 overlaps frame t), PCM written straight into pinned host memory",
                "batch_per_gpu": args.batch,
                "parallelism": f"replicas x{world} (no collective on the data path)",
            },
            "xrt_per_gpu": audio_s / wall / world,
            "stream_event_ms_per_step": ev_ms / args.steps,
        }
        if not args.no_profile:
            rows, per_kernel, nst = kernel_profile(eng, job)
            out["roofline"] = roofline_of(per_kernel)
            tot = sum(r["total_ms"] for r in rows)
            out["kernel_ms_per_step"] = {k: round(v["total_ms"] / nst, 4) for k, v in
                                         sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_ms"])}
            out["kernel_sum_ms_per_step"] = tot / nst
            # per label: [algorithmic MB per launch, HBM-side MB per launch from the committed PMC passes or null]
            out["kernel_mb_per_launch"] = {k: [round(v["bytes"] / v["count"] / 1e6, 2),
                                               (lambda t: None if t is None else round(t / 1e6, 2))(pmc_traffic(k))]
                                           for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_ms"])}
            # the same events grouped by call site (lm.qkv, seanet.convtr2, ...): us per launch, launches per step
            sites = {}
            for r in rows:
                k = sites.setdefault(r["site"] + " " + r["kernel"], [0, 0.0])
                k[0] += r["count"]
                k[1] += r["total_ms"]
            out["site_us_per_launch"] = {k: [round(v[1] / v[0] * 1e3, 2), round(v[0] / nst, 2)]
                                         for k, v in sorted(sites.items(), key=lambda kv: -kv[1][1])}
            # whole-step achieved fraction of the HBM roofline (SURVEY 8d bytes_step formula)
            ctx = args.voice_len + args.text_len + args.frames / 2
            L = cfg.flow_lm.transformer.num_layers
            bytes_step = (eng.lm_weight_bytes() + eng.mimi_weight_bytes()
                          + args.batch * (8 * L * ctx * 1024 + 2.18e6 + 68e3))
            out["step_hbm_roofline"] = {"algorithmic_bytes_per_step": bytes_step,
                                        "bound_us_at_8TBs": bytes_step / 8e12 * 1e6,
                                        "frac": bytes_step / 8e12 / (wall / args.steps)}
        if not args.no_latency and world == 1:
            out["latency_b1"] = first_chunk_latency(eng, args)
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, cfg, W, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    job = None
    eng.close()


if __name__ == "__main__":
    main()
