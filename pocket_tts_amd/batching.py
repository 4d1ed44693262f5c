"""Continuous batching of utterances on one GPU.

The reference serves one request at a time: `/tts` starts a thread that runs `generate_audio_stream` for that
request alone (main.py:80-181) and the model is "not thread safe" (tts_model.py:491-492).  Here a fixed set of
SLOTS shares one batched FlowLM state and one batched codec state; the hipGraphs of the step are captured
once.  A request JOINS a free slot (its voice state + text are prefilled on a batch-1 state and copied into the
slot's row, the slot's codec carries are zeroed), is decoded in lock-step with the other slots with its own
position, EOS bookkeeping and frame count (tts_model.py:756-768 per row), and LEAVES when its loop would
break; the slot is then parked until the next request arrives.  Chunks reach the caller as they are decoded:
fp32 `[frame_samples]` tensors or, with `pcm_format="i16"`, the 16-bit samples of the WAV stream written by
the codec's last kernel (data/audio.py:79).

With temp == 0 every request reproduces `TTSModel.generate_audio` for the same text and voice (same frame
count, waveform equal up to fp32 summation order: batch tiles differ from batch-1 tiles).
"""

from __future__ import annotations

import collections
import logging
import queue
import threading

import torch

from .text import estimate_max_gen_len, prepare_text_prompt, split_into_best_sentences

logger = logging.getLogger(__name__)


def eos_bookkeeping(local_step: int, max_gen_len: int, frames_after_eos: int, eos_step, flag: bool):
    """Per-row restatement of the reference's generation loop (tts_model.py:756-775), evaluated after FlowLM step
    `local_step` (0-based) of an utterance: returns `(eos_step, n_emit)`.  `n_emit` is None while the row keeps
    running; otherwise it is the number of frames the utterance consists of: the latent of the step at which the
    reference's loop breaks is NOT decoded, and without EOS the row stops after `max_gen_len` frames."""
    if local_step >= max_gen_len:
        return eos_step, max_gen_len
    if flag and eos_step is None:
        eos_step = local_step
    if eos_step is not None and local_step >= eos_step + frames_after_eos:
        return eos_step, local_step
    return eos_step, None


def eos_bookkeeping_rows(local_step, max_gen_len, frames_after_eos, eos_step, n_emit, flags, rows=None):
    """`eos_bookkeeping` for many rows at once, in place on the int arrays `eos_step` / `n_emit` (-1 = None).
    `local_step`: scalar or array (rows of a continuous batch are at different steps of their utterances); rows whose
    `n_emit` is already decided, or that `rows` (bool mask) excludes, are left alone."""
    import numpy as np

    run = n_emit < 0
    if rows is not None:
        run = run & rows
    ls = np.broadcast_to(np.asarray(local_step), n_emit.shape)
    maxed = run & (ls >= max_gen_len)
    n_emit[maxed] = np.broadcast_to(max_gen_len, n_emit.shape)[maxed]
    run = run & ~maxed
    first = run & flags & (eos_step < 0)
    eos_step[first] = ls[first]
    done = run & (eos_step >= 0) & (ls >= eos_step + frames_after_eos)
    n_emit[done] = ls[done]


class Request:
    """One submitted text.  Iterate to receive chunks; `result()` waits for the whole waveform."""

    def __init__(self, rid: int):
        self.id = rid
        self._q: queue.Queue = queue.Queue()
        self.frames = 0
        self.error: Exception | None = None
        self._pending_chunks = 0  # text chunks not yet finished

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            yield item

    def result(self) -> torch.Tensor:
        parts = list(self)
        if not parts:
            return torch.zeros(0)
        return torch.cat(parts)


class _Job:
    """one text chunk of a request while it owns a slot"""

    __slots__ = ("req", "tokens", "voice", "gen", "fae", "start", "last")

    def __init__(self, req, tokens, voice, gen, fae, last):
        self.req, self.tokens, self.voice, self.gen, self.fae, self.last = req, tokens, voice, gen, fae, last
        self.start = None      # global step of its first FlowLM step


class ContinuousBatcher:
    def __init__(self, model, slots: int = 16, capacity: int = 1024, pcm_format: str = "f32", noise_seed: int = 0):
        """`capacity`: KV positions per slot (voice + text + generated frames of one chunk must fit).

        The scheduler never waits for the GPU on the step path: FlowLM step g and codec frame g are queued on the two
        streams of the "events" pipeline, and the EOS flags / PCM of a step are read from pinned memory `lag` <= nb steps
        later (as soon as the frame's event has fired).  A row therefore runs up to nb steps past the end of its
        utterance before its slot is parked (those frames are dropped); frame counts and waveforms are exactly those of
        the reference's loop (tts_model.py:756-768 per row)."""
        import numpy as np

        from .engine import StepPipeline

        if pcm_format not in ("f32", "i16"):
            raise ValueError("pcm_format must be 'f32' or 'i16'")
        self.model, self.eng, self.B = model, model.engine, slots
        self.capacity, self.pcm_format = capacity, pcm_format
        eng = self.eng
        self.st = eng.new_lm_state(slots, capacity)
        self.ms = eng.new_mimi_state(slots)
        if model.temp > 0:
            self.st.set_noise(model.temp, noise_seed)
        for b in range(slots):
            self.st.set_row_active(b, False)
        self.pipe = StepPipeline(eng, self.st, self.ms, None, model.lsd_decode_steps, float(model.eos_threshold),
                                 mode="events", pcm_i16=(pcm_format == "i16"))
        self.pipe.restart()
        self.slot: list = [None] * slots                 # running _Job per slot
        # per-slot bookkeeping of the job that owns the slot (arrays: one numpy pass per step instead of a Python loop)
        self.a_start = np.zeros(slots, np.int64)
        self.a_gen = np.zeros(slots, np.int64)
        self.a_fae = np.zeros(slots, np.int64)
        self.a_eos = np.full(slots, -1, np.int64)
        self.a_emit = np.full(slots, 0, np.int64)         # >= 0: no running job in the slot
        self.waiting: collections.deque = collections.deque()
        self.g = 0            # global step counter == pipe.t
        self.collected = 0    # frames [0, collected) have been read (EOS flags) and routed (PCM)
        self._next_id = 0
        self._lock = threading.Lock()
        self._wake = threading.Condition(self._lock)
        self._thread = None
        self._stop = False
        self._chain: dict = {}  # request id -> deque of follow-up chunks (run one after the other)
        self._outstanding: dict = {}  # request id -> Request, from submit() until its final sentinel: what _fail / stop notify
        self._failed: Exception | None = None
        self._closed = False

    # ---- submission (any thread) ---------------------------------------------------------------
    def submit(self, model_state: dict, text: str, frames_after_eos: int | None = None, max_tokens: int = 50) -> Request:
        """Same text handling as `generate_audio_stream` (tts_model.py:618-631): long texts are split into
        chunks that run one after the other, each from the voice state."""
        from .tts_model import _state_current_end

        m = self.model
        if self._failed is not None:
            raise RuntimeError(f"the batcher has stopped after an error: {self._failed}")
        if not text or not text.strip():
            raise ValueError("Text to generate cannot be empty")
        chunks = split_into_best_sentences(m.tokenizer.encode, m.tokenizer.sp, text, max_tokens,
                                           m.pad_with_spaces_for_short_inputs, m.remove_semicolons)
        t_voice = _state_current_end(model_state)
        jobs = []
        with self._lock:
            req = Request(self._next_id)
            self._next_id += 1
        for i, chunk in enumerate(chunks):
            _, guess = prepare_text_prompt(chunk, m.pad_with_spaces_for_short_inputs, m.remove_semicolons)
            fae = frames_after_eos if frames_after_eos is not None else (
                m.model_recommended_frames_after_eos if m.model_recommended_frames_after_eos is not None else guess + 2)
            ids = m.tokenizer.encode(chunk)
            gen = estimate_max_gen_len(len(ids), m.config.mimi.frame_rate)
            need = t_voice + len(ids) + gen + self.pipe.nb + 2  # a row runs up to nb steps past its end before it is parked
            if need > self.capacity:
                raise ValueError(f"request needs {need} KV positions; slot capacity is {self.capacity}")
            jobs.append(_Job(req, torch.tensor(ids, dtype=torch.long)[None, :], model_state, gen, fae, i == len(chunks) - 1))
        req._pending_chunks = len(jobs)
        with self._wake:
            if self._failed is not None or self._closed:
                raise RuntimeError("the batcher is closed" if self._failed is None else
                                   f"the batcher has stopped after an error: {self._failed}")
            self._outstanding[req.id] = req
            self.waiting.append(jobs[0])
            if len(jobs) > 1:
                self._chain[req.id] = collections.deque(jobs[1:])
            self._wake.notify_all()
        return req

    # ---- scheduler (one thread) ----------------------------------------------------------------
    def _admit(self):
        """Waiting jobs join the free slots.  Jobs with the same voice state and token count are prefilled as ONE batch
        (voice KV cloned from its device-resident copy, one GEMM pass for the group) and dealt to their slots; nothing
        synchronises with the host."""
        eng = self.eng
        with self._lock:
            free = [b for b in range(self.B) if self.slot[b] is None and self.a_emit[b] >= 0]
            take = []
            while self.waiting and len(take) < len(free):
                take.append(self.waiting.popleft())
        if not take:
            return
        groups: dict = {}
        for job in take:
            groups.setdefault((id(job.voice), job.tokens.shape[1]), []).append(job)
        # slots: rows of one voice share its keys (KvPrefix) and the decode attention scores a shared prefix once per 4
        # NEIGHBOURING rows (attn_cascade_kernel), so a job prefers a 4-row group that holds only its own voice
        voice_of = {b: id(self.slot[b].voice) for b in range(self.B) if self.slot[b] is not None}

        def pick(vid):
            def score(b):
                mates = [voice_of[r] for r in range(b & ~3, min(self.B, (b & ~3) + 4)) if r in voice_of]
                return (any(v != vid for v in mates), -sum(v == vid for v in mates), b)
            b = min(free, key=score)
            free.remove(b)
            voice_of[b] = vid
            return b

        for (vid, _), jobs in sorted(groups.items(), key=lambda kv: kv[0][0]):
            try:
                self._admit_group(jobs, [pick(vid) for _ in jobs])
            except (ValueError, KeyError, IndexError, TypeError) as e:
                if len(jobs) == 1:
                    # a bad request (malformed voice state, capacity): fail THIS request, keep serving the others.  The
                    # job is in no list any more, so it is notified here (its consumer would block forever).
                    self._end_request(jobs[0].req, e)
                    continue
                for job in jobs:  # find the bad one(s): admit the group's members one by one
                    with self._lock:
                        b = next(b for b in range(self.B) if self.slot[b] is None and self.a_emit[b] >= 0)
                    try:
                        self._admit_group([job], [b])
                    except (ValueError, KeyError, IndexError, TypeError) as e1:
                        self._end_request(job.req, e1)

    def _admit_group(self, jobs, rows):
        eng, model = self.eng, self.model
        voice = model._voice_acquire(jobs[0].voice)  # device-resident voice, no host sync on a hit
        grp = None
        try:
            voice_st, t_voice = voice
            Tt = jobs[0].tokens.shape[1]
            grp = eng.new_lm_state(len(jobs), t_voice + Tt)
            grp.copy_from(voice_st)
            eng.lm_prefill(grp, eng.embed_text(torch.cat([j.tokens for j in jobs], dim=0)))
            for i, b in enumerate(rows):
                self.st.copy_row_from(b, grp, i)   # KV rows, position, BOS as the pending input, row active
            eng.sync()  # the group state is freed below; its clone kernels must have run
        finally:
            if grp is not None:
                grp.close()
            model._voice_release(voice)
        for job, b in zip(jobs, rows):
            # the slot's codec carries: zero on the codec stream, behind the frames already queued there
            self.ms.reset_row(b, self.pipe.s2)
            job.start = self.g
            self.slot[b] = job
            self.a_start[b], self.a_gen[b], self.a_fae[b] = self.g, job.gen, job.fae
            self.a_eos[b], self.a_emit[b] = -1, -1

    def _end_request(self, req, error: Exception | None = None):
        """final sentinel of a request (with `error`: the consumer's iteration raises it); drops its follow-up chunks"""
        with self._wake:
            if self._outstanding.pop(req.id, None) is None:
                return
            self._chain.pop(req.id, None)
        if error is not None:
            req.error = error
        req._q.put(None)

    def _process(self, frame: int):
        """EOS decisions of FlowLM step `frame` and the PCM of codec frame `frame`, for every slot whose job had started
        by then; rows whose loop breaks at this step leave their slot (the break-step frame is not emitted)."""
        import numpy as np

        pipe = self.pipe
        pipe.done_event(frame).synchronize()  # codec frame done => the FlowLM step's flags are on the host too
        q = frame % pipe.nb
        rows = (self.a_emit < 0) & (self.a_start <= frame)
        if not rows.any():
            return
        eos_bookkeeping_rows(frame - self.a_start, self.a_gen, self.a_fae, self.a_eos, self.a_emit,
                             pipe.flag[q].numpy() != 0, rows)
        # one copy of the frame out of the pinned ring (numpy memcpy: no intra-op thread team), rows are views of it
        pcm = torch.from_numpy((pipe.pcm16_of(frame) if self.pcm_format == "i16" else pipe.pcm_of(frame)).numpy().copy())
        for b in np.nonzero(rows)[0]:
            job = self.slot[b]
            if self.a_emit[b] < 0:
                job.req._q.put(pcm[b])
                job.req.frames += 1
                continue
            if self.a_eos[b] < 0:
                logger.warning("Maximum generation length reached without EOS, this very often indicates an error.")
            self.st.set_row_active(int(b), False)
            self.slot[b] = None
            self._finish(job)

    def _finish(self, job):
        req = job.req
        with self._wake:
            req._pending_chunks -= 1
            nxt = self._chain.get(req.id)
            if nxt:
                self.waiting.appendleft(nxt.popleft())
                if not nxt:
                    del self._chain[req.id]
            elif req._pending_chunks == 0:
                self._outstanding.pop(req.id, None)
                req._q.put(None)

    def _check_gpu_error(self):
        if self.st.error():  # synchronises the FlowLM stream: only called when idle / every 512 steps
            raise RuntimeError("libptts: a cooperative FlowLM kernel timed out; audio generated since the last check is invalid")

    def step(self) -> bool:
        """One scheduler iteration: admit, read the steps whose frames are complete, queue FlowLM step g + codec frame g.
        Returns False when there is nothing to run."""
        self._admit()
        pipe, nb = self.pipe, self.pipe.nb
        # mandatory for frame g - nb (its pinned buffers are about to be reused), opportunistic for the later ones
        while self.collected < self.g and (self.g - self.collected >= nb or pipe.done_event(self.collected).query()):
            self._process(self.collected)
            self.collected += 1
        if all(j is None for j in self.slot):
            self._drain()
            return bool(self.waiting)
        pipe.step()
        self.g += 1
        if self.g % 512 == 0:
            self._check_gpu_error()
        return True

    def _drain(self):
        """read the frames still in flight (nothing is queued behind them)"""
        while self.collected < self.g:
            self._process(self.collected)
            self.collected += 1

    def run_until_idle(self):
        while self.step():
            pass
        self._drain()
        self._check_gpu_error()

    # ---- background operation ------------------------------------------------------------------
    def start(self):
        def loop():
            torch.cuda.set_device(self.eng.device)
            while True:
                with self._wake:
                    if self._stop:
                        return
                    idle = not self.waiting and all(j is None for j in self.slot)
                try:
                    if idle:
                        if self.collected < self.g:
                            self._drain()  # the last frames of the rows that just left; may queue a follow-up chunk
                            self._check_gpu_error()
                        with self._wake:
                            if not self._stop and not self.waiting:
                                self._wake.wait(timeout=0.05)
                    else:
                        self.step()
                except Exception as e:  # forward to every waiting consumer, like the reference's result_queue errors
                    self._fail(e)
                    return

        self._stop = False
        self._thread = threading.Thread(target=loop, daemon=True, name="ptts-batcher")
        self._thread.start()

    def _fail(self, e: Exception):
        """the scheduler cannot continue: every outstanding request - waiting, admitted or mid-admission - gets the
        error, and later submit() calls raise"""
        logger.error("batcher failed: %s", e)
        with self._wake:
            self._failed = e
            reqs = list(self._outstanding.values())
            self.waiting.clear()
        for r in reqs:
            self._end_request(r, e)

    def stop(self):
        with self._wake:
            self._stop = True
            self._wake.notify_all()
        if self._thread is not None:
            self._thread.join()
            self._thread = None

    def close(self):
        """stops the scheduler; requests still outstanding receive an error instead of blocking their consumers"""
        self.stop()
        with self._wake:
            self._closed = True
            reqs = list(self._outstanding.values())
            self.waiting.clear()
        for r in reqs:
            self._end_request(r, RuntimeError("the batcher was closed before this request finished"))
        self.pipe.sync()
        self.pipe.close()
        self.st.close()
        self.ms.close()
