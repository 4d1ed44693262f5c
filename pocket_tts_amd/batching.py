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

    __slots__ = ("req", "tokens", "voice", "gen", "fae", "start", "eos_step", "n_emit", "routed", "last")

    def __init__(self, req, tokens, voice, gen, fae, last):
        self.req, self.tokens, self.voice, self.gen, self.fae, self.last = req, tokens, voice, gen, fae, last
        self.start = None      # global step of its first FlowLM step
        self.eos_step = None   # local step of the first EOS flag
        self.n_emit = None     # frames to keep, known when the row leaves
        self.routed = 0        # frames handed to the request so far


class ContinuousBatcher:
    def __init__(self, model, slots: int = 16, capacity: int = 1024, pcm_format: str = "f32", noise_seed: int = 0):
        """`capacity`: KV positions per slot (voice + text + generated frames of one chunk must fit)."""
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
                                 mode="hostsync", pcm_i16=(pcm_format == "i16"))
        self.pipe.restart()
        self.slot: list = [None] * slots                 # running _Job per slot
        self.history = [collections.deque(maxlen=4) for _ in range(slots)]  # jobs whose frames may still be in flight
        self.waiting: collections.deque = collections.deque()
        self.g = 0            # global step counter == pipe.t
        self.collected = 0    # frames [0, collected) have been routed
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
            if t_voice + len(ids) + gen + 1 > self.capacity:
                raise ValueError(f"request needs {t_voice + len(ids) + gen + 1} KV positions; slot capacity is {self.capacity}")
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
        from .tts_model import _state_current_end

        eng = self.eng
        while True:
            with self._lock:
                free = [b for b in range(self.B) if self.slot[b] is None]
                if not free or not self.waiting:
                    return
                job = self.waiting.popleft()
            b = free[0]
            one = None
            try:
                voice_st, t_voice = self.model._voice_lm_state(job.voice)  # device-resident voice, no host sync
                Tt = job.tokens.shape[1]
                one = eng.new_lm_state(1, t_voice + Tt)
                one.copy_from(voice_st)
                eng.lm_prefill(one, eng.embed_text(job.tokens))
                self.st.copy_row_from(b, one)       # KV rows, position, BOS as the pending input, row active
                eng.sync()
            except (ValueError, KeyError, IndexError, TypeError) as e:
                # a bad request (malformed voice state, capacity): fail THIS request, keep serving the others.  The job
                # is in no list any more, so it is notified here (its consumer used to block forever).
                self._end_request(job.req, e)
                continue
            finally:
                if one is not None:
                    one.close()
            # the slot's codec carries: zero on the codec stream, behind the frames already queued there
            self.ms.reset_row(b, self.pipe.s2)
            job.start = self.g
            self.slot[b] = job
            self.history[b].append(job)

    def _end_request(self, req, error: Exception | None = None):
        """final sentinel of a request (with `error`: the consumer's iteration raises it); drops its follow-up chunks"""
        with self._wake:
            if self._outstanding.pop(req.id, None) is None:
                return
            self._chain.pop(req.id, None)
        if error is not None:
            req.error = error
        req._q.put(None)

    def _route(self, frame: int):
        """hand the rows of decoded `frame` to their requests"""
        pipe = self.pipe
        pipe.done_event(frame).synchronize()
        pcm = pipe.pcm16_of(frame) if self.pcm_format == "i16" else pipe.pcm_of(frame)
        for b in range(self.B):
            for job in self.history[b]:
                if job.start is None or frame < job.start:
                    continue
                local = frame - job.start
                if job.n_emit is not None and local >= job.n_emit:
                    continue
                if local != job.routed:
                    continue  # belongs to another job of this slot
                job.req._q.put(pcm[b].clone())
                job.req.frames += 1
                job.routed += 1
        self._finish_done()

    def _finish_done(self):
        for b in range(self.B):
            for job in list(self.history[b]):
                if job.n_emit is not None and job.routed >= job.n_emit and job.req is not None:
                    req = job.req
                    job.req = None
                    self.history[b].remove(job)
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

    def step(self) -> bool:
        """One scheduler iteration: admit, FlowLM step g, route frame g-2, EOS decisions of step g, codec frame g.
        Returns False when there is nothing to run."""
        self._admit()
        if all(j is None for j in self.slot):
            self._drain()
            return bool(self.waiting)
        pipe, g = self.pipe, self.g
        pipe.lm_step_async()
        if g - 2 >= self.collected:  # before its pinned buffer is reused by frame g
            self._route(g - 2)
            self.collected = g - 1
        flags = pipe.wait_flags(g)
        for b, job in enumerate(self.slot):
            if job is None:
                continue
            job.eos_step, job.n_emit = eos_bookkeeping(g - job.start, job.gen, job.fae, job.eos_step, bool(flags[b].item()))
            if job.n_emit is not None:
                if job.eos_step is None:
                    logger.warning("Maximum generation length reached without EOS, this very often indicates an error.")
                self.st.set_row_active(b, False)
                self.slot[b] = None
        pipe.decode_async(g)
        self.g += 1
        self._finish_done()
        return True

    def _drain(self):
        """route the frames still in flight (nothing is running)"""
        while self.collected < self.g:
            self._route(self.collected)
            self.collected += 1

    def run_until_idle(self):
        while self.step():
            pass
        self._drain()

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
                        self._drain()  # the last frames of the rows that just left; may queue a follow-up chunk
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
