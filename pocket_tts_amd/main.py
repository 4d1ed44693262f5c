"""`generate` command with the reference's flags (pocket_tts/main.py:222-327) over the MI355X engine.

    python -m pocket_tts_amd generate --config cfg.yaml --voice voice.safetensors --text "..." \
        --output-path out.wav

Output: 24 kHz mono 16-bit WAV followed by 200 ms of silence (reference data/audio.py:69-72,99-107).
`serve` / `export-voice` (HTTP server, voice encoding) are outside the hot-path scope of this build.
"""

from __future__ import annotations

import argparse
import logging
import sys
import wave

logger = logging.getLogger("pocket_tts_amd")

DEFAULT_TEXT = ("Hello world. I am Kyutai's Pocket TTS. I'm fast enough to run on small CPUs. "
                "I hope you'll like me.")


def write_wav_stream(path, chunks, sample_rate: int) -> int:
    """fp32 chunks -> clamp, int16, raw frames; 200 ms of trailing silence.  Returns samples written."""
    out = sys.stdout.buffer if path == "-" else open(path, "wb")
    n = 0
    with out:
        w = wave.open(out, "wb")
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.setnframes(1_000_000_000)  # streaming: the length is not known up front
        for chunk in chunks:
            pcm = (chunk.clamp(-1, 1) * 32767).short().cpu().numpy()
            w.writeframesraw(pcm.tobytes())
            n += pcm.shape[0]
        silence = int(sample_rate * 0.2)
        w.writeframesraw(bytes(2 * silence))
        if path == "-":
            w._patchheader = lambda: None  # unseekable stream: keep the provisional header
        w.close()
    return n + silence


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="pocket-tts")
    sub = ap.add_subparsers(dest="command", required=True)
    g = sub.add_parser("generate", help="Generate speech")
    g.add_argument("--text", default=None, help="Text to generate ('-' reads stdin)")
    g.add_argument("--voice", default=None, help="Voice state (.safetensors exported by export_model_state)")
    g.add_argument("-q", "--quiet", action="store_true", help="Disable logging output")
    g.add_argument("--language", default=None)
    g.add_argument("--config", default=None, help="Path to a local model config .yaml")
    g.add_argument("--lsd-decode-steps", type=int, default=1)
    g.add_argument("--temperature", type=float, default=0.7)
    g.add_argument("--noise-clamp", type=float, default=None)
    g.add_argument("--eos-threshold", type=float, default=-4.0)
    g.add_argument("--frames-after-eos", type=int, default=None)
    g.add_argument("--output-path", default="./tts_output.wav")
    g.add_argument("--device", default="cuda:0")
    g.add_argument("--max-tokens", type=int, default=50)
    g.add_argument("--quantize", action="store_true")
    g.add_argument("--codec-bf16", action="store_true",
                   help="bf16 weights + activations in the Mimi codec (fp32 accumulate; not in the reference, ~44 dB SNR)")
    return ap


def cli_app(argv=None) -> int:
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.ERROR if args.quiet else logging.INFO)
    text = DEFAULT_TEXT if args.text is None else args.text
    if text == "-":
        text = sys.stdin.read()
    if not text.strip():
        logger.error("No input received from stdin.")
        return 1
    from .tts_model import TTSModel

    model = TTSModel.load_model(language=args.language, config=args.config, temp=args.temperature,
                                lsd_decode_steps=args.lsd_decode_steps, noise_clamp=args.noise_clamp,
                                eos_threshold=args.eos_threshold, quantize=args.quantize, codec_bf16=args.codec_bf16,
                                device=args.device)
    voice = args.voice if args.voice is not None else "alba"
    state = model.get_state_for_audio_prompt(voice)
    chunks = model.generate_audio_stream(state, text, frames_after_eos=args.frames_after_eos, max_tokens=args.max_tokens)
    write_wav_stream(args.output_path, chunks, model.sample_rate)
    if args.output_path != "-":
        logger.info("Results written in %s", args.output_path)
    return 0


if __name__ == "__main__":
    sys.exit(cli_app())
