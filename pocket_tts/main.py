"""`pocket_tts.main:cli_app` - the reference's console-script entry point (pyproject.toml:71-72) - on the MI355X engine."""

from pocket_tts_amd.main import build_parser, cli_app, write_wav_stream  # noqa: F401

if __name__ == "__main__":
    import sys

    sys.exit(cli_app())
