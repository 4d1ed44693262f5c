import sys

from pocket_tts_amd.main import cli_app

sys.exit(cli_app())
