import sys

from .main import cli_app

sys.exit(cli_app())
