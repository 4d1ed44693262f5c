"""Drop-in name: `import pocket_tts` resolves to the MI355X implementation (`pocket_tts_amd`).

The reference package exports exactly `TTSModel` and `export_model_state` (`pocket_tts/__init__.py:6-19`, enforced by
its `tests/test_python_api.py:8-26`); so does this one.  Code written against the reference -
`from pocket_tts import TTSModel`, `pocket_tts.main:cli_app`, `python -m pocket_tts generate ...` - runs unchanged.
"""

__all__ = ["TTSModel", "export_model_state"]


def __getattr__(name):
    if name in __all__:
        import pocket_tts_amd

        return getattr(pocket_tts_amd, name)
    raise AttributeError(name)
