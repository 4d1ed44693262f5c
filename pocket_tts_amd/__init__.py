"""MI355X-native implementation of the Pocket-TTS decode hot path (FlowLM step + Mimi codec decode).

Public surface mirrors the reference package (`pocket_tts/__init__.py:6-19`).
"""

__all__ = ["TTSModel", "export_model_state"]


def __getattr__(name):
    # lazy: importing the package must not require torch / a GPU (config + weights tooling is CPU-only)
    if name in __all__:
        from . import tts_model

        return getattr(tts_model, name)
    raise AttributeError(name)
