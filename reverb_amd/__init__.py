"""reverb_amd -- MI355X-native (gfx950) engine for the Reverb-ASR `recognize_wav` hot path.

Public surface mirrors `asr/wenet/__init__.py:1-6` of the reference:
    from reverb_amd import load_model, ReverbASR, get_available_models, download_model
(`import wenet` resolves to the same objects through the `wenet/` compatibility package.)
Importing this package does not load librvb.so; creating a model does, and fails loudly when the
HIP extension or a GPU is missing -- there is no CPU fallback.
"""
__all__ = ["load_model", "ReverbASR", "get_available_models", "download_model"]


def __getattr__(name):
    if name in __all__:
        from . import reverb
        return getattr(reverb, name)
    raise AttributeError(name)
