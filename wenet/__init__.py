"""Compatibility package: `import wenet; wenet.load_model(...)` keeps working for users of the
reference (asr/wenet/__init__.py:1-6 re-exports the same four names)."""
from reverb_amd.reverb import ReverbASR, download_model, get_available_models, load_model  # noqa: F401
