"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that makes the *unmodified* reference (`/root/reference/asr/wenet`)
importable in the build container (torch 2.10, no torchaudio / whisper /
typeguard / wandb / git).  It registers empty stand-in modules for the
third-party packages the reference imports at module load but never uses on
the recognize_wav hot path, then puts `/root/reference/asr` on sys.path.

Used only by `oracle/gen_golden.py` (to generate tests/golden/*) and by the
CPU-side tests that pin `oracle/` against the real reference when
`/root/reference` exists.  `/root/reference` does not exist on the GPU box;
nothing that runs there may import this file.
"""
import os
import sys
import types
import typing

REFERENCE_ASR = os.environ.get("REVERB_REFERENCE_ASR", "/root/reference/asr")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ASR, "wenet"))


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: True


def _stub(name, **attrs):
    m = _Stub(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m


def install():
    """Make `import wenet` resolve to the reference. Idempotent."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ASR)
    if "wenet" in sys.modules and (getattr(sys.modules["wenet"], "__file__", "") or "").startswith(REFERENCE_ASR):
        return
    # the repo ships its own `wenet` compatibility package; make sure the name resolves to the reference
    for k in [k for k in sys.modules if k == "wenet" or k.startswith("wenet.")]:
        del sys.modules[k]
    import torch
    import torch.nn.modules.conv as C
    # squeezeformer/conv2d.py imports typing names from torch.nn.modules.conv
    for n in ("Union", "Optional"):
        if not hasattr(C, n):
            setattr(C, n, getattr(typing, n))
    if not hasattr(C, "Tensor"):
        C.Tensor = torch.Tensor
    for name in ("torchaudio", "torchaudio.compliance", "torchaudio.transforms",
                 "whisper", "typeguard", "wandb", "git"):
        if name not in sys.modules:
            _stub(name)
    _stub("torchaudio.compliance.kaldi", Tuple=typing.Tuple)
    _stub("whisper.tokenizer", LANGUAGES={"en": "english"})
    if REFERENCE_ASR not in sys.path:
        sys.path.insert(0, REFERENCE_ASR)
