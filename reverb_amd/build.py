"""Build librvb.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

`python -m reverb_amd.build` or `reverb_amd.build.build()`.  hipcc cross-compiles without a GPU.
Objects are cached under reverb_amd/csrc/_build and rebuilt when a source or header is newer.

Two shared objects come out:
  librvb.so        the PRODUCT: exports include/rvb.h + include/rvd.h and nothing else of the C ABI
  librvb_test.so   the same objects + csrc/test_api.hip (rvb_test_*: raw kernel / host-search hooks, csrc/test_api.h) and
                   engine.hip compiled with -DRVB_TEST_API; loaded by tests/ and scripts/ only (reverb_amd._lib.load_test)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librvb.so")
OUT_TEST = os.path.join(HERE, "librvb_test.so")
SOURCES = ["gemm.hip", "gemm2.hip", "attention.hip", "elementwise.hip", "softmax_topk.hip", "fbank.hip", "engine.hip", "diar.hip", "resnet.hip", "conv_gemm.hip", "conv_stream.hip", "conv_block.hip", "conv_row64.hip", "conv_s2.hip", "linkage.hip", "diar_engine.hip", "comm.hip", "search.cpp", "audio.cpp", "mp3.cpp"]
TEST_SOURCES = ["test_api.hip", ("engine.hip", "engine_testapi", ["-DRVB_TEST_API"])]      # (source, object stem, extra flags)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value", "-Wno-unused-variable"]


def _headers(path=None, seen=None):
    """The headers a source really includes (its `#include "..."` lines, followed recursively through csrc/ and include/): a change of
    mp3.h rebuilds the three files that read it, not gemm2.hip's seven minutes of template instantiations."""
    import re
    if path is None:
        hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
        hs.append(os.path.join(HERE, "..", "include", "rvb.h"))
        hs.append(os.path.join(HERE, "..", "include", "rvd.h"))
        return hs
    seen = set() if seen is None else seen
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return []
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(HERE, "..", "include"), os.path.join(HERE, "..")):
            cand = os.path.normpath(os.path.join(base, name))
            if os.path.isfile(cand):
                if cand not in seen:
                    seen.add(cand)
                    _headers(cand, seen)
                break
    return sorted(seen)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(item):
    src, stem, extra = item if isinstance(item, tuple) else (item, os.path.splitext(item)[0], [])
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    obj = os.path.join(bdir, stem + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _headers(path)):
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [HIPCC] + FLAGS + extra + lang + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    if force:
        for f in os.listdir(os.path.join(CSRC, "_build")) if os.path.isdir(os.path.join(CSRC, "_build")) else []:
            os.remove(os.path.join(CSRC, "_build", f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES) + len(TEST_SOURCES))) as ex:
        all_objs = list(ex.map(_compile, SOURCES + TEST_SOURCES))
    objs, test_objs = all_objs[:len(SOURCES)], all_objs[len(SOURCES):]
    engine_obj = objs[SOURCES.index("engine.hip")]
    for out, members in ((OUT, objs), (OUT_TEST, [o for o in objs if o != engine_obj] + test_objs)):
        if _stale(out, members):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + members + ["-lpthread", "-ldl"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", out)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
