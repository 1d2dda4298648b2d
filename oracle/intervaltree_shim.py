"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in for the third-party package `intervaltree==3.1.0` (`/root/reference/diarization/requirements.txt:2`,
not installed here, no network) as far as the reference's word -> speaker join uses it, so that the UNMODIFIED
`/root/reference/diarization/assign_words2speakers.py:24-61` (`speaker_for_segment`) can be executed in this
container and pin `reverb_amd/bin/assign_words2speakers.py`.  What that function touches:

  * `IntervalTree(iterable of Interval)`   set semantics (equal intervals collapse), null intervals raise ValueError
  * `tree[start:stop]`                      = `tree.overlap(start, stop)`: the SET of intervals with
                                              `iv.begin < stop and iv.end > start`; empty when `start >= stop`
  * `for interval in tree`                  iteration over the set of intervals
  * `Interval(begin, end, data=None)`       a namedtuple (`iv[0]`, `iv[1]`, `.data`), hashed on (begin, end) only
  * `Interval.distance_to(other)`           0 when they overlap, else the gap between them

Restated from the published behaviour of intervaltree 3.1.0 (interval.py / intervaltree.py); the tree itself (an
augmented AVL tree there) is replaced by a linear scan, which returns the same sets.  Set ITERATION order is
CPython's hash order in both, so exact ties in the reference's `min` / `max` resolve the same way here as with the
real package only by accident: the pin (tests/test_words2speakers_pin.py) treats exact ties as "any tied answer".

`load_reference_module()` imports the reference file itself with this module registered as `intervaltree` and an
empty `pyannote.database.util`; nothing of the reference is copied.
"""
import importlib.util
import os
import sys
import types
from collections import namedtuple

REFERENCE_FILE = os.environ.get("REVERB_REFERENCE_W2S", "/root/reference/diarization/assign_words2speakers.py")


class Interval(namedtuple("IntervalBase", ["begin", "end", "data"])):
    __slots__ = ()

    def __new__(cls, begin, end, data=None):
        return super(Interval, cls).__new__(cls, begin, end, data)

    def overlaps(self, begin, end=None):
        if end is not None:
            return begin < self.end and end > self.begin
        try:
            return self.overlaps(begin.begin, begin.end)
        except AttributeError:
            return self.begin <= begin < self.end

    def is_null(self):
        return self.begin >= self.end

    def distance_to(self, other):
        if self.overlaps(other):
            return 0
        try:
            if self.begin < other.begin:
                return other.begin - self.end
            return self.begin - other.end
        except AttributeError:
            if self.end <= other:
                return other - self.end
            return self.begin - other

    def __hash__(self):
        return hash((self.begin, self.end))

    def __eq__(self, other):
        return self.begin == other.begin and self.end == other.end and self.data == other.data


class IntervalTree:
    def __init__(self, intervals=None):
        self.all_intervals = set(intervals) if intervals is not None else set()
        for iv in self.all_intervals:
            if iv.is_null():
                raise ValueError("IntervalTree: Null Interval objects not allowed in IntervalTree: {0}".format(iv))

    def overlap(self, begin, end=None):
        if end is None:
            begin, end = begin.begin, begin.end
        if begin >= end:
            return set()
        return {iv for iv in self.all_intervals if iv.begin < end and iv.end > begin}

    def at(self, p):
        return {iv for iv in self.all_intervals if iv.begin <= p < iv.end}

    def __getitem__(self, index):
        try:
            start, stop = index.start, index.stop
        except AttributeError:
            return self.at(index)
        if start is None:
            if stop is None:
                return set(self.all_intervals)
            start = min(iv.begin for iv in self.all_intervals)
        if stop is None:
            stop = max(iv.end for iv in self.all_intervals)
        return self.overlap(start, stop)

    def __iter__(self):
        return iter(self.all_intervals)

    def __len__(self):
        return len(self.all_intervals)


def reference_available() -> bool:
    return os.path.isfile(REFERENCE_FILE)


# The REAL package, when this machine happens to hold it (this image has intervaltree==3.1.0 under a conda
# python's site-packages, not importable by the interpreter the tests run in; it is pure Python and needs `sortedcontainers`,
# which is installed).  It is loaded from where it lies, by path -- nothing of it is copied, no sys.path change.
REAL_SITE = os.environ.get("REVERB_REAL_INTERVALTREE", "/opt/conda/lib/python3.9/site-packages")
_REAL = []


def real_intervaltree():
    """The genuine `intervaltree` module (3.1.0), or None.  With it the live pin runs the reference function against the
    very data structure it was written for; the stand-in classes above remain for machines without it (the GPU box)."""
    if _REAL:
        return _REAL[0]
    mod = None
    try:
        import intervaltree as mod          # installed for this interpreter after all
    except ImportError:
        init = os.path.join(REAL_SITE, "intervaltree", "__init__.py")
        meta = os.path.join(REAL_SITE, "intervaltree-3.1.0.dist-info")
        if os.path.isfile(init) and os.path.isdir(meta):
            saved = {n: m for n, m in sys.modules.items() if n == "intervaltree" or n.startswith("intervaltree.")}
            try:
                import sortedcontainers  # noqa: F401 -- the package's one dependency
                spec = importlib.util.spec_from_file_location("intervaltree", init, submodule_search_locations=[os.path.dirname(init)])
                mod = importlib.util.module_from_spec(spec)
                sys.modules["intervaltree"] = mod
                spec.loader.exec_module(mod)
            except Exception:
                mod = None
            finally:
                for n in [n for n in sys.modules if n == "intervaltree" or n.startswith("intervaltree.")]:
                    del sys.modules[n]
                sys.modules.update(saved)
    _REAL.append(mod)
    return mod


def tree_classes():
    """(IntervalTree, Interval, "real 3.1.0" | "stand-in"): the real package's classes when present, else this module's."""
    real = real_intervaltree()
    if real is not None:
        return real.IntervalTree, real.Interval, "real intervaltree " + getattr(real, "__version__", "3.1.0")
    return IntervalTree, Interval, "stand-in"


class _StandIns:
    """`intervaltree` = this module's classes, `pyannote.database.util.load_rttm` = `load_rttm`, for the span of a `with`."""
    NAMES = ("intervaltree", "pyannote", "pyannote.database", "pyannote.database.util")

    def __init__(self, load_rttm=None, real=True):
        self.load_rttm = load_rttm
        self.real = real            # hand the reference the real package's classes when this machine has them

    def __enter__(self):
        self.saved = {n: sys.modules.get(n) for n in self.NAMES}
        it = types.ModuleType("intervaltree")
        it.IntervalTree, it.Interval, _ = tree_classes() if self.real else (IntervalTree, Interval, None)
        sys.modules["intervaltree"] = it
        for n in self.NAMES[1:]:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
        sys.modules["pyannote.database.util"].load_rttm = self.load_rttm    # only the script's __main__ block uses it
        return self

    def __exit__(self, *exc):
        for n, m in self.saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def load_reference_module(real=True):
    """The reference's assign_words2speakers.py, executed unmodified against the real intervaltree (when present and `real`)
    or the stand-ins above."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_FILE)
    with _StandIns(real=real):
        spec = importlib.util.spec_from_file_location("_reference_assign_words2speakers", REFERENCE_FILE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod


def run_reference_script(rttm_path, ctm_path, stm_path, load_rttm):
    """The reference file run as `__main__` (its lines 64-89: read CTM + RTTM, build the tree, write the STM), unmodified.
    `load_rttm` stands in for `pyannote.database.util.load_rttm`: {uri: annotation with .itertracks(yield_label=True)}."""
    import runpy
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_FILE)
    argv = sys.argv
    try:
        sys.argv = [REFERENCE_FILE, rttm_path, ctm_path, stm_path]
        with _StandIns(load_rttm):
            runpy.run_path(REFERENCE_FILE, run_name="__main__")
    finally:
        sys.argv = argv
