"""Speaker diarization pipeline with the calling convention of `pyannote.audio.Pipeline`, as used by
the reference (/root/reference/diarization/infer_pyannote3.0.py:33-42):

    pipeline = Pipeline.from_pretrained(model_dir)        # :33
    pipeline.to(torch.device("cuda"))                      # :36
    annotation = pipeline("a.wav")                         # :40
    annotation.write_rttm(f)                               # :42

The two neural networks run in librvb.so's HIP kernels (reverb_amd/diar_engine.py, include/rvd.h);
this file is the host logic between and after them, restating pyannote.audio 3.x
`SpeakerDiarization.apply` with the hyper-parameters of pyannote/speaker-diarization-3.1, which
Revai/reverb-diarization-v1 inherits (fine-tuned segmentation model, same pipeline):
powerset decoding, speaker counting, overlap-excluding masks, centroid-linkage agglomerative
clustering (scipy, as pyannote itself uses), reconstruction, binarisation, RTTM.

pyannote.audio is a third-party dependency that is not vendored in /root/reference and not
installed here: the restatement is from its published algorithm and is
**parity-unpinned** against the package itself; the functions are pure numpy so that
tests/test_diarization_host.py can check them on hand-made inputs.
"""
from __future__ import annotations

import math
import os
from collections import namedtuple
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

POWERSET = [(), (0,), (1,), (2,), (0, 1), (0, 2), (1, 2)]      # 3 local speakers, at most 2 at once

DEFAULT_PARAMS = dict(
    clustering=dict(method="centroid", min_cluster_size=12, threshold=0.7045654963945799),
    segmentation=dict(min_duration_off=0.0),
    embedding_exclude_overlap=True,
)

# receptive field of the segmentation model (pyannote >= 3.1 `model.receptive_field`): 991 samples wide,
# one frame every 270 samples
FRAME_DURATION = 991 / 16000.0
FRAME_STEP = 270 / 16000.0


# ------------------------------------------------------------------------------------------------ result type
class Segment(namedtuple("Segment", ["start", "end"])):
    @property
    def duration(self):
        return self.end - self.start

    @property
    def middle(self):
        return 0.5 * (self.start + self.end)


class Annotation:
    """The slice of pyannote.core.Annotation the reference touches: itertracks / write_rttm / labels."""

    def __init__(self, uri: Optional[str] = None):
        self.uri = uri
        self._tracks: List[Tuple[Segment, object, str]] = []

    def add(self, segment: Segment, track, label):
        self._tracks.append((segment, track, label))

    def __setitem__(self, key, label):
        segment, track = key
        self.add(Segment(*segment), track, label)

    def __len__(self):
        return len(self._tracks)

    def __bool__(self):
        return True

    def itertracks(self, yield_label: bool = False):
        for seg, track, label in sorted(self._tracks, key=lambda x: (x[0].start, x[0].end, str(x[1]))):
            yield (seg, track, label) if yield_label else (seg, track)

    def labels(self):
        return sorted({label for _, _, label in self._tracks}, key=lambda x: (str(type(x)), x))

    def rename_labels(self, mapping: Dict) -> "Annotation":
        out = Annotation(self.uri)
        for seg, track, label in self._tracks:
            out.add(seg, track, mapping.get(label, label))
        return out

    def write_rttm(self, file):
        """pyannote.core.Annotation.write_rttm line format."""
        uri = self.uri if self.uri else "<NA>"
        if isinstance(uri, str) and " " in uri:
            raise ValueError(f'Space-separated RTTM file format does not allow file URIs containing spaces (got: "{uri}").')
        for segment, _, label in self.itertracks(yield_label=True):
            if isinstance(label, str) and " " in label:
                raise ValueError(f'Space-separated RTTM file format does not allow labels containing spaces (got: "{label}").')
            file.write(f"SPEAKER {uri} 1 {segment.start:.3f} {segment.duration:.3f} <NA> <NA> {label} <NA> <NA>\n")


def load_rttm(path: str) -> Dict[str, Annotation]:
    """pyannote.database.util.load_rttm: {uri: Annotation}, one track per line (track id = line index)."""
    out: Dict[str, Annotation] = {}
    with open(path) as f:
        for i, line in enumerate(f):
            parts = line.split()
            if not parts:
                continue
            if len(parts) < 8:
                raise ValueError(f"{path}:{i + 1}: not an RTTM line")
            uri, start, dur, spk = parts[1], float(parts[3]), float(parts[4]), parts[7]
            out.setdefault(uri, Annotation(uri)).add(Segment(start, start + dur), i, spk)
    return out


# ------------------------------------------------------------------------------------------------ host stages
def powerset_to_multilabel(logp: np.ndarray) -> np.ndarray:
    """Powerset.to_multilabel(soft=False): argmax class -> 0/1 activity of the 3 local speakers."""
    mapping = np.zeros((len(POWERSET), 3), np.float32)
    for k, spk in enumerate(POWERSET):
        mapping[k, list(spk)] = 1.0
    return mapping[np.argmax(logp, axis=-1)]


def classes_to_multilabel(classes: np.ndarray) -> np.ndarray:
    """Same as powerset_to_multilabel for already-argmaxed class indices (uint8 [chunks, frames])."""
    mapping = np.zeros((len(POWERSET), 3), np.float32)
    for k, spk in enumerate(POWERSET):
        mapping[k, list(spk)] = 1.0
    return mapping[classes]


SEGMENT_PRECISION = 1e-6      # pyannote.core.segment.SEGMENT_PRECISION


def closest_frame(t: float, start: float = 0.0) -> int:
    """pyannote.core.SlidingWindow.closest_frame for the model's receptive field."""
    return int(np.rint((t - start - 0.5 * FRAME_DURATION) / FRAME_STEP))


def _aggregate_cols(cols, chunk_step: float, chunk_duration: float, skip_average: bool, missing: float, epsilon: float) -> np.ndarray:
    """overlap-add of per-class (num_chunks, frames) score planes onto the file-level frame grid -> (num_frames, classes)."""
    num_chunks, nf = cols[0].shape
    num_frames = closest_frame(chunk_duration + (num_chunks - 1) * chunk_step + 0.5 * FRAME_DURATION) + 1
    starts = np.rint((np.arange(num_chunks) * chunk_step) / FRAME_STEP).astype(np.int64)     # closest_frame(c*step + dur/2)
    total = int(max(num_frames, starts[-1] + nf))
    idx = (starts[:, None] + np.arange(nf)[None, :]).ravel()
    agg = np.zeros((total, len(cols)), np.float64)
    cnt = np.zeros((total, len(cols)), np.float64)
    full_cnt = None
    for k, plane in enumerate(cols):
        col = plane.ravel()
        ok = ~np.isnan(col)
        if ok.all():
            agg[:, k] = np.bincount(idx, weights=col, minlength=total)
            if full_cnt is None:
                full_cnt = np.bincount(idx, minlength=total)
            cnt[:, k] = full_cnt
        else:
            agg[:, k] = np.bincount(idx[ok], weights=col[ok], minlength=total)
            cnt[:, k] = np.bincount(idx[ok], minlength=total)
    avg = agg if skip_average else agg / np.maximum(cnt, epsilon)
    avg[cnt == 0.0] = missing
    return avg[:num_frames]


def aggregate(scores: np.ndarray, chunk_step: float, chunk_duration: float, skip_average: bool = False,
              missing: float = 0.0, epsilon: float = 1e-12) -> np.ndarray:
    """pyannote.audio Inference.aggregate(hamming=False, warm_up=(0,0)): overlap-add of per-chunk scores
    (num_chunks, frames, classes), NaN entries ignored, onto the file-level frame grid.  The per-chunk
    `+=` loop of the original is done as one weighted bincount per class (same sums, fp64)."""
    return _aggregate_cols([scores[:, :, k] for k in range(scores.shape[2])], chunk_step, chunk_duration, skip_average, missing, epsilon)


def speaker_count(binarized: np.ndarray, chunk_step: float, chunk_duration: float) -> np.ndarray:
    """SpeakerDiarizationMixin.speaker_count: rounded average number of active speakers per frame."""
    count = aggregate(np.sum(binarized, axis=-1, keepdims=True), chunk_step, chunk_duration)
    return np.rint(count).astype(np.uint8)


_NSPK = np.array([len(p) for p in POWERSET], np.float64)


def _lut() -> np.ndarray:
    lut = np.zeros((len(POWERSET), 3), bool)
    for k, spk in enumerate(POWERSET):
        lut[k, list(spk)] = True
    return lut


class ClassRuns:
    """Run-length view of the argmax powerset classes (uint8 [chunks, frames]): the class of a window is
    piecewise constant (a handful of runs per 589 frames), so every per-frame overlap-add of the host stages
    becomes a difference array over run boundaries + one cumulative sum.  Sums of 0/1 activities are small
    integers, exact in float64 either way, so the results are identical to the per-frame formulation."""

    def __init__(self, classes: np.ndarray, chunk_step: float, chunk_duration: float):
        classes = np.ascontiguousarray(classes)
        W, nf = classes.shape
        self.num_chunks, self.nf = W, nf
        edge = np.empty((W, nf), bool)
        edge[:, 0] = True
        np.not_equal(classes[:, 1:], classes[:, :-1], out=edge[:, 1:])
        self.w, self.a = np.nonzero(edge)                       # run r: frames [a, b) of chunk w, class c
        self.c = classes[self.w, self.a]
        self.b = np.empty_like(self.a)
        if self.a.size:
            self.b[:-1] = self.a[1:]
            self.b[-1] = nf
            self.b[:-1][self.w[1:] != self.w[:-1]] = nf
        # file-level frame grid, as Inference.aggregate lays it out (see _aggregate_cols)
        self.num_frames = closest_frame(chunk_duration + (W - 1) * chunk_step + 0.5 * FRAME_DURATION) + 1 if W else 0
        self.starts = np.rint((np.arange(W) * chunk_step) / FRAME_STEP).astype(np.int64)
        self.total = int(max(self.num_frames, self.starts[-1] + nf)) if W else 0
        self.lo = self.starts[self.w] + self.a                  # file-level frame range [lo, hi) of every run
        self.hi = self.starts[self.w] + self.b

    def overlap_add(self, sel: np.ndarray, col: np.ndarray, ncol: int, weights: Optional[np.ndarray] = None) -> np.ndarray:
        """sum over the runs `sel` of weights * [lo <= t < hi] into column `col` -> float64 (total, ncol).
        The difference array is laid out column by column (each column's running sum is a contiguous pass; round 5: 3 h of
        audio took 90 ms in the row-major form) and handed back as the transposed view, which is also the layout _top_count
        wants; one bincount carries the +w at run starts and the -w at run ends."""
        stride = self.total + 1
        lo, hi = self.lo[sel], self.hi[sel]
        base = col * stride
        w = np.ones(lo.size, np.float64) if weights is None else np.asarray(weights, np.float64)
        d = np.bincount(np.concatenate([lo + base, hi + base]), weights=np.concatenate([w, -w]), minlength=stride * ncol)
        d = d.reshape(ncol, stride)
        np.cumsum(d, axis=1, out=d)
        return d[:, :self.total].T

    def coverage(self) -> np.ndarray:
        """number of chunks that cover each file-level frame: float64 (total,)."""
        d = np.bincount(self.starts, minlength=self.total + 1) - np.bincount(self.starts + self.nf, minlength=self.total + 1)
        return np.cumsum(d)[:self.total].astype(np.float64)

    def histogram(self) -> np.ndarray:
        """frames per powerset class of every chunk: int64 (chunks, 7)."""
        k = len(POWERSET)
        return np.bincount(self.w * k + self.c, weights=self.b - self.a, minlength=self.num_chunks * k) \
            .astype(np.int64).reshape(self.num_chunks, k)


def speaker_count_from_classes(classes: np.ndarray, chunk_step: float, chunk_duration: float, runs: Optional[ClassRuns] = None) -> np.ndarray:
    """speaker_count(classes_to_multilabel(classes)) without materialising the multilabel tensor."""
    r = runs or ClassRuns(classes, chunk_step, chunk_duration)
    if r.num_chunks == 0:
        return np.zeros((0, 1), np.uint8)
    spk = np.nonzero(r.c > 0)[0]
    agg = r.overlap_add(spk, 0, 1, _NSPK[r.c[spk]])
    cnt = r.coverage()[:, None]
    avg = agg / np.maximum(cnt, 1e-12)
    avg[cnt == 0.0] = 0.0
    return np.rint(avg[:r.num_frames]).astype(np.uint8)


def class_histogram(classes: np.ndarray, runs: Optional[ClassRuns] = None) -> np.ndarray:
    """frames per powerset class of every chunk: int64 (chunks, 7)."""
    return (runs or ClassRuns(classes, 1.0, 10.0)).histogram()


def active_from_classes(classes: np.ndarray, runs: Optional[ClassRuns] = None) -> np.ndarray:
    """bool (chunks, 3): the local speaker is active in at least one frame (= np.sum(binarized, axis=1) > 0)."""
    return (class_histogram(classes, runs) @ _lut().astype(np.int64)) > 0


def embedding_items_from_classes(classes: np.ndarray, exclude_overlap: bool, min_num_samples: int = 400,
                                 window_samples: int = 160000, runs: Optional[ClassRuns] = None):
    """The (chunk, local speaker) pairs that are active at all, and their pooling masks as
    embedding_masks() would build them: -> (chunk_idx, speaker_idx, masks float32 [n, frames])."""
    num_chunks, nf = classes.shape
    lut = _lut()
    counts = class_histogram(classes, runs)
    total = counts @ lut.astype(np.int64)                         # frames each local speaker is active in
    alone = counts[:, 1:4]                                        # ... active alone (classes {s} are 1..3)
    wi, si = np.nonzero(total > 0)
    use_clean = np.zeros(wi.shape, bool)
    if exclude_overlap:
        min_num_frames = math.ceil(nf * min_num_samples / window_samples)
        use_clean = alone[wi, si] > min_num_frames
    masks = np.empty((wi.size, nf), np.float32)
    for s in range(3):                                            # per (speaker, mask kind): OR of <= 3 class comparisons
        for clean in (False, True):
            grp = np.nonzero((si == s) & (use_clean == clean))[0]
            if grp.size:
                rows = classes[wi[grp]]
                on = rows == (s + 1)
                if not clean:
                    for k in np.nonzero(lut[:, s])[0]:
                        if k != s + 1:
                            on |= rows == k
                masks[grp] = on
    return wi, si, masks


def reconstruct_from_classes(classes: np.ndarray, hard_clusters: np.ndarray, count: np.ndarray, chunk_step: float,
                             chunk_duration: float, runs: Optional[ClassRuns] = None) -> np.ndarray:
    """reconstruct(classes_to_multilabel(classes), hard_clusters, count, ...) on the run-length view: a run of class c
    in chunk w adds 1 to every cluster that one of the speakers of c is mapped to (once per cluster: the original takes
    the max over the local speakers of a chunk that share a cluster)."""
    r = runs or ClassRuns(classes, chunk_step, chunk_duration)
    k = max(int(np.max(hard_clusters)) + 1, 1) if hard_clusters.size else 1
    lut = _lut()
    hw = hard_clusters[r.w]                                       # (runs, 3) cluster of each local speaker of the run's chunk
    on = lut[r.c] & (hw >= 0)                                     # (runs, 3) speaker talks in the run and has a cluster
    on[:, 1] &= ~(on[:, 0] & (hw[:, 0] == hw[:, 1]))
    on[:, 2] &= ~((on[:, 0] & (hw[:, 0] == hw[:, 2])) | (on[:, 1] & (hw[:, 1] == hw[:, 2])))
    ri, sp = np.nonzero(on)
    act = r.overlap_add(ri, hw[ri, sp], k)[:r.num_frames]
    return _top_count(act, count)


def _top_count(act: np.ndarray, count: np.ndarray) -> np.ndarray:
    """to_diarization: per frame, the `count[t]` most active clusters (ties: lowest cluster index first) are on."""
    max_per_frame = int(np.max(count)) if count.size else 0
    if act.shape[1] < max_per_frame:
        act = np.pad(act, ((0, 0), (0, max_per_frame - act.shape[1])))
    n = min(act.shape[0], count.shape[0])
    act, cnt = act[:n], count[:n, 0].astype(np.int64)
    ncl = act.shape[1]
    if ncl <= 8:
        # rank of cluster k in the stable descending order = clusters that beat it + equal ones with a lower index
        cols = np.ascontiguousarray(act.T)                         # (clusters, frames): contiguous per cluster
        cnt16 = cnt.astype(np.int16)
        out = np.empty(cols.shape, act.dtype)
        for k in range(ncl):
            rank = np.zeros(n, np.int16)
            for j in range(ncl):
                if j != k:
                    rank += (cols[j] >= cols[k]) if j < k else (cols[j] > cols[k])
            out[k] = rank < cnt16
        return np.ascontiguousarray(out.T)
    order = np.argsort(-act, axis=-1, kind="stable")
    binary = np.zeros_like(act)
    ranks = np.arange(ncl)[None, :] < cnt[:, None]
    np.put_along_axis(binary, order, ranks.astype(act.dtype), axis=1)
    return binary


def embedding_masks(binarized: np.ndarray, exclude_overlap: bool, min_num_samples: int = 400,
                    window_samples: int = 160000) -> np.ndarray:
    """SpeakerDiarization.get_embeddings mask choice: frames where the speaker talks alone, unless that
    leaves too few frames, in which case all of the speaker's frames.  -> (chunks, speakers, frames)."""
    num_chunks, nf, nspk = binarized.shape
    masks = np.nan_to_num(binarized, nan=0.0).astype(np.float32)
    if exclude_overlap:
        min_num_frames = math.ceil(nf * min_num_samples / window_samples)
        clean = masks * (np.sum(masks, axis=2, keepdims=True) < 2)
        use_clean = np.sum(clean, axis=1, keepdims=True) > min_num_frames
        masks = np.where(use_clean, clean, masks)
    return np.ascontiguousarray(np.transpose(masks, (0, 2, 1)))


def _fcluster_distance(dendrogram: np.ndarray, t: float) -> np.ndarray:
    """scipy.cluster.hierarchy.fcluster(Z, t, criterion="distance") without its O(n) Python-side validation of Z (4-5 ms of the
    7 ms the host spends between the GPU's dendrogram and the assignment, at 9 000 embeddings): the same compiled routine on
    the same array.  The dendrogram comes from our own kernel (pinned to scipy's by the GPU tests); anything unexpected about
    the private module falls back to the public function."""
    from scipy.cluster.hierarchy import fcluster
    try:
        from scipy.cluster import _hierarchy
        z = np.ascontiguousarray(dendrogram, dtype=np.float64)
        n = z.shape[0] + 1
        if z.ndim != 2 or z.shape[1] != 4 or n < 2:
            raise ValueError
        labels = np.zeros((n,), dtype="i")
        _hierarchy.cluster_dist(z, labels, float(t), int(n))
        return labels
    except Exception:
        return fcluster(dendrogram, t, criterion="distance")


def cluster_embeddings(embeddings: np.ndarray, binarized: np.ndarray, threshold: float, min_cluster_size: int,
                       method: str = "centroid", num_clusters: Optional[int] = None, min_clusters: Optional[int] = None,
                       max_clusters: Optional[int] = None, linkage_fn=None, active: Optional[np.ndarray] = None):
    """pyannote.audio.pipelines.clustering.AgglomerativeClustering.__call__ (metric cosine):
    embeddings (chunks, speakers, dim), binarized (chunks, frames, speakers) -> hard_clusters (chunks, speakers),
    centroids (clusters, dim).  `linkage_fn(unit_vectors) -> Z` replaces scipy's linkage (the pipeline passes the
    GPU implementation, DiarEngine.centroid_linkage, which returns the same dendrogram)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import cdist

    num_chunks, nspk, dim = embeddings.shape
    if active is None:                                   # (chunks, speakers): speaker active in at least one frame
        active = np.any(binarized > 0, axis=1)
    valid = ~np.isnan(embeddings.sum(axis=2))            # an embedding is valid iff none of its entries is NaN
    chunk_idx, speaker_idx = np.where(active * valid)
    train = np.ascontiguousarray(embeddings[chunk_idx, speaker_idx], dtype=np.float32)     # float32, as pyannote holds them
    n = train.shape[0]
    if n == 0:
        return np.zeros((num_chunks, nspk), np.int64), np.zeros((1, dim))
    lo = num_clusters or min_clusters or 1
    lo = max(1, min(n, lo))
    hi = num_clusters or max_clusters or n
    hi = max(1, min(n, hi))
    if hi < 2:
        return np.zeros((num_chunks, nspk), np.int64), np.mean(train, axis=0, keepdims=True)

    mcs = min(min_cluster_size, max(1, round(0.1 * n)))
    if n == 1:
        clusters = np.zeros((1,), np.int64)
    else:
        # pyannote normalises the float32 embeddings in place, scipy then works in fp64
        unit = (train / np.linalg.norm(train, axis=-1, keepdims=True)).astype(np.float64)
        if linkage_fn is not None and method == "centroid":
            dendrogram = linkage_fn(unit)
        else:
            dendrogram = linkage(unit, method=method, metric="euclidean")
        clusters = _fcluster_distance(dendrogram, threshold) - 1
        uniq, counts = np.unique(clusters, return_counts=True)
        large = uniq[counts >= mcs]
        target = num_clusters
        if len(large) < lo:
            target = lo
        elif len(large) > hi:
            target = hi
        if target is not None and len(large) != target:
            # walk the dendrogram away from the threshold until the number of large clusters fits
            dd = np.copy(dendrogram)
            dd[:, 2] = np.arange(n - 1)
            best_it, best_large = n - 1, 1
            for it in np.argsort(np.abs(dendrogram[:, 2] - threshold)):
                if dd[it, 3] < mcs:
                    continue
                cl = fcluster(dd, it, criterion="distance") - 1
                u, cts = np.unique(cl, return_counts=True)
                nl = int(np.sum(cts >= mcs))
                if abs(nl - target) < abs(best_large - target):
                    best_it, best_large = it, nl
                if nl == target:
                    break
            clusters = fcluster(dd, best_it, criterion="distance") - 1
            uniq, counts = np.unique(clusters, return_counts=True)
            large = uniq[counts >= mcs]
        if len(large) == 0:
            clusters[:] = 0
        else:
            small = uniq[counts < mcs]
            if len(small):
                large_c = np.vstack([np.mean(unit[clusters == k], axis=0) for k in large])
                small_c = np.vstack([np.mean(unit[clusters == k], axis=0) for k in small])
                nearest = np.argmin(cdist(large_c, small_c, metric="cosine"), axis=0)
                for s_i, l_i in enumerate(nearest):
                    clusters[clusters == small[s_i]] = large[l_i]
                _, clusters = np.unique(clusters, return_inverse=True)

    k = int(np.max(clusters)) + 1
    centroids = np.vstack([np.mean(train[clusters == i], axis=0) for i in range(k)]).astype(np.float64)
    # assign every (chunk, speaker) embedding to the most similar centroid: soft = 2 - cosine distance; only the
    # trained pairs have embeddings here (the others are NaN and are marked inactive by the caller)
    cn = centroids / np.linalg.norm(centroids, axis=1, keepdims=True)
    tn = unit if n > 1 else (train / np.linalg.norm(train, axis=1, keepdims=True)).astype(np.float64)
    sim = tn @ cn.T                                                              # = 1 - cdist(.., "cosine")
    hard = np.zeros((num_chunks, nspk), np.int64)
    hard[chunk_idx, speaker_idx] = np.argmax(sim, axis=1)
    rest = valid & ~active
    if rest.any():                                              # valid embeddings of inactive speakers (never trained on)
        ri, rs = np.nonzero(rest)
        e = embeddings[ri, rs].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            hard[ri, rs] = np.argmax(np.nan_to_num((e / np.linalg.norm(e, axis=1, keepdims=True)) @ cn.T, nan=-np.inf), axis=1)
    return hard, centroids


def reconstruct(segmentations: np.ndarray, hard_clusters: np.ndarray, count: np.ndarray, chunk_step: float,
                chunk_duration: float) -> np.ndarray:
    """SpeakerDiarization.reconstruct + to_diarization: (frames, clusters) 0/1 matrix in which the `count[t]`
    most active clusters of frame t are on."""
    num_chunks, nf, _ = segmentations.shape
    k = int(np.max(hard_clusters)) + 1
    # one contiguous plane per cluster.  pyannote fills absent (chunk, cluster) pairs with NaN and its aggregation skips
    # them; with skip_average=True that is the same as adding zeros (activities are >= 0), which keeps the fast path
    planes = [np.zeros((num_chunks, nf), np.float32) for _ in range(max(k, 1))]
    for kk in range(k):                              # max over the local speakers of a chunk mapped to cluster kk
        for sp in range(segmentations.shape[2]):
            sel = np.nonzero(hard_clusters[:, sp] == kk)[0]
            if sel.size:
                planes[kk][sel] = np.maximum(planes[kk][sel], segmentations[sel, :, sp])
    return _top_count(_aggregate_cols(planes, chunk_step, chunk_duration, True, 0.0, 1e-12), count)


def to_annotation(binary: np.ndarray, min_duration_off: float = 0.0, uri: Optional[str] = None) -> Annotation:
    """pyannote.audio.utils.signal.Binarize(onset=offset=0.5) on the 0/1 matrix: one track per speaker turn,
    turn boundaries at frame middles (a turn ends at the middle of the first inactive frame, or of the last
    frame when still active there)."""
    ann = Annotation(uri)
    n = binary.shape[0]
    if n == 0:
        return ann
    ts = np.arange(n) * FRAME_STEP + 0.5 * FRAME_DURATION
    for k in range(binary.shape[1]):
        on = binary[:, k] > 0.5
        d = np.diff(on.astype(np.int8))
        starts = list(np.nonzero(d == 1)[0] + 1)
        ends = list(np.nonzero(d == -1)[0] + 1)
        if on[0]:
            starts.insert(0, 0)
        if on[-1]:
            ends.append(n - 1)
        regions = [[ts[a], ts[b]] for a, b in zip(starts, ends)]
        if min_duration_off > 0.0 and regions:       # Annotation.support(collar): bridge short same-speaker gaps
            merged = [regions[0]]
            for s_, e_ in regions[1:]:
                if s_ - merged[-1][1] <= min_duration_off:
                    merged[-1][1] = max(merged[-1][1], e_)
                else:
                    merged.append([s_, e_])
            regions = merged
        for i, (s_, e_) in enumerate(regions):
            if e_ - s_ <= SEGMENT_PRECISION:         # pyannote's Annotation.__setitem__ silently drops empty segments
                continue                             # (a speaker active on the last frame only: start == end)
            ann.add(Segment(float(s_), float(e_)), f"{k}_{i}", k)
    return ann


# ------------------------------------------------------------------------------------------------ pipeline
class SpeakerDiarization:
    """`pipeline(file)` -> Annotation.  Built by Pipeline.from_pretrained."""

    def __init__(self, cfg: dict, segmentation_sd: dict, embedding_sd: dict, params: Optional[dict] = None, dtype: str = "bf16"):
        self.cfg = dict(cfg)
        self.params = {**DEFAULT_PARAMS, **(params or {})}
        self._seg_sd, self._emb_sd = segmentation_sd, embedding_sd
        self.dtype = dtype
        self.device_index: Optional[int] = None
        self._engine = None
        self._emb_fp8_scales = None          # fp8 trunk scales agreed on by the ranks of a sharded run (dist.share_emb_fp8_scales)
        self.timings: Dict[str, float] = {}

    # pyannote API --------------------------------------------------------------------------------
    def to(self, device):
        """Accepts torch.device / "cuda" / "cuda:1" / int.  A CPU device is refused: the networks only exist as
        HIP kernels (the reference falls back to CPU torch, infer_pyannote3.0.py:35; we do not)."""
        idx = 0
        if hasattr(device, "type"):
            if device.type != "cuda":
                raise RuntimeError("reverb_amd diarization runs on an MI355X only (no CPU fallback); got device " + str(device))
            idx = device.index or 0
        elif isinstance(device, str):
            if not device.startswith("cuda"):
                raise RuntimeError("reverb_amd diarization runs on an MI355X only (no CPU fallback); got device " + device)
            idx = int(device.split(":")[1]) if ":" in device else 0
        else:
            idx = int(device)
        if self._engine is not None and idx != self.device_index:
            self._engine.close(); self._engine = None
        self.device_index = idx
        return self

    def instantiate(self, params: dict):
        self.params = {**self.params, **params}
        return self

    @property
    def engine(self):
        if self._engine is None:
            from .diar_engine import DiarEngine
            self._engine = DiarEngine(self.cfg, self._seg_sd, self._emb_sd, dtype=self.dtype, device=self.device_index or 0)
            if self._emb_fp8_scales is not None:      # a rank that had no window when the scales were shared
                self._engine.set_emb_fp8_scales(self._emb_fp8_scales)
        return self._engine

    def _load(self, file) -> Tuple[np.ndarray, str]:
        """-> (int16 mono PCM at 16 kHz, uri).  pyannote's `Audio`: downmix to mono (mean), resample to the model's rate."""
        from . import audio as A
        if isinstance(file, dict):
            uri = file.get("uri")
            if "waveform" in file:
                w = file["waveform"]
                w = w.detach().cpu().numpy() if hasattr(w, "detach") else np.asarray(w)
                w = w.mean(axis=0) if w.ndim == 2 else w          # pyannote Audio: downmix to mono
                pcm = w if w.dtype == np.int16 else np.clip(np.rint(w * 32768.0), -32768, 32767).astype(np.int16)
                rate = int(file.get("sample_rate", 16000))
                if rate != 16000:
                    pcm = self.engine.resample(pcm, rate)
                return pcm, uri or "waveform"
            file = file["audio"]
        path = os.fspath(file)
        pcm, info = A.load_with_info(path)                         # WAVE / FLAC, decoded by librvb on the host
        if info.sample_format != "int16" or pcm.shape[0] > 1:     # (channels, samples) -> mono mean like pyannote Audio
            mono = A.normalized(pcm, info).mean(axis=0) * np.float32(32768.0)
            pcm = np.clip(np.rint(mono), -32768, 32767).astype(np.int16)
        else:
            pcm = pcm[0]
        if info.sample_rate != 16000:                             # resampled on the device (rvd_resample_pcm)
            pcm = self.engine.resample(pcm, info.sample_rate)
        return pcm, os.path.splitext(os.path.basename(path))[0]

    def _runs(self, classes: np.ndarray) -> ClassRuns:
        """run-length view of `classes` (built once per recording: networks() hands it to finish() through `_pre`)"""
        return ClassRuns(classes, self.cfg["step_samples"] / self.cfg["sample_rate"], self.cfg["window_samples"] / self.cfg["sample_rate"])

    def networks(self, pcm: np.ndarray, prepare_finish: bool = True, resident: bool = False):
        """The GPU part on one recording (or one rank's slice of it): argmax powerset classes per window frame
        (uint8 [W, frames]) and one embedding per active (window, local speaker) pair (float32 [W, 3, dim], NaN
        where the speaker is inactive).  `prepare_finish=False` (the sharded path: finish() will see the gathered
        classes of every rank, not this slice) skips the part of finish() that is otherwise precomputed here."""
        import time
        eng = self.engine
        t0 = time.perf_counter()
        W = eng.rerun_resident() if resident else eng.upload(pcm)             # resident: `pcm` went up with an earlier call
        t1 = time.perf_counter()
        classes = eng.segment_classes()                                       # argmax on the GPU
        t2 = time.perf_counter()
        # inactive (window, speaker) pairs are never used downstream: only the active ones are embedded.
        # Round 4: the host work that needs nothing but the classes runs UNDER the embedding network (eng.embed is a foreign
        # call: the GIL is free while it waits for the GPU) -- the pooling masks of everything behind the first trunk pass, and
        # finish()'s speaker count and activity table.  Same functions on the same inputs: results are unchanged.
        excl = bool(self.params["embedding_exclude_overlap"])
        head_windows = self.HEAD_WINDOWS or (eng.emb_windows_per_pass() if hasattr(eng, "emb_windows_per_pass") else 768)
        head = min(W, head_windows)
        runs_all = None
        if head < W:
            wi, si, masks = embedding_items_from_classes(classes[:head], excl, 400, self.cfg["window_samples"])
        else:
            runs_all = self._runs(classes)
            wi, si, masks = embedding_items_from_classes(classes, excl, 400, self.cfg["window_samples"], runs_all)
        t3 = time.perf_counter()
        step = self.cfg["step_samples"] / self.cfg["sample_rate"]
        dur = self.cfg["window_samples"] / self.cfg["sample_rate"]

        def tail_items():
            return embedding_items_from_classes(classes[head:], excl, 400, self.cfg["window_samples"]) if head < W else None

        def count_active():
            # the run-length view of the whole recording: built here (helper thread), handed to finish() WITH the results --
            # finish() takes all three from the future and never builds or reads it concurrently
            runs = runs_all if runs_all is not None else self._runs(classes)
            return speaker_count_from_classes(classes, step, dur, runs), active_from_classes(classes, runs), runs

        pool = self._host_pool()
        f_tail = pool.submit(tail_items)
        f_pre = pool.submit(count_active) if prepare_finish else None
        emb = np.full((W, 3, self.cfg["emb_dim"]), np.nan, np.float32)
        n_items = int(wi.size)
        if wi.size:
            emb[wi, si] = eng.embed(wi.astype(np.int64), masks)
        tail = f_tail.result()
        if tail is not None and tail[0].size:
            wi2, si2, masks2 = tail
            emb[wi2 + head, si2] = eng.embed((wi2 + head).astype(np.int64), masks2)
            n_items += int(wi2.size)
        self._pre = (classes, f_pre) if f_pre is not None else None
        t4 = time.perf_counter()
        self.timings = dict(upload=t1 - t0, segmentation=t2 - t1, host_masks=t3 - t2, embedding=t4 - t3, windows=W, embeddings=n_items)
        return classes, emb

    HEAD_WINDOWS = None         # None = the engine's windows per trunk pass: the first pass starts on these masks alone (tests set a number)

    def _host_pool(self):
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rvd-host")
        return self._pool

    def finish(self, classes: np.ndarray, emb: np.ndarray, uri: Optional[str], num_speakers: Optional[int] = None,
               min_speakers: Optional[int] = None, max_speakers: Optional[int] = None, return_embeddings: bool = False):
        """The host part: speaker count, clustering (linkage on the GPU), reconstruction, RTTM turns."""
        import time
        t0 = time.perf_counter()
        step = self.cfg["step_samples"] / self.cfg["sample_rate"]
        dur = self.cfg["window_samples"] / self.cfg["sample_rate"]
        pre = getattr(self, "_pre", None)
        self._pre = None                                                        # one recording at a time: do not keep it alive
        if pre is not None and pre[0] is classes:                               # computed under the embedding network by networks()
            count, active, runs = pre[1].result()
        else:
            if pre is not None:
                pre[1].result()                                                 # another recording's: let the helper finish first
            runs = self._runs(classes)
            count, active = speaker_count_from_classes(classes, step, dur, runs), None
        if count.size == 0 or np.max(count) == 0:
            return (Annotation(uri), np.zeros((0, self.cfg["emb_dim"]))) if return_embeddings else Annotation(uri)
        if active is None:
            active = active_from_classes(classes, runs)
        cp = self.params["clustering"]
        ms = max_speakers if max_speakers is not None else np.inf
        hard, centroids = cluster_embeddings(emb, None, float(cp["threshold"]), int(cp["min_cluster_size"]), cp.get("method", "centroid"),
                                             num_speakers, min_speakers, max_speakers, linkage_fn=self.engine.centroid_linkage,
                                             active=active)
        t1 = time.perf_counter()
        count = np.minimum(count, ms).astype(np.int8)
        hard = hard.copy()
        hard[~active] = -2
        binary = reconstruct_from_classes(classes, hard, count, step, dur, runs)
        ann = to_annotation(binary, float(self.params["segmentation"].get("min_duration_off", 0.0)), uri)
        mapping = {label: f"SPEAKER_{i:02d}" for i, label in enumerate(ann.labels())}
        ann = ann.rename_labels(mapping)
        t2 = time.perf_counter()
        self.timings.update(clustering=t1 - t0, reconstruction=t2 - t1)
        if return_embeddings:
            return ann, centroids
        return ann

    def __call__(self, file, num_speakers: Optional[int] = None, min_speakers: Optional[int] = None,
                 max_speakers: Optional[int] = None, return_embeddings: bool = False):
        import time
        t0 = time.perf_counter()
        pcm, uri = self._load(file)
        classes, emb = self.networks(pcm)
        out = self.finish(classes, emb, uri, num_speakers, min_speakers, max_speakers, return_embeddings)
        self.timings["total"] = time.perf_counter() - t0
        return out

    apply = __call__


class Pipeline:
    """`pyannote.audio.Pipeline` entry point used by the reference."""

    @staticmethod
    def from_pretrained(checkpoint_path, use_auth_token=None, hparams_file=None, cache_dir=None, dtype: str = "bf16") -> SpeakerDiarization:
        """`checkpoint_path`: a directory holding `config.yaml` (pyannote pipeline config: params.clustering.*,
        params.segmentation.*) and the two checkpoints `segmentation.pt` / `embedding.pt` (torch state dicts, or
        pytorch-lightning checkpoints with a "state_dict" entry).  Hub names such as "Revai/reverb-diarization-v1"
        need the files downloaded first (this process does no network I/O)."""
        import yaml
        path = os.fspath(checkpoint_path)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{path!r} is not a local pipeline directory.  Download the pipeline (config.yaml, segmentation and embedding "
                "checkpoints) and pass the directory; reverb_amd does not fetch from the Hugging Face hub.")
        with open(os.path.join(path, "config.yaml")) as f:
            conf = yaml.safe_load(f) or {}
        params = conf.get("params", {})
        pp = conf.get("pipeline", {}).get("params", {})
        if "embedding_exclude_overlap" in pp:
            params["embedding_exclude_overlap"] = bool(pp["embedding_exclude_overlap"])
        from .synth_diar import DIAR_DIMS
        cfg = dict(DIAR_DIMS)
        cfg.update(conf.get("reverb_amd", {}))

        def load_sd(stem):
            import torch
            for ext in (".pt", ".bin", ".ckpt"):
                p = os.path.join(path, stem + ext)
                if os.path.exists(p):
                    sd = torch.load(p, map_location="cpu", weights_only=False)
                    sd = sd.get("state_dict", sd)
                    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
            raise FileNotFoundError(f"{path}: no {stem}.pt / .bin / .ckpt")

        seg_sd = load_sd("segmentation")
        if not any(k.startswith("sincnet.") for k in seg_sd) or not any(k.startswith("lstm.") for k in seg_sd):
            kinds = sorted({k.split(".")[0] for k in seg_sd})[:8]
            raise NotImplementedError(
                f"{path}: the segmentation checkpoint is not a PyanNet (SincNet + LSTM) model (top-level modules: {kinds}).  "
                "Only the Revai/reverb-diarization-v1 architecture (pyannote/segmentation-3.0) is built; "
                "reverb-diarization-v2 (WavLM-based segmentation) is not supported.")
        return SpeakerDiarization(cfg, seg_sd, load_sd("embedding"), params, dtype=dtype)
