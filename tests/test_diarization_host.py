"""Host logic of the diarization pipeline (reverb_amd/diarization.py) and of the word->speaker join
(reverb_amd/bin/assign_words2speakers.py; reference diarization/assign_words2speakers.py:25-89) on
hand-made inputs.  No GPU, no oracle: these are property / known-answer checks of pure numpy code."""
import io
import os
import re

import numpy as np
import pytest

from reverb_amd import diarization as D
from reverb_amd.bin import assign_words2speakers as A


def test_powerset_mapping():
    logp = np.full((1, 7, 7), -10.0, np.float32)
    for k in range(7):
        logp[0, k, k] = -0.1
    ml = D.powerset_to_multilabel(logp)[0]
    want = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1]], np.float32)
    assert np.array_equal(ml, want)


def test_frame_grid_and_aggregate_average():
    assert D.closest_frame(0.5 * D.FRAME_DURATION) == 0
    assert D.closest_frame(1.0 + 0.5 * D.FRAME_DURATION) == 59          # 1 s hop = 59.26 frames
    W, nf = 5, 589
    scores = np.ones((W, nf, 1))
    scores[2] = 3.0
    avg = D.aggregate(scores, 1.0, 10.0)
    assert avg.shape[0] == int(np.rint((10.0 + W - 1) / D.FRAME_STEP)) + 1
    assert np.isclose(avg[0, 0], 1.0)                                    # only chunk 0 covers frame 0
    s2 = D.closest_frame(2.0 + 0.5 * D.FRAME_DURATION)
    assert np.isclose(avg[s2 + 10, 0], (1 + 1 + 3) / 3.0)               # chunks 0,1,2 overlap there
    assert avg[-1, 0] == 0.0                                             # beyond the last chunk: `missing`
    tot = D.aggregate(scores, 1.0, 10.0, skip_average=True)
    assert np.isclose(tot[s2 + 10, 0], 5.0)
    nan = scores.copy(); nan[2] = np.nan
    avg2 = D.aggregate(nan, 1.0, 10.0)
    assert np.isclose(avg2[s2 + 10, 0], 1.0)                             # NaN chunk ignored


def test_speaker_count_rounds_mean_activity():
    W, nf = 3, 589
    b = np.zeros((W, nf, 3), np.float32)
    b[:, :, 0] = 1.0
    b[1, :, 1] = 1.0                                                     # one of the three chunks sees a 2nd speaker
    c = D.speaker_count(b, 1.0, 10.0)
    s1, s2 = D.closest_frame(1.0 + 0.5 * D.FRAME_DURATION), D.closest_frame(2.0 + 0.5 * D.FRAME_DURATION)
    assert c[0, 0] == 1 and c[s1 + 5, 0] == 2 and c[s2 + 5, 0] == 1      # mean 1, 1.5 -> 2 (rint half-even), 1.33 -> 1


def test_embedding_masks_exclude_overlap_with_fallback():
    b = np.zeros((1, 589, 3), np.float32)
    b[0, 100:200, 0] = 1; b[0, 150:300, 1] = 1                           # overlap in 150..200
    b[0, 400:402, 2] = 1; b[0, 400:402, 0] = 1                           # speaker 2 only ever overlapped
    m = D.embedding_masks(b, True)
    assert m.shape == (1, 3, 589)
    assert m[0, 0].sum() == 50 and m[0, 0, 150:200].sum() == 0
    assert m[0, 1].sum() == 100
    assert m[0, 2].sum() == 2                                            # clean mask empty -> falls back to the full mask
    assert np.array_equal(D.embedding_masks(b, False)[0, 0], b[0, :, 0])


def _toy_embeddings(seed=0):
    rng = np.random.default_rng(seed)
    W, dim = 40, 16
    a, b_ = np.zeros(dim), np.zeros(dim)
    a[0], b_[1] = 1.0, 1.0
    emb = np.zeros((W, 3, dim), np.float32)
    seg = np.zeros((W, 20, 3), np.float32)
    truth = np.full((W, 3), -1)
    for w in range(W):
        emb[w, 0] = 5.0 * (a + 0.05 * rng.standard_normal(dim)); seg[w, :10, 0] = 1; truth[w, 0] = 0
        if w % 2 == 0:
            emb[w, 1] = 0.3 * (b_ + 0.05 * rng.standard_normal(dim)); seg[w, 10:, 1] = 1; truth[w, 1] = 1
        emb[w, 2] = np.nan                                               # never active
    c = (a + b_) / np.sqrt(2) + 0.05 * rng.standard_normal(dim)          # 2 stray embeddings between the groups
    emb[1, 1] = c; seg[1, 10:, 1] = 1
    emb[3, 1] = c + 0.01; seg[3, 10:, 1] = 1
    return emb, seg, truth


def test_clustering_two_speakers_small_cluster_reassigned():
    emb, seg, truth = _toy_embeddings()
    hard, cent = D.cluster_embeddings(emb, seg, threshold=0.7045654963945799, min_cluster_size=12)
    assert cent.shape[0] == 2
    k0 = hard[0, 0]
    assert np.all(hard[:, 0] == k0)
    assert np.all(hard[::2, 1] == 1 - k0)
    forced = D.cluster_embeddings(emb, seg, 0.7045654963945799, 12, num_clusters=1)[0]
    assert np.all(forced == 0)


def test_reconstruct_and_rttm():
    W, nf = 4, 589
    seg = np.zeros((W, nf, 3), np.float32)
    hard = np.full((W, 3), -2)
    for w in range(W):
        # global speaker 0 talks in file seconds [2, 6), speaker 1 in [5, 9): place them per chunk
        t = (np.arange(nf) * D.FRAME_STEP + 0.5 * D.FRAME_DURATION) + w * 1.0
        seg[w, :, w % 3] = ((t >= 2) & (t < 6))                          # local slot differs per chunk
        seg[w, :, (w + 1) % 3] = ((t >= 5) & (t < 9))
        hard[w, w % 3] = 0; hard[w, (w + 1) % 3] = 1
    count = D.speaker_count(seg, 1.0, 10.0)
    binary = D.reconstruct(seg, hard, count, 1.0, 10.0)
    ann = D.to_annotation(binary, 0.0, uri="toy")
    ann = ann.rename_labels({k: f"SPEAKER_{i:02d}" for i, k in enumerate(ann.labels())})
    buf = io.StringIO(); ann.write_rttm(buf)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2
    pat = re.compile(r"^SPEAKER toy 1 (\d+\.\d{3}) (\d+\.\d{3}) <NA> <NA> (SPEAKER_\d\d) <NA> <NA>$")
    got = {m.group(3): (float(m.group(1)), float(m.group(2))) for m in map(pat.match, lines)}
    assert abs(got["SPEAKER_00"][0] - 2.0) < 0.05 and abs(got["SPEAKER_00"][1] - 4.0) < 0.08
    assert abs(got["SPEAKER_01"][0] - 5.0) < 0.05 and abs(got["SPEAKER_01"][1] - 4.0) < 0.08


def _random_classes(rng, W, nf, min_run, max_run):
    """Powerset classes cut from one file-level timeline (overlapping chunks agree where they overlap) plus a
    local-speaker -> cluster map in which two local speakers of a chunk sometimes share a cluster."""
    n_file = int((W - 1) / D.FRAME_STEP) + nf + 10
    line = np.zeros(n_file, np.int64)
    t = 0
    while t < n_file:
        run = int(rng.integers(min_run, max_run))
        line[t:t + run] = rng.integers(0, 8)                              # 0 silence, 1..5 one speaker, 6..7 two at once
        t += run
    classes = np.zeros((W, nf), np.uint8)
    hard = np.full((W, 3), -2, np.int64)
    pair = {(0, 1): 4, (0, 2): 5, (1, 2): 6}
    for w in range(W):
        seg = line[int(np.rint(w / D.FRAME_STEP)):][:nf]
        ids = [g for g in np.unique(seg) if 0 < g <= 5][:3]
        for i, g in enumerate(ids):
            classes[w][seg == g] = i + 1
            hard[w, i] = rng.integers(0, 4) if rng.random() < 0.2 else g - 1
        if len(ids) >= 2:
            classes[w][seg == 6] = pair[(0, 1)]
            classes[w][seg == 7] = pair[(0, 2)] if len(ids) == 3 else pair[(0, 1)]
    return classes, hard


@pytest.mark.parametrize("W,nf,min_run,max_run", [(1, 589, 30, 400), (9, 589, 1, 8), (60, 589, 5, 90), (3, 50, 1, 5), (2, 1, 1, 2)])
def test_run_length_host_path_equals_per_frame_path(W, nf, min_run, max_run):
    """The pipeline works on run-length views of the argmax classes (ClassRuns); the per-frame functions that follow
    pyannote's formulation (speaker_count / embedding_masks / reconstruct on the multilabel tensor) must agree exactly."""
    rng = np.random.default_rng(W * 1000 + nf)
    classes, hard = _random_classes(rng, W, nf, min_run, max_run)
    ml = D.classes_to_multilabel(classes)
    runs = D.ClassRuns(classes, 1.0, 10.0)
    assert np.array_equal(runs.histogram(), np.stack([np.bincount(c, minlength=7) for c in classes]))
    count = D.speaker_count_from_classes(classes, 1.0, 10.0, runs)
    want_count = D.speaker_count(ml, 1.0, 10.0)
    assert count.dtype == want_count.dtype and np.array_equal(count, want_count)
    active = D.active_from_classes(classes, runs)
    assert np.array_equal(active, ml.sum(axis=1) > 0)
    for exclude in (True, False):
        wi, si, masks = D.embedding_items_from_classes(classes, exclude, runs=runs)
        full = D.embedding_masks(ml, exclude)
        assert np.array_equal(np.stack([wi, si]), np.stack(np.nonzero(active)))
        assert masks.dtype == np.float32 and np.array_equal(masks, full[wi, si])
    hard[~active] = -2
    if hard.max() < 0:
        hard[0, 0] = 0
    for cap in (np.inf, 1):
        cnt = np.minimum(count, cap).astype(np.int8)
        got = D.reconstruct_from_classes(classes, hard, cnt, 1.0, 10.0, runs)
        want = D.reconstruct(ml, hard, cnt, 1.0, 10.0)
        assert got.shape == want.shape and np.array_equal(got, want)


def test_top_count_ties_and_many_clusters():
    rng = np.random.default_rng(3)
    for ncl in (1, 3, 8, 11):                                             # 11 takes the argsort branch
        act = rng.integers(0, 3, (200, ncl)).astype(np.float64)            # many ties
        count = rng.integers(0, 4, (200, 1)).astype(np.int8)
        got = D._top_count(act, count)
        width = max(ncl, int(count.max()))
        assert got.shape == (200, width)
        padded = np.pad(act, ((0, 0), (0, width - ncl)))
        for t in range(200):
            order = sorted(range(width), key=lambda k: (-padded[t, k], k))[:count[t, 0]]
            assert sorted(np.nonzero(got[t])[0]) == sorted(order)


def test_rttm_roundtrip_and_label_with_space(tmp_path):
    ann = D.Annotation("u1")
    ann.add(D.Segment(1.0, 2.5), "a", "SPEAKER_01"); ann.add(D.Segment(0.25, 0.75), "b", "SPEAKER_00")
    p = tmp_path / "x.rttm"
    with open(p, "w") as f:
        ann.write_rttm(f)
    assert open(p).read().splitlines()[0] == "SPEAKER u1 1 0.250 0.500 <NA> <NA> SPEAKER_00 <NA> <NA>"
    back = D.load_rttm(str(p))
    assert list(back) == ["u1"]
    assert [(round(s.start, 3), round(s.end, 3), l) for s, _, l in back["u1"].itertracks(yield_label=True)] == \
        [(0.25, 0.75, "SPEAKER_00"), (1.0, 2.5, "SPEAKER_01")]
    bad = D.Annotation("u 1"); bad.add(D.Segment(0, 1), 0, "x")
    with pytest.raises(ValueError):
        bad.write_rttm(io.StringIO())


def test_speaker_for_segment_cases():
    turns = [(0.0, 2.0, "A"), (1.5, 4.0, "B"), (6.0, 7.0, "A")]
    assert A.speaker_for_segment(0.2, 0.5, turns) == "A"                 # exactly one turn
    assert A.speaker_for_segment(1.6, 1.0, turns) == "B"                 # both overlap: B 1.0 s vs A 0.4 s
    assert A.speaker_for_segment(1.4, 0.3, turns) == "A"                 # A 0.3 vs B 0.2
    assert A.speaker_for_segment(4.5, 0.2, turns) == "B"                 # gap: nearest is B (0.5 s) not A (1.3 s)
    assert A.speaker_for_segment(5.5, 0.2, turns) == "A"
    assert A.speaker_for_segment(2.0, 0.0, turns) == "A"                 # empty query -> nearest; both at distance 0 -> earliest
    assert A.speaker_for_segment(1.0, 1.0, []) == ""
    assert A.speaker_for_segment(2.0, 0.5, turns) == "B"                 # half-open: A ends at 2.0


def test_assign_words_cli(tmp_path):
    rttm, ctm, stm = tmp_path / "d.rttm", tmp_path / "w.ctm", tmp_path / "o.stm"
    rttm.write_text("SPEAKER rec 1 0.000 2.000 <NA> <NA> SPEAKER_00 <NA> <NA>\nSPEAKER rec 1 2.000 3.000 <NA> <NA> SPEAKER_01 <NA> <NA>\n")
    ctm.write_text("rec 0 0.10 0.30 hello 0.99\nrec 0 1.90 0.40 there 0.50\nrec 0 9.00 0.20 bye 1.00\n")
    A.main([str(rttm), str(ctm), str(stm)])
    assert stm.read_text().splitlines() == ["rec 1 SPEAKER_00 0.100 0.400 hello", "rec 1 SPEAKER_01 1.900 2.300 there",
                                            "rec 1 SPEAKER_01 9.000 9.200 bye"]


def test_pipeline_loading_errors(tmp_path):
    with pytest.raises(FileNotFoundError, match="not a local pipeline directory"):
        D.Pipeline.from_pretrained("Revai/reverb-diarization-v1", use_auth_token="x")
    from reverb_amd import synth_diar
    d = synth_diar.write_pipeline_dir(str(tmp_path / "pipe"))
    pipe = D.Pipeline.from_pretrained(d)
    assert pipe.params["clustering"]["min_cluster_size"] == 12
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pipe.to("cpu")
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pipe.to(torch.device("cpu"))


def test_v2_segmentation_checkpoint_is_refused_with_a_clear_message(tmp_path):
    """a non-PyanNet (e.g. WavLM-based reverb-diarization-v2) checkpoint must fail at load time, by name."""
    import torch
    from reverb_amd import synth_diar
    d = synth_diar.write_pipeline_dir(str(tmp_path / "pipe"))
    torch.save({"wavlm.encoder.layers.0.attention.k_proj.weight": torch.zeros(4, 4), "classifier.weight": torch.zeros(7, 4)},
               os.path.join(d, "segmentation.pt"))
    with pytest.raises(NotImplementedError, match="reverb-diarization-v2"):
        D.Pipeline.from_pretrained(d)


def test_speaker_active_on_the_last_frame_only_gives_no_empty_turn():
    """start == end == ts[n-1] must not become a 0-duration RTTM line (pyannote drops empty segments;
    make_turns / intervaltree would raise 'Null Interval' on it)."""
    b = np.zeros((50, 2), np.float32)
    b[10:20, 0] = 1
    b[-1, 1] = 1                      # speaker 1: active on the final frame only
    ann = D.to_annotation(b, uri="x")
    segs = [(seg.start, seg.end, lab) for seg, _, lab in ann.itertracks(yield_label=True)]
    assert len(segs) == 1 and segs[0][2] == 0 and segs[0][1] > segs[0][0]
    b[-2:, 1] = 1                     # two frames: a real (one frame-step long) turn
    assert len(list(D.to_annotation(b, uri="x").itertracks(yield_label=True))) == 2


def test_vectorised_word_speaker_join_equals_the_per_word_function():
    rng = np.random.default_rng(12)
    for trial in range(6):
        n_turns = int(rng.integers(0, 40))
        t0 = np.sort(rng.random(n_turns) * 100)
        turns = sorted({(float(s), float(s + 0.2 + rng.random() * 6), f"SPEAKER_{int(rng.integers(0, 4)):02d}") for s in t0})
        starts = rng.random(300) * 110 - 5
        durs = np.where(rng.random(300) < 0.05, 0.0, rng.random(300) * 1.5)
        want = [A.speaker_for_segment(float(s), float(d), turns) for s, d in zip(starts, durs)]
        assert A.speakers_for_words(starts, durs, turns, block=64) == want
    assert A.speakers_for_words([1.0], [0.5], []) == [""]


def test_fcluster_distance_shortcut_equals_scipy():
    """cluster_embeddings cuts the dendrogram with scipy's compiled routine directly (no Python-side validation of Z): same labels
    as scipy.cluster.hierarchy.fcluster, also on centroid dendrograms with inversions and at thresholds on merge heights."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from reverb_amd.diarization import _fcluster_distance
    rng = np.random.default_rng(3)
    for n, d in ((2, 4), (7, 3), (200, 16), (1500, 8)):
        x = rng.standard_normal((n, d))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        z = linkage(x, method="centroid", metric="euclidean")
        for t in (0.0, 0.3, 0.7045654963945799, float(z[len(z) // 2, 2]), 10.0):
            np.testing.assert_array_equal(_fcluster_distance(z, t), fcluster(z, t, criterion="distance"))

