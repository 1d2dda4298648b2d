"""Edge cases of the front door (ReverbASR.transcribe / Engine.decode_resident / diarization pipeline): empty and
sub-frame audio, inputs that subsample to zero or one encoder frame, a chunk boundary straddled by one frame."""
import numpy as np
import pytest

from reverb_amd import synth

pytestmark = pytest.mark.gpu

CHUNK = 400


@pytest.fixture(scope="module")
def eng():
    from reverb_amd.engine import Engine
    cfg = synth.make_config("tiny")
    e = Engine(cfg, synth.make_state_dict(cfg, 0, synth.CTC_GAMMA, 12.33), dtype="f32", device=0, max_chunks=3, chunk_frames=CHUNK)
    yield e
    e.close()


def _decode(eng, pcm, modes=("ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring", "attention")):
    eng.upload_pcm(pcm)
    nf = eng.fbank()
    return nf, eng.decode_resident(nf, list(modes), CHUNK, 4, 0.1, 0.0)


@pytest.mark.parametrize("n_samples", [0, 1, 399])
def test_audio_shorter_than_one_frame_gives_no_chunks(eng, n_samples):
    nf, res = _decode(eng, np.zeros(n_samples, np.int16))
    assert nf == 0
    assert all(res[m] == [] for m in res)


@pytest.mark.parametrize("frames,enc", [(1, 0), (6, 0), (7, 1), (10, 1), (11, 2)])
def test_inputs_that_subsample_to_zero_or_one_encoder_frame(eng, frames, enc):
    """(n - 7)//4 + 1 valid encoder frames (subsampling.py:226): 0 for fewer than 7 input frames.  The CTC-based modes
    then return empty hypotheses; `attention` mode decodes against a fully masked memory exactly as the reference does
    (golden case tiny_bn, 5-frame tail chunk), i.e. it may emit a few tokens of the decoder's prior."""
    pcm = synth.synth_audio(1.0, seed=2)[: 400 + 160 * (frames - 1)]
    nf, res = _decode(eng, pcm)
    assert nf == frames
    assert eng.encoder_lens().tolist() == [enc]
    for m in res:
        assert len(res[m]) == 1
        if enc == 0 and m != "attention":
            assert list(res[m][0].tokens) == []


def test_one_frame_past_a_chunk_boundary(eng):
    pcm = synth.synth_audio(5.0, seed=4)[: 400 + 160 * CHUNK]          # CHUNK + 1 frames -> chunks of CHUNK and 1 frames
    nf, res = _decode(eng, pcm, ("ctc_greedy_search", "attention_rescoring"))
    assert nf == CHUNK + 1
    assert eng.encoder_lens().tolist() == [(CHUNK - 7) // 4 + 1, 0]
    assert len(res["attention_rescoring"]) == 2 and list(res["attention_rescoring"][1].tokens) == []
    # the first chunk is unaffected by the presence of the tail chunk
    nf1, res1 = _decode(eng, pcm[: 400 + 160 * (CHUNK - 1)], ("ctc_greedy_search", "attention_rescoring"))
    assert nf1 == CHUNK
    assert list(res1["attention_rescoring"][0].tokens) == list(res["attention_rescoring"][0].tokens)


def test_more_chunks_than_one_launch_holds(eng):
    pcm = synth.synth_audio(4.0 * 7 + 1.3, seed=6)                      # 8 chunks of 4 s with max_chunks = 3
    nf, res = _decode(eng, pcm, ("attention_rescoring",))
    assert len(res["attention_rescoring"]) == -(-nf // CHUNK) == 8
    from reverb_amd.engine import Engine
    big = Engine(eng.configs, synth.make_state_dict(eng.configs, 0, synth.CTC_GAMMA, 12.33), dtype="f32", device=0, max_chunks=16,
                 chunk_frames=CHUNK)
    big.upload_pcm(pcm)
    want = big.decode_resident(big.fbank(), ["attention_rescoring"], CHUNK, 4, 0.1, 0.0)["attention_rescoring"]
    big.close()
    assert [list(r.tokens) for r in res["attention_rescoring"]] == [list(r.tokens) for r in want]


def test_diarization_rejects_empty_audio_and_handles_silence(tmp_path):
    from reverb_amd import diarization as D, synth_diar
    from reverb_amd._lib import RvbError
    pipe = D.Pipeline.from_pretrained(synth_diar.write_pipeline_dir(str(tmp_path / "p")), dtype="f32").to("cuda")
    with pytest.raises(RvbError):
        pipe({"waveform": np.zeros(0, np.int16), "sample_rate": 16000})
    ann = pipe({"waveform": np.zeros(16000 * 3, np.int16), "sample_rate": 16000, "uri": "silence"})   # 3 s of digital silence
    assert ann.uri == "silence"
    for seg, _, label in ann.itertracks(yield_label=True):
        assert 0.0 <= seg.start < seg.end <= 10.1 and label.startswith("SPEAKER_")


def test_api_is_callable_from_a_worker_thread(tmp_path):
    """The reference's examples/stream.py:45-53 calls `transcribe` from a worker thread, one call at a time."""
    import threading
    from reverb_amd.reverb import load_model
    mdir = synth.write_model_dir(str(tmp_path / "m"), "tiny")
    wav = str(tmp_path / "a.wav")
    synth.write_wav(wav, synth.synth_audio(8.0, seed=8))
    asr = load_model(mdir, gpu=0, dtype="f32", max_chunks=4)
    want = asr.transcribe(wav, mode="attention_rescoring", format="ctm")
    got, err = [], []

    def work():
        try:
            got.append(asr.transcribe(wav, mode="attention_rescoring", format="ctm"))
        except Exception as ex:          # pragma: no cover
            err.append(ex)
    for _ in range(2):
        t = threading.Thread(target=work)
        t.start(); t.join()
    assert not err, err
    assert got == [want, want]


def test_cabi_collective_on_one_rank():
    """rvb_comm_unique_id / rvb_comm_create / rvb_comm_allgather (RCCL bound directly by librvb; the stand-alone communicator
    reverb_amd.dist uses by default on GPUs) and the engine-bound rvb_comm_init / rvb_allgather_results, with a world of one:
    the only size a 1-GPU box can run; the packed-result gather and the packed diarization shard return what went in."""
    from golden_util import Case
    from reverb_amd import dist as rdist
    from reverb_amd.engine import Engine
    from reverb_amd.search import DecodeResult
    case = Case("tiny_ln")
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    uid = rdist.RvbComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = rdist.RvbComm(eng, 1, 0, uid)
    x = np.arange(1000, dtype=np.int32)
    np.testing.assert_array_equal(comm.all_gather(x), x[None])
    hyps = [DecodeResult((1, 2, 3), -1.5, confidence=0.5, times=[4, 5, 6], tokens_confidence=[0.1, 0.2, 0.3]), DecodeResult([7])]
    hyps[1].ctc_frames = [9]
    got = rdist.all_gather_results(hyps, None, comm=comm)
    assert [(list(h.tokens), h.times, h.ctc_frames, h.score) for h in got] == [([1, 2, 3], [4, 5, 6], None, -1.5), ([7], None, [9], 0.0)]
    # the diarization shard's one packed buffer through the same communicator
    rng = np.random.default_rng(0)
    classes = rng.integers(0, 7, size=(3, 589)).astype(np.uint8)
    emb = rng.standard_normal((3, 3, 256)).astype(np.float32)
    emb[1, 2] = np.nan
    host = rdist.gather_words(rdist.pack_diar_shard(classes, emb, 5, 589, 256), None, comm)
    c2, e2 = rdist.unpack_diar_shard(host[0], 5)
    np.testing.assert_array_equal(c2, classes)
    np.testing.assert_array_equal(np.isnan(e2), np.isnan(emb))
    np.testing.assert_array_equal(np.nan_to_num(e2), np.nan_to_num(emb))
    # the collective alone, device to device (the xGMI datapoint of bench.py): runs, is checked, takes a sane time
    ms = comm.time_all_gather(7_200_000, iters=3)
    assert 0.0 < ms < 50.0
    # round 4: what a launcher needs besides the gather (one RCCL communicator per rank: no torch.distributed "nccl" group)
    comm.barrier()
    assert comm.max(3.25) == 3.25
    # ... and the posterior exchange device to device: the engine's top-k buffers, straight from HBM
    x2, lens2 = case.chunked_feats()
    eng.encode(x2, lens2, case.beam)
    v, i = eng.ctc_topk()
    nbytes, (gv, gi) = comm.all_gather_topk(eng, to_host=True)
    assert nbytes == v.nbytes + i.nbytes
    np.testing.assert_array_equal(gv[0], v)
    np.testing.assert_array_equal(gi[0], i)
    assert comm.all_gather_topk(eng)[0] == nbytes            # without the host copy
    # a generous timeout changes nothing
    comm.set_timeout(30.0)
    np.testing.assert_array_equal(comm.all_gather(x), x[None])
    # ... an impossible one (1 ns for a 64 MB gather + two PCIe copies) ends the collective with RVB_E_TIMEOUT instead of
    # blocking, and the communicator refuses further use: the path a dead peer takes on a real world
    from reverb_amd._lib import RvbError
    comm.set_timeout(1e-9)
    big = np.zeros(16_000_000, np.int32)
    with pytest.raises(RvbError, match=r"\(-7\)"):
        comm.all_gather(big)
    with pytest.raises(RvbError, match="aborted after a timeout"):
        comm.all_gather(x)
    comm.close()
    # the engine-bound form of the same collective
    import ctypes as C
    from reverb_amd._lib import check
    idbuf = C.create_string_buffer(rdist.RvbComm.unique_id(), 128)
    check(eng.lib.rvb_comm_init(eng.handle, 1, 0, C.cast(idbuf, C.c_void_p)), "rvb_comm_init")
    y = np.empty_like(x)
    check(eng.lib.rvb_allgather_results(eng.handle, x.ctypes.data, x.nbytes, y.ctypes.data), "rvb_allgather_results")
    np.testing.assert_array_equal(y, x)
    check(eng.lib.rvb_comm_destroy(eng.handle), "rvb_comm_destroy")
    eng.close()
