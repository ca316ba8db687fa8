"""Host-side mirror of the reference interface (onnxruntime.InferenceSession as Mimic3Voice uses it) and the
batch sharding across ranks.  On CPU these run on the test model of the kernels (tests/emu); the same checks
run against the real library in test_gpu_parity.py / test_gpu_session.py."""
import os
import sys

import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from mimic3_amd.session import GraphOptimizationLevel, InferenceSession, InvalidArgument, SessionOptions
from mimic3_amd import sharding
from oracle.vits_oracle import VitsOracle, audio_float_to_int16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _voice(tmp_path, cfg, seed=5):
    w = W.synthetic_weights(cfg, seed=seed)
    d = tmp_path / "voice"
    d.mkdir()
    W.save(d / "generator.m355", cfg, w)
    (d / "generator.onnx").write_bytes(b"")  # what Mimic 3 passes; the engine opens the .m355 beside it
    return d, w


def test_session_mirrors_the_reference_call_sequence(emu_lib, tmp_path):
    """voice.py:392-405 (construct) and voice.py:180-232 (feed, run, squeeze, int16)."""
    cfg = VitsConfig.tiny()
    d, w = _voice(tmp_path, cfg)
    so = SessionOptions()
    so.graph_optimization_level = GraphOptimizationLevel.ORT_DISABLE_ALL  # armv7l branch of the reference
    so.use_deterministic_compute = True
    sess = InferenceSession(str(d / "generator.onnx"), sess_options=so, providers=["CUDAExecutionProvider"], _library=emu_lib)
    assert [i.name for i in sess.get_inputs()] == ["input", "input_lengths", "scales"]
    phoneme_ids = [3, 7, 1, 9, 4, 2]
    text_array = np.expand_dims(np.array(phoneme_ids, dtype=np.int64), 0)
    inputs = {
        "input": text_array,
        "input_lengths": np.array([text_array.shape[1]], dtype=np.int64),
        "scales": np.array([0.0, 1.0, 0.0], dtype=np.float32),
    }
    audio = sess.run(None, inputs)[0].squeeze()
    ref = VitsOracle(cfg, w).infer(text_array, inputs["input_lengths"], inputs["scales"])
    assert audio.shape == (int(ref["audio_lengths"][0]),)
    assert np.abs(audio - ref["audio"][0, 0]).max() < 1e-4
    pcm, lengths = sess.run_pcm16(inputs)
    assert np.array_equal(pcm[0], audio_float_to_int16(audio))
    assert int(lengths[0]) == audio.shape[0]


def test_session_errors_like_onnxruntime(emu_lib, tmp_path):
    cfg = VitsConfig.tiny(n_speakers=3)
    d, _ = _voice(tmp_path, cfg)
    sess = InferenceSession(str(d / "generator.m355"), _library=emu_lib)
    assert [i.name for i in sess.get_inputs()][-1] == "sid"
    feed = {"input": np.ones((1, 4), np.int64), "input_lengths": np.array([4]), "scales": np.array([0, 1, 0], np.float32)}
    with pytest.raises(ValueError, match="sid"):
        sess.run(None, feed)
    feed["sid"] = np.array([1])
    assert sess.run(None, feed)[0].ndim == 3
    with pytest.raises(ValueError, match="Invalid input name"):
        sess.run(None, {**feed, "bogus": np.zeros(1)})
    with pytest.raises(ValueError):
        sess.run(None, {**feed, "input": np.ones((4,), np.int64)})
    with pytest.raises(ValueError, match="speaker id"):
        sess.run(None, {**feed, "sid": np.array([9])})
    with pytest.raises(FileNotFoundError):
        InferenceSession(str(tmp_path / "nowhere" / "generator.onnx"), _library=emu_lib)


def test_micro_batching_coalesces_concurrent_calls(emu_lib):
    """N2: concurrent B = 1 run() calls from worker threads become batched engine calls with identical results."""
    import threading

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=9))
    plain = InferenceSession(blob, _library=emu_lib)
    so = SessionOptions()
    so.micro_batch_window_ms = 200.0
    so.micro_batch_max = 8
    mb = InferenceSession(blob, sess_options=so, _library=emu_lib)
    rng = np.random.default_rng(1)
    feeds = []
    for i in range(8):
        n = int(rng.integers(3, 12))
        feeds.append({"input": rng.integers(1, 20, (1, n)).astype(np.int64), "input_lengths": np.array([n], np.int64),
                      "scales": np.array([0.0, 1.0 if i % 2 else 1.5, 0.0], np.float32)})
    expect = [plain.run(None, f)[0] for f in feeds]
    got = [None] * 8
    errs = []

    def work(i):
        try:
            got[i] = mb.run(None, feeds[i])[0]
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for e, g in zip(expect, got):
        assert e.shape == g.shape and np.array_equal(e, g)
    assert mb._batcher.requests == 8 and mb._batcher.batches < 8  # two scale groups -> 2 (or a few) engine calls
    with pytest.raises(ValueError):  # errors propagate to the waiting caller
        mb.run(None, {"input": np.full((1, 3), 999, np.int64), "input_lengths": np.array([3]), "scales": np.array([0, 1, 0], np.float32)})


def test_micro_batching_with_lanes_runs_batches_concurrently(emu_lib):
    """Micro-batcher + lanes: batches are handed to lane workers; every caller still gets its own row, bit-exact."""
    import threading

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=9))
    plain = InferenceSession(blob, _library=emu_lib)
    so = SessionOptions()
    so.micro_batch_window_ms = 30.0
    so.micro_batch_max = 3
    so.lanes = 2
    mb = InferenceSession(blob, sess_options=so, _library=emu_lib)
    assert mb._batcher._pool is not None
    rng = np.random.default_rng(3)
    feeds = [{"input": rng.integers(1, 20, (1, int(rng.integers(3, 12)))).astype(np.int64), "scales": np.array([0.0, 1.0, 0.0], np.float32)}
             for _ in range(12)]
    for f in feeds:
        f["input_lengths"] = np.array([f["input"].shape[1]], np.int64)
    expect = [plain.run(None, f)[0] for f in feeds]
    got = [None] * len(feeds)

    def work(i):
        got[i] = mb.run(None, feeds[i])[0]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(feeds))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e, g in zip(expect, got):
        assert e.shape == g.shape and np.array_equal(e, g)
    assert mb._batcher.requests == 12 and 4 <= mb._batcher.batches <= 12


def test_lanes_run_concurrent_calls_on_separate_engine_handles(emu_lib):
    """Several engine handles behind one session: concurrent run() calls land on different lanes, results do not
    depend on the lane (the noise is keyed by the session seed and the utterance counter, not by the handle)."""
    import threading

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=4))
    so1, so3 = SessionOptions(), SessionOptions()
    so1.seed = so3.seed = 77
    so3.lanes = 3
    one = InferenceSession(blob, sess_options=so1, _library=emu_lib)
    three = InferenceSession(blob, sess_options=so3, _library=emu_lib)
    assert len(three._engines) == 3 and len({id(e) for e in three._engines}) == 3
    rng = np.random.default_rng(2)
    feeds = [{"input": rng.integers(1, 20, (2, 9)).astype(np.int64), "input_lengths": np.array([9, 5], np.int64),
              "scales": np.array([0.0, 1.0, 0.0], np.float32)} for _ in range(6)]
    expect = [one.run(None, f)[0] for f in feeds]
    got = [None] * 6
    seen = set()
    orig = three._free_lanes.acquire

    def spy_get(*a, **k):
        e = orig(*a, **k)
        seen.add(id(e))
        return e

    three._free_lanes.acquire = spy_get
    barrier = threading.Barrier(6)

    def work(i):
        barrier.wait()
        got[i] = three.run(None, feeds[i])[0]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e, g in zip(expect, got):
        assert np.array_equal(e, g)
    assert len(seen) >= 2  # more than one lane was used
    assert three._free_lanes.idle() == 3  # every lane came back


def test_onnxruntime_shim_module_surface():
    import mimic3_amd.onnxruntime_shim as shim

    for name in ("InferenceSession", "SessionOptions", "GraphOptimizationLevel", "get_available_providers"):
        assert hasattr(shim, name)
    assert shim.GraphOptimizationLevel.ORT_DISABLE_ALL == 0
    saved = sys.modules.pop("onnxruntime", None)
    try:
        assert shim.install() is sys.modules["onnxruntime"]
        import onnxruntime  # noqa: F401  (resolves to the shim)

        assert onnxruntime.InferenceSession is InferenceSession
    finally:
        sys.modules.pop("onnxruntime", None)
        if saved is not None:
            sys.modules["onnxruntime"] = saved


def test_shard_bounds_and_balance():
    for n in (0, 1, 7, 8, 256):
        for ws in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    parts = sharding.balanced_order([10, 1, 1, 1, 9, 2, 2, 8], 2)
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum([10, 1, 1, 1, 9, 2, 2, 8][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 2


def _rank_main(rank, world, port, blob, tmpdir):
    """One process per (emulated) GPU: shard the feed, synthesise locally, gather on rank 0 with gloo."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from mimic3_amd import build
    from mimic3_amd._native import NativeLibrary
    from mimic3_amd.session import InferenceSession as IS

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = NativeLibrary(build.EMU_LIB)
    so = SessionOptions()
    so.seed = 1234
    sess = IS(blob, sess_options=so, _library=lib)
    rng = np.random.default_rng(0)
    B, Tx = 5, 9
    ids = rng.integers(1, 20, (B, Tx)).astype(np.int64)
    lens = np.array([9, 4, 7, 9, 5], np.int64)
    feed = {"input": ids, "input_lengths": lens, "scales": np.array([0.0, 1.0, 0.0], np.float32)}
    local, rows = sharding.shard_feed(feed, world, rank)
    out = sess.run(None, local)[0] if len(rows) else np.zeros((0, 1, 0), np.float32)
    auds = [out[i, 0, : int(sess.last_lengths[i])] for i in range(len(rows))]
    full = sharding.gather_results(auds, rows, B)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "gathered.npz"), **{f"a{i}": a for i, a in enumerate(full)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process(emu_lib, tmp_path):
    """world_size = 2 over gloo on CPU: the N > 1 path (shard -> per-rank engine -> optional gather)."""
    import socket

    import torch.multiprocessing as mp

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=5))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, blob, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    sess = InferenceSession(blob, _library=emu_lib)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, 20, (5, 9)).astype(np.int64)
    lens = np.array([9, 4, 7, 9, 5], np.int64)
    ref = sess.run(None, {"input": ids, "input_lengths": lens, "scales": np.array([0.0, 1.0, 0.0], np.float32)})[0]
    for b in range(5):
        L = int(sess.last_lengths[b])
        assert np.array_equal(got[f"a{b}"], ref[b, 0, :L])


def test_fused_volume_equals_audioop_mul_and_wav_helpers(emu_lib):
    """N4: settings.volume applied inside the int16 kernel gives the bytes audioop.mul gives on the host
    (tts.py:542-543); silence / WAV framing helpers match add_break and the wave module."""
    import audioop
    import io
    import wave

    from mimic3_amd import postprocess as PP

    cfg = VitsConfig.tiny()
    sess = InferenceSession(W.pack(cfg, W.synthetic_weights(cfg, seed=3, frames_per_id=3.0)), _library=emu_lib)
    feed = {"input": np.array([[3, 7, 1, 9, 4, 2, 8], [5, 5, 2, 0, 0, 0, 0]], np.int64),
            "input_lengths": np.array([7, 3], np.int64), "scales": np.array([0, 1, 0], np.float32)}
    base, lengths = sess.run_pcm16(feed)
    assert max(int(np.abs(r).max()) for r in base) >= 32766
    for vol in (50.0, 33.0, 150.0, 7.5, 99.9, 100.0):
        rows, _ = sess.run_pcm16(feed, volume=vol)
        for r, b0 in zip(rows, base):
            want = np.frombuffer(audioop.mul(b0.tobytes(), 2, vol / 100.0), dtype=np.int16)
            assert np.array_equal(r, want), vol
            assert np.array_equal(PP.apply_volume(b0, vol), want), vol
    with pytest.raises(ValueError):
        sess.run_pcm16(feed, volume=0.0)
    edge = np.array([-32768, -32767, -1, 0, 1, 32767], np.int16)
    for vol in (100.0, 250.0, 1.0, 50.0):
        assert np.array_equal(PP.apply_volume(edge, vol), np.frombuffer(audioop.mul(edge.tobytes(), 2, vol / 100.0), dtype=np.int16))
    assert PP.silence(250).shape == (int(0.25 * 22050),) and not PP.silence(250).any()
    blob = PP.utterances_to_wav(base, 22050, break_ms=100)
    with wave.open(io.BytesIO(blob), "rb") as wf:
        assert (wf.getframerate(), wf.getsampwidth(), wf.getnchannels()) == (22050, 2, 1)
        frames = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    assert np.array_equal(frames, np.concatenate([base[0], PP.silence(100), base[1]]))
    ref = io.BytesIO()
    with wave.open(ref, "wb") as wf:
        wf.setframerate(22050); wf.setsampwidth(2); wf.setnchannels(1)
        wf.writeframes(frames.tobytes())
    assert blob == ref.getvalue()
