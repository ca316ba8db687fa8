"""Host-side mirror of the reference interface (onnxruntime.InferenceSession as Mimic3Voice uses it) and the
batch sharding across ranks.  On CPU these run on the test model of the kernels (tests/emu); the same checks
run against the real library in test_gpu_parity.py / test_gpu_session.py."""
import os
import threading
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
    """One process per (emulated) GPU: shard the feed, synthesise locally with the int16 result left in the engine's
    buffer, gather the padded int16 blocks + lengths on rank 0 with ONE tensor gather (gloo here, RCCL on the GPUs)."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from mimic3_amd import build
    from mimic3_amd._native import Engine, NativeLibrary

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = NativeLibrary(build.EMU_LIB)
    eng = Engine(blob, library=lib)
    rng = np.random.default_rng(0)
    B, Tx = 5, 9
    ids = rng.integers(1, 20, (B, Tx)).astype(np.int64)
    lens = np.array([9, 4, 7, 9, 5], np.int64)
    feed = {"input": ids, "input_lengths": lens, "scales": np.array([0.667, 1.0, 0.8], np.float32)}
    local, rows = sharding.shard_feed(feed, world, rank, balance=True)
    # Philox noise is keyed by the GLOBAL utterance index: rows are run one call each so that any split gives the
    # same audio (a contiguous shard would pass utterance_base = first row instead)
    blocks, lengths = [], []
    for i, g in enumerate(rows):
        eng.run(local["input"][i:i + 1], local["input_lengths"][i:i + 1], local["scales"], seed=1234, utterance_base=int(g),
                want_float=False, want_pcm16=True, device_only=True)
        blk, ln = sharding.device_pcm_block(eng)   # aliases the engine's result buffer: no host copy on the GPU path
        blocks.append(blk.clone())
        lengths.append(int(ln[0]))
    import torch

    L = max([b.shape[1] for b in blocks] + [1])
    block = torch.zeros((len(blocks), L), dtype=torch.int16)
    for i, b in enumerate(blocks):
        block[i, : b.shape[1]] = b[0]
    full = sharding.gather_pcm(block, lengths, rows, B)
    assert (full is None) == (rank != 0)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "gathered.npz"), **{f"a{i}": a for i, a in enumerate(full)})
    # the ragged-list front end packs and calls the same collective
    again = sharding.gather_results([np.asarray(block[i, : lengths[i]]) for i in range(len(rows))], rows, B)
    if rank == 0:
        assert all(np.array_equal(a, b) for a, b in zip(full, again))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process(emu_lib, tmp_path):
    """world_size = 2 over gloo on CPU: the N > 1 path (shard -> per-rank engine -> optional tensor gather of padded
    int16 blocks).  Stochastic scales: the Philox draws depend on the global utterance index only, so the two-rank
    result is bitwise the single-process one."""
    import socket

    import torch.multiprocessing as mp

    from mimic3_amd._native import Engine

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=5))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, blob, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    eng = Engine(blob, library=emu_lib)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, 20, (5, 9)).astype(np.int64)
    lens = np.array([9, 4, 7, 9, 5], np.int64)
    ref = eng.run(ids, lens, [0.667, 1.0, 0.8], seed=1234, utterance_base=0, want_float=False, want_pcm16=True)
    for b in range(5):
        L = int(ref["lengths"][b])
        assert got[f"a{b}"].dtype == np.int16 and np.array_equal(got[f"a{b}"], ref["pcm"][b, :L])
    eng.close()


def test_in_process_multi_device_session_round_robin(tmp_path):
    """SURVEY §8f N2 "device round-robin": ONE session (what the unchanged mimic3-server shares between its worker
    threads, mimic3_http/__main__.py:53-61) over several devices — one weight replica per device, lanes spread over
    them, every call on the least-loaded device, results independent of the device.  Runs on the CPU model of the
    kernels with two emulated devices (MI355_EMU_DEVICES=2), in a subprocess so the setting cannot leak."""
    import subprocess
    import textwrap

    code = textwrap.dedent("""
        import os, sys, threading
        import numpy as np
        sys.path.insert(0, %r)
        from mimic3_amd import build, weights as W
        from mimic3_amd._native import NativeLibrary
        from mimic3_amd.config import VitsConfig
        from mimic3_amd.session import InferenceSession, InvalidArgument, SessionOptions

        lib = NativeLibrary(build.EMU_LIB)
        assert lib.device_count() == 2
        cfg = VitsConfig.tiny()
        blob = W.pack(cfg, W.synthetic_weights(cfg, seed=9))
        feed = lambda n: {"input": np.arange(1, n + 1, dtype=np.int64)[None, :] %% 19 + 1, "input_lengths": np.array([n]),
                          "scales": np.array([0, 1, 0], np.float32)}
        one = InferenceSession(blob, _library=lib)
        assert one.devices == [0]
        so = SessionOptions(); so.devices = "all"; so.lanes = 2
        multi = InferenceSession(blob, sess_options=so, _library=lib)
        assert multi.devices == [0, 1] and [e.device for e in multi._engines] == [0, 1, 0, 1]
        # sequential calls alternate devices (ties broken by least-recently-used) ...
        for n in (5, 6, 7, 8):
            assert np.array_equal(multi.run(None, feed(n))[0], one.run(None, feed(n))[0])
        assert multi._free_lanes.calls_per_device == {0: 2, 1: 2}, multi._free_lanes.calls_per_device
        # ... and concurrent ones spread: 4 lanes, 8 threads, nobody starves, both devices used
        outs = {}
        def work(k):
            outs[k] = multi.run_pcm16(feed(4 + k %% 5))[0][0]
        ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        [t.start() for t in ts]; [t.join() for t in ts]
        for k in range(8):
            assert np.array_equal(outs[k], one.run_pcm16(feed(4 + k %% 5))[0][0])
        c = multi._free_lanes.calls_per_device
        assert c[0] + c[1] == 12 and min(c.values()) >= 4, c
        # device list given explicitly / via the environment; bad lists are InvalidArgument
        so2 = SessionOptions(); so2.devices = [1]
        s2 = InferenceSession(blob, sess_options=so2, _library=lib)
        assert s2.devices == [1] and s2._engines[0].device == 1
        os.environ["MI355VITS_DEVICES"] = "1,0"
        assert InferenceSession(blob, _library=lib).devices == [1, 0]
        del os.environ["MI355VITS_DEVICES"]
        for bad in ([0, 0], [2], "0,x"):
            so3 = SessionOptions(); so3.devices = bad
            try:
                InferenceSession(blob, sess_options=so3, _library=lib)
            except InvalidArgument:
                pass
            else:
                raise AssertionError(bad)
        # micro-batching on top: batches are dealt to the devices' lanes
        so4 = SessionOptions(); so4.devices = "all"; so4.micro_batch_window_ms = 20.0
        mb = InferenceSession(blob, sess_options=so4, _library=lib)
        res = {}
        def w2(k):
            res[k] = mb.run_pcm16(feed(3 + k))[0][0]
        ts = [threading.Thread(target=w2, args=(k,)) for k in range(6)]
        [t.start() for t in ts]; [t.join() for t in ts]
        for k in range(6):
            assert np.array_equal(res[k], one.run_pcm16(feed(3 + k))[0][0])
        for s in (one, multi, s2, mb):
            s.close()
        print("ok")
    """ % ROOT)
    env = dict(os.environ, MI355_EMU_DEVICES="2")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout + p.stderr


def test_session_close_releases_engines_and_dispatcher(emu_lib):
    """a session with micro-batching must be collectable — close() (also run by __del__) stops the
    dispatcher thread, fails queued requests instead of hanging them, and destroys every lane."""
    import gc
    import threading
    import weakref

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=9))
    so = SessionOptions()
    so.micro_batch_window_ms = 1.0
    so.lanes = 2
    before = {t.name for t in threading.enumerate()}
    sess = InferenceSession(blob, sess_options=so, _library=emu_lib)
    feed = {"input": np.array([[3, 7, 1]], np.int64), "input_lengths": np.array([3]), "scales": np.array([0, 1, 0], np.float32)}
    assert sess.run(None, feed)[0].shape[0] == 1
    ref = weakref.ref(sess)
    thread = sess._batcher._thread
    del sess
    gc.collect()
    assert ref() is None, "the dispatcher thread kept the session alive"
    thread.join(timeout=5.0)
    assert not thread.is_alive()
    # explicit close: idempotent, later calls raise
    s2 = InferenceSession(blob, _library=emu_lib)
    s2.close()
    s2.close()
    with pytest.raises(RuntimeError, match="closed"):
        s2.run(None, feed)
    assert {t.name for t in threading.enumerate() if t.name.startswith("mi355vits-microbatch")} <= before


def test_lane_pool_shutdown_wakes_queued_callers():
    """a caller that passed the session's _closed check just before close() and queued behind close()'s own lane
    collection must not wait forever — shutdown() wakes it with "session is closed"; later acquires fail at once."""
    import threading
    import time

    from mimic3_amd.session import _LanePool

    class E:
        device = 0

    e = E()
    pool = _LanePool([e])
    assert pool.acquire() is e  # a call in flight
    order = []

    def closer():  # close(): collects the lane, then shuts the pool down
        pool.acquire()
        order.append("closer got the lane")
        pool.shutdown()

    def late():  # queued behind the closer
        try:
            pool.acquire()
            order.append("late got a lane")
        except RuntimeError as ex:
            order.append(str(ex))

    tc = threading.Thread(target=closer)
    tc.start()
    while len(pool._waiters) < 1:
        time.sleep(0.001)
    tl = threading.Thread(target=late)
    tl.start()
    while len(pool._waiters) < 2:
        time.sleep(0.001)
    pool.release(e)  # the call in flight ends
    tc.join(5.0)
    tl.join(5.0)
    assert not tc.is_alive() and not tl.is_alive()
    assert order == ["closer got the lane", "session is closed"]
    with pytest.raises(RuntimeError, match="closed"):
        pool.acquire()


def test_lane_pool_waiter_interrupted_does_not_swallow_a_lane():
    """a caller interrupted while queued for a lane leaves the queue; a lane handed to it meanwhile goes
    back to the pool."""
    import threading

    from mimic3_amd.session import _LanePool

    class E:
        device = 0

    e = E()
    pool = _LanePool([e])
    assert pool.acquire() is e
    got = []

    class Boom(BaseException):
        pass

    def waiter():
        real_event = threading.Event

        class Ev(real_event):
            def wait(self, timeout=None):  # interrupted as soon as it starts waiting
                raise Boom()

        threading.Event = Ev
        try:
            pool.acquire()
        except Boom:
            got.append("interrupted")
        finally:
            threading.Event = real_event

    t = threading.Thread(target=waiter)
    t.start()
    t.join()
    assert got == ["interrupted"] and len(pool._waiters) == 0
    pool.release(e)
    assert pool.idle() == 1 and pool.acquire() is e


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


def test_chunked_streaming_of_long_form_audio(emu_lib):
    """N4: sentences of a long request are synthesised with a look-ahead window on one shared session (lanes + micro-batching
    underneath) and delivered in order as they finish; the streamed WAV body equals the joined one."""
    import io
    import wave

    from mimic3_amd import postprocess as PP
    from mimic3_amd import streaming as ST

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=3, frames_per_id=2.0))
    so = SessionOptions()
    so.lanes = 2
    so.micro_batch_window_ms = 2.0
    so.seed = 5
    sess = InferenceSession(blob, sess_options=so, _library=emu_lib)
    plain = InferenceSession(blob, _library=emu_lib)
    rng = np.random.default_rng(1)
    sentences = [rng.integers(1, cfg.num_symbols, int(rng.integers(3, 14))).tolist() for _ in range(23)]
    det = (0.0, 1.0, 0.0)  # deterministic scales: the streamed audio must equal one-by-one synthesis exactly
    expect = [plain.run_pcm16(ST._feed(s, det, None))[0][0] for s in sentences]

    consumed = []

    def lazily():
        for i, s in enumerate(sentences):
            consumed.append(i)
            yield s

    got = []
    for k, audio in enumerate(ST.stream_sentences(sess, lazily(), scales=det, look_ahead=4)):
        assert len(consumed) <= k + 1 + 4, "more than look_ahead sentences were pulled ahead of the consumer"
        got.append(audio)
    assert len(got) == len(expect) and all(np.array_equal(a, b) for a, b in zip(got, expect))

    body = b"".join(ST.stream_wav(sess, sentences, scales=det, break_ms=50, look_ahead=6))
    assert body[:4] == b"RIFF" and body[8:16] == b"WAVEfmt " and body[40:44] == struct_pack_u32(ST.STREAM_SIZE)
    joined = PP.utterances_to_wav(expect, 22050, break_ms=50)
    assert body[44:] == joined[44:]  # same PCM as the reference-style joined WAV; only the two size fields differ
    with wave.open(io.BytesIO(joined), "rb") as wf:
        assert wf.getnframes() * 2 == len(body) - 44

    # a failing sentence surfaces at its turn, later ones are not delivered
    bad = sentences[:3] + [[cfg.num_symbols + 5]] + sentences[3:6]
    out = []
    with pytest.raises(ValueError, match="phoneme id"):
        for a in ST.stream_sentences(sess, bad, scales=det, look_ahead=3):
            out.append(a)
    assert len(out) == 3
    sess.close()
    plain.close()


def struct_pack_u32(v):
    import struct

    return struct.pack("<I", v)


def test_bench_launch_plan_with_stubbed_device_counts():
    """`python bench.py --gpus N` must start by itself.  bench.plan_launch decides from (--gpus, --single-process,
    the launcher's environment, the visible device count) without touching torch: plain N > 1 re-executes under
    torch.distributed.run (one rank per GPU, loopback rendezvous), a launched rank goes on, too few devices is the ONE error."""
    import sys

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bench import plan_launch

    assert plan_launch(1, False, {}, 1) == ("run",)
    assert plan_launch(1, False, {}, 8) == ("run",)
    kind, msg = plan_launch(8, False, {}, 1)            # this box: one device
    assert kind == "error" and msg == "8 devices requested, 1 visible"
    kind, msg = plan_launch(2, False, {}, 0)
    assert kind == "error" and "no HIP device" in msg
    kind, make = plan_launch(8, False, {}, 8)           # an 8-GPU node, no launcher: re-exec
    assert kind == "reexec"
    cmd = make("/usr/bin/python3", "/x/bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7:] == ["/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    # under the launcher (what the driver does for N > 1): every rank goes on; a disagreeing world size is refused
    env = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}
    assert plan_launch(4, False, env, 8) == ("run",)
    assert plan_launch(8, False, env, 8)[0] == "error"
    assert plan_launch(4, False, env, 2) == ("error", "4 devices requested, 2 visible (LOCAL_RANK 3)")
    # WORLD_SIZE=1 exported by a launcher with --nproc-per-node 1 is a launched run too
    assert plan_launch(1, False, {"WORLD_SIZE": "1"}, 1) == ("run",)
    assert plan_launch(2, False, {"WORLD_SIZE": "1"}, 2)[0] == "error"
    # the threads-in-one-process shape of the reference server stays available
    assert plan_launch(4, True, {}, 4) == ("single",)
    assert plan_launch(4, True, {}, 1) == ("error", "4 devices requested, 1 visible")


def test_bench_cli_fails_only_for_missing_devices():
    """The command itself, as the driver types it: on a box without (enough) devices it exits with the device-count message,
    not with a usage error about launchers."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    tail = (r.stderr or r.stdout).strip().splitlines()[-1]
    assert "devices requested" in tail or "no HIP device visible" in tail, tail
    assert "torch.distributed.run" not in tail and "--single-process" not in tail


def test_planned_batches_of_a_known_request(emu_lib):
    """SURVEY N2 (tts.py:470-515: end_utterance holds every pending sentence): stream_sentences on a LIST plans its batches —
    sentence 0 alone, a head batch, the rest length-sorted inside its phoneme-length class — and yields exactly the chunks of
    one call per sentence, in order; the plan covers every sentence once; padding efficiency is reported."""
    from mimic3_amd import streaming as ST

    lens = [5, 9, 3, 130, 12, 7, 8, 200, 6, 11, 4, 140, 10, 9, 9, 300, 2, 13, 5, 6, 7, 8, 9, 10, 11]
    plan = ST.plan_batches(lens, head=4, max_batch=6)
    assert plan[0] == [0] and sorted(i for b in plan for i in b) == list(range(len(lens)))
    assert [1, 2, 4] in plan and [3] in plan  # the head window, split by phoneme-length class (130 > 128)
    assert [1, 2, 4] not in ST.plan_batches(lens, max_batch=6)  # default: no head window, sentence 0 alone and then the sorted groups
    for b in plan:
        assert len(b) <= 6 and len({ST._tx_class(lens[i]) for i in b}) == 1
    tail = [b for b in plan[1:] if min(b) > 4 and ST._tx_class(lens[b[0]]) == 0]
    assert len(tail) == 3 and all(5 <= len(b) <= 6 for b in tail)  # 17 short sentences in equal shares, not 6 + 6 + 5 by accident
    spans = [max(lens[i] for i in b) - min(lens[i] for i in b) for b in tail]
    assert max(spans) <= 5 and sum(spans) <= 10, spans  # length-sorted: rows of one batch are alike (unsorted windows of six span 7 - 11)
    assert ST.plan_batches([], head=3) == [] and ST.plan_batches([7], head=3) == [[0]]

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=3, frames_per_id=2.0))
    so = SessionOptions()
    so.lanes = 2
    so.seed = 5
    sess = InferenceSession(blob, sess_options=so, _library=emu_lib)
    plain = InferenceSession(blob, _library=emu_lib)
    rng = np.random.default_rng(2)
    sentences = [rng.integers(1, cfg.num_symbols, int(rng.integers(3, 30))).tolist() for _ in range(19)]
    det = (0.0, 1.0, 0.0)
    expect = [plain.run_pcm16(ST._feed(s, det, None))[0][0] for s in sentences]
    stats = {}
    got = list(ST.stream_sentences(sess, sentences, scales=det, look_ahead=4, stats=stats))
    assert len(got) == len(expect) and all(np.array_equal(a, b) for a, b in zip(got, expect))
    assert sum(stats["batches"]) == 19 and stats["batches"][0] == 1 and 0.3 < stats["padding_efficiency"] <= 1.0
    # the lazy path (a generator) gives the same chunks
    lazy = list(ST.stream_sentences(sess, (s for s in sentences), scales=det, look_ahead=4))
    assert all(np.array_equal(a, b) for a, b in zip(lazy, expect))
    # a failing sentence inside a planned batch: its batch-mates are still delivered up to its turn
    bad = [list(s) for s in sentences]
    bad[9] = [cfg.num_symbols + 5]
    out = []
    with pytest.raises(ValueError, match="phoneme id"):
        for a in ST.stream_sentences(sess, bad, scales=det, look_ahead=4):
            out.append(a)
    assert len(out) == 9 and all(np.array_equal(a, b) for a, b in zip(out, expect))
    sess.close()
    plain.close()


def test_micro_batcher_collects_while_every_lane_is_busy(emu_lib):
    """Round 6: a batch is not cut by the clock while the lanes are busy.  Closed-loop clients on one lane: the first request
    runs alone at once (no window for a lone call on an idle session), everything that arrives meanwhile forms the next batch —
    so the mean batch under load is far above what a 1 ms window alone collects; results are bitwise the plain session's."""
    import time

    cfg = VitsConfig.tiny()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=3, frames_per_id=2.0))
    so = SessionOptions()
    so.micro_batch_window_ms = 1.0
    so.micro_batch_max = 16
    sess = InferenceSession(blob, sess_options=so, _library=emu_lib)
    plain = InferenceSession(blob, _library=emu_lib)
    rng = np.random.default_rng(4)
    feeds = [{"input": rng.integers(1, cfg.num_symbols, (1, 9)).astype(np.int64), "input_lengths": np.array([9], np.int64),
              "scales": np.array([0.0, 1.0, 0.0], np.float32)} for _ in range(12)]
    expect = [plain.run_pcm16(f)[0][0] for f in feeds]
    t0 = time.perf_counter()
    lone = sess.run_pcm16(feeds[0])[0][0]
    t_lone = time.perf_counter() - t0
    assert np.array_equal(lone, expect[0])
    t0 = time.perf_counter()
    plain.run_pcm16(feeds[0])
    t_plain = time.perf_counter() - t0
    assert t_lone < t_plain + 0.9e-3 + 0.5 * t_plain, (t_lone, t_plain)  # the 1 ms window was not waited for
    results = [None] * 12
    b0, r0 = sess._batcher.batches, sess._batcher.requests

    def client(k):
        for rep in range(3):
            results[k] = sess.run_pcm16(feeds[k])[0][0]

    ts = [threading.Thread(target=client, args=(k,)) for k in range(12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(np.array_equal(results[k], expect[k]) for k in range(12))
    nb, nr = sess._batcher.batches - b0, sess._batcher.requests - r0
    assert nr == 36 and nr / nb >= 4.0, (nr, nb)  # (the CPU model takes ~10 ms per call: a 1 ms window alone would give ~1-2)
    sess.close()
    plain.close()
