"""The reference-interface mirror on the real device (pytest -m gpu)."""
import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from mimic3_amd.session import InferenceSession, SessionOptions
from oracle.vits_oracle import VitsOracle, audio_float_to_int16

pytestmark = pytest.mark.gpu


def test_session_run_matches_oracle_and_reference_int16(tmp_path):
    cfg = VitsConfig.vctk_low()
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=2.5)
    W.save(tmp_path / "generator.m355", cfg, w)
    so = SessionOptions()
    so.use_deterministic_compute = True
    sess = InferenceSession(str(tmp_path / "generator.onnx"), sess_options=so, providers=["CUDAExecutionProvider"])
    ids = np.expand_dims(np.random.default_rng(3).integers(1, 50, 24).astype(np.int64), 0)
    feed = {"input": ids, "input_lengths": np.array([24], np.int64), "scales": np.array([0.0, 1.0, 0.0], np.float32),
            "sid": np.array([17], np.int64)}
    audio = sess.run(None, feed)[0].squeeze()               # voice.py:230
    ref = VitsOracle(cfg, w).infer(ids, feed["input_lengths"], feed["scales"], sid=feed["sid"])
    assert audio.shape[0] == int(ref["audio_lengths"][0])
    r = ref["audio"][0, 0]
    assert np.sqrt(np.mean((audio - r) ** 2)) / np.sqrt(np.mean(r ** 2)) < 1e-4
    pcm, _ = sess.run_pcm16(feed)
    assert np.array_equal(pcm[0], audio_float_to_int16(audio))  # utils.py:237-244, fused on the GPU


def test_session_is_thread_safe_like_shared_ort_sessions(tmp_path):
    """mimic3_http workers share one session per voice and call run() concurrently (voice.py:277-292)."""
    import threading

    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=2)
    sess = InferenceSession(W.pack(cfg, w))
    rng = np.random.default_rng(0)
    feeds = [{"input": rng.integers(1, 20, (1, 12)).astype(np.int64), "input_lengths": np.array([12]),
              "scales": np.array([0.0, 1.0, 0.0], np.float32)} for _ in range(8)]
    expect = [sess.run(None, f)[0] for f in feeds]
    got = [None] * 8

    def work(i):
        for _ in range(5):
            got[i] = sess.run(None, feeds[i])[0]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e, g in zip(expect, got):
        assert np.array_equal(e, g)


def test_session_loads_generator_onnx_with_folded_weight_norm(tmp_path):
    """The voice as Mimic 3 ships it: generator.onnx + config.json, weight-normed convs folded into anonymous
    initialisers by torch.onnx.export.  No .m355 beside it: converted at load."""
    from tests.onnx_fixture import export_onnx

    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=8, frames_per_id=2.0)
    (tmp_path / "generator.onnx").write_bytes(export_onnx(cfg, w, weight_norm_prefixes=("flow.", "dec.")))
    (tmp_path / "config.json").write_text(cfg.to_json())
    sess = InferenceSession(str(tmp_path / "generator.onnx"))
    assert sess.config.resblock == "2" and tuple(sess.config.upsample_rates) == (8, 8, 4)
    ids = np.expand_dims(np.random.default_rng(5).integers(1, 50, 20).astype(np.int64), 0)
    feed = {"input": ids, "input_lengths": np.array([20], np.int64), "scales": np.array([0.0, 1.0, 0.0], np.float32)}
    audio = sess.run(None, feed)[0].squeeze()
    r = VitsOracle(cfg, w).infer(ids, feed["input_lengths"], feed["scales"])["audio"][0, 0]
    assert audio.shape == r.shape
    assert np.sqrt(np.mean((audio - r) ** 2)) / np.sqrt(np.mean(r ** 2)) < 1e-4


def test_lanes_overlap_concurrent_calls_on_the_device():
    """Three engine handles (HIP streams) behind one session, six threads: same bits as one lane."""
    import threading

    cfg = VitsConfig.tiny_wide()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=12))
    so1, so3 = SessionOptions(), SessionOptions()
    so1.seed = so3.seed = 5
    so3.lanes = 3
    one, three = InferenceSession(blob, sess_options=so1), InferenceSession(blob, sess_options=so3)
    rng = np.random.default_rng(4)
    feeds = [{"input": rng.integers(1, 20, (3, 40)).astype(np.int64), "input_lengths": np.array([40, 17, 29], np.int64),
              "scales": np.array([0.0, 1.0, 0.0], np.float32)} for _ in range(6)]
    expect = [one.run(None, f)[0] for f in feeds]
    got = [None] * 6

    def work(i):
        for _ in range(4):
            got[i] = three.run(None, feeds[i])[0]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e, g in zip(expect, got):
        assert np.array_equal(e, g)


def test_fused_volume_on_the_device_equals_audioop_mul():
    import audioop

    cfg = VitsConfig.apope_low()
    sess = InferenceSession(W.pack(cfg, W.synthetic_weights(cfg, seed=6, frames_per_id=2.0)))
    ids = np.random.default_rng(1).integers(1, 50, (2, 30)).astype(np.int64)
    feed = {"input": ids, "input_lengths": np.array([30, 11], np.int64), "scales": np.array([0.667, 1.0, 0.8], np.float32)}
    sess._seed = 4
    sess._utterances = 0
    base, _ = sess.run_pcm16(feed)
    for vol in (50.0, 33.0, 180.0):
        sess._utterances = 0  # same noise stream as the base run
        rows, _ = sess.run_pcm16(feed, volume=vol)
        for r, b0 in zip(rows, base):
            assert np.array_equal(r, np.frombuffer(audioop.mul(b0.tobytes(), 2, vol / 100.0), dtype=np.int16))


def test_lanes_share_one_weight_replica_and_devices_option(tmp_path):
    """mi355vits_clone: lanes of one device share the weight upload (HBM use grows by workspaces only); the
    multi-device option on a one-GPU box degenerates to devices == [0]; results do not depend on the lane."""
    import torch

    cfg = VitsConfig.vctk_low()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=8, frames_per_id=2.0))
    so = SessionOptions()
    so.devices = "all"
    so.lanes = 3
    so.seed = 11
    free0 = torch.cuda.mem_get_info(0)[0]
    sess = InferenceSession(blob, sess_options=so)
    used = free0 - torch.cuda.mem_get_info(0)[0]
    n_dev = torch.cuda.device_count()
    assert sess.devices == list(range(n_dev)) and len(sess._engines) == 3 * n_dev
    # 76.5 MB of weights (x ~2.4 with the packed copies) once per device, not once per lane
    assert used < n_dev * 400e6, used
    feed = {"input": np.random.default_rng(1).integers(1, 50, (2, 20)).astype(np.int64), "input_lengths": np.array([20, 13]),
            "scales": np.array([0.0, 1.0, 0.0], np.float32), "sid": np.array([5, 77])}
    outs = [sess._engines[k].run(feed["input"], feed["input_lengths"], feed["scales"], feed["sid"])["audio"] for k in range(len(sess._engines))]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    sess.close()


def test_device_side_gather_over_rccl_world_1():
    """The optional result gather on the real stack: the engine's int16 result is wrapped in place (HBM) as a torch
    tensor and moved with one RCCL gather — world_size 1 here (one GPU per box), so this checks the device-pointer
    plumbing and the collective's dtype/shape handling; world_size 2 runs over gloo in the CPU suite."""
    import os

    import torch
    import torch.distributed as dist

    from mimic3_amd import sharding
    from mimic3_amd._native import Engine

    cfg = VitsConfig.tiny()
    eng = Engine(W.pack(cfg, W.synthetic_weights(cfg, seed=5)))
    ids = np.random.default_rng(0).integers(1, 20, (3, 9)).astype(np.int64)
    lens = np.array([9, 4, 7], np.int64)
    ref = eng.run(ids, lens, [0.667, 1.0, 0.8], seed=3, want_float=False, want_pcm16=True)
    eng.run(ids, lens, [0.667, 1.0, 0.8], seed=3, want_float=False, want_pcm16=True, device_only=True)
    blk, ln = sharding.device_pcm_block(eng)
    assert blk.is_cuda and blk.dtype == torch.int16 and tuple(blk.shape) == ref["pcm"].shape
    assert np.array_equal(ln.numpy(), ref["lengths"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        full = sharding.gather_pcm(blk, ln, np.arange(3), 3)
    finally:
        dist.destroy_process_group()
    for b in range(3):
        assert np.array_equal(full[b], ref["pcm"][b, : int(ref["lengths"][b])])
    eng.close()


@pytest.mark.gpu
def test_box_probe_reports_plausible_numbers_and_leaves_the_engine_untouched():
    """mi355vits_probe_device / mi355vits_probe_weights (bench.py carries them in its line so that a slow lease can be told from a
    slow kernel): every figure positive and inside what an MI355X can physically do, the weight-arena probe works on a live handle,
    and a run after the probes gives the same bits as a run before them."""
    from mimic3_amd._native import Engine, default_library, hooks_library

    assert not default_library().has_hooks  # the product library exports include/mi355vits.h only
    lib = hooks_library()                   # the product's objects + include/mi355vits_lab.h
    p = lib.probe_device(0)
    assert p["cus"] > 0
    assert 1e3 < p["l2_stream_GBps"] < 2e5 and 5e2 < p["hbm_copy_GBps"] < 8.1e3, p
    assert 20 < p["l2_hit_latency_ns"] < 2e3 and p["l2_hit_latency_ns"] < p["latency_1GiB_ns"] < 5e3, p
    assert 0 < p["l2_stream_beside_copy_GBps"] <= 1.2 * p["l2_stream_GBps"] and p["table_24MB_stream_GBps"] > 0, p
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=5, frames_per_id=2.0)
    eng = Engine(W.pack(cfg, w), device=0, library=lib)
    ids = np.random.default_rng(5).integers(1, cfg.num_symbols, (2, 40))
    before = eng.run(ids, [40, 31], [0.667, 1.0, 0.8], seed=9, want_pcm16=True)
    q = eng.probe_weights()
    assert q["windows"] >= 1 and 0 < q["arena_stream1_GBps"][0] <= q["arena_stream1_GBps"][2] and 0 < q["arena_stream8_GBps"][0] <= q["arena_stream8_GBps"][2], q
    after = eng.run(ids, [40, 31], [0.667, 1.0, 0.8], seed=9, want_pcm16=True)
    assert np.array_equal(before["audio"], after["audio"]) and np.array_equal(before["pcm"], after["pcm"])
    eng.close()
