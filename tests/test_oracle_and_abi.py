"""CPU-side checks: the oracle against its golden vectors, the weight container, the config mirror, and that
the product library loads and exports the whole C ABI (no compute without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd.config import CVitsConfig, VitsConfig
from oracle.vits_oracle import VitsOracle, audio_float_to_int16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_int16_conversion_against_reference_vectors():
    """tests/golden/int16_reference.npz holds outputs of the reference's own audio_float_to_int16."""
    g = np.load(os.path.join(GOLDEN, "int16_reference.npz"))
    for i in range(int(g["n"])):
        assert np.array_equal(audio_float_to_int16(g[f"in{i}"]), g[f"out{i}"])
    assert list(audio_float_to_int16(np.array([0, .5, -.25, 1e-4], np.float32))) == [0, 32767, -16383, 6]


@pytest.mark.parametrize("name", ["hf_tiny_resblock1.npz", "hf_tiny_resblock1_multispeaker.npz"])
def test_oracle_against_hf_vits_fixture(name):
    """Outputs of HuggingFace VitsModel (independent implementation) frozen by oracle/hf_crosscheck.py."""
    g = np.load(os.path.join(GOLDEN, name))
    cfg = VitsConfig.from_json(str(g["config_json"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    sid = np.full(len(g["lengths"]), int(g["sid"][0])) if "sid" in g.files else None
    r = VitsOracle(cfg, w).infer(g["ids"], g["lengths"], [0, 1, 0], sid=sid, batch_semantics="upstream")
    assert np.array_equal(r["audio_lengths"], g["hf_lengths"])
    for b in range(len(g["lengths"])):
        L = int(g["hf_lengths"][b])
        a, h = r["audio"][b, 0, :L], g["hf_waveform"][b, :L]
        assert np.sqrt(np.mean((a - h) ** 2)) / np.sqrt(np.mean(h ** 2)) < 2e-5


def test_oracle_against_hf_vits_full_size_fixture():
    g = np.load(os.path.join(GOLDEN, "hf_low_resblock1_decimated.npz"))
    cfg = VitsConfig.from_json(str(g["config_json"]))
    w = W.synthetic_weights(cfg, seed=int(g["seed"]), frames_per_id=float(g["frames_per_id"]))
    cs = float(sum(float(np.abs(v).sum()) for v in w.values()))
    if abs(cs - float(g["weight_checksum"])) > 1e-6 * cs:
        pytest.skip("numpy Generator stream differs from the one that made the fixture")
    r = VitsOracle(cfg, w).infer(g["ids"], g["lengths"], [0, 1, 0])
    assert np.array_equal(r["audio_lengths"], g["hf_lengths"])
    a, h = r["audio"][:, 0, ::16], g["hf_waveform_dec16"]
    assert np.sqrt(np.mean((a - h) ** 2)) / np.sqrt(np.mean(h ** 2)) < 2e-5


def test_oracle_golden_is_reproducible():
    g = np.load(os.path.join(GOLDEN, "oracle_apope_low_b1.npz"))
    cfg = VitsConfig.from_json(str(g["config_json"]))
    w = W.synthetic_weights(cfg, seed=int(g["seed"]), frames_per_id=float(g["frames_per_id"]))
    r = VitsOracle(cfg, w).infer(g["ids"], g["lengths"], [0, 1, 0])
    assert np.array_equal(r["audio_lengths"], g["audio_lengths"])
    assert np.array_equal(r["w_ceil"].astype(np.int32), g["w_ceil"])
    assert np.abs(r["audio"][:, 0] - g["audio"]).max() < 1e-4


def test_oracle_per_row_batch_equals_single_rows():
    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=3)
    ora = VitsOracle(cfg, w)
    ids = np.array([[3, 4, 5, 6, 7], [8, 9, 1, 0, 0]], np.int64)
    full = ora.infer(ids, np.array([5, 3]), [0, 1, 0])
    one = ora.infer(ids[1:, :3], np.array([3]), [0, 1, 0])
    L = int(one["audio_lengths"][0])
    assert np.abs(full["audio"][1, 0, :L] - one["audio"][0, 0, :L]).max() < 1e-6


def test_model_shapes_match_voice_file_sizes():
    """15,610,907 fp32 parameters (62,443,628 B) vs generator.onnx 62,792,219 B for apope_low;
    768 B per extra symbol; 2,048 B per extra speaker."""
    assert W.count_parameters(VitsConfig.apope_low(50)) == 15_610_907
    assert (W.count_parameters(VitsConfig.apope_low(51)) - W.count_parameters(VitsConfig.apope_low(50))) * 4 == 768
    a = VitsConfig.vctk_low()
    b = VitsConfig.vctk_low()
    b.n_speakers = 110
    assert (W.count_parameters(b) - W.count_parameters(a)) * 4 == 2048
    multi_extra = (W.count_parameters(VitsConfig.vctk_low()) - W.count_parameters(VitsConfig.apope_low())) * 4 - 109 * 2048
    assert abs(multi_extra - 13_530_880 + 2 * 2048) < 4096  # 4 cond layers + dec.cond + dp.cond (SURVEY table)


def test_container_roundtrip_and_config_mirror():
    cfg = VitsConfig.tiny(n_speakers=3)
    w = W.synthetic_weights(cfg, seed=1)
    cfg2, w2 = W.unpack(W.pack(cfg, w))
    assert cfg2 == cfg
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    j = json.loads(cfg.to_json())
    assert VitsConfig.from_json(j) == cfg
    with pytest.raises(ValueError):
        W.check_weights(cfg, {k: v for k, v in list(w.items())[1:]})
    bad = VitsConfig.tiny()
    bad.resblock = "3"
    with pytest.raises(ValueError):
        bad.validate()


def test_c_struct_matches_header():
    """Field order of CVitsConfig == struct mi355vits_config in include/mi355vits.h."""
    hdr = open(os.path.join(ROOT, "include", "mi355vits.h")).read()
    body = hdr[hdr.index("typedef struct mi355vits_config {"):hdr.index("} mi355vits_config;")]
    names = re.findall(r"(?:int32_t|float)\s+(\w+)", body)
    assert names == [f[0] for f in CVitsConfig._fields_]
    assert ctypes.sizeof(CVitsConfig) == 4 * (len(names) + 4 * 7 + 63)  # arrays: 4 x [8] + 1 x [64]


def test_product_library_exports_the_c_abi():
    from mimic3_amd import build
    from mimic3_amd._native import EXPORTED_SYMBOLS, NativeLibrary

    hdr = open(os.path.join(ROOT, "include", "mi355vits.h")).read()
    declared = set(re.findall(r"\b(mi355vits_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    lib = NativeLibrary(build.build_hip())  # cross-compiles for gfx950 when stale; loading needs no GPU
    assert "gfx950" in lib.version()
    # the product / lab split: unit-test, micro-benchmark and probe hooks are declared in include/mi355vits_lab.h
    # and exported by libmi355vits_hooks.so (the product's objects + csrc/lab_api.cpp), the lab build and the CPU model — the
    # product library carries none of them
    import subprocess

    from mimic3_amd._native import HOOKS_LIBRARY, LAB_SYMBOLS

    lab_hdr = open(os.path.join(ROOT, "include", "mi355vits_lab.h")).read()
    lab_declared = set(re.findall(r"\b(mi355vits_[a-z0-9_]+)\s*\(", lab_hdr))
    assert lab_declared == set(LAB_SYMBOLS) and not (lab_declared & declared), lab_declared ^ set(LAB_SYMBOLS)
    assert not lib.has_hooks
    dyn = subprocess.run(["nm", "-D", "--defined-only", lib.path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(mi355vits_[a-z0-9_]+)\b", dyn))
    assert exported == declared, exported ^ declared  # nothing beyond include/mi355vits.h leaves the product library
    hooks = NativeLibrary(HOOKS_LIBRARY)
    assert hooks.has_hooks and "gfx950" in hooks.version()
    with pytest.raises(RuntimeError, match="mi355vits_lab.h"):
        lib.test_mfma_layout()


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the product engine must fail loudly; it must never import the oracle or the CPU model."""
    import torch

    from mimic3_amd._native import Engine, NativeError

    src = ""
    for f in os.listdir(os.path.join(ROOT, "mimic3_amd")):
        if f.endswith(".py") and f != "build.py":  # build.py only compiles the test model, it never loads it
            src += open(os.path.join(ROOT, "mimic3_amd", f)).read()
    assert "import oracle" not in src and "from oracle" not in src
    assert "libmi355vits_emu" not in src.replace("tests/emu/libmi355vits_emu.so", "")
    if not torch.cuda.is_available():
        cfg = VitsConfig.tiny()
        with pytest.raises(NativeError):
            Engine(W.pack(cfg, W.synthetic_weights(cfg, seed=1)))


def test_plain_c_client_of_the_header(emu_lib, tmp_path):
    """include/mi355vits.h compiled as C (gcc -std=c99 -pedantic), linked against the library (here: its CPU model),
    driving create / get_config / run / free_result / last_error / destroy — the binding any other host language
    would make.  Its PCM equals what the ctypes path returns."""
    from tests.util import run_c_client

    run_c_client(emu_lib, tmp_path)


def test_corrupted_containers_are_rejected_not_fatal(emu_lib):
    """400 corrupted .m355 blobs (bit flips in header / config / tensor table, truncations, extreme field values, random
    bytes behind the magic) through mi355vits_create_from_buffer: every one ends in an error code or a loadable voice.
    Runs in a subprocess so that a crash would fail this test instead of killing pytest."""
    import subprocess
    import sys

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_container.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    assert "rejected" in p.stdout
