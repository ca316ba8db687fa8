#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: 22.05 kHz audio samples/s (whole job) + RTF for the
en_UK/apope_low VITS graph, hot path only (``onnx_model.run`` + ``audio_float_to_int16``, the reference's
own timed region, mimic3_tts/voice.py:229-232).

    python bench.py --gpus N --steps K --warmup W    # N > 1 without a launcher: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (what the driver launches for N > 1)
    python bench.py --gpus N --single-process        # one process, N devices (the unchanged mimic3-server's shape)

A "step" is one pass of the hot path over the job's batch of synthetic phoneme ids.  Workload: BASELINE.json's
batch of 256 independent utterances of 128 phoneme ids, durations forced to 6 frames/id -> 768 latent frames =
196,608 samples = 8.916 s per utterance (SURVEY.md §8d "unit S"), sharded by utterance over the GPUs: all 256 on the
one GPU at ``--gpus 1`` (the configuration the metric names; ``value`` IS the batch-256 number), 256 / N per GPU at
``--gpus N`` (32 each on eight: BASELINE configs[3]) — strong scaling of a fixed job, no collective on the data path.
``--batch B`` fixes the per-GPU batch instead (weak scaling; ``--batch 32`` = rounds 1-4's headline shape).

Timed region = what the reference times around ``run`` + int16: phoneme ids start in host memory (1 KiB per
utterance), the int16 result ENDS in host memory (pinned buffers recycled by the library, D2H inside the
region): ``value`` is host-to-host.  ``device_only`` in the JSON is the same loop with the int16 result left
in HBM (compute-side number).  Batch-1 latency / RTF on the reference's golden-utterance shape (180 ids ->
991 frames) is measured in the same run ("latency_b1"), and BASELINE.json configs[2] (en_US/vctk_low,
32 x 128 ids, sid = b mod 109) as "vctk_low_b32" with its own roofline.

The JSON line also carries
  "roofline"     — dominant kernel group, algorithmic FLOPs / launch time measured with HIP events on the
                   engine's stream (mi355vits_profile_*), against the matrix-core peak of its path, plus HBM numbers;
                   "traffic" = HBM bytes per launch MEASURED IN THIS RUN by two counters-only rocprofv3 passes of a
                   3-step subprocess (or null: --no-traffic, rocprofv3 missing, extra legs) — never a stored table
  "host_ms_per_step" — one handle driven sequentially: call wall time vs device time
  "cpu_baseline" — the PyTorch-CPU oracle (onnxruntime is not installed here) timed on this host's cores on
                   a bounded sample of the same workload (rank 0, N = 1 only), with the engine checked against it
  first level    — the metric's other configurations as NUMBERS: latency_b1_ms / b1_x_realtime / b1_launches (configs[1]), b256_ms,
                   b32_ms / b32_samples_per_s (the 32-row shard of the eight-GPU run on this one device), longform_first_audio_ms /
                   longform_total_ms / longform_padding_efficiency (configs[4]), serve64_* / ragged_b1_calls_x_realtime (the reference's real
                   call shape: one ragged sentence per call), probe_* (the box probe: L2-hit stream GB/s, L2-hit latency, HBM copy GB/s)
  "extra"        — vctk_low b32 (configs[2]), the 32-row shard on ONE GPU with its kernel table, long-form streaming (configs[4]:
                   120 sentences through mimic3_amd.streaming on one session, default math and bf16 weights), the f32-MFMA /
                   bf16-weights / f16x2 math modes; they run after the headline's handles are closed (open idle handles alias
                   HIP streams onto shared hardware queues: tools/floor_diag2.py).
  "device"       — (after an untimed pre-heat of the same steps until the shader clock is steady: "device.preheat")
                   shader / memory / fabric clocks, socket power and cap, partition modes of the HIP device from sysfs, sampled
                   every 10 ms over the timed loop and over the per-kernel table; one-line summaries of it, of the batch-1
                   latency, the batch-256 leg and the streaming leg sit in "config" (the driver's record keeps that object).
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide); MATH_BF16X3 spends 6 bf16 products per f32 multiply-add
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0      # HBM3E spec
SAMPLE_RATE = 22050


class DeviceMonitor:
    """What the box was doing while the numbers were taken: the HIP device's sysfs node (matched by PCI bus id) gives the
    shader / memory / fabric clock levels, the power cap, the performance level and the compute / memory partition modes;
    a sampler thread reads the live shader clock and socket power every 10 ms inside a window (the timed loop, the
    per-kernel table).  A slow box then explains itself in the JSON line (one kernel 40 % slower on the driver's
    box with nothing in the record to say why).  Everything is best effort: missing files give None, never an error."""

    def __init__(self, torch_index=0):
        import glob

        self.dir = self.hwmon = None
        try:
            import torch

            p = torch.cuda.get_device_properties(torch_index)
            bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            for d in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.basename(os.path.realpath(d)) == bus:
                    self.dir = d
                    break
            self.bus = bus
        except Exception:  # noqa: BLE001 - diagnostics only
            self.bus = None
        if self.dir is None:  # a box that shows exactly one GPU node: that one
            cands = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
            if len(cands) == 1:
                self.dir = cands[0]
        if self.dir:
            hw = sorted(glob.glob(os.path.join(self.dir, "hwmon", "hwmon*")))
            self.hwmon = hw[0] if hw else None
        self._stop = None
        self._samples = []

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    def _level(self, name):
        """(current MHz, [levels MHz]) of a pp_dpm_* file."""
        txt = self._read(os.path.join(self.dir, name)) if self.dir else None
        if not txt:
            return None, []
        cur, levels = None, []
        for line in txt.splitlines():
            try:
                mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            except (IndexError, ValueError):
                continue
            levels.append(mhz)
            if line.rstrip().endswith("*"):
                cur = mhz
        return cur, levels

    def _num(self, name, scale):
        v = self._read(os.path.join(self.hwmon, name)) if self.hwmon else None
        try:
            return float(v) * scale
        except (TypeError, ValueError):
            return None

    def snapshot(self):
        if not self.dir:
            return {"available": False, "pci_bus": self.bus}
        sclk, sclk_levels = self._level("pp_dpm_sclk")
        return {
            "available": True, "pci_bus": self.bus, "sysfs": self.dir,
            "sclk_mhz": self._num("freq1_input", 1e-6) or sclk, "sclk_levels_mhz": sclk_levels,
            "mclk_mhz": self._level("pp_dpm_mclk")[0], "fclk_mhz": self._level("pp_dpm_fclk")[0], "socclk_mhz": self._level("pp_dpm_socclk")[0],
            "power_w": self._num("power1_input", 1e-6), "power_cap_w": self._num("power1_cap", 1e-6),
            "temp_c": self._num("temp2_input", 1e-3),
            "perf_level": self._read(os.path.join(self.dir, "power_dpm_force_performance_level")),
            "compute_partition": self._read(os.path.join(self.dir, "current_compute_partition")),
            "memory_partition": self._read(os.path.join(self.dir, "current_memory_partition")),
        }

    def start(self):
        self._samples = []
        if not self.hwmon:
            return
        self._stop = threading.Event()

        def loop(stop=self._stop, out=self._samples):
            while not stop.is_set():
                out.append((self._num("freq1_input", 1e-6), self._num("power1_input", 1e-6)))
                stop.wait(0.01)

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        """-> {"sclk_mhz": {min, median, max}, "power_w": {mean, max}, "samples": n} of the window, or None."""
        if self._stop is None:
            return None
        self._stop.set()
        self._thread.join(timeout=1.0)
        self._stop = None
        f = [a for a, _ in self._samples if a]
        w = [b for _, b in self._samples if b]
        if not f:
            return None
        return {"sclk_mhz": {"min": min(f), "median": float(np.median(f)), "max": max(f)},
                "power_w": {"mean": float(np.mean(w)) if w else None, "max": max(w) if w else None}, "samples": len(f)}

    @staticmethod
    def brief(snap, window):
        """One line (< 128 characters: the driver's record keeps strings that short) for `config.device`."""
        if not snap.get("available"):
            return "device state unavailable (no sysfs node for the HIP device)"
        w = window or {}
        sc = w.get("sclk_mhz") or {}
        pw = w.get("power_w") or {}
        return ("sclk %s-%s MHz (median %s) mclk %s fclk %s; %s W mean (cap %s); %s/%s perf=%s" % (
            _i(sc.get("min")), _i(sc.get("max")), _i(sc.get("median")), _i(snap.get("mclk_mhz")), _i(snap.get("fclk_mhz")),
            _i(pw.get("mean")), _i(snap.get("power_cap_w")), snap.get("compute_partition"), snap.get("memory_partition"),
            snap.get("perf_level")))[:127]


def _i(v):
    return "?" if v is None else str(int(round(v)))


def make_batch(B, Tx, base):
    ids = np.stack([np.random.default_rng(1234 + base + b).integers(1, 50, Tx) for b in range(B)]).astype(np.int64)
    return ids, np.full(B, Tx, np.int64)


class Workload:
    """One voice on one or more devices: ``streams`` engine handles per device, steps dealt to them in turn."""

    def __init__(self, cfg, weights, devices, streams, B, Tx, fpi, rank, world, multispeaker_sid=False, math=None):
        from mimic3_amd import weights as W
        from mimic3_amd._native import Engine

        self.cfg, self.B, self.Tx, self.fpi, self.rank, self.world = cfg, B, Tx, fpi, rank, world
        blob = W.pack(cfg, weights)
        self.engines = []
        for d in devices:  # one weight replica per device, the further handles of a device share it
            first = Engine(blob, device=d)
            if math:
                first.set_math(math)
            self.engines.append(first)
            self.engines.extend(first.clone() for _ in range(max(1, streams) - 1))
        self.math = self.engines[0].math
        self.n_devices = len(devices)
        self.ids, self.lengths = make_batch(B, Tx, rank * B)
        self.forced = np.full((B, Tx), fpi, np.int32)
        self.scales = np.array([0.667, 1.0, 0.8], np.float32)
        self.sid = (np.arange(B) % cfg.n_speakers).astype(np.int64) if multispeaker_sid else None

    def step(self, i, e=None, device_only=False):
        e = e or self.engines[0]
        return e.run(self.ids, self.lengths, self.scales, self.sid, forced_durations=self.forced, seed=1,
                     utterance_base=self.rank * self.B + i * self.world * self.B, want_float=False, want_pcm16=True,
                     device_only=device_only)

    def run_steps(self, n, device_only=False):
        """n steps; with S > 1 handles, S host threads each drive one engine handle (own HIP stream and workspace)
        and take step numbers from a shared counter, so the small-kernel front half of one batch overlaps the
        matrix-core back half of another.  Every step is a complete, independent run() whose int16 result is in
        host memory when it returns (unless device_only)."""
        engines = self.engines
        if len(engines) == 1:
            last = None
            for i in range(n):
                last = self.step(i, device_only=device_only)
            return last
        counter = itertools.count()
        outs = [None] * len(engines)
        errs = []

        def work(k):
            try:
                while True:
                    i = next(counter)
                    if i >= n:
                        return
                    outs[k] = self.step(i, engines[k], device_only)
            except Exception as ex:  # noqa: BLE001 - re-raised below
                errs.append(ex)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(engines))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise errs[0]
        return next((o for o in outs if o is not None), None)

    def size_workspaces(self):
        for e in self.engines:  # every handle sizes its workspace on first use: never inside the timed region
            self.step(0, e)

    def close(self):
        for e in self.engines:
            e.close()


def kernel_table(eng, step, nsteps):
    """Per-kernel HIP-event table of `nsteps` profiled steps on one handle."""
    eng.profile_enable(True)
    eng.profile_reset()
    for i in range(nsteps):
        step(i)
    rep = eng.profile_report()
    eng.profile_enable(False)
    eng.profile_reset()
    tot_ms = sum(v["ms"] for v in rep.values())
    table = []
    for name, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
        sec = v["ms"] * 1e-3
        table.append({
            "kernel": name, "launches_per_step": v["calls"] // nsteps, "ms_per_step": v["ms"] / nsteps,
            "share": v["ms"] / tot_ms if tot_ms else 0.0,
            "tflops": v["flops"] / sec / 1e12 if sec > 0 else 0.0,
            "gbs_algorithmic": v["bytes"] / sec / 1e9 if sec > 0 else 0.0,
        })
    return rep, table, tot_ms


def roofline_of(rep, table, tot_ms, nsteps, workload_key, voice, math="f32", traffic=None):
    """`traffic`: measure_traffic_in_run's result for the dominant kernel, or None (then "traffic" is null: nothing is quoted
    from a stored table)."""
    dom = table[0]
    drec = rep[dom["kernel"]]
    dsec = drec["ms"] * 1e-3
    # the roof of the dominant kernel: kernels named *_b3 / running in MATH_BF16X3 execute six bf16 MFMA products per
    # algorithmic f32 multiply-add, so their matrix-core roof is 2500 / 6 TFLOP/s of ALGORITHMIC f32 work
    on_bf16 = math in ("bf16x3", "f16x2") and any(t in dom["kernel"] for t in ("mrf", "wn_layer_b3", "dec.rb", "upsample.s0", "upsample.s1", "conv_pre"))
    on_f16x2 = math == "f16x2" and "mrf_fused" in dom["kernel"]  # three f16 MFMA products per multiply-add (same MFMA rate)
    mfma = "v_mfma_f32_16x16x32_bf16" if "mrf_p" in dom["kernel"] else "v_mfma_f32_32x32x16_bf16"
    peak = PEAK_BF16_TFLOPS / 3.0 if on_f16x2 else (PEAK_BF16X3_TFLOPS if on_bf16 else PEAK_FP32_TFLOPS)
    return {
        "kernel": dom["kernel"],
        "bound": "mfma",
        "achieved": drec["flops"] / dsec / 1e12,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": drec["flops"] / dsec / 1e12 / peak,
        "matrix_core_path": ("v_mfma_f32_32x32x16_f16 x 3 partial products per multiply-add (operands as 2 x fp16 terms): peak = 2500 / 3 TFLOP/s"
                             if on_f16x2 else
                             (mfma + " x 6 partial products per f32 multiply-add (operands split 3 x bf16, f32 "
                              "accumulate): peak = 2500 TFLOP/s dense bf16 / 6") if on_bf16 else "v_mfma_f32_32x32x2_f32: peak = 157.3 TFLOP/s"),
        "frac_of_f32_mfma_peak": drec["flops"] / dsec / 1e12 / PEAK_FP32_TFLOPS,
        "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
        "traffic_detail": traffic,
        "algorithmic_flops_per_launch": drec["flops"] / drec["calls"],
        "algorithmic_bytes_per_launch": drec["bytes"] / drec["calls"],
        "avg_launch_us": drec["ms"] * 1e3 / drec["calls"],
        "launches": drec["calls"],
        "hbm_algorithmic_gbs": drec["bytes"] / dsec / 1e9,
        "hbm_frac": drec["bytes"] / dsec / 1e9 / PEAK_HBM_GBS,
        "note": "achieved = ALGORITHMIC f32 FLOPs (2 x Cout x Cin x K per output sample, halo recompute not counted) / launch "
                "time by HIP events on the engine stream; the dense Conv1d stacks are matrix-core-bound on MI355X (AI >= 24 "
                "FLOP/B vs ridge 19.7 at f32); hbm_* give the same launches against the 8 TB/s HBM roof as BASELINE asks; "
                "traffic = HBM bytes per launch measured in this run by rocprofv3 counter passes in a subprocess (null when "
                "not measured: extra legs, rocprofv3 missing)",
        "whole_step": {
            "tflops": sum(v["flops"] for v in rep.values()) / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0,
            "hbm_algorithmic_gbs": sum(v["bytes"] for v in rep.values()) / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0,
            "kernel_ms_per_step": tot_ms / nsteps,
        },
    }


def print_table(title, table):
    print(title, file=sys.stderr)
    for row in table:
        print(f"  {row['kernel']:22s} {row['launches_per_step']:4d} launches  {row['ms_per_step']:9.3f} ms/step "
              f"{100 * row['share']:5.1f}%  {row['tflops']:7.2f} TFLOP/s  {row['gbs_algorithmic']:8.1f} GB/s", file=sys.stderr)


def plan_launch(gpus, single_process, env, visible_devices):
    """How `bench.py --gpus N` starts (no torch, no device: tests/test_session_and_sharding.py drives it with stubbed counts).

    ("run",)                  this process is one rank of a WORLD_SIZE == N job (or N == 1): go on
    ("single",)               --single-process: one process, one host thread per device (the mimic3-server shape)
    ("reexec", make_cmd)      WORLD_SIZE unset and N > 1: `python bench.py --gpus N` re-executes itself as
                              `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
                              (one rank per GPU over RCCL, exactly what the driver launches for N > 1)
    ("error", message)        N devices requested, fewer visible / WORLD_SIZE disagrees with --gpus
    """
    launched = "WORLD_SIZE" in env
    world = int(env.get("WORLD_SIZE", "1"))
    if gpus < 1:
        return ("error", f"--gpus {gpus}: need at least one device")
    if visible_devices < 1:
        return ("error", "bench.py needs an MI355X (no HIP device visible); there is no CPU fallback for the measured path")
    if single_process and world == 1:
        if visible_devices < gpus:
            return ("error", f"{gpus} devices requested, {visible_devices} visible")
        return ("single",)
    if launched:
        if world != gpus:
            return ("error", f"--gpus {gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
        local = int(env.get("LOCAL_RANK", "0"))
        if local >= visible_devices:
            return ("error", f"{gpus} devices requested, {visible_devices} visible (LOCAL_RANK {local})")
        return ("run",)
    if gpus == 1:
        return ("run",)
    if visible_devices < gpus:
        return ("error", f"{gpus} devices requested, {visible_devices} visible")

    def make_cmd(python, script, argv):
        import socket

        with socket.socket() as s:  # a free rendezvous port on the loopback
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        return [python, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                "--master-port", str(port), script] + list(argv)

    return ("reexec", make_cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of about 3 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="utterances per GPU per step.  Default: BASELINE's batch 256 divided over the GPUs (256 on one GPU — the "
                         "configuration the metric names —, 32 each on eight: strong scaling of a fixed job); given explicitly, every "
                         "GPU gets that many whatever --gpus is (weak scaling)")
    ap.add_argument("--tx", type=int, default=128)
    ap.add_argument("--frames-per-id", type=int, default=6)
    ap.add_argument("--voice", choices=["apope_low", "vctk_low"], default="apope_low",
                    help="headline voice (BASELINE metric: apope_low; vctk_low is measured as an extra leg either way)")
    ap.add_argument("--math", choices=["bf16x3", "f32", "f16x2"], default=None,
                    help="matrix-core path of the dense convs (default: the engine's, MI355VITS_MATH or bf16x3 = f32 operands "
                         "split exactly into 3 bf16 terms, six MFMA products, f32 accumulate; f32 = v_mfma_f32 only)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MI355VITS_BENCH_STREAMS", "3")),
                    help="engine handles (HIP streams) kept in flight per GPU; steps are dealt to them in turn")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from this one process (threads), no torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 latency leg (profiling runs: keeps per-kernel "
                    "averages pure batch-32)")
    ap.add_argument("--no-extra", action="store_true", help="skip the device-only and vctk_low legs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes that measure roofline.traffic")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall budget of the CPU-baseline sample")
    ap.add_argument("--no-preheat", action="store_true",
                    help="skip the untimed pre-heat steps in front of the warm-up (rounds 1-3 and the driver's BENCH_r01-r03 were "
                         "measured without them; the JSON says which protocol ran: device.preheat)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    plan = plan_launch(args.gpus, args.single_process, os.environ,
                       torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if plan[0] == "error":
        raise SystemExit(plan[1])
    if plan[0] == "reexec":  # plain `python bench.py --gpus N`: become the N-rank job the contract describes
        cmd = plan[1](sys.executable, os.path.abspath(__file__), sys.argv[1:])
        print("bench.py: WORLD_SIZE unset, --gpus %d -> %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(cmd[0], cmd)
    single = plan[0] == "single"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL; used for the barrier and the max-over-ranks only

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from mimic3_amd import weights as W
    from mimic3_amd.config import VitsConfig

    voices = {"apope_low": VitsConfig.apope_low, "vctk_low": VitsConfig.vctk_low}
    cfg = voices[args.voice]()
    weights = W.synthetic_weights(cfg, seed=1234)
    devices = list(range(args.gpus)) if single else [local_rank]
    n_gpus = args.gpus if single else world
    # BASELINE.json: "... batch=1 and batch=256"; configs[3]: "batch=256 sharded across 8 x MI355X".  One step = one pass over the
    # job's 256 utterances: all of them on the one GPU at --gpus 1, 256 / N per GPU at --gpus N (strong scaling of the fixed job)
    strong = args.batch is None
    if strong and 256 % n_gpus:
        raise SystemExit(f"--gpus {n_gpus} does not divide BASELINE's batch of 256: pass --batch (utterances per GPU)")
    B, Tx, fpi = (256 // n_gpus if strong else args.batch), args.tx, args.frames_per_id
    # handles in flight per GPU: a 256-row batch fills the chip for 65 ms and takes 27 GB of workspace per handle (288 GB of HBM) — three
    # overlap one batch's launch-bound text side and its tail with the other batches' decoders: 64.3 - 64.4 ms per step against 64.7 - 66.0
    # with two and 67.1 with one (round 6, one lease, alternating: profiles/r06_b256_handles.txt)
    n_streams = max(1, args.streams) if B < 128 else min(int(os.environ.get("MI355VITS_BENCH_BIG_STREAMS", "3")), max(1, args.streams))
    # single-process: every step of every device is one batch of B utterances; the job's step = n_gpus batches
    wl = Workload(cfg, weights, devices, n_streams, B, Tx, fpi, rank, world, multispeaker_sid=cfg.is_multispeaker,
                  math=args.math)
    eng = wl.engines[0]
    math = wl.math

    def barrier():
        if dist is not None:
            dist.barrier()
        if single:
            for d in devices:
                torch.cuda.synchronize(d)
        else:
            torch.cuda.synchronize()

    def timed(w, steps, warmup, device_only=False):
        w.run_steps(warmup, device_only)
        barrier()
        t0 = time.perf_counter()
        out = w.run_steps(steps, device_only)
        barrier()
        el = time.perf_counter() - t0
        per_rank[:] = [el]
        if dist is not None:  # max over the ranks is the job's time; every rank's own time goes into the line
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            g = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(g, t)
            per_rank[:] = [float(x.item()) for x in g]
            el = max(per_rank)
        return el, out

    per_rank = []

    wl.size_workspaces()
    mon = DeviceMonitor(local_rank)
    dev_before = mon.snapshot()
    # in single-process mode K "steps" = K batches per device = K * n_gpus engine calls
    calls = args.steps * (n_gpus if single else 1)
    # pre-heat: a device that idled through model load starts at its idle clock state, and the first second under load is a
    # transient — the power controller overshoots, clamps the shader clock down and climbs back (1850 -> 1950 MHz at 1.2 -> 1.3 kW),
    # and the three host threads / pinned-buffer pool / hardware queues settle: steps run 4 - 10 % slower in it (profiles/
    # r05_preheat_transient.txt: 9.1 - 9.5 ms per step when timed 0.3 s after the first launch, 8.55 - 8.7 after 1.5 s, same box,
    # same binary).  W = 5 warm-up steps are 50 ms.  So: the same steps, untimed, for AT LEAST 1 s and until the shader clock of the
    # last 0.3 s stays within 2 % (3 s at most; 1 s when the clock cannot be read) — recorded in "device.preheat", never part of
    # `value`.  (Round 4's rule compared the recent clock with the highest one seen, which an idle-clock first sample made
    # unreachable and a loaded first sample satisfied at once: the driver's 20-step run was timed inside the transient.)
    preheat = {"seconds": 0.0, "steps": 0, "enabled": not args.no_preheat}
    t_ph = time.perf_counter()
    hist = []
    box_probe = None
    if rank == 0:
        try:  # ~10 ms: what this lease's chip gives the kernels' access patterns (include/mi355vits_lab.h mi355vits_probe_device;
            # the hooks live in libmi355vits_hooks.so = the product's objects + csrc/lab_api.cpp, never in the timed path)
            from mimic3_amd._native import Engine as _Engine, hooks_library
            hl = hooks_library()
            box_probe = hl.probe_device(local_rank)
            pe = _Engine(W.pack(cfg, weights), device=local_rank, library=hl)  # a replica of the same weights: its arena is probed
            try:
                box_probe.update(pe.probe_weights())
            finally:
                pe.close()
        except Exception as ex:  # noqa: BLE001 - diagnostics only
            box_probe = {"error": str(ex)[:100]}
    t_ph = time.perf_counter()
    while not args.no_preheat:
        wl.run_steps(n_streams * (n_gpus if single else 1))
        preheat["steps"] += n_streams
        el_ph = time.perf_counter() - t_ph
        clk = mon.snapshot().get("sclk_mhz") if mon.dir else None
        hist.append((el_ph, clk))
        if el_ph < 1.0:
            continue
        if dist is not None:  # several ranks: a fixed 1.5 s each, so that nobody idles (and cools) at the barrier below
            if el_ph >= 1.5:
                break
            continue
        if clk is None or el_ph >= 3.0:
            break
        win = max(0.3, 3.5 * el_ph / len(hist))  # (a batch-256 step is 68 ms: three samples need more than 0.3 s)
        recent = [c for t, c in hist if t >= el_ph - win and c]
        if len(recent) >= 3 and max(recent) - min(recent) <= 0.02 * float(np.median(recent)):
            break
    preheat["seconds"] = time.perf_counter() - t_ph
    preheat["sclk_mhz_first_last"] = [hist[0][1], hist[-1][1]] if hist else None
    barrier()
    mon.start()  # (sampling starts with the warm-up steps; it reads two sysfs files every 10 ms)
    elapsed, out = timed(wl, calls, args.warmup * (n_gpus if single else 1))
    headline_per_rank_ms = [e / args.steps * 1e3 for e in per_rank]
    dev_window = mon.stop()
    samples_per_call = int(out["lengths"].sum())
    samples_per_step = samples_per_call * n_gpus
    value = samples_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = {
        "metric": f"22.05 kHz audio samples/sec/node (en_{'UK/apope_low' if args.voice == 'apope_low' else 'US/vctk_low'} "
                  "VITS hot path: run + int16, host ids in -> host int16 out)",
        "value": value,
        "unit": "samples/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "per_rank_ms_per_step": headline_per_rank_ms,  # every rank's own clock over the same K steps (ms_per_step = their max)
        "world_size": (dist.get_world_size() if dist is not None else 1),  # as RCCL sees it (1: no process group)
        "launch": ("single-process, one host thread per device" if single else
                   ("torch.distributed.run, one rank per GPU (RCCL barrier + all_gather of the ranks' times)" if dist is not None else "one process")),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "headline_protocol": "r05+: step = BASELINE's batch 256 over the GPUs (strong); rounds 1-4: 32 utt/GPU (weak) -> compare b32_ms / b256_ms",
        "vs_baseline": None,
        "dtype": "f32" if math == "f32" else (
            "f32 in / out / accumulate; dense-conv operands as 2 x fp16 terms (22 significant bits, 3 f16-MFMA products per multiply-add), "
            "encoder on the f32 MFMA; experimental mode, parity at the f32 tolerances" if math == "f16x2" else
            "f32 (dense convs: operands split exactly into 3 x bf16, 6 bf16-MFMA products per "
            "multiply-add, f32 accumulate; f32 in / f32 out, parity at the f32 tolerances)"),
        "math": math,
        "data": f"synthetic (seeded random-init weights of the {args.voice} shapes, seeded phoneme ids)",
        "config": {
            "workload": (f"{'en_UK/apope_low' if args.voice == 'apope_low' else 'en_US/vctk_low'} batch {B * n_gpus}"
                         + (" (BASELINE's batch=256)" if B * n_gpus == 256 else "") + f" on {n_gpus} GPU(s): {B} utt/GPU x {Tx} ids x {fpi} "
                         f"frames/id = {Tx * fpi * cfg.hop_length} samples each")[:127],
            "global_batch": B * n_gpus, "per_gpu_batch": B, "phonemes": Tx, "frames": Tx * fpi,
            "parallelism": f"batch-shard x{n_gpus}" + (" (one process, one host thread per engine handle)" if single else ""),
            "streams_per_gpu": n_streams,
            "scales": [0.667, 1.0, 0.8],
            "timed_region": "host-to-host: ids H2D + run + int16 + D2H of int16 into recycled pinned buffers",
        },
        "rtf": elapsed / args.steps / (samples_per_step / SAMPLE_RATE),
        "x_realtime": (samples_per_step / SAMPLE_RATE) / (elapsed / args.steps),
        "timed_region_s": elapsed,
        "device": {"before": dev_before, "preheat": preheat, "timed_window": dev_window},
    }
    result["config"]["device"] = DeviceMonitor.brief(dev_before, dev_window)
    if B * n_gpus == 256:
        result["b256_ms"] = ms_per_step  # one pass over BASELINE's batch of 256 (this run's step)
    if box_probe:
        result["box_probe"] = box_probe
        for k in ("l2_stream_GBps", "l2_hit_latency_ns", "hbm_copy_GBps", "l2_stream_beside_copy_GBps", "table_24MB_stream_GBps", "latency_32MB_ns", "latency_1GiB_ns"):
            if k in box_probe:
                result["probe_" + k] = box_probe[k]

    if rank == 0:
        print(f"headline: {value:.4g} samples/s, {ms_per_step:.3f} ms/step over {elapsed:.2f} s (host-to-host)", file=sys.stderr)
        # where a step's wall time goes on ONE handle driven sequentially: the call as the caller sees it (Python marshalling +
        # H2D of the ids + launches + the sync on the frame counts + D2H of the int16 into a pooled pinned buffer) against the
        # device time between the call's first and last HIP event; with `streams` handles in flight the headline hides most
        # of the difference behind another handle's kernels
        walls, devs = [], []
        for i in range(12):
            t1 = time.perf_counter()
            wl.step(i, eng)
            walls.append(time.perf_counter() - t1)
            devs.append(eng.last_run_ms())
        w_ms, d_ms = float(np.median(walls[2:])) * 1e3, float(np.median(devs[2:]))
        result["host_ms_per_step"] = {"one_handle_call_ms": w_ms, "device_ms": d_ms, "host_side_ms": max(0.0, w_ms - d_ms),
                                      "handles_in_flight": n_streams,
                                      "note": "median of 10 sequential calls on one handle; host_side = call wall - device time "
                                              "between the call's first and last event"}

    if not args.no_extra:
        # same loop, int16 result left in HBM: the compute-side number (round 1's headline definition)
        n_dev = max(10, calls // 4)
        el_d, _ = timed(wl, n_dev, 3, device_only=True)
        if rank == 0:
            steps_d = n_dev / (n_gpus if single else 1)
            result["device_only"] = {
                "value": samples_per_step * steps_d / el_d, "unit": "samples/s", "ms_per_step": el_d / steps_d * 1e3,
                "steps": steps_d, "note": "int16 result left in HBM (no D2H inside the timed region)",
            }

    if rank == 0 and not args.no_b1:
        # ---- configs[1]: batch 1, golden-utterance shape (991 frames = 253,696 samples = 11.505 s)
        Txg = 180
        ids1 = np.random.default_rng(99).integers(1, 50, (1, Txg)).astype(np.int64)
        f1 = np.full((1, Txg), 5, np.int32)
        f1[0, :91] = 6
        sid1 = np.zeros(1, np.int64) if cfg.is_multispeaker else None

        def b1():
            return eng.run(ids1, [Txg], wl.scales, sid1, forced_durations=f1, want_float=False, want_pcm16=True)  # incl. D2H

        for _ in range(3):
            b1()
        lat = []
        for _ in range(20):
            t1 = time.perf_counter()
            o1 = b1()
            lat.append(time.perf_counter() - t1)
        n1 = int(o1["lengths"][0])
        med = float(np.median(lat))
        dev_ms = eng.last_run_ms()
        if not args.no_roofline:
            rep1, rows1, tot1 = kernel_table(eng, lambda i: b1(), 3)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_b1.json"), "w") as f:
                json.dump(rows1, f, indent=1)
            print("batch-1 golden shape, per-kernel (HIP events): total %.3f ms over %d launches" %
                  (tot1 / 3, sum(r["launches_per_step"] for r in rows1)), file=sys.stderr)
            for r in rows1[:10]:
                print(f"  {r['kernel']:22s} {r['launches_per_step']:4d} launches {r['ms_per_step']:8.3f} ms", file=sys.stderr)
        result["latency_b1"] = {
            "workload": f"{args.voice} batch 1, 180 ids -> 991 frames = 253,696 samples (golden-utterance shape), "
                        "host ids in -> host int16 out (PCIe included)",
            "ms_median": med * 1e3, "ms_min": float(min(lat)) * 1e3, "rtf": med / (n1 / SAMPLE_RATE),
            "x_realtime": (n1 / SAMPLE_RATE) / med, "device_ms": dev_ms,
        }
        if not args.no_roofline:
            result["latency_b1"]["launches"] = sum(r["launches_per_step"] for r in rows1)
            result["latency_b1"]["kernel_ms"] = tot1 / 3
        # the metric's batch-1 half as first-level numbers (and, as before, as a string where the driver's record keeps `config`)
        result["latency_b1_ms"] = med * 1e3
        result["b1_x_realtime"] = (n1 / SAMPLE_RATE) / med
        if "launches" in result["latency_b1"]:
            result["b1_launches"] = result["latency_b1"]["launches"]
            result["b1_kernel_ms"] = result["latency_b1"]["kernel_ms"]
        result["config"]["batch1"] = ("%.3f ms host-to-host = %.0f x real time (180 ids -> 991 frames = 11.5 s), %s launches, kernels %s ms" % (
            med * 1e3, (n1 / SAMPLE_RATE) / med, result["latency_b1"].get("launches", "?"),
            ("%.3f" % result["latency_b1"]["kernel_ms"]) if "kernel_ms" in result["latency_b1"] else "?"))[:127]

    if rank == 0 and not args.no_roofline:
        nsteps = 3
        for i in range(3):  # the batch-1 leg above leaves the chip mostly idle: back to the loaded clocks before the table
            wl.step(i, device_only=True)
        mon.start()
        rep, table, tot_ms = kernel_table(eng, lambda i: wl.step(i, device_only=True), nsteps)
        table_window = mon.stop()
        traffic = None
        if n_gpus == 1 and not args.no_traffic:
            tail = ["--batch", str(B), "--tx", str(Tx), "--frames-per-id", str(fpi), "--voice", args.voice] + (["--math", math] if args.math else [])
            traffic = measure_traffic_in_run(table[0]["kernel"], tail)
        result["roofline"] = roofline_of(rep, table, tot_ms, nsteps, [B, Tx, fpi], args.voice, math, traffic)
        # the whole per-kernel table, flat (the driver's record keeps first-level scalars of `roofline`): ms per step by label
        for row in table[:16]:
            result["roofline"]["ms:" + row["kernel"]] = round(row["ms_per_step"], 4)
        if table_window:
            result["roofline"]["sclk_mhz_during_table"] = table_window["sclk_mhz"]["median"]
            result["roofline"]["power_w_during_table"] = table_window["power_w"]["mean"]
        result["device"]["kernel_table_window"] = table_window
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
            json.dump(table, f, indent=1)
        print_table("per-kernel (HIP events):", table)

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, weights, wl, args.cpu_seconds)

    # the extra legs open their own handles: the headline's go first.  HIP maps a process's streams onto a few hardware
    # queues; with the headline's idle handles still open two ACTIVE handles of a leg can land on one queue and serialise —
    # measured (tools/floor_diag2.py): 7.87 ms per step alone, 9.45 ms with two idle handles open, 7.88 ms after closing them
    wl.close()
    if rank == 0 and n_gpus == 1 and not args.no_extra and args.voice == "apope_low":
        # ---- BASELINE.json configs[2]: en_US/vctk_low multi-speaker, batch 32 x 128 phonemes, one MI355X
        vcfg = VitsConfig.vctk_low()
        vw = Workload(vcfg, W.synthetic_weights(vcfg, seed=1234), devices, args.streams, 32, 128, fpi, 0, 1,
                      multispeaker_sid=True, math=args.math)
        vw.size_workspaces()
        vsteps = max(60, args.steps // 3)  # >= 60 steps after 10 warm-up: a 20-step leg measures clock ramp and first-touch costs
        el_v, out_v = timed(vw, vsteps, 10)
        sps = int(out_v["lengths"].sum())
        extra = {
            "workload": f"en_US/vctk_low (109 speakers, gin 512), 32 utterances x 128 phoneme ids, sid = b mod 109, forced "
                        f"{fpi} frames/id; host-to-host",
            "value": sps * vsteps / el_v, "unit": "samples/s", "steps": vsteps, "ms_per_step": el_v / vsteps * 1e3,
            "x_realtime": (sps / SAMPLE_RATE) / (el_v / vsteps), "dtype": "f32", "math": math,
        }
        if not args.no_roofline:
            repv, tablev, totv = kernel_table(vw.engines[0], lambda i: vw.step(i, device_only=True), 3)
            extra["roofline"] = roofline_of(repv, tablev, totv, 3, [32, 128, fpi], "vctk_low", math)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_vctk.json"), "w") as f:
                json.dump(tablev, f, indent=1)
            print_table("vctk_low b32, per-kernel (HIP events):", tablev[:12])
        result["extra"] = {"vctk_low_b32": extra}
        vw.close()
        # ---- the other end of configs[3]'s curve on this one device: the 32-row shard a GPU gets when the batch of 256 is spread over
        # eight (headline at --gpus 1: all 256 rows here), or — when --batch made the shard the headline — the whole 256 on one GPU
        ob = 32 if B != 32 else 256
        ow = Workload(cfg, weights, devices, max(1, args.streams) if ob < 128 else min(2, max(1, args.streams)), ob, Tx, fpi, 0, 1,
                      multispeaker_sid=cfg.is_multispeaker, math=args.math)
        ow.size_workspaces()
        osteps = max(60, args.steps) if ob == 32 else max(8, args.steps // 16)
        el_g, out_g = timed(ow, osteps, 10 if ob == 32 else 2)
        spb = int(out_g["lengths"].sum())
        oleg = {"workload": f"en_UK/apope_low, {ob} utterances x {Tx} phoneme ids on ONE GPU, forced {fpi} frames/id; host-to-host",
                "value": spb * osteps / el_g, "unit": "samples/s", "steps": osteps, "ms_per_step": el_g / osteps * 1e3,
                "x_realtime": (spb / SAMPLE_RATE) / (el_g / osteps), "math": math, "global_batch": ob}
        if not args.no_roofline:
            repg, tableg, totg = kernel_table(ow.engines[0], lambda i: ow.step(i, device_only=True), 3 if ob == 32 else 2)
            oleg["roofline"] = roofline_of(repg, tableg, totg, 3 if ob == 32 else 2, [ob, Tx, fpi], args.voice, math)
            oleg["kernels_ms_per_step"] = {r["kernel"]: round(r["ms_per_step"], 4) for r in tableg[:16]}
        result["extra"]["apope_low_b%d_1gpu" % ob] = oleg
        result["b%d_ms" % ob] = oleg["ms_per_step"]
        result["b%d_samples_per_s" % ob] = oleg["value"]
        result["config"]["batch%d_1gpu" % ob] = ("%.4g samples/s = %.2f ms per %d utterances on ONE GPU (%.0f x real time), %d steps" % (
            oleg["value"], oleg["ms_per_step"], ob, oleg["x_realtime"], osteps))[:127]
        ow.close()
        # ---- BASELINE.json configs[4]: long-form text streamed sentence by sentence, default math and bf16 weights
        lf = {"default": longform_stream(cfg, weights, args.math), "bf16w": longform_stream(cfg, weights, "bf16w")}
        lf["lengths_equal_across_modes"] = lf["default"].pop("det_lengths") == lf["bf16w"].pop("det_lengths")
        result["extra"]["longform_stream"] = lf
        result["longform_first_audio_ms"] = lf["default"]["first_audio_ms"]
        result["longform_total_ms"] = lf["default"]["total_ms"]
        result["longform_bf16w_first_audio_ms"] = lf["bf16w"]["first_audio_ms"]
        result["longform_bf16w_total_ms"] = lf["bf16w"]["total_ms"]
        result["longform_padding_efficiency"] = lf["default"]["padding_efficiency"]  # valid / computed output samples of the planned batches
        # ---- the reference's own call shape: one ragged sentence per call from 64 concurrent workers (and one call at a time)
        sv = serving_shape(cfg, weights, args.math)
        result["extra"]["serving_shape"] = sv
        result["serve64_sentences_per_s"] = sv["batch+lanes"]["sentences_per_s"]
        result["serve64_x_realtime"] = sv["batch+lanes"]["x_realtime"]
        result["serve64_latency_ms_p50"] = sv["batch+lanes"]["latency_ms_p50"]
        result["ragged_b1_calls_x_realtime"] = sv["one_call_at_a_time"]["x_realtime"]
        result["config"]["serving_shape"] = (
            "64 clients x 1 ragged sentence per call: %.0f sentences/s = %.0f x RT, p50 %.1f ms, mean batch %.1f; one call at a time %.0f x RT" % (
                sv["batch+lanes"]["sentences_per_s"], sv["batch+lanes"]["x_realtime"], sv["batch+lanes"]["latency_ms_p50"],
                sv["batch+lanes"].get("mean_batch", 0.0), sv["one_call_at_a_time"]["x_realtime"]))[:127]
        result["config"]["longform_stream"] = (
            "120 sentences %.0f s audio: first audio %.1f ms, all in %.0f ms = %.0f x RT; bf16w %.1f / %.0f ms; chunks==calls %s, lengths== %s" % (
                lf["default"]["audio_s"], lf["default"]["first_audio_ms"], lf["default"]["total_ms"], lf["default"]["x_realtime"],
                lf["bf16w"]["first_audio_ms"], lf["bf16w"]["total_ms"],
                lf["default"]["chunks_equal_single_calls"] and lf["bf16w"]["chunks_equal_single_calls"], lf["lengths_equal_across_modes"]))[:127]
        if math != "f32":
            # ---- the same headline workload on the pure f32-MFMA path (v_mfma_f32_32x32x2_f32 everywhere), for reference
            fw = Workload(cfg, weights, devices, args.streams, B, Tx, fpi, rank, world, multispeaker_sid=cfg.is_multispeaker,
                          math="f32")
            fw.size_workspaces()
            fsteps = max(60, args.steps // 4)
            el_f, out_f = timed(fw, fsteps, 10)
            f32_leg = {"math": "f32", "value": int(out_f["lengths"].sum()) * fsteps / el_f, "unit": "samples/s", "steps": fsteps,
                       "ms_per_step": el_f / fsteps * 1e3,
                       "note": "same workload, every dense conv on v_mfma_f32_32x32x2_f32 (MI355VITS_MATH=f32)"}
            if not args.no_roofline:
                repf, tablef, totf = kernel_table(fw.engines[0], lambda i: fw.step(i, device_only=True), 3)
                f32_leg["roofline"] = roofline_of(repf, tablef, totf, 3, [B, Tx, fpi], args.voice, "f32")
                with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_f32.json"), "w") as f:
                    json.dump(tablef, f, indent=1)
            result["extra"]["f32_mfma"] = f32_leg
            fw.close()
            # ---- BASELINE.json configs[4], first slice: bf16 weights (reduced precision: its own tolerance, rel RMS <= 2e-2,
            # tests/test_gpu_parity.py::test_bf16_weights_mode_at_its_own_tolerance) — never the headline
            bw = Workload(cfg, weights, devices, args.streams, B, Tx, fpi, rank, world, multispeaker_sid=cfg.is_multispeaker,
                          math="bf16w")
            bw.size_workspaces()
            el_b, out_b = timed(bw, fsteps, 10)
            bw_leg = {"math": "bf16w", "dtype": "bf16-weights (bf16 x exact-f32 activations, f32 accumulate)",
                      "value": int(out_b["lengths"].sum()) * fsteps / el_b, "unit": "samples/s", "steps": fsteps,
                      "ms_per_step": el_b / fsteps * 1e3,
                      "note": "same workload, weights rounded to bf16 (3 bf16-MFMA products per multiply-add); reduced precision"}
            if not args.no_roofline:
                repb, tableb, totb = kernel_table(bw.engines[0], lambda i: bw.step(i, device_only=True), 3)
                rl = roofline_of(repb, tableb, totb, 3, [B, Tx, fpi], args.voice, "bf16x3")
                rl["peak"] = PEAK_BF16_TFLOPS / 3.0
                rl["frac"] = rl["achieved"] / rl["peak"]
                rl["matrix_core_path"] = "v_mfma_f32_32x32x16_bf16 x 3 partial products per multiply-add: peak = 2500 / 3 TFLOP/s"
                bw_leg["roofline"] = rl
            result["extra"]["bf16_weights"] = bw_leg
            bw.close()
            # ---- experimental: the fused MRF stages with operands as two fp16 terms (22 significant bits, three products);
            # same tolerances as the default in the tests, but not f32-grade by construction — reported beside, never the headline
            hw = Workload(cfg, weights, devices, args.streams, B, Tx, fpi, rank, world, multispeaker_sid=cfg.is_multispeaker,
                          math="f16x2")
            hw.size_workspaces()
            el_h, out_h = timed(hw, fsteps, 10)
            h2_leg = {"math": "f16x2", "dtype": "f32 in / out / accumulate; dense-conv operands as 2 x fp16 (22 bits), 3 MFMA products",
                      "value": int(out_h["lengths"].sum()) * fsteps / el_h, "unit": "samples/s", "steps": fsteps,
                      "ms_per_step": el_h / fsteps * 1e3,
                      "note": "same workload; every kernel of the bf16x3 split runs the two-term fp16 split instead"}
            if not args.no_roofline:
                reph, tableh, toth = kernel_table(hw.engines[0], lambda i: hw.step(i, device_only=True), 3)
                h2_leg["kernels_ms_per_step"] = {r["kernel"]: round(r["ms_per_step"], 4) for r in tableh[:8]}
            result["extra"]["f16x2"] = h2_leg
            hw.close()


    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg, weights, wl, budget_s):
    """The PyTorch-CPU oracle on this host's cores over a bounded sample of the same workload: (a) one utterance per
    call (the reference's own call shape: B = 1 per sentence), (b) four utterances per call, both on <= 32 threads;
    `value` is the better of the two."""
    import torch

    from oracle.vits_oracle import VitsOracle, audio_float_to_int16

    ora = VitsOracle(cfg, weights)
    Tx, fpi = wl.Tx, wl.fpi
    rng = np.random.default_rng(0)
    ncpu = os.cpu_count() or 1

    def leg(nb, threads, budget):
        torch.set_num_threads(threads)
        ids_c, len_c = wl.ids[:nb], wl.lengths[:nb]
        sid_c = None if wl.sid is None else wl.sid[:nb]
        nw = rng.standard_normal((nb, 2, Tx)).astype(np.float32)
        nz = rng.standard_normal((nb, cfg.inter_channels, Tx * fpi)).astype(np.float32)

        last = {}

        def once():
            r = ora.infer(ids_c, len_c, wl.scales, sid=sid_c, noise_w=nw, noise_z=nz, forced_durations=wl.forced[:nb],
                          stage_rows=())
            last["r"] = r
            return [audio_float_to_int16(r["audio"][b, 0, : int(r["audio_lengths"][b])]) for b in range(nb)]

        t_begin = time.perf_counter()
        pcm = once()  # warm-up (thread pools, allocator); also the only sample if a call is too slow for the budget
        warm = time.perf_counter() - t_begin
        times = []
        while warm < 0.4 * budget and (len(times) < 3 or (time.perf_counter() - t_begin < budget and len(times) < 500)):
            t1 = time.perf_counter()
            pcm = once()
            times.append(time.perf_counter() - t1)
        n = sum(len(p) for p in pcm)
        med = float(np.median(times)) if times else warm
        # the checker doing its job in the same run: the engine (the handles just timed, same math mode) on the very
        # inputs the oracle was timed on — both Gaussian draws injected — against the oracle's waveform
        out = wl.engines[0].run(ids_c, len_c, wl.scales, sid_c, noise_w=nw, noise_z=nz, forced_durations=wl.forced[:nb],
                                want_float=True, want_pcm16=True)
        r = last["r"]
        rel, same_len, worst_lsb = 0.0, True, 0
        for b in range(nb):
            L = int(out["lengths"][b])
            same_len = same_len and L == int(r["audio_lengths"][b])
            a64, r64 = out["audio"][b, :L].astype(np.float64), r["audio"][b, 0, :L].astype(np.float64)
            rel = max(rel, float(np.sqrt(np.mean((a64 - r64) ** 2)) / max(1e-30, np.sqrt(np.mean(r64 ** 2)))))
            worst_lsb = max(worst_lsb, int(np.abs(out["pcm"][b, :L].astype(np.int32) - pcm[b].astype(np.int32)).max()))
        return {"utterances_per_call": nb, "threads": threads, "samples": n, "runs": len(times), "ms_median": med * 1e3,
                "samples_per_s": n / med,
                "engine_vs_oracle": {"rel_rms_worst_row": rel, "tolerance": 1e-4, "lengths_equal": same_len,
                                     "int16_max_lsb": worst_lsb, "math": wl.math}}

    # thread sweep inside the same wall budget (32 of the box's 256 hardware threads understates "the same box's
    # host cores"): one utterance per call on 32 threads (the reference's own call shape, B = 1 per sentence), then four
    # utterances per call on 32 / 64 / 128 threads; the best leg is the reported baseline, with ITS thread count in `cores`
    sweep = [(1, min(ncpu, 32))] + [(min(4, wl.B), t) for t in (32, 64, 128) if t <= ncpu]
    legs = [leg(nb, th, budget_s / len(sweep)) for nb, th in sweep]
    best = max(legs, key=lambda l: l["samples_per_s"])
    return {
        "value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
        "sample": f"PyTorch-CPU fp32 oracle (onnxruntime unavailable), {best['utterances_per_call']} utterance(s) x {Tx} ids x "
                  f"{fpi} frames/id = {best['samples']} samples per call, median of {best['runs']} calls after 1 warm-up, "
                  "run + int16",
        "ms_median": best["ms_median"], "x_realtime": best["samples_per_s"] / SAMPLE_RATE,
        "thread_sweep": "; ".join(f"{l['utterances_per_call']}x{l['threads']}thr: {l['samples_per_s']:.3g}/s" for l in legs)[:127],
        "engine_vs_oracle_rel_rms": best["engine_vs_oracle"]["rel_rms_worst_row"],
        "engine_vs_oracle_lengths_equal": best["engine_vs_oracle"]["lengths_equal"],
        "engine_vs_oracle_int16_max_lsb": best["engine_vs_oracle"]["int16_max_lsb"],
        "engine_vs_oracle": best["engine_vs_oracle"],
        "legs": legs, "host_cpus": ncpu, "cpu": _cpu_model(), "torch": torch.__version__,
    }


def longform_stream(cfg, weights, math, n_sentences=120, look_ahead=32):
    """BASELINE.json configs[4] (its GPU half): a long-form request — 120 sentences of 40-160 phoneme ids, about 10k characters,
    NATURAL durations, stochastic scales — delivered as an ordered chunk stream by ``mimic3_amd.streaming.stream_sentences``
    (what ``/api/tts/stream`` runs per request, mimic3_amd/http_stream.py; the reference joins the sentences of
    ``end_utterance`` into one WAV, mimic3_http/app.py:157-227, tts.py:470-515) on ONE shared session: 3 lanes per device, 1 ms
    micro-batch window, every visible device (``devices="all"``).  first_audio_ms = request start -> first chunk in the caller's
    hands; total_ms = last chunk.  In-leg check at deterministic scales: the streamed chunks are BITWISE the per-sentence calls.
    padding_efficiency = valid / computed output samples over the planned batches (a batch computes every row to its longest)."""
    from mimic3_amd import streaming as ST
    from mimic3_amd import weights as W
    from mimic3_amd.session import InferenceSession, SessionOptions

    so = SessionOptions()
    so.lanes = 3
    so.micro_batch_window_ms = 1.0
    so.devices = "all"
    so.math = math
    so.seed = 4242
    sess = InferenceSession(W.pack(cfg, weights), sess_options=so)
    try:
        rng = np.random.default_rng(0)
        sentences = [rng.integers(1, 50, int(rng.integers(40, 161))).astype(np.int64).tolist() for _ in range(n_sentences)]
        list(ST.stream_sentences(sess, sentences[:look_ahead], look_ahead=look_ahead))  # warm-up: workspaces of every lane sized
        runs = []
        stats = {}
        for _ in range(3):
            t0 = time.perf_counter()
            first, n = None, 0
            # a list: stream_sentences PLANS the request (sentence 0 alone, a head batch, the rest length-sorted inside its
            # phoneme-length class: mimic3_amd.streaming.plan_batches; tts.py:470-515 holds all pending sentences the same way)
            for audio in ST.stream_sentences(sess, sentences, look_ahead=look_ahead, stats=stats):
                if first is None:
                    first = time.perf_counter() - t0
                n += audio.shape[0]
            runs.append((time.perf_counter() - t0, first, n))
        total, first, n = sorted(runs)[1]  # the median run
        # round 5's shape for comparison: one call per sentence from a thread pool, batches formed by arrival (a lazy iterable)
        t0 = time.perf_counter()
        first_lazy = None
        for audio in ST.stream_sentences(sess, iter(sentences), look_ahead=look_ahead):
            if first_lazy is None:
                first_lazy = time.perf_counter() - t0
        total_lazy = time.perf_counter() - t0
        # the check: deterministic scales (a call's Philox draws are keyed by its utterance number, which concurrent calls take
        # in arrival order), 16 sentences streamed with 16 in flight == the same sentences one call at a time, bit for bit
        det = (0.0, 1.0, 0.0)
        streamed = list(ST.stream_sentences(sess, sentences[:16], scales=det, look_ahead=16))
        single = [sess.run_pcm16(ST._feed(ids, det, None))[0][0] for ids in sentences[:16]]
        same = all(np.array_equal(a, b) for a, b in zip(streamed, single))
        return {"math": sess.engine.math, "sentences": n_sentences, "phoneme_ids": int(sum(len(x) for x in sentences)), "look_ahead": look_ahead,
                "lanes_per_device": 3, "micro_batch_window_ms": 1.0, "devices": list(sess.devices),
                "first_audio_ms": first * 1e3, "total_ms": total * 1e3, "audio_s": n / SAMPLE_RATE,
                "x_realtime": n / SAMPLE_RATE / total, "samples_per_s": n / total, "runs_total_ms": [r[0] * 1e3 for r in runs],
                "chunks_equal_single_calls": bool(same), "det_lengths": [int(a.shape[0]) for a in streamed],
                "planned_batches": stats.get("batches"), "padding_efficiency": stats.get("padding_efficiency"),
                "text_padding_efficiency": stats.get("text_padding_efficiency"),
                "arrival_batched": {"first_audio_ms": first_lazy * 1e3, "total_ms": total_lazy * 1e3,
                                    "note": "the same request as a lazy iterable: one call per sentence, micro-batched by arrival"}}
    finally:
        sess.close()


def serving_shape(cfg, weights, math, clients=64, seconds=1.5):
    """The reference's REAL call shape as a throughput number (VERDICT r5 missing #5): `clients` threads — mimic3_http's synthesis
    workers, synthesis.py:88-136 — each issuing ONE ragged sentence per call (`voice.py:180-181`: B = 1, 40-160 phoneme ids, NATURAL
    durations, stochastic scales) on one shared session with 3 lanes and a 1 ms micro-batch window; closed loop.  Reports sentences/s,
    audio seconds per second (x real time), latency percentiles, the mean batch the micro-batcher formed, and beside it the same
    sentences one call at a time on a plain session (what a drop-in without caller-side batching gives)."""
    from mimic3_amd import weights as W
    from mimic3_amd.session import InferenceSession, SessionOptions

    rng = np.random.default_rng(0)
    feeds = []
    for _ in range(256):
        n = int(rng.integers(40, 161))
        feeds.append({"input": rng.integers(1, 50, (1, n)).astype(np.int64), "input_lengths": np.array([n], np.int64),
                      "scales": np.array([0.667, 1.0, 0.8], np.float32)})
    blob = W.pack(cfg, weights)
    out = {}
    for name, lanes, window, nclients in (("batch+lanes", 3, 1.0, clients), ("one_call_at_a_time", 1, 0.0, 1)):
        so = SessionOptions()
        so.lanes, so.micro_batch_window_ms, so.micro_batch_max, so.math = lanes, window, 32, math
        sess = InferenceSession(blob, sess_options=so)
        try:
            for f in feeds[:4]:
                sess.run_pcm16(f)
            lat, samples, lock = [], [0], threading.Lock()
            stop = time.perf_counter() + (seconds if nclients > 1 else 0.5 * seconds)

            def client(k):
                i, mine, n = k, [], 0
                while time.perf_counter() < stop:
                    t1 = time.perf_counter()
                    _rows, lengths = sess.run_pcm16(feeds[i % len(feeds)])
                    mine.append(time.perf_counter() - t1)
                    n += int(lengths[0])
                    i += nclients
                with lock:
                    lat.extend(mine)
                    samples[0] += n

            t0 = time.perf_counter()
            ts = [threading.Thread(target=client, args=(k,)) for k in range(nclients)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            wall = time.perf_counter() - t0
            lm = np.sort(np.array(lat)) * 1e3
            out[name] = {"clients": nclients, "lanes": lanes, "micro_batch_window_ms": window, "sentences_per_s": len(lat) / wall,
                         "x_realtime": samples[0] / SAMPLE_RATE / wall, "samples_per_s": samples[0] / wall,
                         "latency_ms_p50": float(lm[len(lm) // 2]), "latency_ms_p95": float(lm[int(0.95 * len(lm))])}
            if sess._batcher is not None:
                out[name]["mean_batch"] = sess._batcher.requests / max(1, sess._batcher.batches)
        finally:
            sess.close()
    return out


# profiler label -> substring of the kernel name as rocprofv3 prints it (the label's launches are that kernel's)
_KERNEL_NEEDLES = {
    "dec.mrf_p.s2": "k_mrf_p<32", "dec.mrf_p.s1": "k_mrf_p<64", "dec.mrf_p": "k_mrf_p<",
    "dec.mrf_fused.s0": "k_mrf_fused<4", "dec.mrf_fused.s1": "k_mrf_fused<2", "dec.mrf_fused.s2": "k_mrf_fused<1",
    "flow.wn_layer_b3": "k_wn_layer_b3", "flow.wn_layer": "k_wn_layer_h192",
}


def measure_traffic_in_run(label, argv_tail, timeout_s=240):
    """HBM bytes per launch of the kernel behind profiler label `label`, MEASURED NOW: two counters-only rocprofv3 passes
    (FETCH_SIZE, WRITE_SIZE — they do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots") of a short run of this
    same script on this same box.  bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: on gfx950 FETCH_SIZE tallies 64 B per
    128-B read request (same guide, "HBM").  None when rocprofv3 is missing or a pass fails — never a stored constant."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3")
    needle = _KERNEL_NEEDLES.get(label)
    if not prof or not needle:
        return None
    means, launches, kname_seen = {}, 0, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mi355vits_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1",
                   "--streams", "1", "--no-cpu-baseline", "--no-extra", "--no-b1", "--no-roofline", "--no-traffic"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
            if not dbs:
                return None
            per, names = {}, {}
            con = sqlite3.connect(dbs[0])
            for disp, kname, val in con.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ?",
                                                (counter,)):
                if needle in kname:
                    per[disp] = per.get(disp, 0.0) + val
                    names[disp] = kname
            con.close()
            if not per:
                return None
            means[counter] = sum(per.values()) / len(per)
            launches = len(per)
            kname_seen = next(iter(names.values()))
        except (subprocess.SubprocessError, OSError, sqlite3.Error):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"hbm_bytes_per_launch": int((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024), "fetch_size_kib": means["FETCH_SIZE"],
            "write_size_kib": means["WRITE_SIZE"], "launches": launches, "kernel": kname_seen,
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two counters-only passes of a 3-step run of "
                      "this script on this box), 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 FETCH_SIZE tallies 64 B per 128-B request)"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
