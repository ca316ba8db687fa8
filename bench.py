#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: 22.05 kHz audio samples/s (whole job) + RTF for the
en_UK/apope_low VITS graph, hot path only (``onnx_model.run`` + ``audio_float_to_int16``, the reference's
own timed region, mimic3_tts/voice.py:229-232).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic phoneme ids.  Workload (weak scaling):
every GPU synthesises ``--batch`` (default 32) independent utterances of 128 phoneme ids, durations forced
to 6 frames/id -> 768 latent frames = 196,608 samples = 8.916 s per utterance (SURVEY.md §8d "unit S");
8 GPUs x 32 = the batch-256 configuration of BASELINE.json.  Inputs are 1 KiB of ids per utterance; they
and the int16 result stay in HBM inside the timed region (PCIe-inclusive numbers: DESIGN.md).
Batch-1 latency / RTF on the reference's golden-utterance shape (180 ids -> 991 frames) is measured in
the same run and reported under "latency_b1".

The JSON line also carries
  "roofline"     — dominant kernel group, algorithmic FLOPs / launch time measured with HIP events on the
                   engine's stream (mi355vits_profile_*), against the fp32 matrix-core peak, plus HBM numbers
  "cpu_baseline" — the PyTorch-CPU oracle (onnxruntime is not installed here) timed on this host's cores on
                   a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E spec
SAMPLE_RATE = 22050


def make_batch(B, Tx, base):
    ids = np.stack([np.random.default_rng(1234 + base + b).integers(1, 50, Tx) for b in range(B)]).astype(np.int64)
    return ids, np.full(B, Tx, np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--tx", type=int, default=128)
    ap.add_argument("--frames-per-id", type=int, default=6)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MI355VITS_BENCH_STREAMS", "3")),
                    help="engine handles (HIP streams) kept in flight per GPU; steps are dealt to them in turn")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 latency leg (profiling runs: keeps per-kernel "
                    "averages pure batch-32)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall budget of the CPU-baseline sample")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback for the measured path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL; used for the barrier and the max-over-ranks only

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from mimic3_amd import weights as W
    from mimic3_amd._native import Engine
    from mimic3_amd.config import VitsConfig

    cfg = VitsConfig.apope_low()
    weights = W.synthetic_weights(cfg, seed=1234)
    blob = W.pack(cfg, weights)
    eng = Engine(blob, device=local_rank)
    engines = [eng] + [Engine(blob, device=local_rank) for _ in range(max(1, args.streams) - 1)]

    B, Tx, fpi = args.batch, args.tx, args.frames_per_id
    ids, lengths = make_batch(B, Tx, rank * B)
    forced = np.full((B, Tx), fpi, np.int32)
    scales = np.array([0.667, 1.0, 0.8], np.float32)

    def step(i, e=None):
        return (e or eng).run(ids, lengths, scales, forced_durations=forced, seed=1,
                              utterance_base=rank * B + i * world * B, want_float=False, want_pcm16=True, device_only=True)

    def run_steps(n):
        """n steps; with --streams S > 1, S host threads each drive one engine handle (own HIP stream and workspace)
        and take step numbers from a shared counter, so the small-kernel front half of one batch overlaps the
        matrix-core back half of another.  Every step is a complete, independent run()."""
        if len(engines) == 1:
            last = None
            for i in range(n):
                last = step(i)
            return last
        import itertools
        import threading

        counter = itertools.count()
        outs = [None] * len(engines)
        errs = []

        def work(k):
            try:
                while True:
                    i = next(counter)
                    if i >= n:
                        return
                    outs[k] = step(i, engines[k])
            except Exception as ex:  # noqa: BLE001 - re-raised below
                errs.append(ex)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(engines))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise errs[0]
        return next((o for o in outs if o is not None), None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for e in engines:  # every handle sizes its workspace on first use: never inside the timed region
        step(0, e)
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    samples_per_step = int(out["lengths"].sum()) * world
    value = samples_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = {
        "metric": "22.05 kHz audio samples/sec/node (en_UK/apope_low VITS hot path: run + int16)",
        "value": value,
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded random-init weights of the en_UK/apope_low shapes, seeded phoneme ids)",
        "config": {
            "workload": f"en_UK/apope_low, {B} utterances/GPU x {Tx} phoneme ids, forced {fpi} frames/id "
                        f"({Tx * fpi} frames = {Tx * fpi * cfg.hop_length} samples each); 8 GPUs = BASELINE batch 256",
            "global_batch": B * world, "phonemes": Tx, "frames": Tx * fpi, "parallelism": f"batch-shard x{world}",
            "streams_per_gpu": len(engines),
            "scales": [0.667, 1.0, 0.8],
        },
        "rtf": elapsed / args.steps / (samples_per_step / SAMPLE_RATE),
        "x_realtime": (samples_per_step / SAMPLE_RATE) / (elapsed / args.steps),
    }

    if rank == 0 and not args.no_b1:
        # ---- configs[1]: batch 1, golden-utterance shape (991 frames = 253,696 samples = 11.505 s)
        Txg = 180
        ids1 = np.random.default_rng(99).integers(1, 50, (1, Txg)).astype(np.int64)
        f1 = np.full((1, Txg), 5, np.int32)
        f1[0, :91] = 6
        for _ in range(3):
            eng.run(ids1, [Txg], scales, forced_durations=f1, want_float=False, want_pcm16=True)
        lat = []
        for _ in range(10):
            t1 = time.perf_counter()
            o1 = eng.run(ids1, [Txg], scales, forced_durations=f1, want_float=False, want_pcm16=True)  # incl. D2H of int16
            lat.append(time.perf_counter() - t1)
        n1 = int(o1["lengths"][0])
        med = float(np.median(lat))
        if not args.no_roofline:
            eng.profile_enable(True)
            eng.profile_reset()
            for _ in range(3):
                eng.run(ids1, [Txg], scales, forced_durations=f1, want_float=False, want_pcm16=True)
            rep1 = eng.profile_report()
            eng.profile_enable(False)
            eng.profile_reset()
            rows1 = sorted(({"kernel": k, "launches": v["calls"] // 3, "ms": v["ms"] / 3} for k, v in rep1.items()),
                           key=lambda r: -r["ms"])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_b1.json"), "w") as f:
                json.dump(rows1, f, indent=1)
            print("batch-1 golden shape, per-kernel (HIP events): total %.3f ms over %d launches" %
                  (sum(r["ms"] for r in rows1), sum(r["launches"] for r in rows1)), file=sys.stderr)
            for r in rows1[:10]:
                print(f"  {r['kernel']:22s} {r['launches']:4d} launches {r['ms']:8.3f} ms", file=sys.stderr)
        result["latency_b1"] = {
            "workload": "en_UK/apope_low batch 1, 180 ids -> 991 frames = 253,696 samples (golden-utterance shape), "
                        "host ids in -> host int16 out (PCIe included)",
            "ms_median": med * 1e3, "ms_min": float(min(lat)) * 1e3, "rtf": med / (n1 / SAMPLE_RATE),
            "x_realtime": (n1 / SAMPLE_RATE) / med, "device_ms": eng.last_run_ms(),
        }

    if rank == 0 and not args.no_roofline:
        eng.profile_enable(True)
        eng.profile_reset()
        for i in range(max(1, min(args.steps, 3))):
            step(i)
        rep = eng.profile_report()
        eng.profile_enable(False)
        nsteps = max(1, min(args.steps, 3))
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
        dom = table[0]
        drec = rep[dom["kernel"]]
        dsec = drec["ms"] * 1e-3
        result["roofline"] = {
            "kernel": dom["kernel"],
            "bound": "mfma",
            "achieved": drec["flops"] / dsec / 1e12,
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": drec["flops"] / dsec / 1e12 / PEAK_FP32_TFLOPS,
            "traffic": (_pmc_traffic(dom["kernel"], B, Tx, fpi) or {}).get("hbm_bytes_per_launch"),
            "traffic_detail": _pmc_traffic(dom["kernel"], B, Tx, fpi),
            "avg_launch_us": drec["ms"] * 1e3 / drec["calls"],
            "launches": drec["calls"],
            "hbm_algorithmic_gbs": drec["bytes"] / dsec / 1e9,
            "hbm_frac": drec["bytes"] / dsec / 1e9 / PEAK_HBM_GBS,
            "note": "fp32 Conv1d is compute-bound on MI355X (AI >= 24 FLOP/B vs ridge 19.7): peak = 157.3 TFLOP/s fp32 "
                    "matrix-core rate; hbm_* give the same launches against the 8 TB/s HBM roof as BASELINE asks",
            "whole_step": {
                "tflops": sum(v["flops"] for v in rep.values()) / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0,
                "hbm_algorithmic_gbs": sum(v["bytes"] for v in rep.values()) / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0,
                "kernel_ms_per_step": tot_ms / nsteps,
            },
        }
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
            json.dump(table, f, indent=1)
        print("per-kernel (HIP events):", file=sys.stderr)
        for row in table:
            print(f"  {row['kernel']:22s} {row['launches_per_step']:4d} launches  {row['ms_per_step']:9.3f} ms/step "
                  f"{100 * row['share']:5.1f}%  {row['tflops']:7.2f} TFLOP/s  {row['gbs_algorithmic']:8.1f} GB/s", file=sys.stderr)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.vits_oracle import VitsOracle, audio_float_to_int16

        torch.set_num_threads(min(os.cpu_count() or 1, 32))  # beyond ~32 threads the small ops only get slower
        ora = VitsOracle(cfg, weights)
        nb = 1
        ids_c, len_c = ids[:nb], lengths[:nb]
        rng = np.random.default_rng(0)
        nw = rng.standard_normal((nb, 2, Tx)).astype(np.float32)
        nz = rng.standard_normal((nb, cfg.inter_channels, Tx * fpi)).astype(np.float32)

        def cpu_once():
            r = ora.infer(ids_c, len_c, scales, noise_w=nw, noise_z=nz, forced_durations=forced[:nb])
            return [audio_float_to_int16(r["audio"][b, 0, : int(r["audio_lengths"][b])]) for b in range(nb)]

        cpu_once()
        times = []
        t_begin = time.perf_counter()
        while len(times) < 3 or (time.perf_counter() - t_begin < args.cpu_seconds and len(times) < 500):
            t1 = time.perf_counter()
            pcm = cpu_once()
            times.append(time.perf_counter() - t1)
        n_cpu = sum(len(p) for p in pcm)
        medc = float(np.median(times))
        result["cpu_baseline"] = {
            "value": n_cpu / medc, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"PyTorch-CPU fp32 oracle (onnxruntime unavailable), {nb} utterance x {Tx} ids x {fpi} frames/id "
                      f"= {n_cpu} samples, median of {len(times)} runs after 1 warm-up, run + int16",
            "ms_median": medc * 1e3, "x_realtime": (n_cpu / SAMPLE_RATE) / medc,
            "cpu": _cpu_model(), "torch": torch.__version__,
        }

    if rank == 0:
        print(json.dumps(result))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(kernel, B, Tx, fpi):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), valid for
    the workload they were collected on; None otherwise (counters cannot be read from inside this process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(kernel)
        if e and e["workload"] == [B, Tx, fpi]:
            return {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
                    "source": e["source"]}
    except (OSError, ValueError, KeyError):
        pass
    return None


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
