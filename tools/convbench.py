#!/usr/bin/env python3
"""Micro-benchmark of the MFMA Conv1d kernel on the shapes of the hot path (run on the GPU box).
Each configuration runs in a fresh process because tile shape / chunk overrides are read once from the environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {
    # name: (B, Cin, Cout, T, K, dil, epi)
    "flow.in_gate": (32, 192, 384, 768, 5, 1, 1),
    "rb.s0.k7": (32, 128, 128, 6144, 7, 3, 0),
    "rb.s0.k3": (32, 128, 128, 6144, 3, 1, 0),
    "ups0.poly": (32, 256, 1024, 769, 2, 1, 0),
    "enc.ffn1": (32, 192, 768, 128, 3, 1, 0),
}

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from mimic3_amd._native import hooks_library

    name = sys.argv[2]
    B, Cin, Cout, T, K, dil, epi = SHAPES[name]
    ms = hooks_library().bench_conv1d(B, Cin, Cout, T, K, dil, epi, reps=10)
    fl = 2.0 * B * T * Cout * Cin * K
    print(json.dumps({"shape": name, "cfg": os.environ.get("MI355VITS_CONV_CFG", "auto"),
                      "chunk": os.environ.get("MI355VITS_CONV_CHUNK", "64"), "ms": ms, "tflops": fl / ms / 1e9}))
    sys.exit(0)

runs = []
for name in SHAPES:
    cfgs = ["auto", "2,2,2,2", "2,1,2,2"] if SHAPES[name][6] == 1 else ["auto", "2,2,2,2", "1,2,2,2", "1,1,2,2"]
    for cfg in cfgs:
        for chunk in ("64", "32"):
            env = dict(os.environ)
            if cfg != "auto":
                env["MI355VITS_CONV_CFG"] = cfg
            env["MI355VITS_CONV_CHUNK"] = chunk
            r = subprocess.run([sys.executable, __file__, "--one", name], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[-200:]
            print(line, flush=True)
