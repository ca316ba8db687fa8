#!/usr/bin/env python3
"""Turn a rocprofv3 output directory (rocpd sqlite) into the text summaries committed under profiles/.

    python tools/rocprof_summary.py stats  gpurun_out/prof_stats            # --kernel-trace --stats run
    python tools/rocprof_summary.py pmc    gpurun_out/prof_pmc1 [more dirs] # --pmc runs (one counter set per dir)

`pmc` prints, per kernel name and counter, the per-launch mean of the counter summed over its instances (XCDs /
channels); HBM bytes follow MI355X_MICROARCH.md: bytes = 2 x FETCH_SIZE(KB) x 1024 / 2 ... see DESIGN.md §6 for
the gfx950 correction actually applied (FETCH_SIZE counts 64 B per 128 B request -> reads = 2 x FETCH_SIZE KiB).
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def _db(d):
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit(f"no rocpd .db under {d}")
    return sqlite3.connect(dbs[0])


def stats(d, limit=40):
    c = _db(d)
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (limit,)):
        print(f"{name[:100]:100s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}")


def pmc(dirs, min_us=20.0):
    rows = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [per-dispatch sum]
    dur = defaultdict(list)
    for d in dirs:
        c = _db(d)
        per = defaultdict(float)
        meta = {}
        for disp, kname, cname, val, dd in c.execute(
                "select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
            per[(disp, cname)] += val
            meta[disp] = (kname, dd)
        seen = set()
        for (disp, cname), v in per.items():
            kname, dd = meta[disp]
            rows[kname][cname].append(v)
            if (d, disp) not in seen:
                seen.add((d, disp))
                dur[kname].append(dd / 1e3)
    print(f"{'kernel':80s} {'launches':>8s} {'avg_us':>10s}  counters (per-launch mean)")
    for kname in sorted(rows, key=lambda k: -sum(dur[k])):
        avg = sum(dur[kname]) / max(1, len(dur[kname]))
        if avg < min_us:
            continue
        cs = "  ".join(f"{cn}={sum(v) / len(v):.4g}" for cn, v in sorted(rows[kname].items()))
        print(f"{kname[:80]:80s} {len(dur[kname]):8d} {avg:10.1f}  {cs}")


# bench.py profiler label -> (kernel name as rocprofv3 prints it, algorithmic bytes per launch for B x Tx x fpi)
def _labels(B, Tx, fpi):
    Ty = Tx * fpi
    # the needles name the MATH_BF16X3 variants (math template argument 1 / k_wn_layer_b3): the default path bench.py times;
    # MI355VITS_MATH=f32 runs match the ", 0>" / k_wn_layer_h192 entries
    return {
        # round 3: the MRF stages of 64 / 32 channels on k_mrf_p (planes split once, weights in registers)
        "dec.mrf_p.s1": ("k_mrf_p<64,", 2 * 4 * B * 64 * Ty * 64),      # read x + write y, [64, 64 Ty]
        "dec.mrf_p.s2": ("k_mrf_p<32,", 2 * 4 * B * 32 * Ty * 256),     # [32, 256 Ty]
        "dec.mrf_fused.s0": ("k_mrf_fused<4, 2, 3, 3, 2, 160, 128, 1>", 2 * 4 * B * 128 * Ty * 8),
        "dec.mrf_fused.s1": ("k_mrf_fused<2, 4, 6, 3, 2, 320, 288, 1>", 2 * 4 * B * 64 * Ty * 64),
        "dec.mrf_fused.s2": ("k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 1>", 2 * 4 * B * 32 * Ty * 256),
        "flow.wn_layer_b3": ("k_wn_layer_b3<false, 3>", 4 * 4 * B * 192 * Ty + 6 * (384 * 192 * 5 + 384 * 192)),  # h r+w, skip r+w, bf16x3 weights
        # MATH_F16X2 (bench --math f16x2)
        "f16x2:dec.mrf_fused.s1": ("k_mrf_fused<2, 4, 6, 3, 2, 320, 288, 3>", 2 * 4 * B * 64 * Ty * 64),
        "f16x2:dec.mrf_fused.s2": ("k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 3>", 2 * 4 * B * 32 * Ty * 256),
        "f16x2:flow.wn_layer_b3": ("k_wn_layer_b3<false, 3, true>", 4 * 4 * B * 192 * Ty + 4 * (384 * 192 * 5 + 384 * 192)),
        "f32:dec.mrf_fused.s1": ("k_mrf_fused<2, 4, 6, 3, 2, 320, 288, 0>", 2 * 4 * B * 64 * Ty * 64),
        "f32:dec.mrf_fused.s2": ("k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 0>", 2 * 4 * B * 32 * Ty * 256),
        "f32:flow.wn_layer": ("k_wn_layer_h192<4>", 4 * 4 * B * 192 * Ty + 4 * (384 * 192 * 5 + 384 * 192)),
    }


def traffic(dirs, out_path, B=32, Tx=128, fpi=6):
    """profiles/pmc_traffic.json: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; on gfx950
    FETCH_SIZE tallies 64 B per 128-B read request, MI355X_MICROARCH.md "HBM")."""
    import json

    sums = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        c = _db(d)
        per = defaultdict(float)
        names = {}
        for disp, kname, cname, val in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection "
                                                 "where counter_name in ('FETCH_SIZE', 'WRITE_SIZE')"):
            per[(disp, cname)] += val
            names[disp] = kname
        for (disp, cname), v in per.items():
            sums[names[disp]][cname].append(v)
    out = {}
    for label, (needle, algo) in _labels(B, Tx, fpi).items():
        for kname, cs in sums.items():
            if needle in kname and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
                w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
                out[label] = {"workload": [B, Tx, fpi], "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                              "algorithmic_bytes_per_launch": int(algo), "fetch_size_kib": f, "write_size_kib": w,
                              "launches": len(cs["FETCH_SIZE"]),
                              "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of {needle}: "
                                        f"2 x FETCH_SIZE + WRITE_SIZE, per launch (gfx950 FETCH_SIZE counts 64 B per 128 B request)"}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def traffic_from_text(txt_path, out_path, B=32, Tx=128, fpi=6):
    """The same table from a `pmc` text summary (per-launch means of FETCH_SIZE / WRITE_SIZE in KiB)."""
    import json
    import re

    rows = {}
    for line in open(txt_path):
        f = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
        w = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
        n = re.search(r"\)\s+(\d+)\s+[0-9.]+\s+[A-Z]", line)
        if f and w:
            rows[line] = (float(f.group(1)), float(w.group(1)), int(n.group(1)) if n else 0)
    out = {}
    for label, (needle, algo) in _labels(B, Tx, fpi).items():
        for line, (f, w, n) in rows.items():
            if needle in line:
                out[label] = {"workload": [B, Tx, fpi], "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                              "algorithmic_bytes_per_launch": int(algo), "fetch_size_kib": f, "write_size_kib": w, "launches": n,
                              "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of {needle}: "
                                        f"2 x FETCH_SIZE + WRITE_SIZE, per launch (gfx950 FETCH_SIZE counts 64 B per 128 B request)"}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def derived(txt_path, min_us=20.0):
    """Per-kernel ratios from a `pmc` text summary: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024
    SIMDs); wait_any / wait_inst / active / valu = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU over
    SQ_WAVE_CYCLES; valu/mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too: `other/mfma` = the same minus one =
    plain vector instructions per MFMA); mfma_x = SQ_INSTS_MFMA as issued (compare with the algorithmic count); lds_confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
    read_MiB = 2 x FETCH_SIZE KiB / 1024 (gfx950: 64 B tallied per 128-B request), write_MiB = WRITE_SIZE KiB / 1024, per launch."""
    import re

    print(f"{'kernel':74s} {'calls':>6s} {'avg_us':>9s} {'mfma_busy':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'valu':>6s} "
          f"{'valu/mfma':>9s} {'other/mfma':>10s} {'insts_mfma':>11s} {'lds_confl':>9s} {'read_MiB':>9s} {'write_MiB':>9s}")
    for line in open(txt_path):
        m = re.match(r"(.{80})\s+(\d+)\s+([0-9.]+)\s+(.*)", line)
        if not m or "=" not in m.group(4):
            continue
        c = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.e+-]+)", m.group(4))}
        avg = float(m.group(3))
        if avg < min_us:
            continue
        g = lambda k: c.get(k, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        ratio = lambda a, b: a / b if b and b == b else float("nan")
        print(f"{m.group(1)[:74]:74s} {int(m.group(2)):6d} {avg:9.1f} {ratio(g('SQ_VALU_MFMA_BUSY_CYCLES'), g('GRBM_GUI_ACTIVE') / 8 * 1024):9.3f} "
              f"{ratio(g('SQ_WAIT_ANY'), wc):8.3f} {ratio(g('SQ_WAIT_INST_ANY'), wc):9.3f} {ratio(g('SQ_ACTIVE_INST_ANY'), wc):7.3f} "
              f"{ratio(g('SQ_ACTIVE_INST_VALU'), wc):6.3f} {ratio(g('SQ_INSTS_VALU'), g('SQ_INSTS_MFMA')):9.1f} {ratio(g('SQ_INSTS_VALU'), g('SQ_INSTS_MFMA')) - 1.0:10.2f} "
              f"{g('SQ_INSTS_MFMA'):11.4g} "
              f"{ratio(g('SQ_LDS_BANK_CONFLICT'), g('SQ_LDS_IDX_ACTIVE')):9.3f} {2 * g('FETCH_SIZE') / 1024:9.1f} {g('WRITE_SIZE') / 1024:9.1f}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "derived":
        derived(sys.argv[2])
        raise SystemExit(0)
    BATCH = int(os.environ.get("BATCH") or 256)  # the profiled bench workload: bench.py's default at --gpus 1 (tools/profile.sh passes BATCH on)
    if len(sys.argv) >= 4 and sys.argv[1] == "traffic-from-text":
        traffic_from_text(sys.argv[3], sys.argv[2], B=BATCH)
        raise SystemExit(0)
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "pmc", "traffic"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[3:], sys.argv[2], B=BATCH)
    else:
        pmc(sys.argv[2:])
