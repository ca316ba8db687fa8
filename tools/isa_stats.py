#!/usr/bin/env python3
"""Per-kernel instruction counts of a hipcc --save-temps .s file: MFMAs, full drains (vmcnt(0)), scratch, LDS, barriers.
usage: tools/isa_stats.py file.s [name-filter]"""
import re
import sys

PATS = [("mfma", r"v_mfma"), ("vmcnt0", r"vmcnt\(0\)"), ("lgkm0", r"lgkmcnt\(0\)"), ("ds_r", r"ds_read"), ("ds_w", r"ds_write"),
        ("vload", r"(buffer|global)_load"), ("vstore", r"(buffer|global)_store"), ("scratch", r"scratch_"), ("barrier", r"s_barrier"),
        ("nop", r"s_nop"), ("valu", r"(?m)^\s+v_(?!mfma)")]
text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(name), text)
    sp = re.search(r"\.set %s\.private_seg_size, (\d+)" % re.escape(name), text)
    cols = " ".join("%s %d" % (k, len(re.findall(p, body))) for k, p in PATS)
    print("%-64s vgpr %s scratchB %s %s lines %d" % (name[:64], vg.group(1) if vg else "?", sp.group(1) if sp else "?", cols, body.count("\n")))
