#!/usr/bin/env python3
"""bench.py on the LAB build of the library (libmi355vits_lab.so, -DMI355_LAB: timing / kernel-choice experiments that the
product library does not carry).  Same flags as bench.py.  Build: python -m mimic3_amd.build lab"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mimic3_amd._native as N  # noqa: E402

N.DEFAULT_LIBRARY = os.environ.get("MI355VITS_LAB_LIB") or os.path.join(ROOT, "mimic3_amd", "csrc", "libmi355vits_lab.so")  # (a variant build)
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
