"""Run by tests/test_oracle_and_abi.py in a subprocess (a crash must not take pytest down): corrupted .m355 containers
through mi355vits_create_from_buffer on the CPU model of the library.  Exit code 0 = every case ended in an error code
or a loadable voice; anything else (signal, exception) fails the test."""
import sys

import numpy as np

from mimic3_amd import build, weights as W
from mimic3_amd._native import Engine, NativeError, NativeLibrary
from mimic3_amd.config import VitsConfig

lib = NativeLibrary(build.build_emu())
cfg = VitsConfig.tiny()
blob = bytearray(W.pack(cfg, W.synthetic_weights(cfg, seed=2)))
header = blob.find(b"\0" * 64, 600)  # somewhere past the config block and into the tensor table
rng = np.random.default_rng(11)
ok = err = 0
for trial in range(400):
    b = bytearray(blob)
    mode = trial % 4
    if mode == 0:    # flip bits in the header / config block / tensor table
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, 12000))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:  # truncate
        b = b[: int(rng.integers(0, len(b)))]
    elif mode == 2:  # overwrite a 4-byte field with an extreme value
        pos = int(rng.integers(8, 12000)) & ~3
        b[pos:pos + 4] = int(rng.choice([0, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 1 << 20])).to_bytes(4, "little")
    else:            # random bytes behind a valid magic
        b = bytearray(b"M355VITS") + bytearray(rng.integers(0, 256, int(rng.integers(0, 4096)), dtype=np.uint8).tobytes())
    try:
        e = Engine(bytes(b), library=lib)
        e.close()
        ok += 1
    except NativeError:
        err += 1
print(f"loaded {ok}, rejected {err}")
sys.exit(0 if ok + err == 400 and err > 100 else 1)
