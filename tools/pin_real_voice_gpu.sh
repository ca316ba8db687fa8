#!/bin/bash
# The one command that turns `parity` green the day a real voice is at hand (VERDICT r2 "next" #9):
#   1. put en_UK/apope_low (generator.onnx, config.json, phonemes.txt — mimic3_tts/const.py:22-24 gives the URL,
#      mimic3_tts/voices.json:378+ the sha256) under  ./voices/en_UK/apope_low/  of this repository (it travels with the
#      gpurun snapshot), together with
#        voices/apope_sample.ids   the phoneme ids of tests/apope_sample.txt (one line of integers; printed by the reference's
#                                  `mimic3 --voice en_UK/apope_low --debug` or by tools/pin_real_voice.py --ids-from-reference)
#        voices/apope_sample_amd64.wav   the reference's golden (tests/apope_sample_amd64.wav)
#   2. run:  bash tools/pin_real_voice_gpu.sh
# Expected tail of the output (tests/test_real_voice_pin.py, tools/pin_real_voice.py):
#   files: generator.onnx sha256 matches voices.json ... ok
#   engine vs golden: <n> of 253696 int16 samples differ (<= 10 %: tests/samples_match.py criterion) ... ok
#   oracle vs golden: ... ok        <- this line pins ResBlock2 and the `scales` plumbing of oracle/vits_oracle.py
#   engine vs oracle on the real weights: rel RMS <= 1e-4 ... ok
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=${MI355VITS_VOICE_DIR:-$R/voices/en_UK/apope_low}
[ -f "$V/generator.onnx" ] || { echo "no voice at $V (see the header of this script)"; exit 2; }
CMD="MI355VITS_VOICE_DIR=$V MI355VITS_SAMPLE_IDS=${MI355VITS_SAMPLE_IDS:-$R/voices/apope_sample.ids} MI355VITS_SAMPLE_WAV=${MI355VITS_SAMPLE_WAV:-$R/voices/apope_sample_amd64.wav} python -m pytest tests/test_real_voice_pin.py -m gpu -q -s"
if command -v gpurun >/dev/null 2>&1 && [ ! -e /dev/kfd ]; then
  REL=${V#$R/}
  gpurun --timeout 900 -- "MI355VITS_VOICE_DIR=\$GRAFT_REPO_ROOT/$REL MI355VITS_SAMPLE_IDS=\$GRAFT_REPO_ROOT/voices/apope_sample.ids MI355VITS_SAMPLE_WAV=\$GRAFT_REPO_ROOT/voices/apope_sample_amd64.wav python -m pytest tests/test_real_voice_pin.py -m gpu -q -s"
else
  cd "$R" && eval "$CMD"
fi
