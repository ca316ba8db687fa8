cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for E in 0 1; do
  if [ $E = 1 ]; then export MI355VITS_ENC_B3=1; else unset MI355VITS_ENC_B3; fi
  echo "== enc_b3 $E"; timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-extra --no-roofline 2>&1 >/dev/null | grep "headline\|total"
done; done
export MI355VITS_ENC_B3=1
timeout 200 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "ffn\|headline\|total\|conv_pre\|rb.s0"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
