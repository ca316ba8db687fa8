cd $GRAFT_REPO_ROOT
for rep in 1 2; do for E in 1 0; do
  if [ $E = 1 ]; then export MI355VITS_NO_B3_PC_STD=1; else unset MI355VITS_NO_B3_PC_STD; fi
  echo "== no_pc_std=$E"; timeout 100 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "rb.s0\|conv_pre\|ffn1\|headline\|total"
  timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-extra --no-b1 --no-roofline 2>&1 >/dev/null | grep "headline"
done; done
unset MI355VITS_NO_B3_PC_STD
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d or split or bench_workload or invariance or properties" 2>&1 | tail -2
