cd $GRAFT_REPO_ROOT
for E in 1 0; do
  if [ $E = 1 ]; then export MI355VITS_NO_B3_PC=1; else unset MI355VITS_NO_B3_PC; fi
  echo "== no_pc=$E"; timeout 100 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "upsample\|headline"
  timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-extra --no-b1 --no-roofline 2>&1 >/dev/null | grep "headline"
done
