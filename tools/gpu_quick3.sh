cd $GRAFT_REPO_ROOT
for rep in 1 2; do for E in 0 1; do
  if [ $E = 1 ]; then export MI355VITS_MRF_S2_4W=1; else unset MI355VITS_MRF_S2_4W; fi
  echo "== s2_4w=$E"; timeout 100 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "mrf_fused.s2\|headline"
  timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-extra --no-b1 --no-roofline 2>&1 >/dev/null | grep "headline"
done; done
