cd $GRAFT_REPO_ROOT
for A in 0 8; do echo "== mrf ablate $A"; MI355VITS_MRF_ABLATE=$A timeout 100 python bench.py --steps 20 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "mrf_fused"; done
