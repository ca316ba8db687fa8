cd $GRAFT_REPO_ROOT
for rep in 1 2; do for A in 16 0; do echo "== conv ablate $A (16 = no XCD renumbering)"; MI355VITS_CONV_ABLATE=$A timeout 100 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "upsample\|headline"; done; done
