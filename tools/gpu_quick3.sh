cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh
timeout 200 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "headline\|mrf\|rb.s0\|wn_layer\|upsample\|conv_pre"
