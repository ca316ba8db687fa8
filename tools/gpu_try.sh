cd $GRAFT_REPO_ROOT
timeout 600 python tools/math_modes_vs_fp64.py 2>&1 | tail -6
