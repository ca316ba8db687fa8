cd $GRAFT_REPO_ROOT
export MI355VITS_NO_POST_FUSION=1
bash tools/gpu_ab.sh
