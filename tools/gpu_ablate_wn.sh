# (lab build: python -m mimic3_amd.build lab — the product library has no ablation switch)
cd $GRAFT_REPO_ROOT
for A in ${ABLATES:-0 1 2 4 3 5 6 7}; do
  echo "== wn ablate $A"
  MI355VITS_WN_ABLATE=$A timeout 120 python tools/lab_bench.py --no-traffic --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "wn_layer"
done
