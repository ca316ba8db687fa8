# (lab build: python -m mimic3_amd.build lab — the product library has no ablation switch)
cd $GRAFT_REPO_ROOT
for A in ${ABLATES:-0 3 4 6}; do
  echo "== ablate $A"
  MI355VITS_CONV_ABLATE=$A timeout 120 python tools/lab_bench.py --no-traffic --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "upsample\|headline\|conv_pre\|rb.s0"
done
