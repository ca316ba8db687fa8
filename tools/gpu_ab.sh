# same-box A/B of two builds of the library: libmi355vits_base.so (built from an earlier commit) vs the current one
cd $GRAFT_REPO_ROOT/mimic3_amd/csrc
cp libmi355vits.so /tmp/new.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then cp mimic3_amd/csrc/libmi355vits_base.so mimic3_amd/csrc/libmi355vits.so; else cp /tmp/new.so mimic3_amd/csrc/libmi355vits.so; fi
    echo "== $which (rep $rep) 3 streams"; timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-extra --no-b1 --no-roofline 2>&1 >/dev/null | grep "headline"
    echo "== $which (rep $rep) 1 stream"; timeout 200 python bench.py --steps 40 --streams 1 --no-cpu-baseline --no-extra --no-b1 --no-roofline 2>&1 >/dev/null | grep "headline"
  done
done
cp /tmp/new.so mimic3_amd/csrc/libmi355vits.so
