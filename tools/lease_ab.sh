# round-5 lease script: A/B of lab switches on one box.  VARIANTS = "tag:ENV=1,ENV2=0 tag2: ..." (tag with no env = the lab defaults);
# per variant one lab_bench run (kernel table printed), optionally TESTS first and PROFILE (rocprofv3 stats + counters, product build) last.
cd $GRAFT_REPO_ROOT
O=gpurun_out
TAG=${TAG:-r05}
python -c "from mimic3_amd._native import hooks_library; print(hooks_library().probe_device())" > $O/${TAG}_probe.txt 2>&1; cat $O/${TAG}_probe.txt | tail -2
if [ -n "$TESTS" ]; then timeout ${TEST_TIMEOUT:-1200} python -m pytest $TESTS -q -m gpu -x > $O/${TAG}_pytest.log 2>&1; tail -4 $O/${TAG}_pytest.log; fi
for v in $VARIANTS; do
  t=${v%%:*}; e=${v#*:}; e=${e//,/ }
  B=tools/lab_bench.py; case "$e" in *PRODUCT=1*) B=bench.py;; esac
  vb=$(echo "$e" | grep -o 'BATCH=[0-9]*' | cut -d= -f2); vs=$(echo "$e" | grep -o 'STEPS=[0-9]*' | cut -d= -f2)  # per-variant batch / steps
  env $e timeout 300 python $B --batch ${vb:-${BATCH:-32}} --steps ${vs:-${STEPS:-40}} --warmup 10 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/${TAG}_bench_$t.json 2> $O/${TAG}_bench_$t.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_bench_$t.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$t: ms/step %.3f | " % d["ms_per_step"] + "  ".join("%s=%s" % (k[3:], v) for k, v in r.items() if k.startswith("ms:")) + " | " + str(d["config"].get("device")) + " | arena " + str({k: v for k, v in (d.get("box_probe") or {}).items() if k.startswith("arena")}))
except Exception as ex:
    print("$t: no bench json:", ex)
PY
done
if [ -n "$PROFILE" ]; then TAG=$TAG STATS_ONLY=$STATS_ONLY bash tools/profile.sh > $O/${TAG}_profile.log 2>&1; python tools/rocprof_summary.py derived $O/${TAG}_rocprof_pmc.txt > $O/${TAG}_pmc_derived.txt 2>&1; head -20 $O/${TAG}_rocprof_stats.txt | cut -c1-160; head -30 $O/${TAG}_pmc_derived.txt | cut -c1-220; fi
