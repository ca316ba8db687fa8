# one parameterised lease script (replaces the per-experiment gpu_r0N_x.sh files):
#   TESTS="tests/test_gpu_parity.py -k bf16"  pytest selection (default: none)        -> gpurun_out/${TAG}_pytest.log
#   BENCH="--steps 60 --no-extra ..."         bench.py arguments (default: no bench)  -> gpurun_out/${TAG}_bench.json / .err
#   LAB=1                                     bench on the lab build (tools/lab_bench.py), with the MI355VITS_* switches of the caller
#   PROFILE=1                                 rocprofv3 kernel stats + PMC passes (tools/profile.sh)
cd $GRAFT_REPO_ROOT
O=gpurun_out
TAG=${TAG:-run}
if [ -n "$TESTS" ]; then timeout ${TEST_TIMEOUT:-1500} python -m pytest $TESTS -q -m gpu -x > $O/${TAG}_pytest.log 2>&1; tail -5 $O/${TAG}_pytest.log; fi
if [ -n "$BENCH" ]; then
  B=bench.py; [ -n "$LAB" ] && B=tools/lab_bench.py
  ( time timeout ${BENCH_TIMEOUT:-900} python $B $BENCH > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err ) 2>&1 | grep real
  python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("ms/step", round(d["ms_per_step"], 3), "value %.4g" % d["value"])
    for k, v in d.get("config", {}).items():
        print("  config.%s: %s" % (k, v))
    r = d.get("roofline", {})
    print("  roofline:", r.get("kernel"), "frac %.3f" % r.get("frac", 0), "traffic", r.get("traffic"))
    print("  " + "  ".join("%s=%s" % (k[3:], v) for k, v in r.items() if k.startswith("ms:")))
    c = d.get("cpu_baseline", {})
    print("  cpu:", c.get("value"), c.get("cores"), c.get("thread_sweep"))
except Exception as e:
    print("no bench json:", e)
PY
  grep -i "error\|Traceback" $O/${TAG}_bench.err | head -5
fi
if [ -n "$PROFILE" ]; then TAG=$TAG bash tools/profile.sh > $O/${TAG}_profile.log 2>&1; head -8 $O/${TAG}_rocprof_stats.txt | cut -c1-150; fi
