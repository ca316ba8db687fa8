cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 900 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
r=json.load(open('gpurun_out/r03_bench_default.json'))
print({k:r[k] for k in ('value','ms_per_step','steps')})
print('host',r.get('host_ms_per_step'))
rl=r['roofline']; print('roofline',rl['kernel'],rl['achieved'],rl['frac'],rl['traffic'],rl['avg_launch_us'])
print('b1',r['latency_b1']['ms_median'])
for k,v in r.get('extra',{}).items(): print(k, v.get('ms_per_step'), v.get('value'))
print('cpu',r['cpu_baseline']['value'], r['cpu_baseline']['engine_vs_oracle'])
PY
grep -A14 "per-kernel (HIP events):" $O/r03_bench_default.err | head -40
