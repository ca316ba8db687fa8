# VERDICT r2 "next" #3: where does the ~9.6 ms floor of the extra legs come from?  One math mode at a time in its own process,
# 1 / 2 / 3 engine handles in flight, 20 vs 200 timed steps
cd $GRAFT_REPO_ROOT
O=gpurun_out
for m in f16x2 bf16x3; do for st in 1 2 3; do for n in 20 200; do
  timeout 300 python bench.py --math $m --streams $st --steps $n --warmup 10 --no-extra --no-cpu-baseline --no-b1 --no-roofline --no-traffic 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); h=r['host_ms_per_step']
print('$m handles=$st steps=$n  ms_per_step=%.3f  one_handle_call=%.3f device=%.3f host_side=%.3f' % (r['ms_per_step'], h['one_handle_call_ms'], h['device_ms'], h['host_side_ms']))"
done; done; done
