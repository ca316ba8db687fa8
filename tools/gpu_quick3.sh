cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed" | tail -2
timeout 300 python bench.py --steps 200 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['engine_vs_oracle'])"
