cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do
  timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_ab_q.json 2> $O/r03_ab_q.err
  echo "product (packed) rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_ab_q.json)"; grep "dec.mrf_p" $O/r03_ab_q.err
  timeout 300 python tools/lab_bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_ab_p.json 2> $O/r03_ab_p.err
  echo "variant (scalar, no slp) rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_ab_p.json)"; grep "dec.mrf_p" $O/r03_ab_p.err
done
for st in 2 3 4; do timeout 300 python bench.py --steps 60 --streams $st --no-extra --no-cpu-baseline --no-traffic --no-b1 --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('streams $st', r['ms_per_step'])"; done
