cd $GRAFT_REPO_ROOT
O=gpurun_out
{
ls /sys/class/drm/
for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | tr '\n' ' '; echo; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level current_compute_partition current_memory_partition; do echo "-- $f"; cat $c/$f 2>&1 | head -12; done; ls $c/hwmon/*/ | tr '\n' ' '; for f in $c/hwmon/*/freq*_input $c/hwmon/*/freq*_label $c/hwmon/*/power1_* ; do echo "$f: $(cat $f 2>&1)"; done; done
echo ==== rocm-smi
rocm-smi --showclocks --showpower --showperflevel --showcomputepartition --showmemorypartition 2>&1 | head -60
echo ==== amd-smi
amd-smi metric -g 0 --clock --power 2>&1 | head -80
amd-smi static -g 0 --limit 2>&1 | head -40
} > $O/r04_devstate.txt 2>&1
B="python bench.py --no-cpu-baseline --no-traffic --no-extra"
timeout 300 $B --steps 20 --warmup 5 > $O/r04_p1_a.json 2> $O/r04_p1_a.err; grep -o '"ms_per_step": [0-9.]*' $O/r04_p1_a.json | head -1; grep "wn_layer_b3\|mrf_p.s1\|upsample" $O/r04_p1_a.err
timeout 300 $B --steps 200 --warmup 10 > $O/r04_p1_b.json 2> $O/r04_p1_b.err; grep -o '"ms_per_step": [0-9.]*' $O/r04_p1_b.json | head -1; grep "wn_layer_b3\|mrf_p.s1\|upsample" $O/r04_p1_b.err
timeout 300 $B --steps 20 --warmup 5 > $O/r04_p1_c.json 2> $O/r04_p1_c.err; grep -o '"ms_per_step": [0-9.]*' $O/r04_p1_c.json | head -1; grep "wn_layer_b3\|mrf_p.s1\|upsample" $O/r04_p1_c.err
