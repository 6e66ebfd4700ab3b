cd $GRAFT_REPO_ROOT
for s in "time_volume.py" "time_convs.py" "time_configs.py" "time_matching_encoder.py" "time_big_tsdf.py" "mc_profile_probe.py" "e2e_profile_probe.py" "stress_cross_wg.py 20" "time_conv_layer.py" "time_volume_split.py" "host_issue_time.py" "time_convs_streams.py" "plan_debug_probe.py" "phase_schedule_probe.py"; do
  echo "== $s"; timeout 120 python scripts/$s > /tmp/out.txt 2>&1; rc=$?; echo "   rc=$rc"; [ $rc -ne 0 ] && tail -5 /tmp/out.txt
done
true
