cd $GRAFT_REPO_ROOT; C=discrete_mean_field_game_amd/csrc
cp $C/libmfg_hip.so /tmp/base.so
for v in base nowl base nowl; do
  if [ $v != base ]; then cp $C/libmfg_hip_$v.so $C/libmfg_hip.so; else cp /tmp/base.so $C/libmfg_hip.so; fi
  echo "== $v"; python tools/perf_probe.py 21,65536,15 2>&1 | grep -E "rollout|sample_d"
done
cp /tmp/base.so $C/libmfg_hip.so
