cd $GRAFT_REPO_ROOT
for wl in 1 2 3 6; do
  echo "== brox tolerance mode, wave levels $wl"
  python tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16,3x16 --opt=sor_exact=0 --opt=sor_wave_levels=$wl 2>&1 | grep config | cut -c1-250
  python tools/sor_one_pair.py brox sor_wave_levels=$wl 2>&1 | grep which | cut -c1-250
done
