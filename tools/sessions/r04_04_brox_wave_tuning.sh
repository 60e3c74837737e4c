cd $GRAFT_REPO_ROOT
for tw in 64 128; do for P in 2 4 6 8; do
  echo "== brox tolerance mode, tile width $tw, prefetch $P"
  python tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16 --opt=sor_exact=0 --opt=sor_tile_w=$tw --opt=sor_wave_p=$P
done; done
for P in 4 8; do echo "== one pair P=$P"; python tools/sor_one_pair.py brox sor_wave_p=$P; done
