cd $GRAFT_REPO_ROOT
for tw in 64 128; do
  echo "== brox tolerance mode, tile width $tw"
  python tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16,3x16 --opt=sor_exact=0 --opt=sor_tile_w=$tw $( [ $tw = 64 ] && echo --check )
done
echo "== brox exact"
python tools/bench_sor_groups.py --only=brox_cfg4 --grid=3x16
