cd $GRAFT_REPO_ROOT
for geom in 1 2; do for K in 1 2 3 4; do
  echo "== geom $geom K $K"
  python tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x16,3x16 --opt=sor_exact=0 --opt=sor_fuse=$K --opt=sor_tile=$geom $( [ $K = 2 ] && echo --check )
done; done
