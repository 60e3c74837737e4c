cd $GRAFT_REPO_ROOT
python tools/check_sor_tile.py --no-big 2>&1 | grep -E "MISMATCH" | head
python tools/sor_one_pair.py hs 2>&1 | grep which | cut -c1-200
python tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x16,3x16 --opt=sor_exact=0 2>&1 | grep config | cut -c1-250
