cd $GRAFT_REPO_ROOT
python tools/ab_bench.py fixed3=,fuse3_cursor=0 cursor=,fuse3_cursor=1 cursor_a2_20=,fuse3_afac2=2.0 cursor_a2_13=,fuse3_afac2=1.3 cursor_no1=,fuse3_afac1=0.5 cursor_a1_12=,fuse3_afac1=1.2 --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-cli --no-single --no-other-mode --fixed-steps 0"
