cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/r04_tr_brox -o t -- python3 $R/tools/bench_sor_groups.py --only=brox_cfg4 --grid=1x16 --no-warm --opt=sor_exact=0 --opt=sor_tile_w=128 > $R/gpurun_out/r04_tr_brox.txt 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/r04_tr_hs -o t -- python3 $R/tools/bench_sor_groups.py --only=hs_cfg3 --grid=1x16 --no-warm --opt=sor_exact=0 --opt=sor_fuse=2 --opt=sor_tile=2 > $R/gpurun_out/r04_tr_hs.txt 2>&1
