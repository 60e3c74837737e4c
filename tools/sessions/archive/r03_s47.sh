#!/bin/bash
# round 3 session 47: kernel stats + SQ counters of one 640x480 TV-L1-with-occlusions solve in its final form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03au; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/occ_t -- python3 $R/tools/bench_tvl1occ.py --size 640x480 --cpu none > $O/trace.log 2>&1 || { tail -3 $O/trace.log; exit 1; }
grep '"size"' $O/trace.log | cut -c1-200
python3 $R/tools/fmt_kernel_stats.py /tmp/occ_t 12 | tee $O/kernel_stats.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/occ_sq -- python3 $R/tools/bench_tvl1occ.py --size 640x480 --cpu none > $O/sq.log 2>&1 || { tail -3 $O/sq.log; exit 1; }
python3 $R/tools/pmc_sum.py /tmp/occ_sq k_rof_window k_occ_chi_fused | tee $O/sq_summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/occ_lds -- python3 $R/tools/bench_tvl1occ.py --size 640x480 --cpu none > $O/lds.log 2>&1 || { tail -3 $O/lds.log; exit 1; }
python3 $R/tools/pmc_sum.py /tmp/occ_lds k_rof_window k_occ_chi_fused | tee $O/lds_summary.txt
