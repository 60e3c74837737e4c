#!/bin/bash
# round 3 session 12: strip-height model under concurrency (waves per round 1024 / 2048 / 3072), both modes, A/B on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout -k 10 900 python tools/ab_bench.py s1024= s2048=,rows_slots=2048 s3072=,rows_slots=3072 s512=,rows_slots=512 --rounds 2 --args "--no-cpu --no-sor --no-occ --no-4k --no-single" > $O/ab_slots.txt 2>&1; cat $O/ab_slots.txt
