#!/bin/bash
# Exact SOR windows: rows per workgroup 128 (row blocks start on 128-byte lines of the hyperplane-major planes) against the default 125 (+ 3 border items = two waves); one box.
mkdir -p gpurun_out
{
for rows in 125 128 124 120; do
  echo "== sor_rows=$rows"
  timeout -k 10 400 python tools/bench_sor_groups.py --grid=3x16 --opt=sor_rows=$rows 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r04_sor_exact_rows_alignment.txt 2>&1
cat gpurun_out/r04_sor_exact_rows_alignment.txt | cut -c1-300
