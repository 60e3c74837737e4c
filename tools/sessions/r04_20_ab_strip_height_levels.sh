#!/bin/bash
# Job-level A/B of k_tvl1_iter3's strip height per pyramid level (OFX_ROWS3 knob); the model picks 32 / 28 / 16 at 1080 / 540 / 270.
mkdir -p gpurun_out
python tools/ab_bench.py base= \
  a=,env:OFX_ROWS3=1080:24 b=,env:OFX_ROWS3=1080:16 c=,env:OFX_ROWS3=1080:20 \
  d=,env:OFX_ROWS3=540:16 e=,env:OFX_ROWS3=540:20 f=,env:OFX_ROWS3=540:12 \
  g=,env:OFX_ROWS3=270:8 h=,env:OFX_ROWS3=270:12 i=,env:OFX_ROWS3=270:24 \
  --rounds 3 --args "--no-cpu --no-sor --no-occ --no-4k --no-cli" > gpurun_out/r04_ab_strip_height_levels.txt 2>&1
grep MEDIAN gpurun_out/r04_ab_strip_height_levels.txt
