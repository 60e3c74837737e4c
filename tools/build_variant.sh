#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>": an A/B build of libofx.so under variants/ (git-ignored; travels to the
# GPU box).  Use with OFX_LIB_PATH=variants/libofx_<name>.so.
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
make -C $R/optical-flow-1_amd/csrc OUT=$R/variants/libofx_$name.so BUILD=$R/variants/build_$name EXTRA="$*" 2>&1 | grep -E "error|warning: " ; ls -la $R/variants/libofx_$name.so
