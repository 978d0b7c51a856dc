#!/bin/bash
# One more build of the product library under another name, for A/B runs on ONE box (tools/ab_libs.sh):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DGP_SK_TIMING]     -> grand_plus_amd/libgrandplus_<name>.so
# With GP_SRC=<dir> the sources come from that directory (e.g. an older revision exported with `git show`).
NAME=$1; shift
SRC=${GP_SRC:-grand_plus_amd/csrc}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -munsafe-fp-atomics \
  -Iinclude -I$SRC "$@" -o grand_plus_amd/libgrandplus_$NAME.so $SRC/gfpush.hip grand_plus_amd/csrc/augment.hip grand_plus_amd/csrc/propagate.hip
