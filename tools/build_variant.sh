#!/bin/bash
# EXTRA="-DFOO=1" adds compiler flags.
# Compiles the current HIP sources into grand_plus_amd/libgrandplus_<name>.so (git-ignored) for tools/ab.sh.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 $EXTRA -O3 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -munsafe-fp-atomics \
  -Iinclude -Igrand_plus_amd/csrc -o grand_plus_amd/libgrandplus_$1.so \
  grand_plus_amd/csrc/gfpush.hip grand_plus_amd/csrc/augment.hip grand_plus_amd/csrc/propagate.hip
