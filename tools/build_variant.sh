#!/bin/bash
# Compiles the current HIP sources into grand_plus_amd/libgrandplus_<name>.so (git-ignored) for tools/ab.sh.
# EXTRA="-DFOO=1" adds compiler flags to the GFPush translation unit.  The two other translation units are
# compiled once into build/ and re-linked.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -munsafe-fp-atomics -Iinclude -Igrand_plus_amd/csrc"
for f in augment propagate; do
  if [ ! -f build/$f.o ] || [ grand_plus_amd/csrc/$f.hip -nt build/$f.o ]; then hipcc $FLAGS -c grand_plus_amd/csrc/$f.hip -o build/$f.o; fi
done
hipcc $FLAGS $EXTRA -c grand_plus_amd/csrc/gfpush.hip -o build/gfpush_$1.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o grand_plus_amd/libgrandplus_$1.so build/gfpush_$1.o build/augment.o build/propagate.o
