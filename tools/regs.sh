#!/bin/bash
# Register / scratch usage of the GFPush kernels for the current sources (EXTRA adds compiler flags).
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 $EXTRA -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Iinclude -Igrand_plus_amd/csrc \
  -c grand_plus_amd/csrc/gfpush.hip -o /tmp/regs_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep gfpush_kernel
rm -f /tmp/regs_$$.o
