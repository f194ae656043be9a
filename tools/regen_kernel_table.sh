#!/bin/bash
# KERNELS.md = registers / LDS / scratch / spills of every kernel in the SHIPPED gnuradio4_amd/libgr4hip.so (llvm-readelf --notes of its gfx950 code objects; no GPU needed).
# Regenerate after every change of csrc/ -- tests/test_abi_host.py::test_kernel_table_is_current compares the committed file with the built library.
cd "$(dirname "$0")/.."
{
  echo "# KERNELS — per-kernel resources of the shipped \`libgr4hip.so\` (generated: \`tools/regen_kernel_table.sh\`; do not edit)"
  echo
  echo "gfx950, wave64, 512 VGPRs per SIMD lane-slot pool: waves per SIMD = floor(512 / VGPR) (AGPRs share the pool). LDS B = static allocation only; dynamic LDS is in DESIGN.md's kernel table."
  echo
  python tools/kernel_resources.py --md 2>/dev/null
} > KERNELS.md
wc -l KERNELS.md
