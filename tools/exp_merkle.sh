#!/bin/bash
# Development tool: prove time of the 2^22 x 4 trace with the specialised and the generic Merkle kernels
for k in "" "SMI_MERKLE_GENERIC=1" "" "SMI_MERKLE_GENERIC=1"; do
  echo "== ${k:-default}"
  env $k REPS=10 python tools/kbench.py prove:22:3:4 2>/dev/null | head -5
done
