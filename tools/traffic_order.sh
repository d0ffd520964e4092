#!/bin/bash
# FETCH_SIZE / L2 hit rate of the conv kernel under different workgroup->tile orders (PV_TILE_ORDER bit0 = M fastest, bit1 = no XCD remap)
export PMC_PASSES=3,4,5
for ord in -1 1 2 3; do
  echo "#### PV_TILE_ORDER=$ord"
  if [ "$ord" = "-1" ]; then unset PV_TILE_ORDER; else export PV_TILE_ORDER=$ord; fi
  python3 tools/pmc.py "conv3 320->320 @64" gemm_conv 2>&1 | grep -v "^$"
  python3 tools/kbench.py "conv3 320->320 @64" 2>&1 | tail -2
done
