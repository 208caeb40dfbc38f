#!/bin/bash
# first pass of the two-pass render: bands per workgroup (FR_DEBUG_FIRST_BANDS) and reduced occupancy (FR_DEBUG_FIRST_LDS)
for rep in 1 2; do for B in 0 2 1; do echo "== bands $B"; FR_DEBUG_FIRST_BANDS=$B python tools/c4_ab.py 11 2>&1 | grep "^C4"; done; done
