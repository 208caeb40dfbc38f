#!/bin/bash
# first pass of the two-pass render at reduced occupancy (unused dynamic LDS) and with 8 bands per workgroup
for B in 0 8; do for L in 0; do echo "== bands $B extra LDS $L"; FR_DEBUG_FIRST_BANDS=$B FR_DEBUG_FIRST_LDS=$L python tools/c4_ab.py 11 2>&1 | grep "^C4"; done; done
