#!/bin/bash
# the second pass's refill policy under the two-pass render (C4): lanes that must be free before a refill x minimum run
for W in 8 12 16 24 32; do for R in 4 8 16; do echo "== want $W minrun $R"; FR_DEBUG_QUEUE_WANT=$W FR_DEBUG_QUEUE_MINRUN=$R python tools/c4_ab.py 11 2>&1 | grep "^C4"; done; done
