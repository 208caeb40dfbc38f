#!/bin/bash
# The second pass's vector instructions on C4 (f64) as a function of its refill threshold (FR_DEBUG_QUEUE_WANT lanes free
# before a refill; minimum run 8): one rocprofv3 PMC pass per setting.  tools/sim/second_pass_drain.py gives wave-iterations
# and episodes per 64-entry chunk for the same settings; profiles/r03_second_pass_cost_model.txt fits
#     instructions per chunk = 6.75 x wave-iterations + E x episodes + F        (E: per episode, F: per finishing batch)
# Usage (GPU box, repo root): bash tools/second_pass_cost_model.sh
for W in 4 8 12 16 24 32 48; do
  FR_DEBUG_QUEUE_WANT=$W FR_DEBUG_QUEUE_MINRUN=8 bash tools/pmc_valu.sh sp_want_$W --no-extras --view julia --iterations 4096 --precision f64 2>/dev/null | grep "second_kernel" | sed "s/^/want $W: /"
done
