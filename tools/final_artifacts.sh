#!/bin/bash
# The evidence bench.py's line points at, for the library as built NOW (run on the GPU box, repo root; ~4 min):
#   rocprofv3 kernel-trace + PMC passes of C4-f32, C4-f64, C2 (tools/pmc_sq.sh) and one default bench.py run.
# Afterwards, in the build container:  python tools/pmc_record.py <key>=gpurun_out/pmc_r03_<...> ...  (see the end of this file)
set -u
TAG=${1:-r03}
bash tools/pmc_sq.sh ${TAG}_c4_f32 --view julia --iterations 4096 --precision f32 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c4_f64 --view julia --iterations 4096 --precision f64 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c2_f64 --no-cpu-baseline > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_bench_default_n1.json 2> gpurun_out/${TAG}_bench.err
tail -c 300 gpurun_out/${TAG}_bench.err
for t in c4_f32 c4_f64 c2_f64; do grep "stats pass" gpurun_out/pmc_${TAG}_$t/summary.txt | cut -c1-120; done
# then:
#   python tools/pmc_record.py 16384x16384_i4096_f32_julia=gpurun_out/pmc_${TAG}_c4_f32 16384x16384_i4096_f64_julia=gpurun_out/pmc_${TAG}_c4_f64 \
#          16384x16384_i1024_f64_default=gpurun_out/pmc_${TAG}_c2_f64
#   cp gpurun_out/pmc_${TAG}_<cfg>/summary.txt profiles/${TAG}_<cfg>_rocprofv3.txt ; python tools/first_pass_classes.py > profiles/${TAG}_c4_first_pass_classes.txt
