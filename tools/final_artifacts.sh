#!/bin/bash
# The evidence bench.py's line points at, for the library as built NOW (run on the GPU box, repo root; ~8 min):
#   rocprofv3 kernel-trace + PMC passes (tools/pmc_sq.sh) of EVERY driver-timed line — C2 f64 / f32, C3, C4 f32 / f64, C5's
#   image — the C4 ablations (tools/c4_ablation.sh) and one default bench.py run.
# Afterwards, in the build container:  bash tools/final_artifacts.sh record <tag>   (pmc_record.py + the summaries into profiles/)
set -u
if [ "${1:-}" = "record" ]; then
  TAG=${2:-r04}
  python tools/pmc_record.py 16384x16384_i1024_f64_default=gpurun_out/pmc_${TAG}_c2_f64 16384x16384_i1024_f32_default=gpurun_out/pmc_${TAG}_c2_f32 \
      16384x16384_i65536_f64_zoom1e6=gpurun_out/pmc_${TAG}_c3_f64 16384x16384_i4096_f32_julia=gpurun_out/pmc_${TAG}_c4_f32 \
      16384x16384_i4096_f64_julia=gpurun_out/pmc_${TAG}_c4_f64 65536x65536_i1024_f64_default=gpurun_out/pmc_${TAG}_c5_f64
  for t in c2_f64 c2_f32 c3_f64 c4_f32 c4_f64 c5_f64; do cp gpurun_out/pmc_${TAG}_$t/summary.txt profiles/${TAG}_${t}_rocprofv3.txt; done
  for p in f32 f64; do cp gpurun_out/c4_ablation_$p.txt profiles/${TAG}_c4_ablation_$p.txt; done
  cp gpurun_out/${TAG}_bench_default_n1.json profiles/${TAG}_bench_default_n1.json
  exit 0
fi
TAG=${1:-r04}
bash tools/pmc_sq.sh ${TAG}_c2_f64 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c2_f32 --precision f32 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c4_f32 --view julia --iterations 4096 --precision f32 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c4_f64 --view julia --iterations 4096 --precision f64 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c5_f64 --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_c3_f64 --view zoom1e6 --iterations 65536 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for t in c2_f64 c2_f32 c4_f32 c4_f64 c5_f64 c3_f64; do grep "stats pass" gpurun_out/pmc_${TAG}_$t/summary.txt | cut -c1-150; done
bash tools/c4_ablation.sh f32 > /dev/null 2>&1
bash tools/c4_ablation.sh f64 > /dev/null 2>&1
tail -4 gpurun_out/c4_ablation_f32.txt
python bench.py > gpurun_out/${TAG}_bench_default_n1.json 2> gpurun_out/${TAG}_bench.err
tail -c 300 gpurun_out/${TAG}_bench.err
