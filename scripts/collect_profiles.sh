#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats and the two PMC passes for the default bench workload.
# Outputs under gpurun_out/; copy the summaries into profiles/ afterwards (scripts/pmc_summary.py).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --exact-sample 0 --handles 1 --closed-loop-steps 0 > gpurun_out/prof_stats.log 2>&1
tail -1 gpurun_out/prof_stats.log | cut -c1-400
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --exact-sample 0 --handles 1 --closed-loop-steps 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --exact-sample 0 --handles 1 --closed-loop-steps 0 > gpurun_out/pmc_write.log 2>&1
echo profiles collected
