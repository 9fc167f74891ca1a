#!/bin/bash
# Round 4, on the GPU box (gpurun): kernel-trace stats, the two HBM PMC passes and two SQ passes (LDS bank conflicts, flat / LDS / vector-memory
# instructions, wave-parked and issue-stall cycles: the non-streaming 59 % of k_solve) for the reduced bench command, and the pattern micro-benchmark.
# Outputs under gpurun_out/; the summaries are written into profiles/ by scripts/pmc_summary.py and scripts/sq_summary.py afterwards (CPU).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out
CMD="bench.py --steps 2 --warmup 1 --no-cpu --exact-sample 0 --handles 1 --closed-loop-steps 0 --no-extra-legs"
rm -rf gpurun_out/r4_prof_stats gpurun_out/r4_pmc_fetch gpurun_out/r4_pmc_write gpurun_out/r4_pmc_sqa gpurun_out/r4_pmc_sqb
rocprofv3 -L > gpurun_out/r4_counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4_prof_stats" -- python3 $CMD > gpurun_out/r4_prof_stats.log 2>&1 && echo stats ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/r4_pmc_fetch" -- python3 $CMD > gpurun_out/r4_pmc_fetch.log 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/r4_pmc_write" -- python3 $CMD > gpurun_out/r4_pmc_write.log 2>&1 && echo write ok
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_LDS --output-format csv -d "$R/gpurun_out/r4_pmc_sqa" -- python3 $CMD > gpurun_out/r4_pmc_sqa.log 2>&1 && echo sqa ok
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$R/gpurun_out/r4_pmc_sqb" -- python3 $CMD > gpurun_out/r4_pmc_sqb.log 2>&1 && echo sqb ok
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o sector_rmw sector_rmw.hip 2> "$R/gpurun_out/r4_sector_build.log" && {
  ./sector_rmw 256 512 8 380 1500 0; ./sector_rmw 256 512 8 1000 600 0; ./sector_rmw 256 512 16 500 1500 0; ./sector_rmw 256 512 8 380 1500 2; ./sector_rmw 256 512 8 380 1500 4; ./sector_rmw 256 512 8 380 1500 0 150; ./sector_rmw 256 512 8 380 1500 0 300; } > "$R/gpurun_out/r4_sector_rmw.txt" 2>&1)
tail -3 gpurun_out/r4_sector_rmw.txt
echo profiles collected
