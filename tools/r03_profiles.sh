#!/bin/bash
# Round-3 evidence in one go (each step writes under gpurun_out/ so that progress is visible): kernel-trace stats of the headline bench,
# of config-2 / config-4 solves and of a whole config-5 solve; a separate --pmc FETCH_SIZE pass over the resident grid (gpurun refuses
# --pmc beside tracing domains other than --kernel-trace).  rocprofv3 gets python3 itself behind `--`.
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
mkdir -p gpurun_out
BENCH_MIN="--no-cpu-baseline --no-microbench --concurrent 0 --no-other-configs --no-validator --sharded-pivots 0"
echo "== bench under rocprofv3 (config 3)"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c3 -- python3 bench.py --steps 2 --warmup 1 $BENCH_MIN > gpurun_out/r03_bench_under_rocprofv3.json 2> gpurun_out/prof_r03_c3.err || exit 2
echo "== config 2"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c2 -- python3 bench.py --workload config2 --steps 2 --warmup 1 $BENCH_MIN > gpurun_out/r03_bench_config2_under_rocprofv3.json 2> gpurun_out/prof_r03_c2.err || exit 3
echo "== config 4"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c4 -- python3 bench.py --workload config4 --steps 2 --warmup 1 $BENCH_MIN > gpurun_out/r03_bench_config4_under_rocprofv3.json 2> gpurun_out/prof_r03_c4.err || exit 4
echo "== config 5, whole solve"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c5 -- python3 tools/gpu_sizeclass.py config5 > gpurun_out/r03_config5_under_rocprofv3.txt 2>&1 || exit 5
echo "== pmc FETCH_SIZE, resident grid"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r03_res -- python3 tools/pmc_scan.py resident > gpurun_out/r03_pmc_resident.txt 2>&1 || exit 6
echo "== pmc FETCH_SIZE, config-5 scans (calibration: flush_kernel streams 512 MiB)"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r03_rc5 -- python3 tools/pmc_scan.py rc5 > gpurun_out/r03_pmc_rc5.txt 2>&1 || exit 7
find gpurun_out -name "*kernel_stats.csv" -newer tools/r03_profiles.sh | head
echo done
