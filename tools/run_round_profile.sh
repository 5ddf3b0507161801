# Round profile, part 1 (run on the GPU box through gpurun): tests, default bench, rocprofv3 kernel stats of the same command, HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md) at the bench's default step count, f32 and bf16 configurations.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r03}
python -m pytest tests -m gpu -x -q > gpurun_out/${R}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.txt
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/prof.err
cp $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc2.err
python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/${R}_bench_bf16_config.json 2> gpurun_out/bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_bf16 -- python3 bench.py --precision bf16 --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/prof_bf16.err
cp $(find gpurun_out/prof_stats_bf16 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bf16_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_bf16 -- python3 bench.py --precision bf16 --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_bf16 -- python3 bench.py --precision bf16 --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc4.err
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch_bf16 gpurun_out/pmc_write_bf16 > gpurun_out/${R}_pmc_traffic.json
head -c 600 gpurun_out/${R}_bench.json
