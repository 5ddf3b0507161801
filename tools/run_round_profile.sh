set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --no-cpu-baseline --e2e-passes 0 > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-passes 0 > /dev/null 2> gpurun_out/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-passes 0 > /dev/null 2> gpurun_out/pmc2.err
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_traffic.json
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -2
cat gpurun_out/bench_default.json
