set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-passes 0 > /dev/null 2> gpurun_out/pmc_sq.err
python tools/pmc_summary.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_summary.txt
cat gpurun_out/pmc_sq_summary.txt
