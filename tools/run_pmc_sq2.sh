# Round profile, part 2: SQ + LDS counters of the hot kernels (separate passes: 8 SQ slots each) at the bench's default step count
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc_sq.err
python tools/pmc_summary.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_summary.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sq_b -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc_sq_b.err
python tools/pmc_summary.py gpurun_out/pmc_sq_b >> gpurun_out/pmc_sq_summary.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_sq_c -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc_sq_c.err || true
python tools/pmc_summary.py gpurun_out/pmc_sq_c >> gpurun_out/pmc_sq_summary.txt || true
cat gpurun_out/pmc_sq_summary.txt
