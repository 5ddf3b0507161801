# L1 / L2 / fabric counters of the hot kernels (separate --pmc passes, kernel counters only) at the bench's default step count
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_cache_summary.txt
: > $OUT
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_cache_$i -- python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/pmc_cache_$i.err || { echo "pass $i failed" >> $OUT; tail -3 gpurun_out/pmc_cache_$i.err >> $OUT; continue; }
  python tools/pmc_summary.py gpurun_out/pmc_cache_$i >> $OUT || true
done
cat $OUT
