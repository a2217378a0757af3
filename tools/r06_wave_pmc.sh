mkdir -p gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L > /tmp/avail.txt 2>&1; grep -o "TCC_[A-Z0-9_]*RDREQ[A-Z0-9_]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|TCC_REQ[a-z_]*\|TCP_TCC_READ_REQ[a-z_]*\|TCP_TOTAL_CACHE_ACCESSES[a-z_]*\|FETCH_SIZE\|TCC_BUBBLE[a-z_]*\|TCC_EA0_RD_UNCACHED[a-z0-9_]*" /tmp/avail.txt | sort -u | tr '\n' ' ' > gpurun_out/r06b/pmc_avail.txt
PMC_COUNTERS="FETCH_SIZE" bash tools/pmc_kernel.sh a ordered_multi python tools/mse_round_probe.py 8 > gpurun_out/r06b/wave_pmc_fetch.txt 2>&1
PMC_COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" bash tools/pmc_kernel.sh b ordered_multi python tools/mse_round_probe.py 8 > gpurun_out/r06b/wave_pmc_tcc.txt 2>&1
bash tools/pmc_kernel.sh c ordered_multi python tools/mse_round_probe.py 8 > gpurun_out/r06b/wave_pmc_sq.txt 2>&1
cat gpurun_out/r06b/pmc_avail.txt; tail -n 12 gpurun_out/r06b/wave_pmc_fetch.txt gpurun_out/r06b/wave_pmc_tcc.txt gpurun_out/r06b/wave_pmc_sq.txt
