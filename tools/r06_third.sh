#!/bin/bash
# round 6, third GPU call: GPU suite (release + the tunable build in its subprocess), default bench, SQ counters of the
# stand-alone selection kernel (two passes of 8 counters)
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
S=$SECONDS
timeout 1700 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1
echo "pytest: rc $? in $((SECONDS - S)) s"; tail -25 $out/pytest_gpu.log
S=$SECONDS
timeout 900 python bench.py --steps 20 --warmup 5 --detail-file $out/bench_steps20_detail.json > $out/bench_steps20.stdout 2> $out/bench_steps20.err
echo "driver-flag bench: rc $? in $((SECONDS - S)) s"; tail -1 $out/bench_steps20.stdout
{
echo "# token_select_kernel<8> at 32768 token slots (BASELINE [256,128] tokens), p = 0.95: launch duration and SQ counters"
python tools/select_loop.py --n 200 --lengths bench --events
python tools/select_loop.py --n 200 --lengths full --events
for L in bench full; do
echo; echo "## lengths = $L, pass 1 (occupancy / stall split)"
bash tools/pmc_kernel.sh sel1 token_select_kernel python tools/select_loop.py --n 100 --lengths $L
echo; echo "## lengths = $L, pass 2 (instruction mix, LDS)"
PMC_COUNTERS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" bash tools/pmc_kernel.sh sel2 token_select_kernel python tools/select_loop.py --n 100 --lengths $L
echo; echo "## lengths = $L, pass 3 (LDS array, scalar, barriers)"
PMC_COUNTERS="SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_WAVE32_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" bash tools/pmc_kernel.sh sel3 token_select_kernel python tools/select_loop.py --n 100 --lengths $L
done
} > $out/token_select_pmc.txt 2>&1
tail -60 $out/token_select_pmc.txt
