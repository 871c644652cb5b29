mkdir -p gpurun_out/prof3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-frames 0 --frames 12 --steps 1 --warmup 0 --no-profile-events"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/prof3/$name -- $B > gpurun_out/prof3/$name.json 2> gpurun_out/prof3/$name.err || echo "FAILED $name"; }
run a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum &&
run b TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum &&
run c TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_EA0_WRREQ_LEVEL_sum &&
run d TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
echo done
