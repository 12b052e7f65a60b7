# HISTORICAL (round 4).  Kept as the record behind profiles/r04_*; the current form of these passes is tools/profile_round.sh / tools/scripts/pmc_tcp.sh.
# Since round 6 every profiler pass here runs under `timeout -k 10` and logs to <pass dir>.out / .err, as those do (a pass that hangs leaves a record).
# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=gpurun_out/prof_c1x3
mkdir -p $P
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
SQ3="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM"
TCP="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum"
i=0
# THE FOURTH SET -- eight derived TCP counters in ONE --pmc set -- is the pass that never returned in round 4 (profiles/r05_conv1x1_tcp.txt):
# it is left out of the loop here; tools/scripts/pmc_tcp.sh collects those counters in sets of three under per-set timeouts.
for set in "$SQ1" "$SQ2" "$SQ3"; do i=$((i+1)); timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/s$i -o pmc -- python3 tools/conv_once.py 16 128 384 256 1 > $P/s$i.out 2> $P/s$i.err || echo "pmc pass $P/s$i failed or timed out (rc $?): see $P/s$i.err"; python3 tools/pmc_summary.py $P/s$i/pmc_counter_collection.csv conv1x1_x3 2>&1; done
