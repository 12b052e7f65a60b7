#!/bin/bash
# Dev tool (GPU box, through gpurun): vector-L1 (TCP) counters of ONE kernel, collected the way round 4's hung pass was not --
#   bash tools/scripts/pmc_tcp.sh <tag> <kernel name filter> <python driver and args ...>
#   e.g.  bash tools/scripts/pmc_tcp.sh c1x1 conv1x1_x3 tools/conv_once.py 16 128 384 256 1
# Round 4's pass put EIGHT derived TCP_*_sum counters into one --pmc set (each a sum over the 16 TCP instances; the block has four
# counter registers per instance), with stdout / stderr discarded; rocprofv3 accepted the set and never came back (killed after
# 7 minutes, profiles/r04_conv1x1_x3.txt).  Here: sets of THREE, one rocprofv3 run per set, each under its own wall-clock timeout,
# everything logged under gpurun_out/prof_tcp_<tag>/, the program directly behind `--`, no trace domain besides --kernel-trace.
# A set that times out is reported and the remaining sets are NOT run (nothing is retried).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=$1; FILTER=$2; shift 2
P=gpurun_out/prof_tcp_$TAG
mkdir -p $P
SETS=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"
      "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum")
i=0
for set in "${SETS[@]}"; do
  i=$((i + 1))
  echo "== set $i: $set" | tee -a $P/log.txt
  timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/s$i -o pmc -- python3 "$@" > $P/s$i.out 2> $P/s$i.err
  rc=$?
  echo "   rc $rc" | tee -a $P/log.txt
  if [ $rc -ne 0 ]; then echo "   set $i failed or timed out: stopping (see $P/s$i.err)" | tee -a $P/log.txt; tail -5 $P/s$i.err | tee -a $P/log.txt; break; fi
  python3 tools/pmc_summary.py $P/s$i/pmc_counter_collection.csv "$FILTER" 2>&1 | tee -a $P/log.txt
done
