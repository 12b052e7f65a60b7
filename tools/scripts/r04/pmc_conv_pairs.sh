# HISTORICAL (round 4).  Kept as the record behind profiles/r04_*; the current form of these passes is tools/profile_round.sh / tools/scripts/pmc_tcp.sh.
# Since round 6 every profiler pass here runs under `timeout -k 10` and logs to <pass dir>.out / .err, as those do (a pass that hangs leaves a record).
# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=gpurun_out/prof_convh2
mkdir -p $P
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
for m in pairs gn; do
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $P/${m}_sq1 -o pmc -- python3 tools/conv_once.py 16 128 128 256 3 $m > $P/${m}_sq1.out 2> $P/${m}_sq1.err || echo "pmc pass $P/${m}_sq1 failed or timed out (rc $?): see $P/${m}_sq1.err"
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $P/${m}_sq2 -o pmc -- python3 tools/conv_once.py 16 128 128 256 3 $m > $P/${m}_sq2.out 2> $P/${m}_sq2.err || echo "pmc pass $P/${m}_sq2 failed or timed out (rc $?): see $P/${m}_sq2.err"
done
for m in pairs gn; do for d in sq1 sq2; do echo "## $m $d: rocprofv3 --pmc ... -- python3 tools/conv_once.py 16 128 128 256 3 $m"; python3 tools/pmc_summary.py $P/${m}_$d/pmc_counter_collection.csv conv3x3_x3; done; done
