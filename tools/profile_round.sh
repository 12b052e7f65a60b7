#!/bin/bash
# Dev tool: produce the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...   (copy what is to be judged into profiles/)
# Counters are collected in their own passes (never together with --stats traces), as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=${1:-r}
P=gpurun_out/prof_$TAG
mkdir -p $P
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-alt > $P/bench_line.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/train -o train -- python3 tools/bench_train.py --size 256 --batch 4 --steps 1 --warmup 1 > $P/train_line.json 2> $P/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/gn_trace -o gn -- python3 tools/gn_stats_once.py 16 128 256 > /dev/null 2>&1
python3 tools/gn_once.py 2>&1 | grep -v amdgpu.ids > $P/gn_once.txt
run_pmc() {   # name, counters..., then -- command
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $P/$name -o pmc -- "$@" > /dev/null 2>&1
}
run_pmc bwd_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4
run_pmc bwd_sq2 $SQ2 -- python3 tools/attn_bwd_once.py 4
run_pmc wg3_sq1 $SQ1 -- python3 tools/wgrad_once.py 4 128 128 256 gn
run_pmc wg3_sq2 $SQ2 -- python3 tools/wgrad_once.py 4 128 128 256 gn
HDIFF_WGRAD_GENERIC=1 run_pmc wgG_sq1 $SQ1 -- python3 tools/wgrad_once.py 4 128 128 256 gn
HDIFF_WGRAD_GENERIC=1 run_pmc wgG_sq2 $SQ2 -- python3 tools/wgrad_once.py 4 128 128 256 gn
run_pmc conv_sq1 $SQ1 -- python3 tools/conv_once.py 16 128 128 256 3 gn
run_pmc conv_sq2 $SQ2 -- python3 tools/conv_once.py 16 128 128 256 3 gn
run_pmc attn_sq1 $SQ1 -- python3 tools/attn_once.py 16
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc attn_$c $c -- python3 tools/attn_once.py 16
  run_pmc conv_$c $c -- python3 tools/conv_once.py 16 128 128 256 3 gn
  run_pmc gn_$c $c -- python3 tools/gn_stats_once.py 16 128 256
  run_pmc bwd_$c $c -- python3 tools/attn_bwd_once.py 4
done
{
  for d in bwd_sq1 bwd_sq2; do echo "## $d: rocprofv3 --pmc ... -- python3 tools/attn_bwd_once.py 4"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv mha_bwd_fused; done
  for d in wg3_sq1 wg3_sq2; do echo "## $d: ... -- python3 tools/wgrad_once.py 4 128 128 256 gn"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv conv_wgrad3x3; done
  for d in wgG_sq1 wgG_sq2; do echo "## $d: HDIFF_WGRAD_GENERIC=1 ... -- python3 tools/wgrad_once.py 4 128 128 256 gn"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv "conv_wgrad_kernel<1>"; done
  for d in conv_sq1 conv_sq2; do echo "## $d: ... -- python3 tools/conv_once.py 16 128 128 256 3 gn"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv conv_igemm; done
  echo "## attn_sq1: ... -- python3 tools/attn_once.py 16"; python3 tools/pmc_summary.py $P/attn_sq1/pmc_counter_collection.csv "fast_kernel<16"
  for k in attn conv gn bwd; do for c in FETCH_SIZE WRITE_SIZE; do echo "## ${k}_$c (KiB per dispatch)"; python3 tools/pmc_summary.py $P/${k}_$c/pmc_counter_collection.csv | grep -A1 -E "fast_kernel<16|igemm_kernel<2, 8, 12, 5, 1|gn_stats_kernel|bwd_fused|dq_reduce|delta"; done; done
} > $P/pmc_summary.txt
ls $P
