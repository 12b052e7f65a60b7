#!/bin/bash
# Dev tool: produce the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag> <commit> [part]   -> gpurun_out/prof_<tag>/...   (copy what is to be judged into profiles/)
# part = traces | pmc1 | pmc2 (one gpurun call each: a call is limited to 20 minutes) | pmcbwd | traintraces | pmcfwd (subsets, after a change to one
# attention kernel alone) | summary (no GPU: run it where the merged
# gpurun_out/ is, i.e. in the build container) | all
# <commit> = `git rev-parse --short HEAD` of the tree that was sent (the box has no .git): it is stamped, with hashes of the
# kernel sources, into gpurun_out/prof_<tag>/roofline_traffic.json, the table bench.py reads for `roofline.traffic`.
# Counters are collected in their own passes (never together with --stats traces), as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=${1:-r}
COMMIT=${2:-unknown}
PART=${3:-all}
P=gpurun_out/prof_$TAG
mkdir -p $P
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
if want traces; then
# kernel-trace + stats of the headline (library default = bf16x3 contraction mode) and of the fp32 mode, and of one optimizer step
rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-extras --no-cpu-baseline > $P/bench_line.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench_f32 -o bench -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-extras --no-cpu-baseline --contract f32 > $P/bench_f32_line.json 2> $P/bench_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/train -o train -- python3 tools/bench_train.py --size 256 --batch 4 --steps 1 --warmup 1 > $P/train_line.json 2> $P/train.err
# BASELINE config C3 itself: one optimizer step at batch 64 (the program directly behind --)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/train64 -o train -- python3 tools/bench_train.py --size 256 --batch 64 --steps 1 --warmup 1 > $P/train64_line.json 2> $P/train64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/gn_trace -o gn -- python3 tools/gn_stats_once.py 16 128 256 > /dev/null 2>&1
python3 tools/gn_once.py 2>&1 | grep -v amdgpu.ids > $P/gn_once.txt
fi
run_pmc() {   # name, counters..., then -- command   (environment of the caller is inherited: HDIFF_CONTRACT=f32 run_pmc ...)
  local name=$1; shift
  local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $P/$name -o pmc -- "$@" > $P/$name.out 2> $P/$name.err \
    || echo "pmc pass $name failed or timed out (rc $?): see $P/$name.err" | tee -a $P/failed.txt
}
if want pmc1; then
run_pmc x3_sq1 $SQ1 -- python3 tools/attn_once.py 16
run_pmc x3_sq2 $SQ2 -- python3 tools/attn_once.py 16
HDIFF_CONTRACT=f32 run_pmc attn_sq1 $SQ1 -- python3 tools/attn_once.py 16
run_pmc bwd_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4
run_pmc bwd_sq2 $SQ2 -- python3 tools/attn_bwd_once.py 4
HDIFF_CONTRACT=f32 run_pmc bwdf32_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4
run_pmc bwd32_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4 256 16384
run_pmc bwd32_sq2 $SQ2 -- python3 tools/attn_bwd_once.py 4 256 16384
run_pmc fwd32_sq1 $SQ1 -- python3 tools/attn_once.py 16 256 16384
run_pmc fwd32_sq2 $SQ2 -- python3 tools/attn_once.py 16 256 16384
run_pmc convx3_sq1 $SQ1 -- python3 tools/conv_once.py 16 128 128 256 3 gn
run_pmc convh2_sq1 $SQ1 -- python3 tools/conv_once.py 16 128 128 256 3 pairs
run_pmc convh2_sq2 $SQ2 -- python3 tools/conv_once.py 16 128 128 256 3 pairs
HDIFF_CONTRACT=f32 run_pmc conv_sq1 $SQ1 -- python3 tools/conv_once.py 16 128 128 256 3 gn
fi
if want pmc2; then
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc x3_$c $c -- python3 tools/attn_once.py 16
  run_pmc convh2_$c $c -- python3 tools/conv_once.py 16 128 128 256 3 pairs
  HDIFF_CONTRACT=f32 run_pmc attn_$c $c -- python3 tools/attn_once.py 16
  HDIFF_CONTRACT=f32 run_pmc conv_$c $c -- python3 tools/conv_once.py 16 128 128 256 3 gn
  run_pmc gn_$c $c -- python3 tools/gn_stats_once.py 16 128 256
  run_pmc bwd_$c $c -- python3 tools/attn_bwd_once.py 4
  HDIFF_CONTRACT=f32 run_pmc bwdf32_$c $c -- python3 tools/attn_bwd_once.py 4
done
fi
if [ "$PART" = pmcbwd ]; then   # only the attention-backward passes of pmc1 / pmc2 (after a change to attention_bwd_h2.hip alone)
run_pmc bwd_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4
run_pmc bwd_sq2 $SQ2 -- python3 tools/attn_bwd_once.py 4
run_pmc bwd32_sq1 $SQ1 -- python3 tools/attn_bwd_once.py 4 256 16384
run_pmc bwd32_sq2 $SQ2 -- python3 tools/attn_bwd_once.py 4 256 16384
for c in FETCH_SIZE WRITE_SIZE; do run_pmc bwd_$c $c -- python3 tools/attn_bwd_once.py 4; done
fi
if [ "$PART" = pmcfwd ]; then   # only the d_head 16 forward's passes and the headline trace (after a change to attention_h2.hip alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-extras --no-cpu-baseline > $P/bench_line.json 2> $P/bench.err
run_pmc x3_sq1 $SQ1 -- python3 tools/attn_once.py 16
run_pmc x3_sq2 $SQ2 -- python3 tools/attn_once.py 16
for c in FETCH_SIZE WRITE_SIZE; do run_pmc x3_$c $c -- python3 tools/attn_once.py 16; done
fi
if [ "$PART" = traintraces ]; then   # only the two training traces of `traces`
rocprofv3 --kernel-trace --stats --output-format csv -d $P/train -o train -- python3 tools/bench_train.py --size 256 --batch 4 --steps 1 --warmup 1 > $P/train_line.json 2> $P/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/train64 -o train -- python3 tools/bench_train.py --size 256 --batch 64 --steps 1 --warmup 1 > $P/train64_line.json 2> $P/train64.err
fi
if want summary; then
{
  for d in x3_sq1 x3_sq2; do echo "## $d: rocprofv3 --pmc ... -- python3 tools/attn_once.py 16   (bf16x3 mode: split passes + mha_flash_fwd_h2_kernel<16, 4>; rocprofv3 prints that name mangled: its demangler does not know __bf16)"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv "mha_flash_fwd_h2_kernel"; done
  echo "## attn_sq1: HDIFF_CONTRACT=f32 ... -- python3 tools/attn_once.py 16"; python3 tools/pmc_summary.py $P/attn_sq1/pmc_counter_collection.csv "fast_kernel<16"
  for d in bwd_sq1 bwd_sq2; do echo "## $d: ... -- python3 tools/attn_bwd_once.py 4   (bf16x3 mode: maxima + split pass + mha_bwd_h2p_kernel (round 6: the pipelined d_head 16 kernel) + slab reduce)"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv mha_bwd_h2p_kernel; done
  for d in bwd32_sq1 bwd32_sq2; do echo "## $d: ... -- python3 tools/attn_bwd_once.py 4 256 16384   (d_head 32, L = 16384: mha_bwd_h2_kernel<32>)"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv mha_bwd_h2_kernel; done
  for d in fwd32_sq1 fwd32_sq2; do echo "## $d: ... -- python3 tools/attn_once.py 16 256 16384   (d_head 32 forward, L = 16384: mha_flash_fwd_x3p_kernel, fp16 pairs)"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv mha_flash_fwd_x3p; done
  echo "## bwdf32_sq1: HDIFF_CONTRACT=f32 ... -- python3 tools/attn_bwd_once.py 4"; python3 tools/pmc_summary.py $P/bwdf32_sq1/pmc_counter_collection.csv mha_bwd_fused
  echo "## convx3_sq1: ... -- python3 tools/conv_once.py 16 128 128 256 3 gn   (bf16 triples: round 3's kernel, still the one for 3x3 convs without a GroupNorm prologue)"; python3 tools/pmc_summary.py $P/convx3_sq1/pmc_counter_collection.csv conv3x3_x3
  for d in convh2_sq1 convh2_sq2; do echo "## $d: ... -- python3 tools/conv_once.py 16 128 128 256 3 pairs   (fp16 pairs: conv3x3_x3_kernel<9, false, true>)"; python3 tools/pmc_summary.py $P/$d/pmc_counter_collection.csv conv3x3_x3; done
  echo "## conv_sq1: HDIFF_CONTRACT=f32 ... -- python3 tools/conv_once.py 16 128 128 256 3 gn"; python3 tools/pmc_summary.py $P/conv_sq1/pmc_counter_collection.csv conv_igemm
  for k in x3 attn conv convh2 gn bwd bwdf32; do for c in FETCH_SIZE WRITE_SIZE; do echo "## ${k}_$c (KiB per dispatch)"; python3 tools/pmc_summary.py $P/${k}_$c/pmc_counter_collection.csv | grep -A1 -E "mha_flash_fwd_h2_kernel|qk_split_h2|qk_rowmax|v_split_h2|fast_kernel<16|igemm_kernel<2, 8, 12, 5, 1|conv3x3_x3|gn_stats_kernel|bwd_fused|bwd_h2|bwd_split|absmax|dq_reduce|delta"; done; done
} > $P/pmc_summary.txt
python3 tools/traffic_json.py $P $COMMIT --out $P/roofline_traffic.json > $P/traffic_line.json 2> $P/traffic.err
fi
ls $P
