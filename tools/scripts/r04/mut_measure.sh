# Round 4's measurement script, tracked in round 5 as it was run then (profiles/r04_* name it).  Variant libraries (tools/bin/libhdiff_*.so:
# build products, not tracked) are built with tools/scripts/ab_build.sh today; knobs this script sets through the environment may have
# become compile-time -D switches of such a build since (tools/README.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "" mut1 mut2 mut4; do
  echo "=== lib: ${lib:-real}"
  if [ -n "$lib" ]; then export HDIFF_LIB=$PWD/tools/bin/libhdiff_$lib.so; else unset HDIFF_LIB; fi
  timeout -k 10 500 python3 -m pytest -q -s tests/test_gpu_configs.py::test_c2_unet_128_against_the_real_reference tests/test_gpu_configs.py::test_c2_sampler_steps_128_against_the_real_reference tests/test_gpu_end_to_end.py::test_c1_sampling_64x64_T50_matches_cpu_path tests/test_gpu_model.py::test_unet_default64_golden_seeded_weights tests/test_gpu_model.py::test_unet_small_forward_golden tests/test_gpu_model.py::test_sampler_small_teacher_forced_and_graph > gpurun_out/mm_${lib:-real}.log 2>&1
  grep -E "max err|per-step|C1 |passed|failed|Error" gpurun_out/mm_${lib:-real}.log | cut -c1-300
done
true
