import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The precision contract: the two contraction modes -- fp32-input MFMA ("f32") and the three-piece
# bf16 split with fp32 accumulation ("bf16x3") -- must BOTH pass the whole golden / oracle / float64 suite of these modules
# at the SAME tolerances.  Tests that never reach a forward contraction kernel (backward-only kernels, elementwise ops)
# run once, and so do the tests that pick a mode themselves (the `bf16x3_mode` fixture of test_gpu_ops.py).  The library's
# default is bf16x3 (HDIFF_CONTRACT=f32 for the other); every fixture restores the mode it found.
CONTRACT_MODULES = {"test_gpu_model", "test_gpu_configs", "test_gpu_end_to_end", "test_gpu_tree_b", "test_gpu_ops",
                    "test_gpu_fullsize", "test_gpu_backward"}
CONTRACT_INDEPENDENT = {"test_small_ops_match_torch",
                        "test_linear_rows_and_gather", "test_linear_and_embedding_backward", "test_downsample_and_tconv_backward",
                        "test_ddpm_step_bit_exact_and_nan_flag", "test_ddpm_step_loop_bookkeeping", "test_q_sample_bit_exact_and_clip",
                        "test_randn_moments_and_determinism", "test_groupnorm_scale_shift", "test_groupnorm_fused_finalize_equals_two_launches",
                        "test_linear_rows_multi_equals_the_launches_it_replaces", "test_conv1x1_direct_gemm_path",
                        "test_conv1x1_and_5x5_stride2",
                        # these run both modes side by side themselves
                        "test_split_bf16_attention_backward_is_another_program_fp32_class_and_reproducible",
                        "test_attention_backward_fp16_pairs_error_class_every_pair", "test_attention_backward_fp16_pairs_ranges",
                        "test_attention_backward_slab_cap_setting",
                        "test_c1_sampling_64x64_T50_matches_cpu_path",                       # loops over both modes itself (one CPU oracle run)
                        "test_flash_attention_fp16_pairs_keeps_every_row_in_the_kernel"}     # a fresh process in the library's default mode
CONTRACT_MODES = ("f32", "bf16x3")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    if mod in CONTRACT_MODULES and metafunc.function.__name__ not in CONTRACT_INDEPENDENT \
            and "hdiff_contract" in metafunc.fixturenames and "bf16x3_mode" not in metafunc.fixturenames:
        metafunc.parametrize("hdiff_contract", CONTRACT_MODES, indirect=True)


@pytest.fixture(autouse=True)
def hdiff_contract(request):
    """Selects the contraction mode for one test (process-wide switch of libhdiff.so) and restores fp32 afterwards."""
    mode = getattr(request, "param", None)
    if mode is None:      # not parametrised: the test runs in whatever mode the library is in (default bf16x3) -- report THAT
        try:
            import hdiff_amd
            yield hdiff_amd.get_contraction_mode() if os.path.exists(hdiff_amd._capi.LIB_PATH) else None
        except Exception:
            yield None
        return
    import hdiff_amd
    before = hdiff_amd.get_contraction_mode()
    hdiff_amd.set_contraction_mode(mode)
    try:
        yield mode
    finally:
        hdiff_amd.set_contraction_mode(before)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
