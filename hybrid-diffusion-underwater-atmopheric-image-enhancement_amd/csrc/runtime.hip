// Error reporting, device query, hipGraph and event helpers of the C ABI (include/hdiff.h).
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace hdiff {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static int g_contract = -1;
int contraction_mode() {
  if (g_contract < 0) {
    // Default: the three-piece bf16 split (fp32-class error, pinned by the whole golden suite at the fp32 tolerances:
    // tests/conftest.py runs it in both modes).  HDIFF_CONTRACT=f32 selects the fp32-input MFMA kernels.
    const char* e = getenv("HDIFF_CONTRACT");
    g_contract = (e && strcmp(e, "f32") == 0) ? HDIFF_CONTRACT_F32 : HDIFF_CONTRACT_BF16X3;
  }
  return g_contract;
}
}  // namespace hdiff

extern "C" {

int hdiff_abi_version(void) { return 6; }   // 6: hdiff_opt_chunk, hdiff_grad_norm_clip_coef, hdiff_adamw_step (the optimizer tail); 3: hdiff_q_sample takes the schedule length, hdiff_mha_flash_bwd a workspace; 4: hdiff_mha_flash_fwd_ws, hdiff_ddpm_step takes the schedule length; 5: hdiff_conv_desc.wp_h2 / act_scale, hdiff_pack_conv_weight_h2, hdiff_gn_act_scale
const char* hdiff_last_error(void) { return hdiff::g_err; }

int hdiff_set_contraction_mode(int mode) {
  HDIFF_CHECK_ARG(mode == HDIFF_CONTRACT_F32 || mode == HDIFF_CONTRACT_BF16X3, "set_contraction_mode: unknown mode %d", mode);
  hdiff::g_contract = mode;
  return HDIFF_OK;
}
int hdiff_get_contraction_mode(void) { return hdiff::contraction_mode(); }

int hdiff_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int hdiff_graph_begin(hdiff_stream_t stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    hdiff::set_error("hipStreamBeginCapture: %s", hipGetErrorString(e));
    return HDIFF_ERR_LAUNCH;
  }
  return HDIFF_OK;
}

int hdiff_graph_end(hdiff_stream_t stream, void** graph_exec_out) {
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    hdiff::set_error("hipStreamEndCapture: %s", hipGetErrorString(e));
    return HDIFF_ERR_LAUNCH;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) {
    hdiff::set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return HDIFF_ERR_LAUNCH;
  }
  *graph_exec_out = (void*)exec;
  return HDIFF_OK;
}

int hdiff_graph_launch(void* graph_exec, hdiff_stream_t stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    hdiff::set_error("hipGraphLaunch: %s", hipGetErrorString(e));
    return HDIFF_ERR_LAUNCH;
  }
  return HDIFF_OK;
}

int hdiff_graph_destroy(void* graph_exec) {
  if (graph_exec) hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return HDIFF_OK;
}

int hdiff_event_create(void** ev) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) {
    hdiff::set_error("hipEventCreate failed");
    return HDIFF_ERR_LAUNCH;
  }
  *ev = (void*)e;
  return HDIFF_OK;
}
int hdiff_event_record(void* ev, hdiff_stream_t stream) {
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) {
    hdiff::set_error("hipEventRecord failed");
    return HDIFF_ERR_LAUNCH;
  }
  return HDIFF_OK;
}
int hdiff_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess ||
      hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    hdiff::set_error("hipEventElapsedTime failed");
    return HDIFF_ERR_LAUNCH;
  }
  return HDIFF_OK;
}
int hdiff_event_destroy(void* ev) {
  if (ev) hipEventDestroy((hipEvent_t)ev);
  return HDIFF_OK;
}

}  // extern "C"
