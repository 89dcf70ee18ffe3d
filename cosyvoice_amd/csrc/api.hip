// Library identity + struct-size probes (the ctypes mirrors in cosyvoice_amd/_lib.py are checked against these).
#include "cv_device.h"
extern "C" int cv_version(void) { return 1; }
extern "C" const char* cv_arch(void) { return "gfx950"; }
extern "C" int cv_sizeof_gemm_params(void) { return (int)sizeof(cv_gemm_params); }
extern "C" int cv_sizeof_norm_params(void) { return (int)sizeof(cv_norm_params); }
extern "C" int cv_sizeof_attn_params(void) { return (int)sizeof(cv_attn_params); }

// ---- hipGraph capture of an ABI launch sequence
extern "C" int cv_graph_begin(void* stream) {
  return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
extern "C" int cv_graph_end(void* stream, void** graph_exec_out) {
  if (!graph_exec_out) return CV_ERR_ARG;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return CV_ERR_LAUNCH;
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) return CV_ERR_LAUNCH;
  *graph_exec_out = (void*)ge;
  return CV_OK;
}
extern "C" int cv_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return CV_ERR_ARG;
  return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
extern "C" int cv_graph_destroy(void* graph_exec) {
  if (!graph_exec) return CV_ERR_ARG;
  return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
