// Library identity + struct-size probes (the ctypes mirrors in cosyvoice_amd/_lib.py are checked against these).
#include "cv_device.h"
extern "C" int cv_version(void) { return 1; }
extern "C" const char* cv_arch(void) { return "gfx950"; }
extern "C" int cv_sizeof_gemm_params(void) { return (int)sizeof(cv_gemm_params); }
extern "C" int cv_sizeof_norm_params(void) { return (int)sizeof(cv_norm_params); }
extern "C" int cv_sizeof_attn_params(void) { return (int)sizeof(cv_attn_params); }

// ---- hipGraph capture of an ABI launch sequence.  A handle owns the captured graph, its executable and the launch list read
// back from the kernel nodes: cv_graph_launch replays the hipGraphExec (cheapest on the host), cv_graph_launch_direct issues
// the same launches one by one with hipLaunchKernel.  The second form exists because hipGraph replays ignore the CU mask of
// the stream they are launched into (probed on gfx950 / ROCm 7.2, tools/cumask_probe.py) while direct launches honour it:
// it is what lets the latency-bound decode loop and the throughput-bound flow solver own disjoint CU sets.
#include <vector>
#include <algorithm>

namespace {
struct cv_launch_node {
  int kind;  // 0 kernel, 1 memset, 2 memcpy
  hipKernelNodeParams k;
  hipMemsetParams ms;
  hipMemcpy3DParms mc;
  int module_launch;  // kernel node whose func is a hipFunction_t
};
struct cv_graph_handle {
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  std::vector<cv_launch_node> list;
  bool direct_ok = false;
};

// insertion-ordered topological sort of the captured nodes (a single-stream capture is already a chain)
bool build_launch_list(cv_graph_handle* h) {
  size_t n = 0;
  if (hipGraphGetNodes(h->g, nullptr, &n) != hipSuccess) return false;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(h->g, nodes.data(), &n) != hipSuccess) return false;
  std::vector<std::vector<size_t>> deps(n);
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) return false;
    std::vector<hipGraphNode_t> d(nd);
    if (nd && hipGraphNodeGetDependencies(nodes[i], d.data(), &nd) != hipSuccess) return false;
    for (size_t j = 0; j < nd; ++j) {
      const size_t idx = (size_t)(std::find(nodes.begin(), nodes.end(), d[j]) - nodes.begin());
      if (idx >= n) return false;
      deps[i].push_back(idx);
    }
  }
  std::vector<char> done(n, 0);
  std::vector<size_t> order;
  while (order.size() < n) {
    bool progressed = false;
    for (size_t i = 0; i < n; ++i) {
      if (done[i]) continue;
      bool ready = true;
      for (size_t d : deps[i]) ready = ready && done[d];
      if (!ready) continue;
      done[i] = 1;
      order.push_back(i);
      progressed = true;
    }
    if (!progressed) return false;
  }
  for (size_t i : order) {
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) return false;
    cv_launch_node ln{};
    if (ty == hipGraphNodeTypeKernel) {
      ln.kind = 0;
      if (hipGraphKernelNodeGetParams(nodes[i], &ln.k) != hipSuccess) return false;
      if (ln.k.extra) return false;  // packed-argument launches are not replayed here
    } else if (ty == hipGraphNodeTypeMemset) {
      ln.kind = 1;
      if (hipGraphMemsetNodeGetParams(nodes[i], &ln.ms) != hipSuccess) return false;
      if (ln.ms.elementSize != 1 && ln.ms.height > 1) return false;
    } else if (ty == hipGraphNodeTypeMemcpy) {
      ln.kind = 2;
      if (hipGraphMemcpyNodeGetParams(nodes[i], &ln.mc) != hipSuccess) return false;
    } else if (ty == hipGraphNodeTypeEmpty) {
      continue;
    } else {
      return false;
    }
    h->list.push_back(ln);
  }
  return true;
}
}  // namespace

extern "C" int cv_graph_begin(void* stream) {
  return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
extern "C" int cv_graph_end(void* stream, void** graph_out) {
  if (!graph_out) return CV_ERR_ARG;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return CV_ERR_LAUNCH;
  cv_graph_handle* h = new cv_graph_handle();
  h->g = g;
  if (hipGraphInstantiate(&h->ge, g, nullptr, nullptr, 0) != hipSuccess) {
    hipGraphDestroy(g);
    delete h;
    return CV_ERR_LAUNCH;
  }
  h->direct_ok = build_launch_list(h);
  *graph_out = (void*)h;
  return CV_OK;
}
extern "C" int cv_graph_launch(void* graph, void* stream) {
  if (!graph) return CV_ERR_ARG;
  return hipGraphLaunch(((cv_graph_handle*)graph)->ge, (hipStream_t)stream) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
extern "C" int cv_graph_num_launches(void* graph) {
  if (!graph) return CV_ERR_ARG;
  cv_graph_handle* h = (cv_graph_handle*)graph;
  return h->direct_ok ? (int)h->list.size() : CV_ERR_UNSUPPORTED;
}
extern "C" int cv_graph_launch_direct(void* graph, void* stream) {
  if (!graph) return CV_ERR_ARG;
  cv_graph_handle* h = (cv_graph_handle*)graph;
  if (!h->direct_ok) return CV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  for (cv_launch_node& ln : h->list) {
    hipError_t e = hipSuccess;
    if (ln.kind == 0) {
      if (!ln.module_launch) {
        e = hipLaunchKernel(ln.k.func, ln.k.gridDim, ln.k.blockDim, ln.k.kernelParams, ln.k.sharedMemBytes, st);
        if (e == hipErrorInvalidDeviceFunction) {
          (void)hipGetLastError();
          ln.module_launch = 1;
        }
      }
      if (ln.module_launch)
        e = hipModuleLaunchKernel((hipFunction_t)ln.k.func, ln.k.gridDim.x, ln.k.gridDim.y, ln.k.gridDim.z, ln.k.blockDim.x,
                                  ln.k.blockDim.y, ln.k.blockDim.z, ln.k.sharedMemBytes, st, ln.k.kernelParams, nullptr);
    } else if (ln.kind == 1) {
      const size_t rows = ln.ms.height ? ln.ms.height : 1;
      if (rows == 1 && ln.ms.elementSize == 4) e = hipMemsetD32Async((hipDeviceptr_t)ln.ms.dst, (int)ln.ms.value, ln.ms.width, st);
      else if (rows == 1 && ln.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)ln.ms.dst, (unsigned short)ln.ms.value, ln.ms.width, st);
      else e = hipMemset2DAsync(ln.ms.dst, ln.ms.pitch, (int)ln.ms.value, ln.ms.width, rows, st);  // elementSize 1 (checked at capture)
    } else {
      e = hipMemcpy3DAsync(&ln.mc, st);
    }
    if (e != hipSuccess) return CV_ERR_LAUNCH;
  }
  return CV_OK;
}
extern "C" int cv_graph_destroy(void* graph) {
  if (!graph) return CV_ERR_ARG;
  cv_graph_handle* h = (cv_graph_handle*)graph;
  hipGraphExecDestroy(h->ge);
  hipGraphDestroy(h->g);
  delete h;
  return CV_OK;
}

// ---- CU-masked streams.  mask bit i selects CU slot i / n_xcd of XCD i % n_xcd (KFD's symmetric mapping, confirmed by timing
// on MI355X: tools/cumask_probe2.py); an XCD left with no CU at all falls back to all of its CUs.
extern "C" int cv_stream_create_cumask(const uint32_t* mask, int32_t nwords, void** stream_out) {
  if (!mask || nwords <= 0 || !stream_out) return CV_ERR_ARG;
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask) != hipSuccess) return CV_ERR_LAUNCH;
  *stream_out = (void*)st;
  return CV_OK;
}
extern "C" int cv_stream_destroy(void* stream) {
  if (!stream) return CV_ERR_ARG;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
