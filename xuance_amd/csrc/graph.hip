// hipGraph capture of op sequences: the update loop (64 minibatches x ~10 launches) and the rollout loop
// (T steps x ~6 launches) are launch-bound at the reference's problem sizes, so the host side captures them once
// and replays them with one hipGraphLaunch.  Every xrl_* entry point is capture-safe (no allocation, no sync).
#include "common.h"

namespace xrl {
int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev] = v;
    }
    return cached[dev];
}
}  // namespace xrl

using namespace xrl;

extern "C" int xrl_graph_begin(xrl_stream_t stream) {
    XRL_CHECK_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return XRL_OK;
}

extern "C" int xrl_graph_end(xrl_stream_t stream, void** graph_exec_out) {
    XRL_CHECK_ARG(graph_exec_out != nullptr);
    hipGraph_t graph = nullptr;
    XRL_CHECK_HIP(hipStreamEndCapture(as_stream(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    XRL_CHECK_HIP(e);
    *graph_exec_out = exec;
    return XRL_OK;
}

extern "C" int xrl_graph_launch(void* graph_exec, xrl_stream_t stream) {
    XRL_CHECK_ARG(graph_exec != nullptr);
    XRL_CHECK_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), as_stream(stream)));
    return XRL_OK;
}

extern "C" int xrl_graph_destroy(void* graph_exec) {
    if (graph_exec) XRL_CHECK_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return XRL_OK;
}
