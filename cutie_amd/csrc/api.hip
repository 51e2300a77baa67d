// C-ABI entry points of libcutie_hip.so (see include/cutie_hip.h): launch-plan executor, HIP-graph capture
// and an on-stream hipEvent timer.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void cutie_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int dispatch(const cutie_op* op, hipStream_t s) {
    switch (op->kind) {
        case CUTIE_OP_CONV: return launch_conv(op, s);
        case CUTIE_OP_QUERY_INIT: return (op->flags & 1) ? launch_attention(op, s) : launch_elementwise(op, s);
        case CUTIE_OP_QFFN: return launch_qchain(op, s);
        case CUTIE_OP_STEM: return launch_stem(op, s);
        case CUTIE_OP_ATTN_Q2P: case CUTIE_OP_ATTN_SELF: case CUTIE_OP_ATTN_P2Q:
            return (op->flags & 12) ? launch_qchain(op, s) : launch_attention(op, s);
        case CUTIE_OP_AUX_MASK:
            return launch_attention(op, s);
        case CUTIE_OP_KEY_PREP: case CUTIE_OP_AFF_SCORE: case CUTIE_OP_AFF_SELECT: case CUTIE_OP_AFF_READOUT:
            return launch_affinity(op, s);
        case CUTIE_OP_RANK_SELECT: case CUTIE_OP_GATHER_ROWS: case CUTIE_OP_CONSOL_AFF: case CUTIE_OP_CONSOL_READ:
            return launch_bank(op, s);
        default:
            if (op->kind > 0 && op->kind < CUTIE_OP__COUNT) return launch_elementwise(op, s);
            cutie_set_error("cutie_exec: unknown op kind %d", op->kind);
            return -3;
    }
}

extern "C" {

int cutie_exec(const cutie_op* ops, int n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < n; ++k) {
        int rc = dispatch(&ops[k], s);
        if (rc != 0) {
            if (rc > 0) cutie_set_error("op %d (kind %d): HIP error %d (%s)", k, ops[k].kind, rc, hipGetErrorString((hipError_t)rc));
            else {
                char tmp[400];
                snprintf(tmp, sizeof(tmp), "%s", g_err);
                cutie_set_error("op %d (kind %d): %s", k, ops[k].kind, tmp);
            }
            return rc > 0 ? -1 : rc;
        }
    }
    return 0;
}

int cutie_exec_one(const cutie_op* op, void* stream) { return cutie_exec(op, 1, stream); }

struct GraphHandle { hipGraph_t graph; hipGraphExec_t exec; };

void* cutie_graph_capture(const cutie_op* ops, int n, void* stream) {
    // Captured on a private stream of the calling thread (capture records, it does not execute): the caller's stream may be the legacy
    // default stream, which cannot be captured, and must not change state.  The instantiated graph is launched on any stream.
    (void)stream;
    static thread_local hipStream_t cap = nullptr;
    if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) {
        cap = nullptr;
        (void)hipGetLastError();
        cutie_set_error("graph: cannot create the capture stream");
        return nullptr;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        cutie_set_error("graph: begin capture failed");
        return nullptr;
    }
    int rc = cutie_exec(ops, n, (void*)cap);
    hipError_t e = hipStreamEndCapture(cap, &graph);
    if (rc != 0 || e != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        if (rc == 0) cutie_set_error("graph: end capture failed (%d)", (int)e);
        (void)hipGetLastError();                             // the failed capture must not leave a sticky error for the next launch check
        return nullptr;
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        hipGraphDestroy(graph);
        (void)hipGetLastError();
        cutie_set_error("graph: instantiate failed");
        return nullptr;
    }
    GraphHandle* h = new GraphHandle{graph, exec};
    return h;
}

int cutie_graph_launch(void* graph, void* stream) {
    GraphHandle* h = (GraphHandle*)graph;
    if (!h) { cutie_set_error("graph: null handle"); return -4; }
    hipError_t e = hipGraphLaunch(h->exec, (hipStream_t)stream);
    if (e != hipSuccess) { cutie_set_error("graph launch: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

void cutie_graph_destroy(void* graph) {
    GraphHandle* h = (GraphHandle*)graph;
    if (!h) return;
    hipGraphExecDestroy(h->exec);
    hipGraphDestroy(h->graph);
    delete h;
}

float cutie_time_ops(const cutie_op* ops, int n, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { cutie_set_error("time: event create failed"); return -1.f; }
    if (cutie_exec(ops, n, stream) != 0) return -1.f;              // warm-up
    hipEventRecord(a, s);
    for (int it = 0; it < iters; ++it)
        if (cutie_exec(ops, n, stream) != 0) return -1.f;
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms / (float)iters;
}

const char* cutie_hip_last_error(void) { return g_err; }
int cutie_hip_abi_version(void) { return 4; }
int cutie_op_struct_size(void) { return (int)sizeof(cutie_op); }
int cutie_hip_build_flags(void) {
#ifdef CUTIE_DIAG
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
