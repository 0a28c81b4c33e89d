// libpam_hip.so: launch plans -- the HRNet forward as a recorded DAG of kernel launches, replayed from C.
//
// Why not a captured hipGraph: the dependency-precise schedule of the conv stack (every consumer stream waits for exactly the tensors it
// reads; pam/hrnet_hip.py) has streams that wait on each other's events in both directions over a module, and capturing that pattern
// with three or more streams segfaults in hipStreamEndCapture on ROCm 7.2 (tools/dag_probe.py), although the same calls run correctly
// eagerly.  A plan stores what the executor would have issued -- launches (function, geometry, a copy of the argument struct) on logical
// streams, event records and waits -- and pam_plan_replay issues it on real streams with real events, in recorded order, at C speed
// (~1.5-3 us per call instead of ~15 us through ctypes).  Mode 1 builds ONE explicit hipGraph from the same plan
// (hipGraphAddKernelNode with the recorded dependencies) and replays that.
// Pointers inside the argument structs are baked in: the caller keeps every buffer of the recorded forward alive and in place.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "pam_launch.hpp"
#include "../../include/pam.h"

namespace {
enum { OP_LAUNCH = 0, OP_RECORD = 1, OP_WAIT = 2 };
struct PlanOp {
    int kind, stream, event;                 // stream: logical (0 = the replay's caller stream), event: id for record / wait
    const void* func; dim3 grid, block; size_t lds; size_t arg_off, arg_bytes;
};
struct Plan {
    std::vector<PlanOp> ops;
    std::vector<char> args;
    int n_streams = 1, n_events = 0, n_launch = 0, cur_stream = 0;
    int device = 0;
    std::vector<hipStream_t> side;           // real streams of logical streams 1 .. n_streams - 1
    std::vector<hipEvent_t> ev;              // one per recorded event
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    std::string err;
};
thread_local Plan* g_rec = nullptr;
}  // namespace

bool pam_plan_recording() { return g_rec != nullptr; }
void pam_plan_add_launch(const void* func, dim3 grid, dim3 block, size_t lds, const void* arg, size_t arg_bytes) {
    Plan* p = g_rec;
    PlanOp o{};
    o.kind = OP_LAUNCH; o.stream = p->cur_stream; o.event = -1; o.func = func; o.grid = grid; o.block = block; o.lds = lds;
    o.arg_off = (p->args.size() + 15) & ~(size_t)15; o.arg_bytes = arg_bytes;
    p->args.resize(o.arg_off + arg_bytes);
    memcpy(p->args.data() + o.arg_off, arg, arg_bytes);
    p->ops.push_back(o);
    p->n_launch += 1;
}

extern "C" int pam_plan_begin(void) {
    if (g_rec) return PAM_E_STATE;
    g_rec = new Plan();
    if (hipGetDevice(&g_rec->device) != hipSuccess) { delete g_rec; g_rec = nullptr; return PAM_E_HIP; }
    return PAM_OK;
}
extern "C" int pam_plan_stream(int idx) {
    if (!g_rec || idx < 0 || idx >= 8) return PAM_E_ARG;
    g_rec->cur_stream = idx;
    if (idx + 1 > g_rec->n_streams) g_rec->n_streams = idx + 1;
    return PAM_OK;
}
extern "C" int pam_plan_record(void) {                   // -> event id (>= 0)
    if (!g_rec) return PAM_E_STATE;
    PlanOp o{}; o.kind = OP_RECORD; o.stream = g_rec->cur_stream; o.event = g_rec->n_events++;
    g_rec->ops.push_back(o);
    return o.event;
}
extern "C" int pam_plan_wait(int event_id) {
    if (!g_rec || event_id < 0 || event_id >= g_rec->n_events) return PAM_E_ARG;
    PlanOp o{}; o.kind = OP_WAIT; o.stream = g_rec->cur_stream; o.event = event_id;
    g_rec->ops.push_back(o);
    return PAM_OK;
}
extern "C" int pam_plan_abort(void) {
    delete g_rec; g_rec = nullptr;
    return PAM_OK;
}
extern "C" int pam_plan_end(void** out) {
    if (!g_rec || !out) return PAM_E_ARG;
    Plan* p = g_rec; g_rec = nullptr;
    bool ok = true;
    p->side.resize(p->n_streams, nullptr);
    for (int i = 1; i < p->n_streams && ok; ++i) ok = hipStreamCreateWithFlags(&p->side[i], hipStreamNonBlocking) == hipSuccess;
    p->ev.resize(p->n_events, nullptr);
    for (int i = 0; i < p->n_events && ok; ++i) ok = hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming) == hipSuccess;
    p->ev_join.resize(p->n_streams, nullptr);
    for (int i = 1; i < p->n_streams && ok; ++i) ok = hipEventCreateWithFlags(&p->ev_join[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete p; return PAM_E_HIP; }
    *out = p;
    return PAM_OK;
}
extern "C" int pam_plan_info(const void* plan, int32_t* out4) {
    const Plan* p = (const Plan*)plan;
    if (!p || !out4) return PAM_E_ARG;
    out4[0] = p->n_launch; out4[1] = p->n_events; out4[2] = p->n_streams; out4[3] = (int)p->ops.size();
    return PAM_OK;
}
extern "C" const char* pam_plan_last_error(const void* plan) { return plan ? ((const Plan*)plan)->err.c_str() : ""; }

#define PCHK(p, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (p)->err = std::string(#call) + ": " + hipGetErrorString(e_); return PAM_E_HIP; } } while (0)

static int replay_eager(Plan* p, hipStream_t caller) {
    auto real = [&](int s) { return s == 0 ? caller : p->side[s]; };
    if (p->n_streams > 1) {
        PCHK(p, hipEventRecord(p->ev_fork, caller));
        for (int i = 1; i < p->n_streams; ++i) PCHK(p, hipStreamWaitEvent(p->side[i], p->ev_fork, 0));
    }
    char* base = p->args.data();
    for (const PlanOp& o : p->ops) {
        if (o.kind == OP_LAUNCH) {
            void* kargs[1] = {base + o.arg_off};
            PCHK(p, hipLaunchKernel(o.func, o.grid, o.block, kargs, o.lds, real(o.stream)));
        } else if (o.kind == OP_RECORD) {
            PCHK(p, hipEventRecord(p->ev[o.event], real(o.stream)));
        } else {
            PCHK(p, hipStreamWaitEvent(real(o.stream), p->ev[o.event], 0));
        }
    }
    for (int i = 1; i < p->n_streams; ++i) {
        PCHK(p, hipEventRecord(p->ev_join[i], p->side[i]));
        PCHK(p, hipStreamWaitEvent(caller, p->ev_join[i], 0));
    }
    return PAM_OK;
}

// One explicit hipGraph from the plan: a launch depends on the previous launch of its logical stream and on whatever the events its stream
// waited for since then stood for (an event = the set of nodes its stream's position depended on when it was recorded).
static int build_graph(Plan* p) {
    PCHK(p, hipGraphCreate(&p->graph, 0));
    std::vector<std::vector<hipGraphNode_t>> front(p->n_streams);       // what the NEXT op of a stream depends on
    std::vector<std::vector<hipGraphNode_t>> evdep(p->n_events);
    char* base = p->args.data();
    auto add_unique = [](std::vector<hipGraphNode_t>& v, hipGraphNode_t n) { for (auto q : v) if (q == n) return; v.push_back(n); };
    for (const PlanOp& o : p->ops) {
        if (o.kind == OP_LAUNCH) {
            hipKernelNodeParams kp{};
            void* kargs[1] = {base + o.arg_off};
            kp.func = (void*)o.func; kp.gridDim = o.grid; kp.blockDim = o.block; kp.sharedMemBytes = (unsigned)o.lds; kp.kernelParams = kargs; kp.extra = nullptr;
            hipGraphNode_t n;
            PCHK(p, hipGraphAddKernelNode(&n, p->graph, front[o.stream].data(), front[o.stream].size(), &kp));
            front[o.stream].assign(1, n);
        } else if (o.kind == OP_RECORD) {
            evdep[o.event] = front[o.stream];
        } else {
            for (auto n : evdep[o.event]) add_unique(front[o.stream], n);
        }
    }
    PCHK(p, hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    return PAM_OK;
}

extern "C" int pam_plan_replay(void* plan, void* stream, int mode) {
    Plan* p = (Plan*)plan;
    if (!p || g_rec) return PAM_E_ARG;
    if (mode == 0) return replay_eager(p, (hipStream_t)stream);
    if (mode == 1) {
        if (!p->exec) { int rc = build_graph(p); if (rc) return rc; }
        PCHK(p, hipGraphLaunch(p->exec, (hipStream_t)stream));
        return PAM_OK;
    }
    return PAM_E_ARG;
}
extern "C" int pam_plan_destroy(void* plan) {
    Plan* p = (Plan*)plan;
    if (!p) return PAM_OK;
    hipSetDevice(p->device);
    // the explicit graph is NOT destroyed: destroying a multi-queue hipGraph corrupts the ROCm 7.2 runtime's heap now and then
    // (tools/graph_destroy_stress.py); what leaks is the executable graph of a plan that was replayed in mode 1
    p->exec = nullptr; p->graph = nullptr;
    for (auto s : p->side) if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    for (auto e : p->ev) if (e) hipEventDestroy(e);
    for (auto e : p->ev_join) if (e) hipEventDestroy(e);
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    delete p;
    return PAM_OK;
}
