// Multi-stream launch replayer: a captured hipGraph re-issued as ordinary asynchronous launches over several
// HIP streams, with events only where a dependency crosses streams.
//
// Why: on ROCm 7.2 a replayed hipGraph runs its branches almost back to back (one queue, a barrier packet at
// every join) and graph launches on different streams do not overlap each other, while eager launches from
// Python overlap fine but cost ~20 us of host time each.  The training step (trainer.py:245-299 of the
// reference) is ~950 short launches with wide branch parallelism -- three discriminator updates, the four
// branches of the generator loss, the branches of every Inception block, weight gradients beside data
// gradients -- so the launches are recorded ONCE (stream capture -> hipGraph, which also records torch's own
// kernels, copies and fills), the graph is walked here (nodes, parameters, edges), every node is given a stream
// (a chain keeps its stream, a fork takes the next one) and each replay re-issues the nodes in topological order
// with hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync: ~3 us of host time per launch, real concurrency.
// The first chain runs on the CALLER's stream; only forks get streams of the replayer's own.
//
// The graph object must outlive the replayer (kernel argument buffers belong to the graph's nodes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <queue>
#include <vector>

#include "sbagan_hip.h"

namespace {

struct RNode {
    hipGraphNodeType type;
    hipKernelNodeParams kp;
    hipFunction_t hfunc;          // non-null: launch through hipModuleLaunchKernel
    hipMemcpy3DParms cp;
    hipMemsetParams ms;
    int stream;                   // index into Replayer::streams
    int record;                   // event index recorded after this node, or -1
    std::vector<int> waits;       // event indices this node's stream waits for first
    bool sink;                    // no successors: the caller's stream must wait for it
    int marker;                   // >= 0: a host-call node (sba_replay_marker): its tag; the callback runs instead of a launch
};

// A marker: a no-op kernel whose captured node stands for "call the host here".  Stream capture records it like any
// launch (with the dependencies of its stream position); a replay calls the registered callback with the node's tag and
// the stream the node was assigned to -- the host then issues whatever belongs at that point of the step and cannot be
// recorded: an RCCL collective on the exchange stream (torch.distributed call), the wait for one, an eager launch
// sequence that depends on host state.  Eagerly (outside capture / replay) the marker is a no-op launch.
__global__ void sba_replay_marker_kernel(int tag) { (void)tag; }

typedef void (*sba_replay_cb)(int tag, void* stream, void* user);

struct Replayer {
    std::vector<RNode> nodes;     // in issue order (topological)
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    hipEvent_t start = nullptr;
    std::vector<int> sink_events; // events recorded after the last node of every stream that ends in a sink
    std::vector<int> tail_event;  // per stream: event recorded at the end of a replay (for the caller to wait on)
    int n_kernels = 0, n_copies = 0, n_memsets = 0, n_waits = 0, n_markers = 0;
    sba_replay_cb cb = nullptr;
    void* cb_user = nullptr;
};

#define RCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "sba_replay: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    (void)hipGetLastError(); return SBA_E_LAUNCH; } } while (0)

static bool is_1d(const hipMemcpy3DParms& p) {
    return p.srcArray == nullptr && p.dstArray == nullptr && p.extent.height <= 1 && p.extent.depth <= 1 &&
           p.srcPos.x == 0 && p.srcPos.y == 0 && p.srcPos.z == 0 && p.dstPos.x == 0 && p.dstPos.y == 0 && p.dstPos.z == 0;
}

}  // namespace

extern "C" int sba_replay_create(void* hip_graph, int max_streams, int flags, void** out) {
    if (!hip_graph || !out || max_streams < 1 || max_streams > 32) return SBA_E_ARG;
    const int verbose = flags & 1;
    const bool on_caller = (flags & 2) != 0;
    hipGraph_t graph = (hipGraph_t)hip_graph;
    size_t n = 0;
    RCHECK(hipGraphGetNodes(graph, nullptr, &n));
    if (n == 0) return SBA_E_ARG;
    std::vector<hipGraphNode_t> gn(n);
    RCHECK(hipGraphGetNodes(graph, gn.data(), &n));
    std::map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[gn[i]] = (int)i;
    size_t ne = 0;
    RCHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) RCHECK(hipGraphGetEdges(graph, ef.data(), et.data(), &ne));
    std::vector<std::vector<int>> deps(n), succ(n);
    for (size_t e = 0; e < ne; ++e) {
        auto a = index.find(ef[e]), b = index.find(et[e]);
        if (a == index.end() || b == index.end()) return SBA_E_ARG;
        deps[b->second].push_back(a->second);
        succ[a->second].push_back(b->second);
    }
    // topological order, ties broken by the capture order (node index): the order the host issued the work in
    std::vector<int> indeg(n), order;
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < n; ++i) { indeg[i] = (int)deps[i].size(); if (!indeg[i]) ready.push((int)i); }
    while (!ready.empty()) {
        const int u = ready.top(); ready.pop();
        order.push_back(u);
        for (int v : succ[u]) if (--indeg[v] == 0) ready.push(v);
    }
    if (order.size() != n) return SBA_E_ARG;       // a cycle: not a captured graph

    Replayer* R = new Replayer();
    std::vector<int> pos(n, -1);                    // graph node -> position in R->nodes
    std::vector<int> stream_of(n, -1);
    std::vector<char> stream_taken(n, 0);           // a successor already continues this node's stream
    std::vector<int> last_use(max_streams, -1);     // issue position of the last node each stream was given
    int used_streams = 0, undecodable = 0;
    std::vector<int> load(max_streams, 0);
    int rr = 0;
    const char* pol = getenv("SBA_REPLAY_POLICY");
    const int policy = pol ? atoi(pol) : 0;
    for (int u : order) {
        RNode r;
        memset(&r.kp, 0, sizeof(r.kp)); memset(&r.cp, 0, sizeof(r.cp)); memset(&r.ms, 0, sizeof(r.ms));
        r.hfunc = nullptr; r.record = -1; r.sink = succ[u].empty(); r.marker = -1;
        if (hipGraphNodeGetType(gn[u], &r.type) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
        if (r.type == hipGraphNodeTypeKernel) {
            if (hipGraphKernelNodeGetParams(gn[u], &r.kp) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
            hipFunction_t f = nullptr;
            if (hipGetFuncBySymbol(&f, r.kp.func) != hipSuccess) {      // not a host-side kernel symbol:
                (void)hipGetLastError();
                r.hfunc = (hipFunction_t)r.kp.func;                      // the node holds a module function
            }
            if (r.kp.func == (void*)sba_replay_marker_kernel && r.kp.kernelParams && r.kp.kernelParams[0]) {
                r.marker = *reinterpret_cast<int*>(r.kp.kernelParams[0]);
                ++R->n_markers;
            }
            ++R->n_kernels;
        } else if (r.type == hipGraphNodeTypeMemcpy) {
            const hipError_t ge = hipGraphMemcpyNodeGetParams(gn[u], &r.cp);
            if (ge != hipSuccess || !is_1d(r.cp)) {
                (void)hipGetLastError();
                ++undecodable;
                if (verbose) {
                    // name the neighbours so that the copy can be traced to the host code that issued it
                    auto kname = [&](int v) -> const char* {
                        hipGraphNodeType t;
                        hipKernelNodeParams kp;
                        if (hipGraphNodeGetType(gn[v], &t) != hipSuccess || t != hipGraphNodeTypeKernel) return "(not a kernel)";
                        if (hipGraphKernelNodeGetParams(gn[v], &kp) != hipSuccess) return "(?)";
                        const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
                        return nm ? nm : "(?)";
                    };
                    fprintf(stderr, "sba_replay: memcpy node %d not decodable as a 1-D copy (GetParams: %s; extent %zu x %zu x %zu)\n",
                            u, hipGetErrorString(ge), r.cp.extent.width, r.cp.extent.height, r.cp.extent.depth);
                    for (int d : deps[u]) fprintf(stderr, "    after  %.120s\n", kname(d));
                    for (int v : succ[u]) fprintf(stderr, "    before %.120s\n", kname(v));
                }
                continue;
            }
            ++R->n_copies;
        } else if (r.type == hipGraphNodeTypeMemset) {
            if (hipGraphMemsetNodeGetParams(gn[u], &r.ms) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
            ++R->n_memsets;
        } else if (r.type != hipGraphNodeTypeEmpty) {
            if (verbose) fprintf(stderr, "sba_replay: node %d has unsupported type %d\n", u, (int)r.type);
            delete R;
            return SBA_E_UNSUPPORTED;
        }
        // stream: continue the stream of the LAST-issued dependency that nobody continues yet, else fork
        int s = -1, best = -1;
        for (int d : deps[u])
            if (!stream_taken[d] && pos[d] > best) { best = pos[d]; s = stream_of[d]; }
        if (s >= 0) {
            for (int d : deps[u]) if (pos[d] == best) stream_taken[d] = 1;
        } else {
            // a fork (or a root): the stream that has been quiet longest -- a finished branch's stream rather
            // than one a long chain is still being issued to
            s = 0;
            if (policy == 1) {              // (A/B aid) the stream with the fewest nodes so far
                for (int t = 1; t < max_streams; ++t) if (load[t] < load[s]) s = t;
            } else if (policy == 2) {       // (A/B aid) round robin over the forks
                s = (rr++) % max_streams;
            } else {
                for (int t = 1; t < max_streams; ++t) if (last_use[t] < last_use[s]) s = t;
            }
        }
        ++load[s];
        used_streams = std::max(used_streams, s + 1);
        stream_of[u] = s;
        r.stream = s;
        pos[u] = (int)R->nodes.size();
        last_use[s] = pos[u];
        R->nodes.push_back(r);
    }
    if (undecodable) { delete R; return SBA_E_UNSUPPORTED; }
    // events where a dependency crosses streams.  Within one stream the issue order IS the dependency order,
    // and a wait on producer p also covers everything issued before p on p's stream.
    std::vector<int> last_waited(used_streams * used_streams, -1);   // [consumer stream][producer stream] -> issue position
    for (int u : order) {
        RNode& r = R->nodes[pos[u]];
        for (int d : deps[u]) {
            RNode& p = R->nodes[pos[d]];
            if (p.stream == r.stream) continue;
            int& lw = last_waited[r.stream * used_streams + p.stream];
            if (lw >= pos[d]) continue;                     // already ordered behind a later node of that stream
            if (p.record < 0) {
                p.record = (int)R->events.size();
                R->events.push_back(nullptr);
            }
            r.waits.push_back(p.record);
            lw = pos[d];
            ++R->n_waits;
        }
    }
    // flags bit 1: stream 0 of the assignment IS the caller's stream (sba_replay_launch): a single-chain graph -- one phase
    // of the step -- then needs no stream of its own (a process has only 4 hardware queues by default; every extra stream
    // shares one).  Not for a caller on the NULL stream: the whole step with its main chain there measured 12.0 against
    // 11.15 ms on a stream of the replayer's own.
    R->streams.assign(used_streams, nullptr);
    for (int s = on_caller ? 1 : 0; s < used_streams; ++s)
        if (hipStreamCreateWithFlags(&R->streams[s], hipStreamNonBlocking) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
    for (auto& e : R->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
    if (hipEventCreateWithFlags(&R->start, hipEventDisableTiming) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
    R->tail_event.assign(used_streams, -1);
    for (int s = 0; s < used_streams; ++s) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
        R->tail_event[s] = (int)R->events.size();
        R->events.push_back(e);
    }
    if (verbose)
        fprintf(stderr, "sba_replay: %zu nodes (%d kernels, %d copies, %d memsets), %zu edges -> %d streams, %d cross-stream waits\n",
                n, R->n_kernels, R->n_copies, R->n_memsets, ne, used_streams, R->n_waits);
    *out = R;
    return SBA_OK;
}

extern "C" int sba_replay_launch(void* handle, void* stream) {
    if (!handle) return SBA_E_ARG;
    Replayer* R = (Replayer*)handle;
    hipStream_t caller = (hipStream_t)stream;
    (void)hipGetLastError();
    static int timing = -1;     // SBA_REPLAY_TIMING=1: print the host time one replay takes to issue (tuning aid)
    if (timing < 0) { const char* e = getenv("SBA_REPLAY_TIMING"); timing = (e && e[0] == '1') ? 1 : 0; }
    const auto t0 = std::chrono::steady_clock::now();
    const bool own0 = !R->streams.empty() && R->streams[0] != nullptr;
    if (R->streams.size() > 1 || own0) RCHECK(hipEventRecord(R->start, caller));
    for (size_t s = own0 ? 0 : 1; s < R->streams.size(); ++s) RCHECK(hipStreamWaitEvent(R->streams[s], R->start, 0));
    for (RNode& r : R->nodes) {
        hipStream_t s = R->streams[r.stream] ? R->streams[r.stream] : caller;
        for (int e : r.waits) RCHECK(hipStreamWaitEvent(s, R->events[e], 0));
        switch (r.type) {
            case hipGraphNodeTypeKernel:
                if (r.marker >= 0) {
                    // host-call node: everything it depends on has been ISSUED (stream-ordered before s's current
                    // position); the callback may enqueue on s and make s wait for other streams
                    if (R->cb) R->cb(r.marker, (void*)s, R->cb_user);
                    break;
                }
                if (r.hfunc)
                    RCHECK(hipModuleLaunchKernel(r.hfunc, r.kp.gridDim.x, r.kp.gridDim.y, r.kp.gridDim.z, r.kp.blockDim.x,
                                                 r.kp.blockDim.y, r.kp.blockDim.z, r.kp.sharedMemBytes, s,
                                                 r.kp.kernelParams, r.kp.extra));
                else
                    RCHECK(hipLaunchKernel(r.kp.func, r.kp.gridDim, r.kp.blockDim, r.kp.kernelParams,
                                           r.kp.sharedMemBytes, s));
                break;
            case hipGraphNodeTypeMemcpy:
                RCHECK(hipMemcpyAsync(r.cp.dstPtr.ptr, r.cp.srcPtr.ptr, r.cp.extent.width, r.cp.kind, s));
                break;
            case hipGraphNodeTypeMemset:
                if (r.ms.height <= 1) {
                    if (r.ms.elementSize == 4) RCHECK(hipMemsetD32Async((hipDeviceptr_t)r.ms.dst, (int)r.ms.value, r.ms.width, s));
                    else if (r.ms.elementSize == 2) RCHECK(hipMemsetD16Async((hipDeviceptr_t)r.ms.dst, (unsigned short)r.ms.value, r.ms.width, s));
                    else RCHECK(hipMemsetAsync(r.ms.dst, (int)r.ms.value, r.ms.width, s));
                } else {
                    RCHECK(hipMemset2DAsync(r.ms.dst, r.ms.pitch, (int)r.ms.value, r.ms.width * r.ms.elementSize, r.ms.height, s));
                }
                break;
            default:
                break;      // empty node: ordering only
        }
        if (r.record >= 0) RCHECK(hipEventRecord(R->events[r.record], s));
    }
    for (size_t s = own0 ? 0 : 1; s < R->streams.size(); ++s) {
        hipEvent_t e = R->events[R->tail_event[s]];
        RCHECK(hipEventRecord(e, R->streams[s]));
        RCHECK(hipStreamWaitEvent(caller, e, 0));
    }
    if (timing) {
        static int calls = 0;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (++calls % 8 == 0) fprintf(stderr, "sba_replay: issued %zu nodes in %.0f us of host time (%.2f us per node)\n",
                                      R->nodes.size(), us, us / R->nodes.size());
    }
    return SBA_OK;
}

extern "C" int sba_replay_info(void* handle, int* info8) {
    if (!handle || !info8) return SBA_E_ARG;
    Replayer* R = (Replayer*)handle;
    info8[0] = (int)R->nodes.size(); info8[1] = R->n_kernels; info8[2] = R->n_copies; info8[3] = R->n_memsets;
    info8[4] = (int)R->streams.size(); info8[5] = R->n_waits; info8[6] = (int)R->events.size(); info8[7] = R->n_markers;
    return SBA_OK;
}

extern "C" int sba_replay_marker(int tag, void* stream) {
    if (tag < 0) return SBA_E_ARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(sba_replay_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, tag);
    return hipGetLastError() == hipSuccess ? SBA_OK : SBA_E_LAUNCH;
}

extern "C" int sba_replay_set_callback(void* handle, void* fn, void* user) {
    if (!handle) return SBA_E_ARG;
    Replayer* R = (Replayer*)handle;
    R->cb = (sba_replay_cb)fn;
    R->cb_user = user;
    return SBA_OK;
}

extern "C" int sba_replay_destroy(void* handle) {
    if (!handle) return SBA_E_ARG;
    Replayer* R = (Replayer*)handle;
    for (hipStream_t s : R->streams) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : R->events) if (e) (void)hipEventDestroy(e);
    if (R->start) (void)hipEventDestroy(R->start);
    delete R;
    return SBA_OK;
}
