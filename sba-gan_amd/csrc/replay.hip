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
#include <cmath>
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
    // the recorded dependencies (positions in `nodes`) and what the stream assignment works from
    std::vector<std::vector<int>> deps;
    std::vector<char> critical;   // per node: 1 = on (or within the slack threshold of) the longest path; sba_replay_prioritize
    std::vector<float> dur_us;    // per node: duration measured alone (sba_replay_prioritize), else empty
    int max_streams = 1, n_high = 0, prio_mode = 0;
    bool on_caller = false;
};

#define RCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "sba_replay: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    (void)hipGetLastError(); return SBA_E_LAUNCH; } } while (0)

static bool is_1d(const hipMemcpy3DParms& p) {
    return p.srcArray == nullptr && p.dstArray == nullptr && p.extent.height <= 1 && p.extent.depth <= 1 &&
           p.srcPos.x == 0 && p.srcPos.y == 0 && p.srcPos.z == 0 && p.dstPos.x == 0 && p.dstPos.y == 0 && p.dstPos.z == 0;
}


// Give every node a stream and place events where a dependency crosses streams.  A chain keeps its stream: a node continues
// the stream of the LAST-issued dependency that nobody continues yet; a fork (or a root) takes the stream that has been quiet
// longest -- a finished branch's stream rather than one a long chain is still being issued to.  With priorities
// (R->prio_mode, sba_replay_prioritize) the streams form two pools -- [0, n_high) for the nodes marked critical, created with
// the higher priority, the rest for everything else -- and a node continues a dependency's stream only inside its own pool.
static int assign_streams(Replayer* R) {
    const int n = (int)R->nodes.size(), S = R->max_streams;
    for (hipStream_t s : R->streams) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : R->events) if (e) (void)hipEventDestroy(e);
    R->streams.clear(); R->events.clear(); R->tail_event.clear(); R->n_waits = 0;
    const bool pools = R->prio_mode != 0 && !R->on_caller && S >= 2;
    const int nh = pools ? std::max(1, std::min(R->n_high, S - 1)) : 0;
    std::vector<char> taken(n, 0);                  // a successor already continues this node's stream
    std::vector<int> last_use(S, -1), load(S, 0);   // issue position of the last node each stream was given
    int used_streams = 0, rr = 0;
    const char* pol = getenv("SBA_REPLAY_POLICY");
    const int policy = pol ? atoi(pol) : 0;
    for (int u = 0; u < n; ++u) {
        RNode& r = R->nodes[u];
        r.waits.clear();
        r.record = -1;
        const int lo = pools ? (R->critical[u] ? 0 : nh) : 0, hi = pools ? (R->critical[u] ? nh : S) : S;
        int s = -1, best = -1;
        for (int d : R->deps[u]) {
            const int sd = R->nodes[d].stream;
            if (!taken[d] && d > best && sd >= lo && sd < hi) { best = d; s = sd; }
        }
        if (s >= 0) {
            taken[best] = 1;
        } else {
            s = lo;
            if (policy == 1) {              // (A/B aid) the stream with the fewest nodes so far
                for (int t = lo + 1; t < hi; ++t) if (load[t] < load[s]) s = t;
            } else if (policy == 2) {       // (A/B aid) round robin over the forks
                s = lo + (rr++) % (hi - lo);
            } else {
                for (int t = lo + 1; t < hi; ++t) if (last_use[t] < last_use[s]) s = t;
            }
        }
        ++load[s];
        used_streams = std::max(used_streams, s + 1);
        r.stream = s;
        last_use[s] = u;
    }
    // events where a dependency crosses streams.  Within one stream the issue order IS the dependency order,
    // and a wait on producer p also covers everything issued before p on p's stream.
    std::vector<int> last_waited(used_streams * used_streams, -1);   // [consumer stream][producer stream] -> issue position
    for (int u = 0; u < n; ++u) {
        RNode& r = R->nodes[u];
        for (int d : R->deps[u]) {
            RNode& p = R->nodes[d];
            if (p.stream == r.stream) continue;
            int& lw = last_waited[r.stream * used_streams + p.stream];
            if (lw >= d) continue;                          // already ordered behind a later node of that stream
            if (p.record < 0) {
                p.record = (int)R->events.size();
                R->events.push_back(nullptr);
            }
            r.waits.push_back(p.record);
            lw = d;
            ++R->n_waits;
        }
    }
    // on_caller: stream 0 of the assignment IS the caller's stream (sba_replay_launch): a single-chain graph -- one phase
    // of the step -- then needs no stream of its own (a process has only 4 hardware queues by default; every extra stream
    // shares one).  Not for a caller on the NULL stream: the whole step with its main chain there measured 12.0 against
    // 11.15 ms on a stream of the replayer's own.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);         // (numerically: least >= greatest)
    const int p_hi = (R->prio_mode == 3) ? 0 : greatest, p_lo = (R->prio_mode == 1) ? 0 : least;
    R->streams.assign(used_streams, nullptr);
    for (int s = R->on_caller ? 1 : 0; s < used_streams; ++s) {
        const hipError_t e = pools ? hipStreamCreateWithPriority(&R->streams[s], hipStreamNonBlocking, s < nh ? p_hi : p_lo)
                                   : hipStreamCreateWithFlags(&R->streams[s], hipStreamNonBlocking);
        if (e != hipSuccess) { (void)hipGetLastError(); return SBA_E_LAUNCH; }
    }
    for (auto& e : R->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return SBA_E_LAUNCH;
    R->tail_event.assign(used_streams, -1);
    for (int s = 0; s < used_streams; ++s) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return SBA_E_LAUNCH;
        R->tail_event[s] = (int)R->events.size();
        R->events.push_back(e);
    }
    return SBA_OK;
}

// one recorded node onto stream s
static int issue_node(Replayer* R, RNode& r, hipStream_t s) {
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
    return SBA_OK;
}

}  // namespace

extern "C" int sba_replay_destroy(void* handle);

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
    R->max_streams = max_streams;
    R->on_caller = on_caller;
    std::vector<int> pos(n, -1);                    // graph node -> position in R->nodes
    int undecodable = 0;
    for (int u : order) {
        RNode r;
        memset(&r.kp, 0, sizeof(r.kp)); memset(&r.cp, 0, sizeof(r.cp)); memset(&r.ms, 0, sizeof(r.ms));
        r.hfunc = nullptr; r.record = -1; r.sink = succ[u].empty(); r.marker = -1; r.stream = 0;
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
        pos[u] = (int)R->nodes.size();
        R->nodes.push_back(r);
        std::vector<int> dp;
        for (int d : deps[u]) dp.push_back(pos[d]);
        R->deps.push_back(dp);
    }
    if (undecodable) { delete R; return SBA_E_UNSUPPORTED; }
    R->critical.assign(R->nodes.size(), 0);
    if (hipEventCreateWithFlags(&R->start, hipEventDisableTiming) != hipSuccess) { delete R; return SBA_E_LAUNCH; }
    const int rc = assign_streams(R);
    if (rc != SBA_OK) { sba_replay_destroy(R); return rc; }
    if (verbose)
        fprintf(stderr, "sba_replay: %zu nodes (%d kernels, %d copies, %d memsets), %zu edges -> %zu streams, %d cross-stream waits\n",
                n, R->n_kernels, R->n_copies, R->n_memsets, ne, R->streams.size(), R->n_waits);
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
        { const int rc = issue_node(R, r, s); if (rc != SBA_OK) return rc; }
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


extern "C" int sba_replay_prioritize(void* handle, void* stream, int mode, int max_streams, int n_high, float slack_frac,
                                     int verbose) {
    if (!handle || mode < 0 || mode > 3 || max_streams < 2 || max_streams > 32 || n_high < 1 || n_high >= max_streams ||
        !(slack_frac >= 0.f && slack_frac <= 1.f))
        return SBA_E_ARG;
    Replayer* R = (Replayer*)handle;
    if (R->on_caller) return SBA_E_ARG;
    hipStream_t caller = (hipStream_t)stream;
    const int n = (int)R->nodes.size();
    (void)hipGetLastError();
    // 1. every node alone, in issue order on the caller's stream, an event between neighbours: durations without contention
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) {
            for (auto& d : ev) if (d) (void)hipEventDestroy(d);
            (void)hipGetLastError();
            return SBA_E_LAUNCH;
        }
    int rc = SBA_OK;
    for (int u = 0; u < n && rc == SBA_OK; ++u) {
        if (hipEventRecord(ev[u], caller) != hipSuccess) rc = SBA_E_LAUNCH;
        if (rc == SBA_OK) rc = issue_node(R, R->nodes[u], caller);
    }
    if (rc == SBA_OK && hipEventRecord(ev[n], caller) != hipSuccess) rc = SBA_E_LAUNCH;
    if (hipStreamSynchronize(caller) != hipSuccess) rc = SBA_E_LAUNCH;
    R->dur_us.assign(n, 0.f);
    for (int u = 0; u < n && rc == SBA_OK; ++u) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[u], ev[u + 1]) != hipSuccess) rc = SBA_E_LAUNCH;
        R->dur_us[u] = ms * 1e3f;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc != SBA_OK) { (void)hipGetLastError(); return rc; }
    // 2. earliest start (tl) and longest path to the end (bl) of every node; slack = longest path of the step - (tl + bl)
    std::vector<double> tl(n, 0.0), bl(n, 0.0), below(n, 0.0);
    for (int u = 0; u < n; ++u)
        for (int d : R->deps[u]) tl[u] = std::max(tl[u], tl[d] + (double)R->dur_us[d]);
    for (int u = n - 1; u >= 0; --u) {
        bl[u] = (double)R->dur_us[u] + below[u];
        for (int d : R->deps[u]) below[d] = std::max(below[d], bl[u]);
    }
    double cp = 0.0, total = 0.0;
    for (int u = 0; u < n; ++u) { cp = std::max(cp, tl[u] + bl[u]); total += R->dur_us[u]; }
    const double thr = slack_frac * cp;
    int ncrit = 0;
    double tcrit = 0.0;
    for (int u = 0; u < n; ++u) {
        const double slack = cp - tl[u] - bl[u];
        bool c = slack <= thr;
        if (!c && slack <= 2.0 * thr)           // hysteresis: a chain does not hop between the pools at the threshold
            for (int d : R->deps[u]) c = c || R->critical[d];
        R->critical[u] = (mode != 0 && c) ? 1 : 0;
        if (c) { ++ncrit; tcrit += R->dur_us[u]; }
    }
    if (verbose) {
        fprintf(stderr, "sba_replay_prioritize: %d nodes, %.0f us of work alone, longest dependency path %.0f us; %d nodes "
                        "(%.0f us) within %.0f us of it -> %d high-priority stream(s) of %d (mode %d)\n",
                n, total, cp, ncrit, tcrit, thr, n_high, max_streams, mode);
        auto node_name = [&](const RNode& r) -> const char* {
            const char* nm = r.type == hipGraphNodeTypeKernel ? (r.marker >= 0 ? "(host call)" : (r.hfunc ? "(module function)" : hipKernelNameRefByPtr(r.kp.func, nullptr)))
                             : r.type == hipGraphNodeTypeMemcpy ? "(copy)" : r.type == hipGraphNodeTypeMemset ? "(fill)" : "(empty)";
            return nm ? nm : "(?)";
        };
        if (verbose > 2)                        // every node: duration alone, slack, kernel
            for (int v = 0; v < n; ++v)
                fprintf(stderr, "  node %4d %7.1f %8.1f  %.100s (%u,%u,%u)\n", v, R->dur_us[v], cp - tl[v] - bl[v], node_name(R->nodes[v]),
                        R->nodes[v].kp.gridDim.x, R->nodes[v].kp.gridDim.y, R->nodes[v].kp.gridDim.z);
        if (verbose > 1) {                      // the longest path, node by node
            int u = -1;
            for (int v = 0; v < n; ++v) if (R->deps[v].empty() && bl[v] >= cp - 1e-6) { u = v; break; }
            std::vector<std::vector<int>> succ(n);
            for (int v = 0; v < n; ++v) for (int d : R->deps[v]) succ[d].push_back(v);
            while (u >= 0) {
                const RNode& r = R->nodes[u];
                fprintf(stderr, "  cp %8.1f %7.1f  %.100s (%u,%u,%u)\n", tl[u], R->dur_us[u], node_name(r), r.kp.gridDim.x, r.kp.gridDim.y, r.kp.gridDim.z);
                int nx = -1;
                for (int v : succ[u]) if (nx < 0 || bl[v] > bl[nx]) nx = v;
                u = nx;
            }
        }
    }
    // 3. the streams again, in two pools
    R->prio_mode = mode;
    R->max_streams = max_streams;
    R->n_high = n_high;
    rc = assign_streams(R);
    if (verbose) {
        fprintf(stderr, "sba_replay_prioritize: %zu streams, %d cross-stream waits\n", R->streams.size(), R->n_waits);
        // which chains share a stream: per stream, the work alone and the nodes by slack (a chain's nodes share one slack value)
        for (size_t st = 0; st < R->streams.size(); ++st) {
            std::map<long, std::pair<int, double>> by_slack;
            double w = 0.0;
            int cnt = 0;
            for (int u = 0; u < n; ++u)
                if (R->nodes[u].stream == (int)st) {
                    auto& b = by_slack[lround(cp - tl[u] - bl[u])];
                    ++b.first; b.second += R->dur_us[u];
                    w += R->dur_us[u]; ++cnt;
                }
            fprintf(stderr, "  stream %zu: %d nodes, %.0f us alone; slack(us):nodes:us", st, cnt, w);
            for (auto& kv : by_slack) if (kv.second.first >= 8) fprintf(stderr, "  %ld:%d:%.0f", kv.first, kv.second.first, kv.second.second);
            fprintf(stderr, "\n");
        }
        // the makespan these streams allow when nothing slows anything down: every node starts when its dependencies AND its
        // predecessor on its stream have finished
        std::vector<double> fin(n, 0.0), sfree(R->streams.size(), 0.0);
        double span = 0.0;
        for (int u = 0; u < n; ++u) {
            double t0 = sfree[R->nodes[u].stream];
            for (int d : R->deps[u]) t0 = std::max(t0, fin[d]);
            fin[u] = t0 + R->dur_us[u];
            sfree[R->nodes[u].stream] = fin[u];
            span = std::max(span, fin[u]);
        }
        fprintf(stderr, "sba_replay_prioritize: with this stream assignment and no contention the recording takes %.0f us "
                        "(longest dependency path %.0f us)\n", span, cp);
    }
    return rc;
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
