// A multi-stream executor for a captured HIP graph (host code; no reference counterpart - the reference queues every op from Python).
//
// Why: a training step is ~600 launches on four streams.  Queued from Python the host needs ~10 ms per step and a slow host makes
// the step host-bound; replayed as a hipGraph the host is out of the loop, but this runtime executes a graph's parallel branches
// level by level (measured: the replayed S1 step takes 13.9 ms with or without its weight-gradient branch, the eager one 12.4 ms
// on four free-running streams).  This executor takes the CAPTURED graph only as a recording - kernel / memset / memcpy nodes with
// their parameters and dependency edges - and launches the nodes itself from a C loop: every node on one of a few HIP streams
// (a node follows one of its predecessors' streams when it can), every cross-stream edge as an event record + stream wait.  The
// device then sees what the eager step shows it - independent queues that overlap freely - at ~1-2 us of host time per node.
//
// The graph stays owned by the caller (torch.cuda.CUDAGraph(keep_graph=True)): the executor reads the nodes' parameter blocks in
// place, and the caller keeps the capture's memory pool alive.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "asr_common.h"

namespace {

struct XNode {
    hipGraphNodeType type;
    hipKernelNodeParams k;
    hipMemsetParams ms;
    hipGraphExec_t sub = nullptr;  // memcpy node: a one-node graph of its own (see create)
    hipGraph_t sub_graph = nullptr;
    int stream;                    // index into GraphX::streams (0 = the launch stream)
    std::vector<int> wait_events;  // events of cross-stream predecessors
    int record_event;              // event recorded after this node (-1: none)
    // a bucket-ready marker (collective.hip): the launch loop queues the all-reduce of cbuf[0..ccount) on this node's stream instead
    bool collective = false;
    float* cbuf = nullptr;
    long long ccount = 0;
    int ctag = 0;
};

// Events between the executor's streams order work on ONE device: they need no system-scope fence.  A default HIP event performs one
// when it is recorded (cache write-back and invalidation so that the host and other devices see the data) - measured ~6.5 us of
// nothing between the recording kernel and the next kernel of its stream, 49 times per S1 step.  (`sysfence` below restores it.)
unsigned graphx_event_flags() {
    constexpr bool sysfence = false;      // (true: HIP's default system-scope fence at every record, S1 replay +0.1-0.2 ms)
    // (hipEventReleaseToDevice instead: 12.06 against 11.99 ms; both flags together are rejected by the runtime)
    return hipEventDisableTiming | (sysfence ? 0u : hipEventDisableSystemFence);
}

struct GraphX {
    std::vector<XNode> nodes;      // in launch (topological) order
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    std::vector<int> tail_event;   // per side stream: the event recorded after its last node (joined into stream 0 at the end)
    hipEvent_t begin = nullptr;
    int n_kernel = 0, n_memset = 0, n_memcpy = 0, n_other = 0, n_collective = 0;
    void* comm = nullptr;          // RCCL communicator of the collective nodes (asr_graphx_set_collective)
    asr_collective_fn coll_fn = nullptr;   // ... or the caller's own all-reduce (a test rig's gloo ranks sharing one GPU)
    void* coll_ctx = nullptr;
    int rotation = 0;              // logical side stream i -> physical side stream (i + rotation) mod n (asr_graphx_set_rotation)
    std::vector<int> map;          // explicit logical -> physical side-stream map (asr_graphx_place_streams); empty: the rotation
};

#define GX_CHECK(call)                                                                 \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            asr_set_error("graphx: %s failed: %s", #call, hipGetErrorString(e__));     \
            return (int)e__;                                                           \
        }                                                                              \
    } while (0)

}  // namespace

extern "C" int asr_graphx_destroy(void* handle) {
    GraphX* g = static_cast<GraphX*>(handle);
    if (!g) return 0;
    for (XNode& x : g->nodes) {
        if (x.sub) hipGraphExecDestroy(x.sub);
        if (x.sub_graph) hipGraphDestroy(x.sub_graph);
    }
    for (size_t i = 1; i < g->streams.size(); ++i) hipStreamDestroy(g->streams[i]);
    for (hipEvent_t e : g->events) hipEventDestroy(e);
    if (g->begin) hipEventDestroy(g->begin);
    delete g;
    return 0;
}

extern "C" int asr_graphx_create(void* hip_graph, int max_streams, void** out_handle) {
    ASR_REQUIRE(hip_graph && out_handle && max_streams >= 1 && max_streams <= 16, -1, "graphx_create: bad arguments");
    hipGraph_t graph = static_cast<hipGraph_t>(hip_graph);
    size_t n = 0;
    GX_CHECK(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> raw(n);
    GX_CHECK(hipGraphGetNodes(graph, raw.data(), &n));
    std::vector<hipGraphNode_t> sorted_raw = raw;
    std::sort(sorted_raw.begin(), sorted_raw.end());
    auto index_of = [&](hipGraphNode_t h) { return (int)(std::lower_bound(sorted_raw.begin(), sorted_raw.end(), h) - sorted_raw.begin()); };
    // dependency lists (indices into sorted_raw) and a topological order that follows the capture order where it can
    std::vector<std::vector<int>> deps(n), succ(n);
    for (size_t i = 0; i < n; ++i) {
        size_t nd = 0;
        GX_CHECK(hipGraphNodeGetDependencies(sorted_raw[i], nullptr, &nd));
        std::vector<hipGraphNode_t> d(nd);
        if (nd) GX_CHECK(hipGraphNodeGetDependencies(sorted_raw[i], d.data(), &nd));
        for (size_t j = 0; j < nd; ++j) {
            const int p = index_of(d[j]);
            deps[i].push_back(p);
            succ[p].push_back((int)i);
        }
    }
    std::vector<int> capture_pos(n);      // position in hipGraphGetNodes' own order: the order the nodes were added in
    for (size_t i = 0; i < n; ++i) capture_pos[index_of(raw[i])] = (int)i;
    std::vector<int> indeg(n), order;
    std::vector<int> ready;
    for (size_t i = 0; i < n; ++i) {
        indeg[i] = (int)deps[i].size();
        if (!indeg[i]) ready.push_back((int)i);
    }
    auto by_capture = [&](int a, int b) { return capture_pos[a] > capture_pos[b]; };      // heap: smallest capture position first
    std::make_heap(ready.begin(), ready.end(), by_capture);
    while (!ready.empty()) {
        std::pop_heap(ready.begin(), ready.end(), by_capture);
        const int u = ready.back();
        ready.pop_back();
        order.push_back(u);
        for (int v : succ[u])
            if (--indeg[v] == 0) {
                ready.push_back(v);
                std::push_heap(ready.begin(), ready.end(), by_capture);
            }
    }
    ASR_REQUIRE(order.size() == n, -1, "graphx_create: the graph has a cycle (%zu of %zu nodes ordered)", order.size(), n);

    GraphX* g = new GraphX();
    struct Guard {       // every early return below (GX_CHECK, ASR_REQUIRE) releases the half-built plan: streams, events, cloned sub-graphs
        GraphX* g;
        ~Guard() { if (g) asr_graphx_destroy(g); }
    } guard{g};
    g->streams.push_back(nullptr);     // slot 0: the stream handed to asr_graphx_launch
    std::vector<int> stream_of(n, -1), tail_of_stream(1, -1), event_of(n, -1);
    std::vector<int> pos_in_order(n);
    for (size_t i = 0; i < n; ++i) pos_in_order[order[i]] = (int)i;
    // Which successor inherits a node's stream: the one with the longest chain of work still behind it ("height").  A captured
    // stream's own next launch and an event-ordered launch on another stream look alike in the graph (both are edges); but the step's
    // main chain is hundreds of nodes long, a side branch (weight-gradient GEMMs, mask hashing, the CTC branch) is a short chain
    // hanging off it - so the main chain keeps its stream through every fork, and each side chain keeps its own.
    std::vector<int> height(n, 1), heir(n, -1);
    for (size_t oi = n; oi-- > 0;) {
        const int u = order[oi];
        for (int v : succ[u]) {
            if (height[v] + 1 > height[u]) height[u] = height[v] + 1;
            if (heir[u] < 0 || height[v] > height[heir[u]] || (height[v] == height[heir[u]] && capture_pos[v] < capture_pos[heir[u]])) heir[u] = v;
        }
    }
    for (size_t oi = 0; oi < n; ++oi) {
        const int u = order[oi];
        int st = -1, from = -1;
        for (int p : deps[u])
            if (heir[p] == u && tail_of_stream[stream_of[p]] == p && (from < 0 || height[p] > height[from])) from = p;
        if (from >= 0) st = stream_of[from];
        if (st < 0) {
            if (tail_of_stream[0] < 0) st = 0;                                           // the first node opens the launch stream
            else {
                // nobody's heir: a stream that is free to take it - its tail has handed its stream on already (or ends there); one whose
                // tail is a predecessor of this node first (no event needed), then a new stream, then the least recently used one
                int reuse = -1, lru = -1;
                for (size_t sidx = 1; sidx < g->streams.size(); ++sidx) {
                    const int t = tail_of_stream[sidx];
                    const bool free_now = t >= 0 && (heir[t] < 0 || stream_of[heir[t]] >= 0);
                    if (!free_now) continue;
                    const bool is_pred = std::find(deps[u].begin(), deps[u].end(), t) != deps[u].end();
                    if (is_pred) { reuse = (int)sidx; break; }
                    if (lru < 0 || pos_in_order[t] < pos_in_order[tail_of_stream[lru]]) lru = (int)sidx;
                }
                if (reuse >= 0) st = reuse;
                else if ((int)g->streams.size() < max_streams) {
                    hipStream_t ns;
                    // (side streams of another priority get hardware queues of their own, and cross-queue ordering is what costs: S1
                    // replay 23.8 / 19.3 ms at the lowest / highest priority against 12.4)
                    const hipError_t ce = hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
                    if (ce != hipSuccess) { asr_set_error("graphx: stream"); return -2; }
                    g->streams.push_back(ns);
                    tail_of_stream.push_back(-1);
                    st = (int)g->streams.size() - 1;
                } else if (lru >= 0) st = lru;
                else {                                                                   // every stream is mid-chain: behind the predecessor launched last
                    int latest = -1;
                    for (int p : deps[u])
                        if (latest < 0 || pos_in_order[p] > pos_in_order[latest]) latest = p;
                    st = latest >= 0 ? stream_of[latest] : 0;
                }
            }
        }
        stream_of[u] = st;
        tail_of_stream[st] = u;
    }
    // events for cross-stream edges; (an edge inside one stream is the stream's own order - `order` is topological)
    g->nodes.resize(n);
    for (size_t oi = 0; oi < n; ++oi) {
        const int u = order[oi];
        XNode& x = g->nodes[oi];
        x.stream = stream_of[u];
        x.record_event = -1;
        GX_CHECK(hipGraphNodeGetType(sorted_raw[u], &x.type));
        if (x.type == hipGraphNodeTypeKernel) {
            GX_CHECK(hipGraphKernelNodeGetParams(sorted_raw[u], &x.k));
            if (x.k.kernelParams == nullptr || x.k.extra != nullptr) {
                asr_set_error("graphx_create: a kernel node passes its arguments through `extra` (not a hipLaunchKernel-style launch)");
                return -3;
            }
            {   // a node captured from hipModuleLaunchKernel carries a hipFunction_t, not a host stub: hipLaunchKernel would only fail at
                // launch time, with earlier nodes already queued - refuse the graph here instead (the caller replays it with hipGraphLaunch)
                hipFuncAttributes fa;
                if (hipFuncGetAttributes(&fa, x.k.func) != hipSuccess) {
                    (void)hipGetLastError();
                    asr_set_error("graphx_create: a kernel node's function is not a host-side kernel stub");
                    return -3;
                }
            }
            g->n_kernel++;
            if (x.k.func == asr_collective_marker_func()) {      // values, not pointers into the node: the block is the graph's
                x.collective = true;
                x.cbuf = *static_cast<float**>(x.k.kernelParams[0]);
                x.ccount = *static_cast<long long*>(x.k.kernelParams[1]);
                x.ctag = *static_cast<int*>(x.k.kernelParams[2]);
                g->n_collective++;
            }
        } else if (x.type == hipGraphNodeTypeMemset) {
            GX_CHECK(hipGraphMemsetNodeGetParams(sorted_raw[u], &x.ms));
            g->n_memset++;
        } else if (x.type == hipGraphNodeTypeMemcpy) {
            // A copy recorded by hipMemcpyAsync is a 1-D memcpy node, and this runtime has no getter for those (hipGraphMemcpyNodeGetParams
            // leaves its 3-D parameter block untouched).  The node is replayed as what it is instead: a clone of the graph with every other
            // node removed, instantiated once, launched on the node's stream (a handful of small copies per step).
            hipGraphNode_t cn = nullptr;
            GX_CHECK(hipGraphClone(&x.sub_graph, graph));
            GX_CHECK(hipGraphNodeFindInClone(&cn, sorted_raw[u], x.sub_graph));
            size_t cnn = 0;
            GX_CHECK(hipGraphGetNodes(x.sub_graph, nullptr, &cnn));
            std::vector<hipGraphNode_t> cl(cnn);
            GX_CHECK(hipGraphGetNodes(x.sub_graph, cl.data(), &cnn));
            for (hipGraphNode_t v : cl)
                if (v != cn) GX_CHECK(hipGraphDestroyNode(v));
            GX_CHECK(hipGraphInstantiate(&x.sub, x.sub_graph, nullptr, nullptr, 0));
            g->n_memcpy++;
        } else if (x.type == hipGraphNodeTypeEmpty) {
            g->n_other++;
        } else {
            asr_set_error("graphx_create: node type %d is not supported (kernel, memset, memcpy and empty nodes are)", (int)x.type);
            return -3;
        }
    }
    // Events for the cross-stream edges.  A recorded event costs the recording stream ~5 us between two kernels (the marker packet),
    // so a waiter on a SIDE stream is pointed at the latest node of the producer's stream that (a) records an event anyway and (b) is
    // issued before the waiter: later in the producer's stream order than the node it needs, hence sufficient, and already queued when
    // the wait is queued.  Producers left without a waiter record nothing.  (Waiters on the busiest stream - the step's main chain -
    // keep their exact producer: they must not wait for more than they need.)  Waits that an earlier wait of the same stream on a
    // later node of the same producer stream already implies are dropped.
    {
        constexpr bool coalesce = true;
        const int ns = (int)g->streams.size();
        std::vector<int> per_stream(ns, 0);
        for (size_t i = 0; i < n; ++i) per_stream[stream_of[i]]++;
        const int busiest = (int)(std::max_element(per_stream.begin(), per_stream.end()) - per_stream.begin());
        std::vector<char> exact(n, 0);      // node is the producer of some cross-stream edge
        for (size_t oi = 0; oi < n; ++oi)
            for (int p : deps[order[oi]])
                if (stream_of[p] != stream_of[order[oi]]) exact[p] = 1;
        std::vector<std::vector<int>> targets(n);      // per node (by order position): the nodes whose events it waits for
        std::vector<int> waiters(n, 0);
        std::vector<int> waited(ns * ns, -1);           // [consumer stream][producer stream]: latest position already waited for
        for (size_t oi = 0; oi < n; ++oi) {
            const int u = order[oi], cs = stream_of[u];
            for (int p : deps[u]) {
                const int ps = stream_of[p];
                if (ps == cs) continue;
                int q = p;
                if (coalesce && cs != busiest)
                    for (size_t oj = (size_t)pos_in_order[p] + 1; oj < oi; ++oj)
                        if (stream_of[order[oj]] == ps && exact[order[oj]]) q = order[oj];
                if (pos_in_order[q] <= waited[cs * ns + ps]) continue;
                waited[cs * ns + ps] = pos_in_order[q];
                targets[oi].push_back(q);
                waiters[q]++;
            }
        }
        for (size_t oi = 0; oi < n; ++oi)
            for (int q : targets[oi]) {
                if (event_of[q] < 0) {
                    hipEvent_t e;
                    if (hipEventCreateWithFlags(&e, graphx_event_flags()) != hipSuccess) { asr_set_error("graphx: event"); return -2; }
                    event_of[q] = (int)g->events.size();
                    g->events.push_back(e);
                    g->nodes[pos_in_order[q]].record_event = event_of[q];
                }
                g->nodes[oi].wait_events.push_back(event_of[q]);
            }
    }
    // every side stream is joined into the launch stream behind its last node
    g->tail_event.assign(g->streams.size(), -1);
    for (size_t s = 1; s < g->streams.size(); ++s) {
        const int t = tail_of_stream[s];
        if (t < 0) continue;
        if (event_of[t] < 0) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, graphx_event_flags()) != hipSuccess) { asr_set_error("graphx: event"); return -2; }
            event_of[t] = (int)g->events.size();
            g->events.push_back(e);
            g->nodes[pos_in_order[t]].record_event = event_of[t];
        }
        g->tail_event[s] = event_of[t];
    }
    if (hipEventCreateWithFlags(&g->begin, graphx_event_flags()) != hipSuccess) { asr_set_error("graphx: event"); return -2; }
    if (getenv("ASR_AMD_GRAPHX_DEBUG")) {      // the launch plan, one line per node: position, stream, type / kernel name, waits, record
        for (size_t oi = 0; oi < n; ++oi) {
            const XNode& x = g->nodes[oi];
            const char* name = x.type == hipGraphNodeTypeKernel ? hipKernelNameRefByPtr(x.k.func, nullptr)
                               : (x.type == hipGraphNodeTypeMemset ? "<memset>" : (x.type == hipGraphNodeTypeMemcpy ? "<memcpy>" : "<empty>"));
            char dl[160];
            int off = 0;
            dl[0] = 0;
            for (int p : deps[order[oi]])
                if (off < 140) off += snprintf(dl + off, sizeof(dl) - off, " %d", pos_in_order[p]);
            fprintf(stderr, "graphx %4zu s%d %.90s waits %zu rec %d deps%s\n", oi, x.stream, name ? name : "?", x.wait_events.size(), x.record_event, dl);
        }
    }
    guard.g = nullptr;      // (built: the caller owns it now)
    *out_handle = g;
    return 0;
}

extern "C" int asr_graphx_set_collective(void* handle, void* rccl_comm, asr_collective_fn fn, void* ctx) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g, -1, "graphx_set_collective: null handle");
    g->comm = rccl_comm;
    g->coll_fn = fn;
    g->coll_ctx = ctx;
    return 0;
}

extern "C" int asr_graphx_collectives(void* handle, int* n_collective, long long* total_count) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g, -1, "graphx_collectives: null handle");
    long long tot = 0;
    for (const XNode& x : g->nodes)
        if (x.collective) tot += x.ccount;
    if (n_collective) *n_collective = g->n_collective;
    if (total_count) *total_count = tot;
    return 0;
}

extern "C" int asr_graphx_info(void* handle, int* n_nodes, int* n_kernels, int* n_streams, int* n_events) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g, -1, "graphx_info: null handle");
    if (n_nodes) *n_nodes = (int)g->nodes.size();
    if (n_kernels) *n_kernels = g->n_kernel;
    if (n_streams) *n_streams = (int)g->streams.size();
    if (n_events) *n_events = (int)g->events.size();
    return 0;
}

// The HIP runtime multiplexes streams onto a few hardware queues (4 by default; more made every configuration slower here) and two
// streams on one queue do not overlap: whether the executor's busiest side chain (the weight gradients) shares the launch stream's
// queue was luck - S2 replayed in 6.6 or 10.0 ms depending on which stream index the chain had been given.  The plan's logical
// streams can be rotated over the physical ones; Trainer.step_auto times the rotations and keeps the best.
extern "C" int asr_graphx_set_rotation(void* handle, int rotation) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g && rotation >= 0, -1, "graphx_set_rotation: bad arguments");
    const int nside = (int)g->streams.size() - 1;
    g->rotation = nside > 0 ? rotation % nside : 0;
    return 0;
}

extern "C" int asr_streams_share_queue(void* stream_a, void* stream_b, int* shared);

// Place the plan's side streams by what the hardware queues look like (asr_streams_share_queue on every pair, ~30 probes of ~0.1 ms):
// logical streams in order of their kernel count go, heaviest first, to the physical stream whose queue carries the least so far -
// the launch stream's queue starts with the main chain's own weight, so it is taken last.  clear != 0: back to the rotation.
extern "C" int asr_graphx_place_streams(void* handle, void* launch_stream, int clear) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g, -1, "graphx_place_streams: null handle");
    g->map.clear();
    const int ns = (int)g->streams.size();
    if (clear || ns <= 2) return 0;
    // queue groups of the physical side streams: group 0 = the launch stream's queue
    std::vector<int> group(ns, -1);
    int ngroups = 1;
    for (int p = 1; p < ns; ++p) {
        int sh = 0;
        if (int rc = asr_streams_share_queue(launch_stream, g->streams[p], &sh)) return rc;
        if (sh) { group[p] = 0; continue; }
        for (int q = 1; q < p && group[p] < 0; ++q) {
            if (group[q] <= 0) continue;
            if (int rc = asr_streams_share_queue(g->streams[q], g->streams[p], &sh)) return rc;
            if (sh) group[p] = group[q];
        }
        if (group[p] < 0) group[p] = ngroups++;
    }
    std::vector<long> weight(ns, 0), load(ngroups, 0);
    for (const XNode& x : g->nodes) weight[x.stream] += 1;
    load[0] = weight[0];
    std::vector<int> logical(ns - 1);
    for (int i = 1; i < ns; ++i) logical[i - 1] = i;
    std::sort(logical.begin(), logical.end(), [&](int a, int b) { return weight[a] > weight[b]; });
    std::vector<char> used(ns, 0);
    std::vector<int> map(ns, 0);
    for (int l : logical) {
        int best = -1;
        for (int p = 1; p < ns; ++p)
            if (!used[p] && (best < 0 || load[group[p]] < load[group[best]])) best = p;
        used[best] = 1;
        map[l] = best;
        load[group[best]] += weight[l];
    }
    g->map = map;
    if (getenv("ASR_AMD_GRAPHX_DEBUG")) {
        for (int l = 1; l < ns; ++l) fprintf(stderr, "graphx place: logical stream %d (%ld nodes) -> physical %d (queue group %d)\n", l, weight[l], map[l], group[map[l]]);
    }
    return 0;
}

static int graphx_launch_nodes(GraphX* g, hipStream_t main);

extern "C" int asr_graphx_launch(void* handle, void* stream) {
    GraphX* g = static_cast<GraphX*>(handle);
    ASR_REQUIRE(g, -1, "graphx_launch: null handle");
    hipStream_t main = static_cast<hipStream_t>(stream);
    g->streams[0] = main;
    // A plan with collective nodes and nothing to run them is refused BEFORE anything is queued; any later failure (a launch the
    // runtime rejects, a collective that errors) leaves a half-queued step: the side streams are drained then, so that the caller's
    // stream order still covers everything that was queued, and the error is returned.
    if (g->n_collective > 0 && !g->coll_fn && !g->comm) {
        asr_set_error("graphx_launch: the step holds %d gradient all-reduce node(s) and no communicator (asr_graphx_set_collective)",
                      g->n_collective);
        return -5;
    }
    const int rc = graphx_launch_nodes(g, main);
    if (rc != 0) {
        // A step with collective nodes that failed part-way is fatal for the JOB, not only for this rank: the peers have queued (or will
        // queue) the matching ncclAllReduce calls and would wait for this rank forever.  The communicator is BORROWED (its owner gave
        // it to asr_graphx_set_collective): it is not freed here - the executor forgets it and returns ASR_ERR_COLLECTIVE_STEP, on which
        // the owner aborts it (asr_rccl_comm_abort: peers fail fast with an asynchronous error) and drops its own handle.
        bool collective_step = false;
        if (g->n_collective > 0 && g->comm) {
            const std::string first = asr_last_error();
            g->comm = nullptr;
            collective_step = true;
            asr_set_error("%s [a step with RCCL all-reduce nodes failed part-way: abort the communicator (asr_rccl_comm_abort) so that peers fail "
                          "fast instead of waiting in an all-reduce this rank never joins]", first.c_str());
        }
        for (size_t s = 1; s < g->streams.size(); ++s) (void)hipStreamSynchronize(g->streams[s]);
        if (collective_step) return ASR_ERR_COLLECTIVE_STEP;
        return rc;
    }
    return rc;
}

static int graphx_launch_nodes(GraphX* g, hipStream_t main) {
    // side streams start behind everything already queued on the launch stream
    if (g->streams.size() > 1) {
        GX_CHECK(hipEventRecord(g->begin, main));
        for (size_t s = 1; s < g->streams.size(); ++s) GX_CHECK(hipStreamWaitEvent(g->streams[s], g->begin, 0));
    }
    const int nside = (int)g->streams.size() - 1;
    for (XNode& x : g->nodes) {
        // logical side stream i runs on physical side stream (i + rotation) mod nside: which streams share a hardware queue with the
        // launch stream is the runtime's choice (asr_graphx_set_rotation)
        hipStream_t st = x.stream == 0 ? main
                         : g->streams[g->map.empty() ? 1 + (x.stream - 1 + g->rotation) % nside : g->map[x.stream]];
        for (int e : x.wait_events) GX_CHECK(hipStreamWaitEvent(st, g->events[e], 0));
        switch (x.type) {
            case hipGraphNodeTypeKernel:
                if (x.collective) {
                    // the bucket's gradients are final in this stream's order (the waits above): sum them over the ranks right here
                    int rc;
                    if (g->coll_fn) rc = g->coll_fn(g->coll_ctx, x.cbuf, x.ccount, x.ctag, st);
                    else {
                        ASR_REQUIRE(g->comm, -5, "graphx_launch: the step holds %d gradient all-reduce node(s) and no communicator "
                                                 "(asr_graphx_set_collective)", g->n_collective);
                        rc = asr_rccl_all_reduce_f32(g->comm, x.cbuf, x.ccount, st);
                    }
                    if (rc != 0) return rc;
                    break;
                }
                GX_CHECK(hipLaunchKernel(x.k.func, x.k.gridDim, x.k.blockDim, x.k.kernelParams, x.k.sharedMemBytes, st));
                break;
            case hipGraphNodeTypeMemset: {
                const hipMemsetParams& m = x.ms;
                if (m.height <= 1) {
                    if (m.elementSize == 4) GX_CHECK(hipMemsetD32Async((hipDeviceptr_t)m.dst, (int)m.value, m.width, st));
                    else if (m.elementSize == 2) GX_CHECK(hipMemsetD16Async((hipDeviceptr_t)m.dst, (unsigned short)m.value, m.width, st));
                    else GX_CHECK(hipMemsetD8Async((hipDeviceptr_t)m.dst, (unsigned char)m.value, m.width, st));
                } else {
                    ASR_REQUIRE(m.elementSize == 1 || m.value == 0, -4, "graphx_launch: 2-D memset of %u-byte elements", m.elementSize);
                    GX_CHECK(hipMemset2DAsync(m.dst, m.pitch, (int)m.value, m.width * m.elementSize, m.height, st));
                }
                break;
            }
            case hipGraphNodeTypeMemcpy:
                GX_CHECK(hipGraphLaunch(x.sub, st));
                break;
            default:
                break;      // empty node: its edges are events already
        }
        if (x.record_event >= 0) GX_CHECK(hipEventRecord(g->events[x.record_event], st));
    }
    for (size_t s = 1; s < g->streams.size(); ++s)
        if (g->tail_event[s] >= 0) GX_CHECK(hipStreamWaitEvent(main, g->events[g->tail_event[s]], 0));
    return 0;
}
