// context.hip -- LeleCtx / LeleBuf: stream, staging arena, weight cache, timers.
#include "common.h"

#include <atomic>
#include <tuple>

namespace lele {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace lele

using namespace lele;

int LeleCtx::arena_reset() {
    arena_used = 0;
    if (!arena_overflow.empty()) {
        LELE_REQUIRE(!capturing, "graph capture: the staging arena overflowed in the op before capture began; sync first");
        // the previous op's kernels may still read the overflow blocks: drain before freeing
        LELE_HIP_CHECK(hipStreamSynchronize(stream));
        for (void* p : arena_overflow) (void)hipFree(p);
        arena_overflow.clear();
        ++generation;
    }
    return 0;
}

/* park the current lane's arena / scratch / temporaries and take lane `to`'s (streams and capture state are the caller's business) */
void LeleCtx::swap_lane_memory(int to) {
    if (parked.size() < (size_t)kMaxLanes) parked.resize(kMaxLanes);
    LaneState& cur = parked[lane];
    LaneState& dst = parked[to];
    cur.arena = arena, cur.arena_cap = arena_cap, cur.arena_used = arena_used;
    cur.arena_overflow.swap(arena_overflow);
    cur.scratch = scratch, cur.scratch_cap = scratch_cap;
    for (int i = 0; i < 3; ++i) cur.tmp[i] = tmp[i];
    arena = dst.arena, arena_cap = dst.arena_cap, arena_used = dst.arena_used;
    arena_overflow.clear();
    arena_overflow.swap(dst.arena_overflow);
    scratch = dst.scratch, scratch_cap = dst.scratch_cap;
    for (int i = 0; i < 3; ++i) tmp[i] = dst.tmp[i];
    lane = to;
}

int LeleCtx::sync_all() {
    LELE_HIP_CHECK(hipStreamSynchronize(stream));
    for (int l = 0; l < kMaxLanes; ++l)
        if (lane_stream[l] && lane_stream[l] != stream) LELE_HIP_CHECK(hipStreamSynchronize(lane_stream[l]));
    if (lane == 0 && !capturing) side_lanes = false;
    return 0;
}

namespace {
std::atomic<int> g_live_ctx[64];
}
namespace lele {
int live_contexts(int device) { return device >= 0 && device < 64 ? g_live_ctx[device].load() : 2; }
}

int LeleCtx::capture_deps_get(std::vector<hipGraphNode_t>* out) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const hipGraphNode_t* deps = nullptr;
    size_t n = 0;
    LELE_HIP_CHECK(hipStreamGetCaptureInfo_v2(stream, &st, nullptr, nullptr, &deps, &n));
    LELE_REQUIRE(st == hipStreamCaptureStatusActive, "lanes: the capture is no longer active (an op invalidated it)");
    out->assign(deps, deps + n);
    return 0;
}

int LeleCtx::tmp_buf(int i, LeleBuf** out) {
    if (!tmp[i]) {
        LELE_REQUIRE(!capturing, "graph capture: this op needs a temporary buffer; run the sequence once before capturing it");
        tmp[i] = new LeleBuf();
        tmp[i]->ctx = this;
    }
    *out = tmp[i];
    return 0;
}

int LeleCtx::check_deverr(const char* where) {
    if (deverr_host && *deverr_host) {
        const unsigned bits = *deverr_host;
        *deverr_host = 0;
        LELE_REQUIRE(false, "%s: a kernel reported a data-dependent violation earlier on this stream:%s (the access was clamped; "
                     "lele's bounds-checked indexing panics here, manipulation.rs:626-633)", where,
                     (bits & LELE_DEVERR_GATHER_INDEX) ? " gather index out of range"
                     : (bits & LELE_DEVERR_FFN_SYNC) ? " a workgroup of the one-launch feed-forward block waited 20 ms for its neighbours -- another "
                                                        "process is using the device; set LELE_HIP_FFN_ONE_LAUNCH=0 (results of that launch are wrong)"
                                                      : " unknown");
    }
    return 0;
}

int LeleCtx::arena_alloc(size_t bytes, void** out) {
    size_t need = (bytes + 255) & ~size_t(255);
    if (arena_used + need <= arena_cap) {
        *out = arena + arena_used;
        arena_used += need;
        return 0;
    }
    LELE_REQUIRE(!capturing, "graph capture: an op needs %zu bytes of scratch beyond the %zu-byte arena", need, arena_cap);
    void* p = nullptr;
    LELE_HIP_CHECK(hipMalloc(&p, need ? need : 256));
    arena_overflow.push_back(p);
    *out = p;
    return 0;
}

int LeleCtx::get_scratch(size_t bytes, void** out) {
    if (bytes > scratch_cap) {
        LELE_REQUIRE(!capturing, "graph capture: scratch would grow; run the sequence once before capturing it");
        LELE_HIP_CHECK(hipStreamSynchronize(stream));
        if (scratch) {
            (void)hipFree(scratch);
            ++generation;
        }
        scratch = nullptr;
        scratch_cap = 0;
        size_t cap = (bytes + (1 << 20)) & ~size_t((1 << 20) - 1);
        LELE_HIP_CHECK(hipMalloc(&scratch, cap));
        scratch_cap = cap;
    }
    *out = scratch;
    return 0;
}

int LeleCtx::dev_ptr(const LeleTensor* t, const void** out) {
    if (!t) {
        *out = nullptr;
        return 0;
    }
    size_t bytes = (size_t)numel(t) * dtype_size(t->dtype);
    if (t->mem == LELE_MEM_DEVICE || bytes == 0) {
        *out = t->data;
        return 0;
    }
    if (t->mem == LELE_MEM_WEIGHT) {
        auto key = std::make_tuple(t->data, bytes, 0);
        auto it = weights.find(key);
        if (it != weights.end()) {
            *out = it->second;
            return 0;
        }
        LELE_REQUIRE(!capturing, "graph capture: weight %p is not cached yet; run the sequence once before capturing it", t->data);
        void* d = nullptr;
        LELE_HIP_CHECK(hipMalloc(&d, bytes));
        LELE_HIP_CHECK(hipMemcpyAsync(d, t->data, bytes, hipMemcpyHostToDevice, stream));
        LELE_HIP_CHECK(hipStreamSynchronize(stream));
        weights[key] = d;
        *out = d;
        return 0;
    }
    LELE_REQUIRE(!capturing, "graph capture: LELE_MEM_HOST inputs cannot be captured; pass device tensors or weights");
    void* d = nullptr;
    LELE_TRY(arena_alloc(bytes, &d));
    // pageable host memory: hipMemcpyAsync stages it before returning, so the caller may reuse `data` at once
    LELE_HIP_CHECK(hipMemcpyAsync(d, t->data, bytes, hipMemcpyHostToDevice, stream));
    *out = d;
    return 0;
}

int LeleBuf::reserve(size_t n) {
    rowstat_valid = false;  // the buffer is about to be written
    if (n > cap) {
        LELE_REQUIRE(!ctx->capturing, "graph capture: an output buffer would grow; run the sequence once before capturing it");
        LELE_TRY(ctx->sync_all());
        if (data) {
            ctx->buf_of_data.erase(data);
            (void)hipFree(data);
            ++ctx->generation;
        }
        data = nullptr;
        cap = 0;
        size_t c = (n + 4095) & ~size_t(4095);
        LELE_HIP_CHECK(hipMalloc(&data, c));
        cap = c;
        ctx->buf_of_data[data] = this;
    }
    bytes = n;
    return 0;
}

int LeleBuf::reserve_rowstat(int64_t rows) {
    if ((size_t)rows <= rowstat_cap) return 0;
    if (ctx->capturing) return 0;  // no allocation while recording: the producer simply does not publish statistics
    LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (rowstat) {
        (void)hipFree(rowstat);
        ++ctx->generation;
    }
    rowstat = nullptr;
    rowstat_cap = 0;
    const size_t c = ((size_t)rows + 511) & ~size_t(511);
    LELE_HIP_CHECK(hipMalloc((void**)&rowstat, c * 8));
    rowstat_cap = c;
    return 0;
}

extern "C" {

const char* lele_hip_last_error(void) { return g_err.c_str(); }

int lele_hip_device_count(int* count) {
    LELE_HIP_CHECK(hipGetDeviceCount(count));
    return 0;
}

int lele_hip_ctx_create(int device, LeleCtx** out) {
    LELE_REQUIRE(out != nullptr, "ctx_create: out is NULL");
    LELE_HIP_CHECK(hipSetDevice(device));
    LeleCtx* c = new LeleCtx();
    c->device = device;
    hipDeviceProp_t prop;
    LELE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    LELE_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->lane_stream[0] = c->stream;
    LELE_HIP_CHECK(hipEventCreate(&c->ev0));
    LELE_HIP_CHECK(hipEventCreate(&c->ev1));
    c->arena_cap = size_t(64) << 20;
    LELE_HIP_CHECK(hipMalloc((void**)&c->arena, c->arena_cap));
    LELE_HIP_CHECK(hipHostMalloc((void**)&c->deverr_host, 64, hipHostMallocMapped));
    *c->deverr_host = 0;
    LELE_HIP_CHECK(hipHostGetDevicePointer((void**)&c->deverr_dev, c->deverr_host, 0));
    if (device >= 0 && device < 64) ++g_live_ctx[device];
    *out = c;
    return 0;
}

int lele_hip_ctx_destroy(LeleCtx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    while (!c->graphs.empty()) (void)lele_hip_graph_destroy(c->graphs.back());  // graphs hold raw addresses of this ctx's memory
    while (!c->comms.empty()) (void)lele_hip_comm_destroy(c->comms.back());     // communicators issue on this ctx's stream
    for (hipEvent_t e : c->qprof.ev) (void)hipEventDestroy(e);
    if (c->lane != 0) (void)lele_hip_lane_set(c, 0);
    for (LeleBuf* t : c->tmp)
        if (t) (void)lele_hip_buf_destroy(t);
    for (size_t l = 1; l < c->parked.size(); ++l) {
        LeleCtx::LaneState& s = c->parked[l];
        if (!c->lane_stream[l]) continue;
        (void)hipStreamSynchronize(c->lane_stream[l]);
        for (LeleBuf* t : s.tmp)
            if (t) (void)lele_hip_buf_destroy(t);
        for (void* p : s.arena_overflow) (void)hipFree(p);
        if (s.arena) (void)hipFree(s.arena);
        if (s.scratch) (void)hipFree(s.scratch);
        (void)hipStreamDestroy(c->lane_stream[l]);
    }
    for (hipEvent_t e : c->lane_events)
        if (e) (void)hipEventDestroy(e);
    if (c->deverr_host) (void)hipHostFree(c->deverr_host);
    for (auto& kv : c->rs_sync) (void)hipFree(kv.second);
    if (c->device >= 0 && c->device < 64) --g_live_ctx[c->device];
    for (void* p : c->arena_overflow) (void)hipFree(p);
    for (auto& kv : c->weights) (void)hipFree(kv.second);
    if (c->arena) (void)hipFree(c->arena);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->mailbox) (void)hipHostFree(c->mailbox);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int lele_hip_sync(LeleCtx* c) {
    LELE_REQUIRE(c, "sync: ctx is NULL");
    LELE_REQUIRE(!c->capturing, "sync: not allowed while a graph is being captured");
    LELE_TRY(c->sync_all());
    return c->check_deverr("sync");
}

void* lele_hip_ctx_stream(LeleCtx* c) { return c ? (void*)c->stream : nullptr; }

/* ---- hipGraph capture of an op sequence -------------------------------------------------------------------------- */
int lele_hip_graph_begin(LeleCtx* c) {
    LELE_REQUIRE(c, "graph_begin: ctx is NULL");
    LELE_REQUIRE(!c->capturing, "graph_begin: a capture is already in progress");
    LELE_REQUIRE(c->lane == 0, "graph_begin: lane %d is current; a capture starts (and ends) on lane 0", c->lane);
    LELE_HIP_CHECK(hipSetDevice(c->device));
    LELE_TRY(c->sync_all());
    LELE_TRY(c->arena_reset());  // frees any overflow blocks now, while synchronising is still legal
    for (auto& ps : c->parked) {  // ... and every parked lane's (all streams are drained): its first captured op could not free them
        for (void* p : ps.arena_overflow) (void)hipFree(p);
        if (!ps.arena_overflow.empty()) ++c->generation;
        ps.arena_overflow.clear();
        ps.arena_used = 0;
    }
    LELE_HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    for (auto& t : c->lane_tail) t.clear();
    c->event_nodes.clear();
    return 0;
}
int lele_hip_graph_end(LeleCtx* c, LeleGraph** out) {
    LELE_REQUIRE(c && out, "graph_end: NULL argument");
    LELE_REQUIRE(c->capturing, "graph_end: no capture in progress");
    LELE_REQUIRE(c->lane == 0, "graph_end: lane %d is current; switch back to lane 0 (lele_hip_lane_set) before ending the capture", c->lane);
    {   // the end of the graph = the end of every lane: whatever follows the launch on the stream is ordered after all of them anyway
        // (a graph launch completes when its last node does), the explicit join only keeps the capture's dependency set whole
        std::vector<hipGraphNode_t> all;
        for (int l = 1; l < LeleCtx::kMaxLanes; ++l) all.insert(all.end(), c->lane_tail[l].begin(), c->lane_tail[l].end());
        if (!all.empty()) LELE_HIP_CHECK(hipStreamUpdateCaptureDependencies(c->stream, all.data(), all.size(), hipStreamAddCaptureDependencies));
    }
    c->capturing = false;
    hipGraph_t g = nullptr;
    LELE_HIP_CHECK(hipStreamEndCapture(c->stream, &g));
    LELE_REQUIRE(g != nullptr, "graph_end: the capture was invalidated (an op allocated or synchronised)");
    hipGraphExec_t e = nullptr;
    hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    LELE_HIP_CHECK(rc);
    LeleGraph* lg = new LeleGraph();
    lg->ctx = c;
    lg->exec = e;
    lg->generation = c->generation;
    c->graphs.push_back(lg);
    *out = lg;
    return 0;
}
int lele_hip_graph_abort(LeleCtx* c) {  // leave capture mode after a failed op, discarding what was recorded
    LELE_REQUIRE(c, "graph_abort: ctx is NULL");
    if (!c->capturing) return 0;
    c->capturing = false;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(c->stream, &g);   // c->stream is lane 0's stream for the whole capture (lane_set does not switch it)
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    if (c->lane != 0) c->swap_lane_memory(0);   // the failed op may have run on a side lane: lane 0's arena / scratch / temporaries back
    for (auto& t : c->lane_tail) t.clear();
    c->event_nodes.clear();
    return 0;
}
int lele_hip_graph_launch(LeleGraph* g) {
    LELE_REQUIRE(g && g->exec, "graph_launch: graph is NULL");
    LELE_REQUIRE(!g->ctx->capturing, "graph_launch: not allowed while capturing");
    LELE_REQUIRE(g->generation == g->ctx->generation,
                 "graph_launch: device memory of this ctx was re-allocated after the graph was recorded (a buffer, the scratch "
                 "block or a row-statistics block grew): the recorded addresses are stale -- record the graph again");
    LELE_HIP_CHECK(hipGraphLaunch(g->exec, g->ctx->stream));
    return 0;
}
int lele_hip_graph_destroy(LeleGraph* g) {
    if (!g) return 0;
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    auto& v = g->ctx->graphs;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == g) {
            v.erase(v.begin() + i);
            break;
        }
    delete g;
    return 0;
}

/* ---- lanes ------------------------------------------------------------------------------------------------------------ */
int lele_hip_lane_set(LeleCtx* c, int lane) {
    LELE_REQUIRE(c, "lane_set: ctx is NULL");
    LELE_REQUIRE(lane >= 0 && lane < LeleCtx::kMaxLanes, "lane_set: lane %d outside [0, %d)", lane, LeleCtx::kMaxLanes);
    if (lane == c->lane) return 0;
    if (lane != 0) c->side_lanes = true;
    LELE_HIP_CHECK(hipSetDevice(c->device));
    if (c->parked.size() < (size_t)LeleCtx::kMaxLanes) c->parked.resize(LeleCtx::kMaxLanes);
    LeleCtx::LaneState& dst = c->parked[lane];
    if (!c->lane_stream[lane]) {  // first use: its own stream and staging arena
        LELE_REQUIRE(!c->capturing, "graph capture: lane %d does not exist yet; run the sequence once (with its lanes) before capturing it", lane);
        LELE_HIP_CHECK(hipStreamCreateWithFlags(&c->lane_stream[lane], hipStreamNonBlocking));
        dst.arena_cap = c->arena_cap;
        LELE_HIP_CHECK(hipMalloc((void**)&dst.arena, dst.arena_cap));
    }
    // where the lane we leave stands -- unless an op has invalidated the capture: then there is nothing to remember, and the switch must
    // still succeed so that a caller's clean-up (`finally: lane_set(0)`, lele_hip_graph_abort) does not mask the original error
    bool live = c->capturing;
    if (live) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &st) != hipSuccess || st != hipStreamCaptureStatusActive) live = false, (void)hipGetLastError();
    }
    if (live) LELE_TRY(c->capture_deps_get(&c->lane_tail[c->lane]));
    c->swap_lane_memory(lane);
    if (live) {  // one stream records everything: the next node hangs off THIS lane's tail (nothing yet: a root of the graph)
        std::vector<hipGraphNode_t>& t = c->lane_tail[lane];
        LELE_HIP_CHECK(hipStreamUpdateCaptureDependencies(c->stream, t.empty() ? nullptr : t.data(), t.size(), hipStreamSetCaptureDependencies));
    } else if (!c->capturing) {
        c->stream = c->lane_stream[lane];
    }
    return 0;
}
int lele_hip_lane_record(LeleCtx* c, int event) {
    LELE_REQUIRE(c && event >= 0 && event < (1 << 16), "lane_record: bad argument");
    LELE_HIP_CHECK(hipSetDevice(c->device));
    if (c->capturing) {
        if (c->event_nodes.size() <= (size_t)event) c->event_nodes.resize((size_t)event + 1);
        return c->capture_deps_get(&c->event_nodes[event]);
    }
    if (c->lane_events.size() <= (size_t)event) c->lane_events.resize((size_t)event + 1, nullptr);
    if (!c->lane_events[event]) LELE_HIP_CHECK(hipEventCreateWithFlags(&c->lane_events[event], hipEventDisableTiming));
    LELE_HIP_CHECK(hipEventRecord(c->lane_events[event], c->stream));
    return 0;
}
int lele_hip_lane_wait(LeleCtx* c, int event) {
    LELE_REQUIRE(c && event >= 0, "lane_wait: bad argument");
    LELE_HIP_CHECK(hipSetDevice(c->device));
    if (c->capturing) {
        LELE_REQUIRE((size_t)event < c->event_nodes.size(), "lane_wait: event %d was not recorded in this capture", event);
        std::vector<hipGraphNode_t>& nodes = c->event_nodes[event];
        if (!nodes.empty()) LELE_HIP_CHECK(hipStreamUpdateCaptureDependencies(c->stream, nodes.data(), nodes.size(), hipStreamAddCaptureDependencies));
        return 0;
    }
    LELE_REQUIRE((size_t)event < c->lane_events.size() && c->lane_events[event], "lane_wait: event %d was never recorded", event);
    LELE_HIP_CHECK(hipStreamWaitEvent(c->stream, c->lane_events[event], 0));
    return 0;
}

int lele_hip_timer_start(LeleCtx* c) {
    LELE_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    return 0;
}
int lele_hip_timer_stop(LeleCtx* c, float* ms) {
    LELE_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    LELE_HIP_CHECK(hipEventSynchronize(c->ev1));
    LELE_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

int lele_hip_buf_create(LeleCtx* c, LeleBuf** out) {
    LELE_REQUIRE(c && out, "buf_create: NULL argument");
    LeleBuf* b = new LeleBuf();
    b->ctx = c;
    *out = b;
    return 0;
}
int lele_hip_buf_destroy(LeleBuf* b) {
    if (!b) return 0;
    if (b->data) {
        (void)hipStreamSynchronize(b->ctx->stream);
        b->ctx->buf_of_data.erase(b->data);
        (void)hipFree(b->data);
        ++b->ctx->generation;
    }
    if (b->rowstat) {  // a recorded graph may have baked this pointer in (quant.hip reads the pairs): it must not replay either
        if (!b->data) (void)hipStreamSynchronize(b->ctx->stream);
        (void)hipFree(b->rowstat);
        ++b->ctx->generation;
    }
    delete b;
    return 0;
}
int lele_hip_buf_reserve(LeleBuf* b, size_t bytes) {
    LELE_REQUIRE(b, "buf_reserve: buf is NULL");
    return b->reserve(bytes);
}
void* lele_hip_buf_data(LeleBuf* b) { return b ? b->data : nullptr; }
int lele_hip_buf_mark_dirty(LeleBuf* b) {
    LELE_REQUIRE(b, "buf_mark_dirty: buf is NULL");
    b->rowstat_valid = false;
    return 0;
}
size_t lele_hip_buf_bytes(LeleBuf* b) { return b ? b->bytes : 0; }
int lele_hip_buf_from_host(LeleBuf* b, const void* src, size_t bytes) {
    LELE_REQUIRE(b, "buf_from_host: buf is NULL");
    LELE_TRY(b->reserve(bytes));
    if (bytes) {
        LELE_HIP_CHECK(hipMemcpyAsync(b->data, src, bytes, hipMemcpyHostToDevice, b->ctx->stream));
        LELE_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));
    }
    return 0;
}
int lele_hip_buf_to_host(LeleBuf* b, void* dst, size_t bytes) {
    LELE_REQUIRE(b, "buf_to_host: buf is NULL");
    LELE_REQUIRE(bytes <= b->bytes, "buf_to_host: %zu bytes requested, result holds %zu", bytes, b->bytes);
    LeleCtx* c = b->ctx;
    if (bytes && bytes <= LeleCtx::kMailboxBytes) {   // small read: through the context's page-locked mailbox
        if (!c->mailbox) {
            LELE_REQUIRE(!c->capturing, "buf_to_host: not allowed while a graph is being captured");
            LELE_HIP_CHECK(hipHostMalloc(&c->mailbox, LeleCtx::kMailboxBytes, hipHostMallocDefault));
        }
        LELE_HIP_CHECK(hipMemcpyAsync(c->mailbox, b->data, bytes, hipMemcpyDeviceToHost, c->stream));
        LELE_HIP_CHECK(hipStreamSynchronize(c->stream));
        memcpy(dst, c->mailbox, bytes);
        return c->check_deverr("buf_to_host");
    }
    if (bytes) LELE_HIP_CHECK(hipMemcpyAsync(dst, b->data, bytes, hipMemcpyDeviceToHost, c->stream));
    LELE_HIP_CHECK(hipStreamSynchronize(c->stream));
    return c->check_deverr("buf_to_host");
}

}  // extern "C"
