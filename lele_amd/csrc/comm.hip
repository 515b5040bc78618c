// comm.hip -- the one exchange step of the sharded recogniser, owned by the C ABI.
//
// lele itself is single-process; its only route to several devices is "one instance per utterance shard".  Utterances are
// independent (dynamic quantisation and CMVN are per utterance), so a GPU never needs another GPU's data while it computes.
// What is exchanged, once, at the very end, is the DECODED token ids (i32, count-prefixed rows; 22 KB per GPU for
// BASELINE configs[3]) so that every rank holds the transcripts of the whole batch: one RCCL all-gather over xGMI, issued on
// the ctx stream straight from the device buffer the greedy decoder wrote -- no host round trip, no torch.
//
// RCCL is bound at run time (dlopen "librccl.so"): the library itself has no link-time dependency on it, a single-GPU
// integration never loads it, and a missing RCCL is an error of lele_hip_comm_* only.  One process per GPU; the 128-byte
// unique id is created by rank 0 and handed to the other ranks by whatever launched them -- lele_hip_comm_init_file does that
// through a file (rank 0 writes <path>.tmp and renames it; the others poll), which needs neither MPI nor torch.
#include "common.h"
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>

#include <dlfcn.h>
#include <errno.h>
#include <time.h>
#include <unistd.h>

using namespace lele;

namespace {

struct UniqueId {
    char internal[128];  // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
};
typedef void* Comm;
// rccl.h:52 ncclSuccess = 0; :448-450 ncclMax = 2; :459-463 ncclInt32 = 2, ncclInt64 = 4
constexpr int kUint8 = 1, kInt32 = 2, kInt64 = 4, kMax = 2;  // ncclDataType_t / ncclRedOp_t values

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

int load_rccl(Rccl** out) {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.handle) {
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        LELE_REQUIRE(h, "comm: librccl.so cannot be loaded (%s)", dlerror());
#define LELE_SYM(field, sym)                                                         \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));                    \
    LELE_REQUIRE(r.field, "comm: librccl.so does not export %s", sym)
        LELE_SYM(GetUniqueId, "ncclGetUniqueId");
        LELE_SYM(CommInitRank, "ncclCommInitRank");
        LELE_SYM(CommDestroy, "ncclCommDestroy");
        LELE_SYM(AllGather, "ncclAllGather");
        LELE_SYM(AllReduce, "ncclAllReduce");
        LELE_SYM(GetErrorString, "ncclGetErrorString");
#undef LELE_SYM
        r.handle = h;
    }
    *out = &r;
    return 0;
}

#define LELE_RCCL_CHECK(api, expr)                                                                   \
    do {                                                                                             \
        int _rc = (expr);                                                                            \
        LELE_REQUIRE(_rc == 0, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_rc), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct LeleComm {
    LeleCtx* ctx = nullptr;
    Rccl* api = nullptr;
    Comm comm = nullptr;
    int rank = 0, world = 1;
    void* scratch = nullptr;  // 64 bytes of device memory for the scalar reductions
};

extern "C" {

int lele_hip_comm_unique_id(uint8_t* id128) {
    LELE_REQUIRE(id128, "comm_unique_id: NULL argument");
    Rccl* api = nullptr;
    LELE_TRY(load_rccl(&api));
    UniqueId id;
    LELE_RCCL_CHECK(api, api->GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

int lele_hip_comm_init(LeleCtx* ctx, const uint8_t* id128, int rank, int world, LeleComm** out) {
    LELE_REQUIRE(ctx && id128 && out, "comm_init: NULL argument");
    LELE_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    Rccl* api = nullptr;
    LELE_TRY(load_rccl(&api));
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    LeleComm* c = new LeleComm();
    c->ctx = ctx;
    c->api = api;
    c->rank = rank;
    c->world = world;
    int rc = api->CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(rc));
        delete c;
        return 2;
    }
    if (hipMalloc(&c->scratch, 64) != hipSuccess) {
        set_error("comm_init: hipMalloc failed");
        (void)api->CommDestroy(c->comm);
        delete c;
        return 1;
    }
    ctx->comms.push_back(c);
    *out = c;
    return 0;
}

// The rendezvous file is [32-byte job token][128-byte id].  The token is LELE_JOB_ID (else TORCHELASTIC_RUN_ID, else empty), which
// whoever launches the ranks sets to something unique per launch (lele_run: its pid and start time): a reader only accepts a file that
// carries ITS token, so a file left behind by an earlier job is never taken for the current one, however young, and a valid one is
// never refused, however late the reader arrives (no wall-clock test: clocks on a shared file system differ).  Rank 0 removes the
// path before it writes.
//
// Without any token in the environment (a launcher that tells its ranks nothing) the all-zero token cannot tell this job's file from
// one an earlier token-less job left under the same name, and a reader that took a dead id would hang in ncclCommInitRank.  No
// clock can decide that either (a reader may arrive many seconds after rank 0 wrote; st_mtime is the file server's clock),
// so the file proves that its writer is ALIVE instead: a token-less file is [32 zero bytes][128-byte id][8-byte beat], rank 0 rewrites
// it with the beat incremented every 20 ms for as long as it sits in ncclCommInitRank (that call returns only once every rank has
// joined, i.e. has read the file) and removes it afterwards; a reader accepts an id once it has seen it under two different beats.
// A file nobody keeps beating -- what a dead job left, whatever its age -- is never accepted, a live one always within two beats.
static void job_token(uint8_t (&tok)[32]) {
    memset(tok, 0, sizeof(tok));
    const char* v = getenv("LELE_JOB_ID");
    if (!v || !*v) v = getenv("TORCHELASTIC_RUN_ID");
    if (!v) return;
    uint64_t h[4] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull, 0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full};  // four FNV-1a lanes
    for (size_t i = 0; v[i]; ++i) {
        const unsigned k = (unsigned)(i & 3);
        h[k] = (h[k] ^ (uint8_t)v[i]) * 0x100000001b3ull;
        h[(k + 1) & 3] ^= h[k] >> 29;
    }
    memcpy(tok, h, sizeof(tok));
    tok[0] |= 1;  // never all zeros: "a token was set"
}

static bool token_is_empty(const uint8_t (&tok)[32]) {
    for (uint8_t b : tok)
        if (b) return false;
    return true;
}

// [token][id] (+ [beat] when the token is empty), complete when it appears: written beside the path and renamed over it
static int write_id_file(const char* path, const uint8_t (&tok)[32], const uint8_t (&id)[128], uint64_t beat) {
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    LELE_REQUIRE(f, "comm_init_file: cannot write %s (%s)", tmp.c_str(), strerror(errno));
    size_t w = fwrite(tok, 1, sizeof(tok), f) + fwrite(id, 1, sizeof(id), f), want = sizeof(tok) + sizeof(id);
    if (token_is_empty(tok)) {
        w += fwrite(&beat, 1, sizeof(beat), f);
        want += sizeof(beat);
    }
    fclose(f);
    LELE_REQUIRE(w == want, "comm_init_file: short write to %s", tmp.c_str());
    LELE_REQUIRE(rename(tmp.c_str(), path) == 0, "comm_init_file: rename to %s failed (%s)", path, strerror(errno));
    return 0;
}

/* the reader's half of lele_hip_comm_init_file: wait (at most timeout_ms) for a rendezvous file of THIS job under `path` and
 * hand out its 128-byte id.  No device, no RCCL: callable (and tested) on a host without a GPU. */
int lele_hip_comm_read_id_file(const char* path, int timeout_ms, uint8_t* id128) {
    LELE_REQUIRE(path && id128, "comm_read_id_file: NULL argument");
    uint8_t tok[32];
    job_token(tok);
    const bool tokenless = token_is_empty(tok);
    const size_t want = 32 + 128 + (tokenless ? 8 : 0);
    const int step_ms = 5;
    int waited = 0;
    bool have_beat = false;
    uint64_t first_beat = 0;
    for (;;) {
        FILE* f = fopen(path, "rb");
        if (f) {
            uint8_t got[32 + 128 + 8 + 1];
            const size_t r = fread(got, 1, sizeof(got), f);
            fclose(f);
            // the rename makes the file appear complete: another size is a foreign file, another token another job's
            if (r == want && memcmp(got, tok, 32) == 0) {
                uint64_t beat = 0;
                memcpy(&beat, got + 160, tokenless ? 8 : 0);
                if (!tokenless || (have_beat && beat != first_beat)) {
                    memcpy(id128, got + 32, 128);
                    return 0;
                }
                if (!have_beat) first_beat = beat, have_beat = true;
            }
        }
        LELE_REQUIRE(waited < timeout_ms,
                     "comm_init_file: waited %d ms for %s (a file of this job: set LELE_JOB_ID per launch; without a token the file "
                     "must be kept alive by a running rank 0)", timeout_ms, path);
        struct timespec ts = {0, step_ms * 1000000L};
        nanosleep(&ts, nullptr);
        waited += step_ms;
    }
}

int lele_hip_comm_init_file(LeleCtx* ctx, const char* path, int rank, int world, int timeout_ms, LeleComm** out) {
    LELE_REQUIRE(ctx && path && out, "comm_init_file: NULL argument");
    uint8_t id[128], tok[32];
    job_token(tok);
    if (rank != 0) {
        LELE_TRY(lele_hip_comm_read_id_file(path, timeout_ms, id));
        return lele_hip_comm_init(ctx, id, rank, world, out);
    }
    (void)unlink(path);  // whatever an earlier job left there
    LELE_TRY(lele_hip_comm_unique_id(id));
    LELE_TRY(write_id_file(path, tok, id, 1));
    if (!token_is_empty(tok) || world == 1) return lele_hip_comm_init(ctx, id, rank, world, out);
    // token-less, other ranks to come: keep the file beating while this thread sits in ncclCommInitRank (which returns when all joined)
    std::atomic<bool> stop{false};
    std::thread beat([&] {
        for (uint64_t b = 2; !stop.load(std::memory_order_acquire); ++b) {
            struct timespec ts = {0, 20 * 1000000L};
            nanosleep(&ts, nullptr);
            uint8_t t0[32] = {0};
            FILE* f = fopen((std::string(path) + ".tmp").c_str(), "wb");  // errors here only delay the readers: they time out and say so
            if (!f) continue;
            const bool ok = fwrite(t0, 1, 32, f) + fwrite(id, 1, 128, f) + fwrite(&b, 1, 8, f) == 168;
            fclose(f);
            if (ok) (void)rename((std::string(path) + ".tmp").c_str(), path);
        }
    });
    const int rc = lele_hip_comm_init(ctx, id, rank, world, out);
    stop.store(true, std::memory_order_release);
    beat.join();
    (void)unlink(path);  // every rank has read it (or the group failed): nothing token-less is left behind for a later job to find
    return rc;
}

int lele_hip_comm_rank(const LeleComm* c, int* rank, int* world) {
    LELE_REQUIRE(c && rank && world, "comm_rank: NULL argument");
    *rank = c->rank;
    *world = c->world;
    return 0;
}

/* every rank contributes `count` i32 values (device memory); out receives [world, count] in rank order, on the ctx stream */
int lele_hip_comm_allgather_i32(LeleComm* c, const LeleTensor* send, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(c && send && out, "comm_allgather_i32: NULL argument");
    LELE_REQUIRE(send->dtype == LELE_I32, "comm_allgather_i32: i32 tensor required");
    LELE_REQUIRE(send->mem == LELE_MEM_DEVICE, "comm_allgather_i32: the send tensor must be device memory (ids stay on the GPU)");
    LeleCtx* ctx = c->ctx;
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t count = numel(send);
    LELE_TRY(out->reserve((size_t)c->world * count * 4));
    if (count) LELE_RCCL_CHECK(c->api, c->api->AllGather(send->data, out->data, (size_t)count, kInt32, c->comm, ctx->stream));
    return set_shape(out_shape, out_rank, {(int64_t)c->world, count});
}

/* any element type: `bytes` of device memory from every rank, [world, ...shape] in rank order, on the ctx stream */
int lele_hip_comm_allgather(LeleComm* c, const LeleTensor* send, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(c && send && out, "comm_allgather: NULL argument");
    LELE_REQUIRE(send->mem == LELE_MEM_DEVICE, "comm_allgather: the send tensor must be device memory");
    LELE_REQUIRE(send->rank + 1 <= LELE_MAX_RANK, "comm_allgather: rank %d leaves no room for the rank axis", send->rank);
    LeleCtx* ctx = c->ctx;
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)numel(send) * dtype_size(send->dtype);
    LELE_TRY(out->reserve((size_t)c->world * bytes));
    if (bytes) LELE_RCCL_CHECK(c->api, c->api->AllGather(send->data, out->data, bytes, kUint8, c->comm, ctx->stream));
    std::vector<int64_t> shape{(int64_t)c->world};
    shape.insert(shape.end(), send->shape, send->shape + send->rank);
    return set_shape_v(out_shape, out_rank, shape);
}

/* MAX over ranks of a host scalar (row widths of ragged shards; the bench's wall time in ns): one tiny all-reduce + sync */
int lele_hip_comm_allreduce_max_i64(LeleComm* c, int64_t* value) {
    LELE_REQUIRE(c && value, "comm_allreduce_max_i64: NULL argument");
    LeleCtx* ctx = c->ctx;
    LELE_REQUIRE(!ctx->capturing, "comm_allreduce_max_i64: not allowed while a graph is being captured");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_HIP_CHECK(hipMemcpyAsync(c->scratch, value, 8, hipMemcpyHostToDevice, ctx->stream));
    LELE_RCCL_CHECK(c->api, c->api->AllReduce(c->scratch, (char*)c->scratch + 8, 1, kInt64, kMax, c->comm, ctx->stream));
    LELE_HIP_CHECK(hipMemcpyAsync(value, (char*)c->scratch + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
    LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* all ranks have finished everything they queued on their ctx streams before any rank returns */
int lele_hip_comm_barrier(LeleComm* c) {
    int64_t one = 1;
    return lele_hip_comm_allreduce_max_i64(c, &one);
}

int lele_hip_comm_destroy(LeleComm* c) {
    if (!c) return 0;
    auto& live = c->ctx->comms;  // a communicator outlives neither its context nor, at process exit, the HIP runtime: the context
    live.erase(std::remove(live.begin(), live.end(), c), live.end());  // destroys the ones still registered before its stream
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)c->api->CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
    return 0;
}

}  // extern "C"
