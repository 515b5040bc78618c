// common.h -- internals shared by the translation units of liblele_hip.so (context, buffers, staging).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lele_hip.h"

namespace lele {

void set_error(const char* fmt, ...);

#define LELE_HIP_CHECK(expr)                                                                     \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            ::lele::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define LELE_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            ::lele::set_error(__VA_ARGS__); \
            return 2;                       \
        }                                   \
    } while (0)

#define LELE_TRY(expr)           \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

inline size_t dtype_size(int dt) {
    switch (dt) {
        case LELE_F32: return 4;
        case LELE_I64: return 8;
        case LELE_I32: return 4;
        default: return 1;
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: remember the opt-in per (kernel, device),
// under a lock (one ctx per host thread is the documented threading model, so launches race on this table)
inline hipError_t ensure_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

inline int64_t numel(const LeleTensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->rank; ++i) n *= t->shape[i];
    return n;
}

}  // namespace lele

// Opaque types of the C ABI -------------------------------------------------------------------------------
struct LeleCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    // staging arena for LELE_MEM_HOST inputs of the call in flight (bump allocator, reset per op)
    char* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    std::vector<void*> arena_overflow;  // freed at the next reset
    // LELE_MEM_WEIGHT cache: (host ptr, bytes, tag) -> device copy (tag distinguishes pre-packed forms)
    std::map<std::tuple<const void*, size_t, int>, void*> weights;
    // device address -> the LeleBuf that owns it (kept current by LeleBuf::reserve), for the row-statistics side channel
    std::map<const void*, LeleBuf*> buf_of_data;
    // scratch that ops may keep across calls (grown on demand)
    void* scratch = nullptr;
    size_t scratch_cap = 0;
    // page-locked mailbox for small device-to-host reads (a probability, token ids, a count): a copy into pageable memory
    // goes through the runtime's own staging path and costs several times the latency of one into pinned memory
    void* mailbox = nullptr;
    static constexpr size_t kMailboxBytes = 64 * 1024;
    // true between lele_hip_graph_begin / _end: every launch on `stream` is being recorded into a hipGraph, so nothing
    // may allocate, free, synchronise or touch pageable host memory (the guards return an error instead)
    bool capturing = false;
    // Bumped whenever device memory a recorded graph may have baked in is freed (a LeleBuf or the scratch block re-allocated,
    // arena overflow blocks released): a LeleGraph remembers the value at graph_end and refuses to replay after a change.
    uint64_t generation = 0;
    // Sticky device-side error word (page-locked host memory mapped into the device): kernels that meet a data-dependent
    // violation the reference would panic! on (an out-of-range gather index) clamp the access and set a bit here; the next
    // lele_hip_sync / buf_to_host reports it as an error.  LELE_DEVERR_* below.
    unsigned* deverr_host = nullptr;
    unsigned* deverr_dev = nullptr;
    std::vector<LeleGraph*> graphs;  // alive graphs recorded on this ctx (destroyed with it)
    std::vector<struct LeleComm*> comms;  // communicators bound to this ctx's stream (destroyed with it, BEFORE the stream)
    // per-stage stopwatch of the quantised linear (bench.py's roofline block): events recorded between its kernels
    struct QProf {
        bool on = false;
        std::vector<hipEvent_t> ev;  // 4 per call: start, after range, after row quantisation, after GEMM
        size_t used = 0;
    } qprof;

    // reset_conv_stats / print_conv_stats: 2-D convolutions issued since the last reset
    int64_t conv_calls = 0, conv_macs = 0;

    // two library-owned result buffers for ops that fall back to an unfused sequence and need somewhere to put the intermediate
    // (add3 / fused_quantized_linear_residual with an operand that broadcasts OUTWARD: the in-place second pass is not possible)
    LeleBuf* tmp[3] = {nullptr, nullptr, nullptr};
    int tmp_buf(int i, LeleBuf** out);

    // ---- lanes: extra streams of this context, for plans whose independent branches should overlap (lele_hip_lane_*).
    // The fields above (stream, arena*, scratch*, tmp) are always those of the CURRENT lane -- every op of the library keeps using
    // them unchanged; switching lanes swaps them with the parked state of the target lane.  Each lane has its own stream, staging
    // arena, scratch block and temporary buffers, so two ops in flight on different lanes never share library-owned memory.
    struct LaneState {
        hipStream_t stream = nullptr;
        char* arena = nullptr;
        size_t arena_cap = 0, arena_used = 0;
        std::vector<void*> arena_overflow;
        void* scratch = nullptr;
        size_t scratch_cap = 0;
        LeleBuf* tmp[3] = {nullptr, nullptr, nullptr};
    };
    static constexpr int kMaxLanes = 4;
    int lane = 0;                       // the current lane
    bool side_lanes = false;            // a lane other than 0 has been current since the streams were last drained (sync_all)
    // quant.hip: arrival counters of the one-launch feed-forward form, one block per grid shape (ncb, nrr); cleared at allocation only
    std::map<std::pair<int, int>, unsigned*> rs_sync;
    std::vector<LaneState> parked;      // parked[l] = state of lane l while it is not current (entry of the current lane: unused)
    hipStream_t lane_stream[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};  // every lane's OWN stream ([0] = the ctx stream)
    std::vector<hipEvent_t> lane_events;  // lele_hip_lane_record / _wait, eager mode
    // While a graph is being captured there is only ONE stream (lane 0's): a capture that spreads over several streams is recorded by
    // the runtime as "parallel capture streams" of each other, and ending such a capture recursed without end once a third stream had
    // joined an earlier one (hip::Stream::EndCapture, ROCm 7.0).  Instead the DAG is built on the capturing stream itself: the dependency
    // set of the NEXT recorded node is replaced by the tail of the lane that issues it (hipStreamUpdateCaptureDependencies), an event
    // is the set of nodes it stands for, a wait adds them.  Plain node-to-node edges, no event nodes, no extra streams.
    std::vector<hipGraphNode_t> lane_tail[kMaxLanes];
    std::vector<std::vector<hipGraphNode_t>> event_nodes;
    int capture_deps_get(std::vector<hipGraphNode_t>* out);
    int sync_all();                     // drain every lane's stream
    void swap_lane_memory(int to);      // park the current lane's arena / scratch / temporaries, take lane `to`'s

    int check_deverr(const char* where);
    int arena_reset();
    int arena_alloc(size_t bytes, void** out);
    int get_scratch(size_t bytes, void** out);
    // Returns a device pointer for tensor t (uploads host data through the arena / weight cache).
    int dev_ptr(const LeleTensor* t, const void** out);
};

struct LeleGraph {
    LeleCtx* ctx = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t generation = 0;
};

#define LELE_DEVERR_GATHER_INDEX 1u
#define LELE_DEVERR_FFN_SYNC 2u   // igemm_rs_kernel<3>: a workgroup waited RS_SYNC_LIMIT for its neighbours (they were not resident)

// Developer switches (kernel variants for A/B timing, stamps, ablations) exist in the LAB build only

// Device-side assertions of the bounds-checking build (LELE_HIP_DEBUG_BOUNDS=1 python -m lele_amd.build -> liblele_hip_dbg.so): the
// loaders of the GEMM core, the window kernels' LDS offsets and the epilogues' store coordinates state what they assume; a violation
// prints the condition, the source line and the block / thread, and traps the kernel (the next synchronisation fails).  In the
// product library the macro is empty.
#ifdef LELE_HIP_DEBUG_BOUNDS
#include <stdio.h>
#define LELE_DEV_ASSERT(cond)                                                                                                      \
    do {                                                                                                                           \
        if (!(cond)) {                                                                                                             \
            printf("lele_hip bounds assertion failed: %s (%s:%d) block (%u, %u, %u) thread %u\n", #cond, __FILE__, __LINE__, blockIdx.x, \
                   blockIdx.y, blockIdx.z, threadIdx.x);                                                                            \
            __builtin_trap();                                                                                                      \
        }                                                                                                                          \
    } while (0)
#else
#define LELE_DEV_ASSERT(cond) ((void)0)
#endif

// (LELE_HIP_LAB=1 python -m lele_amd.build -> liblele_hip_lab.so); in the product library lab_env() is NULL for every name.
// The product's own run-time switches are the seven documented in INTEGRATION.md ("Run-time switches").
#ifdef LELE_HIP_LAB
inline const char* lab_env(const char* name) { return getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif

struct LeleBuf {
    LeleCtx* ctx = nullptr;
    void* data = nullptr;
    size_t cap = 0;
    size_t bytes = 0;  // size of the last result
    // Side channel producer -> consumer: {min, max} of every row of the result, written by the kernel that produced it
    // (LayerNorm) so that a dynamic quantisation reading this buffer next need not scan it again (quant.hip).  Valid only
    // until the buffer is written again: reserve() -- which every op calls on its output -- clears the flag.
    float* rowstat = nullptr;  // [rowstat_rows][2] on the device
    size_t rowstat_cap = 0;    // in rows
    int64_t rowstat_rows = 0, rowstat_len = 0;  // ROWS: pairs = rows, len = row length; WHOLE: pairs = blocks, len = element count
    int64_t rowstat_m = 0, rowstat_batch = 0;   // kind 2: rows per slice and the number of slices the producer saw
    int rowstat_kind = 0;                       // 0 = one pair per row (LayerNorm); 1 = pairs that together cover the whole tensor;
                                                // 2 = rowstat_rows / slices pairs per slice of rowstat_m rows of length rowstat_len
    bool rowstat_valid = false;
    int reserve(size_t n);
    int reserve_rowstat(int64_t rows);  // may decline (returns 0 with rowstat == nullptr untouched) while capturing
};

namespace lele {
// features_ops.hip: power spectrum |FFT|^2 of `rows` real rows of length n_fft (a power of two <= 4096), bit-exact with the
// reference's radix-2 network; out_power is [rows, n_fft/2 + 1]
int live_contexts(int device);  // context.hip: contexts of this process alive on `device`
int fft_twiddles(LeleCtx* ctx, int64_t n, const float** tw_re, const float** tw_im);  // features_ops.hip: fft.rs:136-157 on the device
int fft_rows_power(LeleCtx* ctx, const float* rows_in, int64_t rows, int64_t n_fft, float* out_power);

// quant.hip: dynamic-quantisation parameters of the joint range of several device arrays (16 bytes on the device)
struct QParamsDev {
    float scale, zp, inv_scale;
    int zp_i;
};
int quant_params_of(LeleCtx* ctx, const float* const* srcs, const int64_t* lens, int nsrc, void* prm_dev);

// Destination of an op that takes a LelePitch (lele_hip.h): dense (out_pitch == 0: the buffer is resized as usual, out_offset must
// be 0) or a window of a buffer that ALREADY holds the enclosing tensor (the caller reserved it: growing it here would drop what
// other producers wrote).  `images` chunks of `per_image` elements of `esize` bytes.
inline int pitched_out(LeleBuf* out, const LelePitch* pv, int64_t images, int64_t per_image, size_t esize, void** dst) {
    if (pv->out_pitch == 0) {
        LELE_REQUIRE(pv->out_offset == 0, "pitched op: out_offset without out_pitch");
        LELE_TRY(out->reserve((size_t)(images * per_image) * esize));
        *dst = out->data;
        return 0;
    }
    LELE_REQUIRE(pv->out_offset >= 0 && pv->out_pitch >= per_image, "pitched op: out_pitch %lld is smaller than one image (%lld elements)",
                 (long long)pv->out_pitch, (long long)per_image);
    const int64_t need = images > 0 ? pv->out_offset + (images - 1) * pv->out_pitch + per_image : 0;
    LELE_REQUIRE(out->data && (size_t)need * esize <= out->cap,
                 "pitched op: the destination buffer holds %zu bytes, the window ends at %lld (reserve the enclosing tensor first)", out->cap,
                 (long long)((size_t)need * esize));
    out->rowstat_valid = false;  // the buffer is being written: producer-side statistics of an earlier result are stale
    *dst = (char*)out->data + (size_t)pv->out_offset * esize;
    return 0;
}

inline int set_shape(int64_t* out_shape, int32_t* out_rank, std::initializer_list<int64_t> dims) {
    if (out_rank) *out_rank = (int32_t)dims.size();
    if (out_shape) {
        int i = 0;
        for (int64_t d : dims) out_shape[i++] = d;
    }
    return 0;
}
inline int set_shape_v(int64_t* out_shape, int32_t* out_rank, const std::vector<int64_t>& dims) {
    if (out_rank) *out_rank = (int32_t)dims.size();
    if (out_shape)
        for (size_t i = 0; i < dims.size(); ++i) out_shape[i] = dims[i];
    return 0;
}

#ifdef __HIPCC__
// f32 -> three bf16 pieces, two values at a time: a = h.lo + m.lo + l.lo and b = h.hi + m.hi + l.hi EXACTLY (element 0 in the low half
// of each word, the layout of a bf16 MFMA operand pair).  Every piece is ROUNDED TO NEAREST (v_cvt_pk_bf16_f32), the remainders are
// exact f32 subtractions: |a - h| <= 2^-9 |a| has at most 16 significant bits, |a - h - m| at most 8, so l is exact.
// Why not truncation (a mask and a subtraction, the same instruction count): with truncated pieces m and l carry the SIGN OF THE VALUE,
// so in a split-bf16 product every one of the six kept terms and the three dropped ones (m l, l m, l l) has the sign of a * b -- the
// dropped terms and the matrix core's alignment truncation (a product is cut 2 bits below the accumulator's ulp TOWARDS ZERO:
// tools/mfma_round.hip) then shrink every product the same way, and a sum of K products comes out ~1e-7 (K = 576) too small in
// magnitude.  One layer hides that below f32 round-off; a 100-layer SiLU network does not -- a scale error is the one perturbation that
// adds up coherently through depth: the lifted Yolo26n-seg graph ended 3.3e-5 low (profiles/r05_graph_error_growth.json), ten times
// the spread of an f32 chain.  With rounded pieces m and l take either sign, five of the six terms and all dropped ones become
// zero-mean noise, and what is left of the bias is the hh term's alone.
__device__ __forceinline__ void split3_bf16_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 lele_bf2 __attribute__((ext_vector_type(2)));
    typedef float lele_f2 __attribute__((ext_vector_type(2)));
    auto pack = [](float x, float y) { return __builtin_bit_cast(unsigned, __builtin_convertvector(lele_f2{x, y}, lele_bf2)); };
    h = pack(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pack(ra, rb);
    l = pack(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
#endif
}  // namespace lele
