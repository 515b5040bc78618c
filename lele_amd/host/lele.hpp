// lele.hpp -- header-only C++ mirror of lele's host-side interface over the C ABI of include/lele_hip.h.
//
// lele's host code is Rust (`lele::tensor::TensorView`, `lele::kernels::*`, `lele::features::*`); Rust is not
// available in this image, so this header is the compiled stand-in for the Rust shim of INTEGRATION.md: the same
// names, the same argument order and meaning, the same "panic on misuse" behaviour (here: lele::Error carrying
// lele_hip_last_error()), with `Vec<f32>` workspace slots replaced by lele::Buffer (a LeleBuf in HBM).
//   namespace lele            TensorView (src/tensor.rs:5-71), Ctx, Buffer, Error
//   namespace lele::kernels   one function per reference kernel (file:line in include/lele_hip.h)
//   namespace lele::features  FeatureConfig, SenseVoiceFrontend, Cmvn, Lfr (src/features/*.rs)
// Results stay on the device; TensorView::to_vec<T>() is the lazy D2H (the analogue of `.data` access in Rust).
#pragma once
#include "lele_hip.h"

#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lele {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int rc) {
    if (rc != 0) throw Error(lele_hip_last_error());
}

// One HIP stream + staging arena + weight cache: the per-thread state of a lele model instance (SURVEY 8b "Threading").
class Ctx {
   public:
    explicit Ctx(int device = 0) { check(lele_hip_ctx_create(device, &h_)); }
    ~Ctx() {
        if (h_) lele_hip_ctx_destroy(h_);
    }
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
    LeleCtx* raw() const { return h_; }
    void sync() const { check(lele_hip_sync(h_)); }
    // hipGraph capture of an op sequence (see include/lele_hip.h): begin, issue the ops, end -> Graph; Graph::launch()
    void graph_begin() const { check(lele_hip_graph_begin(h_)); }
    LeleGraph* graph_end() const {
        LeleGraph* g = nullptr;
        const int rc = lele_hip_graph_end(h_, &g);
        if (rc != 0) lele_hip_graph_abort(h_);
        check(rc);
        return g;
    }
    // lanes (include/lele_hip.h, lele_hip_lane_*): independent branches of a plan on other streams of this context -- parallel branches
    // of one recorded graph; every cross-lane dependency is the caller's to order with record / wait
    void lane_set(int lane) const { check(lele_hip_lane_set(h_, lane)); }
    void lane_record(int event) const { check(lele_hip_lane_record(h_, event)); }
    void lane_wait(int event) const { check(lele_hip_lane_wait(h_, event)); }
    static Ctx& current() {  // thread-local default, like lele's thread-local scratch and caches; LELE_HIP_DEVICE picks the GPU
        static thread_local Ctx ctx(default_device());
        return ctx;
    }
    static int default_device() {
        const char* d = std::getenv("LELE_HIP_DEVICE");
        return d && *d ? std::atoi(d) : 0;
    }

   private:
    LeleCtx* h_ = nullptr;
};

class Comm {  // this process's membership of an RCCL communicator (one process per GPU), bound to a ctx stream (lele_hip_comm_*)
   public:
    Comm(const Ctx& ctx, const std::string& id_file, int rank, int world, int timeout_ms = 120000) : rank_(rank), world_(world) {
        check(lele_hip_comm_init_file(ctx.raw(), id_file.c_str(), rank, world, timeout_ms, &h_));
    }
    ~Comm() {
        if (h_) lele_hip_comm_destroy(h_);
    }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    LeleComm* raw() const { return h_; }
    int rank() const { return rank_; }
    int world() const { return world_; }
    void barrier() const { check(lele_hip_comm_barrier(h_)); }
    int64_t max(int64_t v) const {
        check(lele_hip_comm_allreduce_max_i64(h_, &v));
        return v;
    }

   private:
    LeleComm* h_ = nullptr;
    int rank_, world_;
};

class Graph {  // a captured op sequence; replays on the ctx stream with one hipGraphLaunch
   public:
    explicit Graph(LeleGraph* g) : g_(g) {}
    ~Graph() {
        if (g_) lele_hip_graph_destroy(g_);
    }
    Graph(const Graph&) = delete;
    Graph& operator=(const Graph&) = delete;
    void launch() const { check(lele_hip_graph_launch(g_)); }

   private:
    LeleGraph* g_ = nullptr;
};

// A workspace slot: the device-side counterpart of `ws.buf_k: Vec<f32>` (compiler/mod.rs:148-290).
class Buffer {
   public:
    explicit Buffer(Ctx& ctx = Ctx::current()) { check(lele_hip_buf_create(ctx.raw(), &h_)); }
    ~Buffer() {
        if (h_) lele_hip_buf_destroy(h_);
    }
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    LeleBuf* raw() const { return h_; }
    void* data() const { return lele_hip_buf_data(h_); }
    void reserve(size_t bytes) { check(lele_hip_buf_reserve(h_, bytes)); }   // the enclosing tensor of windowed results (LelePitch)
    void upload(const void* src, size_t bytes) { check(lele_hip_buf_from_host(h_, src, bytes)); }
    void download(void* dst, size_t bytes) const { check(lele_hip_buf_to_host(h_, dst, bytes)); }

   private:
    LeleBuf* h_ = nullptr;
};

template <class T>
struct dtype_of;
template <>
struct dtype_of<float> {
    static constexpr int value = LELE_F32;
};
template <>
struct dtype_of<int64_t> {
    static constexpr int value = LELE_I64;
};
template <>
struct dtype_of<int32_t> {
    static constexpr int value = LELE_I32;
};
template <>
struct dtype_of<uint8_t> {
    static constexpr int value = LELE_U8;
};
template <>
struct dtype_of<int8_t> {
    static constexpr int value = LELE_I8;
};

// Borrowed view: host memory (from_slice), an immutable weight (weight -> cached on the device) or a device result.
class TensorView {
   public:
    std::vector<int64_t> shape;
    TensorView() = default;  // TensorView::empty()
    template <class T>
    static TensorView from_slice(const T* data, std::vector<int64_t> shape) {  // tensor.rs:52-61
        return TensorView(data, std::move(shape), dtype_of<T>::value, LELE_MEM_HOST);
    }
    template <class T>
    static TensorView weight(const T* data, std::vector<int64_t> shape) {  // Model::weight_f32, tensor.rs:131-147
        return TensorView(data, std::move(shape), dtype_of<T>::value, LELE_MEM_WEIGHT);
    }
    static TensorView from_device(const Buffer& buf, std::vector<int64_t> shape, int dtype = LELE_F32) {
        TensorView t(buf.data(), std::move(shape), dtype, LELE_MEM_DEVICE);
        t.buf_ = &buf;
        return t;
    }
    // A CHANNEL VIEW of a device tensor (include/lele_hip.h, LelePitch): the tensor starts `offset` elements into the buffer and image n
    // (index along axis 0) starts n * pitch elements after image 0; inside an image it is dense.  Only the *_pitched entry points (and
    // conv2d_res) accept views; `pitch` == 0 and `offset` == 0: dense.
    static TensorView from_device(const Buffer& buf, std::vector<int64_t> shape, int dtype, int64_t offset, int64_t pitch) {
        TensorView t = from_device(buf, std::move(shape), dtype);
        int64_t per = 1;
        for (size_t i = 1; i < t.shape.size(); ++i) per *= t.shape[i];
        t.offset_ = offset;
        t.pitch_ = (pitch == per && offset == 0) ? 0 : pitch;   // the whole tensor: dense
        return t;
    }
    bool is_view() const { return pitch_ != 0 || offset_ != 0; }
    int64_t offset() const { return offset_; }
    int64_t pitch() const { return pitch_; }
    const Buffer* buffer() const { return buf_; }
    TensorView channels(int64_t c0, int64_t c1) const {   // channels [c0, c1) of a rank >= 2 device tensor, no copy
        if (mem_ != LELE_MEM_DEVICE || !buf_ || shape.size() < 2 || c0 < 0 || c1 > shape[1] || c0 > c1)
            throw Error("TensorView::channels: a device tensor of rank >= 2 and 0 <= c0 <= c1 <= C expected");
        int64_t inner = 1;
        for (size_t i = 2; i < shape.size(); ++i) inner *= shape[i];
        std::vector<int64_t> s = shape;
        s[1] = c1 - c0;
        return from_device(*buf_, std::move(s), dtype_, offset_ + c0 * inner, pitch_ ? pitch_ : shape[1] * inner);
    }
    int64_t size() const {
        int64_t n = 1;
        for (int64_t d : shape) n *= d;
        return n;
    }
    size_t dim() const { return shape.size(); }
    int dtype() const { return dtype_; }
    bool is_empty() const { return data_ == nullptr; }
    TensorView with_shape(std::vector<int64_t> s) const {  // view ops share the data (shape.rs:2-13)
        if (is_view()) throw Error("reshape of a channel view: copy it out first (copy_view)");
        TensorView t = *this;
        t.shape = std::move(s);
        return t;
    }
    LeleTensor c() const {
        const void* d = data_;
        if (mem_ == LELE_MEM_DEVICE && buf_ && offset_)
            d = (const char*)buf_->data() + (size_t)offset_ * (dtype_ == LELE_I64 ? 8 : (dtype_ == LELE_U8 || dtype_ == LELE_I8) ? 1 : 4);
        return LeleTensor{d, shape.data(), (int32_t)shape.size(), dtype_, mem_};
    }
    template <class T>
    std::vector<T> to_vec() const {  // `.data` in Rust: host copy (D2H for device results)
        if (dtype_of<T>::value != dtype_) throw Error("TensorView::to_vec: dtype mismatch");
        std::vector<T> v((size_t)size());
        if (v.empty()) return v;
        if (mem_ == LELE_MEM_DEVICE) {
            if (!buf_) throw Error("TensorView::to_vec: device view without a buffer");
            if (is_view()) {   // a channel view: the span that holds it, then image by image
                const int64_t n = shape[0];
                int64_t per = 1;
                for (size_t i = 1; i < shape.size(); ++i) per *= shape[i];
                const int64_t pitch = pitch_ ? pitch_ : per;
                std::vector<T> flat((size_t)(offset_ + (n - 1) * pitch + per));
                buf_->download(flat.data(), flat.size() * sizeof(T));
                for (int64_t i = 0; i < n; ++i) std::memcpy(v.data() + i * per, flat.data() + offset_ + i * pitch, (size_t)per * sizeof(T));
                return v;
            }
            buf_->download(v.data(), v.size() * sizeof(T));
        } else {
            std::memcpy(v.data(), data_, v.size() * sizeof(T));
        }
        return v;
    }

   private:
    TensorView(const void* d, std::vector<int64_t> s, int dt, int mem) : shape(std::move(s)), data_(d), dtype_(dt), mem_(mem) {}
    const void* data_ = nullptr;
    int dtype_ = LELE_F32;
    int mem_ = LELE_MEM_HOST;
    const Buffer* buf_ = nullptr;
    int64_t offset_ = 0, pitch_ = 0;   // channel view (elements); both 0: dense
};

namespace detail {
struct Shape {
    int64_t dims[8] = {0};
    int32_t rank = 0;
    std::vector<int64_t> vec() const { return std::vector<int64_t>(dims, dims + rank); }
};
struct Opt {  // Option<&TensorView>
    LeleTensor t;
    const LeleTensor* p = nullptr;
    explicit Opt(const TensorView* v) {
        if (v) {
            t = v->c();
            p = &t;
        }
    }
};
inline std::vector<int64_t> row_major_strides(const std::vector<int64_t>& shape) {
    std::vector<int64_t> st(shape.size(), 1);
    for (int i = (int)shape.size() - 2; i >= 0; --i) st[i] = st[i + 1] * shape[i + 1];
    return st;
}
inline LeleCtx* ctx() { return Ctx::current().raw(); }
}  // namespace detail

namespace kernels {
using detail::ctx;
using detail::Opt;
using detail::Shape;

#define LELE_RET(out, dt) return TensorView::from_device(out, sh.vec(), dt)

// ---- gemm.rs
inline TensorView matmul(const TensorView& a, const TensorView& b, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    check(lele_hip_matmul(ctx(), &ta, &tb, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView matmul_fused_add(const TensorView& a, const TensorView& b, const TensorView& bias, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c(), tc = bias.c();
    check(lele_hip_matmul_fused_add(ctx(), &ta, &tb, &tc, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView gemm(const TensorView& a, const TensorView& b, const TensorView* c, float alpha, float beta, bool trans_a,
                       bool trans_b, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    Opt oc(c);
    check(lele_hip_gemm(ctx(), &ta, &tb, oc.p, alpha, beta, trans_a, trans_b, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}

// ---- conv2d.rs / conv1d.rs
inline TensorView conv2d_activation(const TensorView& x, const TensorView& w, const TensorView* bias,
                                    const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                                    const std::vector<int64_t>& strides, int act, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c();
    Opt ob(bias);
    check(lele_hip_conv2d(ctx(), &tx, &tw, ob.p, dilations.data(), dilations.size(), group, pads.data(), pads.size(),
                          strides.data(), strides.size(), act, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView conv2d(const TensorView& x, const TensorView& w, const TensorView* bias, const std::vector<int64_t>& dilations,
                         int64_t group, const std::vector<int64_t>& pads, const std::vector<int64_t>& strides, Buffer& out) {
    return conv2d_activation(x, w, bias, dilations, group, pads, strides, LELE_ACT_NONE, out);
}
inline TensorView conv2d_fused(const TensorView& x, const TensorView& w, const TensorView* bias,
                               const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                               const std::vector<int64_t>& strides, bool relu, Buffer& out) {
    return conv2d_activation(x, w, bias, dilations, group, pads, strides, relu ? LELE_ACT_RELU : LELE_ACT_NONE, out);
}
inline TensorView conv2d_silu(const TensorView& x, const TensorView& w, const TensorView* bias,
                              const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                              const std::vector<int64_t>& strides, Buffer& out) {
    return conv2d_activation(x, w, bias, dilations, group, pads, strides, LELE_ACT_SILU, out);
}
// act(conv2d(x) + bias) + res in one call: a bottleneck's conv2d_silu followed by add (lele_hip_conv2d_res; the same bits).  Dense
// operands here; channel views (LelePitch) are the batch graph runner's business (lele_amd/plan.py).
inline TensorView conv2d_res(const TensorView& x, const TensorView& w, const TensorView* bias, const TensorView& res,
                             const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                             const std::vector<int64_t>& strides, int act, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c(), tr = res.c();
    Opt ob(bias);
    check(lele_hip_conv2d_res(ctx(), &tx, &tw, ob.p, &tr, dilations.data(), dilations.size(), group, pads.data(), pads.size(), strides.data(),
                              strides.size(), act, nullptr, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView conv1d_fused(const TensorView& x, const TensorView& w, const TensorView* bias,
                               const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                               const std::vector<int64_t>& strides, bool relu, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c();
    Opt ob(bias);
    check(lele_hip_conv1d(ctx(), &tx, &tw, ob.p, dilations.data(), dilations.size(), group, pads.data(), pads.size(),
                          strides.data(), strides.size(), relu, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView conv1d(const TensorView& x, const TensorView& w, const TensorView* bias, const std::vector<int64_t>& dilations,
                         int64_t group, const std::vector<int64_t>& pads, const std::vector<int64_t>& strides, Buffer& out) {
    return conv1d_fused(x, w, bias, dilations, group, pads, strides, false, out);
}
inline TensorView conv_transpose(const TensorView& x, const TensorView& w, const TensorView* bias,
                                 const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                                 const std::vector<int64_t>& strides, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c();
    Opt ob(bias);
    check(lele_hip_conv_transpose(ctx(), &tx, &tw, ob.p, dilations.data(), dilations.size(), group, pads.data(), pads.size(),
                                  strides.data(), strides.size(), out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}

// ---- rnn.rs
struct LstmOut {
    TensorView y, h, c;
};
inline LstmOut lstm(const TensorView& x, const TensorView& w, const TensorView& r, const TensorView* bias,
                    const TensorView* sequence_lens, const TensorView* initial_h, const TensorView* initial_c, Buffer& out_y,
                    Buffer& out_h, Buffer& out_c) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c(), tr = r.c();
    Opt ob(bias), os(sequence_lens), oh(initial_h), oc(initial_c);
    check(lele_hip_lstm(ctx(), &tx, &tw, &tr, ob.p, os.p, oh.p, oc.p, out_y.raw(), out_h.raw(), out_c.raw(), sh.dims, &sh.rank));
    const std::vector<int64_t> ys = sh.vec(), hs = {1, 1, ys.back()};
    return {TensorView::from_device(out_y, ys), TensorView::from_device(out_h, hs), TensorView::from_device(out_c, hs)};
}
struct GruOut {
    TensorView y, h;
};
inline GruOut gru(const TensorView& x, const TensorView& w, const TensorView& r, const TensorView* bias,
                  const TensorView* initial_h, bool linear_before_reset, Buffer& out_y, Buffer& out_h) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c(), tr = r.c();
    Opt ob(bias), oh(initial_h);
    check(lele_hip_gru(ctx(), &tx, &tw, &tr, ob.p, oh.p, linear_before_reset, out_y.raw(), out_h.raw(), sh.dims, &sh.rank));
    const std::vector<int64_t> ys = sh.vec();
    return {TensorView::from_device(out_y, ys), TensorView::from_device(out_h, {1, 1, ys.back()})};
}

// ---- quantization.rs
inline TensorView fused_quantized_linear(const TensorView& input, const TensorView& weight_int8, const TensorView& weight_scale,
                                         const TensorView& weight_zero, const TensorView* bias, bool apply_relu, Buffer& out) {
    Shape sh;
    LeleTensor a = input.c(), b = weight_int8.c(), s = weight_scale.c(), z = weight_zero.c();
    Opt ob(bias);
    check(lele_hip_fused_quantized_linear(ctx(), &a, &b, &s, &z, ob.p, apply_relu, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
struct DqlOut {
    TensorView y, scale, zero_point;
};
inline DqlOut dynamic_quantize_linear(const TensorView& x, Buffer& out_y, Buffer& out_scale, Buffer& out_zp) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_dynamic_quantize_linear(ctx(), &tx, out_y.raw(), out_scale.raw(), out_zp.raw(), sh.dims, &sh.rank));
    return {TensorView::from_device(out_y, sh.vec()), TensorView::from_device(out_scale, {1}),
            TensorView::from_device(out_zp, {1})};
}
inline TensorView mat_mul_integer_with_scale_bias(const TensorView& a, const TensorView& b, const TensorView* a_zero_point,
                                                  const TensorView* b_zero_point, const TensorView* scale, const TensorView* bias,
                                                  bool relu, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    Opt za(a_zero_point), zb(b_zero_point), sc(scale), bi(bias);
    check(lele_hip_mat_mul_integer_with_scale_bias(ctx(), &ta, &tb, za.p, zb.p, sc.p, bi.p, relu, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView mat_mul_integer(const TensorView& a, const TensorView& b, const TensorView* a_zero_point,
                                  const TensorView* b_zero_point, Buffer& out) {
    return mat_mul_integer_with_scale_bias(a, b, a_zero_point, b_zero_point, nullptr, nullptr, false, out);
}

// ---- math.rs: activations, element-wise, reductions
inline TensorView unary(int op, const TensorView& x, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_unary(ctx(), op, &tx, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
#define LELE_UNARY(name, OP) \
    inline TensorView name(const TensorView& x, Buffer& out) { return unary(OP, x, out); }
LELE_UNARY(exp, LELE_U_EXP)
LELE_UNARY(sigmoid, LELE_U_SIGMOID)
LELE_UNARY(tanh_kernel, LELE_U_TANH)
LELE_UNARY(silu, LELE_U_SILU)
LELE_UNARY(erf, LELE_U_ERF)
LELE_UNARY(gelu, LELE_U_GELU)
LELE_UNARY(fast_gelu, LELE_U_FAST_GELU)
LELE_UNARY(relu, LELE_U_RELU)
LELE_UNARY(sqrt, LELE_U_SQRT)
LELE_UNARY(log, LELE_U_LOG)
LELE_UNARY(sin, LELE_U_SIN)
LELE_UNARY(cos, LELE_U_COS)
LELE_UNARY(neg, LELE_U_NEG)
LELE_UNARY(reciprocal, LELE_U_RECIPROCAL)
LELE_UNARY(softplus, LELE_U_SOFTPLUS)
LELE_UNARY(not_, LELE_U_NOT)
LELE_UNARY(abs, LELE_U_ABS)
LELE_UNARY(floor, LELE_U_FLOOR)
LELE_UNARY(ceil, LELE_U_CEIL)
#undef LELE_UNARY
inline TensorView binary(int op, const TensorView& a, const TensorView& b, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    check(lele_hip_binary(ctx(), op, &ta, &tb, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, a.dtype() == LELE_I64 ? LELE_I64 : LELE_F32);
}
#define LELE_BINARY(name, OP) \
    inline TensorView name(const TensorView& a, const TensorView& b, Buffer& out) { return binary(OP, a, b, out); }
LELE_BINARY(add, LELE_B_ADD)
LELE_BINARY(sub, LELE_B_SUB)
LELE_BINARY(mul, LELE_B_MUL)
LELE_BINARY(div, LELE_B_DIV)
LELE_BINARY(pow, LELE_B_POW)
LELE_BINARY(max, LELE_B_MAX)
LELE_BINARY(min, LELE_B_MIN)
LELE_BINARY(equal, LELE_B_EQUAL)
LELE_BINARY(less, LELE_B_LESS)
LELE_BINARY(greater, LELE_B_GREATER)
LELE_BINARY(prelu, LELE_B_PRELU)
LELE_BINARY(mod_f32, LELE_B_MOD)
LELE_BINARY(and_, LELE_B_AND)
LELE_BINARY(or_, LELE_B_OR)
#undef LELE_BINARY
inline TensorView where_op(const TensorView& cond, const TensorView& x, const TensorView& y, Buffer& out) {
    Shape sh;
    LeleTensor tc = cond.c(), tx = x.c(), ty = y.c();
    check(lele_hip_where(ctx(), &tc, &tx, &ty, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView clip(const TensorView& x, const float* min_v, const float* max_v, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_clip(ctx(), &tx, min_v != nullptr, min_v ? *min_v : 0.0f, max_v != nullptr, max_v ? *max_v : 0.0f, out.raw(),
                        sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView reduce(int op, const TensorView& x, const std::vector<int64_t>& axes, bool keepdims, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_reduce(ctx(), op, &tx, axes.data(), axes.size(), keepdims, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView reduce_sum(const TensorView& x, const std::vector<int64_t>& axes, bool keepdims, Buffer& out) {
    return reduce(LELE_R_SUM, x, axes, keepdims, out);
}
inline TensorView reduce_mean(const TensorView& x, const std::vector<int64_t>& axes, bool keepdims, Buffer& out) {
    return reduce(LELE_R_MEAN, x, axes, keepdims, out);
}
inline TensorView reduce_max(const TensorView& x, const std::vector<int64_t>& axes, bool keepdims, Buffer& out) {
    return reduce(LELE_R_MAX, x, axes, keepdims, out);
}
inline TensorView reduce_l2(const TensorView& x, const std::vector<int64_t>& axes, bool keepdims, Buffer& out) {
    return reduce(LELE_R_L2, x, axes, keepdims, out);
}

// ---- norm.rs
inline TensorView layer_norm(const TensorView& x, const TensorView& scale, const TensorView& bias, int64_t axis, float epsilon,
                             Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), ts = scale.c(), tb = bias.c();
    check(lele_hip_layer_norm(ctx(), &tx, &ts, &tb, (int32_t)axis, epsilon, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView rms_norm(const TensorView& x, const TensorView& weight, int64_t axis, float epsilon, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = weight.c();
    check(lele_hip_rms_norm(ctx(), &tx, &tw, (int32_t)axis, epsilon, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView softmax(const TensorView& x, int64_t axis, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_softmax(ctx(), &tx, (int32_t)axis, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView batch_norm(const TensorView& x, const TensorView& scale, const TensorView& bias, const TensorView& mean,
                             const TensorView& var, float epsilon, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), ts = scale.c(), tb = bias.c(), tm = mean.c(), tv = var.c();
    check(lele_hip_batch_norm(ctx(), &tx, &ts, &tb, &tm, &tv, epsilon, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}

// ---- manipulation.rs / shape.rs / conv2d.rs:1051-1502: data movement
inline TensorView strided(const TensorView& x, const std::vector<int64_t>& oshape, const std::vector<int64_t>& strides,
                          int64_t offset, const std::vector<int64_t>* mods, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_strided_copy(ctx(), &tx, oshape.data(), strides.data(), mods ? mods->data() : nullptr, (int32_t)oshape.size(),
                                offset, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, x.dtype());
}
inline TensorView transpose(const TensorView& x, std::vector<int64_t> perm, Buffer& out) {  // manipulation.rs:644
    const int64_t nd = (int64_t)x.dim();
    if (perm.empty())
        for (int64_t i = nd - 1; i >= 0; --i) perm.push_back(i);
    const auto istr = detail::row_major_strides(x.shape);
    std::vector<int64_t> osh, ost;
    for (int64_t p : perm) {
        if (p < 0) p += nd;
        osh.push_back(x.shape[(size_t)p]);
        ost.push_back(istr[(size_t)p]);
    }
    return strided(x, osh, ost, 0, nullptr, out);
}
inline TensorView slice(const TensorView& x, const std::vector<int64_t>& starts, const std::vector<int64_t>& ends,
                        const std::vector<int64_t>& axes, const std::vector<int64_t>& steps, Buffer& out) {  // :209-380
    const int64_t nd = (int64_t)x.dim();
    std::vector<int64_t> a_start((size_t)nd, 0), a_end = x.shape, a_step((size_t)nd, 1);
    for (size_t i = 0; i < starts.size(); ++i) {
        int64_t ax = axes.empty() ? (int64_t)i : (axes[i] < 0 ? axes[i] + nd : axes[i]);
        const int64_t dim = x.shape[(size_t)ax], step = i < steps.size() ? steps[i] : 1;
        const int64_t s64 = starts[i], e64 = ends[i];
        const bool e_max = e64 > INT64_MAX / 2, e_min = e64 < INT64_MIN / 2;
        const int64_t start = s64 > dim ? dim : (s64 < -dim ? -dim : s64);
        const int64_t end = e_max ? dim : (e_min ? -dim : (e64 > dim ? dim : (e64 < -dim ? -dim : e64)));
        const int64_t ns = start < 0 ? start + dim : start;
        const int64_t ne = e_max ? (step > 0 ? dim : -1) : e_min ? (step > 0 ? 0 : -1) : (end < 0 ? end + dim : end);
        if (step > 0) {
            a_start[(size_t)ax] = std::min(std::max<int64_t>(ns, 0), dim);
            a_end[(size_t)ax] = std::min(std::max<int64_t>(ne, 0), dim);
        } else {
            a_start[(size_t)ax] = std::min(std::max<int64_t>(ns, 0), dim - 1);
            a_end[(size_t)ax] = std::min(std::max<int64_t>(ne, -1), dim - 1);
        }
        a_step[(size_t)ax] = step;
    }
    const auto istr = detail::row_major_strides(x.shape);
    std::vector<int64_t> osh, ost;
    int64_t off = 0;
    for (size_t d = 0; d < (size_t)nd; ++d) {
        const int64_t s = a_start[d], e = a_end[d], st = a_step[d];
        osh.push_back(st > 0 ? std::max<int64_t>(0, (e - s + st - 1) / st) : std::max<int64_t>(0, (s - e + (-st) - 1) / (-st)));
        ost.push_back(istr[d] * st);
        off += s * istr[d];
    }
    return strided(x, osh, ost, off, nullptr, out);
}
inline TensorView expand(const TensorView& x, const std::vector<int64_t>& shape, Buffer& out) {  // math.rs:2168
    const size_t nd = std::max(x.dim(), shape.size());
    const auto istr = detail::row_major_strides(x.shape);
    std::vector<int64_t> osh, ost;
    for (size_t i = 0; i < nd; ++i) {
        const int64_t oi = (int64_t)i - (int64_t)(nd - x.dim()), ot = (int64_t)i - (int64_t)(nd - shape.size());
        const int64_t din = oi >= 0 ? x.shape[(size_t)oi] : 1;
        const int64_t dt = ot >= 0 ? (shape[(size_t)ot] ? shape[(size_t)ot] : din) : 1;
        if (din == dt || dt == 1)
            osh.push_back(din);
        else if (din == 1)
            osh.push_back(dt);
        else
            throw Error("Expand: incompatible shapes");
        ost.push_back((oi >= 0 && din != 1) ? istr[(size_t)oi] : 0);
    }
    return strided(x, osh, ost, 0, nullptr, out);
}
inline TensorView tile(const TensorView& x, const std::vector<int64_t>& repeats, Buffer& out) {  // math.rs:2249
    if (repeats.size() != x.dim()) throw Error("Tile: repeats length must match input rank");
    std::vector<int64_t> osh;
    for (size_t i = 0; i < x.dim(); ++i) osh.push_back(x.shape[i] * repeats[i]);
    return strided(x, osh, detail::row_major_strides(x.shape), 0, &x.shape, out);
}
inline std::vector<TensorView> split(const TensorView& x, int64_t axis, const std::vector<int64_t>& splits,
                                     std::vector<Buffer*>& outputs) {  // manipulation.rs:1091
    const int64_t nd = (int64_t)x.dim(), ax = axis < 0 ? axis + nd : axis;
    if (ax < 0 || ax >= nd) throw Error("Split: axis out of bounds");
    int64_t total = 0;
    for (int64_t s : splits) total += s;
    if (total != x.shape[(size_t)ax]) throw Error("Split: splits sum mismatch");
    const auto istr = detail::row_major_strides(x.shape);
    std::vector<TensorView> res;
    int64_t pos = 0;
    for (size_t i = 0; i < splits.size(); ++i) {
        std::vector<int64_t> osh = x.shape;
        osh[(size_t)ax] = splits[i];
        res.push_back(strided(x, osh, istr, pos * istr[(size_t)ax], nullptr, *outputs[i]));
        pos += splits[i];
    }
    return res;
}
inline TensorView concat(const std::vector<const TensorView*>& inputs, int64_t axis, Buffer& out) {  // manipulation.rs:108
    Shape sh;
    std::vector<LeleTensor> ts;
    for (const TensorView* t : inputs) ts.push_back(t->c());
    std::vector<const LeleTensor*> ps;
    for (const LeleTensor& t : ts) ps.push_back(&t);
    check(lele_hip_concat(ctx(), ps.data(), ps.size(), axis, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, inputs.empty() ? LELE_F32 : inputs[0]->dtype());
}
inline TensorView gather(const TensorView& data, const TensorView& indices, int64_t axis, Buffer& out) {
    Shape sh;
    LeleTensor td = data.c(), ti = indices.c();
    check(lele_hip_gather(ctx(), &td, &ti, axis, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, data.dtype());
}
inline TensorView gather_elements(const TensorView& x, const TensorView& indices, int64_t axis, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), ti = indices.c();
    check(lele_hip_gather_elements(ctx(), &tx, &ti, axis, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView max_pool2d(const TensorView& x, const std::vector<int64_t>& kernel_shape, const std::vector<int64_t>& strides,
                             const std::vector<int64_t>& pads, const std::vector<int64_t>& dilations, bool ceil_mode, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_max_pool2d(ctx(), &tx, kernel_shape.data(), kernel_shape.size(), strides.data(), strides.size(), pads.data(),
                              pads.size(), dilations.data(), dilations.size(), ceil_mode, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView resize_nearest(const TensorView& x, int64_t out_h, int64_t out_w, bool asymmetric, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_resize_nearest(ctx(), &tx, out_h, out_w, asymmetric, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
// ---- channel views (include/lele_hip.h, LelePitch): operands that are views of a wider tensor, results written into a window of an
// already reserved one.  `Window` = (offset, pitch) in elements of the destination; NULL: a dense result, the buffer is resized.
struct Window {
    int64_t offset = 0, pitch = 0;
};
namespace pv_detail {
inline LelePitch pitch_of(const TensorView* x, const TensorView* y, const Window* w) {
    return LelePitch{x ? x->pitch() : 0, y ? y->pitch() : 0, w ? w->offset : 0, w ? w->pitch : 0};
}
inline TensorView windowed(Buffer& out, const Shape& sh, const LelePitch& pv, int dt = LELE_F32) {
    return TensorView::from_device(out, sh.vec(), dt, pv.out_offset, pv.out_pitch);
}
}  // namespace pv_detail
inline TensorView copy_view(const TensorView& x, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c();
    const LelePitch pv = pv_detail::pitch_of(&x, nullptr, w);
    check(lele_hip_copy_pitched(ctx(), &tx, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv, x.dtype());
}
inline TensorView transpose_cp(const TensorView& x, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c();
    const LelePitch pv = pv_detail::pitch_of(&x, nullptr, w);
    check(lele_hip_transpose_cp_pitched(ctx(), &tx, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
inline TensorView binary_pitched(int op, const TensorView& a, const TensorView& b, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    const LelePitch pv = pv_detail::pitch_of(&a, &b, w);
    check(lele_hip_binary_pitched(ctx(), op, &ta, &tb, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
inline TensorView resize_nearest_pitched(const TensorView& x, int64_t out_h, int64_t out_w, bool asymmetric, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c();
    const LelePitch pv = pv_detail::pitch_of(&x, nullptr, w);
    check(lele_hip_resize_nearest_pitched(ctx(), &tx, out_h, out_w, asymmetric, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
inline TensorView max_pool2d_pitched(const TensorView& x, const std::vector<int64_t>& kernel_shape, const std::vector<int64_t>& strides,
                                     const std::vector<int64_t>& pads, const std::vector<int64_t>& dilations, bool ceil_mode, Buffer& out,
                                     const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c();
    const LelePitch pv = pv_detail::pitch_of(&x, nullptr, w);
    check(lele_hip_max_pool2d_pitched(ctx(), &tx, kernel_shape.data(), kernel_shape.size(), strides.data(), strides.size(), pads.data(), pads.size(),
                                      dilations.data(), dilations.size(), ceil_mode, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
inline TensorView conv2d_pitched(const TensorView& x, const TensorView& wt, const TensorView* bias, const std::vector<int64_t>& dilations, int64_t group,
                                 const std::vector<int64_t>& pads, const std::vector<int64_t>& strides, int act, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c(), tw = wt.c();
    Opt ob(bias);
    const LelePitch pv = pv_detail::pitch_of(&x, nullptr, w);
    check(lele_hip_conv2d_pitched(ctx(), &tx, &tw, ob.p, dilations.data(), dilations.size(), group, pads.data(), pads.size(), strides.data(), strides.size(),
                                  act, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
inline TensorView conv2d_res_pitched(const TensorView& x, const TensorView& wt, const TensorView* bias, const TensorView& res,
                                     const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                                     const std::vector<int64_t>& strides, int act, Buffer& out, const Window* w = nullptr) {
    Shape sh;
    LeleTensor tx = x.c(), tw = wt.c(), tr = res.c();
    Opt ob(bias);
    const LelePitch pv = pv_detail::pitch_of(&x, &res, w);
    check(lele_hip_conv2d_res(ctx(), &tx, &tw, ob.p, &tr, dilations.data(), dilations.size(), group, pads.data(), pads.size(), strides.data(),
                              strides.size(), act, &pv, out.raw(), sh.dims, &sh.rank));
    return pv_detail::windowed(out, sh, pv);
}
struct TopkOut {
    TensorView values, indices;
};
inline TopkOut topk(const TensorView& x, int64_t k, bool largest, Buffer& out_values, Buffer& out_indices) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_topk(ctx(), &tx, k, largest, out_values.raw(), out_indices.raw(), sh.dims, &sh.rank));
    return {TensorView::from_device(out_values, sh.vec()), TensorView::from_device(out_indices, sh.vec())};
}
inline TensorView cast_to_f32(const TensorView& x, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_cast(ctx(), &tx, LELE_F32, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView cast_to_i64(const TensorView& x, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_cast(ctx(), &tx, LELE_I64, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_I64);
}
// ---- ConvInteger family (conv2d.rs:2216-2761) and the app-side steps (examples/sensevoice/src/{audio,tokenizer}.rs)
inline TensorView conv_integer(const TensorView& x, const TensorView& w, const TensorView* x_zero_point,
                               const TensorView* w_zero_point, const std::vector<int64_t>& dilations, int64_t group,
                               const std::vector<int64_t>& pads, const std::vector<int64_t>& strides, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c();
    Opt zx(x_zero_point), zw(w_zero_point);
    check(lele_hip_conv_integer(ctx(), &tx, &tw, zx.p, zw.p, dilations.data(), dilations.size(), group, pads.data(), pads.size(),
                                strides.data(), strides.size(), out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
struct ConvIntOut {
    TensorView out, scale;  // scale: [1] f32 on the device (upstream returns a host float)
};
inline ConvIntOut conv_integer_from_f32_multi(const std::vector<const TensorView*>& sources, const TensorView& w,
                                              const TensorView* w_zero_point, const std::vector<int64_t>& dilations,
                                              int64_t group, const std::vector<int64_t>& pads,
                                              const std::vector<int64_t>& strides, Buffer& out, Buffer& out_scale) {
    Shape sh;
    std::vector<LeleTensor> ts;
    for (const TensorView* t : sources) ts.push_back(t->c());
    std::vector<const LeleTensor*> ps;
    for (const LeleTensor& t : ts) ps.push_back(&t);
    LeleTensor tw = w.c();
    Opt zw(w_zero_point);
    check(lele_hip_conv_integer_from_f32(ctx(), ps.data(), ps.size(), &tw, zw.p, dilations.data(), dilations.size(), group,
                                         pads.data(), pads.size(), strides.data(), strides.size(), out.raw(), out_scale.raw(),
                                         sh.dims, &sh.rank));
    return {TensorView::from_device(out, sh.vec()), TensorView::from_device(out_scale, {1})};
}
inline ConvIntOut conv_integer_from_f32(const TensorView& x, const TensorView& w, const TensorView* w_zero_point,
                                        const std::vector<int64_t>& dilations, int64_t group, const std::vector<int64_t>& pads,
                                        const std::vector<int64_t>& strides, Buffer& out, Buffer& out_scale) {
    return conv_integer_from_f32_multi({&x}, w, w_zero_point, dilations, group, pads, strides, out, out_scale);
}
inline TensorView fused_scale_bias(const TensorView& data, const TensorView* scale_dev, float scale_mul, const TensorView& bias,
                                   bool silu, Buffer& out) {
    Shape sh;
    LeleTensor td = data.c(), tb = bias.c();
    Opt sc(scale_dev);
    check(lele_hip_fused_scale_bias(ctx(), &td, sc.p, scale_mul, &tb, silu, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView wav_to_f32(const TensorView& bytes, int bits_per_sample, int num_channels, Buffer& out) {
    Shape sh;
    LeleTensor tb = bytes.c();
    check(lele_hip_wav_to_f32(ctx(), &tb, bits_per_sample, num_channels, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView argmax_last(const TensorView& x, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_argmax_last(ctx(), &tx, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_I32);
}
inline std::pair<TensorView, TensorView> token_filter(const TensorView& ids, const TensorView& skip, Buffer& out_ids,
                                                      Buffer& out_counts) {  // tokenizer.rs:63-71
    Shape sh;
    LeleTensor ti = ids.c(), ts = skip.c();
    check(lele_hip_token_filter(ctx(), &ti, &ts, out_ids.raw(), out_counts.raw(), sh.dims, &sh.rank));
    std::vector<int64_t> s(sh.dims, sh.dims + sh.rank), r(sh.dims, sh.dims + (sh.rank > 0 ? sh.rank - 1 : 0));
    return {TensorView::from_device(out_ids, s, LELE_I32), TensorView::from_device(out_counts, r, LELE_I32)};
}
inline TensorView image_preprocess(const TensorView& rgb, int target, Buffer& out) {  // yolo26n-seg image.rs:62-111
    Shape sh;
    LeleTensor tr = rgb.c();
    check(lele_hip_image_preprocess(ctx(), &tr, target, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
struct SegOutputs {
    // one image: f32 [300, 38] (first `count` rows valid, zeros behind them), i32 [1], u8 [H, W]; a batch of N: [N, 300, 38], [N], [N, H, W]
    TensorView dets, count, mask;
};
inline SegOutputs yolo_seg_postprocess(const TensorView& logits, const TensorView& mask_features, int img_width, int img_height,
                                       float threshold, int num_classes, Buffer& dets, Buffer& count, Buffer& mask) {  // image.rs:127-265
    LeleTensor tl = logits.c(), tm = mask_features.c();
    check(lele_hip_yolo_seg_postprocess(ctx(), &tl, &tm, img_width, img_height, threshold, num_classes, dets.raw(), count.raw(),
                                        mask.raw()));
    const int64_t n = logits.size() / (300 * 38);
    if (n == 1)
        return {TensorView::from_device(dets, {300, 38}, LELE_F32), TensorView::from_device(count, {1}, LELE_I32),
                TensorView::from_device(mask, {img_height, img_width}, LELE_U8)};
    return {TensorView::from_device(dets, {n, 300, 38}, LELE_F32), TensorView::from_device(count, {n}, LELE_I32),
            TensorView::from_device(mask, {n, img_height, img_width}, LELE_U8)};
}
// examples/silero/src/main.rs:151-228: speech segments (sample indices) from per-chunk probabilities; host-only
struct VadConfig {  // main.rs:18-28
    float threshold = 0.3f, min_silence_ms = 200.0f, min_speech_ms = 400.0f, speech_pad_ms = 120.0f, merge_gap_ms = 200.0f;
};
inline std::vector<std::pair<size_t, size_t>> vad_segments(const std::vector<float>& probs, size_t chunk_size, size_t padded_len,
                                                           size_t audio_len, uint32_t sample_rate, const VadConfig& cfg = VadConfig()) {
    auto ms_to_samples = [&](float ms) { return (size_t)std::lround((float)sample_rate * (ms / 1000.0f)); };
    const size_t min_silence = std::max<size_t>(ms_to_samples(cfg.min_silence_ms), 1);
    const size_t min_speech = std::max<size_t>(ms_to_samples(cfg.min_speech_ms), 1);
    const size_t speech_pad = ms_to_samples(cfg.speech_pad_ms), merge_gap = ms_to_samples(cfg.merge_gap_ms);
    std::vector<std::pair<size_t, size_t>> segs, merged;
    bool triggered = false;
    size_t start = 0, silence = 0;
    for (size_t i = 0; i < probs.size(); ++i) {
        const size_t offset = i * chunk_size, frame_end = std::min(offset + chunk_size, padded_len);
        if (probs[i] >= cfg.threshold) {
            if (!triggered) {
                triggered = true;
                start = offset > speech_pad ? offset - speech_pad : 0;
            }
            silence = 0;
        } else if (triggered) {
            silence += frame_end - offset;
            if (silence >= min_silence) {
                const size_t end = std::min(frame_end + speech_pad, audio_len);
                if (end > start && end - start >= min_speech) segs.emplace_back(start, end);
                triggered = false;
                silence = 0;
            }
        }
    }
    if (triggered && audio_len > start && audio_len - start >= min_speech) segs.emplace_back(start, audio_len);
    std::stable_sort(segs.begin(), segs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (const auto& seg : segs) {
        if (!merged.empty()) {
            auto& last = merged.back();
            const size_t gap = seg.first > last.second ? seg.first - last.second : 0;
            if (seg.first <= last.second || gap <= merge_gap) {
                last.second = std::max(last.second, seg.second);
                continue;
            }
        }
        merged.push_back(seg);
    }
    return merged;
}
// fused forms emitted by lele_amd.compiler (bit-identical to the sequences they replace)
inline TensorView fused_quantized_linear_residual(const TensorView& input, const TensorView& weight_int8, const TensorView& weight_scale,
                                                  const TensorView& weight_zero, const TensorView* bias, bool apply_relu, const TensorView& res1,
                                                  const TensorView* res2, Buffer& out) {
    Shape sh;
    LeleTensor ti = input.c(), tw = weight_int8.c(), ts = weight_scale.c(), tz = weight_zero.c(), t1 = res1.c();
    Opt ob(bias), o2(res2);
    check(lele_hip_fused_quantized_linear_residual(ctx(), &ti, &tw, &ts, &tz, ob.p, apply_relu, &t1, o2.p, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
// (x1, layer_norm(x1, ln_scale, ln_bias, -1, epsilon)) with x1 = fused_quantized_linear[_residual](..., res1, res2)
struct SumAndNorm {
    TensorView sum, norm;
};
inline SumAndNorm fused_quantized_linear_residual_ln(const TensorView& input, const TensorView& weight_int8, const TensorView& weight_scale,
                                                     const TensorView& weight_zero, const TensorView* bias, bool apply_relu, const TensorView* res1,
                                                     const TensorView* res2, const TensorView& ln_scale, const TensorView& ln_bias, float epsilon,
                                                     Buffer& out, Buffer& ln_out) {
    Shape sh;
    LeleTensor ti = input.c(), tw = weight_int8.c(), ts = weight_scale.c(), tz = weight_zero.c(), tg = ln_scale.c(), tb = ln_bias.c();
    Opt ob(bias), o1(res1), o2(res2);
    check(lele_hip_fused_quantized_linear_residual_ln(ctx(), &ti, &tw, &ts, &tz, ob.p, apply_relu, o1.p, o2.p, &tg, &tb, epsilon, out.raw(),
                                                      ln_out.raw(), sh.dims, &sh.rank));
    return {TensorView::from_device(out, sh.vec(), LELE_F32), TensorView::from_device(ln_out, sh.vec(), LELE_F32)};
}
// the output half of a SAN-M attention block: FSMN memory block of v_src as the first residual, projection, Adds, LayerNorm
inline SumAndNorm sanm_out_block(const TensorView& input, const TensorView& weight_int8, const TensorView& weight_scale, const TensorView& weight_zero,
                                 const TensorView* bias, bool apply_relu, const TensorView& v_src, const TensorView& fsmn_w, const TensorView* fsmn_bias,
                                 int64_t x_offset, int64_t pad_left, int64_t pad_right, const TensorView* res2, const TensorView& ln_scale,
                                 const TensorView& ln_bias, float epsilon, Buffer& out, Buffer& ln_out) {
    Shape sh;
    LeleTensor ti = input.c(), tw = weight_int8.c(), ts = weight_scale.c(), tz = weight_zero.c(), tv = v_src.c(), tf = fsmn_w.c(), tg = ln_scale.c(),
               tb = ln_bias.c();
    Opt ob(bias), ofb(fsmn_bias), o2(res2);
    check(lele_hip_sanm_out_block(ctx(), &ti, &tw, &ts, &tz, ob.p, apply_relu, &tv, &tf, ofb.p, x_offset, pad_left, pad_right, o2.p, &tg, &tb, epsilon,
                                  out.raw(), ln_out.raw(), sh.dims, &sh.rank));
    return {TensorView::from_device(out, sh.vec(), LELE_F32), TensorView::from_device(ln_out, sh.vec(), LELE_F32)};
}
// fused_quantized_linear[_residual](fused_quantized_linear(input, w1.., true), w2.., apply_relu2, res1, res2)
inline TensorView fused_ffn_quantized(const TensorView& input, const TensorView& w1_int8, const TensorView& w1_scale, const TensorView& w1_zero,
                                      const TensorView* b1, const TensorView& w2_int8, const TensorView& w2_scale, const TensorView& w2_zero,
                                      const TensorView* b2, bool apply_relu2, const TensorView* res1, const TensorView* res2, Buffer& out) {
    Shape sh;
    LeleTensor ti = input.c(), tw1 = w1_int8.c(), ts1 = w1_scale.c(), tz1 = w1_zero.c(), tw2 = w2_int8.c(), ts2 = w2_scale.c(), tz2 = w2_zero.c();
    Opt ob1(b1), ob2(b2), o1(res1), o2(res2);
    check(lele_hip_fused_ffn_quantized(ctx(), &ti, &tw1, &ts1, &tz1, ob1.p, &tw2, &ts2, &tz2, ob2.p, apply_relu2, o1.p, o2.p, out.raw(), sh.dims,
                                       &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline SumAndNorm fused_ffn_quantized_ln(const TensorView& input, const TensorView& w1_int8, const TensorView& w1_scale, const TensorView& w1_zero,
                                         const TensorView* b1, const TensorView& w2_int8, const TensorView& w2_scale, const TensorView& w2_zero,
                                         const TensorView* b2, bool apply_relu2, const TensorView* res1, const TensorView* res2,
                                         const TensorView& ln_scale, const TensorView& ln_bias, float epsilon, Buffer& out, Buffer& ln_out) {
    Shape sh;
    LeleTensor ti = input.c(), tw1 = w1_int8.c(), ts1 = w1_scale.c(), tz1 = w1_zero.c(), tw2 = w2_int8.c(), ts2 = w2_scale.c(), tz2 = w2_zero.c(),
               tg = ln_scale.c(), tb = ln_bias.c();
    Opt ob1(b1), ob2(b2), o1(res1), o2(res2);
    check(lele_hip_fused_ffn_quantized_ln(ctx(), &ti, &tw1, &ts1, &tz1, ob1.p, &tw2, &ts2, &tz2, ob2.p, apply_relu2, o1.p, o2.p, &tg, &tb, epsilon,
                                          out.raw(), ln_out.raw(), sh.dims, &sh.rank));
    return {TensorView::from_device(out, sh.vec(), LELE_F32), TensorView::from_device(ln_out, sh.vec(), LELE_F32)};
}
inline TensorView softmax_scaled(const TensorView& x, const TensorView& scale, int64_t axis, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), ts = scale.c();
    check(lele_hip_softmax_scaled(ctx(), &tx, &ts, (int32_t)axis, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView add3(const TensorView& a, const TensorView& b, const TensorView& c, Buffer& out) {
    Shape sh;
    LeleTensor ta = a.c(), tb = b.c(), tc = c.c();
    check(lele_hip_add3(ctx(), &ta, &tb, &tc, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView halves_pow_add_sqrt(const TensorView& x, int64_t axis, const std::vector<int64_t>& lo, const std::vector<int64_t>& hi,
                                      const TensorView& exp_lo, const TensorView& exp_hi, Buffer& out) {
    if (lo.size() != 2 || hi.size() != 2) throw Error("halves_pow_add_sqrt: lo and hi are [start, end] pairs");
    Shape sh;
    LeleTensor tx = x.c(), t0 = exp_lo.c(), t1 = exp_hi.c();
    check(lele_hip_halves_pow_add_sqrt(ctx(), &tx, (int32_t)axis, lo[0], lo[1], hi[0], hi[1], &t0, &t1, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
inline TensorView depthwise_conv1d_tlc(const TensorView& x, const TensorView& w, const TensorView* bias, int64_t pad_left,
                                       int64_t pad_right, bool relu, int64_t x_offset, bool add_input, Buffer& out) {
    Shape sh;
    LeleTensor tx = x.c(), tw = w.c();
    Opt ob(bias);
    check(lele_hip_depthwise_conv1d_tlc(ctx(), &tx, x_offset, &tw, ob.p, pad_left, pad_right, relu, add_input, out.raw(), sh.dims, &sh.rank));
    LELE_RET(out, LELE_F32);
}
// view operators: shape bookkeeping only (shape.rs:2-52, 105-185)
inline TensorView reshape(const TensorView& x, const std::vector<int64_t>& target) {
    const int64_t total = x.size();
    std::vector<int64_t> s;
    int64_t known = 1;
    int infer = -1;
    for (size_t i = 0; i < target.size(); ++i) {
        int64_t d = target[i];
        if (d == 0 && i < x.dim()) d = x.shape[i];  // 0 copies the input dimension
        if (d == -1) {
            infer = (int)i;
            d = 1;
        } else {
            known *= d;
        }
        s.push_back(d);
    }
    if (infer >= 0) s[(size_t)infer] = known ? total / known : 0;
    int64_t n = 1;
    for (int64_t d : s) n *= d;
    if (n != total) throw Error("Reshape: element count mismatch");
    return x.with_shape(s);
}
inline TensorView flatten(const TensorView& x, int64_t axis) {
    const int64_t nd = (int64_t)x.dim(), ax = axis < 0 ? axis + nd : axis;
    int64_t a = 1, b = 1;
    for (int64_t i = 0; i < nd; ++i) (i < ax ? a : b) *= x.shape[(size_t)i];
    return x.with_shape({a, b});
}
inline TensorView identity(const TensorView& x) { return x; }
inline TensorView unsqueeze(const TensorView& x, std::vector<int64_t> axes) {  // shape.rs:136-156
    std::vector<int64_t> s = x.shape;
    const int64_t rank = (int64_t)s.size() + (int64_t)axes.size();
    std::sort(axes.begin(), axes.end());
    for (int64_t a : axes) {
        const int64_t idx = a < 0 ? a + rank : a;
        if (idx <= (int64_t)s.size()) s.insert(s.begin() + idx, 1); else s.push_back(1);
    }
    return x.with_shape(s);
}
inline TensorView squeeze(const TensorView& x, const std::vector<int64_t>* axes) {  // shape.rs:157-185
    std::vector<int64_t> s;
    const int64_t nd = (int64_t)x.dim();
    for (int64_t i = 0; i < nd; ++i) {
        bool drop = x.shape[(size_t)i] == 1;
        if (drop && axes) {
            drop = false;
            for (int64_t a : *axes) drop = drop || (a < 0 ? a + nd : a) == i;
        }
        if (!drop) s.push_back(x.shape[(size_t)i]);
    }
    return x.with_shape(s);
}
// manipulation.rs:382-587: pads = [begin..., end...] (covering the trailing dims when shorter); mode constant / edge / reflect
inline TensorView pad(const TensorView& x, const std::vector<int64_t>& pads, const float* constant_value, const std::string& mode,
                      Buffer& out) {
    const size_t rank = x.dim();
    std::vector<int64_t> raw;
    for (int64_t p : pads) raw.push_back(std::max<int64_t>(0, p));
    if (raw.size() < rank * 2) {
        const size_t half = raw.size() / 2, missing = rank - half;
        std::vector<int64_t> full(rank * 2, 0);
        for (size_t i = 0; i < half; ++i) {
            full[missing + i] = raw[i];
            full[rank + missing + i] = raw[half + i];
        }
        raw = full;
    }
    const int m = mode == "constant" ? 0 : mode == "edge" ? 1 : mode == "reflect" ? 2 : -1;
    if (m < 0) throw Error("Pad: unknown mode " + mode);
    float cv = constant_value ? *constant_value : 0.0f;
    uint32_t bits;
    std::memcpy(&bits, &cv, 4);
    Shape sh;
    LeleTensor tx = x.c();
    check(lele_hip_pad(ctx(), &tx, raw.data(), m, bits, out.raw(), sh.dims, &sh.rank));
    return TensorView::from_device(out, sh.vec(), x.dtype());
}
inline TensorView constant_of_shape(const std::vector<int64_t>& shape, float value, Buffer& out) {  // shape.rs:122-135
    uint32_t bits;
    std::memcpy(&bits, &value, 4);
    Shape sh;
    check(lele_hip_fill(ctx(), shape.data(), (int32_t)shape.size(), LELE_F32, bits, out.raw(), sh.dims, &sh.rank));
    return TensorView::from_device(out, sh.vec(), LELE_F32);
}

#undef LELE_RET
}  // namespace kernels

namespace features {
using detail::Shape;

struct FeatureConfig {  // src/features/pipeline.rs:8-36 (Default)
    int64_t sample_rate = 16000;
    int64_t n_mels = 80;
    float frame_length_ms = 25.0f;
    float frame_shift_ms = 10.0f;
    int64_t lfr_m = 7;
    int64_t lfr_n = 6;
};

class SenseVoiceFrontend {  // pipeline.rs:29-193
   public:
    explicit SenseVoiceFrontend(const FeatureConfig& cfg = FeatureConfig(), Ctx& ctx = Ctx::current()) {
        LeleFeatureConfig c{cfg.sample_rate, cfg.n_mels, cfg.frame_length_ms, cfg.frame_shift_ms, cfg.lfr_m, cfg.lfr_n};
        check(lele_hip_frontend_create(ctx.raw(), &c, &h_));
    }
    ~SenseVoiceFrontend() {
        if (h_) lele_hip_frontend_destroy(h_);
    }
    SenseVoiceFrontend(const SenseVoiceFrontend&) = delete;
    SenseVoiceFrontend& operator=(const SenseVoiceFrontend&) = delete;
    // pcm: f32 [len] -> [T, n_mels*lfr_m]; an empty TensorView when len < frame length (pipeline.rs:70-73)
    TensorView compute(const TensorView& pcm, Buffer& out) const {
        Shape sh;
        LeleTensor t = pcm.c();
        check(lele_hip_frontend_compute(h_, &t, out.raw(), sh.dims, &sh.rank));
        if (sh.rank == 0 || sh.dims[0] == 0) return TensorView();
        return TensorView::from_device(out, sh.vec());
    }
    TensorView compute_batch(const TensorView& pcm, Buffer& out) const {
        Shape sh;
        LeleTensor t = pcm.c();
        check(lele_hip_frontend_compute_batch(h_, &t, out.raw(), sh.dims, &sh.rank));
        return TensorView::from_device(out, sh.vec());
    }

   private:
    LeleFrontend* h_ = nullptr;
};

struct Cmvn {  // cmvn.rs
    float eps = 1e-5f;
    TensorView compute(const TensorView& x, Buffer& out) const {
        Shape sh;
        LeleTensor t = x.c();
        check(lele_hip_cmvn(detail::ctx(), &t, eps, out.raw(), sh.dims, &sh.rank));
        return TensorView::from_device(out, sh.vec());
    }
};
struct Lfr {  // lfr.rs
    int64_t m = 7, n = 6;
    TensorView compute(const TensorView& x, Buffer& out) const {
        Shape sh;
        LeleTensor t = x.c();
        check(lele_hip_lfr(detail::ctx(), &t, m, n, out.raw(), sh.dims, &sh.rank));
        return TensorView::from_device(out, sh.vec());
    }
};
}  // namespace features

}  // namespace lele
