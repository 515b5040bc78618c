// plan_runner.hpp -- native execution of a compiled plan (lele_amd.compiler output, format "lele_amd.plan/2"; "/3" = the same after
// lele_amd.plan.fold_channel_views: channel views, windows, conv2d_res).
//
// The compiled counterpart of lele's generated `forward()`: lele's compiler emits Rust that is then compiled; here the
// same decisions are data (plan JSON + weights.bin in lele's layout, src/compiler/mod.rs:1381-1505) and this header
// walks the statement list through the C ABI via lele.hpp -- no Python anywhere on the serving path.  Header-only,
// C++17, no dependencies beyond lele.hpp.  Mirrors lele_amd/plan.py statement for statement (tests run both on the
// same plan and compare bits).
#pragma once
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "lele.hpp"

namespace lele {
namespace plan {

// ------------------------------------------------------------------------------------------------ JSON (reader only)
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false, is_int = false;
    double num = 0.0;
    int64_t inum = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* find(const std::string& key) const {
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(const std::string& key) const {
        const Json* j = find(key);
        if (!j) throw Error("plan: missing key '" + key + "'");
        return *j;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    int64_t as_int() const { return is_int ? inum : (int64_t)num; }
    double as_num() const { return is_int ? (double)inum : num; }
};

class JsonParser {
   public:
    explicit JsonParser(const std::string& text) : s_(text) {}
    Json parse() {
        Json j = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return j;
    }

   private:
    const std::string& s_;
    size_t i_ = 0;
    [[noreturn]] void fail(const char* what) const { throw Error(std::string("plan JSON: ") + what + " at byte " + std::to_string(i_)); }
    void ws() {
        while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
    }
    Json value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        Json j;
        if (c == '{') {
            j.kind = Json::Obj;
            ++i_;
            ws();
            if (s_[i_] == '}') return ++i_, j;
            for (;;) {
                ws();
                Json k = string_();
                ws();
                if (s_[i_++] != ':') fail("expected ':'");
                j.obj.emplace_back(k.str, value());
                ws();
                if (s_[i_] == ',') { ++i_; continue; }
                if (s_[i_] == '}') return ++i_, j;
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            j.kind = Json::Arr;
            ++i_;
            ws();
            if (s_[i_] == ']') return ++i_, j;
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (s_[i_] == ',') { ++i_; continue; }
                if (s_[i_] == ']') return ++i_, j;
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') return string_();
        if (s_.compare(i_, 4, "true") == 0) return i_ += 4, j.kind = Json::Bool, j.b = true, j;
        if (s_.compare(i_, 5, "false") == 0) return i_ += 5, j.kind = Json::Bool, j;
        if (s_.compare(i_, 4, "null") == 0) return i_ += 4, j;
        if (s_.compare(i_, 3, "NaN") == 0) return i_ += 3, j.kind = Json::Num, j.num = NAN, j;
        if (s_.compare(i_, 8, "Infinity") == 0) return i_ += 8, j.kind = Json::Num, j.num = INFINITY, j;
        if (s_.compare(i_, 9, "-Infinity") == 0) return i_ += 9, j.kind = Json::Num, j.num = -INFINITY, j;
        // number
        const size_t b = i_;
        bool flt = false;
        if (s_[i_] == '-') ++i_;
        while (i_ < s_.size() && (std::isdigit((unsigned char)s_[i_]) || s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E' || s_[i_] == '+' || s_[i_] == '-')) {
            flt = flt || s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E';
            ++i_;
        }
        if (b == i_) fail("unexpected character");
        const std::string tok = s_.substr(b, i_ - b);
        j.kind = Json::Num;
        if (flt) {
            j.num = std::strtod(tok.c_str(), nullptr);
        } else {
            j.is_int = true;
            j.inum = std::strtoll(tok.c_str(), nullptr, 10);
            j.num = (double)j.inum;
        }
        return j;
    }
    Json string_() {
        if (s_[i_] != '"') fail("expected string");
        ++i_;
        Json j;
        j.kind = Json::Str;
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                const char e = s_[i_++];
                if (e == 'n') c = '\n';
                else if (e == 't') c = '\t';
                else if (e == 'u') {  // plans hold ASCII names; keep the low byte of \uXXXX
                    c = (char)std::strtol(s_.substr(i_, 4).c_str(), nullptr, 16);
                    i_ += 4;
                } else c = e;
            }
            j.str.push_back(c);
        }
        ++i_;
        return j;
    }
};

// ------------------------------------------------------------------------------------------------ values
// A small host array for the integer / tiny-float side arithmetic (rank <= 1 here: shapes, axes, slice bounds).
struct HostArr {
    bool is_int = true;
    bool scalar = false;  // rank 0
    std::vector<int64_t> i;
    std::vector<float> f;
    size_t size() const { return is_int ? i.size() : f.size(); }
    double get(size_t k) const { return is_int ? (double)i[k] : (double)f[k]; }
};
struct Val {
    enum Kind { None, Tensor, Host, List } kind = None;
    std::vector<TensorView> list;  // split_owned results of a lifted plan
    TensorView t;
    HostArr h;
    std::shared_ptr<std::vector<char>> keep;  // storage of a host-backed tensor
};

inline std::vector<int64_t> ints_of(const HostArr& h) {
    std::vector<int64_t> v;
    for (size_t k = 0; k < h.size(); ++k) v.push_back(h.is_int ? h.i[k] : (int64_t)h.f[k]);
    return v;
}

// ------------------------------------------------------------------------------------------------ host ops (hostops.py)
inline HostArr host_eval(const std::string& op, const std::vector<const HostArr*>& x, const Json& attrs,
                         const std::vector<int64_t>* shape_of_first) {
    auto need = [&](size_t n) {
        if (x.size() < n) throw Error("host op " + op + ": operand missing");
    };
    HostArr r;
    if (op == "Shape" || op == "Size") {
        if (!shape_of_first) throw Error("host op " + op + ": no shape");
        if (op == "Shape") r.i = *shape_of_first;
        else {
            int64_t n = 1;
            for (int64_t d : *shape_of_first) n *= d;
            r.i = {n};
            r.scalar = true;
        }
        return r;
    }
    need(1);
    const HostArr& a = *x[0];
    if (op == "Identity") return a;
    if (op == "Cast") {
        const int64_t to = attrs.at("to").as_int();
        r.scalar = a.scalar;
        if (to == 7 || to == 6 || to == 9) {
            r.i = ints_of(a);
        } else {
            r.is_int = false;
            for (size_t k = 0; k < a.size(); ++k) r.f.push_back((float)a.get(k));
        }
        return r;
    }
    if (op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || op == "Expand" || op == "Transpose") {
        r = a;  // rank <= 1 arrays: only the scalar / vector distinction can change
        if (op == "Unsqueeze") r.scalar = false;
        if (op == "Squeeze" && a.size() == 1) r.scalar = true;
        if (op == "Reshape" && x.size() > 1 && x[1]) r.scalar = x[1]->size() == 0;
        if (op == "Expand" && x.size() > 1 && x[1] && x[1]->size() == 1 && a.size() == 1) {
            const int64_t n = x[1]->i[0];
            r.scalar = false;
            if (a.is_int) r.i.assign((size_t)n, a.i[0]); else r.f.assign((size_t)n, a.f[0]);
        }
        return r;
    }
    if (op == "Concat") {
        r.is_int = a.is_int;
        for (const HostArr* p : x) {
            if (!p) continue;
            if (p->is_int != r.is_int) throw Error("host Concat: mixed element types");
            r.i.insert(r.i.end(), p->i.begin(), p->i.end());
            r.f.insert(r.f.end(), p->f.begin(), p->f.end());
        }
        return r;
    }
    if (op == "Gather") {
        need(2);
        const HostArr& idx = *x[1];
        r.is_int = a.is_int;
        r.scalar = idx.scalar;
        for (size_t k = 0; k < idx.size(); ++k) {
            int64_t j = idx.i[k];
            if (j < 0) j += (int64_t)a.size();
            if (j < 0 || j >= (int64_t)a.size()) throw Error("host Gather: index out of range");
            if (a.is_int) r.i.push_back(a.i[(size_t)j]); else r.f.push_back(a.f[(size_t)j]);
        }
        return r;
    }
    if (op == "Slice") {
        need(3);
        int64_t st = x[1]->i.at(0), en = x[2]->i.at(0), step = x.size() > 4 && x[4] && x[4]->size() ? x[4]->i[0] : 1;
        const int64_t d = (int64_t)a.size();
        if (st < 0) st += d;
        if (en < 0) en += d;
        r.is_int = a.is_int;
        if (step > 0) {
            st = std::min(std::max<int64_t>(st, 0), d);
            en = std::min(std::max<int64_t>(en, 0), d);
            for (int64_t k = st; k < en; k += step) { if (a.is_int) r.i.push_back(a.i[(size_t)k]); else r.f.push_back(a.f[(size_t)k]); }
        } else {
            st = std::min(std::max<int64_t>(st, 0), d - 1);
            en = std::min(std::max<int64_t>(en, -1), d - 1);
            for (int64_t k = st; k > en; k += step) { if (a.is_int) r.i.push_back(a.i[(size_t)k]); else r.f.push_back(a.f[(size_t)k]); }
        }
        return r;
    }
    if (op == "Range") {
        need(3);
        r.is_int = a.is_int;
        if (a.is_int) for (int64_t v = a.i[0]; x[2]->i[0] > 0 ? v < x[1]->i[0] : v > x[1]->i[0]; v += x[2]->i[0]) r.i.push_back(v);
        else for (float v = a.f[0]; x[2]->f[0] > 0 ? v < x[1]->f[0] : v > x[1]->f[0]; v += x[2]->f[0]) r.f.push_back(v);
        return r;
    }
    if (op == "ConstantOfShape") {
        const std::vector<int64_t> shp = ints_of(a);
        if (shp.size() > 1) throw Error("host ConstantOfShape: rank > 1 is not supported by the native runner");
        const double v = attrs.has("value") ? attrs.at("value").as_num() : 0.0;
        r.is_int = attrs.has("value") && attrs.at("value").is_int;
        r.scalar = shp.empty();
        const size_t n = shp.empty() ? 1 : (size_t)shp[0];
        if (r.is_int) r.i.assign(n, (int64_t)v); else r.f.assign(n, (float)v);
        return r;
    }
    if (op == "Neg" || op == "Not") {
        r = a;
        for (auto& v : r.i) v = op == "Neg" ? -v : (v == 0);
        for (auto& v : r.f) v = op == "Neg" ? -v : (float)(v == 0.0f);
        return r;
    }
    if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div" || op == "Equal" || op == "Less" || op == "Greater" || op == "Max" || op == "Min" || op == "Where") {
        need(2);
        const bool where = op == "Where";
        const HostArr &p = where ? *x[1] : a, &q = where ? *x[2] : *x[1];
        const size_t n = std::max(std::max(p.size(), q.size()), where ? a.size() : (size_t)0);
        auto bc = [&](const HostArr& h, size_t k) { return h.get(h.size() == 1 ? 0 : k); };
        const bool cmp = op == "Equal" || op == "Less" || op == "Greater";
        r.is_int = cmp || (p.is_int && q.is_int);
        r.scalar = p.scalar && q.scalar && (!where || a.scalar);
        for (size_t k = 0; k < n; ++k) {
            const double u = bc(p, k), v = bc(q, k);
            double w;
            if (op == "Add") w = u + v;
            else if (op == "Sub") w = u - v;
            else if (op == "Mul") w = u * v;
            else if (op == "Div") w = r.is_int ? (v == 0 ? 0 : std::trunc(u / v)) : u / v;
            else if (op == "Equal") w = u == v;
            else if (op == "Less") w = u < v;
            else if (op == "Greater") w = u > v;
            else if (op == "Max") w = std::max(u, v);
            else if (op == "Min") w = std::min(u, v);
            else w = bc(a, k) != 0 ? u : v;
            if (r.is_int) r.i.push_back((int64_t)w); else r.f.push_back((float)w);
        }
        return r;
    }
    throw Error("host op " + op + " is not supported by the native runner");
}

// ------------------------------------------------------------------------------------------------ view chains (kernels.py view_copy)
inline bool reshape_strides(const std::vector<int64_t>& shape, const std::vector<int64_t>& strides, const std::vector<int64_t>& nw,
                            std::vector<int64_t>* out) {
    std::vector<std::pair<int64_t, int64_t>> old;
    for (size_t i = 0; i < shape.size(); ++i)
        if (shape[i] != 1) old.emplace_back(shape[i], strides[i]);
    out->assign(nw.size(), 0);
    size_t oi = 0, ni = 0;
    while (ni < nw.size() && oi < old.size()) {
        if (nw[ni] == 1) { ++ni; continue; }
        int64_t np = nw[ni], op = old[oi].first;
        size_t nj = ni + 1, oj = oi + 1;
        while (np != op) {
            if (np < op) {
                if (nj >= nw.size()) return false;
                np *= nw[nj++];
            } else {
                if (oj >= old.size()) return false;
                op *= old[oj++].first;
            }
        }
        for (size_t k = oi; k + 1 < oj; ++k)
            if (old[k].second != old[k + 1].first * old[k + 1].second) return false;
        (*out)[nj - 1] = old[oj - 1].second;
        for (size_t k = nj - 1; k > ni; --k) (*out)[k - 1] = (*out)[k] * nw[k];
        ni = nj;
        oi = oj;
    }
    if (oi < old.size()) return false;
    for (size_t k = ni; k < nw.size(); ++k)
        if (nw[k] != 1) return false;
    return true;
}

struct ViewGeom {
    std::vector<int64_t> shape, strides;
    int64_t offset = 0;
};
inline ViewGeom walk_chain(const std::vector<int64_t>& src_shape, const Json& chain) {
    ViewGeom g;
    g.shape = src_shape;
    g.strides = detail::row_major_strides(src_shape);
    for (const Json& step : chain.arr) {
        const std::string& kind = step.arr.at(0).str;
        const int64_t nd = (int64_t)g.shape.size();
        if (kind == "slice") {
            int64_t axis = step.arr.at(1).as_int();
            const int64_t start = step.arr.at(2).as_int(), len = step.arr.at(3).as_int();
            if (axis < 0) axis += nd;
            if (axis < 0 || axis >= nd || start < 0 || start + len > g.shape[(size_t)axis]) throw Error("view: slice outside the dimension");
            g.offset += start * g.strides[(size_t)axis];
            g.shape[(size_t)axis] = len;
        } else if (kind == "reshape") {
            std::vector<int64_t> tgt;
            for (const Json& d : step.arr.at(1).arr) tgt.push_back(d.as_int());
            int64_t total = 1, known = 1;
            for (int64_t d : g.shape) total *= d;
            int infer = -1;
            for (size_t i = 0; i < tgt.size(); ++i) {
                if (tgt[i] == 0 && i < g.shape.size()) tgt[i] = g.shape[i];
                if (tgt[i] == -1) infer = (int)i; else known *= tgt[i];
            }
            if (infer >= 0) tgt[(size_t)infer] = known ? total / known : 0;
            std::vector<int64_t> st;
            if (!reshape_strides(g.shape, g.strides, tgt, &st)) throw Error("view: this reshape needs a copy");
            g.shape = tgt;
            g.strides = st;
        } else if (kind == "transpose") {
            std::vector<int64_t> ns, nt;
            for (const Json& p : step.arr.at(1).arr) {
                int64_t q = p.as_int();
                if (q < 0) q += nd;
                ns.push_back(g.shape.at((size_t)q));
                nt.push_back(g.strides.at((size_t)q));
            }
            g.shape = ns;
            g.strides = nt;
        } else {
            throw Error("view: unknown step " + kind);
        }
    }
    return g;
}

// Intermediates of fused forms that ran as their node sequence.  The buffers live with the runner and are REUSED by position: a
// run takes them in the order its fallbacks execute, the next run rewinds and takes the same ones again -- repeated runs of a plan
// that keeps taking a fallback hold a constant number of device buffers and allocate nothing in the steady state.
struct FallbackPool {
    std::vector<std::unique_ptr<Buffer>> bufs;
    size_t next = 0;
    Buffer& take() {
        if (next == bufs.size()) bufs.push_back(std::make_unique<Buffer>());
        return *bufs[next++];
    }
    void rewind() { next = 0; }
};

// the chain run as the operators it stands for (slice, reshape of a contiguous tensor, transpose) with real copies: the fallback
// when the chain is not ONE strided view of its source -- shapes are not known when a plan is compiled.
inline TensorView materialise_chain(const TensorView& x, const Json& chain, Buffer* out, FallbackPool& pool) {
    TensorView cur = x;
    for (size_t i = 0; i < chain.arr.size(); ++i) {
        const Json& step = chain.arr[i];
        const std::string& kind = step.arr.at(0).str;
        const bool last = i + 1 == chain.arr.size();
        auto dst = [&]() -> Buffer& { return last && out ? *out : pool.take(); };
        if (kind == "slice") {
            const int64_t axis = step.arr.at(1).as_int(), start = step.arr.at(2).as_int(), len = step.arr.at(3).as_int();
            cur = kernels::slice(cur, {start}, {start + len}, {axis}, {1}, dst());
        } else if (kind == "reshape") {
            std::vector<int64_t> tgt;
            for (const Json& d : step.arr.at(1).arr) tgt.push_back(d.as_int());
            cur = kernels::reshape(cur, tgt);
        } else if (kind == "transpose") {
            std::vector<int64_t> perm;
            for (const Json& d : step.arr.at(1).arr) perm.push_back(d.as_int());
            cur = kernels::transpose(cur, perm, dst());
        } else {
            throw Error("view: unknown step " + kind);
        }
    }
    return cur;
}

inline TensorView view_copy(const TensorView& x, const Json& chain, Buffer& out, FallbackPool* pool = nullptr) {
    try {
        const ViewGeom g = walk_chain(x.shape, chain);
        return kernels::strided(x, g.shape, g.strides, g.offset, nullptr, out);
    } catch (const Error&) {
        if (!pool) throw;
        return materialise_chain(x, chain, &out, *pool);
    }
}

// kernels.py matmul_view: matmul of two views, the product optionally stored transposed (out_perm) and reshaped
inline TensorView matmul_view_direct(const TensorView& a, const Json& a_chain, const TensorView& b, const Json& b_chain,
                                     const std::vector<int64_t>* out_perm, const std::vector<int64_t>* out_reshape, Buffer& out);

// kernels.py matmul_view.  Geometries the strided GEMM does not take (a view that needs a copy, a rank-2 run-time B against a batched
// A, ...) run the node sequence the op stands for: materialise the views, `matmul`, transpose / reshape the product.
inline TensorView matmul_view(const TensorView& a, const Json& a_chain, const TensorView& b, const Json& b_chain,
                              const std::vector<int64_t>* out_perm, const std::vector<int64_t>* out_reshape, Buffer& out,
                              FallbackPool* pool = nullptr) {
    try {
        return matmul_view_direct(a, a_chain, b, b_chain, out_perm, out_reshape, out);
    } catch (const Error&) {
        if (!pool) throw;
    }
    TensorView am = a_chain.arr.empty() ? a : materialise_chain(a, a_chain, nullptr, *pool);
    TensorView bm = b_chain.arr.empty() ? b : materialise_chain(b, b_chain, nullptr, *pool);
    const bool post = out_perm || out_reshape;
    TensorView res = kernels::matmul(am, bm, post ? pool->take() : out);
    if (out_perm) res = kernels::transpose(res, *out_perm, out);
    if (out_reshape) res = kernels::reshape(res, *out_reshape);
    return res;
}

inline TensorView matmul_view_direct(const TensorView& a, const Json& a_chain, const TensorView& b, const Json& b_chain,
                                     const std::vector<int64_t>* out_perm, const std::vector<int64_t>* out_reshape, Buffer& out) {
    const ViewGeom ga = walk_chain(a.shape, a_chain), gb = walk_chain(b.shape, b_chain);
    const size_t r = ga.shape.size();
    if (r != gb.shape.size() || r < 2 || r > 4) throw Error("matmul_view: operand views must have equal rank 2..4");
    for (size_t i = 0; i + 2 < r; ++i)
        if (ga.shape[i] != gb.shape[i]) throw Error("matmul_view: batch dimensions differ");
    const int64_t m = ga.shape[r - 2], k = ga.shape[r - 1], n = gb.shape[r - 1];
    if (k != gb.shape[r - 2]) throw Error("MatMul K dim mismatch");
    const int64_t bo = r >= 3 ? ga.shape[0] : 1, bi = r == 4 ? ga.shape[1] : 1;
    auto view = [&](const std::vector<int64_t>& st, int64_t off) {
        return LeleMatView{off, r >= 3 ? st[0] : 0, r == 4 ? st[1] : 0, st[r - 2], st[r - 1]};
    };
    std::vector<int64_t> logical(ga.shape.begin(), ga.shape.end() - 2);
    logical.push_back(m);
    logical.push_back(n);
    std::vector<int64_t> perm;
    for (size_t i = 0; i < r; ++i) perm.push_back(out_perm ? ((*out_perm)[i] + (int64_t)r) % (int64_t)r : (int64_t)i);
    std::vector<int64_t> phys;
    for (int64_t p : perm) phys.push_back(logical[(size_t)p]);
    const std::vector<int64_t> pstr = detail::row_major_strides(phys);
    std::vector<int64_t> lstr(r, 0);
    for (size_t j = 0; j < r; ++j) lstr[(size_t)perm[j]] = pstr[j];
    std::vector<int64_t> oshape = phys;
    if (out_reshape) {
        Json fake;  // reuse the reshape rule of walk_chain on the contiguous physical shape
        fake.kind = Json::Arr;
        Json step, dims, name;
        step.kind = Json::Arr;
        dims.kind = Json::Arr;
        name.kind = Json::Str;
        name.str = "reshape";
        for (int64_t d : *out_reshape) {
            Json e;
            e.kind = Json::Num;
            e.is_int = true;
            e.inum = d;
            dims.arr.push_back(e);
        }
        step.arr = {name, dims};
        fake.arr.push_back(step);
        oshape = walk_chain(phys, fake).shape;
    }
    const LeleMatView av = view(ga.strides, ga.offset), bv = view(gb.strides, gb.offset), ov = view(lstr, 0);
    detail::Shape sh;
    LeleTensor ta = a.c(), tb = b.c();
    check(lele_hip_matmul_view(detail::ctx(), &ta, &av, &tb, &bv, bo, bi, m, k, n, &ov, oshape.data(), (int32_t)oshape.size(), out.raw(), sh.dims,
                               &sh.rank));
    return TensorView::from_device(out, sh.vec(), LELE_F32);
}

// kernels.py attention_view: softmax(Q K^T * scale) V in one launch when the kernel takes the geometry, else the three calls
inline TensorView attention_view(const TensorView& q, const Json& q_chain, const TensorView& k, const Json& k_chain, const TensorView& v,
                                 const Json& v_chain, const TensorView& scale, const std::vector<int64_t>* out_perm,
                                 const std::vector<int64_t>* out_reshape, Buffer& out, Buffer& tmp0, Buffer& tmp1, FallbackPool* pool = nullptr) {
    ViewGeom gq, gk, gv;
    bool is_view = true;
    try {
        gq = walk_chain(q.shape, q_chain), gk = walk_chain(k.shape, k_chain), gv = walk_chain(v.shape, v_chain);
    } catch (const Error&) {  // a chain that is not ONE strided view of its source: the three calls below materialise it, as the
        if (!pool) throw;     // unfused statements did
        is_view = false;
    }
    const size_t r = gq.shape.size();
    const char* off = getenv("LELE_HIP_ATTENTION_FUSED");
    bool fused = is_view && !(off && off[0] == '0') && r == gk.shape.size() && r == gv.shape.size() && r >= 2 && r <= 4;
    if (fused) {
        for (size_t i = 0; i + 2 < r; ++i) fused = fused && gq.shape[i] == gk.shape[i] && gq.shape[i] == gv.shape[i];
        fused = fused && gq.shape[r - 1] == 128 && gk.shape[r - 2] == 128 && gv.shape[r - 1] == 128 && gk.shape[r - 1] == gv.shape[r - 2] &&
                gk.shape[r - 1] <= 512 && gq.strides[r - 1] == 1 && gk.strides[r - 2] == 1 && gv.strides[r - 1] == 1 && gq.offset % 4 == 0 &&
                gk.offset % 4 == 0;
        for (size_t i = 0; fused && i + 1 < r; ++i) fused = gq.strides[i] % 4 == 0;
        for (size_t i = 0; fused && i < r; ++i) fused = i == r - 2 || gk.strides[i] % 4 == 0;
        if (fused && out_perm) fused = (((*out_perm)[r - 1] + (int64_t)r) % (int64_t)r) == (int64_t)r - 1;
        if (fused) {  // fewer than 96 blocks of 16 query rows: the three-call sequence (the library itself picks 16- or 32-row blocks)
            int64_t blocks = (gq.shape[r - 2] + 15) / 16;
            for (size_t i = 0; i + 2 < r; ++i) blocks *= gq.shape[i];
            const char* mb = getenv("LELE_HIP_ATTENTION_MIN_BLOCKS");
            fused = blocks >= (mb && *mb ? atoll(mb) : 96);
        }
    }
    if (!fused) {
        TensorView sc = matmul_view(q, q_chain, k, k_chain, nullptr, nullptr, tmp0, pool);
        TensorView pr = kernels::softmax_scaled(sc, scale, -1, tmp1);
        Json none;
        none.kind = Json::Arr;
        return matmul_view(pr, none, v, v_chain, out_perm, out_reshape, out, pool);
    }
    const int64_t tq = gq.shape[r - 2], dh = gq.shape[r - 1], tk = gk.shape[r - 1];
    const int64_t bo = r >= 3 ? gq.shape[0] : 1, bi = r == 4 ? gq.shape[1] : 1;
    auto view = [&](const std::vector<int64_t>& st, int64_t o) { return LeleMatView{o, r >= 3 ? st[0] : 0, r == 4 ? st[1] : 0, st[r - 2], st[r - 1]}; };
    std::vector<int64_t> logical(gq.shape.begin(), gq.shape.end() - 2);
    logical.push_back(tq);
    logical.push_back(dh);
    std::vector<int64_t> perm;
    for (size_t i = 0; i < r; ++i) perm.push_back(out_perm ? ((*out_perm)[i] + (int64_t)r) % (int64_t)r : (int64_t)i);
    std::vector<int64_t> phys;
    for (int64_t p : perm) phys.push_back(logical[(size_t)p]);
    const std::vector<int64_t> pstr = detail::row_major_strides(phys);
    std::vector<int64_t> lstr(r, 0);
    for (size_t j = 0; j < r; ++j) lstr[(size_t)perm[j]] = pstr[j];
    std::vector<int64_t> oshape = phys;
    if (out_reshape) {
        Json fake, step, dims, name;
        fake.kind = Json::Arr;
        step.kind = Json::Arr;
        dims.kind = Json::Arr;
        name.kind = Json::Str;
        name.str = "reshape";
        for (int64_t d : *out_reshape) {
            Json e;
            e.kind = Json::Num;
            e.is_int = true;
            e.inum = d;
            dims.arr.push_back(e);
        }
        step.arr = {name, dims};
        fake.arr.push_back(step);
        oshape = walk_chain(phys, fake).shape;
    }
    const LeleMatView qv = view(gq.strides, gq.offset), kv = view(gk.strides, gk.offset), vv = view(gv.strides, gv.offset), ov = view(lstr, 0);
    detail::Shape sh;
    LeleTensor tq_ = q.c(), tk_ = k.c(), tv_ = v.c(), ts_ = scale.c();
    check(lele_hip_attention_view(detail::ctx(), &tq_, &qv, &tk_, &kv, &tv_, &vv, bo, bi, tq, tk, dh, &ts_, &ov, oshape.data(), (int32_t)oshape.size(),
                                  out.raw(), sh.dims, &sh.rank));
    return TensorView::from_device(out, sh.vec(), LELE_F32);
}

// ------------------------------------------------------------------------------------------------ runner
class Runner {
   public:
    Runner(const std::string& plan_json, const std::string& weights_path) {
        plan_ = JsonParser(plan_json).parse();
        // "lele_amd.plan/2": compiled from ONNX (lele_amd.compiler); no format tag: lifted from lele-generated Rust
        // (tools/lift_generated.py) -- weights keyed by byte offset, output buffers named inside the argument lists
        // "lele_amd.plan/3": /2 after plan.fold_channel_views -- channel views (`chview`), Concat buffers sized up front (`reserve`),
        // results written into windows of them (`window` on a call), conv2d_res / copy_view / transpose_cp: the shape-specialised batch
        // form bench.py times.  Same weights keys as /2.
        v2_ = plan_.has("format") && (plan_.at("format").str == "lele_amd.plan/2" || plan_.at("format").str == "lele_amd.plan/3");
        if (plan_.has("format") && !v2_)
            throw Error("plan format \"" + plan_.at("format").str + "\" is not supported by the native runner (it runs lele_amd.plan/2, /3 and lifted plans)");
        std::ifstream f(weights_path, std::ios::binary);
        if (!f) throw Error("cannot open " + weights_path);
        blob_.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        for (const auto& kv : plan_.at("weights").obj) load_weight(kv.first, kv.second);
        for (const Json& s : plan_.at("slots").arr) slots_[s.str] = std::make_unique<Buffer>();
    }
    const Json& plan() const { return plan_; }
    size_t calls() const { return calls_; }

    // inputs by name -> outputs in plan order.  Device inputs must stay alive; i64 inputs are host arrays.
    std::vector<Val> run(const std::map<std::string, Val>& inputs) {
        env_.clear();
        for (const auto& kv : inputs) env_[kv.first] = kv.second;
        calls_ = 0;
        stmt_ = 0;
        fallback_pool_.rewind();
        exec(plan_.at("statements"));
        std::vector<Val> out;
        for (const Json& o : plan_.at("outputs").arr) out.push_back(env_.at(o.str));
        return out;
    }

   private:
    using TV = TensorView;
    Json plan_;
    bool v2_ = true;
    std::vector<char> blob_;
    std::unordered_map<std::string, std::unique_ptr<Buffer>> named_;  // `newbuf` buffers and split_owned outputs
    std::unordered_map<std::string, std::unique_ptr<Buffer>> slots_;
    std::unordered_map<std::string, std::pair<TV, std::shared_ptr<std::vector<char>>>> weights_;
    std::unordered_map<std::string, Val> env_;
    size_t calls_ = 0, stmt_ = 0;
    FallbackPool fallback_pool_;  // intermediates of fused forms that ran as their node sequence (rewound at every run)
    Buffer attn_tmp0_, attn_tmp1_;  // scores / probabilities of an attention_view statement that runs as the three-call sequence

    std::string wkey(const Json& w) const { return v2_ ? weight_key(w) : std::to_string(w.arr[1].as_int()); }
    static std::string weight_key(const Json& w) {
        std::string k = std::to_string(w.arr[1].as_int()) + ":" + w.arr[0].str + ":";
        for (size_t i = 0; i < w.arr[3].arr.size(); ++i) k += (i ? "x" : "") + std::to_string(w.arr[3].arr[i].as_int());
        return k;
    }
    static float half_to_float(uint16_t h) {
        const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 31, man = h & 1023;
        uint32_t bits;
        if (exp == 0) {
            if (man == 0) bits = sign;
            else {
                int e = -1;
                uint32_t m = man;
                do { ++e; m <<= 1; } while (!(m & 1024));
                bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 1023) << 13);
            }
        } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
        else bits = sign | ((exp + 112) << 23) | (man << 13);
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
    }
    void load_weight(const std::string& key, const Json& w) {  // plan.py load_weights_bin: everything but i64 is handed over as f32
        const bool four = w.arr.size() == 4;  // compiled: [kind, offset, bytes, shape]; lifted: key = offset, [kind, bytes, shape]
        const std::string& kind = w.arr[0].str;
        const size_t off = four ? (size_t)w.arr[1].as_int() : (size_t)std::stoll(key), len = (size_t)w.arr[four ? 2 : 1].as_int();
        if (off + len > blob_.size()) throw Error("weights.bin is shorter than view " + key);
        std::vector<int64_t> shape;
        for (const Json& d : w.arr[four ? 3 : 2].arr) shape.push_back(d.as_int());
        const char* p = blob_.data() + off;
        auto store = std::make_shared<std::vector<char>>();
        auto as_f32 = [&](size_t n, const std::function<float(size_t)>& get) {
            store->resize(n * 4);
            float* d = reinterpret_cast<float*>(store->data());
            for (size_t k = 0; k < n; ++k) d[k] = get(k);
            weights_[key] = {TV::weight(d, shape), store};
        };
        if (kind == "weight_f32") {
            store->assign(p, p + len);  // own 4-byte-aligned copy (the blob offset is 16-aligned, but keep it simple and safe)
            weights_[key] = {TV::weight(reinterpret_cast<const float*>(store->data()), shape), store};
        } else if (kind == "weight_u8") as_f32(len, [&](size_t k) { return (float)(uint8_t)p[k]; });
        else if (kind == "weight_i8") as_f32(len, [&](size_t k) { return (float)(int8_t)p[k]; });
        else if (kind == "weight_f16") as_f32(len / 2, [&](size_t k) { uint16_t h; std::memcpy(&h, p + 2 * k, 2); return half_to_float(h); });
        else if (kind == "weight_f64") as_f32(len / 8, [&](size_t k) { double v; std::memcpy(&v, p + 8 * k, 8); return (float)v; });
        else if (kind == "weight_i32" || kind == "weight_i32_f32") as_f32(len / 4, [&](size_t k) { int32_t v; std::memcpy(&v, p + 4 * k, 4); return (float)v; });
        else if (kind == "weight_i64_f32") as_f32(len / 8, [&](size_t k) { int64_t v; std::memcpy(&v, p + 8 * k, 8); return (float)v; });
        else if (kind == "weight_i64") {
            store->assign(p, p + len);
            weights_[key] = {TV::weight(reinterpret_cast<const int64_t*>(store->data()), shape), store};
        } else if (kind == "weight_i32_i64") {  // src/compiler/mod.rs:1162 / tensor.rs:237: little-endian i32 words widened to i64
            const size_t n = len / 4;
            store->resize(n * 8);
            int64_t* d = reinterpret_cast<int64_t*>(store->data());
            for (size_t k = 0; k < n; ++k) { int32_t v; std::memcpy(&v, p + 4 * k, 4); d[k] = (int64_t)v; }
            weights_[key] = {TV::weight(d, shape), store};
        } else throw Error("weights view kind '" + kind + "' is not handled");
    }

    // ---- argument evaluation
    const Val& ref(const std::string& name) const {
        auto it = env_.find(name);
        if (it == env_.end()) throw Error("plan: undefined value '" + name + "'");
        return it->second;
    }
    TV tensor(const Json& n) {
        if (n.has("ref")) {
            const Val& v = ref(n.at("ref").str);
            if (v.kind == Val::Tensor) return v.t;
            if (v.kind == Val::Host) return host_tensor(v.h);
            throw Error("plan: value '" + n.at("ref").str + "' is not a tensor");
        }
        if (n.has("some")) return tensor(n.at("some"));
        if (n.has("weight")) return weights_.at(wkey(n.at("weight"))).first;
        if (n.has("array")) {
            HostArr h;
            h.is_int = n.has("dtype") && n.at("dtype").str == "i64";
            for (const Json& e : n.at("array").arr) { if (h.is_int) h.i.push_back(e.as_int()); else h.f.push_back((float)e.as_num()); }
            return host_tensor(h);
        }
        throw Error("plan: tensor operand expected");
    }
    std::vector<std::shared_ptr<std::vector<char>>> temp_;  // host-backed operands of the current statement
    TV host_tensor(const HostArr& h) {
        auto store = std::make_shared<std::vector<char>>();
        std::vector<int64_t> shape;
        if (!h.scalar) shape.push_back((int64_t)h.size());
        temp_.push_back(store);
        if (h.is_int) {
            store->resize(h.i.size() * 8);
            std::memcpy(store->data(), h.i.data(), store->size());
            return TV::from_slice(reinterpret_cast<const int64_t*>(store->data()), shape);
        }
        store->resize(h.f.size() * 4);
        std::memcpy(store->data(), h.f.data(), store->size());
        return TV::from_slice(reinterpret_cast<const float*>(store->data()), shape);
    }
    bool is_none(const Json& n) const { return n.has("none"); }
    std::vector<int64_t> ints(const Json& n) {
        if (n.has("list")) {
            std::vector<int64_t> v;
            for (const Json& e : n.at("list").arr) v.push_back(e.at("int").as_int());
            return v;
        }
        if (n.has("ints")) {
            const Val& v = ref(n.at("ints").str);
            if (v.kind != Val::Host) throw Error("plan: '" + n.at("ints").str + "' is not a host integer value");
            return ints_of(v.h);
        }
        if (n.has("some")) return ints(n.at("some"));
        if (n.has("weight_list") || n.has("weight_scalar")) {
            const TensorView& w = weights_.at(wkey(n.has("weight_list") ? n.at("weight_list") : n.at("weight_scalar"))).first;
            std::vector<int64_t> v;
            if (w.dtype() == LELE_I64) v = w.to_vec<int64_t>();
            else for (float f : w.to_vec<float>()) v.push_back((int64_t)f);
            return v;
        }
        if (n.has("ref")) {
            const Val& v = ref(n.at("ref").str);
            if (v.kind == Val::Host) return ints_of(v.h);
        }
        throw Error("plan: integer list expected");
    }
    std::vector<float> floats(const Json& n) {
        const Json& l = n.has("some") ? n.at("some") : n;
        if (l.has("weight_list")) return weights_.at(wkey(l.at("weight_list"))).first.to_vec<float>();
        std::vector<float> v;
        for (const Json& e : l.at("list").arr) v.push_back((float)(e.has("float") ? e.at("float").as_num() : e.at("int").as_num()));
        return v;
    }
    int64_t integer(const Json& n) {
        if (n.has("first")) return ints(n.at("first")).at(0);
        if (n.has("weight_scalar")) return ints(n).at(0);
        return n.at("int").as_int();
    }
    static float number(const Json& n) { return (float)(n.has("float") ? n.at("float").as_num() : n.at("int").as_num()); }
    static bool boolean(const Json& n) { return n.at("bool").b; }

    int event_base_ = 0;  // DAG plans: this runner's first event id (one runner per plan per context here: 0)

    void exec(const Json& statements) {
        for (const Json& st : statements.arr) {
            ++stmt_;
            const std::string& op = st.at("op").str;
            // a DAG plan (lele_amd/lanes.py): the statement's lane, the points of other lanes it waits for, the event it leaves behind
            if (st.has("lane")) {
                check(lele_hip_lane_set(detail::ctx(), (int)st.at("lane").as_int()));
                if (st.has("wait"))
                    for (const Json& e : st.at("wait").arr) check(lele_hip_lane_wait(detail::ctx(), event_base_ + (int)e.as_int()));
            }
            if (op == "join") {  // a point on lane 0: behind the last statement of every side lane (the plan's end), or -- with a `record` and
                // nothing to wait for -- the run's starting point that every side lane's first statement waits for (lanes.py)
                check(lele_hip_lane_set(detail::ctx(), 0));
                for (const Json& e : st.at("wait").arr) check(lele_hip_lane_wait(detail::ctx(), event_base_ + (int)e.as_int()));
                if (st.has("record")) check(lele_hip_lane_record(detail::ctx(), event_base_ + (int)st.at("record").as_int()));
                continue;
            }
            struct Record {  // runs when the statement is done, whatever path it took
                const Json& st;
                int base;
                ~Record() {
                    if (st.has("record")) (void)lele_hip_lane_record(detail::ctx(), base + (int)st.at("record").as_int());
                }
            } record{st, event_base_};
            if (op == "host") host_stmt(st);
            else if (op == "if") if_stmt(st);
            else if (op == "call") call_stmt(st);
            else if (op == "ints") {   // `let x = &[..];`
                Val v;
                v.kind = Val::Host;
                for (const Json& e : st.at("value").arr) v.h.i.push_back(e.as_int());
                env_[st.at("out").arr[0].str] = v;
            } else if (op == "newbuf") {  // `let mut buf_x = Vec::new();` -> a persistent device buffer of that name
                if (!named_.count(st.at("out").arr[0].str)) named_[st.at("out").arr[0].str] = std::make_unique<Buffer>();
            } else if (op == "swap_remove") {  // Vec::swap_remove: take element i, the last element takes its place
                Val& lst = env_.at(st.at("list").str);
                const size_t i = (size_t)st.at("index").as_int();
                Val v;
                v.kind = Val::Tensor;
                v.t = lst.list.at(i);
                lst.list[i] = lst.list.back();
                lst.list.pop_back();
                env_[st.at("out").arr[0].str] = v;
            } else if (op == "alias") {
                env_[st.at("out").arr[0].str] = env_.at(st.at("src").str);
            } else if (op == "reserve") {   // the buffer of a Concat whose operands are written in place (fold_channel_views)
                Buffer& b = *slots_.at(st.at("slots").arr.at(0).str);
                std::vector<int64_t> shape;
                int64_t n = 1;
                for (const Json& d : st.at("shape").arr) shape.push_back(d.as_int()), n *= d.as_int();
                b.reserve((size_t)n * 4);
                Val v;
                v.kind = Val::Tensor;
                v.t = TV::from_device(b, shape, LELE_F32);
                env_[st.at("out").arr[0].str] = v;
            } else if (op == "chview") {    // channels [c0, c1) of a device tensor, no copy
                const Val& src = ref(st.at("src").str);
                if (src.kind != Val::Tensor) throw Error("plan: chview of a value that is not a tensor");
                Val v;
                v.kind = Val::Tensor;
                v.t = src.t.channels(st.at("c0").as_int(), st.at("c1").as_int());
                env_[st.at("out").arr[0].str] = v;
            } else throw Error("plan: statement kind '" + op + "' is not supported by the native runner");
        }
    }

    // `let (outs) = if cond.data[0] != 0 {..} else {..}` (lele: src/compiler/ops/control_flow.rs:18-150).  The condition is
    // read on the host (a device value is fetched: that waits for the stream and is refused inside a graph capture); the
    // taken branch's statements run; device results are copied into this statement's buffers (`.to_owned()` upstream).
    void if_stmt(const Json& st) {
        const Val& c = ref(st.at("cond").at("ref").str);
        bool taken = false;
        if (c.kind == Val::Host) taken = c.h.size() > 0 && (c.h.is_int ? c.h.i[0] != 0 : c.h.f[0] != 0.0f);
        else if (c.kind == Val::Tensor) {
            if (c.t.size() > 0) taken = c.t.dtype() == LELE_I64 ? c.t.to_vec<int64_t>()[0] != 0 : c.t.to_vec<float>()[0] != 0.0f;
        } else throw Error("plan: the condition of an `if` is neither a host value nor a tensor");
        const Json& arm = st.at(taken ? "then" : "else");
        exec(arm.at("statements"));
        size_t k = 0;
        Json none;
        none.kind = Json::Arr;
        for (size_t i = 0; i < st.at("out").arr.size(); ++i) {
            const Json& res = arm.at("results").arr.at(i);
            const std::string& name = st.at("out").arr[i].str;
            if (st.at("kinds").arr.at(i).str == "host") {
                if (res.has("ref")) { env_[name] = ref(res.at("ref").str); continue; }
                Val v;
                v.kind = Val::Host;
                const Json& cst = res.at("const");
                v.h.is_int = res.at("dtype").str == "i64";
                v.h.scalar = cst.kind != Json::Arr;
                auto push = [&](const Json& e) { if (v.h.is_int) v.h.i.push_back(e.as_int()); else v.h.f.push_back((float)e.as_num()); };
                if (cst.kind == Json::Arr) for (const Json& e : cst.arr) push(e); else push(cst);
                env_[name] = v;
            } else {
                temp_.clear();
                ++calls_;
                Val v;
                v.kind = Val::Tensor;
                v.t = view_copy(tensor(res), none, *slots_.at(st.at("slots").arr.at(k++).str));
                env_[name] = v;
            }
        }
    }

    void host_stmt(const Json& st) {
        const std::string& op = st.at("onnx").str;
        std::vector<HostArr> hold;
        hold.reserve(st.at("in").arr.size());
        std::vector<const HostArr*> x;
        std::vector<int64_t> shape0;
        bool have_shape = false;
        for (const Json& n : st.at("in").arr) {
            if (n.kind == Json::Null) { x.push_back(nullptr); continue; }
            HostArr h;
            if (n.has("const")) {
                const Json& c = n.at("const");
                h.is_int = n.at("dtype").str == "i64";
                h.scalar = c.kind != Json::Arr;
                auto push = [&](const Json& e) { if (h.is_int) h.i.push_back(e.as_int()); else h.f.push_back((float)e.as_num()); };
                if (c.kind == Json::Arr) for (const Json& e : c.arr) {
                    if (e.kind == Json::Arr) throw Error("host op " + op + ": rank > 1 constant is not supported by the native runner");
                    push(e);
                } else push(c);
            } else {
                const Val& v = ref(n.at("ref").str);
                if (v.kind == Val::Tensor) {
                    if (op != "Shape" && op != "Size") throw Error("host op " + op + " reads a device tensor");
                    if (!have_shape) shape0 = v.t.shape, have_shape = true;
                } else {
                    h = v.h;
                    if ((op == "Shape" || op == "Size") && !have_shape) { shape0 = h.scalar ? std::vector<int64_t>{} : std::vector<int64_t>{(int64_t)h.size()}; have_shape = true; }
                }
            }
            hold.push_back(std::move(h));
            x.push_back(&hold.back());
        }
        Val out;
        out.kind = Val::Host;
        out.h = host_eval(op, x, st.at("attrs"), have_shape ? &shape0 : nullptr);
        env_[st.at("out").arr[0].str] = out;
    }

    void set(const Json& st, size_t k, const TV& t) {
        Val v;
        v.kind = Val::Tensor;
        v.t = t;
        env_[st.at("out").arr.at(k).str] = v;
    }
    std::vector<Buffer*> bufs_;  // output buffers of the statement in flight: compiled plans name them in "slots",
                                 // lifted ones inside the argument list ({"slot"} workspace slots, {"buf"} named buffers)
    Buffer& slot(const Json&, size_t k) { return *bufs_.at(k); }

    void call_stmt(const Json& st) {
        namespace K = kernels;
        const std::string& fn = st.at("fn").str;
        std::vector<Json> a;
        bufs_.clear();
        if (st.has("slots"))
            for (const Json& sname : st.at("slots").arr) bufs_.push_back(slots_.at(sname.str).get());
        for (const Json& arg : st.at("args").arr) {
            if (arg.has("slot")) bufs_.push_back(slots_.at(arg.at("slot").str).get());
            else if (arg.has("buf")) {
                auto& b = named_[arg.at("buf").str];
                if (!b) b = std::make_unique<Buffer>();
                bufs_.push_back(b.get());
            } else a.push_back(arg);
        }
        temp_.clear();
        ++calls_;
        auto opt = [&](const Json& n, TV& hold) -> const TV* { if (is_none(n)) return nullptr; hold = tensor(n); return &hold; };
        TV h0, h1, h2, h3;
        // views
        if (fn == "reshape") return set(st, 0, K::reshape(tensor(a[0]), ints(a[1])));
        if (fn == "flatten") return set(st, 0, K::flatten(tensor(a[0]), integer(a[1])));
        if (fn == "unsqueeze") return set(st, 0, K::unsqueeze(tensor(a[0]), ints(a[1])));
        if (fn == "squeeze") { const auto ax = ints(a[1]); return set(st, 0, K::squeeze(tensor(a[0]), ax.empty() ? nullptr : &ax)); }
        if (fn == "identity") return set(st, 0, tensor(a[0]));
        if (fn == "split_owned") {  // owned results: one persistent buffer per output of THIS statement
            const std::vector<int64_t> sizes = ints(a[2]);
            std::vector<Buffer*> outs;
            for (size_t k = 0; k < sizes.size(); ++k) {
                auto& b = named_["split@" + std::to_string(stmt_) + "." + std::to_string(k)];
                if (!b) b = std::make_unique<Buffer>();
                outs.push_back(b.get());
            }
            Val v;
            v.kind = Val::List;
            v.list = K::split(tensor(a[0]), integer(a[1]), sizes, outs);
            env_[st.at("out").arr[0].str] = v;
            return;
        }
        // ---- channel views: a result that is a window of an already reserved tensor, and / or operands that are views (plan/3)
        kernels::Window win, *pw = nullptr;
        Buffer* wbuf = nullptr;
        if (st.has("window")) {
            const Val& whole = ref(st.at("window").at("of").str);
            if (whole.kind != Val::Tensor || !whole.t.buffer() || whole.t.dim() < 2) throw Error("plan: the window of '" + fn + "' is not inside a device tensor");
            int64_t inner = 1;
            for (size_t i = 2; i < whole.t.shape.size(); ++i) inner *= whole.t.shape[i];
            win.offset = whole.t.offset() + st.at("window").at("c0").as_int() * inner;
            win.pitch = whole.t.pitch() ? whole.t.pitch() : whole.t.shape[1] * inner;
            pw = &win;
            wbuf = const_cast<Buffer*>(whole.t.buffer());
        }
        auto dest = [&]() -> Buffer& { return wbuf ? *wbuf : slot(st, 0); };
        auto viewed = [&](std::initializer_list<size_t> which) {
            if (pw) return true;
            for (size_t k : which)
                if (a[k].has("ref")) {
                    const Val& v = ref(a[k].at("ref").str);
                    if (v.kind == Val::Tensor && v.t.is_view()) return true;
                }
            return false;
        };
        if (fn == "copy_view") return set(st, 0, K::copy_view(tensor(a[0]), dest(), pw));
        if (fn == "transpose_cp") return set(st, 0, K::transpose_cp(tensor(a[0]), dest(), pw));
        if (fn == "conv2d_res")
            return set(st, 0, K::conv2d_res_pitched(tensor(a[0]), tensor(a[1]), opt(a[2], h0), tensor(a[3]), ints(a[4]), integer(a[5]), ints(a[6]), ints(a[7]),
                                                    (int)integer(a[8]), dest(), pw));
        if ((fn == "conv2d" || fn == "conv2d_silu" || fn == "conv2d_fused") && viewed({0})) {
            const int act = fn == "conv2d_silu" ? LELE_ACT_SILU : (fn == "conv2d_fused" && boolean(a[7])) ? LELE_ACT_RELU : LELE_ACT_NONE;
            return set(st, 0, K::conv2d_pitched(tensor(a[0]), tensor(a[1]), opt(a[2], h0), ints(a[3]), integer(a[4]), ints(a[5]), ints(a[6]), act, dest(), pw));
        }
        if ((fn == "add" || fn == "sub" || fn == "mul" || fn == "div") && viewed({0, 1})) {
            const int bop = fn == "add" ? LELE_B_ADD : fn == "sub" ? LELE_B_SUB : fn == "mul" ? LELE_B_MUL : LELE_B_DIV;
            return set(st, 0, K::binary_pitched(bop, tensor(a[0]), tensor(a[1]), dest(), pw));
        }
        if (fn == "max_pool2d" && viewed({0}))
            return set(st, 0, K::max_pool2d_pitched(tensor(a[0]), ints(a[1]), ints(a[2]), ints(a[3]), ints(a[4]), boolean(a[5]), dest(), pw));
        if (pw && fn != "resize_nearest") throw Error("plan: '" + fn + "' cannot write into a window");
        Buffer& o = dest();
        static const std::map<std::string, int> unary = {{"exp", LELE_U_EXP}, {"sigmoid", LELE_U_SIGMOID}, {"tanh_kernel", LELE_U_TANH}, {"silu", LELE_U_SILU},
            {"erf", LELE_U_ERF}, {"relu", LELE_U_RELU}, {"sqrt", LELE_U_SQRT}, {"log", LELE_U_LOG}, {"sin", LELE_U_SIN}, {"cos", LELE_U_COS}, {"neg", LELE_U_NEG},
            {"reciprocal", LELE_U_RECIPROCAL}, {"softplus", LELE_U_SOFTPLUS}, {"not_", LELE_U_NOT}, {"abs", LELE_U_ABS}, {"floor", LELE_U_FLOOR}, {"ceil", LELE_U_CEIL}};
        static const std::map<std::string, int> binary = {{"add", LELE_B_ADD}, {"sub", LELE_B_SUB}, {"mul", LELE_B_MUL}, {"div", LELE_B_DIV}, {"pow", LELE_B_POW},
            {"max", LELE_B_MAX}, {"min", LELE_B_MIN}, {"equal", LELE_B_EQUAL}, {"less", LELE_B_LESS}, {"greater", LELE_B_GREATER}, {"prelu", LELE_B_PRELU},
            {"mod_f32", LELE_B_MOD}, {"and_", LELE_B_AND}, {"or_", LELE_B_OR}};
        if (unary.count(fn)) return set(st, 0, K::unary(unary.at(fn), tensor(a[0]), o));
        if (binary.count(fn)) return set(st, 0, K::binary(binary.at(fn), tensor(a[0]), tensor(a[1]), o));
        if (fn == "matmul") return set(st, 0, K::matmul(tensor(a[0]), tensor(a[1]), o));
        if (fn == "matmul_fused_add") return set(st, 0, K::matmul_fused_add(tensor(a[0]), tensor(a[1]), tensor(a[2]), o));
        if (fn == "gemm") return set(st, 0, K::gemm(tensor(a[0]), tensor(a[1]), opt(a[2], h0), number(a[3]), number(a[4]), boolean(a[5]), boolean(a[6]), o));
        if (fn == "conv2d" || fn == "conv1d" || fn == "conv2d_silu" || fn == "conv_transpose")
            return set(st, 0, (fn == "conv2d" ? K::conv2d : fn == "conv1d" ? K::conv1d : fn == "conv2d_silu" ? K::conv2d_silu : K::conv_transpose)(
                                  tensor(a[0]), tensor(a[1]), opt(a[2], h0), ints(a[3]), integer(a[4]), ints(a[5]), ints(a[6]), o));
        if (fn == "conv2d_fused" || fn == "conv1d_fused")
            return set(st, 0, (fn == "conv2d_fused" ? K::conv2d_fused : K::conv1d_fused)(tensor(a[0]), tensor(a[1]), opt(a[2], h0), ints(a[3]), integer(a[4]),
                                                                                         ints(a[5]), ints(a[6]), boolean(a[7]), o));
        if (fn == "conv_integer")
            return set(st, 0, K::conv_integer(tensor(a[0]), tensor(a[1]), opt(a[2], h0), opt(a[3], h1), ints(a[4]), integer(a[5]), ints(a[6]), ints(a[7]), o));
        if (fn == "fused_quantized_linear")
            return set(st, 0, K::fused_quantized_linear(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), boolean(a[5]), o));
        if (fn == "fused_quantized_linear_residual")
            return set(st, 0, K::fused_quantized_linear_residual(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), boolean(a[5]),
                                                                 tensor(a[6]), opt(a[7], h1), o));
        if (fn == "fused_quantized_linear_residual_ln") {
            auto r = K::fused_quantized_linear_residual_ln(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), boolean(a[5]), opt(a[6], h1),
                                                           opt(a[7], h2), tensor(a[8]), tensor(a[9]), number(a[10]), o, slot(st, 1));
            set(st, 0, r.sum), set(st, 1, r.norm);
            return;
        }
        if (fn == "sanm_out_block") {
            auto r = K::sanm_out_block(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), boolean(a[5]), tensor(a[6]), tensor(a[7]),
                                       opt(a[8], h1), integer(a[9]), integer(a[10]), integer(a[11]), opt(a[12], h2), tensor(a[13]), tensor(a[14]),
                                       number(a[15]), o, slot(st, 1));
            set(st, 0, r.sum), set(st, 1, r.norm);
            return;
        }
        if (fn == "fused_ffn_quantized")
            return set(st, 0, K::fused_ffn_quantized(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), tensor(a[5]), tensor(a[6]),
                                                     tensor(a[7]), opt(a[8], h1), boolean(a[9]), opt(a[10], h2), opt(a[11], h3), o));
        if (fn == "fused_ffn_quantized_ln") {
            auto r = K::fused_ffn_quantized_ln(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), opt(a[4], h0), tensor(a[5]), tensor(a[6]), tensor(a[7]),
                                               opt(a[8], h1), boolean(a[9]), opt(a[10], h2), opt(a[11], h3), tensor(a[12]), tensor(a[13]), number(a[14]), o,
                                               slot(st, 1));
            set(st, 0, r.sum), set(st, 1, r.norm);
            return;
        }
        if (fn == "mat_mul_integer") return set(st, 0, K::mat_mul_integer(tensor(a[0]), tensor(a[1]), opt(a[2], h0), opt(a[3], h1), o));
        if (fn == "dynamic_quantize_linear") {
            auto r = K::dynamic_quantize_linear(tensor(a[0]), o, slot(st, 1), slot(st, 2));
            set(st, 0, r.y), set(st, 1, r.scale), set(st, 2, r.zero_point);
            return;
        }
        if (fn == "lstm") {
            auto r = K::lstm(tensor(a[0]), tensor(a[1]), tensor(a[2]), opt(a[3], h0), opt(a[4], h1), opt(a[5], h2), opt(a[6], h3), o, slot(st, 1), slot(st, 2));
            set(st, 0, r.y), set(st, 1, r.h), set(st, 2, r.c);
            return;
        }
        if (fn == "gru") {
            auto r = K::gru(tensor(a[0]), tensor(a[1]), tensor(a[2]), opt(a[3], h0), opt(a[4], h1), boolean(a[5]), o, slot(st, 1));
            set(st, 0, r.y), set(st, 1, r.h);
            return;
        }
        if (fn == "layer_norm") return set(st, 0, K::layer_norm(tensor(a[0]), tensor(a[1]), tensor(a[2]), integer(a[3]), number(a[4]), o));
        if (fn == "batch_norm") return set(st, 0, K::batch_norm(tensor(a[0]), tensor(a[1]), tensor(a[2]), tensor(a[3]), tensor(a[4]), number(a[5]), o));
        if (fn == "softmax") return set(st, 0, K::softmax(tensor(a[0]), integer(a[1]), o));
        if (fn == "softmax_scaled") return set(st, 0, K::softmax_scaled(tensor(a[0]), tensor(a[1]), integer(a[2]), o));
        if (fn == "add3") return set(st, 0, K::add3(tensor(a[0]), tensor(a[1]), tensor(a[2]), o));
        if (fn == "halves_pow_add_sqrt") return set(st, 0, K::halves_pow_add_sqrt(tensor(a[0]), integer(a[1]), ints(a[2]), ints(a[3]), tensor(a[4]), tensor(a[5]), o));
        if (fn == "depthwise_conv1d_tlc")
            return set(st, 0, K::depthwise_conv1d_tlc(tensor(a[0]), tensor(a[1]), opt(a[2], h0), integer(a[3]), integer(a[4]), boolean(a[5]), integer(a[6]),
                                                      boolean(a[7]), o));
        if (fn == "max_pool2d") return set(st, 0, K::max_pool2d(tensor(a[0]), ints(a[1]), ints(a[2]), ints(a[3]), ints(a[4]), boolean(a[5]), o));
        if (fn == "resize_nearest") {  // kernels.py resize_nearest: sizes win; scales multiply in f64 and truncate (conv2d.rs:1261-1382)
            const TV x = tensor(a[0]);
            if (x.dim() != 4) throw Error("Resize: expected rank-4 input");
            int64_t oh, ow;
            if (!is_none(a[2])) {
                const auto sz = ints(a[2]);
                if (sz.size() < 4) throw Error("Resize: sizes must have at least 4 elements");
                oh = sz[2], ow = sz[3];
                if (oh <= 0 || ow <= 0) throw Error("Resize: sizes H and W must be positive");
            } else if (!is_none(a[1])) {
                const auto sc = floats(a[1]);
                const double shh = sc.size() >= 3 ? (double)sc[2] : 1.0, sww = sc.size() >= 4 ? (double)sc[3] : 1.0;
                oh = (int64_t)((double)x.shape[2] * shh), ow = (int64_t)((double)x.shape[3] * sww);
            } else throw Error("Resize: either scales or sizes must be provided");
            if (pw || x.is_view()) return set(st, 0, K::resize_nearest_pitched(x, oh, ow, a[3].at("str").str == "asymmetric", o, pw));
            return set(st, 0, K::resize_nearest(x, oh, ow, a[3].at("str").str == "asymmetric", o));
        }
        if (fn == "transpose") {
            const TV x = tensor(a[0]);
            std::vector<int64_t> perm = ints(a[1]);
            if (st.has("may_alias") && perm.size() == x.shape.size()) {   // only size-1 axes move: no byte changes -> a view
                std::vector<int64_t> kept, shape;
                for (int64_t& p : perm) {
                    if (p < 0) p += (int64_t)perm.size();
                    if (x.shape.at((size_t)p) != 1) kept.push_back(p);
                    shape.push_back(x.shape.at((size_t)p));
                }
                if (std::is_sorted(kept.begin(), kept.end())) {
                    --calls_;
                    return set(st, 0, x.with_shape(shape));
                }
            }
            return set(st, 0, K::transpose(x, perm, o));
        }
        if (fn == "view_copy") return set(st, 0, view_copy(tensor(a[0]), a[1].at("chain"), o, &fallback_pool_));
        if (fn == "matmul_view") {
            std::vector<int64_t> perm, resh;
            if (!is_none(a[4])) perm = ints(a[4]);
            if (!is_none(a[5])) resh = ints(a[5]);
            return set(st, 0, matmul_view(tensor(a[0]), a[1].at("chain"), tensor(a[2]), a[3].at("chain"), is_none(a[4]) ? nullptr : &perm,
                                          is_none(a[5]) ? nullptr : &resh, o, &fallback_pool_));
        }
        if (fn == "attention_view") {
            std::vector<int64_t> perm, resh;
            if (!is_none(a[7])) perm = ints(a[7]);
            if (!is_none(a[8])) resh = ints(a[8]);
            return set(st, 0, attention_view(tensor(a[0]), a[1].at("chain"), tensor(a[2]), a[3].at("chain"), tensor(a[4]), a[5].at("chain"),
                                             tensor(a[6]), is_none(a[7]) ? nullptr : &perm, is_none(a[8]) ? nullptr : &resh, o, attn_tmp0_, attn_tmp1_, &fallback_pool_));
        }
        if (fn == "concat") {
            std::vector<TV> hold;
            if (a[0].has("refs")) for (const Json& e : a[0].at("refs").arr) hold.push_back(ref(e.str).t);
            else for (const Json& e : a[0].at("list").arr) hold.push_back(tensor(e));
            std::vector<const TV*> ptrs;
            for (const TV& t : hold) ptrs.push_back(&t);
            return set(st, 0, K::concat(ptrs, integer(a[1]), o));
        }
        if (fn == "where_op") return set(st, 0, K::where_op(tensor(a[0]), tensor(a[1]), tensor(a[2]), o));
        if (fn == "gather") return set(st, 0, K::gather(tensor(a[0]), tensor(a[1]), integer(a[2]), o));
        if (fn == "gather_elements") return set(st, 0, K::gather_elements(tensor(a[0]), tensor(a[1]), integer(a[2]), o));
        if (fn == "slice") return set(st, 0, K::slice(tensor(a[0]), ints(a[1]), ints(a[2]), ints(a[3]), ints(a[4]), o));
        if (fn == "expand") return set(st, 0, K::expand(tensor(a[0]), ints(a[1]), o));
        if (fn == "tile") return set(st, 0, K::tile(tensor(a[0]), ints(a[1]), o));
        if (fn == "split") {
            std::vector<Buffer*> outs = bufs_;
            auto r = K::split(tensor(a[0]), integer(a[1]), ints(a[2]), outs);
            for (size_t k = 0; k < r.size(); ++k) set(st, k, r[k]);
            return;
        }
        if (fn == "pad") {
            float cv = 0.0f;
            const float* pcv = nullptr;
            if (!is_none(a[2])) {
                const TV c = tensor(a[2]);
                const auto v = c.to_vec<float>();
                if (!v.empty()) cv = v[0], pcv = &cv;
            }
            return set(st, 0, K::pad(tensor(a[0]), ints(a[1]), pcv, a[3].at("str").str, o));
        }
        if (fn == "reduce_mean" || fn == "reduce_sum" || fn == "reduce_max" || fn == "reduce_l2") {
            const int op = fn == "reduce_sum" ? 0 : fn == "reduce_mean" ? 1 : fn == "reduce_max" ? 2 : 3;
            return set(st, 0, K::reduce(op, tensor(a[0]), ints(a[1]), boolean(a[2]), o));
        }
        if (fn == "clip") {
            float lo = 0, hi = 0;
            const float *plo = nullptr, *phi = nullptr;
            if (!is_none(a[1])) lo = tensor(a[1]).to_vec<float>().at(0), plo = &lo;
            if (!is_none(a[2])) hi = tensor(a[2]).to_vec<float>().at(0), phi = &hi;
            return set(st, 0, K::clip(tensor(a[0]), plo, phi, o));
        }
        if (fn == "cast_to_i64") return set(st, 0, K::cast_to_i64(tensor(a[0]), o));
        if (fn == "topk") {
            const TV x = tensor(a[0]);
            const int64_t axis = integer(a[2]);
            if (axis != -1 && axis != (int64_t)x.dim() - 1) throw Error("TopK: only the last axis is supported (conv2d.rs:1385)");
            auto r = K::topk(x, integer(a[1]), boolean(a[3]), o, slot(st, 1));
            set(st, 0, r.values), set(st, 1, r.indices);
            return;
        }
        if (fn == "constant_of_shape") return set(st, 0, K::constant_of_shape(ints(a[0]), number(a[1]), o));
        throw Error("plan: kernel '" + fn + "' is not supported by the native runner");
    }
};

}  // namespace plan
}  // namespace lele
