// host_demo.cpp -- exercises lele_amd/host/lele.hpp (the C++ mirror of lele's Rust host interface) end to end.
//   host_demo probe                      : no GPU needed; checks that creating a context without a device fails loudly
//   host_demo run <pcm.f32> <out.f32> [<feats.f32>] : SenseVoiceFrontend -> Cmvn on the PCM file, writes [T,560] f32; also runs
//                                          the reference's matmul / layer_norm / transpose known-answer cases
// Driven by tests/test_host_cpp.py, which compares the output file with the oracle.
#include "lele.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

using lele::Buffer;
using lele::TensorView;
namespace K = lele::kernels;

static int fail(const char* what) {
    std::printf("FAIL %s\n", what);
    return 1;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "probe";
    if (mode == "probe") {
        int n = -1;
        lele_hip_device_count(&n);
        if (n > 0) {
            std::printf("PROBE devices=%d\n", n);
            return 0;
        }
        try {
            lele::Ctx ctx(0);
        } catch (const lele::Error& e) {
            std::printf("PROBE no-device error: %s\n", e.what());
            return 0;
        }
        return fail("context creation succeeded without a device");
    }
    if (mode == "latency") {  // host-side cost of one C-ABI call (device-resident operands, tiny kernel), and of a graph replay
        try {
            std::vector<float> h(512, 1.0f);
            Buffer a, b, o;
            a.upload(h.data(), h.size() * 4);
            b.upload(h.data(), h.size() * 4);
            TensorView ta = TensorView::from_device(a, {512}), tb = TensorView::from_device(b, {512});
            for (int i = 0; i < 100; ++i) K::add(ta, tb, o);
            lele::Ctx::current().sync();
            const int n = 20000;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) K::add(ta, tb, o);
            auto t1 = std::chrono::steady_clock::now();  // issue cost only (asynchronous)
            lele::Ctx::current().sync();
            auto t2 = std::chrono::steady_clock::now();
            const double issue = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
            const double total = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
            std::printf("LATENCY issue_us_per_call=%.3f end_to_end_us_per_call=%.3f calls=%d\n", issue, total, n);
            return 0;
        } catch (const lele::Error& e) {
            std::printf("FAIL lele::Error: %s\n", e.what());
            return 1;
        }
    }
    if (argc < 4) return fail("usage: host_demo run <pcm.f32> <out.f32>");
    try {
        // tests/verify_operators.rs:6-31 test_matmul_simple
        const float a[] = {1, 2, 3, 4, 5, 6}, b[] = {7, 8, 9, 10, 11, 12};
        Buffer o1, o2, o3;
        TensorView c = K::matmul(TensorView::from_slice(a, {2, 3}), TensorView::from_slice(b, {3, 2}), o1);
        const std::vector<float> cv = c.to_vec<float>();
        const float want[] = {58, 64, 139, 154};
        if (c.shape != std::vector<int64_t>{2, 2}) return fail("matmul shape");
        for (int i = 0; i < 4; ++i)
            if (std::fabs(cv[i] - want[i]) > 1e-5f) return fail("matmul value");
        // op chaining on device results: transpose(matmul) then add with itself
        TensorView ct = K::transpose(c, {1, 0}, o2);
        TensorView s = K::add(ct, ct, o3);
        const std::vector<float> sv = s.to_vec<float>();
        const float want2[] = {116, 278, 128, 308};
        for (int i = 0; i < 4; ++i)
            if (sv[i] != want2[i]) return fail("transpose/add chain");
        if (K::reshape(s, {-1}).shape != std::vector<int64_t>{4}) return fail("reshape view");
        // error path keeps the reference's message (gemm.rs:134-137 style shape check)
        bool threw = false;
        try {
            const float bad[] = {1, 2, 3, 4};
            K::matmul(TensorView::from_slice(a, {2, 3}), TensorView::from_slice(bad, {2, 2}), o1);
        } catch (const lele::Error&) {
            threw = true;
        }
        if (!threw) return fail("shape mismatch did not raise");

        std::ifstream in(argv[2], std::ios::binary | std::ios::ate);
        if (!in) return fail("cannot open pcm file");
        const size_t bytes = (size_t)in.tellg();
        std::vector<float> pcm(bytes / 4);
        in.seekg(0);
        in.read(reinterpret_cast<char*>(pcm.data()), (std::streamsize)bytes);
        lele::features::SenseVoiceFrontend fe;  // FeatureConfig::default()
        Buffer feats, normed;
        TensorView f = fe.compute(TensorView::from_slice(pcm.data(), {(int64_t)pcm.size()}), feats);
        if (f.is_empty()) return fail("front-end returned an empty tensor");
        if (argc > 4) {  // the intermediate [T,560] features, so the test can check each stage against the oracle
            const std::vector<float> fv = f.to_vec<float>();
            std::ofstream ff(argv[4], std::ios::binary);
            ff.write(reinterpret_cast<const char*>(fv.data()), (std::streamsize)(fv.size() * 4));
        }
        TensorView n = lele::features::Cmvn().compute(f, normed);
        const std::vector<float> out = n.to_vec<float>();
        std::ofstream of(argv[3], std::ios::binary);
        of.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * 4));
        std::printf("OK rows=%lld cols=%lld\n", (long long)n.shape[0], (long long)n.shape[1]);
        return 0;
    } catch (const lele::Error& e) {
        std::printf("FAIL lele::Error: %s\n", e.what());
        return 1;
    }
}
