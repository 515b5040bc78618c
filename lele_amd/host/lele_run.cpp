// lele_run -- run a compiled plan natively: plan JSON + weights.bin (lele's layout) through the C ABI, no Python.
//
//   lele_run <plan.json> <weights.bin> [--input name=file.bin:f32|i64:d0,d1,...]... [--out prefix] [--runs N] [--graph]
//
// Inputs are raw little-endian arrays; every plan output is written to <prefix><index>.bin (f32, row-major) and one JSON
// line with shapes and timings goes to stdout.  --graph records the statement sequence once as a hipGraph and times the
// replay (lele_hip_graph_*).  Build: g++ -std=c++17 -O2 -I include -I lele_amd/host lele_run.cpp -L lele_amd -llele_hip
#include <chrono>
#include <iostream>

#include "plan_runner.hpp"

using namespace lele;

static std::vector<char> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    return std::vector<char>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    try {
        if (argc < 3) {
            std::fprintf(stderr, "usage: lele_run <plan.json> <weights.bin> [--input name=file:dtype:dims]... [--out prefix] [--runs N] [--graph]\n");
            return 2;
        }
        const std::vector<char> text = read_file(argv[1]);
        plan::Runner runner(std::string(text.begin(), text.end()), argv[2]);
        std::map<std::string, plan::Val> inputs;
        std::vector<std::unique_ptr<Buffer>> keep;
        std::string prefix;
        int runs = 0;
        bool graph = false;
        for (int i = 3; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--input" && i + 1 < argc) {
                const std::string spec = argv[++i];
                const size_t eq = spec.find('='), c1 = spec.find(':', eq), c2 = spec.find(':', c1 + 1);
                if (eq == std::string::npos || c1 == std::string::npos || c2 == std::string::npos) throw Error("bad --input " + spec);
                const std::string name = spec.substr(0, eq), file = spec.substr(eq + 1, c1 - eq - 1), dt = spec.substr(c1 + 1, c2 - c1 - 1);
                std::vector<int64_t> dims;
                std::stringstream ss(spec.substr(c2 + 1));
                for (std::string tok; std::getline(ss, tok, ',');) dims.push_back(std::stoll(tok));
                const std::vector<char> raw = read_file(file);
                plan::Val v;
                if (dt == "i64") {  // integer inputs are host values (plan.py: "all i64 tensors live on the host")
                    v.kind = plan::Val::Host;
                    v.h.i.resize(raw.size() / 8);
                    std::memcpy(v.h.i.data(), raw.data(), v.h.i.size() * 8);
                    if (dims.size() > 1) {  // a rank > 1 integer tensor is only ever an operand of device ops: keep its shape
                        auto store = std::make_shared<std::vector<char>>(raw);
                        v.kind = plan::Val::Tensor;
                        v.keep = store;
                        v.t = TensorView::from_slice(reinterpret_cast<const int64_t*>(store->data()), dims);
                    }
                } else {
                    keep.push_back(std::make_unique<Buffer>());
                    keep.back()->upload(raw.data(), raw.size());
                    v.kind = plan::Val::Tensor;
                    v.t = TensorView::from_device(*keep.back(), dims);
                }
                inputs[name] = v;
            } else if (a == "--out" && i + 1 < argc) prefix = argv[++i];
            else if (a == "--runs" && i + 1 < argc) runs = std::atoi(argv[++i]);
            else if (a == "--graph") graph = true;
            else throw Error("unknown argument " + a);
        }
        Ctx& ctx = Ctx::current();
        std::vector<plan::Val> outs = runner.run(inputs);
        ctx.sync();
        std::ostringstream js;
        js << "{\"model\": \"" << runner.plan().at("source").str << "\", \"kernel_calls\": " << runner.calls() << ", \"outputs\": [";
        for (size_t k = 0; k < outs.size(); ++k) {
            if (outs[k].kind != plan::Val::Tensor) throw Error("plan output is not a tensor");
            const TensorView& t = outs[k].t;
            js << (k ? ", [" : "[");
            for (size_t d = 0; d < t.dim(); ++d) js << (d ? ", " : "") << t.shape[d];
            js << "]";
            if (!prefix.empty()) {
                const std::vector<float> v = t.to_vec<float>();
                std::ofstream f(prefix + std::to_string(k) + ".bin", std::ios::binary);
                f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * 4));
            }
        }
        js << "]";
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        if (runs > 0) {
            runner.run(inputs);
            ctx.sync();
            const auto t0 = now();
            for (int r = 0; r < runs; ++r) runner.run(inputs);
            ctx.sync();
            js << ", \"eager_ms\": " << ms(t0, now()) / runs;
            if (graph) {
                ctx.graph_begin();
                runner.run(inputs);
                Graph g(ctx.graph_end());
                g.launch();
                ctx.sync();
                const auto t1 = now();
                for (int r = 0; r < runs; ++r) g.launch();
                ctx.sync();
                js << ", \"graph_ms\": " << ms(t1, now()) / runs;
            }
        }
        js << "}";
        std::cout << js.str() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lele_run: %s\n", e.what());
        return 1;
    }
}
