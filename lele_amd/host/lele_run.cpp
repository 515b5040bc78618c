// lele_run -- run a compiled plan natively: plan JSON + weights.bin (lele's layout) through the C ABI, no Python.
//
//   lele_run <plan.json> <weights.bin> [--input name=file.bin:f32|i64:d0,d1,...]... [--out prefix] [--runs N] [--graph]
//            [--ranks N [--decode]]
//
// --ranks N: the utterance-sharded serving loop without Python.  The process forks N ranks BEFORE touching HIP; rank r
// runs the plan on GPU r (LELE_HIP_DEVICE) with its own copy of the inputs (weak scaling: every rank a full shard), the ranks meet in
// an RCCL communicator opened through the C ABI (lele_hip_comm_init_file: rank 0 writes the unique id to a file, no MPI / torch), and
// with --decode the greedy arg-max of output 0 is taken on the device and ONE all-gather moves the i32 ids (lele_hip_comm_allgather_i32);
// rank 0 prints the JSON line (MAX over ranks of the step time).
//
// Inputs are raw little-endian arrays; every plan output is written to <prefix><index>.bin (f32, row-major) and one JSON
// line with shapes and timings goes to stdout.  --graph records the statement sequence once as a hipGraph and times the
// replay (lele_hip_graph_*).  Build: g++ -std=c++17 -O2 -I include -I lele_amd/host lele_run.cpp -L lele_amd -llele_hip
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <ctime>
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
        // --ranks: fork first (no HIP state exists yet), children carry their rank in the environment
        int ranks = 0;
        bool decode = false;
        for (int i = 3; i < argc; ++i) {
            if (std::string(argv[i]) == "--ranks" && i + 1 < argc) ranks = std::atoi(argv[i + 1]);
            if (std::string(argv[i]) == "--decode") decode = true;
        }
        int rank = 0, world = 1;
        std::string id_file;
        if (ranks >= 1 && !std::getenv("LELE_RANK")) {
            id_file = "/tmp/lele_run_" + std::to_string((long)getpid()) + ".id";
            std::remove(id_file.c_str());
            // the token the ranks of THIS launch look for in the rendezvous file (lele_hip_comm_init_file)
            if (!std::getenv("LELE_JOB_ID")) setenv("LELE_JOB_ID", ("lele_run-" + std::to_string((long)getpid()) + "-" + std::to_string((long long)time(nullptr))).c_str(), 1);
            std::vector<pid_t> kids;
            for (int r = 0; r < ranks; ++r) {
                const pid_t pid = fork();
                if (pid < 0) throw Error("fork failed");
                if (pid == 0) {
                    setenv("LELE_RANK", std::to_string(r).c_str(), 1);
                    setenv("LELE_WORLD", std::to_string(ranks).c_str(), 1);
                    setenv("LELE_COMM_FILE", id_file.c_str(), 1);
                    if (!std::getenv("LELE_RUN_SHARE_GPU")) setenv("LELE_HIP_DEVICE", std::to_string(r).c_str(), 1);
                    else setenv("LELE_HIP_FFN_ONE_LAUNCH", "0", 0);   // processes sharing a device: no kernel that waits for its own workgroups
                    rank = r;
                    break;
                }
                kids.push_back(pid);
            }
            if ((int)kids.size() == ranks) {  // the parent: wait for the ranks, pass the worst status on
                int rc = 0;
                for (pid_t k : kids) {
                    int st = 0;
                    waitpid(k, &st, 0);
                    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
                }
                std::remove(id_file.c_str());
                return rc;
            }
        }
        if (const char* e = std::getenv("LELE_RANK")) {
            rank = std::atoi(e);
            world = std::atoi(std::getenv("LELE_WORLD"));
            id_file = std::getenv("LELE_COMM_FILE");
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
            else if (a == "--ranks" && i + 1 < argc) ++i;
            else if (a == "--decode") {}
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
        if (ranks >= 1) {  // the exchange step: ids of every rank on every rank, through the C ABI's own communicator
            Comm comm(ctx, id_file, rank, world);
            Buffer ids_buf, all_buf;
            comm.barrier();
            const auto t2 = now();
            std::vector<plan::Val> o2 = runner.run(inputs);
            int64_t gathered[LELE_MAX_RANK] = {0};
            int32_t grank = 0;
            if (decode) {
                TensorView ids = kernels::argmax_last(o2[0].t, ids_buf);
                int64_t flat = 1;
                for (size_t d = 0; d < ids.dim(); ++d) flat *= ids.shape[d];
                const int64_t fshape[1] = {flat};
                LeleTensor ti{ids_buf.data(), fshape, 1, LELE_I32, LELE_MEM_DEVICE};
                check(lele_hip_comm_allgather_i32(comm.raw(), &ti, all_buf.raw(), gathered, &grank));
            }
            ctx.sync();
            const int64_t step_us = comm.max((int64_t)(ms(t2, now()) * 1000.0));
            js << ", \"ranks\": " << world << ", \"step_ms_max_over_ranks\": " << step_us / 1000.0;
            if (decode) {
                std::vector<int32_t> host((size_t)(gathered[0] * gathered[1]));
                all_buf.download(host.data(), host.size() * 4);
                int64_t sum = 0;
                for (int32_t v : host) sum += v;
                js << ", \"gathered_ids\": [" << gathered[0] << ", " << gathered[1] << "], \"ids_checksum\": " << sum;
            }
        }
        js << "}";
        if (rank == 0) std::cout << js.str() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "lele_run: %s\n", e.what());
        return 1;
    }
}
