// launch_chain.hip -- micro-benchmark: what would a HIP graph buy a small LoCoHD call on this runtime?
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 profiles/ubench/launch_chain.hip -o launch_chain && ./launch_chain
//
// A small from_primitives call (10^4 unique anchors; a rank's 125 000-pair share under strong scaling) is a chain of nine dependent
// launches on one stream -- memset, k_prep_count, k_prep_scan, k_prep_scatter, [k_pair_anchor_recs], k_env_group, k_pair_meta, k_sweep_duo,
// the INDIRECT companion -- of 3 .. 50 us each, followed by one host wait (kernel trace: DESIGN.md section 6, round 6).  The trace shows
// 4-5 us between the first five launches: the CPU's enqueue rate.  This program times the same SHAPE with stand-in kernels (a memset
// and eight kernels that spin for the durations of the trace) three ways: launched one by one, launched one by one from a thread that is
// already ahead of the GPU (the chain enqueued twice, only the second timed: the steady state of back-to-back calls without a wait), and
// replayed from an instantiated graph -- every time with the host wait a synchronous call ends in.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long long ticks, unsigned* sink) {  // one workgroup per CU busy for ~ticks of the 100 MHz constant clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] += 1u;
}

static const double kDurUs[8] = {4.7, 5.3, 5.3, 11.5, 49.0, 7.0, 23.0, 10.7};  // the unique-anchor call's kernels (profiles/r06, kernel trace)

static void enqueue_chain(hipStream_t s, void* zero, size_t zero_bytes, unsigned* sink, double scale) {
    CHECK(hipMemsetAsync(zero, 0, zero_bytes, s));
    for (int k = 0; k < 8; ++k) k_spin<<<256, 64, 0, s>>>((long long)(kDurUs[k] * scale * 100.0), sink);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    void* zero;
    unsigned* sink;
    const size_t zero_bytes = 96 << 10;
    CHECK(hipMalloc(&zero, zero_bytes));
    CHECK(hipMalloc(&sink, 64));
    double gpu_us = 0.0;
    for (double d : kDurUs) gpu_us += d;
    for (double scale : {1.0, 0.25}) {  // the trace's durations, and a chain of kernels a quarter as long (launch-bound)
        auto timed = [&](auto&& body) {
            for (int i = 0; i < 20; ++i) body();
            CHECK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) body();
            CHECK(hipStreamSynchronize(s));
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        const double eager_sync = timed([&] { enqueue_chain(s, zero, zero_bytes, sink, scale); CHECK(hipStreamSynchronize(s)); });
        const double eager_nosync = timed([&] { enqueue_chain(s, zero, zero_bytes, sink, scale); });
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue_chain(s, zero, zero_bytes, sink, scale);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double graph_sync = timed([&] { CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s)); });
        const double graph_nosync = timed([&] { CHECK(hipGraphLaunch(ge, s)); });
        printf("{\"chain\": \"memset + 8 kernels\", \"kernel_us_sum\": %.1f, \"eager_with_wait_us\": %.1f, \"eager_back_to_back_us\": %.1f, "
               "\"graph_with_wait_us\": %.1f, \"graph_back_to_back_us\": %.1f}\n",
               gpu_us * scale, eager_sync, eager_nosync, graph_sync, graph_nosync);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
