// Kernel microbenchmarks (GPU box): launch-boundary cost and linear-kernel bandwidth sweeps.
// Build: make -C kangaroo_amd/csrc microbench     Run: kangaroo_amd/csrc/microbench [what]
#include "../kangaroo_amd/csrc/kg_common.h"
#include "../kangaroo_amd/csrc/kg_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
void kg_set_error(const char*, hipError_t) {}

__global__ void k_empty() {}
__global__ void k_flag(const int* f, int* o) { if (*f == 0) return; o[threadIdx.x] = 1; }
__global__ void k_stream_read(const kg_u4* p, size_t n16, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        kg_u4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        kg_u4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static float time_graph(hipStream_t st, hipGraphExec_t ge, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <class F> static hipGraphExec_t capture(hipStream_t st, F f) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    f();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
}

static void bench_launch(hipStream_t st) {
    int* flag; int* o;
    CK(hipMalloc(&flag, 256)); CK(hipMalloc(&o, 4096));
    CK(hipMemset(flag, 0, 256));
    const int n = 200;
    for (int kind = 0; kind < 3; ++kind) {
        auto body = [&]() {
            for (int i = 0; i < n; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
                else if (kind == 1) hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, st);
                else hipLaunchKernelGGL(k_flag, dim3(512), dim3(512), 0, st, flag, o);
            }
        };
        hipGraphExec_t ge = capture(st, body);
        float ms = time_graph(st, ge, 20);
        // eager
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        body(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 5; ++r) body(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ems = 0; CK(hipEventElapsedTime(&ems, e0, e1));
        const char* names[3] = {"empty 1x64", "empty 512x512", "flag-return 512x512"};
        printf("launch %-22s graph %.2f us/kernel   eager %.2f us/kernel\n", names[kind], ms * 1e3 / n, ems * 1e3 / (5 * n));
    }
}

struct Shape { const char* name; int N, K; };

int main(int argc, char** argv) {
    std::string what = argc > 1 ? argv[1] : "all";
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    kg_linear_init();
    if (what == "all" || what == "launch") bench_launch(st);

    // a pool much larger than the 256 MiB Infinity Cache, walked cyclically
    const size_t pool_bytes = (size_t)3 << 30;
    char* pool; CK(hipMalloc(&pool, pool_bytes));
    CK(hipMemset(pool, 0x11, pool_bytes));
    KgDyn* dyn; CK(hipMalloc(&dyn, sizeof(KgDyn) * 2));
    CK(kg_launch_set_dyn(dyn, 1, 1, 0, st) == 0 ? hipSuccess : hipErrorUnknown);
    CK(kg_launch_set_dyn(dyn + 1, 1, 7, 0, st) == 0 ? hipSuccess : hipErrorUnknown);
    char* x; CK(hipMalloc(&x, 16 * 32768 * 2)); CK(hipMemset(x, 0, 16 * 32768 * 2));
    char* out; CK(hipMalloc(&out, 16 * 65536 * 2));
    char* nw; CK(hipMalloc(&nw, 65536)); CK(hipMemset(nw, 0, 65536));
    float* hp; CK(hipMalloc(&hp, 16 * 4096 * 16));
    unsigned* sink; CK(hipMalloc(&sink, 256));
    CK(hipStreamSynchronize(st));

    if (what == "all" || what == "stream") {
        for (size_t mb : {33, 90, 180, 1024}) {
            for (int blocks : {256, 512, 1024, 2048, 4096}) {
                size_t bytes = mb << 20;
                int reps = 20;
                hipGraphExec_t ge = capture(st, [&]() {
                    size_t off = 0;
                    for (int i = 0; i < reps; ++i) {
                        if (off + bytes > pool_bytes) off = 0;
                        hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, st, (const kg_u4*)(pool + off), bytes / 16, sink);
                        off += bytes;
                    }
                });
                float ms = time_graph(st, ge, 5) / reps;
                printf("stream-read %5zu MB blocks %5d: %.1f us  %.0f GB/s\n", mb, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
                CK(hipGraphExecDestroy(ge));
            }
        }
    }
    if (what == "all" || what == "linear") {
        Shape shapes[] = {{"qkv", 12288, 4096}, {"o", 4096, 4096}, {"gateup", 22016, 4096}, {"down", 4096, 11008}, {"head", 32000, 4096}};
        int only_rows = argc > 2 ? atoi(argv[2]) : 0;
        int force_pro = argc > 3 ? atoi(argv[3]) : -1;
        int set_rows = argc > 4 ? atoi(argv[4]) : 7;
        CK(kg_launch_set_dyn(dyn + 1, 1, set_rows, 0, st) == 0 ? hipSuccess : hipErrorUnknown);
        CK(hipStreamSynchronize(st));
        for (auto& sh : shapes) {
            for (int rows : {1, 7}) {
                if (only_rows && rows != only_rows) continue;
                float best = 1e9; int bb = 0, bw = 0;
                for (int nw_ : {2, 4, 8}) {
                    for (int nb : {256, 512, 768, 1024, 1536, 2048, 4096}) {
                        const int G = (sh.N + 15) / 16;
                        if (nb > G && nb != 256) { if (nb / 2 >= G) continue; }
                        size_t bytes = (size_t)((sh.N + 15) / 16 * 16) * sh.K * 2;
                        int reps = 16;
                        bool ok = true;
                        hipGraphExec_t ge = capture(st, [&]() {
                            size_t off = 0;
                            for (int i = 0; i < reps; ++i) {
                                if (off + bytes > pool_bytes) off = 0;
                                KgLinArgs a; memset(&a, 0, sizeof(a));
                                a.W = pool + off; a.N = sh.N; a.K = sh.K; a.dyn = dyn + (rows == 1 ? 0 : 1); a.x = x; a.x_stride = sh.K;
                                a.out = out; a.out_stride = sh.N; a.norm_w = nw; a.eps = 1e-6f;
                                int pro = (sh.K == 4096 && strcmp(sh.name, "o") != 0) ? PRO_RMSNORM : PRO_NONE;
                                if (force_pro >= 0) pro = force_pro;
                                if (kg_launch_linear(a, KG_F16, rows, pro, EPI_STORE, nb, nw_, st) != 0) ok = false;
                                off += bytes;
                            }
                        });
                        if (!ok) { printf("%s rows %d nb %d nw %d: launch refused\n", sh.name, rows, nb, nw_); CK(hipGraphExecDestroy(ge)); continue; }
                        float ms = time_graph(st, ge, 4) / reps;
                        printf("linear %-7s rows %d waves %d blocks %5d: %7.1f us %6.0f GB/s\n", sh.name, rows, nw_, nb, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
                        if (ms < best) { best = ms; bb = nb; bw = nw_; }
                        CK(hipGraphExecDestroy(ge));
                    }
                }
                size_t bytes = (size_t)sh.N * sh.K * 2;
                printf("BEST   %-7s rows %d: waves %d blocks %d  %.1f us  %.0f GB/s\n", sh.name, rows, bw, bb, best * 1e3, bytes / (best * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
