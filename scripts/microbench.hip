// Kernel microbenchmarks (GPU box): launch-boundary cost and linear-kernel bandwidth sweeps.
// Build: make -C kangaroo_amd/csrc microbench     Run: kangaroo_amd/csrc/microbench [what]
#define KG_LIN_TLOG
#include "../kangaroo_amd/csrc/kg_linear_body.h"
__device__ unsigned long long kg_lin_tlog[64 * 1024 * 8];
#include "../kangaroo_amd/csrc/gemm.hip"       // the prefill GEMM kernels with their timeline stamps compiled in
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
void kg_set_error(const char*, hipError_t) {}

__global__ void k_empty() {}
__global__ void k_xcc(int* out) { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); if (threadIdx.x == 0) out[blockIdx.x] = v & 15; }
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

// Grid-wide barrier for a persistent kernel whose blocks are all resident: monotone counter,
// bounded spin (a wave that gives up raises *abort so that every later wait falls through).
__device__ __forceinline__ void grid_arrive(unsigned* ctr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* ctr, unsigned target, unsigned* abort_flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
#define AG __HIP_MEMORY_SCOPE_AGENT
// variant 0: one counter, acquire/release atomics.  1: one counter, relaxed atomics + explicit fences.
// 2: one flag word per block (packed), thread t polls block t's flag.  3: per-XCD counters (blockIdx % 8)
// summed by the pollers.
template <int VV> __device__ __forceinline__ void gbar(unsigned* base, unsigned phase, unsigned* abort_flag) {
    const unsigned nb = gridDim.x;
    constexpr int V = VV & 3;
    constexpr bool FENCE = VV < 4;
    __syncthreads();
    if (VV == 0) { grid_wait((grid_arrive(base), base), phase * nb, abort_flag); return; }
    if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (V == 0) __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, AG);
        if (V == 1) __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, AG);
        if (V == 2) __hip_atomic_store(base + blockIdx.x, phase, __ATOMIC_RELAXED, AG);
        if (V == 3) __hip_atomic_fetch_add(base + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, AG);
    }
    unsigned spins = 0;
    bool ok = false;
    if (V <= 1) { if (threadIdx.x == 0) while (!ok) { ok = __hip_atomic_load(base, __ATOMIC_RELAXED, AG) >= phase * nb; if (++spins > (1u << 22)) { *abort_flag = 1; break; } } }
    if (V == 2) { if (threadIdx.x < nb) while (!ok) { ok = __hip_atomic_load(base + threadIdx.x, __ATOMIC_RELAXED, AG) >= phase; if (++spins > (1u << 22)) { *abort_flag = 1; break; } } }
    if (V == 3) { if (threadIdx.x < 8) while (!ok) { ok = __hip_atomic_load(base + threadIdx.x * 32, __ATOMIC_RELAXED, AG) >= phase * (nb / 8); if (++spins > (1u << 22)) { *abort_flag = 1; break; } } }
    __syncthreads();
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <int V> __global__ __launch_bounds__(512) void k_barrier_chain(unsigned* ctr, unsigned* abort_flag, int phases, float* data) {
    const unsigned nb = gridDim.x;
    float v = 0.f;
    for (int p = 0; p < phases; ++p) {
        // dependent work: block b writes slot (p&1, b+1), block b+1 reads it after the barrier
        if (threadIdx.x == 0) __hip_atomic_store(data + (p & 1) * nb + (blockIdx.x + 1) % nb, v + 1.f, __ATOMIC_RELAXED, AG);
        gbar<V>(ctr, (unsigned)(p + 1), abort_flag);
        v = __hip_atomic_load(data + (p & 1) * nb + blockIdx.x, __ATOMIC_RELAXED, AG);
    }
    if (threadIdx.x == 0) data[2 * nb + blockIdx.x] = v;
}

// timeline copy of the verify pass's gate/up kernel (stand-alone form of the body, with stamps)
__global__ void __launch_bounds__(512) k_lin_tl(const KgLinArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    KgLinArgs a = a_in;
    // where X comes from: 0 one copy for all blocks, 1 a copy per block, 2 a copy per blockIdx % 8
    const size_t copy = (size_t)16 * a.K * 2;
    if (a.head_part_stride == 1) a.x = (const char*)a.x + copy * blockIdx.x;
    if (a.head_part_stride == 2) a.x = (const char*)a.x + copy * (blockIdx.x & 7);
    kg_linear_body<KG_F16, MODE_MFMA, PRO_RMSNORM, EPI_STORE, false>(a, smem, nullptr, blockDim.x >> 6);
}

// the same for a projection without a norm (o_proj of the verify pass)
__global__ void __launch_bounds__(512) k_lin_tl_o(const KgLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    kg_linear_body<KG_F16, MODE_MFMA, PRO_NONE, EPI_STORE, false>(a, smem, nullptr, blockDim.x >> 6);
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

    if (what == "all" || what == "barrier") {
        unsigned* ctr; float* data;
        CK(hipMalloc(&ctr, 8192)); CK(hipMalloc(&data, 8192));
        for (int variant = 0; variant < 8; ++variant) {
            for (int phases : {1000}) {
                const int nthreads = 512;
                CK(hipMemsetAsync(ctr, 0, 8192, st)); CK(hipMemsetAsync(data, 0, 8192, st));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, st));
                if (variant == 0) hipLaunchKernelGGL(k_barrier_chain<0>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 1) hipLaunchKernelGGL(k_barrier_chain<1>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 2) hipLaunchKernelGGL(k_barrier_chain<2>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 4) hipLaunchKernelGGL(k_barrier_chain<4>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 5) hipLaunchKernelGGL(k_barrier_chain<5>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 6) hipLaunchKernelGGL(k_barrier_chain<6>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 7) hipLaunchKernelGGL(k_barrier_chain<7>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                if (variant == 3) hipLaunchKernelGGL(k_barrier_chain<3>, dim3(256), dim3(nthreads), 0, st, ctr, ctr + 1024, phases, data);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                float h[768]; unsigned ab = 0;
                CK(hipMemcpy(h, data, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, ctr + 1024, 4, hipMemcpyDeviceToHost));
                printf("grid barrier variant %d, 256 blocks x %3d threads, %4d phases: %.2f us/phase  (abort %u, chain value %.0f expect %d)\n", variant, nthreads, phases, ms * 1e3 / phases, ab, h[512], phases);
            }
        }
    }

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
    if (what == "gemm") {
        // 128-row prefill passes at the 7B shapes, 16 launches per graph over rotating weights
        kg_gemm_init();
        const int rows = argc > 2 ? atoi(argv[2]) : 128;
        KgDyn* dynr; CK(hipMalloc(&dynr, sizeof(KgDyn)));
        CK(kg_launch_set_dyn(dynr, 1, rows, 0, st) == 0 ? hipSuccess : hipErrorUnknown);
        char* xr; CK(hipMalloc(&xr, (size_t)KG_GEMM_ROWS * 16384 * 2)); CK(hipMemset(xr, 0, (size_t)KG_GEMM_ROWS * 16384 * 2));
        char* outr; CK(hipMalloc(&outr, (size_t)KG_GEMM_ROWS * 32768 * 2));
        char* resr; CK(hipMalloc(&resr, (size_t)KG_GEMM_ROWS * 32768 * 2)); CK(hipMemset(resr, 0, (size_t)KG_GEMM_ROWS * 32768 * 2));
        float* kp; CK(hipMalloc(&kp, (size_t)4 * KG_GEMM_ROWS * 8192 * 4));
        struct { const char* name; int N, K, epi; } cfg[] = {{"qkv-like store", 12288, 4096, EPI_STORE}, {"o", 4096, 4096, EPI_RESID}, {"gateup", 22016, 4096, EPI_SILU_MUL},
                                                            {"down", 4096, 11008, EPI_RESID}, {"gateup 13B", 27648, 5120, EPI_SILU_MUL}, {"down 13B", 5120, 13824, EPI_RESID}};
        for (auto& c : cfg) {
            const int nlaunch = 16;
            const size_t bytes = (size_t)c.N * c.K * 2;
            hipGraphExec_t ge = capture(st, [&]() {
                size_t off = 0;
                for (int i = 0; i < nlaunch; ++i) {
                    if (off + bytes > pool_bytes) off = 0;
                    KgLinArgs a; memset(&a, 0, sizeof(a));
                    a.W = pool + off; a.N = c.N; a.K = c.K; a.dyn = dynr; a.x = xr; a.x_stride = c.K;
                    a.out = outr; a.out_stride = c.epi == EPI_SILU_MUL ? c.N / 2 : c.N; a.resid = resr; a.resid_stride = c.N; a.kpart = kp;
                    if (kg_launch_gemm(a, KG_F16, PRO_NONE, c.epi, st) != 0) printf("launch refused\n");
                    off += bytes;
                }
            });
            float ms = time_graph(st, ge, 3) / nlaunch;
            printf("gemm %-15s rows %3d: %6.1f us per pass  %5.0f GB/s of weights  %5.0f TFLOP/s\n", c.name, rows, ms * 1e3, bytes / (ms * 1e-3) / 1e9,
                   2.0 * rows * c.N * c.K / (ms * 1e-3) / 1e12);
            CK(hipGraphExecDestroy(ge));
        }
        return 0;
    }
    if (what == "gemm_timeline") {
        // 16 back-to-back 64-row GEMM passes (gate/up shape, then down shape): per block, the time spent
        // in each part of a pipeline step, summed over the steps
        kg_gemm_init();
        KgDyn* dyn64; CK(hipMalloc(&dyn64, sizeof(KgDyn)));
        CK(kg_launch_set_dyn(dyn64, 1, 64, 0, st) == 0 ? hipSuccess : hipErrorUnknown);
        char* x64; CK(hipMalloc(&x64, (size_t)64 * 16384 * 2)); CK(hipMemset(x64, 0, (size_t)64 * 16384 * 2));
        char* out64; CK(hipMalloc(&out64, (size_t)64 * 32768 * 2));
        struct { const char* name; int N, K, epi; } cfg[] = {{"gateup", 22016, 4096, EPI_SILU_MUL}, {"qkv-like store", 12288, 4096, EPI_STORE}, {"down", 4096, 11008, EPI_STORE}};
        for (auto& c : cfg) {
            const int nlaunch = 16;
            const size_t bytes = (size_t)c.N * c.K * 2;
            const int nb = kg_gemm_grid_x(c.N, 1), nw_ = (c.N / 16 + nb - 1) / nb;      // nw_: row groups per block
            hipGraphExec_t ge = capture(st, [&]() {
                size_t off = 0;
                for (int i = 0; i < nlaunch; ++i) {
                    if (off + bytes > pool_bytes) off = 0;
                    KgLinArgs a; memset(&a, 0, sizeof(a));
                    a.W = pool + off; a.N = c.N; a.K = c.K; a.dyn = dyn64; a.x = x64; a.x_stride = c.K;
                    a.out = out64; a.out_stride = c.epi == EPI_SILU_MUL ? c.N / 2 : c.N; a.logits_stride = i;
                    if (kg_launch_gemm(a, KG_F16, PRO_NONE, c.epi, st) != 0) printf("launch refused\n");
                    off += bytes;
                }
            });
            float ms = time_graph(st, ge, 2) / nlaunch;
            std::vector<unsigned long long> t((size_t)nlaunch * nb * 8);
            CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(kg_lin_tlog), t.size() * 8));
            double s_pro = 0, s_issue = 0, s_comp = 0, s_store = 0, s_bar = 0, s_epi = 0, s_tot = 0;
            for (int i = 1; i < nlaunch; ++i)
                for (int b = 0; b < nb; ++b) {
                    const unsigned long long* s = &t[((size_t)i * nb + b) * 8];
                    s_pro += (double)(s[1] - s[0]); s_issue += (double)s[2]; s_comp += (double)s[3]; s_store += (double)s[4]; s_bar += (double)s[5];
                    s_epi += (double)(s[7] - s[6]); s_tot += (double)(s[7] - s[0]);
                }
            {   // how many blocks of one launch enter late (i.e. had to wait for a CU slot)
                unsigned long long first = ~0ull; int late = 0;
                for (int b = 0; b < nb; ++b) { const unsigned long long e0 = t[((size_t)5 * nb + b) * 8]; if (e0 < first) first = e0; }
                for (int b = 0; b < nb; ++b) if (t[((size_t)5 * nb + b) * 8] > first + 1000) ++late;
                printf("   blocks entering more than 10 us after the first: %d of %d\n", late, nb);
            }
            const double n = (double)(nlaunch - 1) * nb * 100.0;      // -> us per block
            printf("gemm %-15s %.1f us per launch, %d blocks x %d row groups; per block: total %.1f us = prologue %.1f + weight issue %.1f + LDS reads/MFMA %.1f + barrier %.1f + epilogue %.1f | loader wave: X wait + LDS store %.1f\n",
                   c.name, ms * 1e3, nb, nw_, s_tot / n, s_pro / n, s_issue / n, s_comp / n, s_bar / n, s_epi / n, s_store / n);
            CK(hipGraphExecDestroy(ge));
        }
        return 0;
    }
    if (what == "timeline") {
        // 32 back-to-back launches of a gate/up-sized MFMA pass (7 rows) in one graph; per block the
        // stamps are: entry, parameters known, X in LDS, first row group reduced, end.
        const bool oproj = argc > 2 && std::string(argv[2]) == "o";       // "o": the 4096 x 4096 projection without a norm
        const int N = oproj ? 4096 : 22016, K = 4096, nlaunch = 32, nb = oproj ? 256 : 230, nw_ = 8;
        size_t bytes = (size_t)N * K * 2;
        char* xcopies; CK(hipMalloc(&xcopies, (size_t)256 * 16 * K * 2)); CK(hipMemset(xcopies, 0, (size_t)256 * 16 * K * 2));
        for (int xmode = 0; xmode < (oproj ? 1 : 3); ++xmode) {
        (void)hipFuncSetAttribute((const void*)k_lin_tl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_lin_tl_o, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipGraphExec_t ge = capture(st, [&]() {
            size_t off = 0;
            for (int i = 0; i < nlaunch; ++i) {
                if (off + bytes > pool_bytes) off = 0;
                KgLinArgs a; memset(&a, 0, sizeof(a));
                a.W = pool + off; a.N = N; a.K = K; a.dyn = dyn + 1; a.x = xcopies; a.x_stride = K; a.head_part_stride = xmode;
                a.out = out; a.out_stride = N; a.norm_w = nw; a.eps = 1e-6f;
                a.side_bytes = 512; a.logits_stride = i;
                const size_t smem = 512 + (size_t)2 * nw_ * 256 * 4 + (size_t)7 * (K * 2 + 32);
                if (oproj) hipLaunchKernelGGL(k_lin_tl_o, dim3(nb), dim3(nw_ * 64), smem, st, a);
                else hipLaunchKernelGGL(k_lin_tl, dim3(nb), dim3(nw_ * 64), smem, st, a);
                off += bytes;
            }
        });
        float ms = time_graph(st, ge, 2) / nlaunch;
        printf("X source %d: timeline kernel %.1f us per launch\n", xmode, ms * 1e3);
        std::vector<unsigned long long> t((size_t)nlaunch * nb * 8);
        CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(kg_lin_tlog), t.size() * 8));
        double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int i = 1; i < nlaunch; ++i) {
            unsigned long long prev_end = 0, first_entry = ~0ull, last_entry = 0, first_end = ~0ull, last_end = 0;
            for (int b = 0; b < nb; ++b) { unsigned long long e = t[((size_t)(i - 1) * nb + b) * 8 + 4]; if (e > prev_end) prev_end = e; }
            double d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
            for (int b = 0; b < nb; ++b) {
                const unsigned long long* s = &t[((size_t)i * nb + b) * 8];
                if (s[0] < first_entry) first_entry = s[0];
                if (s[0] > last_entry) last_entry = s[0];
                if (s[4] < first_end) first_end = s[4];
                if (s[4] > last_end) last_end = s[4];
                d1 += (double)(s[1] - s[0]); d2 += (double)(s[2] - s[1]); d3 += (double)(s[3] - s[2]); d4 += (double)(s[4] - s[3]);
                d5 += (double)(s[5] - s[1]); d6 += (double)(s[6] - s[5]); d7 += (double)((long long)(s[7] - s[0]));
            }
            acc[0] += (double)((long long)(first_entry - prev_end)) * 0.01;    // previous kernel's last block end -> this kernel's first block entry
            acc[1] += (double)(last_entry - first_entry) * 0.01;               // entry skew over blocks
            acc[2] += d1 / nb * 0.01; acc[3] += d2 / nb * 0.01; acc[4] += d3 / nb * 0.01;
            acc[5] += (double)(last_end - first_end) * 0.01;                   // end skew over blocks
            acc[6] += d4 / nb * 0.01; acc[7] += d5 / nb * 0.01; acc[8] += d6 / nb * 0.01; acc[9] += d7 / nb * 0.01;
            ++cnt;
        }
        printf("gap prev-end -> first entry %.2f us | entry skew %.2f | entry->params %.2f | params->X in LDS %.2f | X->first row group reduced %.2f | -> end %.2f | end skew %.2f\n",
               acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[6] / cnt, acc[5] / cnt);
        {   // when do the blocks of each index residue mod 8 (= one XCD each) end, relative to the launch's first entry?
            double e8[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            double mx8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 1; i < nlaunch; ++i) {
                unsigned long long first_entry = ~0ull;
                for (int b = 0; b < nb; ++b) { const unsigned long long e0 = t[((size_t)i * nb + b) * 8]; if (e0 < first_entry) first_entry = e0; }
                for (int b = 0; b < nb; ++b) {
                    const double e = (double)(t[((size_t)i * nb + b) * 8 + 4] - first_entry) * 0.01;
                    e8[b & 7] += e; n8[b & 7]++; if (e > mx8[b & 7]) mx8[b & 7] = e;
                }
            }
            int* xo; CK(hipMalloc(&xo, 64 * sizeof(int)));
            hipLaunchKernelGGL(k_xcc, dim3(16), dim3(64), 0, st, xo);
            int xh[16]; CK(hipMemcpyAsync(xh, xo, sizeof(xh), hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            printf("   XCD of blocks 0..15 of a launch on this stream:");
            for (int b = 0; b < 16; ++b) printf(" %d", xh[b]);
            printf("\n");
            printf("   block end after first entry, by block index mod 8 (mean / max us):");
            for (int x = 0; x < 8; ++x) printf(" %d: %.1f / %.1f |", x, e8[x] / n8[x], mx8[x]);
            printf("\n");
        }
        printf("   thread 0 of a block: params -> first two weight batches requested %.2f us | -> its X share in LDS %.2f | -> block barrier passed %.2f | the block's last wave enters %.2f us after its first\n",
               acc[7] / cnt, acc[8] / cnt, acc[3] / cnt - acc[7] / cnt - acc[8] / cnt, acc[9] / cnt);
        CK(hipGraphExecDestroy(ge));
        }
        return 0;
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
