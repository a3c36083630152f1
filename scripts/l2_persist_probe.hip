// Does data loaded by one kernel stay in an XCD's L2 for the NEXT kernel on the stream?
// Kernel A: the blocks on XCD x read region x (bytes per XCD on the command line).  Kernel B, launched right
// after: its blocks read (a) the region of their own XCD (L2-warm if L2 contents survive the boundary), or
// (b) a region nobody touched (cold).  B's duration is timed with HIP events around many A/B pairs and with
// in-kernel clock stamps.  Build: hipcc --offload-arch=gfx950 -O3 scripts/l2_persist_probe.hip -o scripts/l2_persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

// blocks of one XCD share the region: block takes slices by a per-XCD ticket counter
__global__ void __launch_bounds__(512) k_read(const u4* base, size_t per_xcd_16, size_t region_off_16, unsigned* tickets, unsigned* sink,
                                              unsigned long long* stamps, int nt, int predict, int* x0_dev) {
    __shared__ unsigned s_t;
    // predict: take the XCD from the block index and the offset an earlier launch observed, as a kernel
    // that cannot read the hardware id ahead of time (a prefetching predecessor) would have to
    if (!predict && blockIdx.x == 0 && threadIdx.x == 0) x0_dev[0] = xcc_id();
    const int x = predict ? ((x0_dev[0] + (int)blockIdx.x) & 7) : xcc_id();
    const u4* reg = base + region_off_16 + (size_t)x * per_xcd_16;
    const size_t slice = 512 * 8;                      // 16-byte units per ticket: 64 KiB
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(tickets + x * 32, 1u);
        __syncthreads();
        const size_t s0 = (size_t)s_t * slice;
        __syncthreads();
        if (s0 >= per_xcd_16) break;
        u4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (nt ? __builtin_nontemporal_load(reg + s0 + i * 512 + threadIdx.x) : reg[s0 + i * 512 + threadIdx.x]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i].x ^ v[i].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ void k_zero(unsigned* p, int n) { if (threadIdx.x < n) p[threadIdx.x] = 0; }

int main(int argc, char** argv) {
    const size_t per_xcd = (argc > 1 ? atol(argv[1]) : 2048) * 1024;      // KiB per XCD
    const int nb = 256;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* pool; const size_t pool_bytes = (size_t)2 << 30; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
    unsigned *tick, *sink; CK(hipMalloc(&tick, 4096)); CK(hipMalloc(&sink, 256));
    unsigned long long* stamps; CK(hipMalloc(&stamps, nb * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t per16 = per_xcd / 16;
    int* x0_dev; CK(hipMalloc(&x0_dev, 256)); CK(hipMemset(x0_dev, 0, 256));
    const int use_graph = argc > 2 ? atoi(argv[2]) : 0, predict = argc > 3 ? atoi(argv[3]) : 0;
    // the stream between A and B (mode 2): MiB per XCD, plain or non-temporal loads
    const size_t mid_mib = argc > 4 ? atol(argv[4]) : 8; const int mid_nt = argc > 5 ? atoi(argv[5]) : 0;
    printf("graph %d predict %d, stream between A and B: %zu MiB per XCD, %s loads\n", use_graph, predict, mid_mib, mid_nt ? "nt" : "plain");
    for (int nt = 0; nt < 2; ++nt)
    for (int mode = 0; mode < 3; ++mode) {         // 0: B reads what A read; 1: B reads an untouched region; 2: a stream between A and B
        float tot = 0; double kspan = 0; const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            const size_t a_off = ((size_t)(r % 16) * 8 * per_xcd * 2) / 16;          // a fresh region per repetition (beyond the 256 MiB cache after 16 reps it recycles; fine for L2)
            const size_t b_off = mode == 1 ? a_off + (8 * per16) : a_off;
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            if (use_graph) CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, st, tick, 256);
            hipLaunchKernelGGL(k_read, dim3(nb), dim3(512), 0, st, (const u4*)pool, per16, a_off, tick, sink, stamps, 0, 0, x0_dev);
            if (mode == 2) {
                hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, st, tick, 256);
                hipLaunchKernelGGL(k_read, dim3(nb), dim3(512), 0, st, (const u4*)(pool + ((size_t)1 << 30)), (mid_mib << 20) / 16, 0, tick, sink, stamps, mid_nt, 0, x0_dev);
            }
            hipLaunchKernelGGL(k_zero, dim3(1), dim3(256), 0, st, tick, 256);
            if (!use_graph) CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_read, dim3(nb), dim3(512), 0, st, (const u4*)pool, per16, b_off, tick, sink, stamps, nt, predict, x0_dev);
            if (use_graph) {
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            if (use_graph) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(2 * nb);
            CK(hipMemcpy(h.data(), stamps, nb * 16, hipMemcpyDeviceToHost));
            unsigned long long lo = ~0ull, hi = 0;
            for (int b = 0; b < nb; ++b) { if (h[2 * b] < lo) lo = h[2 * b]; if (h[2 * b + 1] > hi) hi = h[2 * b + 1]; }
            if (r >= 2) { tot += ms; kspan += (double)(hi - lo) * 0.01; }
        }
        const char* names[3] = {"B re-reads A's regions (same XCD)", "B reads untouched regions", "B re-reads A's regions after the stream"};
        printf("%5zu KiB per XCD, %s loads, %-46s: events %.2f us, first entry -> last exit %.2f us  (%.0f GB/s)\n", per_xcd >> 10, nt ? "nt" : "plain", names[mode],
               tot / reps * 1e3, kspan / reps, 8.0 * per_xcd / (kspan / reps * 1e-6) / 1e9);
    }
    return 0;
}
