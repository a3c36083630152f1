// How does the block -> XCD placement of one kernel relate to the previous kernel's on the same stream?
// A captured graph of kernels with grids g[0..n) is replayed; every launch records the XCC id of a few of its blocks.
// Printed per launch: XCC of block 0, whether block b sits on (xcc0 + b) % 8 for all probed b, and the offset
// predicted from the previous launch as (previous xcc0 + previous grid) % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_rec(int* out, int slot, int spin) {
    int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0 && blockIdx.x < 64) out[slot * 64 + blockIdx.x] = v & 15;
    if (spin) { unsigned long long t = wall_clock64(); while (wall_clock64() - t < (unsigned)spin) { } }
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int grids[] = {256, 512, 256, 230, 256, 1, 256, 100, 333, 256, 512, 256};
    const int dims[] = {512, 256, 512, 512, 256, 256, 512, 512, 64, 512, 256, 512};
    const int n = sizeof(grids) / sizeof(int), reps = 4;
    int* out; CK(hipMalloc(&out, n * 64 * sizeof(int)));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_rec, dim3(grids[i]), dim3(dims[i]), 0, st, out, i, mode ? 300 : 0);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(out, 0xff, n * 64 * sizeof(int), st));
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            std::vector<int> h(n * 64);
            CK(hipMemcpy(h.data(), out, n * 64 * sizeof(int), hipMemcpyDeviceToHost));
            printf("mode %d replay %d:", mode, r);
            int prev0 = -1, prevg = 0;
            for (int i = 0; i < n; ++i) {
                const int x0 = h[i * 64];
                bool rr = true;
                for (int b = 0; b < 64 && b < grids[i]; ++b) if (h[i * 64 + b] != (x0 + b) % 8) rr = false;
                const int pred = prev0 < 0 ? -1 : (prev0 + prevg) % 8;
                printf("  g%d:x0=%d%s%s", grids[i], x0, rr ? "" : "(!rr)", pred < 0 ? "" : (pred == x0 ? " =pred" : " !=pred"));
                prev0 = x0; prevg = grids[i];
            }
            printf("\n");
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
