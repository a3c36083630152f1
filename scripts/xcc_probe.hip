// Which XCD does block (x, y, z) of a 3-D grid land on?  Prints, per y/z, the XCC ids of x = 0..31,
// and whether xcc == (x + gx * y + gx * gy * z) % 8 holds for every block, for a plain launch and
// for the same launch replayed from a graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = v & 0xf;
}
int main() {
    const dim3 grid(32, 16, 2);
    const int n = grid.x * grid.y * grid.z;
    unsigned* d; hipMalloc(&d, n * sizeof(unsigned));
    std::vector<unsigned> h(n);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemsetAsync(d, 0xff, n * sizeof(unsigned), st);
        if (mode == 0) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, grid, dim3(256), 0, st, d);
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, grid, dim3(256), 0, st, d);
            hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, st); hipGraphLaunch(ge, st);
        }
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += (h[i] != (unsigned)(i % 8));
        printf("mode %d: %d of %d blocks off the id %% 8 rule\n", mode, bad, n);
        for (int y = 0; y < 3; ++y) { printf("  y=%d:", y); for (int x = 0; x < 32; ++x) printf(" %u", h[x + 32 * y]); printf("\n"); }
    }
    return 0;
}
