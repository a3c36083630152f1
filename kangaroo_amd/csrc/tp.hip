// Exchange of a tensor-parallel layer's partial sums (gfx950, peers over xGMI).
//
// o_proj and down_proj of a sharded layer leave fp32 partial sums [rows][d] in the producing rank's slot of its own
// exchange buffer.  One launch of kg_tp_reduce_kernel per projection then does the whole all-reduce in one hop:
//   push   every block copies its piece of the local slot(s) into the same slot of every peer's buffer (plain stores
//          through the peer mapping), fences at system scope and raises flag[parity][rank][block] there to the
//          exchange's sequence number
//   wait   one thread per remote rank spins on that rank's flag in the LOCAL buffer
//   sum    slots 0 .. tp-1 added in rank order (the same order on every rank: identical bits everywhere), rounded,
//          added to the residual rows, rounded again: round16(h + round16(y)) as the residual epilogue computes it
// Blocks are independent (a block waits only for the peers' blocks of the same index), so no grid-wide step exists.
// The buffer has two halves used by alternating exchanges: a peer can be at most one exchange ahead (it cannot
// finish exchange n + 1 without this rank's push of n + 1, which follows this rank's sum of n), so the half being
// summed is never the one being written.  Flags only grow: sequence = passes completed * exchanges per pass + index
// + 1, the pass counter being advanced by a kernel of its own between passes (kg_tp_pass_end_kernel), so every block
// of a launch reads the same value.  A pass that is switched off (dyn->active == 0) neither waits nor counts; the
// ranks agree on that because the loop's decisions are replicated.
// The exchange buffer is uncached device memory (every access goes to memory, so a peer's stores are seen without
// cache maintenance); remote slots and flags are read with system-scope loads all the same.
#include "kg_common.h"
#include "kg_kernels.h"

#define KG_SYS __HIP_MEMORY_SCOPE_SYSTEM
#define KG_TP_SPIN_LIMIT (1u << 25)      // x ~0.3 us: about ten seconds

__device__ __forceinline__ unsigned* tp_flags(float* xbuf, int tp, int d) { return (unsigned*)(xbuf + kg_tp_data_floats(tp, d)); }
__device__ __forceinline__ unsigned* tp_ctrl(float* xbuf, int tp, int d) { return tp_flags(xbuf, tp, d) + 2 * (size_t)tp * KG_TP_FLAG_BLOCKS; }

template <int DT>
__global__ void __launch_bounds__(256) kg_tp_reduce_kernel(const KgTpArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int m = blockIdx.y;
    if (m >= dyn->n_rows) return;
    const int col = 4 * (blockIdx.x * 256 + threadIdx.x);
    const bool ok = col < a.d;                              // d is a multiple of 32
    const int par = a.e & 1, tp = a.tp;
    const size_t slot = kg_tp_slot_floats(a.d), off = (size_t)m * a.d + col;
    const unsigned all = (1u << tp) - 1u, remote = all & ~a.local_mask;
    if (remote) {
        unsigned* ctrl = tp_ctrl(a.xbuf, tp, a.d);
        const unsigned seq = __hip_atomic_load(ctrl, __ATOMIC_RELAXED, KG_SYS) * (unsigned)a.exch_per_pass + (unsigned)a.e + 1u;
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        for (int r = 0; r < tp; ++r) {
            if (!((a.local_mask >> r) & 1u)) continue;
            const kg_f4 v = ok ? *(const kg_f4*)(a.xbuf + (size_t)(par * tp + r) * slot + off) : kg_f4{0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < tp; ++p)
                if (((remote >> p) & 1u) && ok) *(kg_f4*)(a.peer[p] + (size_t)(par * tp + r) * slot + off) = v;
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int r = 0; r < tp; ++r) {
                if (!((a.local_mask >> r) & 1u)) continue;
                for (int p = 0; p < tp; ++p)
                    if ((remote >> p) & 1u)
                        __hip_atomic_store(tp_flags(a.peer[p], tp, a.d) + (size_t)(par * tp + r) * KG_TP_FLAG_BLOCKS + blk, seq, __ATOMIC_RELEASE, KG_SYS);
            }
        }
        if ((int)threadIdx.x < tp && ((remote >> threadIdx.x) & 1u)) {
            unsigned* f = tp_flags(a.xbuf, tp, a.d) + (size_t)(par * tp + threadIdx.x) * KG_TP_FLAG_BLOCKS + blk;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, KG_SYS) - seq) < 0) {
                __builtin_amdgcn_s_sleep(10);
                if ((++spins & 255u) == 0 && (spins > KG_TP_SPIN_LIMIT || __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, KG_SYS) != 0u)) {
                    __hip_atomic_store(ctrl + 1, 1u, __ATOMIC_RELAXED, KG_SYS);      // give up: every later wait falls through, the host reports it
                    break;
                }
            }
        }
        __syncthreads();
    }
    if (!ok) return;
    kg_f4 acc;
    {
        const float* p0 = a.xbuf + (size_t)(par * tp) * slot + off;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __hip_atomic_load(p0 + i, __ATOMIC_RELAXED, KG_SYS);
    }
    for (int r = 1; r < tp; ++r) {
        const float* pr = a.xbuf + (size_t)(par * tp + r) * slot + off;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __hip_atomic_load(pr + i, __ATOMIC_RELAXED, KG_SYS);
    }
    const unsigned short* res = (const unsigned short*)a.resid + (size_t)m * a.resid_stride + col;
    unsigned short* out = (unsigned short*)a.out + (size_t)m * a.out_stride + col;
    const unsigned long long r4 = *(const unsigned long long*)res;
    unsigned long long o4 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned short o = T::from_f(T::to_f((unsigned short)(r4 >> (16 * i))) + kg_round<DT>(acc[i]));
        o4 |= (unsigned long long)o << (16 * i);
    }
    *(unsigned long long*)out = o4;
}

int kg_launch_tp_reduce(const KgTpArgs& a, int max_rows, int dtype, hipStream_t st) {
    if (a.tp < 2 || a.tp > KG_TP_MAX || a.d % 32 || max_rows < 1 || max_rows > KG_TP_ROWS_CAP || !a.xbuf || !a.resid || !a.out) return -1;
    if ((a.resid_stride | a.out_stride) & 3) return -1;
    const dim3 grid((a.d + 1023) / 1024, max_rows);
    if ((size_t)grid.x * grid.y > KG_TP_FLAG_BLOCKS) return -1;
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_tp_reduce_kernel<KG_F16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(kg_tp_reduce_kernel<KG_BF16>, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

__global__ void kg_tp_pass_end_kernel(const KgDyn* dyn, float* xbuf, int tp, int d) {
    if (dyn->active == 0) return;
    unsigned* ctrl = tp_ctrl(xbuf, tp, d);
    __hip_atomic_store(ctrl, __hip_atomic_load(ctrl, __ATOMIC_RELAXED, KG_SYS) + 1u, __ATOMIC_RELAXED, KG_SYS);
}
int kg_launch_tp_pass_end(const KgDyn* dyn, float* xbuf, int tp, int d, hipStream_t st) {
    hipLaunchKernelGGL(kg_tp_pass_end_kernel, dim3(1), dim3(1), 0, st, dyn, xbuf, tp, d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
