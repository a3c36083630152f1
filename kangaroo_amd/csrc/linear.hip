// Stand-alone launches of the weight-streaming linear pass (see kg_linear_body.h).
#include "kg_linear_body.h"

template <int DT, int MODE, int PRO, int EPI>
__global__ void __launch_bounds__(512) kg_linear_kernel(const KgLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    kg_linear_body<DT, MODE, PRO, EPI, false>(a, smem, nullptr, blockDim.x >> 6);
}

// ---------------------------------------------------------------------------------
template <int DT, int MODE, int PRO>
static int launch_epi(const KgLinArgs& a, int epi, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
    switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO, EPI_STORE>), grid, block, smem, st, a); break;
    case EPI_RESID: hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO, EPI_RESID>), grid, block, smem, st, a); break;
    case EPI_ROPE_KV: hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO, EPI_ROPE_KV>), grid, block, smem, st, a); break;
    case EPI_SILU_MUL: hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO, EPI_SILU_MUL>), grid, block, smem, st, a); break;
    case EPI_HEAD: hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO, EPI_HEAD>), grid, block, smem, st, a); break;
    case EPI_STORE_F32:
        if (PRO != PRO_NONE) return -1;          // only the row-split projections of a tensor-parallel layer produce partial sums
        hipLaunchKernelGGL((kg_linear_kernel<DT, MODE, PRO_NONE, EPI_STORE_F32>), grid, block, smem, st, a); break;
    default: return -1;
    }
    return 0;
}

template <int DT>
static int launch_dt(const KgLinArgs& a, int mode, int pro, int epi, dim3 grid, dim3 block, size_t smem, hipStream_t st) {
    if (mode == MODE_VEC) {
        return pro == PRO_NONE ? launch_epi<DT, MODE_VEC, PRO_NONE>(a, epi, grid, block, smem, st)
                               : launch_epi<DT, MODE_VEC, PRO_RMSNORM>(a, epi, grid, block, smem, st);
    } else if (mode == MODE_MFMA) {
        return pro == PRO_NONE ? launch_epi<DT, MODE_MFMA, PRO_NONE>(a, epi, grid, block, smem, st)
                               : launch_epi<DT, MODE_MFMA, PRO_RMSNORM>(a, epi, grid, block, smem, st);
    } else {
        if (pro != PRO_NONE) return -1;
        return launch_epi<DT, MODE_MFMA_GX, PRO_NONE>(a, epi, grid, block, smem, st);
    }
}

static const size_t KG_LDS_LIMIT = 160 * 1024;

size_t kg_linear_smem(int mfma, int max_rows, int K, int nwaves) {
    size_t part = (size_t)2 * nwaves * (mfma ? 256 : 16) * 4;
    size_t x = (size_t)(mfma ? max_rows : 1) * ((size_t)K * 2 + 32);
    return part + x;
}

// max_rows: upper bound of dyn->n_rows for this launch (sizes LDS); rows == 1 selects VEC.
int kg_launch_linear(const KgLinArgs& a_in, int dtype, int max_rows, int pro, int epi, int nblocks, int nwaves, hipStream_t st) {
    KgLinArgs a = a_in;
    if (a.K % 32 != 0 || a.K <= 0 || a.N <= 0) return -1;
    if (max_rows < 1 || max_rows > KG_MAX_ROWS) return -1;
    if (nwaves < 1 || nwaves > 8) return -1;
    int mode = (max_rows == 1) ? MODE_VEC : MODE_MFMA;
    if (mode == MODE_MFMA && pro == PRO_RMSNORM && a.K > 8192) return -3;      // a lane holds at most 16 chunks of a row to normalise
    const int G = (a.N + 15) / 16;
    if (nblocks > G) nblocks = G;
    if (nblocks < 1) nblocks = 1;
    size_t side = 0;
    if (epi == EPI_RESID) side = (size_t)((G + nblocks - 1) / nblocks) * max_rows * 16 * 2;
    if (epi == EPI_ROPE_KV) side = (size_t)max_rows * a.head_dim * 4;
    side = (side + 255) / 256 * 256 + 512;          // + [rows][nwaves] floats for the deferred RMSNorm reduction
    a.side_bytes = (int)side;
    size_t smem = side + kg_linear_smem(mode != MODE_VEC, max_rows, a.K, nwaves);
    if (mode == MODE_MFMA) {
        // X rows that would leave less than two blocks per CU (or not fit at all) are read
        // as B fragments straight from global memory (L2) when no prologue has to rewrite them
        if (pro == PRO_NONE && smem > 80 * 1024) {
            mode = MODE_MFMA_GX;
            a.gx_slot_rows = max_rows <= 8 ? 8 : 16;
            smem = side + (size_t)2 * nwaves * 256 * 4 + (size_t)nwaves * 2 * a.gx_slot_rows * (8 * 64 + 16);
            if (smem > KG_LDS_LIMIT) return -3;
        }
        else if (smem > KG_LDS_LIMIT) return -3;
    }
    dim3 grid(nblocks), block(nwaves * 64);
    int rc = (dtype == KG_F16) ? launch_dt<KG_F16>(a, mode, pro, epi, grid, block, smem, st)
                               : launch_dt<KG_BF16>(a, mode, pro, epi, grid, block, smem, st);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// One-time opt-in for > 64 KiB of dynamic LDS on the M-row kernels.
template <int DT, int MODE, int PRO, int EPI> static void set_attr() {
    (void)hipFuncSetAttribute((const void*)kg_linear_kernel<DT, MODE, PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KG_LDS_LIMIT);
}
template <int DT, int MODE, int PRO> static void set_attr_epi() {
    set_attr<DT, MODE, PRO, EPI_STORE>(); set_attr<DT, MODE, PRO, EPI_RESID>(); set_attr<DT, MODE, PRO, EPI_ROPE_KV>();
    set_attr<DT, MODE, PRO, EPI_SILU_MUL>(); set_attr<DT, MODE, PRO, EPI_HEAD>();
    if (PRO == PRO_NONE) set_attr<DT, MODE, PRO_NONE, EPI_STORE_F32>();
}
void kg_linear_init() {
    set_attr_epi<KG_F16, MODE_MFMA, PRO_NONE>(); set_attr_epi<KG_F16, MODE_MFMA, PRO_RMSNORM>();
    set_attr_epi<KG_BF16, MODE_MFMA, PRO_NONE>(); set_attr_epi<KG_BF16, MODE_MFMA, PRO_RMSNORM>();
    set_attr_epi<KG_F16, MODE_MFMA_GX, PRO_NONE>(); set_attr_epi<KG_BF16, MODE_MFMA_GX, PRO_NONE>();
    set_attr_epi<KG_F16, MODE_VEC, PRO_NONE>(); set_attr_epi<KG_F16, MODE_VEC, PRO_RMSNORM>();
    set_attr_epi<KG_BF16, MODE_VEC, PRO_NONE>(); set_attr_epi<KG_BF16, MODE_VEC, PRO_RMSNORM>();
}
