// Persistent pass kernel (gfx950): a whole multi-layer pass (e.g. the verify pass over the
// deep layers + lm_head) as ONE launch of one block per CU.
//
// Why: at bsz=1 every linear op is a 5-30 us weight stream, and a kernel boundary between
// two of them costs the drain of one stream plus the ramp of the next (launch, parameter
// fetch, first HBM round trip), during which HBM idles.  Here the blocks stay resident and
// walk a list of phases; between phases they meet at a grid barrier, and each block
// requests the first weight tiles of the NEXT phase before it waits there, so the weight
// stream runs across the barrier.  Nothing is flushed or invalidated: everything one phase
// hands to the next (hidden rows, q / new K,V rows, attention output, MLP activations) is
// written and read with device-coherent accesses, weights and tables stay ordinary loads.
#include "kg_linear_body.h"
#include "kg_attn_body.h"

template <int DT, int HD>
__device__ __forceinline__ void attn_phase(const KgAttnArgs& a, char* smem, KgGridBar* gb) {
    gb->wait();
    const KgDyn* dyn = a.dyn;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // keep this phase's per-thread constants out of the kernel prologue
    const int team = tid >> 8, ttid = tid & 255;
    const int pos_c = dyn->pos[tid & 15];
    const int M = dyn->n_rows;
    const int len = dyn->pos[0] + M;
    int chunk = (a.t_max + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + 31) / 32 * 32;
    const int n_act = (len + chunk - 1) / chunk;
    const int items = a.H * n_act, teams = (int)gridDim.x * 2;
    KgAttnTeamLds<HD>* lds = (KgAttnTeamLds<HD>*)smem + team;
    for (int base = 0; base < items; base += teams) {
        const int item = base + (int)blockIdx.x * 2 + team;
        const bool valid = item < items;
        // heads vary fastest: the splits of one head land on different blocks
        const int h = valid ? item % a.H : 0, split = valid ? item / a.H : 0;
        kg_attn_mfma_item<DT, HD, true>(a, valid, h, split, n_act, chunk, M, len, pos_c, ttid, lds[0]);
    }
    gb->arrive();
}

template <int DT>
__global__ void __launch_bounds__(512) kg_persist_kernel(const KgPhase* phases, int n_phases, unsigned* bar_ctr, unsigned* abort_flag, unsigned long long* tlog) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    KgGridBar gb;
    gb.init(bar_ctr, abort_flag, tlog);
    if (phases[0].lin.dyn->active == 0) return;            // same answer in every block: nobody waits for anybody
    for (int i = 0; i < n_phases; ++i) {
        const int kind = phases[i].kind, nw = phases[i].nwaves;
        if (tlog && blockIdx.x == 0 && threadIdx.x == 0) tlog[2 * i] = wall_clock64();
        if (kind == KG_PH_ATTN) {
            const KgAttnArgs a = phases[i].attn;
            switch (a.hd) {
            case 32: attn_phase<DT, 32>(a, smem, &gb); break;
            case 64: attn_phase<DT, 64>(a, smem, &gb); break;
            default: attn_phase<DT, 128>(a, smem, &gb); break;
            }
            continue;
        }
        const KgLinArgs a = phases[i].lin;
        switch (kind) {
        case KG_PH_QKV: kg_linear_body<DT, MODE_MFMA, PRO_RMSNORM, EPI_ROPE_KV, true>(a, smem, &gb, nw); break;
        case KG_PH_O: kg_linear_body<DT, MODE_MFMA, PRO_NONE, EPI_RESID, true>(a, smem, &gb, nw); break;
        case KG_PH_GATEUP: kg_linear_body<DT, MODE_MFMA, PRO_RMSNORM, EPI_SILU_MUL, true>(a, smem, &gb, nw); break;
        case KG_PH_DOWN: kg_linear_body<DT, MODE_MFMA, PRO_NONE, EPI_RESID, true>(a, smem, &gb, nw); break;
        case KG_PH_DOWN_GX: kg_linear_body<DT, MODE_MFMA_GX, PRO_NONE, EPI_RESID, true>(a, smem, &gb, nw); break;
        default: kg_linear_body<DT, MODE_MFMA, PRO_RMSNORM, EPI_HEAD, true>(a, smem, &gb, nw); break;
        }
    }
    if (tlog && blockIdx.x == 0 && threadIdx.x == 0) tlog[2 * n_phases] = wall_clock64();
}

static const size_t KG_PERSIST_LDS_LIMIT = 160 * 1024;

// Fills the launcher-owned fields of a linear phase (mode, LDS layout) and returns the LDS
// bytes it needs, or 0 when the phase cannot run in a persistent pass.
size_t kg_persist_plan_linear(KgPhase& p, int max_rows, int nblocks) {
    KgLinArgs& a = p.lin;
    if (a.K % 32 != 0 || a.K <= 0 || a.N <= 0 || max_rows < 2 || max_rows > KG_MAX_ROWS || p.nwaves < 1 || p.nwaves > 8) return 0;
    const int G = (a.N + 15) / 16;
    if ((p.kind == KG_PH_QKV || p.kind == KG_PH_GATEUP || p.kind == KG_PH_HEAD) && a.K > 8192) return 0;
    size_t side = 0;
    if (p.kind == KG_PH_O || p.kind == KG_PH_DOWN || p.kind == KG_PH_DOWN_GX) side = (size_t)((G + nblocks - 1) / nblocks) * max_rows * 16 * 2;
    if (p.kind == KG_PH_QKV) side = (size_t)max_rows * a.head_dim * 4;
    side = (side + 255) / 256 * 256 + 512;
    a.side_bytes = (int)side;
    size_t smem = side + kg_linear_smem(1, max_rows, a.K, p.nwaves);
    if (p.kind == KG_PH_DOWN && smem > 80 * 1024) p.kind = KG_PH_DOWN_GX;
    if (p.kind == KG_PH_DOWN_GX) {
        a.gx_slot_rows = max_rows <= 8 ? 8 : 16;
        smem = side + (size_t)2 * p.nwaves * 256 * 4 + (size_t)p.nwaves * 2 * a.gx_slot_rows * (8 * 64 + 16);
    }
    return smem <= KG_PERSIST_LDS_LIMIT ? smem : 0;
}

size_t kg_persist_attn_smem(int hd) {
    switch (hd) {
    case 32: return 2 * sizeof(KgAttnTeamLds<32>);
    case 64: return 2 * sizeof(KgAttnTeamLds<64>);
    case 128: return 2 * sizeof(KgAttnTeamLds<128>);
    default: return 0;
    }
}

void kg_persist_init() {
    (void)hipFuncSetAttribute((const void*)kg_persist_kernel<KG_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KG_PERSIST_LDS_LIMIT);
    (void)hipFuncSetAttribute((const void*)kg_persist_kernel<KG_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KG_PERSIST_LDS_LIMIT);
}

// Blocks of the persistent kernel that can be resident at once, or 0 if it cannot run.
int kg_persist_max_blocks(int dtype, size_t smem) {
    int per_cu = 0;
    hipError_t e = (dtype == KG_F16)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kg_persist_kernel<KG_F16>, 512, smem)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kg_persist_kernel<KG_BF16>, 512, smem);
    if (e != hipSuccess || per_cu < 1) return 0;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return ncu;     // one block per CU, whatever the occupancy allows beyond that
}

// phases: device array.  bar: 8 x 32 counters (zero) followed by the abort flag at [256].
int kg_launch_persist(const KgPhase* phases_dev, int n_phases, unsigned* bar, int dtype, int nblocks, size_t smem, unsigned long long* tlog, hipStream_t st) {
    if (!phases_dev || n_phases < 1 || !bar || nblocks < 8 || (nblocks & 7) || smem > KG_PERSIST_LDS_LIMIT) return -1;
    if (dtype == KG_F16) hipLaunchKernelGGL((kg_persist_kernel<KG_F16>), dim3(nblocks), dim3(512), smem, st, phases_dev, n_phases, bar, bar + 256, tlog);
    else hipLaunchKernelGGL((kg_persist_kernel<KG_BF16>), dim3(nblocks), dim3(512), smem, st, phases_dev, n_phases, bar, bar + 256, tlog);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
