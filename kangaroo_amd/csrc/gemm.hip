// Prefill linears (gfx950): Y[m][n] = sum_k X[m][k] W[n][k] for up to 64 rows (tokens) per pass.
//
// Decode streams every weight once per token; a prompt of T0 tokens run through the M<=16
// decode kernels streams them T0/16 times.  Here a pass carries 64 rows, so the weights are
// read T0/64 times, still far below the machine balance (64 flop per weight byte pair), i.e.
// HBM-bound like everything else on this path.
//
// Same packed weight tiles as the decode kernels.  A wave owns one 16-row group of W and
// walks all of K: per k-batch of 8 tiles it holds the 8 A fragments in registers (two
// batches in flight) and multiplies them with four 16-row column blocks of X^T, which the
// block stages in LDS batch by batch (64 rows x 256 k, double-buffered; RMSNorm's gamma is
// applied while staging and the row factor afterwards, as in the decode kernels).  The
// accumulators go through the decode kernels' own epilogue code, so rounding is identical.
#include "kg_linear_body.h"

#define GEMM_NW 2                   // waves (= weight row groups) per block
#define GEMM_XROW (8 * 64 + 16)     // bytes of one X row of a k-batch in LDS (+ pad)
#define GEMM_STAGE (KG_GEMM_ROWS * 32 / (GEMM_NW * 64))     // 16-byte pieces of a batch each thread stages

template <int DT, int PRO, int EPI>
__global__ void __launch_bounds__(GEMM_NW * 64) kg_gemm_kernel(const KgLinArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int pos0 = dyn->pos[0];
    const int M = dyn->n_rows;
    if (dyn->active == 0) return;
    const int G = (a.N + 15) >> 4, KT = a.K >> 5, NB = (KT + 7) >> 3;
    const int rg = blockIdx.x * GEMM_NW + wave;        // >= G in the last block: loads out of range, nothing stored
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs0 = smem;
    char* xs1 = smem + KG_GEMM_ROWS * GEMM_XROW;
    float* ssum = (float*)(smem + 2 * KG_GEMM_ROWS * GEMM_XROW);       // [KG_GEMM_ROWS] sum of squares per row

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)((size_t)G * KT * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(((size_t)(M - 1) * a.x_stride + a.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(PRO == PRO_RMSNORM ? a.norm_w : a.x), 0, a.K * 2, 0x00020000);

    auto issue_w = [&](kg_u4 (&wv)[8], int b) {
        const unsigned base = (unsigned)rg * (unsigned)KT * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = (rg < G) && (8 * b + u < KT);
            wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, ok ? base + (unsigned)(8 * b + u) * 1024u : 0xFFFFFFFFu, 0, 2);
        }
    };
    // staging: piece q = tid + 128 j covers row q >> 5, 16-byte column q & 31 of the batch
    const int scol = tid & 31, srow0 = tid >> 5;
    auto stage_load = [&](kg_u4 (&sx)[GEMM_STAGE], kg_u4& gw, int b) {
        const bool kok = (8 * b + (scol >> 2)) < KT;
#pragma unroll
        for (int j = 0; j < GEMM_STAGE; ++j) {
            const int row = srow0 + (GEMM_NW * 2) * j;
            const bool ok = kok && row < M;
            sx[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(((size_t)row * a.x_stride + 256 * b + 8 * scol) * 2) : 0xFFFFFFFFu, 0, 0);
        }
        if (PRO == PRO_RMSNORM) gw = __builtin_amdgcn_raw_buffer_load_b128(nrsrc, kok ? (unsigned)((256 * b + 8 * scol) * 2) : 0xFFFFFFFFu, 0, 0);
    };
    float ss[GEMM_STAGE];
#pragma unroll
    for (int j = 0; j < GEMM_STAGE; ++j) ss[j] = 0.f;
    auto stage_store = [&](const kg_u4 (&sx)[GEMM_STAGE], const kg_u4& gw, char* xs) {
        const kg_us8 gg = __builtin_bit_cast(kg_us8, gw);
#pragma unroll
        for (int j = 0; j < GEMM_STAGE; ++j) {
            const int row = srow0 + (GEMM_NW * 2) * j;
            kg_u4 v = sx[j];
            if (PRO == PRO_RMSNORM) {
                kg_us8 h = __builtin_bit_cast(kg_us8, v);
                kg_us8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = T::to_f(h[e]);
                    ss[j] += f * f;
                    o[e] = T::from_f(T::to_f(gg[e]) * f);
                }
                v = __builtin_bit_cast(kg_u4, o);
            }
            *(kg_u4*)(xs + row * GEMM_XROW + scol * 16) = v;
        }
    };

    kg_f4 acc[KG_GEMM_ROWS / 16];
#pragma unroll
    for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) acc[cb] = kg_f4{0.f, 0.f, 0.f, 0.f};
    const int ncb = (M + 15) >> 4;                       // column blocks in use (uniform)
    auto compute = [&](const kg_u4 (&wv)[8], const char* xs) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) {
                if (cb < ncb) {
                    const kg_u4 x = *(const kg_u4*)(xs + (16 * cb + r) * GEMM_XROW + u * 64 + g * 16);
                    acc[cb] = T::mfma(wv[u], x, acc[cb]);
                }
            }
        }
    };

    kg_u4 wA[8], wB[8], sx[GEMM_STAGE], gw = {0u, 0u, 0u, 0u};
    stage_load(sx, gw, 0);
    issue_w(wA, 0);
    stage_store(sx, gw, xs0);
    __syncthreads();
    // batch b: weights in (wc), X in xs[b & 1]; the next batch's weights and X pieces are
    // requested first, its X is written to the other buffer after the MFMAs of this batch
    auto step = [&](kg_u4 (&wc)[8], kg_u4 (&wn)[8], int b) {
        stage_load(sx, gw, b + 1);
        issue_w(wn, b + 1);
        compute(wc, (b & 1) ? xs1 : xs0);
        stage_store(sx, gw, (b & 1) ? xs0 : xs1);
        __syncthreads();
    };
    for (int b = 0; b < NB; b += 2) {
        step(wA, wB, b);
        step(wB, wA, b + 1);         // past the last batch: zeros in, zeros added
    }

    // ---- row factors of the deferred RMSNorm ----
    if (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int j = 0; j < GEMM_STAGE; ++j) {
            float t = ss[j];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) t += __shfl_xor(t, o, 64);
            if ((lane & 31) == 0) ssum[srow0 + (GEMM_NW * 2) * j] = t;
        }
    }
    // ---- epilogue inputs through LDS (the X buffers are free now), laid out as the decode epilogues expect ----
    char* side = (EPI == EPI_RESID) ? xs0 + (size_t)wave * KG_GEMM_ROWS * 32 : xs0;
    if (EPI == EPI_RESID) {
        const unsigned short* res = (const unsigned short*)a.resid;
        for (int e = lane; e < M * 16; e += 64) {
            const int m = e >> 4, n = rg * 16 + (e & 15);
            ((unsigned short*)side)[e] = (n < a.N) ? res[(size_t)m * a.resid_stride + n] : (unsigned short)0;
        }
    } else if (EPI == EPI_ROPE_KV) {
        const int half = a.head_dim >> 1;
        for (int i = tid; i < M * a.head_dim; i += GEMM_NW * 64) {
            const int m = i / a.head_dim, jj = i - m * a.head_dim;
            const size_t p = (size_t)(pos0 + m) * half;
            ((float*)side)[i] = (jj < half) ? a.rope_cos[p + jj] : a.rope_sin[p + jj - half];
        }
    }
    __syncthreads();
    if (rg >= G) return;
    HeadRun run; run.mx = -INFINITY; run.sum = 0.f; run.idx = 0;
#pragma unroll
    for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) {
        const int m = 16 * cb + r;
        if (m < M) {
            const float rstd = (PRO == PRO_RMSNORM) ? rsqrtf(ssum[m] / (float)a.K + a.eps) : 1.f;
            // lane holds D[weight row 4g + i][column m], i = 0..3: two adjacent-row pairs
            epilogue_pair<DT, EPI, false>(a, rg * 16 + 4 * g, m, pos0 + m, acc[cb][0] * rstd, acc[cb][1] * rstd, run, side, 0, M);
            epilogue_pair<DT, EPI, false>(a, rg * 16 + 4 * g + 2, m, pos0 + m, acc[cb][2] * rstd, acc[cb][3] * rstd, run, side, 0, M);
        }
    }
}

static size_t gemm_smem() { return (size_t)2 * KG_GEMM_ROWS * GEMM_XROW + KG_GEMM_ROWS * sizeof(float); }

template <int DT, int PRO>
static int launch_epi(const KgLinArgs& a, int epi, dim3 grid, hipStream_t st) {
    const dim3 block(GEMM_NW * 64);
    const size_t smem = gemm_smem();
    switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((kg_gemm_kernel<DT, PRO, EPI_STORE>), grid, block, smem, st, a); break;
    case EPI_RESID: hipLaunchKernelGGL((kg_gemm_kernel<DT, PRO, EPI_RESID>), grid, block, smem, st, a); break;
    case EPI_ROPE_KV: hipLaunchKernelGGL((kg_gemm_kernel<DT, PRO, EPI_ROPE_KV>), grid, block, smem, st, a); break;
    case EPI_SILU_MUL: hipLaunchKernelGGL((kg_gemm_kernel<DT, PRO, EPI_SILU_MUL>), grid, block, smem, st, a); break;
    default: return -1;
    }
    return 0;
}

// Rows and positions come from a.dyn: n_rows <= KG_GEMM_ROWS rows at positions pos[0], pos[0]+1, ...
int kg_launch_gemm(const KgLinArgs& a, int dtype, int pro, int epi, hipStream_t st) {
    if (a.K % 32 != 0 || a.K <= 0 || a.N <= 0) return -1;
    if (epi == EPI_ROPE_KV && (size_t)KG_GEMM_ROWS * a.head_dim * 4 > (size_t)KG_GEMM_ROWS * GEMM_XROW) return -1;
    const int G = (a.N + 15) / 16;
    const dim3 grid((G + GEMM_NW - 1) / GEMM_NW);
    int rc;
    if (dtype == KG_F16) rc = (pro == PRO_NONE) ? launch_epi<KG_F16, PRO_NONE>(a, epi, grid, st) : launch_epi<KG_F16, PRO_RMSNORM>(a, epi, grid, st);
    else rc = (pro == PRO_NONE) ? launch_epi<KG_BF16, PRO_NONE>(a, epi, grid, st) : launch_epi<KG_BF16, PRO_RMSNORM>(a, epi, grid, st);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int DT, int PRO> static void set_attr_all() {
    const int lim = (int)gemm_smem();
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, PRO, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, PRO, EPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, PRO, EPI_ROPE_KV>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, PRO, EPI_SILU_MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
}
void kg_gemm_init() {
    set_attr_all<KG_F16, PRO_NONE>(); set_attr_all<KG_F16, PRO_RMSNORM>();
    set_attr_all<KG_BF16, PRO_NONE>(); set_attr_all<KG_BF16, PRO_RMSNORM>();
    (void)hipGetLastError();
}
