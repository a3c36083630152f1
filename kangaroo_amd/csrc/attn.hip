// Decode attention over the in-place KV cache (gfx950): q rows 1..16 against T cached keys.
//
// One block = one (query head, T-split, block of <= MR query rows).  K and V rows are
// streamed straight into VGPRs: a wave-level load covers 64/LPR consecutive cache rows
// (1 KiB, fully coalesced), LPR = hd/8 lanes per row.  Scores use v_dot2 + an
// LPR-lane shuffle reduction, softmax is online (running max / sum per row) in fp32,
// P.V accumulates per lane for its 8 head dims.  Waves of a block, then T-splits, are
// merged with the usual (max, sum, acc) rule.  The T-splits of one head meet without a
// second kernel: every block publishes its partial result with device-coherent stores,
// then bumps a per-head counter; the block that arrives last reads all partials back and
// writes the final rows (always summing in split order, so the result does not depend on
// which block that is).  The causal bound of row m is its own
// cache slot pos[m] (keys at slots > pos[m] are excluded), which replaces the
// reference's additive mask tensors (kangaroo/adapter.py:38-67, 327-348).
#include "kg_common.h"
#include "kg_kernels.h"

#define ATT_U 4         // K and V loads in flight per lane per batch
#define ATT_NW 4        // waves per block

#define KG_AGENT __HIP_MEMORY_SCOPE_AGENT

// Device-coherent access to the split partials: these bypass the per-XCD L2, so a block on
// another XCD reads what was written without any cache maintenance in between.
__device__ __forceinline__ void coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, KG_AGENT); }
__device__ __forceinline__ float coh_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, KG_AGENT); }

// Called by every thread of a block after its partial stores.  Returns true in all threads
// of the block that completed the head's set of n_act partials (and re-arms the counter).
__device__ __forceinline__ bool split_arrive(unsigned* counter, int n_act) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's partial stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, KG_AGENT);
        const int last = (old == (unsigned)(n_act - 1));
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, KG_AGENT);
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// out[m][h][:] = sum_s w_s o_s / sum_s w_s l_s,  w_s = exp(m_s - max_s m_s), rows m0 .. m0+mr-1
template <int DT>
__device__ __forceinline__ void split_combine(const KgAttnArgs& a, int h, int hd, int m0, int mr, int ns) {
    typedef KgT<DT> T;
    for (int i = threadIdx.x; i < mr * hd; i += blockDim.x) {
        const int m = m0 + i / hd, dmi = i % hd;
        float mx = -INFINITY;
        for (int s = 0; s < ns; ++s) mx = fmaxf(mx, coh_load(a.part_ml + (((size_t)s * a.max_rows + m) * a.H + h) * 2));
        float o = 0.f, l = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float* ml = a.part_ml + (((size_t)s * a.max_rows + m) * a.H + h) * 2;
            const float m_s = coh_load(ml);
            const float f = (m_s == -INFINITY) ? 0.f : expf(m_s - mx);
            o += f * coh_load(a.part_o + (((size_t)s * a.max_rows + m) * a.H + h) * hd + dmi);
            l += f * coh_load(ml + 1);
        }
        ((unsigned short*)a.out)[(size_t)m * a.H * hd + (size_t)h * hd + dmi] = T::from_f(o / l);
    }
}

template <int DT, int LPR, int MR>
__global__ void __launch_bounds__(ATT_NW * 64) kg_attn_kernel(const KgAttnArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int M = dyn->n_rows;
    const int m0 = blockIdx.z * MR;
    if (m0 >= M) return;
    const int mr = min(MR, M - m0);
    const int len = dyn->pos[0] + M;               // rows of a pass sit at consecutive positions
    const int h = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int hd = LPR * 8;
    const int kvh = h / (a.H / a.Hkv);
    constexpr int RPI = 64 / LPR;                  // cache rows per wave-level load
    constexpr int BR = ATT_U * RPI;                // rows per batch
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % LPR, rsub = lane / LPR;

    // Split boundaries depend only on the cache size, never on the current length: a row's
    // keys are then summed in the same order whether it is decoded alone or verified
    // together with later rows, which keeps verify and plain decoding bit-identical.
    int chunk = (a.t_max + nsplit - 1) / nsplit;
    chunk = (chunk + BR - 1) / BR * BR;
    const int n_act = (len + chunk - 1) / chunk;   // splits that hold keys; the others do nothing
    if (split >= n_act) return;
    const int tbeg = split * chunk, tend = min(len, tbeg + chunk);

    // query fragments (this lane's 8 dims of every row) and causal limits
    kg_u4 qf[MR];
    int lim[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int mm = (m < mr) ? (m0 + m) : m0;
        qf[m] = *((const kg_u4*)((const unsigned short*)a.q + (size_t)mm * a.H * hd + (size_t)h * hd) + c);
        lim[m] = (m < mr) ? dyn->pos[mm] : -1;
    }
    float mrun[MR], lsum[MR], acc[MR][8];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        mrun[m] = -INFINITY; lsum[m] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    }
    const float inv_scale = sqrtf((float)hd);
    const unsigned short* kbase = (const unsigned short*)a.k_cache + (size_t)kvh * a.t_max * hd;
    const unsigned short* vbase = (const unsigned short*)a.v_cache + (size_t)kvh * a.t_max * hd;

    for (int tb = tbeg + wave * BR; tb < tend; tb += ATT_NW * BR) {
        kg_u4 kv[ATT_U], vv[ATT_U];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int t = tb + u * RPI + rsub;
            const bool ok = t < tend;
            const size_t off = (size_t)(ok ? t : tbeg) * hd;
            kv[u] = *((const kg_u4*)(kbase + off) + c);
            vv[u] = *((const kg_u4*)(vbase + off) + c);
            if (!ok) { vv[u].x = 0; vv[u].y = 0; vv[u].z = 0; vv[u].w = 0; }
        }
        float sc[ATT_U][MR];
#pragma unroll
        for (int u = 0; u < ATT_U; ++u) {
            const int t = tb + u * RPI + rsub;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= mr) { sc[u][m] = -INFINITY; continue; }     // uniform: work follows the valid rows
                float d = 0.f;
                d = T::dot2(kv[u].x, qf[m].x, d);
                d = T::dot2(kv[u].y, qf[m].y, d);
                d = T::dot2(kv[u].z, qf[m].z, d);
                d = T::dot2(kv[u].w, qf[m].w, d);
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
                // reference op boundaries: q.k^T stored, then divided by sqrt(hd) and stored
                d = kg_round<DT>(kg_round<DT>(d) / inv_scale);
                sc[u][m] = (t < tend && t <= lim[m]) ? d : -INFINITY;
            }
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < mr) {
                float bm = sc[0][m];
#pragma unroll
                for (int u = 1; u < ATT_U; ++u) bm = fmaxf(bm, sc[u][m]);
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) bm = fmaxf(bm, __shfl_xor(bm, o, 64));
                const float mnew = fmaxf(mrun[m], bm);
                if (mnew > -INFINITY) {
                    const float scale = (mrun[m] == -INFINITY) ? 0.f : expf(mrun[m] - mnew);
                    lsum[m] *= scale;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[m][j] *= scale;
#pragma unroll
                    for (int u = 0; u < ATT_U; ++u) {
                        const float p = (sc[u][m] == -INFINITY) ? 0.f : expf(sc[u][m] - mnew);
                        lsum[m] += p;
                        kg_us8 vh = __builtin_bit_cast(kg_us8, vv[u]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m][j] += p * T::to_f(vh[j]);
                    }
                    mrun[m] = mnew;
                }
            }
        }
    }
    // ---- merge the row groups of this wave (lanes that share c) ----
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m >= mr) continue;
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            lsum[m] += __shfl_xor(lsum[m], o, 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m][j] += __shfl_xor(acc[m][j], o, 64);
        }
    }
    // ---- merge waves through LDS ----
    __shared__ float s_o[ATT_NW][MR][LPR * 8];
    __shared__ float s_ml[ATT_NW][MR][2];
    if (rsub == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_o[wave][m][c * 8 + j] = acc[m][j];
            if (c == 0) { s_ml[wave][m][0] = mrun[m]; s_ml[wave][m][1] = lsum[m]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < mr * hd; i += blockDim.x) {
        const int m = i / hd, dmi = i - m * hd;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_NW; ++w) mx = fmaxf(mx, s_ml[w][m][0]);
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_NW; ++w) {
            const float wm = s_ml[w][m][0];
            const float f = (wm == -INFINITY) ? 0.f : expf(wm - mx);
            o += f * s_o[w][m][dmi];
            l += f * s_ml[w][m][1];
        }
        const int mm = m0 + m;
        if (n_act == 1) {
            ((unsigned short*)a.out)[(size_t)mm * a.H * hd + (size_t)h * hd + dmi] = T::from_f(o / l);
        } else {
            coh_store(a.part_o + (((size_t)split * a.max_rows + mm) * a.H + h) * hd + dmi, o);
            if (dmi == 0) {
                float* ml = a.part_ml + (((size_t)split * a.max_rows + mm) * a.H + h) * 2;
                coh_store(ml, mx); coh_store(ml + 1, l);
            }
        }
    }
    if (n_act > 1 && split_arrive(a.counters + blockIdx.z * a.H + h, n_act)) split_combine<DT>(a, h, hd, m0, mr, n_act);
}

// ---------------------------------------------------------------------------------
// MFMA variant (head_dim 32 / 64 / 128, V cached transposed [Hkv][hd][t_max]).
// Per wave and 32 keys:  S^T = K q^T  as two 16-key tiles of v_mfma_f32_16x16x32 (A = K rows,
// B = q^T, so every lane owns one query column and 8 keys), softmax statistics per column
// with two cross-lane steps for ALL query rows at once, then  O += P V  with the same
// instruction (A = P straight from the score registers, B = V^T rows, 16 head dims per
// MFMA).  The key <-> MFMA-row assignment of the two score tiles is chosen so that the 8
// probabilities a lane holds are 8 consecutive keys, i.e. one 16-byte piece of a V^T row.
// Cost is independent of the number of query rows (<= 16), so the verify pass, prefill
// chunks and single-token decode share it, and a row's result does not depend on which
// other rows ride along (columns of an MFMA are independent).
template <int DT, int HD>
__global__ void __launch_bounds__(ATT_NW * 64) kg_attn_mfma_kernel(const KgAttnArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int pos_c = dyn->pos[c];
    if (dyn->active == 0) return;
    const int M = dyn->n_rows;
    const int len = __shfl(pos_c, 0, 64) + M;      // rows of a pass sit at consecutive positions
    const int h = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int kvh = h / (a.H / a.Hkv);
    constexpr int KS = HD / 32;                    // k-steps of the score MFMAs
    constexpr int DB = HD / 16;                    // 16-wide head-dim blocks of the output
    int chunk = (a.t_max + nsplit - 1) / nsplit;   // fixed split boundaries (see kg_attn_kernel)
    chunk = (chunk + 31) / 32 * 32;
    const int n_act = (len + chunk - 1) / chunk;   // splits that hold keys; the others do nothing
    if (split >= n_act) return;
    const int tbeg = split * chunk, tend = min(len, tbeg + chunk);
    const int lim = (c < M) ? pos_c : -1;          // causal bound of this lane's query column

    kg_u4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kg_u4 z = {0u, 0u, 0u, 0u};
        qf[s] = (c < M) ? *((const kg_u4*)((const unsigned short*)a.q + ((size_t)c * a.H + h) * HD + 32 * s + 8 * g)) : z;
    }
    float mrun = -INFINITY, lsum = 0.f;
    kg_f4 O[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) O[b] = kg_f4{0.f, 0.f, 0.f, 0.f};
    const float inv_scale = sqrtf((float)HD);
    const unsigned short* kbase = (const unsigned short*)a.k_cache + (size_t)kvh * a.t_max * HD;
    const unsigned short* vbase = (const unsigned short*)a.v_cache + (size_t)kvh * HD * a.t_max;
    const int krow = 8 * (c >> 2) + (c & 3);       // tile 0: MFMA row c <-> key tb + krow; tile 1: + 4

    for (int tb = tbeg + 32 * wave; tb < tend; tb += 32 * ATT_NW) {
        kg_u4 kf[2][KS], vf[DB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                kf[t][s] = *((const kg_u4*)(kbase + (size_t)(tb + krow + 4 * t) * HD + 32 * s + 8 * g));
#pragma unroll
        for (int b = 0; b < DB; ++b)
            vf[b] = *((const kg_u4*)(vbase + (size_t)(16 * b + c) * a.t_max + tb + 8 * g));
        float sc[8];
        float bm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kg_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) d = T::mfma(kf[t][s], qf[s], d);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = tb + 8 * g + 4 * t + i;
                // reference op boundaries: q.k^T stored, then divided by sqrt(hd) and stored
                const float v = kg_round<DT>(kg_round<DT>(d[i]) / inv_scale);
                sc[4 * t + i] = (key < tend && key <= lim) ? v : -INFINITY;
                bm = fmaxf(bm, sc[4 * t + i]);
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mnew = fmaxf(mrun, bm);
        const float scale = (mrun == -INFINITY) ? 0.f : expf(mrun - mnew);     // mnew == -inf only if mrun is too
        unsigned short ph[8];
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = (sc[j] == -INFINITY) ? 0.f : expf(sc[j] - mnew);
            psum += p;
            ph[j] = T::from_f(p);
        }
        lsum = lsum * scale + psum;
        mrun = mnew;
        kg_u4 pf;
        pf.x = ph[0] | ((unsigned)ph[1] << 16); pf.y = ph[2] | ((unsigned)ph[3] << 16);
        pf.z = ph[4] | ((unsigned)ph[5] << 16); pf.w = ph[6] | ((unsigned)ph[7] << 16);
        // accumulator rows are query rows 4g..4g+3: fetch their rescale factors from the
        // lanes of this 16-lane group that own those columns
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __shfl(scale, (lane & 48) | (4 * g + i), 64);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) O[b][i] *= f[i];
            O[b] = T::mfma(pf, vf[b], O[b]);
        }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);

    // ---- merge waves through LDS ----
    __shared__ float s_o[ATT_NW][KG_MAX_ROWS][HD];
    __shared__ float s_ml[ATT_NW][KG_MAX_ROWS][2];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_o[wave][4 * g + i][16 * b + c] = O[b][i];
    if (g == 0) { s_ml[wave][c][0] = mrun; s_ml[wave][c][1] = lsum; }
    __syncthreads();
    for (int i = threadIdx.x; i < M * HD; i += blockDim.x) {
        const int m = i / HD, dmi = i - m * HD;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_NW; ++w) mx = fmaxf(mx, s_ml[w][m][0]);
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_NW; ++w) {
            const float wm = s_ml[w][m][0];
            const float fw = (wm == -INFINITY) ? 0.f : expf(wm - mx);
            o += fw * s_o[w][m][dmi];
            l += fw * s_ml[w][m][1];
        }
        if (n_act == 1) {
            ((unsigned short*)a.out)[((size_t)m * a.H + h) * HD + dmi] = T::from_f(o / l);
        } else {
            coh_store(a.part_o + (((size_t)split * a.max_rows + m) * a.H + h) * HD + dmi, o);
            if (dmi == 0) {
                float* ml = a.part_ml + (((size_t)split * a.max_rows + m) * a.H + h) * 2;
                coh_store(ml, mx); coh_store(ml + 1, l);
            }
        }
    }
    if (n_act > 1 && split_arrive(a.counters + h, n_act)) split_combine<DT>(a, h, HD, 0, M, n_act);
}

template <int DT, int LPR>
static void launch_lpr(const KgAttnArgs& a, hipStream_t st) {
    if (a.max_rows == 1) {
        dim3 grid(a.H, a.nsplit, 1);
        hipLaunchKernelGGL((kg_attn_kernel<DT, LPR, 1>), grid, dim3(ATT_NW * 64), 0, st, a);
    } else {
        dim3 grid(a.H, a.nsplit, (a.max_rows + 7) / 8);
        hipLaunchKernelGGL((kg_attn_kernel<DT, LPR, 8>), grid, dim3(ATT_NW * 64), 0, st, a);
    }
}
template <int DT>
static int launch_dt(const KgAttnArgs& a, hipStream_t st) {
    if (a.v_transposed) {
        dim3 grid(a.H, a.nsplit, 1);
        switch (a.hd) {
        case 32: hipLaunchKernelGGL((kg_attn_mfma_kernel<DT, 32>), grid, dim3(ATT_NW * 64), 0, st, a); break;
        case 64: hipLaunchKernelGGL((kg_attn_mfma_kernel<DT, 64>), grid, dim3(ATT_NW * 64), 0, st, a); break;
        case 128: hipLaunchKernelGGL((kg_attn_mfma_kernel<DT, 128>), grid, dim3(ATT_NW * 64), 0, st, a); break;
        default: return -1;
        }
    } else {
        switch (a.hd) {
        case 16: launch_lpr<DT, 2>(a, st); break;
        case 32: launch_lpr<DT, 4>(a, st); break;
        case 64: launch_lpr<DT, 8>(a, st); break;
        case 128: launch_lpr<DT, 16>(a, st); break;
        default: return -1;
        }
    }
    return 0;
}

int kg_launch_attn(const KgAttnArgs& a, int dtype, hipStream_t st) {
    if (!a.counters || a.nsplit < 1 || a.max_rows < 1 || a.max_rows > KG_MAX_ROWS || a.H % a.Hkv != 0) return -1;
    if (a.v_transposed && (a.t_max % 32 != 0)) return -1;
    int rc = (dtype == KG_F16) ? launch_dt<KG_F16>(a, st) : launch_dt<KG_BF16>(a, st);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
