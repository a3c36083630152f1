// Decode attention, MFMA variant, as a device function usable from a kernel of its own and
// from a persistent pass kernel (see attn.hip for the overview).
#pragma once
#include "kg_common.h"
#include "kg_kernels.h"

#define ATT_NW 4        // waves per team: one team works on one (query head, T-split)
#ifndef KG_AGENT
#define KG_AGENT __HIP_MEMORY_SCOPE_AGENT
#endif
#ifndef KG_AUX_COH
#define KG_AUX_COH 16
#endif
#ifndef KG_PERSIST_COH_LOADS
#define KG_PERSIST_COH_LOADS 1
#endif

// Device-coherent access to the split partials: these bypass the per-XCD L2, so a block on
// another XCD reads what was written without any cache maintenance in between.
__device__ __forceinline__ void coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, KG_AGENT); }
__device__ __forceinline__ float coh_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, KG_AGENT); }

#define KG_ATTN_MAX_SPLITS 32

// LDS scratch of split_combine (one per team that may run it)
struct KgAttnCombineLds {
    float f[KG_ATTN_MAX_SPLITS][KG_MAX_ROWS];      // weight of split s for row m: exp(m_s - max_s m_s)
    float l[KG_MAX_ROWS];                          // sum_s f_s l_s
};

// out[m][h][:] = sum_s f_s o_s / sum_s f_s l_s,  f_s = exp(m_s - max_s m_s), rows m0 .. m0+mr-1 (mr <= 16);
// run by the `nthr` threads tid = 0..nthr-1 of one team, which share `cl`.  EVERY thread of the block
// must make the call (it contains a block barrier); teams with active == false only keep it company.
// The partials come from memory with device-coherent loads, each a round trip of a microsecond
// or more, and a block can issue only one wave-level load every few cycles, so they are requested
// in bulk and wide: each thread takes four consecutive output elements of a row (one 16-byte load per
// split), only the ns splits in use are touched, and everything is in flight before anything
// waits.  The sums run in split order.  hd is a multiple of 4.
template <int DT, bool COH>
__device__ __forceinline__ void split_combine(const KgAttnArgs& a, int h, int hd, int m0, int mr, int ns, int tid, int nthr,
                                              KgAttnCombineLds& cl, bool active) {
    typedef KgT<DT> T;
    constexpr int SB = 8;                             // splits per batch of loads
    const int hq = hd >> 2;                           // quads per row
    const int nq = active ? mr * hq : 0;              // quads of this head's output rows
    const size_t sstride = (size_t)a.max_rows * a.H * hd;       // floats between consecutive splits
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)a.part_o, 0, (int)((size_t)ns * sstride * 4 > 0x7fffffffu ? 0x7fffffffu : (size_t)ns * sstride * 4), 0x00020000);
    kg_f4 v[2][SB];
    auto quad_off = [&](int q) {                      // float offset of quad q inside split 0
        const int ml = q / hq, dq = q - ml * hq;
        return ((size_t)(m0 + ml) * a.H + h) * hd + 4 * dq;
    };
    auto load_v = [&](int q0, int s0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = q0 + k * nthr + tid;
            const unsigned base = (q < nq) ? (unsigned)(quad_off(q) * 4) : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < SB; ++j)
                if (s0 + j < ns) v[k][j] = __builtin_bit_cast(kg_f4, __builtin_amdgcn_raw_buffer_load_b128(pr, (q < nq) ? base + (unsigned)((s0 + j) * sstride * 4) : 0xFFFFFFFFu, 0, KG_AUX_COH));
        }
    };
    if (nq > 0) load_v(0, 0);
    // phase A: one thread per row
    if (active && tid < mr) {
        const int m = m0 + tid;
        float ms[KG_ATTN_MAX_SPLITS], ls[KG_ATTN_MAX_SPLITS];
#pragma unroll
        for (int s = 0; s < KG_ATTN_MAX_SPLITS; ++s) {
            if (s < ns) {
                const unsigned long long u = __hip_atomic_load((const unsigned long long*)(a.part_ml + (((size_t)s * a.max_rows + m) * a.H + h) * 2), __ATOMIC_RELAXED, KG_AGENT);
                ms[s] = __uint_as_float((unsigned)u); ls[s] = __uint_as_float((unsigned)(u >> 32));
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < KG_ATTN_MAX_SPLITS; ++s) if (s < ns) mx = fmaxf(mx, ms[s]);
        float l = 0.f;
#pragma unroll
        for (int s = 0; s < KG_ATTN_MAX_SPLITS; ++s) {
            if (s < ns) {
                const float f = (ms[s] == -INFINITY) ? 0.f : expf(ms[s] - mx);
                cl.f[s][tid] = f;
                l += f * ls[s];
            }
        }
        cl.l[tid] = l;
    }
    __syncthreads();
    // phase B: quads q = tid + nthr * k
    for (int q0 = 0; q0 < nq; q0 += 2 * nthr) {
        kg_f4 o[2] = {kg_f4{0.f, 0.f, 0.f, 0.f}, kg_f4{0.f, 0.f, 0.f, 0.f}};
        for (int s0 = 0; s0 < ns; s0 += SB) {
            if (q0 > 0 || s0 > 0) load_v(q0, s0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = q0 + k * nthr + tid;
                const int ml = (q < nq) ? q / hq : 0;
#pragma unroll
                for (int j = 0; j < SB; ++j)
                    if (s0 + j < ns) {
                        const float f = cl.f[s0 + j][ml];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[k][e] += f * v[k][j][e];
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = q0 + k * nthr + tid;
            if (q < nq) {
                const int ml = q / hq, dq = q - ml * hq;
                unsigned short* dst = (unsigned short*)a.out + (size_t)(m0 + ml) * a.H * hd + (size_t)h * hd + 4 * dq;
                const float linv = cl.l[ml];
                unsigned short r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = T::from_f(o[k][e] / linv);
                if (COH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store(dst + e, r[e], __ATOMIC_RELAXED, KG_AGENT);
                } else {
                    *(unsigned long long*)dst = (unsigned long long)r[0] | ((unsigned long long)r[1] << 16) | ((unsigned long long)r[2] << 32) | ((unsigned long long)r[3] << 48);
                }
            }
        }
    }
}

// LDS of one team
template <int HD> struct KgAttnTeamLds {
    float o[ATT_NW][KG_MAX_ROWS][HD];
    float ml[ATT_NW][KG_MAX_ROWS][2];
    KgAttnCombineLds cl;
    int last;
};

// One 32-key tile of one wave: K rows as A fragments of the two score MFMAs, V^T pieces as B
// fragments of the P.V MFMAs.
template <int HD> struct KgAttnTile {
    kg_u4 kf[2][HD / 32];
    kg_u4 vf[HD / 16];
};

template <int HD, int AUX>
__device__ __forceinline__ void attn_tile_load(KgAttnTile<HD>& t, const __amdgpu_buffer_rsrc_t& kr, const __amdgpu_buffer_rsrc_t& vr,
                                               int t_max, int tb, int lane) {
    const int c = lane & 15, g = lane >> 4;
    const int krow = 8 * (c >> 2) + (c & 3);       // tile 0: MFMA row c <-> key tb + krow; tile 1: + 4
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < HD / 32; ++s)
            t.kf[tt][s] = __builtin_amdgcn_raw_buffer_load_b128(kr, (unsigned)(((size_t)(tb + krow + 4 * tt) * HD + 32 * s + 8 * g) * 2), 0, AUX);
#pragma unroll
    for (int b = 0; b < HD / 16; ++b)
        t.vf[b] = __builtin_amdgcn_raw_buffer_load_b128(vr, (unsigned)(((size_t)(16 * b + c) * t_max + tb + 8 * g) * 2), 0, AUX);
}

// scores, online softmax and P.V of one tile; (mrun, lsum, O) are the wave's running state
template <int DT, int HD>
__device__ __forceinline__ void attn_tile_compute(const KgAttnTile<HD>& t, const kg_u4 (&qf)[HD / 32], int tb, int tend, int lim,
                                                  float inv_scale, int lane, float& mrun, float& lsum, kg_f4 (&O)[HD / 16]) {
    typedef KgT<DT> T;
    constexpr int KS = HD / 32, DB = HD / 16;
    const int g = lane >> 4;
    float sc[8];
    float bm = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        kg_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) d = T::mfma(t.kf[tt][s], qf[s], d);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = tb + 8 * g + 4 * tt + i;
            // reference op boundaries: q.k^T stored, then divided by sqrt(hd) and stored
            const float v = kg_round<DT>(kg_round<DT>(d[i]) / inv_scale);
            sc[4 * tt + i] = (key < tend && key <= lim) ? v : -INFINITY;
            bm = fmaxf(bm, sc[4 * tt + i]);
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
        O[b] = T::mfma(pf, t.vf[b], O[b]);
    }
}

// ---------------------------------------------------------------------------------
// Per wave and 32 keys:  S^T = K q^T  as two 16-key tiles of v_mfma_f32_16x16x32 (A = K rows,
// B = q^T, so every lane owns one query column and 8 keys), softmax statistics per column
// with two cross-lane steps for ALL query rows at once, then  O += P V  with the same
// instruction (A = P straight from the score registers, B = V^T rows, 16 head dims per
// MFMA).  The key <-> MFMA-row assignment of the two score tiles is chosen so that the 8
// probabilities a lane holds are 8 consecutive keys, i.e. one 16-byte piece of a V^T row.
// Cost is independent of the number of query rows (<= 16), so the verify pass, prefill
// chunks and single-token decode share it, and a row's result does not depend on which
// other rows ride along (columns of an MFMA are independent).
//
// One call = one team of ATT_NW waves on item (h, split); `valid` = false teams only keep the
// block barriers company (every team of a block must make the same calls).  ttid = thread
// index inside the team.  COH: q, K, V and the output go through device-coherent accesses
// (the caller is a persistent kernel in which other blocks produced them).
template <int DT, int HD, bool COH>
__device__ __forceinline__ void kg_attn_mfma_item(const KgAttnArgs& a, bool valid, int h, int split, int n_act, int chunk,
                                                  int M, int len, int pos_c, int ttid, KgAttnTeamLds<HD>& lds) {
    typedef KgT<DT> T;
    const int lane = ttid & 63, wave = __builtin_amdgcn_readfirstlane(ttid >> 6);
    const int c = lane & 15, g = lane >> 4;
    constexpr int KS = HD / 32;                    // k-steps of the score MFMAs
    constexpr int DB = HD / 16;                    // 16-wide head-dim blocks of the output
    constexpr int AUX = (COH && KG_PERSIST_COH_LOADS) ? KG_AUX_COH : 0;
    if (valid) {
        const int kvh = h / (a.H / a.Hkv);
        const int tbeg = split * chunk, tend = min(len, tbeg + chunk);
        const int lim = (c < M) ? pos_c : -1;          // causal bound of this lane's query column
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (int)((size_t)M * a.H * HD * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)a.k_cache + (size_t)kvh * a.t_max * HD), 0,
                                                                            (int)((size_t)a.t_max * HD * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)a.v_cache + (size_t)kvh * HD * a.t_max), 0,
                                                                            (int)((size_t)a.t_max * HD * 2), 0x00020000);
        kg_u4 qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)       // rows >= M are out of range: zeros
            qf[s] = __builtin_amdgcn_raw_buffer_load_b128(qr, (unsigned)((((size_t)c * a.H + h) * HD + 32 * s + 8 * g) * 2), 0, AUX);
        float mrun = -INFINITY, lsum = 0.f;
        kg_f4 O[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b) O[b] = kg_f4{0.f, 0.f, 0.f, 0.f};
        const float inv_scale = sqrtf((float)HD);

        for (int tb = tbeg + 32 * wave; tb < tend; tb += 32 * ATT_NW) {
            KgAttnTile<HD> t;
            attn_tile_load<HD, AUX>(t, kr, vr, a.t_max, tb, lane);
            attn_tile_compute<DT, HD>(t, qf, tb, tend, lim, inv_scale, lane, mrun, lsum, O);
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        // ---- merge waves through LDS ----
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) lds.o[wave][4 * g + i][16 * b + c] = O[b][i];
        if (g == 0) { lds.ml[wave][c][0] = mrun; lds.ml[wave][c][1] = lsum; }
    }
    __syncthreads();
    if (valid) {
        for (int i = ttid; i < M * HD; i += ATT_NW * 64) {
            const int m = i / HD, dmi = i - m * HD;
            float mx = -INFINITY;
#pragma unroll
            for (int w = 0; w < ATT_NW; ++w) mx = fmaxf(mx, lds.ml[w][m][0]);
            float o = 0.f, l = 0.f;
#pragma unroll
            for (int w = 0; w < ATT_NW; ++w) {
                const float wm = lds.ml[w][m][0];
                const float fw = (wm == -INFINITY) ? 0.f : expf(wm - mx);
                o += fw * lds.o[w][m][dmi];
                l += fw * lds.ml[w][m][1];
            }
            if (n_act == 1) {
                unsigned short* dst = (unsigned short*)a.out + ((size_t)m * a.H + h) * HD + dmi;
                const unsigned short v = T::from_f(o / l);
                if (COH) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, KG_AGENT);
                else *dst = v;
            } else {
                coh_store(a.part_o + (((size_t)split * a.max_rows + m) * a.H + h) * HD + dmi, o);
                if (dmi == 0) {
                    float* ml = a.part_ml + (((size_t)split * a.max_rows + m) * a.H + h) * 2;
                    coh_store(ml, mx); coh_store(ml + 1, l);
                }
            }
        }
    }
    if (n_act > 1) {                                           // block-uniform: every team sees the same pass
        // The team that completes the head's set of partials merges them.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's partial stores have been acknowledged
        __syncthreads();
        if (ttid == 0) {
            int last = 0;
            if (valid) {
                const unsigned old = __hip_atomic_fetch_add(a.counters + h, 1u, __ATOMIC_RELAXED, KG_AGENT);
                last = (old == (unsigned)(n_act - 1));
                if (last) __hip_atomic_store(a.counters + h, 0u, __ATOMIC_RELAXED, KG_AGENT);
            }
            lds.last = last;
        }
        __syncthreads();
        split_combine<DT, COH>(a, h, HD, 0, M, n_act, ttid, ATT_NW * 64, lds.cl, lds.last != 0);
    }
    __syncthreads();                                           // the team's LDS is free for the next item
}
