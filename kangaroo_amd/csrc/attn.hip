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
#include "kg_attn_body.h"

#define ATT_U 4         // K and V loads in flight per lane per batch
#define KG_ATTN_PART_LDS (68 * 1024)   // LDS for a block's local chunk partials (8 chunks x 16 rows x 130 floats), beside its team buffers

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
    if (n_act > 1) {
        __shared__ KgAttnCombineLds cl;
        const bool last = split_arrive(a.counters + blockIdx.z * a.H + h, n_act);
        split_combine<DT, false>(a, h, hd, m0, mr, n_act, threadIdx.x, blockDim.x, cl, last);
    }
}

// ---------------------------------------------------------------------------------
// MFMA variant (head_dim 32 / 64 / 128, V cached transposed [Hkv][hd][t_max]).
//
// The keys are cut into fixed chunks (boundaries depend on t_max and nsplit only); a chunk is
// worked on by a team of 4 waves (32-key tiles, wave w takes tiles w, w+4, ...), whose
// results meet in LDS: that gives one partial (max, sum, acc) per chunk and query row, and
// the final rows are the merge of the chunk partials in chunk order.  This arithmetic is the
// same however the chunks are spread over teams and blocks.  A block owns `cpb` consecutive
// chunks of one head; its NT teams take them in turn (team t: chunks t, t + NT, ...) and walk
// them with the next tile's K/V already requested while the current one is computed.  If the
// chunks in use fit one block (contexts up to cpb * chunk keys) their partials stay in LDS and
// the block writes the final rows itself: no global partials, no counter, no further memory
// round trips.  Otherwise the partials go to memory and the block that arrives last at the
// head's counter merges them (a chain of four round trips, ~6 us).
// Blocks whose chunks hold no keys yet have nothing to do while the others sit in a chain of memory round trips that
// leaves the memory system idle.  They fetch the head of the next projection's weight stream instead: consecutive
// launches place block b on XCD (b + c) mod 8 with one c, so a block with linear index b shares an XCD with the
// consumer's blocks of the same index mod 8 (`rho`), and what it reads stays in that XCD's L2 for the consumer.  The
// blocks of one XCD share out that XCD's bytes by (idle_idx, idle_cnt); an index nobody holds (head counts that are
// not a multiple of 8) leaves its share unfetched.  The values are discarded; every address is range-checked by
// the buffer descriptor.
__device__ __forceinline__ void attn_idle_prefetch(const KgAttnArgs& a, unsigned rho, int idle_idx, int idle_cnt, int tid, int nthr) {
    const unsigned upg = (unsigned)a.pf_bytes_per_group >> 4;                 // 16-byte units fetched per row group
    const unsigned gpr = (unsigned)a.pf_groups >> 3;                          // row groups per XCD
    const unsigned U = gpr * upg;
    const unsigned u0 = (unsigned)((unsigned long long)U * idle_idx / idle_cnt), u1 = (unsigned)((unsigned long long)U * (idle_idx + 1) / idle_cnt);
    const unsigned long long total = (unsigned long long)a.pf_groups * a.pf_group_bytes;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.pf_w, 0, (int)(total > 0x7fffffffull ? 0x7fffffffull : total), 0x00020000);
    constexpr int PB = 8;
    for (int it = 0; it < 8; ++it) {                                          // at most 64 requests per thread
        const unsigned ub = u0 + (unsigned)tid + (unsigned)(it * PB * nthr);
        if (ub >= u1) break;
        kg_u4 v[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const unsigned u = ub + (unsigned)(j * nthr);
            const unsigned g = rho + 8u * (u / upg), o = u % upg;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, (u < u1) ? g * (unsigned)a.pf_group_bytes + o * 16u : 0xFFFFFFFFu, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) asm volatile("" :: "v"(v[j].x), "v"(v[j].y), "v"(v[j].z), "v"(v[j].w));
    }
}

template <int DT, int HD, int NT>
__global__ void __launch_bounds__(NT * ATT_NW * 64) kg_attn_mfma_kernel(const KgAttnArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int team = __builtin_amdgcn_readfirstlane(tid / (ATT_NW * 64)), ttid = tid % (ATT_NW * 64);
    const int wave = __builtin_amdgcn_readfirstlane(ttid >> 6);
    const int c = lane & 15, g = lane >> 4;
    // blockIdx.z selects a block of 16 query rows (prefill passes carry up to KG_GEMM_ROWS rows);
    // rows of a pass sit at consecutive positions
    const int row0 = 16 * blockIdx.z;
    // Every scalar the kernel needs (pass parameters from memory, kernel arguments) is fetched in one
    // batch up front: fetched where first used, each costs a scalar-cache miss of its own in a chain
    // of dependent waits (measured 2.3 us from entry to the first K/V request; the keys need none of it).
    const int d_active = dyn->active, Mtot = dyn->n_rows, d_pos0 = dyn->pos[0];
    asm volatile("" :: "s"(d_active), "s"(Mtot), "s"(d_pos0), "s"(a.H), "s"(a.Hkv), "s"(a.t_max), "s"(a.nsplit), "s"(a.max_rows),
                 "s"(a.q), "s"(a.k_cache), "s"(a.v_cache), "s"(a.part_o), "s"(a.part_ml), "s"(a.counters), "s"(a.out));
    const int pos_c = d_pos0 + row0 + c;
    if (d_active == 0) return;
    if (row0 >= Mtot) return;
    const int M = min(16, Mtot - row0);
    const int len = d_pos0 + row0 + M;
    const int h = blockIdx.x;
    int chunk = (a.t_max + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + 31) / 32 * 32;
    const int n_act = (len + chunk - 1) / chunk;   // chunks that hold keys
    const int cpb = (a.nsplit + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c0 = blockIdx.y * cpb, c1 = min(n_act, c0 + cpb);
    const int nblk_act = (n_act + cpb - 1) / cpb;  // blocks of this head that have chunks
    if (c0 >= n_act) {
        if (a.pf_w && blockIdx.z == 0 && n_act >= a.pf_min_chunks) {
            const int n_idle_y = (int)gridDim.y - nblk_act;
            for (int i = 0; i < a.pf_delay; i += 16) __builtin_amdgcn_s_sleep(16);
            attn_idle_prefetch(a, (unsigned)(h + a.H * (int)blockIdx.y) & 7u, (h >> 3) * n_idle_y + ((int)blockIdx.y - nblk_act), ((a.H + 7) >> 3) * n_idle_y,
                               tid, NT * ATT_NW * 64);
        }
        return;
    }
    const bool local = (nblk_act == 1);
    const int nit = (c1 - c0 + NT - 1) / NT;       // chunk rounds of this block (same for every team)

    constexpr int KS = HD / 32, DB = HD / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    KgAttnTeamLds<HD>* lds_all = (KgAttnTeamLds<HD>*)smem_;
    KgAttnTeamLds<HD>& lds = lds_all[team];
    float* pl_o = (float*)(smem_ + NT * sizeof(KgAttnTeamLds<HD>));   // local partials: [cpb][M][HD] then [cpb][M][2]
    float* pl_ml = pl_o + (size_t)cpb * M * HD;

    const int kvh = h / (a.H / a.Hkv);
    const int lim = (c < M) ? pos_c : -1;          // causal bound of this lane's query column
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (int)((size_t)(row0 + M) * a.H * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)a.k_cache + (size_t)kvh * a.t_max * HD), 0,
                                                                        (int)((size_t)a.t_max * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)a.v_cache + (size_t)kvh * HD * a.t_max), 0,
                                                                        (int)((size_t)a.t_max * HD * 2), 0x00020000);
    kg_u4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)       // rows >= M are out of range: zeros
        qf[s] = __builtin_amdgcn_raw_buffer_load_b128(qr, (unsigned)((((size_t)(row0 + c) * a.H + h) * HD + 32 * s + 8 * g) * 2), 0, 0);
    const float inv_scale = sqrtf((float)HD);

    // prefetch cursor over this wave's tiles: (chunk round pk, first key ptb)
    int pk = 0, ptb = (c0 + team) * chunk + 32 * wave;
    auto settle = [&]() {
        while (pk < nit) {
            const int ch = c0 + team + NT * pk;
            if (ch < c1 && ptb < min(len, ch * chunk + chunk)) break;
            ++pk;
            ptb = (c0 + team + NT * pk) * chunk + 32 * wave;
        }
    };
    settle();
    KgAttnTile<HD> nxt;
    if (pk < nit) attn_tile_load<HD, 0>(nxt, kr, vr, a.t_max, ptb, lane);

    for (int k = 0; k < nit; ++k) {
        const int ch = c0 + team + NT * k;
        const bool valid = ch < c1;                // team-uniform; a team without a chunk only keeps the barriers company
        const int tbeg = ch * chunk, tend = valid ? min(len, tbeg + chunk) : tbeg;
        float mrun = -INFINITY, lsum = 0.f;
        kg_f4 O[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b) O[b] = kg_f4{0.f, 0.f, 0.f, 0.f};
        for (int tb = tbeg + 32 * wave; tb < tend; tb += 32 * ATT_NW) {
            const KgAttnTile<HD> cur = nxt;        // == tile (ch, tb): the cursor visits tiles in this order
            ptb += 32 * ATT_NW;
            settle();
            if (pk < nit) attn_tile_load<HD, 0>(nxt, kr, vr, a.t_max, ptb, lane);
            attn_tile_compute<DT, HD>(cur, qf, tb, tend, lim, inv_scale, lane, mrun, lsum, O);
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        // ---- merge the waves of this chunk through the team's LDS ----
        if (valid) {
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) lds.o[wave][4 * g + i][16 * b + c] = O[b][i];
            if (g == 0) { lds.ml[wave][c][0] = mrun; lds.ml[wave][c][1] = lsum; }
        }
        __syncthreads();
        if (valid) {
            // one thread per (row, four consecutive head dims): M * HD / 4 <= 512 quads for the team's 256 threads
            for (int q = ttid; q < M * (HD / 4); q += ATT_NW * 64) {
                const int m = q / (HD / 4), dmi = 4 * (q - m * (HD / 4));
                float mx = -INFINITY;
#pragma unroll
                for (int w = 0; w < ATT_NW; ++w) mx = fmaxf(mx, lds.ml[w][m][0]);
                kg_f4 o = kg_f4{0.f, 0.f, 0.f, 0.f};
                float l = 0.f;
#pragma unroll
                for (int w = 0; w < ATT_NW; ++w) {
                    const float wm = lds.ml[w][m][0];
                    const float fw = (wm == -INFINITY) ? 0.f : expf(wm - mx);
                    const kg_f4 ow = *(const kg_f4*)&lds.o[w][m][dmi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += fw * ow[e];
                    l += fw * lds.ml[w][m][1];
                }
                if (local) {
                    *(kg_f4*)&pl_o[((size_t)(ch - c0) * M + m) * HD + dmi] = o;
                    if (dmi == 0) { pl_ml[((ch - c0) * M + m) * 2] = mx; pl_ml[((ch - c0) * M + m) * 2 + 1] = l; }
                } else {
                    float* dst = a.part_o + (((size_t)ch * a.max_rows + row0 + m) * a.H + h) * HD + dmi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) coh_store(dst + e, o[e]);
                    if (dmi == 0) {
                        float* ml = a.part_ml + (((size_t)ch * a.max_rows + row0 + m) * a.H + h) * 2;
                        coh_store(ml, mx); coh_store(ml + 1, l);
                    }
                }
            }
        }
        __syncthreads();                           // team LDS is free for the next chunk; local partials are visible
    }
    if (local) {
        // same merge as split_combine, partials read from LDS
        const int ns = c1 - c0;
        for (int i = tid; i < M * HD; i += NT * ATT_NW * 64) {
            const int m = i / HD, dmi = i - m * HD;
            float mx = -INFINITY;
            for (int s = 0; s < ns; ++s) mx = fmaxf(mx, pl_ml[(s * M + m) * 2]);
            float o = 0.f, l = 0.f;
            for (int s = 0; s < ns; ++s) {
                const float m_s = pl_ml[(s * M + m) * 2];
                const float f = (m_s == -INFINITY) ? 0.f : expf(m_s - mx);
                o += f * pl_o[((size_t)s * M + m) * HD + dmi];
                l += f * pl_ml[(s * M + m) * 2 + 1];
            }
            ((unsigned short*)a.out)[((size_t)(row0 + m) * a.H + h) * HD + dmi] = T::from_f(o / l);
        }
        return;
    }
    const bool last = split_arrive(a.counters + blockIdx.z * a.H + h, nblk_act);
    split_combine<DT, false>(a, h, HD, row0, M, n_act, tid, NT * ATT_NW * 64, lds_all[0].cl, last);
}

// ---------------------------------------------------------------------------------
// Causal attention of a whole prompt (rows at positions pos0 .. pos0 + rows - 1 over cache slots [0, pos0 + rows)), for the
// prompt pass that runs layer by layer: one WAVE owns 16 query rows and walks all their keys in 32-key tiles with the tile
// code of the decode kernel (scores and P.V on v_mfma_f32_16x16x32, online softmax, next tile requested before the current
// one is used), so there are no key splits, no partials and no merge; a block is four such waves (64 rows) of one head,
// whose K/V requests meet in L2.  Row blocks are dealt to waves from both ends of the prompt so that a block's work is
// even under the causal bound.  V is read from the transposed cache layout [Hkv][hd][t_max].
template <int DT, int HD>
__global__ void __launch_bounds__(256) kg_attn_rows_kernel(const unsigned short* q, const unsigned short* k_cache, const unsigned short* v_cache, unsigned short* out,
                                                           int rows, int pos0, int H, int Hkv, int t_max) {
    typedef KgT<DT> T;
    constexpr int KS = HD / 32, DB = HD / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, kvh = h / (H / Hkv);
    const int nrb = (rows + 15) >> 4;
    // wave slot s = 4 * blockIdx.x + wave takes row block s / 2 from the front (s even) or from the back (s odd)
    const int slot = 4 * blockIdx.x + wave;
    if (slot >= nrb) return;
    const int rb = (slot & 1) ? (nrb - 1 - (slot >> 1)) : (slot >> 1);
    const int row0 = 16 * rb, M = min(16, rows - row0);
    const int len = pos0 + row0 + M;
    const int lim = (c < M) ? (pos0 + row0 + c) : -1;
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, (int)((size_t)rows * H * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)(k_cache + (size_t)kvh * t_max * HD), 0, (int)((size_t)t_max * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)(v_cache + (size_t)kvh * HD * t_max), 0, (int)((size_t)t_max * HD * 2), 0x00020000);
    kg_u4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)       // rows beyond the prompt are out of range: zeros
        qf[s] = __builtin_amdgcn_raw_buffer_load_b128(qr, (unsigned)((((size_t)(row0 + c) * H + h) * HD + 32 * s + 8 * g) * 2), 0, 0);
    const float inv_scale = sqrtf((float)HD);
    float mrun = -INFINITY, lsum = 0.f;
    kg_f4 O[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) O[b] = kg_f4{0.f, 0.f, 0.f, 0.f};
    KgAttnTile<HD> nxt;
    attn_tile_load<HD, 0>(nxt, kr, vr, t_max, 0, lane);
    for (int tb = 0; tb < len; tb += 32) {
        const KgAttnTile<HD> cur = nxt;
        if (tb + 32 < len) attn_tile_load<HD, 0>(nxt, kr, vr, t_max, tb + 32, lane);
        attn_tile_compute<DT, HD>(cur, qf, tb, len, lim, inv_scale, lane, mrun, lsum, O);
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    // accumulator rows are query rows 4g .. 4g + 3, their sums sit in the lanes that own those columns
    float linv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) linv[i] = 1.0f / __shfl(lsum, (lane & 48) | (4 * g + i), 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = row0 + 4 * g + i;
        if (4 * g + i < M) {
            unsigned short* dst = out + ((size_t)m * H + h) * HD + c;
#pragma unroll
            for (int b = 0; b < DB; ++b) dst[16 * b] = T::from_f(O[b][i] * linv[i]);
        }
    }
}

// The same attention for head_dim 128 with the keys shared: a block of four waves owns 128 query rows (32 per wave, two
// MFMA column tiles) and walks the keys in tiles of 64 that are staged ONCE per block in LDS by loads that write LDS
// directly (K rows of 256 bytes, V^T rows of 128 bytes; two buffers of 32 KB; the XOR swizzles that keep the fragment
// reads off each other's banks sit on the source addresses).  Per tile and wave: S^T = K q^T (rows of the MFMA = keys,
// assigned so that the 8 probabilities a lane ends up with are 8 consecutive keys), online softmax per query column (a
// lane owns ONE query per column tile: no cross-lane traffic but the two steps of the column maximum), O^T += V^T P^T.
// Arithmetic and rounding points are attn_tile_compute's.  Row blocks are dealt longest first.
template <int DT>
__global__ void __launch_bounds__(256, 2) kg_attn_rows128_kernel(const unsigned short* q, const unsigned short* k_cache, const unsigned short* v_cache, unsigned short* out,
                                                                  int rows, int pos0, int H, int Hkv, int t_max) {
    typedef KgT<DT> T;
    constexpr int HD = 128, KT = 64, TILE = 2 * KT * HD * 2;      // bytes of one K + V^T tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, kvh = h / (H / Hkv);
    const int nblk = (rows + 127) >> 7, rb = nblk - 1 - (int)blockIdx.x;
    const int r0 = 128 * rb + 32 * wave;                          // this wave's first row
    const int len_blk = pos0 + min(rows, 128 * rb + 128);         // keys the block walks
    const int len_w = pos0 + min(rows, r0 + 32);                  // keys this wave's rows can see (<= 0 rows: nothing)
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, (int)((size_t)rows * H * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)(k_cache + (size_t)kvh * t_max * HD), 0, (int)((size_t)t_max * HD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)(v_cache + (size_t)kvh * HD * t_max), 0, (int)((size_t)t_max * HD * 2), 0x00020000);
    kg_u4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)     // rows beyond the prompt are out of range: zeros
            qf[qt][ks] = __builtin_amdgcn_raw_buffer_load_b128(qr, (unsigned)((((size_t)(r0 + 16 * qt + fr) * H + h) * HD + 32 * ks + 8 * fg) * 2), 0, 0);
    // ---- staging: per tile and thread 4 loads of K (4 rows of 256 B per wave-level load) and 4 of V^T (8 rows of 128 B) ----
    unsigned k_src[4], v_src[4];           // byte offsets at tile 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr_ = (4 * i + wave) * 4 + (lane >> 4), ks_ = lane & 15;                   // LDS row (key), 16-byte slot
        k_src[i] = (unsigned)(((size_t)kr_ * HD + 8 * (ks_ ^ (4 * ((kr_ >> 3) & 3) + (kr_ & 3)))) * 2);
        const int vr_ = (4 * i + wave) * 8 + (lane >> 3), vs_ = lane & 7;                    // LDS row (head dim), 16-byte slot
        v_src[i] = (unsigned)(((size_t)vr_ * t_max + 8 * (vs_ ^ ((vr_ >> 1) & 7))) * 2);
    }
    const int ntile = (len_blk + KT - 1) / KT;
    auto request = [&](int t) {
        char* buf = smem + (t & 1) * TILE;
        const bool ok = t < ntile;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, (__attribute__((address_space(3))) void*)(buf + (4 * i + wave) * 1024), 16,
                                                     ok ? k_src[i] + (unsigned)((size_t)t * KT * HD * 2) : 0xFFFFFFFFu, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (__attribute__((address_space(3))) void*)(buf + KT * HD * 2 + (4 * i + wave) * 1024), 16,
                                                     ok ? v_src[i] + (unsigned)(t * KT * 2) : 0xFFFFFFFFu, 0, 0, 0);
    };
    const float inv_scale = sqrtf((float)HD), rcp_scale = 1.0f / inv_scale;
    float mrun[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
    kg_f4 O[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int b = 0; b < 8; ++b) O[qt][b] = kg_f4{0.f, 0.f, 0.f, 0.f};
    int lim[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) lim[qt] = (r0 + 16 * qt + fr < rows) ? pos0 + r0 + 16 * qt + fr : -1;

    request(0);
    for (int t = 0; t < ntile; ++t) {
        request(t + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tile t has landed (this thread's share) ...
        __builtin_amdgcn_s_barrier();                             // ... and everybody's
        const int tb = t * KT;
        if (tb < len_w) {                                         // tiles beyond this wave's last row: nothing to add
            const char* kb = smem + (t & 1) * TILE;
            const char* vb = kb + KT * HD * 2;
            // Only the tiles that reach a row's diagonal need the causal test (rows of the wave: r0 .. r0 + 31).
            const bool diag = tb + KT - 1 > pos0 + r0;
            float sc[2][16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                // MFMA row fr of score tile kt is key 32 (kt >> 1) + 8 (fr >> 2) + 4 (kt & 1) + (fr & 3)
                const int krow = 32 * (kt >> 1) + 8 * (fr >> 2) + 4 * (kt & 1) + (fr & 3);
                kg_u4 kf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const kg_u4*)(kb + krow * 256 + (((4 * ks + fg) ^ fr) << 4));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    kg_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) d = T::mfma(kf[ks], qf[qt][ks], d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // q.k^T stored, then divided by sqrt(hd) and stored.  The quotient by a constant: one product with the
                        // rounded reciprocal and one correction step give the correctly rounded fp32 quotient (Markstein): for
                        // every finite 16-bit x and sqrt(32 / 64 / 128) the same bits as the division, but for the sign of a
                        // zero quotient and x below 2^-126 (checked exhaustively on the host).
                        const float x = kg_round<DT>(d[e]);
                        const float q0 = x * rcp_scale;
                        sc[qt][4 * kt + e] = kg_round<DT>(__builtin_fmaf(__builtin_fmaf(-q0, inv_scale, x), rcp_scale, q0));
                    }
                }
            }
            if (diag) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int key = tb + 32 * (j >> 3) + 8 * fg + (j & 7);       // j = 4 kt + e: 32 (kt >> 1) + 4 (kt & 1) + e
                        if (key > lim[qt]) sc[qt][j] = -INFINITY;
                    }
            }
            kg_u4 pf[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float bm = sc[qt][0];
#pragma unroll
                for (int j = 1; j < 16; ++j) bm = fmaxf(bm, sc[qt][j]);
                bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
                bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
                const float mnew = fmaxf(mrun[qt], bm);
                // exp(x - m) as 2^(x log2 e - m log2 e); a masked score is -inf and gives 0, and so does a first tile's mrun;
                // a row with no key yet (rows beyond the prompt) keeps m = -inf: its products are never stored
                const float ml = (mnew == -INFINITY) ? 0.f : mnew * 1.44269504088896340736f;
                const float scale = __builtin_amdgcn_exp2f(__builtin_fmaf(mrun[qt], 1.44269504088896340736f, -ml));
                float psum = 0.f;
                float p[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qt][j], 1.44269504088896340736f, -ml));
                    psum += p[j];
                }
                lsum[qt] = lsum[qt] * scale + psum;
                mrun[qt] = mnew;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    pf[qt][ks].x = kg_pack2<DT>(p[8 * ks + 0], p[8 * ks + 1]); pf[qt][ks].y = kg_pack2<DT>(p[8 * ks + 2], p[8 * ks + 3]);
                    pf[qt][ks].z = kg_pack2<DT>(p[8 * ks + 4], p[8 * ks + 5]); pf[qt][ks].w = kg_pack2<DT>(p[8 * ks + 6], p[8 * ks + 7]);
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) O[qt][b] = O[qt][b] * scale;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                kg_u4 vf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vf[ks] = *(const kg_u4*)(vb + (16 * b + fr) * 128 + (((4 * ks + fg) ^ (((16 * b + fr) >> 1) & 7)) << 4));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) O[qt][b] = T::mfma(vf[ks], pf[qt][ks], O[qt][b]);     // D[d = 4 fg + e][q = fr]
            }
        }
        __builtin_amdgcn_s_barrier();                             // everybody has left buffer t & 1: the next request may refill it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = lsum[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int m = r0 + 16 * qt + fr;
        if (m < rows) {
            const float linv = 1.0f / l;
            unsigned short* dst = out + ((size_t)m * H + h) * HD + 4 * fg;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const kg_f4 o = O[qt][b] * linv;
                *(unsigned long long*)(dst + 16 * b) = (unsigned long long)T::from_f(o[0]) | ((unsigned long long)T::from_f(o[1]) << 16) |
                                                       ((unsigned long long)T::from_f(o[2]) << 32) | ((unsigned long long)T::from_f(o[3]) << 48);
            }
        }
    }
}

int kg_launch_attn_rows(const void* q, const void* k_cache, const void* v_cache, void* out, int rows, int pos0, int H, int Hkv, int hd, int t_max, int dtype, hipStream_t st) {
    if (!q || !k_cache || !v_cache || !out || rows < 1 || pos0 < 0 || pos0 + rows > t_max || H % Hkv || (t_max & 31)) return -1;
    static int use128 = -1;
    if (use128 < 0) {
        const char* e = getenv("KG_ATTN_ROWS128");                 // measurements: 0 keeps the one-wave-per-16-rows kernel
        use128 = (e && atoi(e) == 0) ? 0 : 1;
        (void)hipFuncSetAttribute((const void*)kg_attn_rows128_kernel<KG_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)kg_attn_rows128_kernel<KG_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    }
    if (hd == 128 && use128 && (t_max & 63) == 0 && (size_t)t_max * 128 * 2 < 0x7fffffffu) {
        const dim3 g128((rows + 127) / 128, H);
        if (dtype == KG_F16) hipLaunchKernelGGL(kg_attn_rows128_kernel<KG_F16>, g128, dim3(256), 65536, st, (const unsigned short*)q, (const unsigned short*)k_cache, (const unsigned short*)v_cache, (unsigned short*)out, rows, pos0, H, Hkv, t_max);
        else hipLaunchKernelGGL(kg_attn_rows128_kernel<KG_BF16>, g128, dim3(256), 65536, st, (const unsigned short*)q, (const unsigned short*)k_cache, (const unsigned short*)v_cache, (unsigned short*)out, rows, pos0, H, Hkv, t_max);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 grid(((rows + 15) / 16 + 3) / 4, H);
#define KG_AR(D, HD_) hipLaunchKernelGGL((kg_attn_rows_kernel<D, HD_>), grid, dim3(256), 0, st, (const unsigned short*)q, (const unsigned short*)k_cache, (const unsigned short*)v_cache, (unsigned short*)out, rows, pos0, H, Hkv, t_max)
    if (dtype == KG_F16) { switch (hd) { case 32: KG_AR(KG_F16, 32); break; case 64: KG_AR(KG_F16, 64); break; case 128: KG_AR(KG_F16, 128); break; default: return -1; } }
    else { switch (hd) { case 32: KG_AR(KG_BF16, 32); break; case 64: KG_AR(KG_BF16, 64); break; case 128: KG_AR(KG_BF16, 128); break; default: return -1; } }
#undef KG_AR
    return hipGetLastError() == hipSuccess ? 0 : -2;
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
        const int nby = a.blocks_per_head < 1 ? 1 : (a.blocks_per_head > a.nsplit ? a.nsplit : a.blocks_per_head);
        const int cpb = (a.nsplit + nby - 1) / nby;
        const int nt = (a.teams == 2 && cpb >= 2) ? 2 : 1;          // a second team only pays with several chunks per block
        dim3 grid(a.H, nby, (a.max_rows + 15) / 16);
        const size_t part = (size_t)cpb * (a.max_rows < 16 ? a.max_rows : 16) * (a.hd + 2) * sizeof(float);
        if (part > (size_t)KG_ATTN_PART_LDS) return -1;
#define KG_ATTN_LAUNCH(HD_) do { \
            const size_t smem = part + nt * sizeof(KgAttnTeamLds<HD_>); \
            if (nt == 2) hipLaunchKernelGGL((kg_attn_mfma_kernel<DT, HD_, 2>), grid, dim3(2 * ATT_NW * 64), smem, st, a); \
            else hipLaunchKernelGGL((kg_attn_mfma_kernel<DT, HD_, 1>), grid, dim3(ATT_NW * 64), smem, st, a); } while (0)
        switch (a.hd) {
        case 32: KG_ATTN_LAUNCH(32); break;
        case 64: KG_ATTN_LAUNCH(64); break;
        case 128: KG_ATTN_LAUNCH(128); break;
        default: return -1;
        }
#undef KG_ATTN_LAUNCH
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

// One-time opt-in for > 64 KiB of LDS (team buffers + local chunk partials of 16-row passes).
template <int DT, int HD> static void attn_set_attr() {
    (void)hipFuncSetAttribute((const void*)kg_attn_mfma_kernel<DT, HD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KG_ATTN_PART_LDS + sizeof(KgAttnTeamLds<HD>)));
    (void)hipFuncSetAttribute((const void*)kg_attn_mfma_kernel<DT, HD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KG_ATTN_PART_LDS + 2 * sizeof(KgAttnTeamLds<HD>)));
}
void kg_attn_init() {
    attn_set_attr<KG_F16, 32>(); attn_set_attr<KG_F16, 64>(); attn_set_attr<KG_F16, 128>();
    attn_set_attr<KG_BF16, 32>(); attn_set_attr<KG_BF16, 64>(); attn_set_attr<KG_BF16, 128>();
    (void)hipGetLastError();
}

int kg_launch_attn(const KgAttnArgs& a_in, int dtype, hipStream_t st) {
    KgAttnArgs a = a_in;
    // the hand-over through L2 needs the consumer's blocks and row groups to fall evenly on the 8 XCDs
    if (a.pf_w && ((a.pf_grid & 7) || (a.pf_groups & 7) || a.pf_group_bytes <= 0 || a.pf_bytes_per_group < 16)) a.pf_w = nullptr;
    if (a.pf_w) a.pf_bytes_per_group = (a.pf_bytes_per_group > a.pf_group_bytes ? a.pf_group_bytes : a.pf_bytes_per_group) & ~15;
    if (!a.counters || a.nsplit < 1 || a.nsplit > KG_ATTN_MAX_SPLITS || a.max_rows < 1 || a.max_rows > (a.v_transposed ? KG_GEMM_ROWS : KG_MAX_ROWS) || a.H % a.Hkv != 0) return -1;
    if (a.v_transposed && (a.t_max % 32 != 0)) return -1;
    int rc = (dtype == KG_F16) ? launch_dt<KG_F16>(a, st) : launch_dt<KG_BF16>(a, st);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
