// Weight-streaming linear kernels for bsz=1 decode (gfx950).
//
//   y[m][n] = sum_k X[m][k] * W[n][k]        m < n_rows <= 16
//
// W is stored tile-major ("packed"): tile (rg, kt) holds rows 16*rg..16*rg+15 and
// k = 32*kt..32*kt+31 as 1 KiB in the order [g = k/8][r = row][j = k%8], so that lane
// l = 16*g + r of a wave reads its 16 B at byte 16*l: every wave-level load is one
// contiguous 1 KiB, and consecutive k-tiles of a row group are consecutive in memory.
// The same 16 B are the lane's A fragment of v_mfma_f32_16x16x32 (rows on l&15,
// k-group on l>>4).
//
// Two arithmetic modes share that stream:
//   VEC   one row (q = 1): v_dot2_f32 per lane, 4-lane k reduction by wave shuffles.
//   MFMA  up to 16 rows (the (k+1)-token verify pass): X^T is the B operand.
// A block's waves split K of one row group; partial sums meet in LDS, then a few
// threads run the fused epilogue (residual add, RoPE + KV append, SiLU*mul, or the
// lm_head argmax / softmax partials).  The prologue fills X in LDS (plain copy or
// RMSNorm) after the first weight loads are already in flight.
#pragma once
#include "kg_common.h"
#include "kg_kernels.h"

// Optional in-kernel timeline for the microbenchmarks (scripts/microbench.hip defines KG_LIN_TLOG and
// the buffer): thread 0 of every block stamps the 100 MHz clock at a few points of the pass:
// 0 entry, 1 parameters known, 2 X in LDS (after the block barrier), 3 first row group reduced, 4 end.
#ifdef KG_LIN_TLOG
extern __device__ unsigned long long kg_lin_tlog[];
#define KG_TSTAMP(a, i) do { if (threadIdx.x == 0) kg_lin_tlog[((size_t)(a).logits_stride * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define KG_TSTAMP(a, i) do { } while (0)
#endif

#ifndef KG_U_VEC
#define KG_U_VEC 8      // weight tiles (KiB) per batch and wave; two batches are in flight
#endif
#ifndef KG_U_MFMA
#define KG_U_MFMA 8
#endif

enum { MODE_VEC = 0, MODE_MFMA = 1, MODE_MFMA_GX = 2 };

// Accesses to data that another block wrote earlier in the SAME kernel (persistent passes):
// device-coherent, i.e. they bypass the per-XCD L2 instead of relying on cache maintenance
// at a kernel boundary.  COH = false gives the ordinary cached accesses.
#define KG_AGENT __HIP_MEMORY_SCOPE_AGENT
#ifndef KG_AUX_COH
#define KG_AUX_COH 16           // buffer-load cache policy: sc1 (agent scope)
#endif
// Persistent passes: what another block wrote earlier in the same launch is read with
// device-coherent (sc1) loads, the counterpart of the sc1 stores.  Every layer also has its own
// activation buffers, so a line is written once per launch and only read afterwards; with that,
// ordinary cached loads (0) gave identical results in all tests here and fetch the rows once per
// XCD instead of once per block, but only the sc1/sc1 pairing is a documented hand-off.
#ifndef KG_PERSIST_COH_LOADS
#define KG_PERSIST_COH_LOADS 1
#endif
template <bool COH> __device__ __forceinline__ void st16(unsigned short* p, unsigned short v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, KG_AGENT);
    else *p = v;
}

template <int DT> __device__ __forceinline__ float silu_mul_ref(float g, float u) {
    // reference op boundaries: gate, up stored; silu stored; product stored
    float s = kg_round<DT>(g / (1.0f + expf(-g)));
    return kg_round<DT>(s * u);
}

struct HeadRun {
    float mx; float sum; int idx;
};
__device__ __forceinline__ void head_push(HeadRun& h, float v, int n) {
    if (v > h.mx) {
        h.sum = h.sum * expf(h.mx - v) + 1.0f;
        h.mx = v;
        h.idx = n;
    } else {
        h.sum += expf(v - h.mx);
    }
}
__device__ __forceinline__ void head_merge(HeadRun& a, float bmx, float bsum, int bidx) {
    float M = fmaxf(a.mx, bmx);
    float sa = (a.mx == -INFINITY) ? 0.0f : a.sum * expf(a.mx - M);
    float sb = (bmx == -INFINITY) ? 0.0f : bsum * expf(bmx - M);
    bool take_b = (bmx > a.mx) || (bmx == a.mx && bidx < a.idx);
    a.idx = take_b ? bidx : a.idx;
    a.mx = M;
    a.sum = sa + sb;
}

// Epilogue for one (row pair, column m).  n0 = packed row index of v0; v1 is row n0+1.
template <int DT, int EPI, bool COH>
__device__ __forceinline__ void epilogue_pair(const KgLinArgs& a, int n0, int m, int pos, float v0, float v1, HeadRun& run,
                                              const char* side, int it, int M) {
    typedef KgT<DT> T;
    unsigned short* out = (unsigned short*)a.out;
    if (EPI == EPI_STORE) {
        if (n0 < a.N) st16<COH>(&out[(size_t)m * a.out_stride + n0], T::from_f(v0));
        if (n0 + 1 < a.N) st16<COH>(&out[(size_t)m * a.out_stride + n0 + 1], T::from_f(v1));
    } else if (EPI == EPI_STORE_F32) {
        // unrounded fp32 sums: the partial products of a tensor-parallel shard, added up across ranks by the caller
        float* o32 = (float*)a.out;
        if (n0 < a.N) o32[(size_t)m * a.out_stride + n0] = v0;
        if (n0 + 1 < a.N) o32[(size_t)m * a.out_stride + n0 + 1] = v1;
    } else if (EPI == EPI_RESID) {
        // residual rows of this block were staged in LDS by the prologue: [it][m][16]
        const unsigned short* res = (const unsigned short*)side + ((size_t)it * M + m) * 16 + (n0 & 15);
        if (n0 < a.N) st16<COH>(&out[(size_t)m * a.out_stride + n0], T::from_f(T::to_f(res[0]) + kg_round<DT>(v0)));
        if (n0 + 1 < a.N) st16<COH>(&out[(size_t)m * a.out_stride + n0 + 1], T::from_f(T::to_f(res[1]) + kg_round<DT>(v1)));
    } else if (EPI == EPI_SILU_MUL) {
        // packed rows (2i, 2i+1) = (gate_i, up_i)
        int i = n0 >> 1;
        if (n0 + 1 < a.N) st16<COH>(&out[(size_t)m * a.out_stride + i], T::from_f(silu_mul_ref<DT>(kg_round<DT>(v0), kg_round<DT>(v1))));
    } else if (EPI == EPI_ROPE_KV) {
        // packed rows inside a head: (2j, 2j+1) = original (j, j + hd/2)
        if (n0 + 1 >= a.N) return;
        const int hd = a.head_dim, half = hd >> 1;
        const int head = n0 / hd, hl = n0 - head * hd, j = hl >> 1;
        float x1 = kg_round<DT>(v0), x2 = kg_round<DT>(v1);
        if (n0 < a.q_dim + a.kv_dim) {
            // cos / sin rows of the pass's positions were staged in LDS: [m][2][hd/2]
            const float* cs = (const float*)side + (size_t)m * hd;
            float c = kg_round<DT>(cs[j]);
            float s = kg_round<DT>(cs[half + j]);
            float o1 = kg_round<DT>(x1 * c) + kg_round<DT>(-x2 * s);
            float o2 = kg_round<DT>(x2 * c) + kg_round<DT>(x1 * s);
            x1 = o1; x2 = o2;    // stored through from_f below (one more rounding, as the reference's add)
        }
        unsigned short* dst;
        if (n0 < a.q_dim) {
            dst = (unsigned short*)a.q_out + (size_t)m * a.q_dim + head * hd;
        } else if (n0 < a.q_dim + a.kv_dim) {
            int kvh = head - a.q_dim / hd;
            dst = (unsigned short*)a.k_cache + ((size_t)kvh * a.t_max + pos) * hd;
        } else {
            int kvh = head - (a.q_dim + a.kv_dim) / hd;
            if (a.v_transposed) {
                unsigned short* vt = (unsigned short*)a.v_cache + (size_t)kvh * hd * a.t_max + pos;
                st16<COH>(&vt[(size_t)j * a.t_max], T::from_f(x1));
                st16<COH>(&vt[(size_t)(j + half) * a.t_max], T::from_f(x2));
                return;
            }
            dst = (unsigned short*)a.v_cache + ((size_t)kvh * a.t_max + pos) * hd;
        }
        st16<COH>(&dst[j], T::from_f(x1));
        st16<COH>(&dst[j + half], T::from_f(x2));
    } else if (EPI == EPI_HEAD) {
        float l0 = kg_round<DT>(v0), l1 = kg_round<DT>(v1);
        if (a.logits_out) {
            unsigned short* lo = (unsigned short*)a.logits_out;
            if (n0 < a.N) st16<COH>(&lo[(size_t)m * a.logits_stride + n0], T::from_f(v0));
            if (n0 + 1 < a.N) st16<COH>(&lo[(size_t)m * a.logits_stride + n0 + 1], T::from_f(v1));
        }
        if (n0 < a.N) head_push(run, l0, n0);
        if (n0 + 1 < a.N) head_push(run, l1, n0 + 1);
    }
}

// ---- prologue: fill X rows in LDS -------------------------------------------------
// xs: LDS base of X, row stride xs_stride bytes.  The whole block works on every row:
// thread t owns the 16-byte chunks t, t + blockDim, ... of each row, so the sum of squares
// of a row is formed in the same order whatever the number of rows in the pass.
//
// Fast path (a row has at most one chunk per thread): prologue_load issues the chunks of
// all rows as range-checked buffer loads BEFORE the first weight tiles are requested and
// keeps them in registers; prologue_finish runs after the weight loads are in flight, so
// the norm and the LDS fill overlap the first HBM round trip (the memory counter retires
// in order: x requested first is waited for with the weight loads still outstanding).
// RMSNorm follows the reference's order of operations (kangaroo/adapter.py:254-259): the row factor
// rsqrt(mean(x^2) + eps) is formed first (per-thread partial sums, one wave butterfly, one LDS slot per
// wave, one block barrier), x * factor is rounded to the model dtype, and the product with gamma is rounded
// again.  The extra barrier sits in the shadow of the first weight round trip.
// NS register slots of 16 bytes: slot i holds chunk set i % SETS of row i / SETS (SETS = chunks of a row per thread)
template <int NS> struct ProRegs { kg_u4 v[NS]; kg_u4 gw[2]; };

// Row factors of the RMSNorm: every thread adds the squares of the chunks it owns, a wave butterfly and
// one LDS slot per (row, wave) follow, and after a block barrier every thread sums the wave slots in wave
// order.  The order of the additions depends on the thread count only, never on the rows riding along.
template <int RM>
__device__ __forceinline__ void rms_row_factors(float (&ss)[RM], float* red2, int M, int K, float eps, int lane, int wave, int nwaves) {
    // red2 holds 8 slots per row whatever the wave count (unused ones are zero), so a row's partial sums
    // come back as two 16-byte reads
#pragma unroll
    for (int m = 0; m < RM; ++m) {
        if (m < M) {
            const float t = kg_wave_sum_dpp(ss[m]);
            if (lane == 0) red2[m * 8 + wave] = t;
            if (lane == 1 && wave == 0) { for (int w = nwaves; w < 8; ++w) red2[m * 8 + w] = 0.f; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < RM; ++m) {
        if (m < M) {
            const kg_f4 a = *(const kg_f4*)(red2 + m * 8), b = *(const kg_f4*)(red2 + m * 8 + 4);
            const float tot = (((a[0] + a[1]) + a[2]) + a[3]) + (((b[0] + b[1]) + b[2]) + b[3]);
            ss[m] = rsqrtf(tot / (float)K + eps);
        }
    }
}

template <int DT> __device__ __forceinline__ kg_u4 rms_apply(kg_u4 xv, kg_u4 gv, float rstd) { return KgT<DT>::rms_apply(xv, gv, rstd); }
template <int DT> __device__ __forceinline__ float sum_sq8(kg_u4 xv) { return KgT<DT>::sum_sq8(xv); }

// K beyond what the register slots hold: two sweeps over the rows (sum of squares, then normalise and store)
template <int DT, int PRO, int RM, bool COH_LD>
__device__ __forceinline__ void prologue_slow(const KgLinArgs& a, char* xs, int xs_stride, float* red2, int M, int row_off,
                                              int tid, int nthreads, int lane, int wave, int nwaves) {
    const int nchunk = a.K >> 3;
    const kg_u4* wn = (const kg_u4*)a.norm_w;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0,
        (int)(((size_t)(M - 1 + row_off) * a.x_stride + a.K) * 2), 0x00020000);
    float ss[RM];
#pragma unroll
    for (int m = 0; m < RM; ++m) ss[m] = 0.f;
    if (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int m = 0; m < RM; ++m) {
            if (m < M) {
                for (int c = tid; c < nchunk; c += nthreads)
                    ss[m] += sum_sq8<DT>(__builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(((size_t)(m + row_off) * a.x_stride) * 2 + (size_t)c * 16), 0, COH_LD ? KG_AUX_COH : 0));
            }
        }
        rms_row_factors<RM>(ss, red2, M, a.K, a.eps, lane, wave, nwaves);
    }
#pragma unroll
    for (int m = 0; m < RM; ++m) {
        if (m < M) {
            kg_u4* dst = (kg_u4*)(xs + (size_t)m * xs_stride);
            for (int c = tid; c < nchunk; c += nthreads) {
                const kg_u4 xv = __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(((size_t)(m + row_off) * a.x_stride) * 2 + (size_t)c * 16), 0, COH_LD ? KG_AUX_COH : 0);
                dst[c] = (PRO == PRO_NONE) ? xv : rms_apply<DT>(xv, wn[c], ss[m]);
            }
        }
    }
}

template <int PRO, int NS, int SETS, bool COH_LD>
__device__ __forceinline__ void prologue_load(const KgLinArgs& a, ProRegs<NS>& pr, int M, int row_off, int tid, int nthreads) {
    const int K = a.K, nchunk = K >> 3;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0,
        (int)(((size_t)(M - 1 + row_off) * a.x_stride + K) * 2), 0x00020000);
    // A load whose offset is out of range moves no data but still takes its turn in the CU's address path (measured:
    // as long as a real one), so the slots beyond the first 8 are only requested when the pass has rows for them.
    // The branch is block-uniform and both sides keep a fixed number of loads, so the counted waits stay exact enough.
    auto load_slot = [&](int i) {
        const int m = i / SETS, c = tid + (i % SETS) * nthreads;
        const bool ok = (m < M) && (c < nchunk);
        const unsigned off = ok ? (unsigned)(((size_t)(m + row_off) * a.x_stride) * 2 + (size_t)c * 16) : 0xFFFFFFFFu;
        pr.v[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, COH_LD ? KG_AUX_COH : 0);
    };
#pragma unroll
    for (int i = 0; i < (NS < 8 ? NS : 8); ++i) load_slot(i);
    if (NS > 8 && M * SETS > 8) {
#pragma unroll
        for (int i = 8; i < NS; ++i) load_slot(i);
    }
    if (PRO == PRO_RMSNORM) {
        const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc((void*)a.norm_w, 0, K * 2, 0x00020000);
#pragma unroll
        for (int j = 0; j < SETS; ++j) {
            const int c = tid + j * nthreads;
            pr.gw[j] = __builtin_amdgcn_raw_buffer_load_b128(nr, (c < nchunk) ? (unsigned)c * 16u : 0xFFFFFFFFu, 0, 0);
        }
    }
}

template <int DT, int PRO, int NS, int RM, int SETS>
__device__ __forceinline__ void prologue_finish(const KgLinArgs& a, const ProRegs<NS>& pr, char* xs, int xs_stride, float* red2, int M,
                                                int tid, int nthreads, int lane, int wave, int nwaves) {
    const int nchunk = a.K >> 3;
    float ss[RM];
#pragma unroll
    for (int m = 0; m < RM; ++m) ss[m] = 0.f;
    const bool upper = NS > 8 && M * SETS > 8;             // slots 8.. were requested (see prologue_load)
    if (PRO == PRO_RMSNORM) {
        auto add_slot = [&](int i) {
            const int m = i / SETS, c = tid + (i % SETS) * nthreads;
            if (i / SETS < RM) {
                if (m < M && c < nchunk) ss[i / SETS < RM ? i / SETS : 0] += sum_sq8<DT>(pr.v[i]);
            }
        };
#pragma unroll
        for (int i = 0; i < (NS < 8 ? NS : 8); ++i) add_slot(i);
        if (upper) {
#pragma unroll
            for (int i = 8; i < NS; ++i) add_slot(i);
        }
        rms_row_factors<RM>(ss, red2, M, a.K, a.eps, lane, wave, nwaves);
    }
    auto store_slot = [&](int i) {
        const int m = i / SETS, c = tid + (i % SETS) * nthreads;
        if (i / SETS < RM) {
            if (m < M && c < nchunk)
                *(kg_u4*)(xs + (size_t)m * xs_stride + (size_t)c * 16) =
                    (PRO == PRO_NONE) ? pr.v[i] : rms_apply<DT>(pr.v[i], pr.gw[i % SETS], ss[i / SETS < RM ? i / SETS : 0]);
        }
    };
#pragma unroll
    for (int i = 0; i < (NS < 8 ? NS : 8); ++i) store_slot(i);
    if (upper) {
#pragma unroll
        for (int i = 8; i < NS; ++i) store_slot(i);
    }
}

// The whole linear pass of one block.  PERSIST = false: body of a kernel of its own
// (blockDim = nwaves * 64).  PERSIST = true: one phase of a persistent pass kernel whose
// blocks stay resident: the first weight batch is requested BEFORE waiting for the previous
// phase (weights never depend on it), inputs written by other blocks are read and outputs
// are written device-coherently, waves beyond `nwaves` only keep the block barriers company,
// and the phase ends by signalling the grid barrier.
template <int DT, int MODE, int PRO, int EPI, bool PERSIST>
__device__ __forceinline__ void kg_linear_body(const KgLinArgs& a, char* smem, KgGridBar* gb, int nwaves) {
    typedef KgT<DT> T;
    KG_TSTAMP(a, 0);
#ifdef KG_LIN_TLOG
    if (threadIdx.x == blockDim.x - 64) kg_lin_tlog[((size_t)a.logits_stride * gridDim.x + blockIdx.x) * 8 + 7] = wall_clock64();     // entry of the block's last wave
#endif
    constexpr bool COH = PERSIST;                       // outputs: written through to memory
    constexpr bool COH_LD = PERSIST && KG_PERSIST_COH_LOADS; // inputs written by other blocks earlier in this kernel
    const KgDyn* dyn = a.dyn;
    // wave index through readfirstlane: everything derived from it (k-range, loop counters,
    // row-group cursor) is then provably wave-uniform and the loop's control flow is scalar
    int tid_ = threadIdx.x;
    // In a persistent kernel every phase re-derives its per-thread constants from an opaque
    // copy of the thread index; otherwise they are all hoisted to the kernel's entry and
    // stay live (and spill) across every other phase.
    if (PERSIST) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one round trip for all per-pass parameters: switch, row count, positions (lane l holds pos[l])
    constexpr int KG_U = (MODE == MODE_MFMA) ? KG_U_MFMA : KG_U_VEC;
    constexpr bool EARLY = PERSIST;
    const bool idle = PERSIST && wave >= nwaves;     // block-uniform per wave
    const int G = (a.N + 15) >> 4, KT = a.K >> 5;
    const int t0 = (KT * wave) / nwaves, t1 = (KT * (wave + 1)) / nwaves;
    // Weight tiles come through buffer loads: every load instruction is unconditional (so the
    // compiler's counted waits stay exact and nothing branches around a load), and a tile
    // past the wave's k-range or past the block's last row group gets an out-of-range
    // offset, which the hardware range check answers with zeros without touching memory.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)((size_t)G * KT * 1024), 0x00020000);
    auto issue_w = [&](kg_u4 (&wv)[KG_U], int lrg, int lt) {
        const unsigned base = (unsigned)lrg * (unsigned)KT * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int u = 0; u < KG_U; ++u) {
            const bool ok = (lrg < G) && (lt + u < t1);
            const unsigned off = ok ? base + (unsigned)(lt + u) * 1024u : 0xFFFFFFFFu;
            wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 2);
        }
    };
    kg_u4 bufA[KG_U], bufB[KG_U];
    kg_u4 bufAx[KG_U], bufBx[KG_U];
    if (EARLY && !idle) {
        // both batches of the first row group(s): enough bytes in flight to cover the barrier
        issue_w(bufA, blockIdx.x, t0);
        int elt = t0 + KG_U, elrg = blockIdx.x;
        if (elt >= t1) { elt = t0; elrg += gridDim.x; }
        issue_w(bufB, elrg, elt);
    }
    if (PERSIST) gb->wait();
    const int posv = (EPI == EPI_ROPE_KV) ? dyn->pos[lane & (KG_MAX_ROWS - 1)] : 0;
    const int active = dyn->active;
    const int n_rows = dyn->n_rows;
    if (!PERSIST && active == 0) return;
    KG_TSTAMP(a, 1);
    const int M = (MODE == MODE_VEC) ? 1 : n_rows;
    constexpr int RM = (MODE == MODE_VEC) ? 1 : KG_MAX_ROWS;
    constexpr int GX_ROW = 8 * 64 + 16;                           // bytes of one X row of a batch in a wave's LDS slot (+ pad)
    static_assert(MODE != MODE_MFMA_GX || KG_U == 8, "the X slot layout assumes 8 tiles per batch");
    const int PART = (MODE == MODE_VEC) ? 16 : 256;            // floats per wave partial
    char* side = smem;                                           // epilogue inputs staged once (residual rows / RoPE rows)
    float* red2 = (float*)(smem + a.side_bytes - 512);           // [rows][nwaves] partial sums of squares (last 512 B of the side region)
    float* part = (float*)(smem + a.side_bytes);                 // [2][nwaves][PART]
    char* xs = smem + a.side_bytes + 2 * nwaves * PART * 4;
    const int my_pos = (EPI == EPI_ROPE_KV) ? __shfl(posv, (tid >> 3) & (KG_MAX_ROWS - 1), 64) : 0;
    const int xs_stride = a.K * 2 + 32;
    const int nthreads = nwaves * 64;

    const int tlast = max(t1 - 1, 0);
    int bid_ = blockIdx.x;
    if (PERSIST) asm volatile("" : "+s"(bid_));
    const int bid = bid_;
    const int nrg_blk = (G - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    if (idle) {
        // same barrier sequence as the working waves: the norm's (if any), one after the prologue, one per row group
        if (PRO == PRO_RMSNORM && MODE == MODE_VEC) __syncthreads();
        if (MODE == MODE_MFMA && PRO == PRO_RMSNORM && a.K * 2 <= 2 * nwaves * PART * 4) __syncthreads();     // gamma handed round through LDS
        __syncthreads();
        for (int i = 0; i < nrg_blk; ++i) __syncthreads();
        gb->arrive();
        return;
    }
    const int g = lane >> 4, r = lane & 15;

    // B-operand / x source for this lane
    const int mrow = (r < M) ? r : 0;
    const char* xl = xs + (size_t)((MODE == MODE_VEC) ? 0 : mrow) * xs_stride + g * 16;

    // ---- small loads first: X rows and what the epilogues need (residual rows, RoPE rows).
    // A global load inside the streaming loop would have to be waited for with vmcnt(0),
    // i.e. drain every weight tile in flight, so everything is fetched here and parked in LDS.
    const int row_off = (MODE == MODE_VEC && a.x_last_row) ? n_rows - 1 : 0;
    // X rows in registers while the first weights are requested: one 16-byte chunk of every row per thread,
    // or (K beyond 8 * threads) two chunks of up to half as many rows
    constexpr int NS = (MODE == MODE_VEC) ? 2 : KG_MAX_ROWS;
    const int pro_sets = ((a.K >> 3) <= nthreads) ? 1 : 2;
    const bool pro_fast = (a.K >> 3) <= pro_sets * nthreads && M * pro_sets <= NS;
    ProRegs<NS> pr;
    // M-row passes with an RMSNorm: wave w owns rows w, w + nwaves, ... entirely (lane l holds the 16-byte chunks
    // l, l + 64, ... of the row and of gamma), so a row's sum of squares is ONE wave reduction, the row factor
    // never leaves the wave, and no barrier or LDS round trip sits between the loads and the normalised row.
    // A row's arithmetic does not depend on how many rows ride along or on which wave takes it.
    constexpr bool ROWPRO = (MODE == MODE_MFMA && PRO == PRO_RMSNORM);
    constexpr int NCM = ROWPRO ? 16 : 1;                       // chunks of a row per lane: K <= 8192
    kg_u4 rx[NCM], rgm[NCM];
    const int nchunk_ = a.K >> 3;
    const __amdgpu_buffer_rsrc_t xrow_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0,
        ROWPRO ? (int)(((size_t)(M - 1) * a.x_stride + a.K) * 2) : 0, 0x00020000);
    // Chunks 0..7 of a row (K <= 4096) are always requested, chunks 8..11 and 12..15 only when the row has them: an
    // out-of-range load moves no data but occupies the CU's address path like a real one (gate/up at K = 4096:
    // 0.9 us per launch).  The conditions are block-uniform.
    const bool rp_t1 = ROWPRO && nchunk_ > 512, rp_t2 = ROWPRO && nchunk_ > 768;
    auto rp_tiers = [&](auto&& f) {
#pragma unroll
        for (int i = 0; i < (NCM < 8 ? NCM : 8); ++i) f(i);
        if (rp_t1) {
#pragma unroll
            for (int i = 8; i < (NCM < 12 ? NCM : 12); ++i) f(i);
        }
        if (rp_t2) {
#pragma unroll
            for (int i = 12; i < NCM; ++i) f(i);
        }
    };
    auto rowpro_load = [&](int row) {
        rp_tiers([&](int i) {
            const int c = lane + 64 * i;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xrow_rsrc, (row < M && c < nchunk_) ? (unsigned)(((size_t)row * a.x_stride) * 2 + (size_t)c * 16) : 0xFFFFFFFFu,
                                                          0, COH_LD ? KG_AUX_COH : 0);
        });
    };
    // gamma: every wave with a row needs all of it.  When it fits the (still unused) cross-wave reduction area the
    // block fetches it ONCE, 16 bytes per thread, and hands it round through LDS: per block 8 wave-level loads
    // instead of 8 per row, and the CU's address path, which the burst of requests at kernel entry saturates, is
    // free that much earlier.  Waves without a row request nothing at all.
    const bool g_lds = ROWPRO && a.K * 2 <= 2 * nwaves * PART * 4;          // block-uniform
    kg_u4 gq[2];
    if (ROWPRO) {
        const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc((void*)a.norm_w, 0, a.K * 2, 0x00020000);
        if (wave < M) rowpro_load(wave);
        if (g_lds) {
            gq[0] = __builtin_amdgcn_raw_buffer_load_b128(nr, (tid < nchunk_) ? (unsigned)tid * 16u : 0xFFFFFFFFu, 0, 0);
            if (nchunk_ > nthreads) gq[1] = __builtin_amdgcn_raw_buffer_load_b128(nr, (tid + nthreads < nchunk_) ? (unsigned)(tid + nthreads) * 16u : 0xFFFFFFFFu, 0, 0);
        } else if (wave < M) {
            rp_tiers([&](int i) {
                const int c = lane + 64 * i;
                rgm[i] = __builtin_amdgcn_raw_buffer_load_b128(nr, (c < nchunk_) ? (unsigned)c * 16u : 0xFFFFFFFFu, 0, 0);
            });
        }
    } else if (MODE != MODE_MFMA_GX && pro_fast) {
        if (pro_sets == 1) prologue_load<PRO, NS, 1, COH_LD>(a, pr, M, row_off, tid, nthreads);
        else prologue_load<PRO, NS, 2, COH_LD>(a, pr, M, row_off, tid, nthreads);
    }
    unsigned side_raw[2] = {0u, 0u};
    const int side_n = (EPI == EPI_RESID) ? nrg_blk * M * 16 : (EPI == EPI_ROPE_KV) ? M * a.head_dim : 0;
    const bool side_fast = side_n <= 2 * nthreads;
    if (EPI == EPI_RESID) {
        if (side_fast) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0,
                (int)(((size_t)(M - 1) * a.resid_stride + a.N) * 2), 0x00020000);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = tid + e * nthreads;
                const int row = i & 15, m = (i >> 4) % M, itx = (i >> 4) / M;
                const int n = ((int)blockIdx.x + itx * (int)gridDim.x) * 16 + row;
                const bool ok = (i < side_n) && (n < a.N);
                side_raw[e] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, ok ? (unsigned)(((size_t)m * a.resid_stride + n) * 2) : 0xFFFFFFFFu, 0, COH_LD ? KG_AUX_COH : 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0,
                (int)(((size_t)(M - 1) * a.resid_stride + a.N) * 2), 0x00020000);
            unsigned short* sd = (unsigned short*)side;
            for (int i = tid; i < side_n; i += nthreads) {
                const int row = i & 15, m = (i >> 4) % M, itx = (i >> 4) / M;
                const int n = ((int)blockIdx.x + itx * (int)gridDim.x) * 16 + row;
                sd[i] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, (n < a.N) ? (unsigned)(((size_t)m * a.resid_stride + n) * 2) : 0xFFFFFFFFu, 0, COH_LD ? KG_AUX_COH : 0);
            }
        }
    } else if (EPI == EPI_ROPE_KV) {
        const int half = a.head_dim >> 1;
        if (side_fast) {
            const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)a.rope_cos, 0, 0x7FFFFFF0, 0x00020000);
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.rope_sin, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = tid + e * nthreads;
                const int m = i / a.head_dim, jj = i - m * a.head_dim;
                const int p = __shfl(posv, m & (KG_MAX_ROWS - 1), 64);
                const bool ok = i < side_n;
                const unsigned off = (unsigned)(((size_t)p * half + (jj < half ? jj : jj - half)) * 4);
                const unsigned c = __builtin_amdgcn_raw_buffer_load_b32(cr, (ok && jj < half) ? off : 0xFFFFFFFFu, 0, 0);
                const unsigned sn = __builtin_amdgcn_raw_buffer_load_b32(sr, (ok && jj >= half) ? off : 0xFFFFFFFFu, 0, 0);
                side_raw[e] = c | sn;
            }
        } else {
            float* sd = (float*)side;
            for (int i = tid; i < side_n; i += nthreads) {
                const int m = i / a.head_dim, jj = i - m * a.head_dim;
                const int p = dyn->pos[m];
                sd[i] = (jj < half) ? a.rope_cos[(size_t)p * half + jj] : a.rope_sin[(size_t)p * half + jj - half];
            }
        }
    }

    int rg = blockIdx.x;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0,
        (MODE == MODE_MFMA_GX) ? (int)(((size_t)(M - 1) * a.x_stride + a.K) * 2) : 0, 0x00020000);
    // MODE_MFMA_GX (X too large for LDS as a whole): each wave moves the X columns of its next
    // batch (rows x 8 tiles x 64 B, row-contiguous 512-byte pieces) through a wave-private LDS
    // slot.  The X loads are issued ahead of the batch's weight tiles, so waiting for them
    // never waits for those weights (the counter retires in order).
    auto issue_x = [&](kg_u4 (&xv)[KG_U], int lrg, int lt) {
        if (MODE == MODE_MFMA_GX) {
            auto load_piece = [&](int j) {
                const int flat = j * 64 + lane, row = flat >> 5, ch = flat & 31;
                const bool okx = (lrg < G) && (row < M) && (lt + (ch >> 2) < t1);
                const unsigned xo = okx ? (unsigned)(((size_t)row * a.x_stride) * 2) + (unsigned)lt * 64u + (unsigned)ch * 16u : 0xFFFFFFFFu;
                xv[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo, 0, COH_LD ? KG_AUX_COH : 0);
            };
            // load j covers rows 2j and 2j + 1: with 8-row slots the upper half would request nothing, and a request
            // out of range still takes its turn in the address path (the branch is block-uniform)
#pragma unroll
            for (int j = 0; j < KG_U / 2; ++j) load_piece(j);
            if (a.gx_slot_rows > 8) {
#pragma unroll
                for (int j = KG_U / 2; j < KG_U; ++j) load_piece(j);
            }
        }
    };
    auto issue = [&](kg_u4 (&wv)[KG_U], kg_u4 (&xv)[KG_U], int lrg, int lt) {
        issue_x(xv, lrg, lt);
        issue_w(wv, lrg, lt);
    };

    // two batches of weight tiles in flight while the prologue finishes
    int lrg = rg, lt = t0;
    if (EARLY) issue_x(bufAx, lrg, lt);
    else issue(bufA, bufAx, lrg, lt);
    lt += KG_U;
    if (lt >= t1) { lt = t0; lrg += gridDim.x; }
    // M-row passes with a norm request their second batch only after the rows are in LDS.  The rows come
    // from the memory side (another XCD wrote them) and land ~3.5 us after the request while every CU floods the
    // same queues with weight requests; with both batches requested first, a wave reached its wait for the rows
    // only after 2.5 us of issuing and the block's barrier fell 6 us after entry.  One batch ahead, the rows are
    // in LDS after 4.3 us and the second batch is in flight while the first is consumed (gate/up -0.6 us per
    // launch; the other prologues and the 1-row kernels measured the same or slower this way).
    constexpr bool DEFER_B = ROWPRO && !EARLY;
    if (!DEFER_B) {
        if (EARLY) issue_x(bufBx, lrg, lt);
        else issue(bufB, bufBx, lrg, lt);
        lt += KG_U;
        if (lt >= t1) { lt = t0; lrg += gridDim.x; }
    }

    KG_TSTAMP(a, 5);
    if (side_fast) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = tid + e * nthreads;
            if (i < side_n) {
                if (EPI == EPI_RESID) ((unsigned short*)side)[i] = (unsigned short)side_raw[e];
                if (EPI == EPI_ROPE_KV) ((unsigned*)side)[i] = side_raw[e];
            }
        }
    }
    if (ROWPRO) {
        char* gam = (char*)part;                                // free until the first row group is reduced
        if (g_lds) {
            if (tid < nchunk_) *(kg_u4*)(gam + (size_t)tid * 16) = gq[0];
            if (nchunk_ > nthreads && tid + nthreads < nchunk_) *(kg_u4*)(gam + (size_t)(tid + nthreads) * 16) = gq[1];
        }
        auto row_factor = [&]() {
            float ssq = 0.f;
            rp_tiers([&](int i) { ssq += T::sum_sq8(rx[i]); });          // requested chunks out of range came back as zeros
            return rsqrtf(kg_wave_sum_dpp(ssq) / (float)a.K + a.eps);
        };
        float rstd = 0.f;
        if (wave < M) rstd = row_factor();
        if (g_lds) __syncthreads();                             // gamma is in LDS
        for (int row = wave; row < M; row += nwaves) {          // wave-uniform trip count
            if (row != wave) { rowpro_load(row); rstd = row_factor(); }
            rp_tiers([&](int i) {
                const int c = lane + 64 * i;
                if (c < nchunk_) {
                    const kg_u4 gv = g_lds ? *(const kg_u4*)(gam + (size_t)c * 16) : rgm[i];
                    *(kg_u4*)(xs + (size_t)row * xs_stride + (size_t)c * 16) = T::rms_apply(rx[i], gv, rstd);
                }
            });
        }
    } else if (MODE != MODE_MFMA_GX) {
        if (!pro_fast) prologue_slow<DT, PRO, RM, COH_LD>(a, xs, xs_stride, red2, M, row_off, tid, nthreads, lane, wave, nwaves);
        else if (pro_sets == 1) prologue_finish<DT, PRO, NS, RM, 1>(a, pr, xs, xs_stride, red2, M, tid, nthreads, lane, wave, nwaves);
        else prologue_finish<DT, PRO, NS, RM, 2>(a, pr, xs, xs_stride, red2, M, tid, nthreads, lane, wave, nwaves);
    }
    if (DEFER_B) {
        issue(bufB, bufBx, lrg, lt);
        lt += KG_U;
        if (lt >= t1) { lt = t0; lrg += gridDim.x; }
    }
    KG_TSTAMP(a, 6);
    __syncthreads();
    KG_TSTAMP(a, 2);

    HeadRun run; run.mx = -INFINITY; run.sum = 0.f; run.idx = 0;
    int it = 0, ct = t0;
    kg_f4 acc = {0.f, 0.f, 0.f, 0.f};
    float acc0 = 0.f, acc1 = 0.f;

    // One step: consume the batch held in (wv, xv), refill the same registers with the batch
    // two ahead (the two buffers alternate, so no register copies and up to 2*KG_U tiles stay
    // in flight per wave), and at the end of a row group reduce across waves and run the
    // epilogue.
    auto step = [&](kg_u4 (&wv)[KG_U], kg_u4 (&xv)[KG_U], int which, bool refill) {
        char* xslot = xs + (size_t)(wave * 2 + which) * a.gx_slot_rows * GX_ROW;      // MODE_MFMA_GX only
        if (MODE == MODE_MFMA_GX) {
#pragma unroll
            for (int j = 0; j < KG_U / 2; ++j) {
                const int flat = j * 64 + lane, row = flat >> 5, ch = flat & 31;
                *(kg_u4*)(xslot + row * GX_ROW + ch * 16) = xv[j];
            }
            if (a.gx_slot_rows > 8) {
#pragma unroll
                for (int j = KG_U / 2; j < KG_U; ++j) {
                    const int flat = j * 64 + lane, row = flat >> 5, ch = flat & 31;
                    *(kg_u4*)(xslot + row * GX_ROW + ch * 16) = xv[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KG_U; ++u) {
            const kg_u4 w = wv[u];                    // zeros when the tile was out of range
            const int tt = min(ct + u, tlast);
            if (MODE == MODE_VEC) {
                kg_u4 x = *(const kg_u4*)(xl + (size_t)tt * 64);
                acc0 = T::dot2(w.x, x.x, acc0);
                acc1 = T::dot2(w.y, x.y, acc1);
                acc0 = T::dot2(w.z, x.z, acc0);
                acc1 = T::dot2(w.w, x.w, acc1);
            } else if (MODE == MODE_MFMA) {
                kg_u4 x = *(const kg_u4*)(xl + (size_t)tt * 64);
                acc = T::mfma(w, x, acc);
            } else {
                kg_u4 x = *(const kg_u4*)(xslot + mrow * GX_ROW + u * 64 + g * 16);
                acc = T::mfma(w, x, acc);
            }
        }
        if (refill) {
            issue(wv, xv, lrg, lt);
            lt += KG_U;
            if (lt >= t1) { lt = t0; lrg += gridDim.x; }
        }
        ct += KG_U;
        if (ct >= t1 && rg < G) {
            // ---- cross-wave reduction through LDS (slot alternates per row group) ----
            float* slot = part + (size_t)(it & 1) * nwaves * PART;
            if (MODE == MODE_VEC) {
                float v = acc0 + acc1;
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lane < 16) slot[wave * 16 + lane] = v;
            } else {
                // lane holds D[row 4g+i][col r]; store as [col][row] so a pair reads 2 floats
                *(kg_f4*)(slot + wave * 256 + r * 16 + g * 4) = acc;
            }
            __syncthreads();
            if (it == 0) KG_TSTAMP(a, 3);
            const int n_epi = (MODE == MODE_VEC) ? 8 : 128;
            if (tid < n_epi) {
                const int m = tid >> 3, p = tid & 7;
                if (m < M) {
                    float v0 = 0.f, v1 = 0.f;
                    for (int w = 0; w < nwaves; ++w) {
                        const float* sp = slot + w * PART + m * 16 + 2 * p;
                        v0 += sp[0]; v1 += sp[1];
                    }
                    epilogue_pair<DT, EPI, COH>(a, rg * 16 + 2 * p, m, my_pos, v0, v1, run, side, it, M);
                }
            }
            rg += gridDim.x; ct = t0; ++it;
            acc = kg_f4{0.f, 0.f, 0.f, 0.f}; acc0 = 0.f; acc1 = 0.f;
        }
    };
    // A wave consumes nsteps batches (one per KG_U tiles of its k range and row group; a wave without a k range
    // still takes one step per row group for the block barriers).  Step k refills its buffer with batch k + 2.
    // The main loop runs the pairs of steps whose refills both exist: a single back edge, two batches in flight,
    // counted waits.  The last (up to three) steps stand outside it and request nothing: an out-of-range request
    // moves no data but still occupies the CU's address path, which for a projection of two batches per wave
    // (o_proj) sat between the two consumes (measured 0.45 us per launch).
    const int nsteps = nrg_blk * max(1, (t1 - t0 + KG_U - 1) / KG_U);
    int ks = 0;
    while (ks + 3 < nsteps) {
        step(bufA, bufAx, 0, true);
        step(bufB, bufBx, 1, true);
        ks += 2;
    }
    if (ks < nsteps) { step(bufA, bufAx, 0, ks + 2 < nsteps); ++ks; }
    if (ks < nsteps) { step(bufB, bufBx, 1, false); ++ks; }
    if (ks < nsteps) { step(bufA, bufAx, 0, false); ++ks; }
    if (EPI == EPI_HEAD) {
        // merge the 8 pair-threads of each column, then one record per (block, column)
        const int n_epi = (MODE == MODE_VEC) ? 8 : 128;
        if (tid < 128) {      // whole waves take part in the shuffles
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                float bmx = __shfl_xor(run.mx, o, 64);
                float bs = __shfl_xor(run.sum, o, 64);
                int bi = __shfl_xor(run.idx, o, 64);
                head_merge(run, bmx, bs, bi);
            }
            const int m = tid >> 3;
            if (tid < n_epi && (tid & 7) == 0 && m < M && (int)blockIdx.x < G) {     // blocks without a row group have no record
                float* hp = a.head_part + ((size_t)m * a.head_part_stride + blockIdx.x) * 4;
                hp[0] = run.mx; hp[1] = run.sum; hp[2] = __int_as_float(run.idx); hp[3] = 0.f;
            }
        }
    }
    KG_TSTAMP(a, 4);
    if (PERSIST) gb->arrive();
}
