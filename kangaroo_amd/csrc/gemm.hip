// Prefill linears (gfx950): Y[m][n] = sum_k X[m][k] W[n][k] for up to 128 rows (tokens) per pass.
//
// Decode streams every weight once per token; a prompt of T0 tokens run through the M<=16
// decode kernels streams them T0/16 times.  Here a pass carries 128 rows, so the weights are
// read T0/128 times.
//
// Same packed weight tiles as the decode kernels.  A wave owns one or two adjacent 16-row groups of W and
// walks all of K: per k-batch of 4 tiles it holds their A fragments in registers (a ring of batches, all
// but one in flight) and multiplies them with eight 16-row column blocks of X^T from LDS; with two row
// groups every X^T fragment read from LDS feeds both.  X^T is
// staged batch by batch (128 rows x 128 k, double-buffered, XOR-swizzled) by four extra LOADER waves
// of the block (rows that need an RMSNorm are normalised by kg_rmsnorm_kernel beforehand, once per pass
// instead of once per block).  The split matters because a wave's loads retire in order: a wave that both
// streams weights several batches ahead and waits for the next X batch drains its own weight
// queue at every step (measured: half of a block's time went into that wait).  The accumulators
// go through the decode kernels' own epilogue code, so rounding is identical.
#include "kg_linear_body.h"
#include <cstdlib>

#ifndef GEMM_KB
#define GEMM_KB 4                   // weight tiles (32 k each) per k-batch
#endif
#define GEMM_LW 4                   // loader waves per block (stage X; the other GEMM_NW waves stream weights and run the MFMAs)
#define GEMM_XROW (GEMM_KB * 64)    // bytes of one X row of a k-batch in LDS: exactly one 256-byte bank row
// 16-byte piece p of row `row` sits at piece slot p ^ (row & 15): the 16 lanes ds_read_b128 serves
// together read 16 different rows at (nearly) one column, which the XOR spreads over all 16 slots.
#define GEMM_XOFF(row, p) ((row) * GEMM_XROW + ((((p) ^ (row)) & 15) << 4))
static_assert(GEMM_KB == 4, "the LDS swizzle assumes 256-byte rows");

// GEMM_NW = waves (= weight row groups) per block: 4 shares one staged X batch among more
// row groups; 2 keeps 128 blocks for the 4096-row projections.
// GEMM_RPW = weight row groups per compute wave: 2 halves the LDS reads per MFMA; 1 leaves registers for a
// deeper ring of weight batches (more bytes in flight per wave), which is what counts when the row groups
// are too few to fill the CUs with two-group waves.
template <int DT, int EPI, int GEMM_NW, int GEMM_RPW>
__global__ void __launch_bounds__((GEMM_NW + GEMM_LW) * 64) kg_gemm_kernel(const KgLinArgs a) {
    constexpr int GEMM_STAGE = KG_GEMM_ROWS * (GEMM_KB * 4) / (GEMM_LW * 64);    // 16-byte pieces of a batch each loader thread stages
    constexpr int PPR = GEMM_KB * 4;                                   // 16-byte pieces per X row of a batch
    constexpr int RPJ = GEMM_LW * 64 / PPR;                            // rows one staging step covers
    typedef KgT<DT> T;
    KG_TSTAMP(a, 0);
    const KgDyn* dyn = a.dyn;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int pos0 = dyn->pos[0];
    const int M = dyn->n_rows;
    if (dyn->active == 0) return;
    const int G = (a.N + 15) >> 4, KT = a.K >> 5, NB = (KT + GEMM_KB - 1) / GEMM_KB;
    const bool loader = wave >= GEMM_NW;               // wave-uniform role
    const int ltid = tid - GEMM_NW * 64;               // loader thread index (negative in compute waves)
    const int rg0 = (blockIdx.x * GEMM_NW + (loader ? 0 : wave)) * GEMM_RPW;     // >= G in the last block: loads out of range, nothing stored
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs0 = smem;
    char* xs1 = smem + KG_GEMM_ROWS * GEMM_XROW;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)((size_t)G * KT * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(((size_t)(M - 1) * a.x_stride + a.K) * 2), 0x00020000);

    // Blocks walk K in rotated order (block i starts at batch 5 i mod NB): otherwise every block
    // asks for the same X columns at the same moment and those few L2 channels serialise the grid.
    // K ranges on grid.y (ksplit > 1): this block walks batches [kb0, kb0 + NBL)
    const int KS = a.ksplit > 1 ? a.ksplit : 1;
    const int kb0 = (int)(((long long)NB * blockIdx.y) / KS), NBL = (int)(((long long)NB * (blockIdx.y + 1)) / KS) - kb0;
    const int bstart = NBL > 0 ? (int)((blockIdx.x * 5u) % (unsigned)NBL) : 0;
    auto kbatch = [&](int step) { int b = step + bstart; if (b >= NBL) b -= NBL; return (step < NBL) ? kb0 + b : NB; };   // NB: out of range
    auto issue_w = [&](kg_u4 (&wv)[GEMM_RPW][GEMM_KB], int step) {
        const int b = kbatch(step);
#pragma unroll
        for (int j = 0; j < GEMM_RPW; ++j) {
            const unsigned base = (unsigned)(rg0 + j) * (unsigned)KT * 1024u + (unsigned)lane * 16u;
#pragma unroll
            for (int u = 0; u < GEMM_KB; ++u) {
                const bool ok = (rg0 + j < G) && (GEMM_KB * b + u < KT);
                wv[j][u] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, ok ? base + (unsigned)(GEMM_KB * b + u) * 1024u : 0xFFFFFFFFu, 0, 2);
            }
        }
    };
    // staging: piece q = tid + nthreads * j covers row q / PPR, 16-byte column q % PPR of the batch
    const int scol = (ltid & (PPR - 1)), srow0 = (ltid >= 0 ? ltid : 0) / PPR;
    auto stage_load = [&](kg_u4 (&sx)[GEMM_STAGE], int step) {
        const int b = kbatch(step);
        const bool kok = (GEMM_KB * b + (scol >> 2)) < KT;
#pragma unroll
        for (int j = 0; j < GEMM_STAGE; ++j) {
            const int row = srow0 + RPJ * j;
            const bool ok = kok && row < M;
            sx[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(((size_t)row * a.x_stride + 32 * GEMM_KB * b + 8 * scol) * 2) : 0xFFFFFFFFu, 0, 0);
        }
    };
    auto stage_store = [&](const kg_u4 (&sx)[GEMM_STAGE], char* xs) {
#pragma unroll
        for (int j = 0; j < GEMM_STAGE; ++j) *(kg_u4*)(xs + GEMM_XOFF(srow0 + RPJ * j, scol)) = sx[j];
    };

    kg_f4 acc[GEMM_RPW][KG_GEMM_ROWS / 16];
#pragma unroll
    for (int j = 0; j < GEMM_RPW; ++j)
#pragma unroll
        for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) acc[j][cb] = kg_f4{0.f, 0.f, 0.f, 0.f};
    // All column blocks always run (rows past M are zeros in LDS): a branch per block would
    // put every LDS read right in front of its MFMA with a full wait in between.
    auto load_x = [&](kg_u4 (&x)[KG_GEMM_ROWS / 16], const char* xs, int u) {
#pragma unroll
        for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) x[cb] = *(const kg_u4*)(xs + GEMM_XOFF(16 * cb + r, 4 * u + g));
    };
    auto compute = [&](const kg_u4 (&wv)[GEMM_RPW][GEMM_KB], const char* xs) {
        if (GEMM_RPW == 1) {
            // one row group: every fragment feeds one MFMA, so all of a batch's fragments are requested at once
            kg_u4 x[GEMM_KB][KG_GEMM_ROWS / 16];
#pragma unroll
            for (int u = 0; u < GEMM_KB; ++u) load_x(x[u], xs, u);
#pragma unroll
            for (int u = 0; u < GEMM_KB; ++u)
#pragma unroll
                for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) acc[0][cb] = T::mfma(wv[0][u], x[u][cb], acc[0][cb]);
        } else {
            // two row groups: the fragments of two k tiles at a time (registers), each feeding two MFMAs
#pragma unroll
            for (int h = 0; h < GEMM_KB / 2; ++h) {
                kg_u4 x[2][KG_GEMM_ROWS / 16];
                load_x(x[0], xs, 2 * h);
                load_x(x[1], xs, 2 * h + 1);
#pragma unroll
                for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                    for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb)
#pragma unroll
                        for (int j = 0; j < GEMM_RPW; ++j) acc[j][cb] = T::mfma(wv[j][2 * h + uu], x[uu][cb], acc[j][cb]);
            }
        }
    };

    // Software pipeline, one block barrier per k-batch.  Compute waves: ring of D register batches of
    // weights, all but one in flight.  Loader waves: X(b) sits in LDS buffer b & 1; X(b+1) .. X(b+XS)
    // are on their way in XS register sets; during step b the set holding X(b+1) goes to the other
    // LDS buffer and is re-requested for X(b+1+XS).
    constexpr int D = (GEMM_RPW == 2) ? 3 : 4;
    constexpr int XS = (GEMM_RPW == 2) ? 3 : 4;
    constexpr int U = (GEMM_RPW == 2) ? 6 : 4;          // steps per unrolled round: a multiple of D, XS and 2
    static_assert(U % D == 0 && U % XS == 0 && U % 2 == 0, "ring indices must repeat every unrolled round");
#ifdef KG_LIN_TLOG
    unsigned long long tl_issue = 0, tl_compute = 0, tl_store = 0, tl_barrier = 0;
#define GEMM_TL(var, t0_) do { const unsigned long long t1_ = wall_clock64(); var += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define GEMM_TL(var, t0_) do { } while (0)
#endif
    if (loader) {
        kg_u4 sx[XS][GEMM_STAGE];
        // RoPE epilogue: the cos / sin rows of the pass's positions go to their own LDS region now, in the shadow
        // of the first X round trip (requested first, so waiting for them never waits for X)
        constexpr int RS = (EPI == EPI_ROPE_KV) ? 16 : 1;       // 16-byte pieces per loader thread: rows x head_dim <= 128 x 128 floats
        kg_u4 rsv[RS];
        if (EPI == EPI_ROPE_KV) {
            const int half = a.head_dim >> 1, ppr = a.head_dim >> 2;          // pieces per row
#pragma unroll
            for (int j = 0; j < RS; ++j) {
                const int q = ltid + j * (GEMM_LW * 64), m = q / ppr, c4 = (q - m * ppr) * 4;
                rsv[j] = kg_u4{0u, 0u, 0u, 0u};
                if (m < M) {
                    const size_t p = (size_t)(pos0 + m) * half;
                    rsv[j] = *(const kg_u4*)((c4 < half) ? a.rope_cos + p + c4 : a.rope_sin + p + c4 - half);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < XS; ++i) stage_load(sx[i], i);
        if (EPI == EPI_ROPE_KV) {
            const int ppr = a.head_dim >> 2;
            char* rope_side = smem + 2 * KG_GEMM_ROWS * GEMM_XROW;
#pragma unroll
            for (int j = 0; j < RS; ++j) {
                const int q = ltid + j * (GEMM_LW * 64);
                if (q < KG_GEMM_ROWS * ppr) *(kg_u4*)(rope_side + (size_t)q * 16) = rsv[j];
            }
        }
        stage_store(sx[0], xs0);
        stage_load(sx[0], XS);
        __syncthreads();
        for (int b = 0; b < NBL; b += U) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int bb = b + i;
                if (bb >= NBL) break;                   // block-uniform: both roles leave after the same barrier
#ifdef KG_LIN_TLOG
                unsigned long long tl0 = wall_clock64();
#endif
                stage_store(sx[(i + 1) % XS], (bb & 1) ? xs0 : xs1);      // X(bb + 1)
                GEMM_TL(tl_store, tl0);
                stage_load(sx[(i + 1) % XS], bb + 1 + XS);
                GEMM_TL(tl_issue, tl0);
                __syncthreads();
                GEMM_TL(tl_barrier, tl0);
            }
        }
    } else {
        kg_u4 w[D][GEMM_RPW][GEMM_KB];
#pragma unroll
        for (int i = 0; i < D - 1; ++i) issue_w(w[i], i);
        __syncthreads();
        KG_TSTAMP(a, 1);
        for (int b = 0; b < NBL; b += U) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int bb = b + i;
                if (bb >= NBL) break;
#ifdef KG_LIN_TLOG
                unsigned long long tl0 = wall_clock64();
#endif
                issue_w(w[(i + D - 1) % D], bb + D - 1);
                GEMM_TL(tl_issue, tl0);
                compute(w[i % D], (bb & 1) ? xs1 : xs0);
                GEMM_TL(tl_compute, tl0);
                __syncthreads();
                GEMM_TL(tl_barrier, tl0);
            }
        }
    }

#ifdef KG_LIN_TLOG
    if (threadIdx.x == 0) {
        unsigned long long* tl = &kg_lin_tlog[((size_t)a.logits_stride * gridDim.x + blockIdx.x) * 8];
        tl[2] = tl_issue; tl[3] = tl_compute; tl[5] = tl_barrier; tl[6] = wall_clock64();
    }
    if (threadIdx.x == GEMM_NW * 64) {      // first loader thread: its X wait + LDS store time goes to slot 4
        unsigned long long* tl = &kg_lin_tlog[((size_t)a.logits_stride * gridDim.x + blockIdx.x) * 8];
        tl[4] = tl_store;
    }
#endif
    // ---- epilogue inputs through LDS (the X buffers are free now), laid out as the decode epilogues expect ----
    if (EPI == EPI_RESID && (KS > 1 || a.part_only)) {
        // partial sums of this K range; kg_gemm_reduce_kernel adds the ranges and the residual
        if (loader) return;
        float* part = a.kpart + (size_t)blockIdx.y * KG_GEMM_ROWS * a.N;
#pragma unroll
        for (int j = 0; j < GEMM_RPW; ++j) {
            if (rg0 + j >= G) break;
#pragma unroll
            for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) {
                const int m = 16 * cb + r;
                if (m < M) {
                    const int n = (rg0 + j) * 16 + 4 * g;          // N is a multiple of 16 on this path (the launcher checks)
                    *(kg_f4*)(part + (size_t)m * a.N + n) = acc[j][cb];
                }
            }
        }
        return;
    }
    char* side = (EPI == EPI_RESID) ? xs0 + (size_t)(loader ? 0 : wave) * GEMM_RPW * KG_GEMM_ROWS * 32
               : (EPI == EPI_ROPE_KV) ? smem + 2 * KG_GEMM_ROWS * GEMM_XROW : xs0;       // cos / sin rows: staged by the loader waves at entry
    if (EPI == EPI_RESID && !loader) {
        const unsigned short* res = (const unsigned short*)a.resid;
#pragma unroll
        for (int j = 0; j < GEMM_RPW; ++j)
            for (int e = lane; e < M * 16; e += 64) {
                const int m = e >> 4, n = (rg0 + j) * 16 + (e & 15);
                ((unsigned short*)side)[j * KG_GEMM_ROWS * 16 + e] = (n < a.N) ? res[(size_t)m * a.resid_stride + n] : (unsigned short)0;
            }
    }
    __syncthreads();
    if (loader) { KG_TSTAMP(a, 7); return; }
    HeadRun run; run.mx = -INFINITY; run.sum = 0.f; run.idx = 0;
#pragma unroll
    for (int j = 0; j < GEMM_RPW; ++j) {
        const int rg = rg0 + j;
        if (rg >= G) break;
        const char* sj = (EPI == EPI_RESID) ? side + (size_t)j * KG_GEMM_ROWS * 32 : side;
#pragma unroll
        for (int cb = 0; cb < KG_GEMM_ROWS / 16; ++cb) {
            const int m = 16 * cb + r;
            if (m < M) {
                // lane holds D[weight row 4g + i][column m], i = 0..3: two adjacent-row pairs
                epilogue_pair<DT, EPI, false>(a, rg * 16 + 4 * g, m, pos0 + m, acc[j][cb][0], acc[j][cb][1], run, sj, 0, M);
                epilogue_pair<DT, EPI, false>(a, rg * 16 + 4 * g + 2, m, pos0 + m, acc[j][cb][2], acc[j][cb][3], run, sj, 0, M);
            }
        }
    }
    KG_TSTAMP(a, 7);
}

// out[m][n] = round16(resid[m][n] + round16(sum over K ranges of part[s][m][n])): the EPI_RESID epilogue
// of a K-split pass, ranges added in order.
template <int DT>
__global__ void kg_gemm_reduce_kernel(const KgLinArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int M = dyn->n_rows;
    const int m = blockIdx.y;
    if (m >= M) return;
    const unsigned short* res = (const unsigned short*)a.resid + (size_t)m * a.resid_stride;
    unsigned short* out = (unsigned short*)a.out + (size_t)m * a.out_stride;
    // four consecutive columns per thread (N is a multiple of 16 here: the launcher checks)
    const int n = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= a.N) return;
    kg_f4 p[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] = (s < a.ksplit) ? *(const kg_f4*)(a.kpart + ((size_t)s * KG_GEMM_ROWS + m) * a.N + n) : kg_f4{0.f, 0.f, 0.f, 0.f};
    const kg_f4 v = ((p[0] + p[1]) + p[2]) + p[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[n + i] = T::from_f(T::to_f(res[n + i]) + kg_round<DT>(v[i]));
}

// out32[m][n] = sum over K ranges of part[s][m][n], ranges added in order, nothing else: the fp32 partial sums of a
// tensor-parallel shard's row-split projection (a.out is float [rows][out_stride]); the sum over the ranks, the
// rounding and the residual follow in kg_tp_reduce_kernel.
__global__ void kg_gemm_reduce_f32_kernel(const KgLinArgs a) {
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int m = blockIdx.y;
    if (m >= dyn->n_rows) return;
    const int n = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= a.N) return;
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    kg_f4 p[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] = (s < ks) ? *(const kg_f4*)(a.kpart + ((size_t)s * KG_GEMM_ROWS + m) * a.N + n) : kg_f4{0.f, 0.f, 0.f, 0.f};
    *(kg_f4*)((float*)a.out + (size_t)m * a.out_stride + n) = ((p[0] + p[1]) + p[2]) + p[3];
}

// The same sum, one block per row, followed by the RMSNorm of the finished row for the next projection
// (a.out_norm_w / a.out_norm): out as above, out_norm[m][n] = round16(gamma[n] * round16(out[m][n] * rstd)).
// Thread t owns the 8-column pieces t, t + 256, ... in both halves, exactly as kg_rmsnorm_kernel does, so the
// result is bit-identical to kg_gemm_reduce_kernel followed by kg_rmsnorm_kernel.
template <int DT>
__global__ void __launch_bounds__(256) kg_gemm_reduce_norm_kernel(const KgLinArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int m = blockIdx.x;
    if (m >= dyn->n_rows) return;
    __shared__ float red[4];
    const unsigned short* res = (const unsigned short*)a.resid + (size_t)m * a.resid_stride;
    unsigned short* out = (unsigned short*)a.out + (size_t)m * a.out_stride;
    unsigned short* xn = (unsigned short*)a.out_norm + (size_t)m * a.N;
    float ss = 0.f;
    for (int i = threadIdx.x; i < (a.N >> 3); i += 256) {
        const int n = 8 * i;
        kg_f4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        kg_f4 p[4], q[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* pp = a.kpart + ((size_t)s * KG_GEMM_ROWS + m) * a.N + n;
            p[s] = (s < a.ksplit) ? *(const kg_f4*)pp : lo;
            q[s] = (s < a.ksplit) ? *(const kg_f4*)(pp + 4) : hi;
        }
        lo = ((p[0] + p[1]) + p[2]) + p[3];
        hi = ((q[0] + q[1]) + q[2]) + q[3];
        const kg_us8 r8 = __builtin_bit_cast(kg_us8, *(const kg_u4*)(res + n));
        kg_us8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = T::from_f(T::to_f(r8[e]) + kg_round<DT>(e < 4 ? lo[e] : hi[e - 4]));
            const float f = T::to_f(o[e]);
            ss += f * f;
        }
        *(kg_u4*)(out + n) = __builtin_bit_cast(kg_u4, o);
    }
    ss = kg_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float rstd = rsqrtf(ss / (float)a.N + a.eps);
    for (int i = threadIdx.x; i < (a.N >> 3); i += 256) {
        const kg_us8 h = __builtin_bit_cast(kg_us8, *(const kg_u4*)(out + 8 * i));          // this thread's own stores
        const kg_us8 g = __builtin_bit_cast(kg_us8, ((const kg_u4*)a.out_norm_w)[i]);
        kg_us8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = T::from_f(T::to_f(g[e]) * kg_round<DT>(T::to_f(h[e]) * rstd));
        *(kg_u4*)(xn + 8 * i) = __builtin_bit_cast(kg_u4, o);
    }
}

static int g_gemm_ncu = 256;
static void gemm_geometry(int G, int ks, int* nw_out, int* rpw_out);
// blocks on grid.x of a pass over N output rows (tools that index per-block records need it)
int kg_gemm_grid_x(int N, int ksplit) {
    int nw = 2, rpw = 2;
    const int G = (N + 15) / 16;
    gemm_geometry(G, ksplit > 1 ? ksplit : 1, &nw, &rpw);
    return (G + nw * rpw - 1) / (nw * rpw);
}
static size_t gemm_smem(int epi = -1, int head_dim = 128) {
    // two X^T buffers; the RoPE epilogue adds the cos / sin rows of the pass [rows][head_dim] floats
    return (size_t)2 * KG_GEMM_ROWS * GEMM_XROW + ((epi == EPI_ROPE_KV || epi < 0) ? (size_t)KG_GEMM_ROWS * head_dim * 4 : 0);
}

template <int DT, int NW, int RPW>
static int launch_epi(const KgLinArgs& a, int epi, hipStream_t st) {
    const int G = (a.N + 15) / 16;
    const dim3 grid((G + NW * RPW - 1) / (NW * RPW), a.ksplit > 1 ? a.ksplit : 1), block((NW + GEMM_LW) * 64);
    const size_t smem = gemm_smem(epi, a.head_dim);
    switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((kg_gemm_kernel<DT, EPI_STORE, NW, RPW>), grid, block, smem, st, a); break;
    case EPI_RESID: hipLaunchKernelGGL((kg_gemm_kernel<DT, EPI_RESID, NW, RPW>), grid, block, smem, st, a); break;
    case EPI_ROPE_KV: hipLaunchKernelGGL((kg_gemm_kernel<DT, EPI_ROPE_KV, NW, RPW>), grid, block, smem, st, a); break;
    case EPI_SILU_MUL: hipLaunchKernelGGL((kg_gemm_kernel<DT, EPI_SILU_MUL, NW, RPW>), grid, block, smem, st, a); break;
    default: return -1;
    }
    return 0;
}
// Geometry of a pass: compute waves per block (2..4) and row groups per wave (1 or 2).  Measured over the
// 7B and 13B projections at 128 rows: the fastest geometry is always the one that finishes in the fewest
// rounds of blocks and, among those, has the most blocks (every block stages X for itself, but a CU without
// a block streams nothing); at equal block counts more waves per block win.
static void gemm_geometry(int G, int ks, int* nw_out, int* rpw_out) {
    long best_score = -1;
    for (int rpw = 2; rpw >= 1; --rpw)
        for (int nw = 4; nw >= 2; --nw) {
            const int blocks = (G + nw * rpw - 1) / (nw * rpw) * ks;
            const int rounds = (blocks + g_gemm_ncu - 1) / g_gemm_ncu;
            const int resident = blocks < g_gemm_ncu ? blocks : g_gemm_ncu;
            const long score = ((long)(1000 - rounds) << 20) + ((long)resident << 4) + nw;
            if (score > best_score) { best_score = score; *nw_out = nw; *rpw_out = rpw; }
        }
    const char* fn = getenv("KG_GEMM_NW");
    const char* fr = getenv("KG_GEMM_RPW");
    if (fn) *nw_out = atoi(fn);
    if (fr) *rpw_out = atoi(fr);
}
template <int DT>
static int launch_nw(const KgLinArgs& a, int epi, hipStream_t st) {
    int nw = 2, rpw = 2;
    gemm_geometry((a.N + 15) / 16, a.ksplit > 1 ? a.ksplit : 1, &nw, &rpw);
    switch (nw * 10 + rpw) {
    case 42: return launch_epi<DT, 4, 2>(a, epi, st);
    case 32: return launch_epi<DT, 3, 2>(a, epi, st);
    case 22: return launch_epi<DT, 2, 2>(a, epi, st);
    case 41: return launch_epi<DT, 4, 1>(a, epi, st);
    case 31: return launch_epi<DT, 3, 1>(a, epi, st);
    default: return launch_epi<DT, 2, 1>(a, epi, st);
    }
}

// Rows and positions come from a.dyn: n_rows <= KG_GEMM_ROWS rows at positions pos[0], pos[0]+1, ...
int kg_launch_gemm(const KgLinArgs& a_in, int dtype, int pro, int epi, hipStream_t st) {
    KgLinArgs a = a_in;
    if (a.K % 32 != 0 || a.K <= 0 || a.N <= 0) return -1;
    // Few row groups (the d-row projections): one wave per group over all of K leaves most of the
    // machine without a weight request in flight, so K is cut into up to 4 ranges on grid.y.
    {
        const int G = (a.N + 15) / 16, NB = ((a.K >> 5) + GEMM_KB - 1) / GEMM_KB;
        int ks = 1;
        if (epi == EPI_RESID && pro == PRO_NONE && a.kpart && G <= 2 * g_gemm_ncu && a.N % 16 == 0) ks = NB >= 16 ? 4 : (NB >= 8 ? 2 : 1);
        a.ksplit = ks;
    }
    if (a.part_only && (epi != EPI_RESID || !a.kpart || a.N % 16 || (a.out_stride & 3))) return -1;
    if (epi == EPI_ROPE_KV && (a.head_dim > 128 || (a.head_dim & 7) || !a.rope_cos || !a.rope_sin)) return -1;     // the loader waves stage rows x head_dim <= 128 x 128 floats
    int rc;
    if (pro != PRO_NONE) return -1;          // rows are normalised by kg_launch_rmsnorm before the pass
    rc = (dtype == KG_F16) ? launch_nw<KG_F16>(a, epi, st) : launch_nw<KG_BF16>(a, epi, st);
    if (rc) return rc;
    if (a.part_only) {
        hipLaunchKernelGGL(kg_gemm_reduce_f32_kernel, dim3((a.N / 4 + 255) / 256, KG_GEMM_ROWS), dim3(256), 0, st, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const bool norm_after = (a.out_norm_w != nullptr && a.out_norm != nullptr);
    if (norm_after && epi != EPI_RESID) return -1;
    if (a.ksplit > 1 && norm_after) {
        if (dtype == KG_F16) hipLaunchKernelGGL(kg_gemm_reduce_norm_kernel<KG_F16>, dim3(KG_GEMM_ROWS), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(kg_gemm_reduce_norm_kernel<KG_BF16>, dim3(KG_GEMM_ROWS), dim3(256), 0, st, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (a.ksplit > 1) {
        const dim3 grid((a.N / 4 + 255) / 256, KG_GEMM_ROWS);
        if (dtype == KG_F16) hipLaunchKernelGGL(kg_gemm_reduce_kernel<KG_F16>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(kg_gemm_reduce_kernel<KG_BF16>, grid, dim3(256), 0, st, a);
    }
    if (hipGetLastError() != hipSuccess) return -2;
    if (norm_after) return kg_launch_rmsnorm(a.dyn, a.out, a.out_stride, a.out_norm_w, a.eps, a.N, a.out_norm, a.N, dtype, st);
    return 0;
}

template <int DT, int NW, int RPW> static void set_attr_all() {
    const int lim = (int)gemm_smem();
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, EPI_STORE, NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, EPI_RESID, NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, EPI_ROPE_KV, NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)kg_gemm_kernel<DT, EPI_SILU_MUL, NW, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
}
template <int DT> static void set_attr_dt() {
    set_attr_all<DT, 2, 1>(); set_attr_all<DT, 3, 1>(); set_attr_all<DT, 4, 1>();
    set_attr_all<DT, 2, 2>(); set_attr_all<DT, 3, 2>(); set_attr_all<DT, 4, 2>();
}
void kg_gemm_init() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) g_gemm_ncu = ncu;
    set_attr_dt<KG_F16>(); set_attr_dt<KG_BF16>();
    (void)hipGetLastError();
}
