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
#include "kg_common.h"
#include "kg_kernels.h"

#ifndef KG_U
#define KG_U 8          // weight tiles (KiB) per wave kept in flight per batch
#endif

enum { MODE_VEC = 0, MODE_MFMA = 1, MODE_MFMA_GX = 2 };

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
template <int DT, int EPI>
__device__ __forceinline__ void epilogue_pair(const KgLinArgs& a, int n0, int m, int pos, float v0, float v1, HeadRun& run) {
    typedef KgT<DT> T;
    unsigned short* out = (unsigned short*)a.out;
    if (EPI == EPI_STORE) {
        if (n0 < a.N) out[(size_t)m * a.out_stride + n0] = T::from_f(v0);
        if (n0 + 1 < a.N) out[(size_t)m * a.out_stride + n0 + 1] = T::from_f(v1);
    } else if (EPI == EPI_RESID) {
        const unsigned short* res = (const unsigned short*)a.resid;
        if (n0 < a.N) {
            float r = T::to_f(res[(size_t)m * a.resid_stride + n0]);
            out[(size_t)m * a.out_stride + n0] = T::from_f(r + kg_round<DT>(v0));
        }
        if (n0 + 1 < a.N) {
            float r = T::to_f(res[(size_t)m * a.resid_stride + n0 + 1]);
            out[(size_t)m * a.out_stride + n0 + 1] = T::from_f(r + kg_round<DT>(v1));
        }
    } else if (EPI == EPI_SILU_MUL) {
        // packed rows (2i, 2i+1) = (gate_i, up_i)
        int i = n0 >> 1;
        if (n0 + 1 < a.N) out[(size_t)m * a.out_stride + i] = T::from_f(silu_mul_ref<DT>(kg_round<DT>(v0), kg_round<DT>(v1)));
    } else if (EPI == EPI_ROPE_KV) {
        // packed rows inside a head: (2j, 2j+1) = original (j, j + hd/2)
        if (n0 + 1 >= a.N) return;
        const int hd = a.head_dim, half = hd >> 1;
        const int head = n0 / hd, hl = n0 - head * hd, j = hl >> 1;
        float x1 = kg_round<DT>(v0), x2 = kg_round<DT>(v1);
        if (n0 < a.q_dim + a.kv_dim) {
            float c = kg_round<DT>(a.rope_cos[(size_t)pos * half + j]);
            float s = kg_round<DT>(a.rope_sin[(size_t)pos * half + j]);
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
            dst = (unsigned short*)a.v_cache + ((size_t)kvh * a.t_max + pos) * hd;
        }
        dst[j] = T::from_f(x1);
        dst[j + half] = T::from_f(x2);
    } else if (EPI == EPI_HEAD) {
        float l0 = kg_round<DT>(v0), l1 = kg_round<DT>(v1);
        if (a.logits_out) {
            unsigned short* lo = (unsigned short*)a.logits_out;
            if (n0 < a.N) lo[(size_t)m * a.logits_stride + n0] = T::from_f(v0);
            if (n0 + 1 < a.N) lo[(size_t)m * a.logits_stride + n0 + 1] = T::from_f(v1);
        }
        if (n0 < a.N) head_push(run, l0, n0);
        if (n0 + 1 < a.N) head_push(run, l1, n0 + 1);
    }
}

// ---- prologue: fill X rows in LDS -------------------------------------------------
// xs: LDS base of X, row stride xs_stride bytes.  Each wave takes rows m = wave, wave+W, ...
template <int DT, int PRO>
__device__ __forceinline__ void prologue_fill(const KgLinArgs& a, char* xs, int xs_stride, int M, int row_off, int wave, int nwaves, int lane) {
    typedef KgT<DT> T;
    const int K = a.K;
    const int nchunk = K >> 3;                      // 16-byte chunks per row
    for (int m = wave; m < M; m += nwaves) {
        const kg_u4* src = (const kg_u4*)((const unsigned short*)a.x + (size_t)(m + row_off) * a.x_stride);
        kg_u4* dst = (kg_u4*)(xs + (size_t)m * xs_stride);
        if (PRO == PRO_NONE) {
            for (int c = lane; c < nchunk; c += 64) dst[c] = src[c];
        } else {
            float ss = 0.f;
            for (int c = lane; c < nchunk; c += 64) {
                kg_u4 v = src[c];
                kg_us8 h = __builtin_bit_cast(kg_us8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = T::to_f(h[j]); ss += f * f; }
            }
            ss = kg_wave_sum(ss);
            const float rstd = rsqrtf(ss / (float)K + a.eps);
            const kg_u4* wn = (const kg_u4*)a.norm_w;
            for (int c = lane; c < nchunk; c += 64) {
                kg_us8 h = __builtin_bit_cast(kg_us8, src[c]);
                kg_us8 g = __builtin_bit_cast(kg_us8, wn[c]);
                kg_us8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // normalise in fp32, store in the model dtype, then scale by the weight
                    float f = kg_round<DT>(T::to_f(h[j]) * rstd);
                    o[j] = T::from_f(T::to_f(g[j]) * f);
                }
                dst[c] = __builtin_bit_cast(kg_u4, o);
            }
        }
    }
}

template <int DT, int MODE, int PRO, int EPI>
__global__ void __launch_bounds__(512) kg_linear_kernel(const KgLinArgs a) {
    typedef KgT<DT> T;
    const KgDyn* dyn = a.dyn;
    if (dyn->active == 0) return;
    const int M = (MODE == MODE_VEC) ? 1 : dyn->n_rows;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int PART = (MODE == MODE_VEC) ? 16 : 256;            // floats per wave partial
    float* part = (float*)smem;                                  // [2][nwaves][PART]
    char* xs = smem + 2 * nwaves * PART * 4;
    const int xs_stride = a.K * 2 + 32;

    const int G = (a.N + 15) >> 4, KT = a.K >> 5;
    const int t0 = (KT * wave) / nwaves, t1 = (KT * (wave + 1)) / nwaves;
    const kg_u4* Wp = (const kg_u4*)a.W;
    const int g = lane >> 4, r = lane & 15;

    // B-operand / x source for this lane
    const int mrow = (r < M) ? r : 0;
    const char* xl = xs + (size_t)((MODE == MODE_VEC) ? 0 : mrow) * xs_stride + g * 16;
    const kg_u4* xg = (const kg_u4*)((const unsigned short*)a.x + (size_t)mrow * a.x_stride) + g;   // MODE_MFMA_GX

    int rg = blockIdx.x;
    kg_u4 cur[KG_U], nxt[KG_U];
    kg_u4 curx[KG_U], nxtx[KG_U];
    const kg_u4* dummy = Wp + (size_t)lane;

    auto issue = [&](kg_u4 (&wv)[KG_U], kg_u4 (&xv)[KG_U], int lrg, int lt) {
        const kg_u4* base = Wp + ((size_t)lrg * KT) * 64 + lane;
#pragma unroll
        for (int u = 0; u < KG_U; ++u) {
            const bool ok = (lrg < G) && (lt + u < t1);
            const kg_u4* p = ok ? (base + (size_t)(lt + u) * 64) : dummy;
            wv[u] = kg_ldg_stream(p);
            if (MODE == MODE_MFMA_GX) {
                const kg_u4* q = ok ? (xg + (size_t)(lt + u) * 4) : xg;
                xv[u] = *q;
            }
        }
    };

    int lrg = rg, lt = t0;
    issue(cur, curx, lrg, lt);
    lt += KG_U;
    if (lt >= t1) { lt = t0; lrg += gridDim.x; }

    if (MODE != MODE_MFMA_GX) {
        prologue_fill<DT, PRO>(a, xs, xs_stride, M, (MODE == MODE_VEC && a.x_last_row) ? dyn->n_rows - 1 : 0, wave, nwaves, lane);
        __syncthreads();
    }

    HeadRun run; run.mx = -INFINITY; run.sum = 0.f; run.idx = 0;
    int it = 0;
    for (; rg < G; rg += gridDim.x, ++it) {
        kg_f4 acc = {0.f, 0.f, 0.f, 0.f};
        float acc0 = 0.f, acc1 = 0.f;
        for (int ct = t0; ct < t1; ct += KG_U) {
            issue(nxt, nxtx, lrg, lt);
            lt += KG_U;
            if (lt >= t1) { lt = t0; lrg += gridDim.x; }
#pragma unroll
            for (int u = 0; u < KG_U; ++u) {
                const bool ok = (ct + u < t1);
                kg_u4 wv = cur[u];
                if (!ok) { wv.x = 0; wv.y = 0; wv.z = 0; wv.w = 0; }
                if (MODE == MODE_VEC) {
                    const int tt = ok ? (ct + u) : t0;
                    kg_u4 xv = *(const kg_u4*)(xl + (size_t)tt * 64);
                    acc0 = T::dot2(wv.x, xv.x, acc0);
                    acc1 = T::dot2(wv.y, xv.y, acc1);
                    acc0 = T::dot2(wv.z, xv.z, acc0);
                    acc1 = T::dot2(wv.w, xv.w, acc1);
                } else if (MODE == MODE_MFMA) {
                    const int tt = ok ? (ct + u) : t0;
                    kg_u4 xv = *(const kg_u4*)(xl + (size_t)tt * 64);
                    acc = T::mfma(wv, xv, acc);
                } else {
                    acc = T::mfma(wv, curx[u], acc);
                }
            }
#pragma unroll
            for (int u = 0; u < KG_U; ++u) { cur[u] = nxt[u]; if (MODE == MODE_MFMA_GX) curx[u] = nxtx[u]; }
        }
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
        const int n_epi = (MODE == MODE_VEC) ? 8 : 128;
        if (tid < n_epi) {
            const int m = tid >> 3, p = tid & 7;
            if (m < M) {
                float v0 = 0.f, v1 = 0.f;
                for (int w = 0; w < nwaves; ++w) {
                    const float* s = slot + w * PART + m * 16 + 2 * p;
                    v0 += s[0]; v1 += s[1];
                }
                epilogue_pair<DT, EPI>(a, rg * 16 + 2 * p, m, dyn->pos[m], v0, v1, run);
            }
        }
    }
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
            if (tid < n_epi && (tid & 7) == 0 && m < M) {
                float* hp = a.head_part + ((size_t)m * a.head_part_stride + blockIdx.x) * 4;
                hp[0] = run.mx; hp[1] = run.sum; hp[2] = __int_as_float(run.idx); hp[3] = 0.f;
            }
        }
    }
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
int kg_launch_linear(const KgLinArgs& a, int dtype, int max_rows, int pro, int epi, int nblocks, int nwaves, hipStream_t st) {
    if (a.K % 32 != 0 || a.K <= 0 || a.N <= 0) return -1;
    if (max_rows < 1 || max_rows > KG_MAX_ROWS) return -1;
    if (nwaves < 1 || nwaves > 8) return -1;
    int mode = (max_rows == 1) ? MODE_VEC : MODE_MFMA;
    size_t smem = kg_linear_smem(mode != MODE_VEC, max_rows, a.K, nwaves);
    if (mode == MODE_MFMA) {
        // X rows that would leave less than two blocks per CU (or not fit at all) are read
        // as B fragments straight from global memory (L2) when no prologue has to rewrite them
        if (pro == PRO_NONE && smem > 80 * 1024) { mode = MODE_MFMA_GX; smem = (size_t)2 * nwaves * 256 * 4; }
        else if (smem > KG_LDS_LIMIT) return -3;
    }
    const int G = (a.N + 15) / 16;
    if (nblocks > G) nblocks = G;
    if (nblocks < 1) nblocks = 1;
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
}
void kg_linear_init() {
    set_attr_epi<KG_F16, MODE_MFMA, PRO_NONE>(); set_attr_epi<KG_F16, MODE_MFMA, PRO_RMSNORM>();
    set_attr_epi<KG_BF16, MODE_MFMA, PRO_NONE>(); set_attr_epi<KG_BF16, MODE_MFMA, PRO_RMSNORM>();
    set_attr_epi<KG_F16, MODE_VEC, PRO_NONE>(); set_attr_epi<KG_F16, MODE_VEC, PRO_RMSNORM>();
    set_attr_epi<KG_BF16, MODE_VEC, PRO_NONE>(); set_attr_epi<KG_BF16, MODE_VEC, PRO_RMSNORM>();
}
