// Adapter training step (gfx950): the memory-bound pieces around the library GEMMs.
//
// The reference trains the adapter (one attention-only decoder layer + final RMSNorm) against the frozen
// lm_head with a soft-label cross entropy (train.py:258-296).  Its matrix products are plain GEMMs over
// batch x sequence rows (thousands of rows: compute-bound, left to the BLAS library by the host code in
// kangaroo_amd/train.py); everything between them streams activations once and is fused here:
//   RMSNorm forward / backward (kangaroo/adapter.py:254-259), rotary embedding and its adjoint (:87-95),
//   the masked softmax of the attention scores and its backward (:195-211, masks of :38-67),
//   log-softmax + soft-label cross entropy + expected acceptance + gradient of the logits in one pass over
//   the two logit rows (train.py:265-279), and gradient clipping + AdamW + the 16-bit working copy of the
//   weights in one pass over a parameter (train.py:227,285-286).
// Activations are 16-bit (fp16 or bf16), statistics and parameters fp32.  Every reduction runs in a fixed
// order, so a step is reproducible bit for bit.
#include "kg_common.h"
#include <cmath>

void kg_set_error(const char* what, hipError_t e);

namespace {

constexpr int TB = 256;           // threads per block of the row kernels

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = kg_wave_sum(v);
    __syncthreads();              // red may still be read by the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// A piece = W consecutive 16-bit values <-> floats.  W = 8: one 16-byte access (rows are 16-byte aligned when
// the row length is a multiple of 8: the launchers pick W by that); W = 1: any row length.
template <int DT, int W> __device__ __forceinline__ void loadw(const unsigned short* p, float (&f)[W]) {
    if (W == 8) {
        const kg_us8 v = __builtin_bit_cast(kg_us8, *(const kg_u4*)p);
#pragma unroll
        for (int e = 0; e < W; ++e) f[e] = KgT<DT>::to_f(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e) f[e] = KgT<DT>::to_f(p[e]);
    }
}
template <int DT, int W> __device__ __forceinline__ void storew(unsigned short* p, const float (&f)[W]) {
    if (W == 8) {
        kg_us8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = KgT<DT>::from_f(f[e % W]);
        *(kg_u4*)p = __builtin_bit_cast(kg_u4, v);
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e) p[e] = KgT<DT>::from_f(f[e]);
    }
}

// y = round16(gamma * x * rstd), rstd = rsqrt(mean(x^2) + eps) kept for the backward pass
template <int DT, int W>
__global__ void __launch_bounds__(TB) rmsnorm_fwd_kernel(const unsigned short* x, const float* gamma, float eps, int d, unsigned short* y, float* rstd_out) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const unsigned short* xr = x + row * d;
    constexpr int NP = 32 / W;                     // pieces per thread: d <= 8192
    float v[NP][W];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int i0 = (threadIdx.x + c * TB) * W;
        if (i0 < d) {
            loadw<DT, W>(xr + i0, v[c]);
#pragma unroll
            for (int e = 0; e < W; ++e) ss += v[c][e] * v[c][e];
        }
    }
    ss = block_sum(ss, red);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (threadIdx.x == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int i0 = (threadIdx.x + c * TB) * W;
        if (i0 < d) {
            float o[W];
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = gamma[i0 + e] * (v[c][e] * rstd);
            storew<DT, W>(y + row * d + i0, o);
        }
    }
}

// With g = gamma * dy and xh = x * rstd:  dx = rstd * (g - xh * mean(g * xh)),  dgamma = sum over rows of dy * xh.
// Block b walks rows b, b + gridDim.x, ... and leaves its share of dgamma in part[b][d]; a second kernel adds the
// shares in block order.
template <int DT, int W>
__global__ void __launch_bounds__(TB) rmsnorm_bwd_kernel(const unsigned short* dy, const unsigned short* x, const float* gamma, const float* rstd,
                                                          int rows, int d, unsigned short* dx, float* part) {
    __shared__ float red[4];
    constexpr int NP = 32 / W;                     // pieces per thread: d <= 8192
    float dg[NP][W], gm[NP][W];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int i0 = (threadIdx.x + c * TB) * W;
#pragma unroll
        for (int e = 0; e < W; ++e) { dg[c][e] = 0.f; gm[c][e] = (i0 < d) ? gamma[i0 + e] : 0.f; }
    }
    for (size_t row = blockIdx.x; row < (size_t)rows; row += gridDim.x) {
        const float rs = rstd[row];
        float g[NP][W], xh[NP][W];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const int i0 = (threadIdx.x + c * TB) * W;
            if (i0 < d) {
                float dyv[W];
                loadw<DT, W>(dy + row * d + i0, dyv);
                loadw<DT, W>(x + row * d + i0, xh[c]);
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    xh[c][e] *= rs;
                    g[c][e] = gm[c][e] * dyv[e];
                    dot += g[c][e] * xh[c][e];
                    dg[c][e] += dyv[e] * xh[c][e];
                }
            }
        }
        dot = block_sum(dot, red) / (float)d;
        if (dx) {
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const int i0 = (threadIdx.x + c * TB) * W;
                if (i0 < d) {
                    float o[W];
#pragma unroll
                    for (int e = 0; e < W; ++e) o[e] = rs * (g[c][e] - xh[c][e] * dot);
                    storew<DT, W>(dx + row * d + i0, o);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int i0 = (threadIdx.x + c * TB) * W;
        if (i0 < d) {
#pragma unroll
            for (int e = 0; e < W; ++e) part[(size_t)blockIdx.x * d + i0 + e] = dg[c][e];
        }
    }
}
__global__ void column_sum_kernel(const float* part, int nblk, int d, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * d + i];
    out[i] = s;
}

// Rotary embedding on [rows][n_heads][hd] in place, position = row % S.  Pair (j, j + hd/2):
// forward (a, b) -> (a cos - b sin, b cos + a sin); inverse = the adjoint (the rotation by the opposite angle).
template <int DT>
__global__ void rope_kernel(unsigned short* x, size_t rows, int S, int n_heads, int hd, const float* cs, const float* sn, int inverse) {
    typedef KgT<DT> T;
    const int half = hd >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = rows * n_heads * half;
    if (idx >= total) return;
    const int j = (int)(idx % half);
    const size_t rh = idx / half, row = rh / n_heads;
    const int pos = (int)(row % S);
    const float c = cs[(size_t)pos * half + j], s = inverse ? -sn[(size_t)pos * half + j] : sn[(size_t)pos * half + j];
    unsigned short* p = x + rh * hd;
    const float a = T::to_f(p[j]), b = T::to_f(p[j + half]);
    p[j] = T::from_f(a * c - b * s);
    p[j + half] = T::from_f(b * c + a * s);
}

// Row (b, h, i) of the scores [B][H][S][S]: p_j = softmax_j(scale * s_j) over the keys j <= i with
// key_mask[b][j] != 0 (the causal mask plus the padding mask of kangaroo/adapter.py:327-348), zeros elsewhere up to the end
// of the 256-column block of the diagonal (columns beyond it are left as they are).
// Thread t owns the pieces t, t + 256, ... (S <= 4096); pieces wholly in the future are never read.
template <int DT, int W>
__global__ void __launch_bounds__(TB) softmax_fwd_kernel(unsigned short* s, const int* key_mask, int H, int S, float scale) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int i = (int)(row % S), b = (int)(row / ((size_t)H * S));
    unsigned short* sr = s + row * S;
    const int* km = key_mask + (size_t)b * S;
    constexpr int NP = 16 / W;
    float v[NP][W];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (threadIdx.x + c * TB) * W;
#pragma unroll
        for (int e = 0; e < W; ++e) v[c][e] = -INFINITY;
        if (j0 <= i && j0 < S) {
            float f[W];
            loadw<DT, W>(sr + j0, f);
            int kmv[W];
            if (W == 8) {       // the mask row is 32-byte aligned when S is a multiple of 8
                const kg_u4 a = *(const kg_u4*)(km + j0), b4 = *(const kg_u4*)(km + j0 + 4);
                kmv[0] = a.x; kmv[1 % W] = a.y; kmv[2 % W] = a.z; kmv[3 % W] = a.w; kmv[4 % W] = b4.x; kmv[5 % W] = b4.y; kmv[6 % W] = b4.z; kmv[7 % W] = b4.w;
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) kmv[e] = km[j0 + e];
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const bool ok = j0 + e <= i && kmv[e] != 0;
                v[c][e] = ok ? f[e] * scale : -INFINITY;
                mx = fmaxf(mx, v[c][e]);
            }
        }
    }
    mx = block_max(mx, red);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int e = 0; e < W; ++e) { v[c][e] = (v[c][e] == -INFINITY) ? 0.f : __expf(v[c][e] - mx); sum += v[c][e]; }
    sum = block_sum(sum, red);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    // zeros are written up to the end of the 256-column block that holds the diagonal: the products that read the
    // probabilities walk the keys in tiles of at most 256 up to the tile of the diagonal and never look beyond it
    const int lim = min(S, ((i >> 8) + 1) << 8);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (threadIdx.x + c * TB) * W;
        if (j0 < lim) {
#pragma unroll
            for (int e = 0; e < W; ++e) v[c][e] *= inv;
            storew<DT, W>(sr + j0, v[c]);
        }
    }
}
// ds_j = scale * p_j * (dp_j - sum_k dp_k p_k), written over dp; p is zero beyond the query's position, so those
// pieces are written as zeros without being read
template <int DT, int W>
__global__ void __launch_bounds__(TB) softmax_bwd_kernel(const unsigned short* p, unsigned short* dp, int S, float scale) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int i = (int)(row % S);
    const unsigned short* pr = p + row * S;
    unsigned short* dr = dp + row * S;
    constexpr int NP = 16 / W;
    float pv[NP][W], dv[NP][W];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (threadIdx.x + c * TB) * W;
        if (j0 <= i && j0 < S) {
            loadw<DT, W>(pr + j0, pv[c]);
            loadw<DT, W>(dr + j0, dv[c]);
#pragma unroll
            for (int e = 0; e < W; ++e) dot += pv[c][e] * dv[c][e];
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) { pv[c][e] = 0.f; dv[c][e] = 0.f; }
        }
    }
    dot = block_sum(dot, red);
    const int lim = min(S, ((i >> 8) + 1) << 8);        // as the forward kernel: nothing reads beyond the diagonal's 256-column block
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (threadIdx.x + c * TB) * W;
        if (j0 < lim) {
            float o[W];
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = scale * pv[c][e] * (dv[c][e] - dot);
            storew<DT, W>(dr + j0, o);
        }
    }
}

// The same two kernels with a WAVE per row (S a multiple of 8, at most 512 * NP keys; NP pieces of 8 per lane): no block
// barrier and no LDS, four rows per block.  A row of 2048 keys is 4 KB: a block per row spends its time in two block-wide
// reductions, not on memory.
template <int DT, int NP>
__global__ void __launch_bounds__(256) softmax_fwd_wave_kernel(unsigned short* s, const int* key_mask, size_t rows, int H, int S, float scale) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % S), b = (int)(row / ((size_t)H * S));
    unsigned short* sr = s + row * S;
    const int* km = key_mask + (size_t)b * S;
    float v[NP][8];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (lane + c * 64) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = -INFINITY;
        if (j0 <= i && j0 < S) {
            float f[8];
            loadw<DT, 8>(sr + j0, f);
            const kg_u4 a = *(const kg_u4*)(km + j0), b4 = *(const kg_u4*)(km + j0 + 4);
            const unsigned kmv[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = j0 + e <= i && kmv[e] != 0;
                v[c][e] = ok ? f[e] * scale : -INFINITY;
                mx = fmaxf(mx, v[c][e]);
            }
        }
    }
    mx = kg_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[c][e] = (v[c][e] == -INFINITY) ? 0.f : __expf(v[c][e] - mx); sum += v[c][e]; }
    sum = kg_wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    const int lim = min(S, ((i >> 8) + 1) << 8);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (lane + c * 64) * 8;
        if (j0 < lim) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] *= inv;
            storew<DT, 8>(sr + j0, v[c]);
        }
    }
}
template <int DT, int NP>
__global__ void __launch_bounds__(256) softmax_bwd_wave_kernel(const unsigned short* p, unsigned short* dp, size_t rows, int S, float scale) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % S);
    const unsigned short* pr = p + row * S;
    unsigned short* dr = dp + row * S;
    float pv[NP][8], dv[NP][8];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (lane + c * 64) * 8;
        if (j0 <= i && j0 < S) {
            loadw<DT, 8>(pr + j0, pv[c]);
            loadw<DT, 8>(dr + j0, dv[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += pv[c][e] * dv[c][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { pv[c][e] = 0.f; dv[c][e] = 0.f; }
        }
    }
    dot = kg_wave_sum(dot);
    const int lim = min(S, ((i >> 8) + 1) << 8);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int j0 = (lane + c * 64) * 8;
        if (j0 < lim) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = scale * pv[c][e] * (dv[c][e] - dot);
            storew<DT, 8>(dr + j0, o);
        }
    }
}

// One row of out-logits o and target-logits t (train.py:265-279):
//   loss = lse(o) - sum_v softmax(t)_v o_v,   acc = sum_v min(softmax(t)_v, softmax(o)_v),
//   dlogits_v = grad_scale * (softmax(o)_v - softmax(t)_v)   (written over o; zeros when the row's loss_mask is 0),
//   rank = number of out-logits above the out-logit of the target's argmax (lowest index on ties): the target token is
//   among the top k predictions iff rank < k (train.py:298-305); -1 for rows outside the loss mask.
template <int DT, int W>
__global__ void __launch_bounds__(TB) softce_kernel(unsigned short* o, const unsigned short* t, const int* loss_mask, int V, size_t ld, float grad_scale,
                                                     float* row_loss, float* row_acc, int* row_rank) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    unsigned short* orow = o + row * ld;
    const unsigned short* trow = t + row * ld;
    if (loss_mask[row] == 0) {
        const float z[W] = {};
        for (int c = threadIdx.x; c < V / W; c += TB) storew<DT, W>(orow + W * c, z);
        if (threadIdx.x == 0) { row_loss[row] = 0.f; row_acc[row] = 0.f; if (row_rank) row_rank[row] = -1; }
        return;
    }
    // thread t takes pieces t, t + 256, ...; the row pair (4 * V bytes) stays in L2 between the sweeps
    const int NPC = V / W;
    float mo = -INFINITY, mt = -INFINITY;
    for (int c = threadIdx.x; c < NPC; c += TB) {
        float fo[W], ft[W];
        loadw<DT, W>(orow + W * c, fo); loadw<DT, W>(trow + W * c, ft);
#pragma unroll
        for (int e = 0; e < W; ++e) { mo = fmaxf(mo, fo[e]); mt = fmaxf(mt, ft[e]); }
    }
    mo = block_max(mo, red);
    mt = block_max(mt, red);
    float o_at_target = 0.f;
    if (row_rank) {
        // lowest index holding the target's maximum, then its out-logit (block-uniform after the reductions)
        int it = 0x7fffffff;
        for (int c = threadIdx.x; c < NPC && it == 0x7fffffff; c += TB) {
            float ft[W];
            loadw<DT, W>(trow + W * c, ft);
#pragma unroll
            for (int e = W - 1; e >= 0; --e) if (ft[e] == mt) it = W * c + e;
        }
        it = -(int)block_max(-(float)it, red);          // indices < 2^24: exact in fp32
        o_at_target = T::to_f(orow[it]);
    }
    float so = 0.f, st = 0.f, cr = 0.f;
    for (int c = threadIdx.x; c < NPC; c += TB) {
        float fo[W], ft[W];
        loadw<DT, W>(orow + W * c, fo); loadw<DT, W>(trow + W * c, ft);
#pragma unroll
        for (int e = 0; e < W; ++e) { const float et = expf(ft[e] - mt); so += expf(fo[e] - mo); st += et; cr += et * fo[e]; }
    }
    so = block_sum(so, red);
    st = block_sum(st, red);
    cr = block_sum(cr, red);
    const float lse_o = mo + logf(so), inv_st = 1.f / st;
    float acc = 0.f, above = 0.f;
    for (int c = threadIdx.x; c < NPC; c += TB) {
        float fo[W], ft[W], g[W];
        loadw<DT, W>(orow + W * c, fo); loadw<DT, W>(trow + W * c, ft);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float po = expf(fo[e] - lse_o), pt = expf(ft[e] - mt) * inv_st;
            acc += fminf(po, pt);
            above += (fo[e] > o_at_target) ? 1.f : 0.f;
            g[e] = grad_scale * (po - pt);
        }
        storew<DT, W>(orow + W * c, g);
    }
    acc = block_sum(acc, red);
    above = block_sum(above, red);
    if (threadIdx.x == 0) { row_loss[row] = lse_o - cr * inv_st; row_acc[row] = acc; if (row_rank) row_rank[row] = (int)above; }
}

// torch.optim.AdamW step t on one parameter, after clamping the unscaled gradient to [-clip, clip]
// (accelerator.clip_grad_value_, train.py:285); also refreshes the 16-bit copy the GEMMs read.
template <int GDT, int WDT>
__global__ void adamw_kernel(float* p, float* m, float* v, const void* grad, float inv_scale, float clip, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, size_t n, unsigned short* p16, const int* skip) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip && *skip != 0) return;          // a non-finite gradient somewhere in this step: leave parameters and moments alone
    float g = (GDT == 2) ? ((const float*)grad)[i] : KgT<GDT == 2 ? 0 : GDT>::to_f(((const unsigned short*)grad)[i]);
    g *= inv_scale;
    g = fminf(fmaxf(g, -clip), clip);
    float w = p[i] * (1.f - lr * wd);
    const float mm = b1 * m[i] + (1.f - b1) * g, vv = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mm; v[i] = vv;
    w -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    p[i] = w;
    if (p16) p16[i] = KgT<WDT>::from_f(w);
}

// y[i] = round16(y[i] + x[i]): the residual add of kangaroo/adapter.py:304 on stored 16-bit tensors
template <int DT>
__global__ void add16_kernel(unsigned short* y, const unsigned short* x, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const kg_us8 a = __builtin_bit_cast(kg_us8, ((const kg_u4*)y)[i]), b = __builtin_bit_cast(kg_us8, ((const kg_u4*)x)[i]);
    kg_us8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = KgT<DT>::from_f(KgT<DT>::to_f(a[e]) + KgT<DT>::to_f(b[e]));
    ((kg_u4*)y)[i] = __builtin_bit_cast(kg_u4, o);
}

// *flag |= 1 when a value of g is an infinity or a NaN (the exponent bits all set); flag is only ever raised here
__global__ void nonfinite_kernel(const unsigned* g, size_t n, int* flag) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

int done(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { kg_set_error(what, e); return -2; }
    return 0;
}

}  // namespace

extern "C" int kg_train_rmsnorm_fwd(int dtype, const void* x, const float* gamma, float eps, int rows, int d, void* y, float* rstd, void* stream) {
    if (!x || !gamma || !y || !rstd || rows < 1 || d < 1 || d > 32 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = (d & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
#define KG_RF(D, W) hipLaunchKernelGGL((rmsnorm_fwd_kernel<D, W>), dim3(rows), dim3(TB), 0, st, (const unsigned short*)x, gamma, eps, d, (unsigned short*)y, rstd)
    if (dtype == KG_F16) { if (wide) KG_RF(KG_F16, 8); else KG_RF(KG_F16, 1); }
    else { if (wide) KG_RF(KG_BF16, 8); else KG_RF(KG_BF16, 1); }
#undef KG_RF
    return done("kg_train_rmsnorm_fwd");
}

extern "C" size_t kg_train_rmsnorm_bwd_scratch(int rows, int d) { return (size_t)(rows < 256 ? rows : 256) * d * sizeof(float); }

extern "C" int kg_train_rmsnorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* rstd, int rows, int d,
                                    void* dx, float* dgamma, void* scratch, void* stream) {
    if (!dy || !x || !gamma || !rstd || !dgamma || !scratch || rows < 1 || d < 1 || d > 32 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rows < 256 ? rows : 256;
    const bool wide = (d & 7) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
#define KG_RB(D, W) hipLaunchKernelGGL((rmsnorm_bwd_kernel<D, W>), dim3(nblk), dim3(TB), 0, st, (const unsigned short*)dy, (const unsigned short*)x, gamma, rstd, rows, d, (unsigned short*)dx, (float*)scratch)
    if (dtype == KG_F16) { if (wide) KG_RB(KG_F16, 8); else KG_RB(KG_F16, 1); }
    else { if (wide) KG_RB(KG_BF16, 8); else KG_RB(KG_BF16, 1); }
#undef KG_RB
    hipLaunchKernelGGL(column_sum_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const float*)scratch, nblk, d, dgamma);
    return done("kg_train_rmsnorm_bwd");
}

extern "C" int kg_train_rope(int dtype, void* x, long long rows, int S, int n_heads, int head_dim, const float* cos_t, const float* sin_t, int inverse, void* stream) {
    if (!x || !cos_t || !sin_t || rows < 1 || S < 1 || n_heads < 1 || head_dim < 2 || (head_dim & 1)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)rows * n_heads * (head_dim / 2);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == KG_F16) hipLaunchKernelGGL(rope_kernel<KG_F16>, grid, dim3(256), 0, st, (unsigned short*)x, (size_t)rows, S, n_heads, head_dim, cos_t, sin_t, inverse);
    else hipLaunchKernelGGL(rope_kernel<KG_BF16>, grid, dim3(256), 0, st, (unsigned short*)x, (size_t)rows, S, n_heads, head_dim, cos_t, sin_t, inverse);
    return done("kg_train_rope");
}

extern "C" int kg_train_softmax_fwd(int dtype, void* scores, const int* key_mask, int B, int H, int S, float scale, void* stream) {
    if (!scores || !key_mask || B < 1 || H < 1 || S < 1 || S > 16 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * H * S;
    const dim3 grid((unsigned)rows);
    const bool wide = (S & 7) == 0 && (((uintptr_t)scores | (uintptr_t)key_mask) & 15) == 0;
    if (wide) {
        const dim3 gw((unsigned)((rows + 3) / 4));
        unsigned short* sc = (unsigned short*)scores;
#define KG_SMF(D, N) hipLaunchKernelGGL((softmax_fwd_wave_kernel<D, N>), gw, dim3(256), 0, st, sc, key_mask, rows, H, S, scale)
        if (dtype == KG_F16) { if (S <= 512) KG_SMF(KG_F16, 1); else if (S <= 1024) KG_SMF(KG_F16, 2); else if (S <= 2048) KG_SMF(KG_F16, 4); else KG_SMF(KG_F16, 8); }
        else { if (S <= 512) KG_SMF(KG_BF16, 1); else if (S <= 1024) KG_SMF(KG_BF16, 2); else if (S <= 2048) KG_SMF(KG_BF16, 4); else KG_SMF(KG_BF16, 8); }
#undef KG_SMF
    } else if (dtype == KG_F16) hipLaunchKernelGGL((softmax_fwd_kernel<KG_F16, 1>), grid, dim3(TB), 0, st, (unsigned short*)scores, key_mask, H, S, scale);
    else hipLaunchKernelGGL((softmax_fwd_kernel<KG_BF16, 1>), grid, dim3(TB), 0, st, (unsigned short*)scores, key_mask, H, S, scale);
    return done("kg_train_softmax_fwd");
}

extern "C" int kg_train_softmax_bwd(int dtype, const void* probs, void* dprobs, int B, int H, int S, float scale, void* stream) {
    if (!probs || !dprobs || B < 1 || H < 1 || S < 1 || S > 16 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * H * S;
    const dim3 grid((unsigned)rows);
    const bool wide = (S & 7) == 0 && (((uintptr_t)probs | (uintptr_t)dprobs) & 15) == 0;
    if (wide) {
        const dim3 gw((unsigned)((rows + 3) / 4));
        const unsigned short* pp = (const unsigned short*)probs; unsigned short* dpp = (unsigned short*)dprobs;
#define KG_SMB(D, N) hipLaunchKernelGGL((softmax_bwd_wave_kernel<D, N>), gw, dim3(256), 0, st, pp, dpp, rows, S, scale)
        if (dtype == KG_F16) { if (S <= 512) KG_SMB(KG_F16, 1); else if (S <= 1024) KG_SMB(KG_F16, 2); else if (S <= 2048) KG_SMB(KG_F16, 4); else KG_SMB(KG_F16, 8); }
        else { if (S <= 512) KG_SMB(KG_BF16, 1); else if (S <= 1024) KG_SMB(KG_BF16, 2); else if (S <= 2048) KG_SMB(KG_BF16, 4); else KG_SMB(KG_BF16, 8); }
#undef KG_SMB
    } else if (dtype == KG_F16) hipLaunchKernelGGL((softmax_bwd_kernel<KG_F16, 1>), grid, dim3(TB), 0, st, (const unsigned short*)probs, (unsigned short*)dprobs, S, scale);
    else hipLaunchKernelGGL((softmax_bwd_kernel<KG_BF16, 1>), grid, dim3(TB), 0, st, (const unsigned short*)probs, (unsigned short*)dprobs, S, scale);
    return done("kg_train_softmax_bwd");
}

extern "C" int kg_train_softce(int dtype, void* logits_out, const void* logits_tgt, const int* loss_mask, int rows, int vocab, long long row_stride, float grad_scale,
                               float* row_loss, float* row_acc, int* row_rank, void* stream) {
    if (!logits_out || !logits_tgt || !loss_mask || !row_loss || !row_acc || rows < 1 || vocab < 1 || vocab >= (1 << 24) || row_stride < vocab) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = ((vocab | row_stride) & 7) == 0 && (((uintptr_t)logits_out | (uintptr_t)logits_tgt) & 15) == 0;
#define KG_SOFTCE(D, W) hipLaunchKernelGGL((softce_kernel<D, W>), dim3(rows), dim3(TB), 0, st, (unsigned short*)logits_out, (const unsigned short*)logits_tgt, loss_mask, vocab, (size_t)row_stride, grad_scale, row_loss, row_acc, row_rank)
    if (dtype == KG_F16) { if (wide) KG_SOFTCE(KG_F16, 8); else KG_SOFTCE(KG_F16, 1); }
    else { if (wide) KG_SOFTCE(KG_BF16, 8); else KG_SOFTCE(KG_BF16, 1); }
#undef KG_SOFTCE
    return done("kg_train_softce");
}

extern "C" int kg_train_add(int dtype, void* y, const void* x, long long n, void* stream) {
    if (!y || !x || n < 8 || (n & 7) || (((uintptr_t)y | (uintptr_t)x) & 15)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t n8 = (size_t)n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    if (dtype == KG_F16) hipLaunchKernelGGL(add16_kernel<KG_F16>, grid, dim3(256), 0, st, (unsigned short*)y, (const unsigned short*)x, n8);
    else hipLaunchKernelGGL(add16_kernel<KG_BF16>, grid, dim3(256), 0, st, (unsigned short*)y, (const unsigned short*)x, n8);
    return done("kg_train_add");
}

extern "C" int kg_train_nonfinite(const float* g, long long n, int* flag, void* stream) {
    if (!g || !flag || n < 1) return -1;
    const size_t blocks = ((size_t)n + 255) / 256;
    hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)g, (size_t)n, flag);
    return done("kg_train_nonfinite");
}

// grad_dtype: KG_F16 / KG_BF16 / 2 (fp32); copy_dtype: KG_F16 / KG_BF16 (p16 may be NULL); step is 1-based.
extern "C" int kg_train_adamw(float* p, float* m, float* v, const void* grad, int grad_dtype, float inv_scale, float clip, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, long long n, void* p16, int copy_dtype, const int* skip, void* stream) {
    if (!p || !m || !v || !grad || n < 1 || step < 1 || grad_dtype < 0 || grad_dtype > 2) return -1;
    hipStream_t st = (hipStream_t)stream;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const dim3 grid((unsigned)(((size_t)n + 255) / 256));
    unsigned short* c16 = (unsigned short*)p16;
#define KG_ADAMW(G, W) hipLaunchKernelGGL((adamw_kernel<G, W>), grid, dim3(256), 0, st, p, m, v, grad, inv_scale, clip, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, (size_t)n, c16, skip)
    if (copy_dtype == KG_BF16) { if (grad_dtype == 0) KG_ADAMW(0, KG_BF16); else if (grad_dtype == 1) KG_ADAMW(1, KG_BF16); else KG_ADAMW(2, KG_BF16); }
    else { if (grad_dtype == 0) KG_ADAMW(0, KG_F16); else if (grad_dtype == 1) KG_ADAMW(1, KG_F16); else KG_ADAMW(2, KG_F16); }
#undef KG_ADAMW
    return done("kg_train_adamw");
}
