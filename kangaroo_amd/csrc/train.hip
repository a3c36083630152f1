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

// y = round16(gamma * x * rstd), rstd = rsqrt(mean(x^2) + eps) kept for the backward pass
template <int DT>
__global__ void __launch_bounds__(TB) rmsnorm_fwd_kernel(const unsigned short* x, const float* gamma, float eps, int d, unsigned short* y, float* rstd_out) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const unsigned short* xr = x + row * d;
    float ss = 0.f;
    for (int i = threadIdx.x; i < d; i += TB) { const float f = T::to_f(xr[i]); ss += f * f; }
    ss = block_sum(ss, red);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (threadIdx.x == 0) rstd_out[row] = rstd;
    for (int i = threadIdx.x; i < d; i += TB) y[row * d + i] = T::from_f(gamma[i] * (T::to_f(xr[i]) * rstd));
}

// With g = gamma * dy and xh = x * rstd:  dx = rstd * (g - xh * mean(g * xh)),  dgamma = sum over rows of dy * xh.
// Block b walks rows b, b + gridDim.x, ... and leaves its share of dgamma in part[b][d]; a second kernel adds the
// shares in block order.
template <int DT>
__global__ void __launch_bounds__(TB) rmsnorm_bwd_kernel(const unsigned short* dy, const unsigned short* x, const float* gamma, const float* rstd,
                                                          int rows, int d, unsigned short* dx, float* part) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    constexpr int MAXC = 32;                       // columns per thread: d <= 8192
    float dg[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) dg[c] = 0.f;
    for (size_t row = blockIdx.x; row < (size_t)rows; row += gridDim.x) {
        const float rs = rstd[row];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int i = threadIdx.x + c * TB;
            if (i < d) {
                const float g = gamma[i] * T::to_f(dy[row * d + i]), xh = T::to_f(x[row * d + i]) * rs;
                dot += g * xh;
                dg[c] += T::to_f(dy[row * d + i]) * xh;
            }
        }
        dot = block_sum(dot, red) / (float)d;
        if (dx) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int i = threadIdx.x + c * TB;
                if (i < d) {
                    const float g = gamma[i] * T::to_f(dy[row * d + i]), xh = T::to_f(x[row * d + i]) * rs;
                    dx[row * d + i] = T::from_f(rs * (g - xh * dot));
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = threadIdx.x + c * TB;
        if (i < d) part[(size_t)blockIdx.x * d + i] = dg[c];
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
// key_mask[b][j] != 0 (the causal mask plus the padding mask of kangaroo/adapter.py:327-348), zeros elsewhere.
template <int DT>
__global__ void __launch_bounds__(TB) softmax_fwd_kernel(unsigned short* s, const int* key_mask, int H, int S, float scale) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int i = (int)(row % S), b = (int)(row / ((size_t)H * S));
    unsigned short* sr = s + row * S;
    const int* km = key_mask + (size_t)b * S;
    constexpr int MAXC = 16;                       // S <= 4096
    float v[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int j = threadIdx.x + c * TB;
        const bool ok = j <= i && j < S && km[j] != 0;
        v[c] = ok ? T::to_f(sr[j]) * scale : -INFINITY;
        mx = fmaxf(mx, v[c]);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { v[c] = (v[c] == -INFINITY) ? 0.f : __expf(v[c] - mx); sum += v[c]; }
    sum = block_sum(sum, red);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int j = threadIdx.x + c * TB;
        if (j < S) sr[j] = T::from_f(v[c] * inv);
    }
}
// ds_j = scale * p_j * (dp_j - sum_k dp_k p_k), written over dp
template <int DT>
__global__ void __launch_bounds__(TB) softmax_bwd_kernel(const unsigned short* p, unsigned short* dp, int S, float scale) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const unsigned short* pr = p + row * S;
    unsigned short* dr = dp + row * S;
    constexpr int MAXC = 16;
    float pv[MAXC], dv[MAXC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int j = threadIdx.x + c * TB;
        pv[c] = j < S ? T::to_f(pr[j]) : 0.f;
        dv[c] = j < S ? T::to_f(dr[j]) : 0.f;
        dot += pv[c] * dv[c];
    }
    dot = block_sum(dot, red);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int j = threadIdx.x + c * TB;
        if (j < S) dr[j] = T::from_f(scale * pv[c] * (dv[c] - dot));
    }
}

// One row of out-logits o and target-logits t (train.py:265-279):
//   loss = lse(o) - sum_v softmax(t)_v o_v,   acc = sum_v min(softmax(t)_v, softmax(o)_v),
//   dlogits_v = grad_scale * (softmax(o)_v - softmax(t)_v)   (written over o; zeros when the row's loss_mask is 0).
template <int DT>
__global__ void __launch_bounds__(TB) softce_kernel(unsigned short* o, const unsigned short* t, const int* loss_mask, int V, float grad_scale,
                                                     float* row_loss, float* row_acc) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    unsigned short* orow = o + row * V;
    const unsigned short* trow = t + row * V;
    if (loss_mask[row] == 0) {
        for (int i = threadIdx.x; i < V; i += TB) orow[i] = 0;
        if (threadIdx.x == 0) { row_loss[row] = 0.f; row_acc[row] = 0.f; }
        return;
    }
    float mo = -INFINITY, mt = -INFINITY;
    for (int i = threadIdx.x; i < V; i += TB) { mo = fmaxf(mo, T::to_f(orow[i])); mt = fmaxf(mt, T::to_f(trow[i])); }
    mo = block_max(mo, red);
    mt = block_max(mt, red);
    float so = 0.f, st = 0.f, cr = 0.f;
    for (int i = threadIdx.x; i < V; i += TB) {
        const float ov = T::to_f(orow[i]), et = expf(T::to_f(trow[i]) - mt);
        so += expf(ov - mo); st += et; cr += et * ov;
    }
    so = block_sum(so, red);
    st = block_sum(st, red);
    cr = block_sum(cr, red);
    const float lse_o = mo + logf(so), inv_st = 1.f / st;
    float acc = 0.f;
    for (int i = threadIdx.x; i < V; i += TB) {
        const float po = expf(T::to_f(orow[i]) - lse_o), pt = expf(T::to_f(trow[i]) - mt) * inv_st;
        acc += fminf(po, pt);
        orow[i] = T::from_f(grad_scale * (po - pt));
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) { row_loss[row] = lse_o - cr * inv_st; row_acc[row] = acc; }
}

// torch.optim.AdamW step t on one parameter, after clamping the unscaled gradient to [-clip, clip]
// (accelerator.clip_grad_value_, train.py:285); also refreshes the 16-bit copy the GEMMs read.
template <int GDT, int WDT>
__global__ void adamw_kernel(float* p, float* m, float* v, const void* grad, float inv_scale, float clip, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, size_t n, unsigned short* p16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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

int done(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { kg_set_error(what, e); return -2; }
    return 0;
}

}  // namespace

extern "C" int kg_train_rmsnorm_fwd(int dtype, const void* x, const float* gamma, float eps, int rows, int d, void* y, float* rstd, void* stream) {
    if (!x || !gamma || !y || !rstd || rows < 1 || d < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KG_F16) hipLaunchKernelGGL(rmsnorm_fwd_kernel<KG_F16>, dim3(rows), dim3(TB), 0, st, (const unsigned short*)x, gamma, eps, d, (unsigned short*)y, rstd);
    else hipLaunchKernelGGL(rmsnorm_fwd_kernel<KG_BF16>, dim3(rows), dim3(TB), 0, st, (const unsigned short*)x, gamma, eps, d, (unsigned short*)y, rstd);
    return done("kg_train_rmsnorm_fwd");
}

extern "C" size_t kg_train_rmsnorm_bwd_scratch(int rows, int d) { return (size_t)(rows < 256 ? rows : 256) * d * sizeof(float); }

extern "C" int kg_train_rmsnorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* rstd, int rows, int d,
                                    void* dx, float* dgamma, void* scratch, void* stream) {
    if (!dy || !x || !gamma || !rstd || !dgamma || !scratch || rows < 1 || d < 1 || d > 32 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rows < 256 ? rows : 256;
    if (dtype == KG_F16) hipLaunchKernelGGL(rmsnorm_bwd_kernel<KG_F16>, dim3(nblk), dim3(TB), 0, st, (const unsigned short*)dy, (const unsigned short*)x, gamma, rstd, rows, d, (unsigned short*)dx, (float*)scratch);
    else hipLaunchKernelGGL(rmsnorm_bwd_kernel<KG_BF16>, dim3(nblk), dim3(TB), 0, st, (const unsigned short*)dy, (const unsigned short*)x, gamma, rstd, rows, d, (unsigned short*)dx, (float*)scratch);
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
    const dim3 grid((unsigned)((size_t)B * H * S));
    if (dtype == KG_F16) hipLaunchKernelGGL(softmax_fwd_kernel<KG_F16>, grid, dim3(TB), 0, st, (unsigned short*)scores, key_mask, H, S, scale);
    else hipLaunchKernelGGL(softmax_fwd_kernel<KG_BF16>, grid, dim3(TB), 0, st, (unsigned short*)scores, key_mask, H, S, scale);
    return done("kg_train_softmax_fwd");
}

extern "C" int kg_train_softmax_bwd(int dtype, const void* probs, void* dprobs, int B, int H, int S, float scale, void* stream) {
    if (!probs || !dprobs || B < 1 || H < 1 || S < 1 || S > 16 * TB) return -1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((size_t)B * H * S));
    if (dtype == KG_F16) hipLaunchKernelGGL(softmax_bwd_kernel<KG_F16>, grid, dim3(TB), 0, st, (const unsigned short*)probs, (unsigned short*)dprobs, S, scale);
    else hipLaunchKernelGGL(softmax_bwd_kernel<KG_BF16>, grid, dim3(TB), 0, st, (const unsigned short*)probs, (unsigned short*)dprobs, S, scale);
    return done("kg_train_softmax_bwd");
}

extern "C" int kg_train_softce(int dtype, void* logits_out, const void* logits_tgt, const int* loss_mask, int rows, int vocab, float grad_scale,
                               float* row_loss, float* row_acc, void* stream) {
    if (!logits_out || !logits_tgt || !loss_mask || !row_loss || !row_acc || rows < 1 || vocab < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KG_F16) hipLaunchKernelGGL(softce_kernel<KG_F16>, dim3(rows), dim3(TB), 0, st, (unsigned short*)logits_out, (const unsigned short*)logits_tgt, loss_mask, vocab, grad_scale, row_loss, row_acc);
    else hipLaunchKernelGGL(softce_kernel<KG_BF16>, dim3(rows), dim3(TB), 0, st, (unsigned short*)logits_out, (const unsigned short*)logits_tgt, loss_mask, vocab, grad_scale, row_loss, row_acc);
    return done("kg_train_softce");
}

// grad_dtype: KG_F16 / KG_BF16 / 2 (fp32); copy_dtype: KG_F16 / KG_BF16 (p16 may be NULL); step is 1-based.
extern "C" int kg_train_adamw(float* p, float* m, float* v, const void* grad, int grad_dtype, float inv_scale, float clip, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, long long n, void* p16, int copy_dtype, void* stream) {
    if (!p || !m || !v || !grad || n < 1 || step < 1 || grad_dtype < 0 || grad_dtype > 2) return -1;
    hipStream_t st = (hipStream_t)stream;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const dim3 grid((unsigned)(((size_t)n + 255) / 256));
    unsigned short* c16 = (unsigned short*)p16;
#define KG_ADAMW(G, W) hipLaunchKernelGGL((adamw_kernel<G, W>), grid, dim3(256), 0, st, p, m, v, grad, inv_scale, clip, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, (size_t)n, c16)
    if (copy_dtype == KG_BF16) { if (grad_dtype == 0) KG_ADAMW(0, KG_BF16); else if (grad_dtype == 1) KG_ADAMW(1, KG_BF16); else KG_ADAMW(2, KG_BF16); }
    else { if (grad_dtype == 0) KG_ADAMW(0, KG_F16); else if (grad_dtype == 1) KG_ADAMW(1, KG_F16); else KG_ADAMW(2, KG_F16); }
#undef KG_ADAMW
    return done("kg_train_adamw");
}
