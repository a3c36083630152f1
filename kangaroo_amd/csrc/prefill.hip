// Row kernels of the prompt pass that runs layer by layer over the WHOLE prompt (engine.hip: prefill_rows_pass): what
// sits between the matrix products of csrc/mm.hip.  Same arithmetic, rounding for rounding, as the fused prologues and
// epilogues of the decode kernels (kg_linear_body.h), on `rows` rows at consecutive positions instead of a KgDyn block.
#include "kg_common.h"
#include "kg_kernels.h"

// out[m][:] = embed[tokens[pos0 + m]][:]
__global__ void kg_rows_embed_kernel(const int* tokens, int pos0, const kg_u4* embed, int d8, kg_u4* out) {
    const int m = blockIdx.x;
    const kg_u4* src = embed + (size_t)tokens[pos0 + m] * d8;
    kg_u4* dst = out + (size_t)m * d8;
    for (int i = threadIdx.x; i < d8; i += blockDim.x) dst[i] = src[i];
}
int kg_launch_rows_embed(const int* tokens, int pos0, int rows, const void* embed, int d, void* out, hipStream_t st) {
    if (d % 8 || rows < 1) return -1;
    hipLaunchKernelGGL(kg_rows_embed_kernel, dim3(rows), dim3(256), 0, st, tokens, pos0, (const kg_u4*)embed, d / 8, (kg_u4*)out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// LlamaRMSNorm (kangaroo/adapter.py:254-259): x * rsqrt(mean x^2 + eps) rounded to 16 bits, times gamma, rounded again.
// One block per row; thread t owns the 16-byte pieces t, t + 256, ... (K % 8 == 0), as kg_rmsnorm_kernel does.
template <int DT>
__global__ void __launch_bounds__(256) kg_rows_rmsnorm_kernel(const unsigned short* x, int ld, const unsigned short* w, float eps, int K, unsigned short* out, int ld_out) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const unsigned short* xr = x + (size_t)blockIdx.x * ld;
    unsigned short* orow = out + (size_t)blockIdx.x * ld_out;
    float ss = 0.f;
    for (int i = threadIdx.x; i < (K >> 3); i += 256) {
        const kg_us8 h = __builtin_bit_cast(kg_us8, ((const kg_u4*)xr)[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = T::to_f(h[e]); ss += f * f; }
    }
    ss = kg_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float rstd = rsqrtf(ss / (float)K + eps);
    for (int i = threadIdx.x; i < (K >> 3); i += 256) {
        const kg_us8 h = __builtin_bit_cast(kg_us8, ((const kg_u4*)xr)[i]);
        const kg_us8 g = __builtin_bit_cast(kg_us8, ((const kg_u4*)w)[i]);
        kg_us8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = T::from_f(T::to_f(g[e]) * kg_round<DT>(T::to_f(h[e]) * rstd));
        ((kg_u4*)orow)[i] = __builtin_bit_cast(kg_u4, o);
    }
}
int kg_launch_rows_rmsnorm(const void* x, int ld, const void* w, float eps, int K, void* out, int ld_out, int rows, int dtype, hipStream_t st) {
    if (!x || !w || !out || rows < 1 || (K & 7) || (ld & 7) || (ld_out & 7)) return -1;
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_rows_rmsnorm_kernel<KG_F16>, dim3(rows), dim3(256), 0, st, (const unsigned short*)x, ld, (const unsigned short*)w, eps, K, (unsigned short*)out, ld_out);
    else hipLaunchKernelGGL(kg_rows_rmsnorm_kernel<KG_BF16>, dim3(rows), dim3(256), 0, st, (const unsigned short*)x, ld, (const unsigned short*)w, eps, K, (unsigned short*)out, ld_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The end of a projection whose K was cut into ranges (csrc/mm.hip, ksplit): the ranges' fp32 partial sums are added in range
// order, rounded to 16 bits as the stored product would have been, added to the residual row and rounded again; with a
// weight vector the finished row is also normalised for the next projection (same arithmetic as kg_rows_rmsnorm_kernel on
// the stored row).  One block per row; a thread owns the 16-byte pieces t, t + 256, ...
template <int DT>
__global__ void __launch_bounds__(256) kg_rows_reduce_kernel(const float* part, int ksplit, size_t split_stride, int N, const unsigned short* resid, int ldr,
                                                             unsigned short* h_out, int ldh, const unsigned short* w, float eps, unsigned short* xn, int ldx) {
    typedef KgT<DT> T;
    __shared__ float red[4];
    const size_t m = blockIdx.x;
    const float* pr = part + m * (size_t)N;
    const unsigned short* rr = resid + m * ldr;
    unsigned short* hr = h_out + m * ldh;
    float ss = 0.f;
    for (int i = threadIdx.x; i < (N >> 3); i += 256) {
        kg_f4 lo = ((const kg_f4*)pr)[2 * i], hi = ((const kg_f4*)pr)[2 * i + 1];
        for (int r = 1; r < ksplit; ++r) {
            const kg_f4* q = (const kg_f4*)(pr + (size_t)r * split_stride);
            lo = lo + q[2 * i]; hi = hi + q[2 * i + 1];
        }
        const kg_us8 rs = __builtin_bit_cast(kg_us8, ((const kg_u4*)rr)[i]);
        kg_us8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = kg_round<DT>(T::to_f(rs[e]) + kg_round<DT>(e < 4 ? lo[e] : hi[e - 4]));
            o[e] = T::from_f(f); ss += f * f;
        }
        ((kg_u4*)hr)[i] = __builtin_bit_cast(kg_u4, o);
    }
    if (w == nullptr) return;
    ss = kg_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float rstd = rsqrtf(ss / (float)N + eps);
    unsigned short* orow = xn + m * ldx;
    for (int i = threadIdx.x; i < (N >> 3); i += 256) {
        const kg_us8 h = __builtin_bit_cast(kg_us8, ((const kg_u4*)hr)[i]);          // this thread's own stores
        const kg_us8 g = __builtin_bit_cast(kg_us8, ((const kg_u4*)w)[i]);
        kg_us8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = T::from_f(T::to_f(g[e]) * kg_round<DT>(T::to_f(h[e]) * rstd));
        ((kg_u4*)orow)[i] = __builtin_bit_cast(kg_u4, o);
    }
}
int kg_launch_rows_reduce(const float* part, int ksplit, long long split_stride, int N, const void* resid, int ldr, void* h_out, int ldh,
                          const void* w, float eps, void* xn, int ldx, int rows, int dtype, hipStream_t st) {
    if (!part || !resid || !h_out || ksplit < 1 || rows < 1 || (N & 7) || (ldr & 7) || (ldh & 7) || (split_stride & 3) || (w && (!xn || (ldx & 7)))) return -1;
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_rows_reduce_kernel<KG_F16>, dim3(rows), dim3(256), 0, st, part, ksplit, (size_t)split_stride, N, (const unsigned short*)resid, ldr,
                                            (unsigned short*)h_out, ldh, (const unsigned short*)w, eps, (unsigned short*)xn, ldx);
    else hipLaunchKernelGGL(kg_rows_reduce_kernel<KG_BF16>, dim3(rows), dim3(256), 0, st, part, ksplit, (size_t)split_stride, N, (const unsigned short*)resid, ldr,
                            (unsigned short*)h_out, ldh, (const unsigned short*)w, eps, (unsigned short*)xn, ldx);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// RoPE + cache append on the stored rows of the fused q/k/v product (kangaroo/adapter.py:80-95,181-189).  qkv[m] holds the
// packed row order of wqkv: [q heads | k heads | v heads], inside a head the pairs (j, j + hd/2) adjacent as (2j, 2j + 1).
// q goes to q_out[m] in the original order, K to cache slot pos0 + m, V likewise (transposed layout: [Hkv][hd][t_max]).
// The arithmetic is the EPI_ROPE_KV epilogue's: cos / sin rounded to 16 bits, every product and the sum rounded.
template <int DT>
__global__ void __launch_bounds__(256) kg_rows_rope_kv_kernel(const unsigned short* qkv, int pos0, int hd, int q_dim, int kv_dim, const float* rope_cos, const float* rope_sin,
                                                              unsigned short* q_out, unsigned short* k_cache, unsigned short* v_cache, int t_max, int v_transposed) {
    typedef KgT<DT> T;
    const int m = blockIdx.x, pos = pos0 + m, half = hd >> 1, N = q_dim + 2 * kv_dim;
    const unsigned short* row = qkv + (size_t)m * N;
    const float* cs = rope_cos + (size_t)pos * half;
    const float* sn = rope_sin + (size_t)pos * half;
    // a transposed V cache is written by kg_rows_v_transposed_kernel (whole lines instead of one element per line)
    const int n_end = v_transposed ? q_dim + kv_dim : N;
    for (int pr = threadIdx.x; pr < (n_end >> 1); pr += 256) {
        const int n0 = 2 * pr;
        const unsigned v2 = *(const unsigned*)(row + n0);
        float x1 = T::to_f((unsigned short)(v2 & 0xffffu)), x2 = T::to_f((unsigned short)(v2 >> 16));
        const int head = n0 / hd, j = (n0 - head * hd) >> 1;
        if (n0 < q_dim + kv_dim) {
            const float c = kg_round<DT>(cs[j]), s = kg_round<DT>(sn[j]);
            const float o1 = kg_round<DT>(x1 * c) + kg_round<DT>(-x2 * s);
            const float o2 = kg_round<DT>(x2 * c) + kg_round<DT>(x1 * s);
            x1 = o1; x2 = o2;
        }
        unsigned short* dst;
        if (n0 < q_dim) {
            dst = q_out + (size_t)m * q_dim + head * hd;
        } else if (n0 < q_dim + kv_dim) {
            const int kvh = head - q_dim / hd;
            dst = k_cache + ((size_t)kvh * t_max + pos) * hd;
        } else {
            const int kvh = head - (q_dim + kv_dim) / hd;
            if (v_transposed) {
                unsigned short* vt = v_cache + (size_t)kvh * hd * t_max + pos;
                vt[(size_t)j * t_max] = T::from_f(x1);
                vt[(size_t)(j + half) * t_max] = T::from_f(x2);
                continue;
            }
            dst = v_cache + ((size_t)kvh * t_max + pos) * hd;
        }
        dst[j] = T::from_f(x1);
        dst[j + half] = T::from_f(x2);
    }
}
// The V rows of the fused q/k/v product into the transposed cache layout [Hkv][hd][t_max]: a block takes 64 consecutive
// positions of one head, turns them round in LDS and writes 128-byte lines (64 positions of one head dim).  Element 2j of
// a head's stored row is dim j, element 2j + 1 dim j + hd / 2 (the packed row order of wqkv).
template <int HD>
__global__ void __launch_bounds__(256) kg_rows_v_transposed_kernel(const unsigned short* qkv, int rows, int pos0, int N, int v_off, unsigned short* v_cache, int t_max) {
    __shared__ unsigned short t[HD][66];                   // [dim][position], one spare word per row
    const int r0 = blockIdx.x * 64, kvh = blockIdx.y;
    const unsigned short* src = qkv + (size_t)r0 * N + v_off + kvh * HD;
    constexpr int CPR = HD / 8;                            // 16-byte pieces per stored row
    for (int c = threadIdx.x; c < 64 * CPR; c += 256) {
        const int r = c / CPR, ch = c - r * CPR;
        kg_u4 v = kg_u4{0u, 0u, 0u, 0u};
        if (r0 + r < rows) v = *(const kg_u4*)(src + (size_t)r * N + 8 * ch);
        const kg_us8 h = __builtin_bit_cast(kg_us8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 8 * ch + e, dim = (n >> 1) + (n & 1) * (HD / 2);
            t[dim][r] = h[e];
        }
    }
    __syncthreads();
    unsigned short* dst = v_cache + (size_t)kvh * HD * t_max + pos0 + r0;
    const bool wide = ((pos0 + r0) & 7) == 0 && (t_max & 7) == 0;
    for (int c = threadIdx.x; c < HD * 8; c += 256) {
        const int dim = c >> 3, p8 = (c & 7) * 8;
        if (wide && r0 + p8 + 8 <= rows) {
            kg_us8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = t[dim][p8 + e];
            *(kg_u4*)(dst + (size_t)dim * t_max + p8) = __builtin_bit_cast(kg_u4, o);
        } else {
            for (int e = 0; e < 8 && r0 + p8 + e < rows; ++e) dst[(size_t)dim * t_max + p8 + e] = t[dim][p8 + e];
        }
    }
}

int kg_launch_rows_rope_kv(const void* qkv, int rows, int pos0, int head_dim, int q_dim, int kv_dim, const float* rope_cos, const float* rope_sin,
                           void* q_out, void* k_cache, void* v_cache, int t_max, int v_transposed, int dtype, hipStream_t st) {
    if (!qkv || !rope_cos || !rope_sin || !q_out || !k_cache || !v_cache || rows < 1 || pos0 < 0 || pos0 + rows > t_max || (head_dim & 1)) return -1;
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_rows_rope_kv_kernel<KG_F16>, dim3(rows), dim3(256), 0, st, (const unsigned short*)qkv, pos0, head_dim, q_dim, kv_dim, rope_cos, rope_sin,
                                            (unsigned short*)q_out, (unsigned short*)k_cache, (unsigned short*)v_cache, t_max, v_transposed);
    else hipLaunchKernelGGL(kg_rows_rope_kv_kernel<KG_BF16>, dim3(rows), dim3(256), 0, st, (const unsigned short*)qkv, pos0, head_dim, q_dim, kv_dim, rope_cos, rope_sin,
                            (unsigned short*)q_out, (unsigned short*)k_cache, (unsigned short*)v_cache, t_max, v_transposed);
    if (v_transposed) {
        const int N = q_dim + 2 * kv_dim;
        if ((N & 7) || (q_dim & 7) || (kv_dim & 7)) return -1;
        const dim3 grid((rows + 63) / 64, kv_dim / head_dim);
        switch (head_dim) {
            case 32: hipLaunchKernelGGL(kg_rows_v_transposed_kernel<32>, grid, dim3(256), 0, st, (const unsigned short*)qkv, rows, pos0, N, q_dim + kv_dim, (unsigned short*)v_cache, t_max); break;
            case 64: hipLaunchKernelGGL(kg_rows_v_transposed_kernel<64>, grid, dim3(256), 0, st, (const unsigned short*)qkv, rows, pos0, N, q_dim + kv_dim, (unsigned short*)v_cache, t_max); break;
            case 128: hipLaunchKernelGGL(kg_rows_v_transposed_kernel<128>, grid, dim3(256), 0, st, (const unsigned short*)qkv, rows, pos0, N, q_dim + kv_dim, (unsigned short*)v_cache, t_max); break;
            default: return -1;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
