// Small glue kernels: per-pass parameters, embedding gather, standalone RMSNorm, and
// the device-side state machine of the draft / verify loop
// (evaluation/inference_kangaroo.py:46-124 of the reference), so that a whole round
// replays from one captured graph with no host decision inside it.
#include "kg_common.h"
#include "kg_kernels.h"

__global__ void kg_set_dyn_kernel(KgDyn* dyn, int active, int n_rows, int pos0) {
    int i = threadIdx.x;
    if (i == 0) { dyn->active = active; dyn->n_rows = n_rows; }
    if (i < KG_MAX_ROWS) dyn->pos[i] = pos0 + i;
}
int kg_launch_set_dyn(KgDyn* dyn, int active, int n_rows, int pos0, hipStream_t st) {
    if (n_rows < 0 || n_rows > KG_GEMM_ROWS) return -1;      // > KG_MAX_ROWS: prefill passes, which use pos[0] + m
    hipLaunchKernelGGL(kg_set_dyn_kernel, dim3(1), dim3(64), 0, st, dyn, active, n_rows, pos0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The same parameters with the first position read on the device (a pass inside a captured graph whose position
// changes from replay to replay).  A position outside [0, t_max - n_rows] switches the pass off instead of letting its
// cache writes leave the cache.
__global__ void kg_set_dyn_from_kernel(KgDyn* dyn, int n_rows, const int* pos0_p, int t_max) {
    const int i = threadIdx.x, pos0 = *pos0_p;
    const int ok = (pos0 >= 0 && pos0 + n_rows <= t_max) ? 1 : 0;
    if (i == 0) { dyn->active = ok; dyn->n_rows = n_rows; }
    if (i < KG_MAX_ROWS) dyn->pos[i] = ok ? pos0 + i : 0;
}
int kg_launch_set_dyn_from(KgDyn* dyn, int n_rows, const int* pos0_p, int t_max, hipStream_t st) {
    if (n_rows < 0 || n_rows > KG_GEMM_ROWS || !pos0_p) return -1;
    hipLaunchKernelGGL(kg_set_dyn_from_kernel, dim3(1), dim3(64), 0, st, dyn, n_rows, pos0_p, t_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// n words of zeros, as a kernel: inside a captured graph a memset node was seen to overlap the kernel node that
// follows it (arrival counters wiped while blocks were counting), kernel nodes keep their order.
__global__ void kg_zero_u32_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
int kg_launch_zero_u32(unsigned* p, int n, hipStream_t st) {
    if (!p || n < 0) return -1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(kg_zero_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// out[m][:] = embed[tokens[pos[0] + m]][:]   (16-bit elements, d % 8 == 0; rows of a pass sit at consecutive positions)
__global__ void kg_embed_kernel(const KgDyn* dyn, const int* tokens, const kg_u4* embed, int d8, kg_u4* out, int out_stride8) {
    if (dyn->active == 0) return;
    const int m = blockIdx.x;
    if (m >= dyn->n_rows) return;
    const int tok = tokens[dyn->pos[0] + m];
    const kg_u4* src = embed + (size_t)tok * d8;
    kg_u4* dst = out + (size_t)m * out_stride8;
    for (int i = threadIdx.x; i < d8; i += blockDim.x) dst[i] = src[i];
}
int kg_launch_embed(const KgDyn* dyn, const int* tokens, const void* embed, int d, void* out, int out_stride, hipStream_t st) {
    if (d % 8 || out_stride % 8) return -1;
    hipLaunchKernelGGL(kg_embed_kernel, dim3(KG_GEMM_ROWS), dim3(256), 0, st, dyn, tokens, (const kg_u4*)embed, d / 8, (kg_u4*)out, out_stride / 8);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int DT>
__global__ void kg_rmsnorm_kernel(const KgDyn* dyn, const unsigned short* x, int x_stride, const unsigned short* w, float eps, int K,
                                  unsigned short* out, int out_stride) {
    typedef KgT<DT> T;
    if (dyn->active == 0) return;
    const int m = blockIdx.x;
    if (m >= dyn->n_rows) return;
    __shared__ float red[4];
    const unsigned short* xr = x + (size_t)m * x_stride;
    unsigned short* orow = out + (size_t)m * out_stride;
    const bool vec = ((K | x_stride | out_stride) & 7) == 0;     // 16-byte pieces
    float ss = 0.f;
    if (vec) {
        for (int i = threadIdx.x; i < (K >> 3); i += blockDim.x) {
            const kg_us8 h = __builtin_bit_cast(kg_us8, ((const kg_u4*)xr)[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = T::to_f(h[e]); ss += f * f; }
        }
    } else {
        for (int i = threadIdx.x; i < K; i += blockDim.x) { float f = T::to_f(xr[i]); ss += f * f; }
    }
    ss = kg_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float rstd = rsqrtf(ss / (float)K + eps);
    if (vec) {
        for (int i = threadIdx.x; i < (K >> 3); i += blockDim.x) {
            const kg_us8 h = __builtin_bit_cast(kg_us8, ((const kg_u4*)xr)[i]);
            const kg_us8 g = __builtin_bit_cast(kg_us8, ((const kg_u4*)w)[i]);
            kg_us8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = T::from_f(T::to_f(g[e]) * kg_round<DT>(T::to_f(h[e]) * rstd));
            ((kg_u4*)orow)[i] = __builtin_bit_cast(kg_u4, o);
        }
    } else {
        for (int i = threadIdx.x; i < K; i += blockDim.x) {
            float f = kg_round<DT>(T::to_f(xr[i]) * rstd);
            orow[i] = T::from_f(T::to_f(w[i]) * f);
        }
    }
}
int kg_launch_rmsnorm(const KgDyn* dyn, const void* x, int x_stride, const void* w, float eps, int K, void* out, int out_stride, int dtype, hipStream_t st) {
    if (dtype == KG_F16)
        hipLaunchKernelGGL(kg_rmsnorm_kernel<KG_F16>, dim3(KG_GEMM_ROWS), dim3(256), 0, st, dyn, (const unsigned short*)x, x_stride, (const unsigned short*)w, eps, K, (unsigned short*)out, out_stride);
    else
        hipLaunchKernelGGL(kg_rmsnorm_kernel<KG_BF16>, dim3(KG_GEMM_ROWS), dim3(256), 0, st, dyn, (const unsigned short*)x, x_stride, (const unsigned short*)w, eps, K, (unsigned short*)out, out_stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- lm_head partial merge: (max, sum exp(l - max), argmax with lowest-index ties) ----
struct HeadRec { float mx, sum; int idx; };
__device__ __forceinline__ void rec_merge(HeadRec& a, const HeadRec& b) {
    const float M = fmaxf(a.mx, b.mx);
    const float sa = (a.mx == -INFINITY) ? 0.f : a.sum * expf(a.mx - M);
    const float sb = (b.mx == -INFINITY) ? 0.f : b.sum * expf(b.mx - M);
    const bool take_b = (b.mx > a.mx) || (b.mx == a.mx && b.idx < a.idx);
    a.idx = take_b ? b.idx : a.idx;
    a.mx = M;
    a.sum = sa + sb;
}
// all 256 threads of the block call this; result valid in every thread
__device__ HeadRec head_reduce_row(const float* part, int nblocks, float* s_mx, float* s_sum, int* s_idx) {
    HeadRec r; r.mx = -INFINITY; r.sum = 0.f; r.idx = 0x7fffffff;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        const kg_f4 v = *(const kg_f4*)(part + (size_t)b * 4);      // one record: (max, sum, argmax bits, -)
        HeadRec o; o.mx = v[0]; o.sum = v[1]; o.idx = __float_as_int(v[2]);
        rec_merge(r, o);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        HeadRec t; t.mx = __shfl_xor(r.mx, o, 64); t.sum = __shfl_xor(r.sum, o, 64); t.idx = __shfl_xor(r.idx, o, 64);
        rec_merge(r, t);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_mx[wave] = r.mx; s_sum[wave] = r.sum; s_idx[wave] = r.idx; }
    __syncthreads();
    HeadRec f; f.mx = s_mx[0]; f.sum = s_sum[0]; f.idx = s_idx[0];
    for (int w = 1; w < nw; ++w) { HeadRec t; t.mx = s_mx[w]; t.sum = s_sum[w]; t.idx = s_idx[w]; rec_merge(f, t); }
    return f;
}

__global__ void kg_head_finalize_kernel(const KgDyn* dyn, const float* head_part, int stride, int nblocks, int* tok_out, float* pmax_out) {
    __shared__ float s_mx[8], s_sum[8]; __shared__ int s_idx[8];
    if (dyn->active == 0) return;
    const int M = dyn->n_rows;
    for (int m = 0; m < M; ++m) {
        HeadRec r = head_reduce_row(head_part + (size_t)m * stride * 4, nblocks, s_mx, s_sum, s_idx);
        if (threadIdx.x == 0) { tok_out[m] = r.idx; pmax_out[m] = 1.0f / r.sum; }
    }
}
int kg_launch_head_finalize(const KgDyn* dyn, const float* head_part, int head_part_stride, int nblocks, int max_rows,
                            int* tok_out, float* pmax_out, hipStream_t st) {
    (void)max_rows;
    hipLaunchKernelGGL(kg_head_finalize_kernel, dim3(1), dim3(256), 0, st, dyn, head_part, head_part_stride, nblocks, tok_out, pmax_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------
// Loop state machine.  Flags: KS_ALIVE = the round runs; KS_DRAFT_ON = the adapter +
// head of the coming step run; dyn_shallow.active = the shallow pass of the coming
// step runs.  A step's shallow pass runs iff the previous step's adapter ran
// (the reference breaks before the adapter call, inference_kangaroo.py:74-75).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double kg_thresh(const int* st) {
    unsigned long long b = ((unsigned long long)(unsigned)st[KS_THRESH_HI] << 32) | (unsigned)st[KS_THRESH_LO];
    return __builtin_bit_cast(double, b);
}

// h[:] = embed[tok][:]: the input row of the next shallow pass (all threads of the block)
__device__ __forceinline__ void kg_gather_embed(const KgLoopBufs& b, int tok) {
    const int n16 = b.d * b.elt / 16;
    const kg_u4* src = (const kg_u4*)b.embed + (size_t)tok * n16;
    kg_u4* dst = (kg_u4*)b.h;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
}

// Start-of-round state for the round that begins at start index s (:46-49): run by one thread, at the end
// of the previous round's accept kernel (and once after the prefill).  Also presets the verify pass's
// rows (positions s, s+1, ...; one row until a draft step adds more).
__device__ __forceinline__ void kg_round_begin_state(const KgLoopBufs& b, int* st, int s, int stop) {
    const int steps = st[KS_STEPS];
    const int alive = (stop == 0) && (s < st[KS_MAX_INFER] - 1 - steps);             // :46
    st[KS_ALIVE] = alive;
    st[KS_END] = s + 1;                                                              // :49
    st[KS_DRAFTED] = 0;
    st[KS_DRAFT_ON] = alive && (steps > 0);
    b.dyn_shallow->active = alive;
    b.dyn_shallow->n_rows = 1;
    b.dyn_shallow->pos[0] = s;                                                       // token at end_index-1
    KgDyn* dv = b.dyn_verify;                                                        // rows s .. e-1 (:86-89)
    dv->active = alive;
    dv->n_rows = 1;
#pragma unroll
    for (int i = 0; i < KG_MAX_ROWS; ++i) dv->pos[i] = s + i;
}
__global__ void kg_round_begin_kernel(KgLoopBufs b) {
    int* st = b.state;
    const int s = st[KS_START];
    kg_gather_embed(b, b.tokens[s]);
    if (threadIdx.x != 0) return;
    kg_round_begin_state(b, st, s, st[KS_STOP]);
}

// after the shallow pass of step 0: adapter rows and positions (:55-61, :70-71); later steps get theirs
// from the previous step's kg_step_end_kernel
__global__ void kg_step_mid_kernel(KgLoopBufs b, int step) {
    int* st = b.state;
    const int on = st[KS_DRAFT_ON];
    const int s = st[KS_START], e = st[KS_END], alen = st[KS_ADAPTER_LEN];
    const bool patch = (step == 0) && (alen < e - 1);
    if (threadIdx.x == 0) {
        KgDyn* d = b.dyn_adapter;
        d->active = on;
        if (patch) { d->n_rows = 2; d->pos[0] = s - 1; d->pos[1] = s; }
        else { d->n_rows = 1; d->pos[0] = e - 1; }
    }
    if (step == 0 && on) {
        // adapter_in rows: [prev_exit, exited[0]] when the adapter cache misses one row, else [exited[0]]
        const int n16 = b.d * b.elt / 16;
        const kg_u4* ex = (const kg_u4*)b.exited;
        const kg_u4* pv = (const kg_u4*)b.prev_exit;
        kg_u4* ai = (kg_u4*)b.adapter_in;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) {
            if (patch) { ai[i] = pv[i]; ai[n16 + i] = ex[i]; }
            else ai[i] = ex[i];
        }
    }
}

// after the draft head of `step`: token, confidence, next-step switches (:79-83, :74)
__global__ void kg_step_end_kernel(KgLoopBufs b, int step, const float* head_part, int nblocks) {
    __shared__ float s_mx[8], s_sum[8]; __shared__ int s_idx[8];
    int* st = b.state;
    const int on = st[KS_DRAFT_ON];          // uniform
    if (!on) {
        if (threadIdx.x == 0) { b.dyn_shallow->active = 0; b.dyn_adapter->active = 0; }
        return;
    }
    HeadRec r = head_reduce_row(head_part, nblocks, s_mx, s_sum, s_idx);
    kg_gather_embed(b, r.idx);                                // the next shallow pass runs on this token
    if (threadIdx.x == 0) {
        const int e = st[KS_END];
        const float score = 1.0f / r.sum;                     // softmax max = exp(0) / sum exp(l - max)
        b.tokens[e] = r.idx;                                  // :80
        st[KS_END] = e + 1;                                   // :82
        st[KS_ADAPTER_LEN] = e;                               // adapter cache now covers [0, e)
        st[KS_SCORE] = __float_as_int(score);
        st[KS_DRAFTED] = st[KS_DRAFTED] + 1;
        st[KS_NDRAFT_TOTAL] = st[KS_NDRAFT_TOTAL] + 1;
        b.head_pmax[step] = score;
        // next step: its shallow pass always runs; it breaks before the adapter when it
        // is the last step or the confidence fell below the threshold
        const int steps = st[KS_STEPS];
        const bool brk = (step + 1 == steps) || ((double)score < kg_thresh(st));
        st[KS_DRAFT_ON] = brk ? 0 : 1;
        b.dyn_shallow->active = 1;
        b.dyn_shallow->n_rows = 1;
        b.dyn_shallow->pos[0] = e;                            // the token just drafted
        b.dyn_verify->n_rows = e + 1 - st[KS_START];          // the verify pass covers positions s .. e (:86-89)
        // the next step's adapter pass: one row, the token just drafted (:60, :77)
        KgDyn* da = b.dyn_adapter;
        da->active = brk ? 0 : 1;
        da->n_rows = 1;
        da->pos[0] = e;
    }
}

// Top-k verification (evaluation/inference_kangaroo_MTBench.py:377-389): rank[i] = how many verify
// logits of row i are larger than the logit of the token drafted for position s+1+i.  The drafted
// token is among the k largest iff rank < k.  One block per row; logits are the stored 16-bit rows.
template <int DT>
__global__ void kg_topk_rank_kernel(KgLoopBufs b, const unsigned short* logits, int vocab, int* rank) {
    typedef KgT<DT> T;
    __shared__ int s_cnt[4];
    const int* st = b.state;
    if (st[KS_ALIVE] == 0) return;
    const int s = st[KS_START], q = st[KS_END] - s;
    const int i = blockIdx.x;
    if (i >= q - 1) return;                              // the last row has no drafted successor
    const unsigned short* row = logits + (size_t)i * vocab;
    const float lt = T::to_f(row[b.tokens[s + 1 + i]]);
    int cnt = 0;
    for (int v = threadIdx.x; v < vocab; v += blockDim.x) cnt += (T::to_f(row[v]) > lt) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[i] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}


// Top-p verification (evaluation/inference_kangaroo_MTBench.py:356-374).  Row r of the verify logits,
// divided by the temperature and soft-maxed, is sorted by falling probability; its nucleus is every
// token whose predecessors in that order carry a total probability <= p (the first token always).
// member[r * KG_MAX_ROWS + j] = 1 when the token drafted for position s+1+j lies in row r's nucleus:
// the reference tests a drafted token against the nuclei of ALL rows of the round pooled together
// (masked_select flattens them, :365), so the accept kernel takes the OR over r.  One block per row;
// equal probabilities are ordered by token id, as a stable descending sort orders them.
template <int DT>
__global__ void kg_topp_member_kernel(KgLoopBufs b, const unsigned short* logits, int vocab, float temperature, float top_p, int* member) {
    typedef KgT<DT> T;
    constexpr int MAXJ = KG_MAX_ROWS - 1;
    __shared__ float s_red[4][MAXJ + 2];
    const int* st = b.state;
    if (st[KS_ALIVE] == 0) return;
    const int s = st[KS_START], q = st[KS_END] - s;
    const int r = blockIdx.x;
    if (r >= q) return;
    const int nj = q - 1;                                // drafted tokens of this round
    const unsigned short* row = logits + (size_t)r * vocab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int v = tid; v < vocab; v += blockDim.x) mx = fmaxf(mx, T::to_f(row[v]) / temperature);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) s_red[wave][0] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0][0], s_red[1][0]), fmaxf(s_red[2][0], s_red[3][0]));
    __syncthreads();
    float lt[MAXJ]; int it[MAXJ]; float mass[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        it[j] = j < nj ? b.tokens[s + 1 + j] : 0;
        lt[j] = T::to_f(row[it[j]]);
        mass[j] = 0.f;
    }
    float z = 0.f;
    for (int v = tid; v < vocab; v += blockDim.x) {
        const float l = T::to_f(row[v]);
        const float e = __expf(l / temperature - mx);
        z += e;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) mass[j] += (l > lt[j] || (l == lt[j] && v < it[j])) ? e : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        z += __shfl_xor(z, o, 64);
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) mass[j] += __shfl_xor(mass[j], o, 64);
    }
    if (lane == 0) {
        s_red[wave][MAXJ + 1] = z;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) s_red[wave][1 + j] = mass[j];
    }
    __syncthreads();
    if (tid < nj) {
        const float zz = s_red[0][MAXJ + 1] + s_red[1][MAXJ + 1] + s_red[2][MAXJ + 1] + s_red[3][MAXJ + 1];
        const float m = s_red[0][1 + tid] + s_red[1][1 + tid] + s_red[2][1 + tid] + s_red[3][1 + tid];
        member[r * KG_MAX_ROWS + tid] = (m == 0.f || m / zz <= top_p) ? 1 : 0;
    }
}

// "Typical" verification (evaluation/inference_kangaroo_MTBench.py:293-314).  For verify row r:
// p = softmax(logits / temperature), H = -sum p log(p + 1e-5), thr = min(epsilon, exp(-H) * delta);
// typ[2r] = 1 when max p > thr (:305), typ[2r + 1] = float bits of thr (the reference rebinds its draft
// threshold to it, :301).  One block per row.
template <int DT>
__global__ void kg_typical_kernel(KgLoopBufs b, const unsigned short* logits, int vocab, float temperature, float epsilon, float delta, int* typ) {
    typedef KgT<DT> T;
    __shared__ float s_red[4];
    const int* st = b.state;
    if (st[KS_ALIVE] == 0) return;
    const int s = st[KS_START], q = st[KS_END] - s;
    const int r = blockIdx.x;
    if (r >= q) return;
    const unsigned short* row = logits + (size_t)r * vocab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, t) : v + t; }
        __syncthreads();
        if (lane == 0) s_red[wave] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) : (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    };
    float mx = -INFINITY;
    for (int v = tid; v < vocab; v += blockDim.x) mx = fmaxf(mx, T::to_f(row[v]) / temperature);
    mx = block_reduce(mx, true);
    float z = 0.f;
    for (int v = tid; v < vocab; v += blockDim.x) z += expf(T::to_f(row[v]) / temperature - mx);
    z = block_reduce(z, false);
    float h = 0.f;
    for (int v = tid; v < vocab; v += blockDim.x) {
        const float p = expf(T::to_f(row[v]) / temperature - mx) / z;
        h += p * logf(p + 1e-5f);
    }
    h = -block_reduce(h, false);
    if (tid == 0) {
        const float thr = fminf(epsilon, expf(-h) * delta);
        typ[2 * r] = (1.0f / z > thr) ? 1 : 0;
        typ[2 * r + 1] = __float_as_int(thr);
    }
}

// accept / reject, cache rollback by length (:91-119).  top_k == -1: the typical rule (see kg_typical_kernel);
// top_k == 1: the greedy rule of :97; top_k > 1:
// a drafted token survives while rank[i] < top_k (kg_topk_rank_kernel); top_k == 0: while it lies in the
// pooled top-p nuclei, rank + KG_MAX_ROWS holding kg_topp_member_kernel's table.
__global__ void kg_accept_kernel(KgLoopBufs b, const float* head_part, int stride, int nblocks, const int* rank, int top_k) {
    __shared__ int s_tok[KG_MAX_ROWS];
    __shared__ int s_next;
    int* st = b.state;
    if (st[KS_ALIVE] == 0) return;
    const int s = st[KS_START], e = st[KS_END];
    const int q = e - s;
    // argmax of every verify row (:92): wave w reduces rows w, w + 4, ...; the records of all its rows
    // are requested before any is used (one memory round trip for the whole kernel instead of one per row)
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        constexpr int RW = KG_MAX_ROWS / 4, PL = 4;             // rows per wave, records per lane held at once
        kg_f4 rec[RW][PL];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int m = wave + 4 * j;
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                const int blk = lane + 64 * k;
                rec[j][k] = (m < q && blk < nblocks) ? *(const kg_f4*)(head_part + ((size_t)m * stride + blk) * 4) : kg_f4{-INFINITY, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int m = wave + 4 * j;
            if (m >= q) continue;                               // wave-uniform
            HeadRec r; r.mx = -INFINITY; r.sum = 0.f; r.idx = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                if (lane + 64 * k < nblocks) { HeadRec o; o.mx = rec[j][k][0]; o.sum = rec[j][k][1]; o.idx = __float_as_int(rec[j][k][2]); rec_merge(r, o); }
            }
            for (int blk = lane + 64 * PL; blk < nblocks; blk += 64) {
                const kg_f4 v = *(const kg_f4*)(head_part + ((size_t)m * stride + blk) * 4);
                HeadRec o; o.mx = v[0]; o.sum = v[1]; o.idx = __float_as_int(v[2]);
                rec_merge(r, o);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                HeadRec t; t.mx = __shfl_xor(r.mx, o, 64); t.sum = __shfl_xor(r.sum, o, 64); t.idx = __shfl_xor(r.idx, o, 64);
                rec_merge(r, t);
            }
            if (lane == 0) { s_tok[m] = r.idx; b.head_tok[m] = r.idx; }
        }
    }
    __syncthreads();
    // keep the last early-exit row for the adapter's missing-cache patch of the next round (:58)
    {
        const int n16 = b.d * b.elt / 16;
        const kg_u4* ex = (const kg_u4*)b.exited + (size_t)(q - 1) * n16;
        kg_u4* pv = (kg_u4*)b.prev_exit;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) pv[i] = ex[i];
    }
    if (threadIdx.x == 0 && top_k == -1) {
        // typical rule (MTBench :294-314): rank = kg_typical_kernel's table.  start_index moves on by one at every
        // row that ends a run of matches, whether its posterior test passes or not; no token is written; a failed
        // test or an eos stops the generation; later rows are compared against the shifted start_index.
        const int eos = st[KS_EOS], max_infer = st[KS_MAX_INFER];
        int snew = s, stop = 0, thr_bits = 0, have_thr = 0;
        for (int i = 0; i < q; ++i) {
            const int o = s_tok[i];
            if (i == q - 1 || o == eos || o != b.tokens[snew + 1 + i]) {                  // :295
                thr_bits = rank[2 * i + 1]; have_thr = 1;                                 // :301
                snew += 1;                                                                // :307 / :309
                if (rank[2 * i] == 0) stop = 1;                                           // :310
                if (o == eos) stop = 1;                                                   // :312
                if (stop || snew >= max_infer) break;                                     // :313
            }
        }
        if (have_thr) {
            // the draft loop's confidence threshold is that fp32 value from now on (:301 rebinds the name used at :258)
            const double d = (double)__int_as_float(thr_bits);
            const unsigned long long u = __builtin_bit_cast(unsigned long long, d);
            st[KS_THRESH_LO] = (int)(unsigned)(u & 0xffffffffu);
            st[KS_THRESH_HI] = (int)(unsigned)(u >> 32);
        }
        s_next = b.tokens[snew];
        const int r = st[KS_ROUND];
        st[KS_HDR + 2 * r] = snew - s;
        st[KS_HDR + 2 * r + 1] = q - 1;
        st[KS_ROUND] = r + 1;
        st[KS_START] = snew;
        st[KS_STOP] = stop;
        st[KS_Q_PREV] = q;
        if (st[KS_ADAPTER_LEN] > snew) st[KS_ADAPTER_LEN] = snew;
        kg_round_begin_state(b, st, snew, stop);
    }
    if (threadIdx.x == 0 && top_k != -1) {
        const int eos = st[KS_EOS];
        int snew = s, stop = 0;
        for (int i = 0; i < q; ++i) {
            const int o = s_tok[i];
            bool rejected = true;
            if (i < q - 1) {
                if (top_k > 1) rejected = rank[i] >= top_k;
                else if (top_k == 1) rejected = o != b.tokens[s + 1 + i];
                else {
                    int in = 0;
                    for (int r = 0; r < q; ++r) in |= rank[KG_MAX_ROWS + r * KG_MAX_ROWS + i];
                    rejected = in == 0;
                }
            }
            if (i == q - 1 || o == eos || rejected) {                                 // :97
                b.tokens[s + 1 + i] = o;                                              // :98
                snew = s + 1 + i;                                                     // :99
                stop = (o == eos);                                                    // :100
                s_next = o;
                break;
            }
        }
        const int r = st[KS_ROUND];
        st[KS_HDR + 2 * r] = snew - s;                                                // :104
        st[KS_HDR + 2 * r + 1] = q - 1;
        st[KS_ROUND] = r + 1;
        st[KS_START] = snew;
        st[KS_STOP] = stop;
        st[KS_Q_PREV] = q;
        // base caches: slot == position, everything at slots >= snew is dead and gets
        // overwritten; the adapter cache is tracked by length (:114-118)
        if (st[KS_ADAPTER_LEN] > snew) st[KS_ADAPTER_LEN] = snew;
        kg_round_begin_state(b, st, snew, stop);              // the next round's :46-49
    }
    __syncthreads();
    kg_gather_embed(b, s_next);                               // the next round's first shallow pass runs on this token
}

// ---- vanilla greedy decode: one token per graph replay ----
// state for the token at start index s: run by one thread, after the prefill and at the end of every step
__device__ __forceinline__ void kg_vanilla_begin_state(const KgLoopBufs& b, int* st, int s, int stop) {
    const int alive = (stop == 0) && (s + 1 < st[KS_MAX_INFER]);
    st[KS_ALIVE] = alive;
    b.dyn_shallow->active = alive;
    b.dyn_shallow->n_rows = 1;
    b.dyn_shallow->pos[0] = s;
}
__global__ void kg_vanilla_begin_kernel(KgLoopBufs b) {
    int* st = b.state;
    const int s = st[KS_START];
    kg_gather_embed(b, b.tokens[s]);
    if (threadIdx.x != 0) return;
    kg_vanilla_begin_state(b, st, s, st[KS_STOP]);
}
__global__ void kg_vanilla_end_kernel(KgLoopBufs b, const float* head_part, int nblocks) {
    __shared__ float s_mx[8], s_sum[8]; __shared__ int s_idx[8];
    int* st = b.state;
    if (st[KS_ALIVE] == 0) return;
    HeadRec r = head_reduce_row(head_part, nblocks, s_mx, s_sum, s_idx);
    kg_gather_embed(b, r.idx);                                // input row of the next step
    if (threadIdx.x == 0) {
        const int s = st[KS_START];
        const int stop = (r.idx == st[KS_EOS]);
        b.tokens[s + 1] = r.idx;
        st[KS_START] = s + 1;
        st[KS_STOP] = stop;
        st[KS_ROUND] = st[KS_ROUND] + 1;
        kg_vanilla_begin_state(b, st, s + 1, stop);
    }
}

// Per-slice softmax records of stored logit rows, in the layout the lm_head kernel leaves them in
// ([row][stride][4] = max, sum exp(l - max), argmax bits, -): block (b, m) covers the vocabulary slice
// [b * per, (b + 1) * per) of row m.  Used by the kernel-level control entry points, which start from logits.
template <int DT>
__global__ void kg_logit_records_kernel(const unsigned short* logits, int vocab, int per, float* head_part, int stride) {
    typedef KgT<DT> T;
    __shared__ float s_mx[4], s_sum[4]; __shared__ int s_idx[4];
    const int b = blockIdx.x, m = blockIdx.y;
    const unsigned short* row = logits + (size_t)m * vocab;
    const int v0 = b * per, v1 = min(vocab, v0 + per);
    HeadRec r; r.mx = -INFINITY; r.sum = 0.f; r.idx = 0x7fffffff;
    for (int v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
        HeadRec o; o.mx = T::to_f(row[v]); o.sum = 1.f; o.idx = v;
        rec_merge(r, o);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        HeadRec t; t.mx = __shfl_xor(r.mx, o, 64); t.sum = __shfl_xor(r.sum, o, 64); t.idx = __shfl_xor(r.idx, o, 64);
        rec_merge(r, t);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mx[wave] = r.mx; s_sum[wave] = r.sum; s_idx[wave] = r.idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        HeadRec f; f.mx = s_mx[0]; f.sum = s_sum[0]; f.idx = s_idx[0];
        for (int w = 1; w < 4; ++w) { HeadRec t; t.mx = s_mx[w]; t.sum = s_sum[w]; t.idx = s_idx[w]; rec_merge(f, t); }
        float* hp = head_part + ((size_t)m * stride + b) * 4;
        hp[0] = f.mx; hp[1] = f.sum; hp[2] = __int_as_float(f.idx); hp[3] = 0.f;
    }
}
int kg_launch_logit_records(const void* logits, int vocab, int rows, int nblocks, float* head_part, int stride, int dtype, hipStream_t st) {
    if (rows < 1 || rows > KG_MAX_ROWS || nblocks < 1 || nblocks > stride) return -1;
    const int per = (vocab + nblocks - 1) / nblocks;
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_logit_records_kernel<KG_F16>, dim3(nblocks, rows), dim3(256), 0, st, (const unsigned short*)logits, vocab, per, head_part, stride);
    else hipLaunchKernelGGL(kg_logit_records_kernel<KG_BF16>, dim3(nblocks, rows), dim3(256), 0, st, (const unsigned short*)logits, vocab, per, head_part, stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#define KG_L1(k, ...) do { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, st, __VA_ARGS__); return hipGetLastError() == hipSuccess ? 0 : -2; } while (0)
int kg_launch_round_begin(const KgLoopBufs& b, hipStream_t st) { KG_L1(kg_round_begin_kernel, b); }
int kg_launch_step_mid(const KgLoopBufs& b, int step, hipStream_t st) { KG_L1(kg_step_mid_kernel, b, step); }
int kg_launch_step_end(const KgLoopBufs& b, int step, const float* head_part, int nblocks, hipStream_t st) { KG_L1(kg_step_end_kernel, b, step, head_part, nblocks); }
int kg_launch_accept(const KgLoopBufs& b, const float* head_part, int stride, int nblocks, const int* rank, int top_k, hipStream_t st) { KG_L1(kg_accept_kernel, b, head_part, stride, nblocks, rank, top_k); }
int kg_launch_topp_member(const KgLoopBufs& b, const void* logits, int vocab, float temperature, float top_p, int* member, int dtype, hipStream_t st) {
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_topp_member_kernel<KG_F16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, temperature, top_p, member);
    else hipLaunchKernelGGL(kg_topp_member_kernel<KG_BF16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, temperature, top_p, member);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int kg_launch_typical(const KgLoopBufs& b, const void* logits, int vocab, float temperature, float epsilon, float delta, int* typ, int dtype, hipStream_t st) {
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_typical_kernel<KG_F16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, temperature, epsilon, delta, typ);
    else hipLaunchKernelGGL(kg_typical_kernel<KG_BF16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, temperature, epsilon, delta, typ);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int kg_launch_topk_rank(const KgLoopBufs& b, const void* logits, int vocab, int* rank, int dtype, hipStream_t st) {
    if (dtype == KG_F16) hipLaunchKernelGGL(kg_topk_rank_kernel<KG_F16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, rank);
    else hipLaunchKernelGGL(kg_topk_rank_kernel<KG_BF16>, dim3(KG_MAX_ROWS), dim3(256), 0, st, b, (const unsigned short*)logits, vocab, rank);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int kg_launch_vanilla_begin(const KgLoopBufs& b, hipStream_t st) { KG_L1(kg_vanilla_begin_kernel, b); }
int kg_launch_vanilla_end(const KgLoopBufs& b, const float* head_part, int nblocks, hipStream_t st) { KG_L1(kg_vanilla_end_kernel, b, head_part, nblocks); }
