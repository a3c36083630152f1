// Shared device-side definitions for the Kangaroo decode kernels (gfx950 / CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KG_MAX_ROWS 16          // rows one M-row pass can carry (verify needs steps+1 <= 7)
#define KG_GEMM_ROWS 128        // rows one prefill (GEMM) pass can carry
#define KG_TILE_ROWS 16         // weight rows per packed tile
#define KG_TILE_K 32            // k elements per packed tile
#define KG_TILE_ELEMS 512       // 16 x 32 elements = 1 KiB of 16-bit weights

enum { KG_F16 = 0, KG_BF16 = 1 };

// Per-pass dynamic parameters.  They live in device memory so that a captured graph
// can be replayed while positions, row counts and the on/off switch change.
struct KgDyn {
    int active;                 // 0: every kernel of the pass returns at once
    int n_rows;                 // valid rows (1..KG_MAX_ROWS)
    int pos[KG_MAX_ROWS];       // position id == KV-cache slot of each row
    int pad[14];
};

// Decode-loop state (one int32 array in device memory).  Indices:
enum {
    KS_START = 0,       // start_index s: tokens [0..s] are final
    KS_END = 1,         // end_index e: next free token slot
    KS_ADAPTER_LEN = 2, // adapter KV length
    KS_STOP = 3,        // eos was produced
    KS_ALIVE = 4,       // this round runs (loop condition held at round start)
    KS_DRAFT_ON = 5,    // the next adapter/head step runs
    KS_ROUND = 6,       // verify rounds finished
    KS_CTX = 7,         // context length T0
    KS_MAX_INFER = 8,   // min(max_length, T0 + max_new_tokens)
    KS_STEPS = 9,       // SPECULATIVE_DECODING_STEPS
    KS_EOS = 10,
    KS_THRESH_LO = 11,  // threshold as a double: low / high 32 bits
    KS_THRESH_HI = 17,
    KS_SCORE = 12,      // float bits of the last draft confidence
    KS_DRAFTED = 13,    // adapter steps taken this round
    KS_Q_PREV = 14,     // rows of the previous verify pass
    KS_VANILLA_POS = 15,// vanilla decode: position of the token being fed
    KS_NDRAFT_TOTAL = 16,
    KS_HDR = 32,        // accept log starts here: [KS_HDR + 2*r] = accept, [+1] = drafted
};

typedef _Float16 kg_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 kg_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 kg_b2 __attribute__((ext_vector_type(2)));
typedef __bf16 kg_b8 __attribute__((ext_vector_type(8)));
typedef float kg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int kg_u4 __attribute__((ext_vector_type(4)));
typedef unsigned short kg_us8 __attribute__((ext_vector_type(8)));

template <int DT> struct KgT;
template <> struct KgT<KG_F16> {
    static __device__ __forceinline__ float to_f(unsigned short u) {
        _Float16 h = __builtin_bit_cast(_Float16, u);
        return (float)h;
    }
    static __device__ __forceinline__ unsigned short from_f(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(kg_h2, a), __builtin_bit_cast(kg_h2, b), c, false);
    }
    static __device__ __forceinline__ kg_f4 mfma(kg_u4 a, kg_u4 b, kg_f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(kg_h8, a), __builtin_bit_cast(kg_h8, b), c, 0, 0, 0);
    }
    // sum of the squares of 8 stored values (fp32 accumulation)
    static __device__ __forceinline__ float sum_sq8(kg_u4 v) {
        float s = dot2(v.x, v.x, 0.f);
        s = dot2(v.y, v.y, s); s = dot2(v.z, v.z, s);
        return dot2(v.w, v.w, s);
    }
    // gamma * round16(x * rstd), rounded to 16 bits again.  The product of two fp16 values is exact in fp32, so the
    // packed fp16 multiply (one rounding to nearest even) gives the same bits as multiplying in fp32 and rounding.
    static __device__ __forceinline__ kg_u4 rms_apply(kg_u4 xv, kg_u4 gv, float rstd) {
        const kg_h8 x = __builtin_bit_cast(kg_h8, xv), g = __builtin_bit_cast(kg_h8, gv);
        kg_h8 y;
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (_Float16)((float)x[j] * rstd);
        return __builtin_bit_cast(kg_u4, y * g);
    }
};
template <> struct KgT<KG_BF16> {
    static __device__ __forceinline__ float to_f(unsigned short u) {
        return __builtin_bit_cast(float, ((unsigned int)u) << 16);
    }
    static __device__ __forceinline__ unsigned short from_f(float f) {
        __bf16 h = (__bf16)f;   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(kg_b2, a), __builtin_bit_cast(kg_b2, b), c, false);
    }
    static __device__ __forceinline__ kg_f4 mfma(kg_u4 a, kg_u4 b, kg_f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(kg_b8, a), __builtin_bit_cast(kg_b8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float sum_sq8(kg_u4 v) {
        float s = dot2(v.x, v.x, 0.f);
        s = dot2(v.y, v.y, s); s = dot2(v.z, v.z, s);
        return dot2(v.w, v.w, s);
    }
    static __device__ __forceinline__ kg_u4 rms_apply(kg_u4 xv, kg_u4 gv, float rstd) {
        const kg_us8 x = __builtin_bit_cast(kg_us8, xv), g = __builtin_bit_cast(kg_us8, gv);
        kg_us8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f(to_f(g[j]) * to_f(from_f(to_f(x[j]) * rstd)));
        return __builtin_bit_cast(kg_u4, o);
    }
};

// two fp32 values rounded (to nearest even) to the storage type and packed, low half first: one v_cvt_pk_f16_f32 /
// v_cvt_pk_bf16_f32 on gfx950, the same bits as two from_f calls
template <int DT> __device__ __forceinline__ unsigned kg_pack2(float a, float b);
template <> __device__ __forceinline__ unsigned kg_pack2<KG_F16>(float a, float b) {
    typedef _Float16 kg_h2v __attribute__((ext_vector_type(2)));
    typedef float kg_f2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(kg_f2v{a, b}, kg_h2v));
}
template <> __device__ __forceinline__ unsigned kg_pack2<KG_BF16>(float a, float b) {
    typedef __bf16 kg_b2v __attribute__((ext_vector_type(2)));
    typedef float kg_f2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(kg_f2v{a, b}, kg_b2v));
}

// round an fp32 value to the storage type and back (the reference's per-op store)
template <int DT> __device__ __forceinline__ float kg_round(float f) { return KgT<DT>::to_f(KgT<DT>::from_f(f)); }

__device__ __forceinline__ float kg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sum without LDS traffic: four DPP steps leave every lane of a 16-lane row with the row's sum
// (quad swaps, then the mirror of each half row and of the row), four lane reads and three adds join the rows.
// Needs all 64 lanes active.  Every lane gets the same bits.
__device__ __forceinline__ float kg_wave_sum_dpp(float v) {
#define KG_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
    KG_DPP_ADD(0xB1);       // quad_perm [1,0,3,2]
    KG_DPP_ADD(0x4E);       // quad_perm [2,3,0,1]
    KG_DPP_ADD(0x141);      // row_half_mirror
    KG_DPP_ADD(0x140);      // row_mirror
#undef KG_DPP_ADD
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float kg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Grid-wide barrier of a persistent kernel whose blocks are ALL resident (grid <= CUs x
// blocks per CU).  Arrivals are spread over 8 counters (blockIdx % 8, i.e. one per XCD under
// round-robin dispatch), 128 B apart; counters only grow within a launch and must be zero at
// its start.  There is no cache maintenance here: data exchanged across the barrier has to be
// written and read device-coherently (see KG_AUX_COH / st16 in kg_linear_body.h), and arrive()
// waits until this thread's stores have been acknowledged.  A wait that spins too long raises
// *abort_flag, after which every wait falls through, so the grid always drains.
struct KgGridBar {
    unsigned* ctr;
    unsigned* abort_flag;
    unsigned done;              // barriers this block has signalled
    unsigned per_ctr;           // arrivals per counter and barrier (gridDim.x / 8)
    unsigned long long* tlog;   // optional timeline of block 0: [2*i] = phase i entered, [2*i+1] = its wait returned
    __device__ __forceinline__ void init(unsigned* counters, unsigned* abort_ptr, unsigned long long* timeline) {
        ctr = counters; abort_flag = abort_ptr; done = 0; per_ctr = gridDim.x >> 3; tlog = timeline;
    }
    // every thread of the block, after the phase's last store
    __device__ __forceinline__ void arrive() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++done;
    }
    // every thread of the block: returns once all blocks have signalled `done` barriers
    __device__ __forceinline__ void wait() {
        if (done == 0) { if (tlog && blockIdx.x == 0 && threadIdx.x == 0) tlog[1] = wall_clock64(); return; }
        if (threadIdx.x < 8) {
            const unsigned target = done * per_ctr;
            unsigned spins = 0;
            while (__hip_atomic_load(ctr + threadIdx.x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if ((++spins & 63u) == 0) {
                    if (spins > (1u << 18) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (tlog && blockIdx.x == 0 && threadIdx.x == 0) tlog[2 * done + 1] = wall_clock64();
    }
};

// streamed-once weight load: 16 B per lane, non-temporal
__device__ __forceinline__ kg_u4 kg_ldg_stream(const kg_u4* p) { return __builtin_nontemporal_load(p); }

#define KG_CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { kg_set_error(#x, e_); return -2; } } while (0)
