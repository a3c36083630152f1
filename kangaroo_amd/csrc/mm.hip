// Dense matrix product for the compute-bound parts of the path (gfx950): C[M][N] = A[M][K] . B[N][K]^T, both operands
// with the contraction index contiguous ("x W^T", the form of torch.nn.functional.linear; kangaroo/adapter.py:170-172,222,
// train.py:265-273).  16-bit inputs (fp16 / bf16), fp32 accumulation on v_mfma_f32_16x16x32, 16-bit or fp32 output.
//
// A block owns a BM x BN tile of C and walks K in steps of 64.  Per step its threads fetch BM x 64 of A and BN x 64 of
// B with 16-byte buffer loads in whole 128-byte lines (eight consecutive lanes cover one row segment), park them in
// registers while the MFMAs of the current step run, and write them to the other LDS buffer afterwards: one block
// barrier per step.  LDS rows are 128 bytes; the 16-byte chunk index is XORed with (row >> 1) & 7 so that the 16 rows
// x 16 bytes of a fragment read (ds_read_b128) fall on distinct banks.  Wave (wr, wc) computes a (BM / WM) x (BN / WN)
// part of the tile, the B-matrix rows as the MFMA's A operand so that a lane ends up with four CONSECUTIVE output
// columns of one row (an 8-byte store).
// Two shapes: 256 x 256 with 8 waves of 128 x 64 (one block per CU: 128 KB of LDS, all 512 KB of registers) for
// products with at least one tile per CU, and 128 x 128 with 4 waves of 64 x 64 (two blocks per CU) for the rest.
// What decides between them is the path from L2 to the CUs, which carries about 10 TB/s: a 128 x 128 tile moves 32 KB
// per 2.1 MFLOP step (64 flop per byte: ~670 TFLOP/s at that rate, which is what it measures), a 256 x 256 tile twice
// the flops per byte.
//
// B can also be given in the decode path's packed layout (1 KiB tiles of 16 rows x 32 k, include/kangaroo_hip.h): the
// staging loads then walk a tile's 64 pieces in storage order, one full tile per wave-level load.
// Rows / columns beyond M / N and k beyond K are read as zeros through range-checked buffer loads (the offset of a
// piece that does not exist is forced out of range), so any M, N and any K that is a multiple of 8 work.
// B can finally be given K-MAJOR, [K][N] with N contiguous (C = A . B, the form of the products that contract over the
// rows of a stored matrix: dV^T = dO^T . P, dK^T = Q^T . dS): its tiles are staged as [k][n] rows and the fragments, which
// want eight consecutive k of one n, come out of LDS through the transposing read ds_read_b64_tr_b16 (two per
// fragment: a group of 16 lanes fetches a 4 x 16 block and each lane receives one column).  32-byte chunks of a row are
// XORed with (k & 3) | ((k >> 3) & 1) << 2, which spreads the eight rows a 32-lane half reads over all banks.
// grid.z walks `batch` independent products with element strides sA / sB / sC (attention heads).
// Tiles are dealt to blocks so that the blocks resident on one XCD (block index mod 8) work on neighbouring tiles of a
// band of 8 tile rows: they share the L2 copies of that band's A rows and of the B columns they walk.
#include "kg_common.h"
#include "kg_kernels.h"
#include <type_traits>

#define MM_BK 64

__device__ __forceinline__ unsigned mm_lds_off(int row, int chunk) { return (unsigned)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); }

typedef short mm_v4s __attribute__((ext_vector_type(4)));
// silu(g) * u with a 16-bit store after every op, as the reference computes it on stored tensors (kangaroo/adapter.py:242)
template <int DT> __device__ __forceinline__ float mm_silu_mul(float g, float u) {
    const float s = kg_round<DT>(g / (1.0f + expf(-g)));
    return kg_round<DT>(s * u);
}
__device__ __forceinline__ unsigned mm_kmajor_key(int k) { return (unsigned)((k & 3) | (((k >> 3) & 1) << 2)); }

// Tiles are dealt so that the blocks that share an XCD (block index mod 8) take consecutive tiles (bijective for any tile
// count) of a band of 8 tile rows.
template <int BM, int BN>
__device__ __forceinline__ void mm_block_tile(const KgMmArgs& a, int& m0, int& n0) {
    const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN, nt = tm * tn;
    int id = blockIdx.x;
    {
        const int q = nt >> 3, r = nt & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int band = id / (8 * tn), in_band = id - band * 8 * tn;
    const int band_rows = min(8, tm - band * 8);
    m0 = (band * 8 + in_band % band_rows) * BM; n0 = (in_band / band_rows) * BN;
}

// The epilogue of a tile: the wave's part goes through its own LDS region and leaves in whole rows (a lane's accumulators
// are 4 columns of 16 different rows: stored directly, every wave-level store would touch 16 rows with 32 bytes each).
// Every wave of the block must be past its last LDS read of the pipeline, and no write to LDS may be in flight.
template <int DT, int BM, int BN, int WM, int WN, bool OUT_F32>
__device__ __forceinline__ void mm_epilogue(const KgMmArgs& a, kg_f4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int m0, int n0, int zo, int zi,
                                            int wave, int lane, size_t c_off = 0) {
    typedef KgT<DT> T;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    const int wr = wave / WN, wc = wave % WN, fr = lane & 15, fg = lane >> 4;
    const size_t cb = (size_t)zo * a.sC + (size_t)zi * a.sC2 + c_off;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;              // the wave's part of the tile
    constexpr int ESZ = OUT_F32 ? 4 : 2;
    constexpr int ROWB = WCOLS * ESZ + 16;                       // padded row of the staging image (keeps 16-byte alignment, staggers banks)
    static_assert((size_t)WM * WN * 16 * ROWB <= 2 * (size_t)(BM + BN) * MM_BK * 2, "staging image fits the pipeline's LDS");
    char* stg = smem + wave * 16 * ROWB;                         // 16 rows at a time
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // all waves are past the last step's barrier; a wave only touches its own region from here on
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            kg_f4 v = acc[i][j];
            if (a.alpha != 1.0f) v = v * a.alpha;
            char* dst = stg + fr * ROWB + (16 * j + 4 * fg) * ESZ;
            if (OUT_F32) *(kg_f4*)dst = v;
            else *(unsigned long long*)dst = (unsigned long long)T::from_f(v[0]) | ((unsigned long long)T::from_f(v[1]) << 16) | ((unsigned long long)T::from_f(v[2]) << 32) | ((unsigned long long)T::from_f(v[3]) << 48);
        }
        // 16 rows x WCOLS elements: 16-byte pieces, consecutive lanes along a row
        constexpr int PPR = WCOLS * ESZ / 16, NP = 16 * PPR;     // pieces per row, pieces of the 16 rows
#pragma unroll
        for (int c = lane; c < NP; c += 64) {
            const int r = c / PPR, pc = c - r * PPR;
            const int m = m0 + wr * WROWS + 16 * i + r, n = n0 + wc * WCOLS + pc * (16 / ESZ);
            const kg_u4 v = *(const kg_u4*)(stg + r * ROWB + pc * 16);
            if (!OUT_F32 && a.epi == MM_EPI_SILU_MUL) {
                // the piece holds four (gate, up) pairs: four outputs at columns n / 2 ..
                if (m < a.M && n < a.N) {
                    const kg_us8 h = __builtin_bit_cast(kg_us8, v);
                    unsigned short o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = T::from_f(mm_silu_mul<DT>(T::to_f(h[2 * e]), T::to_f(h[2 * e + 1])));
                    unsigned short* out = (unsigned short*)a.C + cb + (size_t)m * a.ldc + (n >> 1);
                    if (n + 8 <= a.N && ((a.ldc | (n >> 1)) & 3) == 0) *(unsigned long long*)out = (unsigned long long)o[0] | ((unsigned long long)o[1] << 16) | ((unsigned long long)o[2] << 32) | ((unsigned long long)o[3] << 48);
                    else for (int e = 0; e < 4 && n + 2 * e + 1 < a.N; ++e) out[e] = o[e];
                }
                continue;
            }
            if (!OUT_F32 && a.epi == MM_EPI_RESID && m < a.M && n < a.N) {
                const unsigned short* rs = (const unsigned short*)a.resid + (size_t)m * a.ldr + n;
                const kg_us8 h = __builtin_bit_cast(kg_us8, v);
                unsigned short* out = (unsigned short*)a.C + cb + (size_t)m * a.ldc + n;
                if (n + 8 <= a.N && ((a.ldc | a.ldr | n) & 7) == 0) {
                    const kg_us8 r8 = __builtin_bit_cast(kg_us8, *(const kg_u4*)rs);
                    kg_us8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = T::from_f(T::to_f(r8[e]) + T::to_f(h[e]));
                    *(kg_u4*)out = __builtin_bit_cast(kg_u4, o);
                } else {
                    for (int e = 0; e < 8 && n + e < a.N; ++e) out[e] = T::from_f(T::to_f(rs[e]) + T::to_f(h[e]));
                }
                continue;
            }
            if (m < a.M && n < a.N) {
                char* out = (char*)a.C + (cb + (size_t)m * a.ldc + n) * ESZ;
                if (n + 16 / ESZ <= a.N && (((a.ldc | n) * ESZ) & 15) == 0 && ((cb * ESZ) & 15) == 0) *(kg_u4*)out = v;
                else {
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
                    for (int e = 0; e < 16 / ESZ && n + e < a.N; ++e) {
                        if (OUT_F32) ((float*)out)[e] = __builtin_bit_cast(float, w[e]);
                        else ((unsigned short*)out)[e] = (unsigned short)(w[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
    }
}

// BMODE 0: B [N][K] row-major; 1: B in the packed tile layout; 2: B [K][N] (K-major)
template <int DT, int BM, int BN, int WM, int WN, int BMODE, bool OUT_F32>
__global__ void __launch_bounds__(WM * WN * 64) kg_mm_kernel(const KgMmArgs a) {
    constexpr bool PACKED_B = (BMODE == 1), KMAJOR_B = (BMODE == 2);
    typedef KgT<DT> T;
    constexpr int MM_BM = BM, NTHR = WM * WN * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;            // MFMA tiles of a wave
    constexpr int PA = BM * 8 / NTHR, PB = BN * 8 / NTHR;          // 16-byte pieces per thread and step
    constexpr int A_BYTES = BM * MM_BK * 2, STAGE = (BM + BN) * MM_BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;

    // ---- tile of this block ----
    int m0, n0;
    mm_block_tile<BM, BN>(a, m0, n0);
    if (a.causal == 1 && n0 > m0 + MM_BM - 1) return;                  // output tile entirely above the diagonal: never read
    const int zo = (int)blockIdx.z / a.batch_inner, zi = (int)blockIdx.z - zo * a.batch_inner;
    const unsigned short* Ab = (const unsigned short*)a.A + (size_t)zo * a.sA + (size_t)zi * a.sA2;
    const unsigned short* Bb = (const unsigned short*)a.B + (size_t)zo * a.sB + (size_t)(zi / a.b_inner_div) * a.sB2;
    const size_t a_bytes = ((size_t)(a.M - 1) * a.lda + a.K) * 2;
    const size_t b_bytes = PACKED_B ? (size_t)((a.N + 15) / 16) * (a.K / 32) * 1024 : (KMAJOR_B ? ((size_t)(a.K - 1) * a.ldb + a.N) * 2 : ((size_t)(a.N - 1) * a.ldb + a.K) * 2);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(a_bytes > 0x7fffffffu ? 0x7fffffffu : a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(b_bytes > 0x7fffffffu ? 0x7fffffffu : b_bytes), 0x00020000);

    // K range of this tile: a causal product with a lower-triangular A (P . V) or a lower-triangular C (dV = P^T . dO
    // handled by the caller through transposes) only needs k up to the tile's last row
    const int k_end = (a.causal == 2) ? min(a.K, m0 + MM_BM) : a.K;
    // an upper-triangular A: nothing before the tile's first row; a K-major B whose row k ends at column k: nothing before
    // the tile's first column
    const int kt0 = (a.causal == 3) ? m0 / MM_BK : ((a.causal == 4) ? n0 / MM_BK : 0);
    const int KT = (k_end + MM_BK - 1) / MM_BK - kt0;

    // ---- staging: 4 pieces of 16 bytes per thread and operand ----
    // piece c = tid + NTHR * i of an operand's step: row c >> 3, chunk c & 7 (a packed B: the 64 pieces of a 1 KiB tile
    // in storage order); everything about a piece but its k0 is fixed, and cheap enough to recompute from c
    unsigned a_src[PA], b_src[PB];        // byte offset of the piece at k0 = 0 (0xFFFFFFFF: row out of range)
    unsigned a_dst[PA], b_dst[PB];        // its place in an LDS buffer
    int a_k[PA], b_k[PB];                 // its first k inside a step
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int c = tid + NTHR * i, row = c >> 3, ch = c & 7;
        a_k[i] = 8 * ch; a_dst[i] = mm_lds_off(row, ch);
        a_src[i] = (m0 + row < a.M) ? (unsigned)(((size_t)(m0 + row) * a.lda + 8 * ch) * 2) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int c = tid + NTHR * i;
        if (PACKED_B) {
            const int r = c & 15, g = (c >> 4) & 3, ktl = (c >> 6) & 1, rgl = c >> 7, row = rgl * 16 + r, ch = ktl * 4 + g;
            b_k[i] = 8 * ch; b_dst[i] = A_BYTES + mm_lds_off(row, ch);
            b_src[i] = (n0 + row < a.N) ? (unsigned)((((size_t)((n0 >> 4) + rgl) * (a.K >> 5) + ktl) * 512 + g * 128 + r * 8) * 2) : 0xFFFFFFFFu;
        } else if (KMAJOR_B) {
            const int k = c / (BN / 8), c16 = c % (BN / 8);              // row k of the step, 16-byte piece c16 of its BN columns
            b_k[i] = k; b_dst[i] = A_BYTES + (unsigned)(k * (BN * 2) + ((((unsigned)c16 >> 1) ^ mm_kmajor_key(k)) << 5) + ((c16 & 1) << 4));
            b_src[i] = (n0 + 8 * c16 < a.N) ? (unsigned)(((size_t)k * a.ldb + n0 + 8 * c16) * 2) : 0xFFFFFFFFu;
        } else {
            const int row = c >> 3, ch = c & 7;
            b_k[i] = 8 * ch; b_dst[i] = A_BYTES + mm_lds_off(row, ch);
            b_src[i] = (n0 + row < a.N) ? (unsigned)(((size_t)(n0 + row) * a.ldb + 8 * ch) * 2) : 0xFFFFFFFFu;
        }
    }
    kg_u4 sa[PA], sb[PB];
    auto stage_load = [&](int kt) {
        const int k0 = (kt0 + kt) * MM_BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = (k0 + a_k[i] < k_end) && a_src[i] != 0xFFFFFFFFu;
            sa[i] = __builtin_amdgcn_raw_buffer_load_b128(ar, ok ? a_src[i] + (unsigned)(k0 * 2) : 0xFFFFFFFFu, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const bool ok = (k0 + b_k[i] < k_end) && b_src[i] != 0xFFFFFFFFu;
            const unsigned boff = PACKED_B ? b_src[i] + (unsigned)((k0 >> 5) * 1024) : (KMAJOR_B ? b_src[i] + (unsigned)((size_t)k0 * a.ldb * 2) : b_src[i] + (unsigned)(k0 * 2));
            sb[i] = __builtin_amdgcn_raw_buffer_load_b128(br, ok ? boff : 0xFFFFFFFFu, 0, 0);
        }
    };
    auto stage_store = [&](int buf) {
        char* l = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) *(kg_u4*)(l + a_dst[i]) = sa[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *(kg_u4*)(l + b_dst[i]) = sb[i];
    };

    kg_f4 acc[TM][TN];                    // [m tile][n tile]: lane holds C[m = 16 mi + fr][n = 16 ni + 4 fg + e]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = kg_f4{0.f, 0.f, 0.f, 0.f};

    if (KT > 0) { stage_load(0); stage_store(0); }
    if (KT > 1) stage_load(1);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        // The registers hold step kt + 1, requested a whole step ago: park it in the other LDS buffer (last read in step
        // kt - 1, which every wave has left) and request step kt + 2 before this step's MFMAs, which hide both.
        if (kt + 1 < KT) stage_store((kt + 1) & 1);
        if (kt + 2 < KT) stage_load(kt + 2);
        const char* la = smem + (kt & 1) * STAGE + wr * (BM / WM) * 128;
        const char* lb = smem + (kt & 1) * STAGE + A_BYTES + wc * (BN / WN) * 128;
#pragma unroll
        for (int ks = 0; ks < MM_BK / 32; ++ks) {
            kg_u4 fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (KMAJOR_B) {
                    // rows k = 32 ks + 8 fg + (0..3 | 4..7), columns 16 j .. 16 j + 15 of this wave's part: lane 4q + p of a 16-lane
                    // group addresses row q, columns 4p .. 4p + 3 and receives column (lane & 15) of the four rows
                    const int q = fr >> 2, pp = fr & 3, kr = 32 * ks + 8 * fg + q;
                    const int c32 = (wc * (BN / WN) + 16 * j) >> 4;
                    const char* base = smem + (kt & 1) * STAGE + A_BYTES;
                    const mm_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mm_v4s*)(base + kr * (BN * 2) + ((((unsigned)c32) ^ mm_kmajor_key(kr)) << 5) + 8 * pp));
                    const mm_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mm_v4s*)(base + (kr + 4) * (BN * 2) + ((((unsigned)c32) ^ mm_kmajor_key(kr + 4)) << 5) + 8 * pp));
                    fb[j].x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16); fb[j].y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                    fb[j].z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16); fb[j].w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
                } else {
                    fb[j] = *(const kg_u4*)(lb + mm_lds_off(16 * j + fr, 4 * ks + fg));
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const kg_u4*)(la + mm_lds_off(16 * i + fr, 4 * ks + fg));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(fb[j], fa[i], acc[i][j]);       // D[n = 4 fg + e][m = fr]
        }
        __syncthreads();
    }

    mm_epilogue<DT, BM, BN, WM, WN, OUT_F32>(a, acc, smem, m0, n0, zo, zi, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// The 256 x 256 tile again, built for the case that decides the large products: one block per CU, two waves per SIMD.
// A K step of 64 is staged as four half tiles of 16 KB (128 rows x 128 bytes) by buffer loads that write LDS directly
// (no register stop, no ds_write pass): the LDS destination of such a load is the wave's base + lane * 16, so the image is
// linear and the XOR swizzle of the fragment reads sits on the SOURCE address of each lane instead.  Rows are assigned to
// half tiles by the phase that reads them: A_0 / A_1 hold the first / second 64 rows of both wave rows' parts, B_0 / B_1 the
// first / second 32 columns of all four wave columns' parts.  A step is four phases, one 64 x 32 quadrant of the wave's
// 128 x 64 part each (16 MFMAs between two block barriers):
//     phase 0  reads A_0, B_0   quadrant (0,0)        phase 2  reads A_1   quadrant (1,1)
//     phase 1  reads B_1        quadrant (0,1)        phase 3  -            quadrant (1,0)   (B_0 stays in registers)
// so every half tile is read in exactly one phase and can be refilled two phases later.  Every phase requests the half tile
// that is read six phases on and then waits until at most four half tiles (8 loads per thread) are outstanding: what the
// NEXT phase reads has then landed, and the barrier that follows makes that true of every wave's share.  The loads are
// never drained inside the loop.  The two wave rows run one barrier apart, so that on every SIMD one wave is in its MFMA
// cluster while the other issues its LDS reads and loads.  Half tiles past the last step are requested out of range
// (nothing is fetched) so that the counts stay uniform.
// ---------------------------------------------------------------------------------------------------------------------
#define MM8_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MM8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int DT, int BMODE, bool OUT_F32>
__global__ void __launch_bounds__(512) kg_mm8_kernel(const KgMmArgs a) {
    static_assert(BMODE == 0 || BMODE == 1, "row-major or packed B");
    typedef KgT<DT> T;
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
    constexpr int HALF = 128 * 128, BUF = 4 * HALF;                 // bytes: a half tile, one step's four half tiles
    constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    int m0, n0;
    mm_block_tile<BM, BN>(a, m0, n0);
    if (a.causal == 1 && n0 > m0 + BM - 1) return;
    const int zo = (int)blockIdx.z / a.batch_inner, zi = (int)blockIdx.z - zo * a.batch_inner;
    const unsigned short* Ab = (const unsigned short*)a.A + (size_t)zo * a.sA + (size_t)zi * a.sA2;
    const unsigned short* Bb = (const unsigned short*)a.B + (size_t)zo * a.sB + (size_t)(zi / a.b_inner_div) * a.sB2;
    const size_t a_bytes = ((size_t)(a.M - 1) * a.lda + a.K) * 2;
    const size_t b_bytes = BMODE == 1 ? (size_t)((a.N + 15) / 16) * (a.K / 32) * 1024 : ((size_t)(a.N - 1) * a.ldb + a.K) * 2;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(a_bytes > 0x7fffffffu ? 0x7fffffffu : a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(b_bytes > 0x7fffffffu ? 0x7fffffffu : b_bytes), 0x00020000);
    // a K range of whole steps per grid.y (ksplit > 1: no causal mode)
    const int k_per = a.ksplit > 1 ? ((a.K + MM_BK - 1) / MM_BK + a.ksplit - 1) / a.ksplit * MM_BK : a.K;
    const int k_lo = a.ksplit > 1 ? (int)blockIdx.y * k_per : 0;
    const int k_end = a.ksplit > 1 ? min(a.K, k_lo + k_per) : ((a.causal == 2) ? min(a.K, m0 + BM) : a.K);
    const int kt0 = a.ksplit > 1 ? k_lo / MM_BK : ((a.causal == 3) ? m0 / MM_BK : 0);
    const int KT = (k_end + MM_BK - 1) / MM_BK - kt0;

    // ---- sources: load g (0, 1) of a half tile fills LDS rows 8 (8 g + wave) .. + 7, lane l row l >> 3, 16-byte slot l & 7 ----
    unsigned a_src[2][2], b_src[2][2];      // [half][g]: byte offset at k0 = 0 (0xFFFFFFFF: row out of range)
    int a_k[2], b_k[2];                     // first k of the lane's piece inside a step
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int q = (8 * g + wave) * 8 + (lane >> 3), ch = (lane & 7) ^ ((q >> 1) & 7);
        a_k[g] = 8 * ch; b_k[g] = BMODE == 1 ? 32 * (wave & 1) + 8 * (lane >> 4) : 8 * ch;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = m0 + (q >> 6) * 128 + h * 64 + (q & 63);
            a_src[h][g] = row < a.M ? (unsigned)(((size_t)row * a.lda + 8 * ch) * 2) : 0xFFFFFFFFu;
            if (BMODE == 1) {
                // packed B: the half tile's LDS image is the 16 packed tiles themselves (8 row groups x 2 k tiles of 1 KiB), load g of
                // wave w copies tile 8 g + w whole: 1 KiB of consecutive memory, and a fragment read is 1 KiB of consecutive LDS
                const int idx = 8 * g + wave, rgl = idx >> 1, n = n0 + (rgl >> 1) * 64 + h * 32 + (rgl & 1) * 16;
                b_src[h][g] = n < a.N ? (unsigned)((((size_t)(n >> 4) * (a.K >> 5) + (idx & 1)) * 512 + lane * 8) * 2) : 0xFFFFFFFFu;
            } else {
                const int n = n0 + (q >> 5) * 64 + h * 32 + (q & 31);
                b_src[h][g] = n < a.N ? (unsigned)(((size_t)n * a.ldb + 8 * ch) * 2) : 0xFFFFFFFFu;
            }
        }
    }
    // half tile `off` (compile-time LDS offset) of step u: two loads per thread, whatever u
    auto request = [&](char* dst, const unsigned (&src)[2], const __amdgpu_buffer_rsrc_t& rs, bool is_b, int u) {
        const int k0 = (kt0 + u) * MM_BK;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // a step past the last one: the advance alone throws the offset out of range (no branch on the uniform test)
            const unsigned adv = u < KT ? ((BMODE == 1 && is_b) ? (unsigned)((k0 >> 5) * 1024) : (unsigned)(k0 * 2)) : 0x80000000u;
            const bool ok = k0 + (is_b ? b_k[g] : a_k[g]) < k_end && src[g] < 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (8 * g + wave) * 1024), 16,
                                                     (ok ? src[g] : 0x7FFFFFFFu) + adv, 0, 0, 0);
        }
    };

    kg_f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = kg_f4{0.f, 0.f, 0.f, 0.f};
    kg_u4 fa[4][2], fb0[2][2], fb1[2][2];   // [tile][k half of the step]

    auto read_a = [&](const char* half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *(const kg_u4*)(half + mm_lds_off(wr * 64 + 16 * i + fr, 4 * ks + fg));
    };
    auto read_b = [&](const char* half, kg_u4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fb[j][ks] = BMODE == 1 ? *(const kg_u4*)(half + ((wc * 2 + j) * 2 + ks) * 1024 + lane * 16)
                                       : *(const kg_u4*)(half + mm_lds_off(wc * 32 + 16 * j + fr, 4 * ks + fg));
    };
    auto cluster = [&](auto mh_c, auto nh_c, const kg_u4 (&fb)[2][2]) {
        constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value;
        MM8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 * MH + i][2 * NH + j] = T::mfma(fb[j][ks], fa[i][ks], acc[4 * MH + i][2 * NH + j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // one step out of buffer B (compile time); t is the step's number
    auto step = [&](auto b_c, int t) {
        constexpr int B = decltype(b_c)::value;
        char* cur = smem + B * BUF;
        char* oth = smem + (B ^ 1) * BUF;
        // phase 0
        read_b(cur + OFF_B0, fb0); __builtin_amdgcn_sched_barrier(0); read_a(cur + OFF_A0);
        request(oth + OFF_B1, b_src[1], br, true, t + 1);
        MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();
        cluster(I0(), I0(), fb0);
        __builtin_amdgcn_s_barrier();
        // phase 1
        read_b(cur + OFF_B1, fb1);
        request(oth + OFF_A1, a_src[1], ar, false, t + 1);
        MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();
        cluster(I0(), I1(), fb1);
        __builtin_amdgcn_s_barrier();
        // phase 2
        read_a(cur + OFF_A1);
        request(cur + OFF_A0, a_src[0], ar, false, t + 2);
        MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();
        cluster(I1(), I1(), fb1);
        __builtin_amdgcn_s_barrier();
        // phase 3
        request(cur + OFF_B0, b_src[0], br, true, t + 2);
        MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();
        cluster(I1(), I0(), fb0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- six half tiles ahead: all of step 0, A_0 and B_0 of step 1 ----
    request(smem + OFF_A0, a_src[0], ar, false, 0);
    request(smem + OFF_B0, b_src[0], br, true, 0);
    request(smem + OFF_B1, b_src[1], br, true, 0);
    request(smem + OFF_A1, a_src[1], ar, false, 0);
    request(smem + BUF + OFF_A0, a_src[0], ar, false, 1);
    request(smem + BUF + OFF_B0, b_src[0], br, true, 1);
    MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();        // A_0 and B_0 of step 0 are in LDS, every wave's share
    if (wr == 1) __builtin_amdgcn_s_barrier();            // the second wave row runs one barrier behind
    int t = 0;
    for (; t + 1 < KT; t += 2) { step(I0(), t); step(I1(), t + 1); }
    if (t < KT) step(I0(), t);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    MM8_WAIT_VM(0);                                       // the requests past the last step
    __builtin_amdgcn_s_barrier();
    mm_epilogue<DT, BM, BN, WM, WN, OUT_F32>(a, acc, smem, m0, n0, zo, zi, wave, lane, a.ksplit > 1 ? (size_t)blockIdx.y * (size_t)a.sCk : 0);
}

// The same construction for a 128 x 256 tile (products with too few rows to fill the chip with 256 x 256 tiles: the prompt
// pass).  A wave computes 64 x 64; a K step is three half tiles (A, B_0, B_1) and two phases of 16 MFMAs (all four row
// tiles x the first / second 32 columns); three LDS buffers of 48 KB, so a half tile is requested four phases before it
// is read: phase 0 of step t requests A and B_0 of step t + 2, phase 1 its B_1, into the buffer step t - 1 was read from
// two phases earlier.  Outstanding after the request, at most: phase 0 five half tiles (B_1 of this step has landed),
// phase 1 four (A and B_0 of the next step have landed).
template <int DT, int BMODE, bool OUT_F32>
__global__ void __launch_bounds__(512) kg_mm8h_kernel(const KgMmArgs a) {
    static_assert(BMODE == 0 || BMODE == 1, "row-major or packed B");
    typedef KgT<DT> T;
    constexpr int BM = 128, BN = 256, WM = 2, WN = 4;
    constexpr int HALF = 128 * 128, BUF = 3 * HALF;
    constexpr int OFF_A = 0, OFF_B0 = HALF, OFF_B1 = 2 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    int m0, n0;
    mm_block_tile<BM, BN>(a, m0, n0);
    if (a.causal == 1 && n0 > m0 + BM - 1) return;
    const int zo = (int)blockIdx.z / a.batch_inner, zi = (int)blockIdx.z - zo * a.batch_inner;
    const unsigned short* Ab = (const unsigned short*)a.A + (size_t)zo * a.sA + (size_t)zi * a.sA2;
    const unsigned short* Bb = (const unsigned short*)a.B + (size_t)zo * a.sB + (size_t)(zi / a.b_inner_div) * a.sB2;
    const size_t a_bytes = ((size_t)(a.M - 1) * a.lda + a.K) * 2;
    const size_t b_bytes = BMODE == 1 ? (size_t)((a.N + 15) / 16) * (a.K / 32) * 1024 : ((size_t)(a.N - 1) * a.ldb + a.K) * 2;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(a_bytes > 0x7fffffffu ? 0x7fffffffu : a_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(b_bytes > 0x7fffffffu ? 0x7fffffffu : b_bytes), 0x00020000);
    // a K range of whole steps per grid.y (ksplit > 1: no causal mode)
    const int k_per = a.ksplit > 1 ? ((a.K + MM_BK - 1) / MM_BK + a.ksplit - 1) / a.ksplit * MM_BK : a.K;
    const int k_lo = a.ksplit > 1 ? (int)blockIdx.y * k_per : 0;
    const int k_end = a.ksplit > 1 ? min(a.K, k_lo + k_per) : ((a.causal == 2) ? min(a.K, m0 + BM) : a.K);
    const int kt0 = a.ksplit > 1 ? k_lo / MM_BK : ((a.causal == 3) ? m0 / MM_BK : 0);
    const int KT = (k_end + MM_BK - 1) / MM_BK - kt0;

    unsigned a_src[2], b_src[2][2];         // [g], [half][g]
    int a_k[2], b_k[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int q = (8 * g + wave) * 8 + (lane >> 3), ch = (lane & 7) ^ ((q >> 1) & 7);
        a_k[g] = 8 * ch; b_k[g] = BMODE == 1 ? 32 * (wave & 1) + 8 * (lane >> 4) : 8 * ch;
        a_src[g] = m0 + q < a.M ? (unsigned)(((size_t)(m0 + q) * a.lda + 8 * ch) * 2) : 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (BMODE == 1) {
                // packed B: the half tile's LDS image is the 16 packed tiles themselves (8 row groups x 2 k tiles of 1 KiB), load g of
                // wave w copies tile 8 g + w whole: 1 KiB of consecutive memory, and a fragment read is 1 KiB of consecutive LDS
                const int idx = 8 * g + wave, rgl = idx >> 1, n = n0 + (rgl >> 1) * 64 + h * 32 + (rgl & 1) * 16;
                b_src[h][g] = n < a.N ? (unsigned)((((size_t)(n >> 4) * (a.K >> 5) + (idx & 1)) * 512 + lane * 8) * 2) : 0xFFFFFFFFu;
            } else {
                const int n = n0 + (q >> 5) * 64 + h * 32 + (q & 31);
                b_src[h][g] = n < a.N ? (unsigned)(((size_t)n * a.ldb + 8 * ch) * 2) : 0xFFFFFFFFu;
            }
        }
    }
    auto request = [&](char* dst, const unsigned (&src)[2], const __amdgpu_buffer_rsrc_t& rs, bool is_b, int u) {
        const int k0 = (kt0 + u) * MM_BK;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const unsigned adv = u < KT ? ((BMODE == 1 && is_b) ? (unsigned)((k0 >> 5) * 1024) : (unsigned)(k0 * 2)) : 0x80000000u;
            const bool ok = k0 + (is_b ? b_k[g] : a_k[g]) < k_end && src[g] < 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (8 * g + wave) * 1024), 16,
                                                     (ok ? src[g] : 0x7FFFFFFFu) + adv, 0, 0, 0);
        }
    };

    kg_f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = kg_f4{0.f, 0.f, 0.f, 0.f};
    kg_u4 fa[4][2], fb0[2][2], fb1[2][2];

    auto read_a = [&](const char* half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *(const kg_u4*)(half + mm_lds_off(wr * 64 + 16 * i + fr, 4 * ks + fg));
    };
    auto read_b = [&](const char* half, kg_u4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fb[j][ks] = BMODE == 1 ? *(const kg_u4*)(half + ((wc * 2 + j) * 2 + ks) * 1024 + lane * 16)
                                       : *(const kg_u4*)(half + mm_lds_off(wc * 32 + 16 * j + fr, 4 * ks + fg));
    };
    auto cluster = [&](auto nh_c, const kg_u4 (&fb)[2][2]) {
        constexpr int NH = decltype(nh_c)::value;
        MM8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 * NH + j] = T::mfma(fb[j][ks], fa[i][ks], acc[i][2 * NH + j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    auto step = [&](auto b_c, int t) {
        constexpr int B = decltype(b_c)::value;
        char* cur = smem + B * BUF;
        char* nx2 = smem + ((B + 2) % 3) * BUF;
        // phase 0
        read_b(cur + OFF_B0, fb0); __builtin_amdgcn_sched_barrier(0); read_a(cur + OFF_A);
        request(nx2 + OFF_A, a_src, ar, false, t + 2);
        request(nx2 + OFF_B0, b_src[0], br, true, t + 2);
        MM8_WAIT_VM(10); __builtin_amdgcn_s_barrier();
        cluster(I0(), fb0);
        __builtin_amdgcn_s_barrier();
        // phase 1
        read_b(cur + OFF_B1, fb1);
        request(nx2 + OFF_B1, b_src[1], br, true, t + 2);
        MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();
        cluster(I1(), fb1);
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int u = 0; u < 2; ++u) {
        request(smem + u * BUF + OFF_A, a_src, ar, false, u);
        request(smem + u * BUF + OFF_B0, b_src[0], br, true, u);
        request(smem + u * BUF + OFF_B1, b_src[1], br, true, u);
    }
    MM8_WAIT_VM(8); __builtin_amdgcn_s_barrier();        // A and B_0 of step 0 are in LDS, every wave's share
    if (wr == 1) __builtin_amdgcn_s_barrier();
    int t = 0;
    for (; t + 2 < KT; t += 3) { step(I0(), t); step(I1(), t + 1); step(I2(), t + 2); }
    if (t < KT) {
        step(I0(), t);
        if (t + 1 < KT) step(I1(), t + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    MM8_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    mm_epilogue<DT, BM, BN, WM, WN, OUT_F32>(a, acc, smem, m0, n0, zo, zi, wave, lane, a.ksplit > 1 ? (size_t)blockIdx.y * (size_t)a.sCk : 0);
}

static int g_mm_ncu = 256;
// flop rate of a CU busy with one 256 x 256 direct-to-LDS tile / one 128 x 256 tile, in the units of mm_launch's model
#define MM_RATE_BIG8 1.75
#define MM_RATE_HALF 1.4
template <int DT, int BM, int BN, int WM, int WN, int BMODE>
static void mm_launch_tile(const KgMmArgs& a, hipStream_t st) {
    const size_t smem = 2 * (size_t)(BM + BN) * MM_BK * 2;
    const dim3 grid(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN), 1, a.batch), block(WM * WN * 64);
    if (a.out_f32) hipLaunchKernelGGL((kg_mm_kernel<DT, BM, BN, WM, WN, BMODE, true>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((kg_mm_kernel<DT, BM, BN, WM, WN, BMODE, false>), grid, block, smem, st, a);
}
template <int DT, int BMODE>
static void mm_launch(const KgMmArgs& a, hipStream_t st) {
    // Rounds of blocks x time of a round.  A CU holds one 256 x 256 block, which runs ~1.4 times as fast per flop as one
    // of the two 128 x 128 blocks it holds instead (they share the CU, and a small tile moves twice the bytes per flop
    // through the L2-to-CU path): a round of large tiles costs 65536 / 1.4, a round of small ones 16384 / 0.5.  Measured:
    // 8192 x 4096 x 4096 is 296 us against 410 us (model 0.71), 640 x 22016 x 4096 232 against 167 us.
    const double t_big = (double)(((long)((a.M + 255) / 256) * ((a.N + 255) / 256) * a.batch + g_mm_ncu - 1) / g_mm_ncu) * (65536.0 / 1.4);
    const double t_small = (double)(((long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch + 2 * g_mm_ncu - 1) / (2 * g_mm_ncu)) * (16384.0 / 0.5);
    const double t_half = (double)(((long)((a.M + 127) / 128) * ((a.N + 255) / 256) * a.batch + g_mm_ncu - 1) / g_mm_ncu) * (32768.0 / MM_RATE_HALF);
    int pick = a.tile;                                    // 1: 128 x 128; 2 / 3: 256 x 256 staged through registers / loads write LDS; 4: 128 x 256
    if (pick == 0) {
        const double t_b = BMODE == 2 ? t_big : t_big * (1.4 / MM_RATE_BIG8);
        pick = (t_b <= t_small && (BMODE == 2 || t_b <= t_half)) ? 3 : ((BMODE != 2 && t_half < t_small) ? 4 : 1);
    }
    if (a.ksplit > 1 && pick < 3) pick = 4;
    if (BMODE == 2 && pick >= 3) pick = pick == 3 ? 2 : 1;   // a K-major B has no direct-to-LDS form (the image is not lane-linear)
    if (pick >= 3) {
        constexpr int BM8 = BMODE == 2 ? 0 : BMODE;
        const int bm = pick == 3 ? 256 : 128;
        const dim3 grid(((a.M + bm - 1) / bm) * ((a.N + 255) / 256), a.ksplit > 1 ? a.ksplit : 1, a.batch), block(512);
        if (pick == 3) {
            if (a.out_f32) hipLaunchKernelGGL((kg_mm8_kernel<DT, BM8, true>), grid, block, 131072, st, a);
            else hipLaunchKernelGGL((kg_mm8_kernel<DT, BM8, false>), grid, block, 131072, st, a);
        } else {
            if (a.out_f32) hipLaunchKernelGGL((kg_mm8h_kernel<DT, BM8, true>), grid, block, 147456, st, a);
            else hipLaunchKernelGGL((kg_mm8h_kernel<DT, BM8, false>), grid, block, 147456, st, a);
        }
    } else if (pick == 2) mm_launch_tile<DT, 256, 256, 2, 4, BMODE>(a, st);
    else mm_launch_tile<DT, 128, 128, 2, 2, BMODE>(a, st);
}

int kg_launch_mm(const KgMmArgs& a, int dtype, hipStream_t st) {
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.K < 8 || (a.K & 7) || a.batch < 1 || a.batch > 65535) return -1;
    if ((a.lda & 7) || (a.b_packed != 1 && (a.ldb & 7)) || (a.sA & 7) || (a.sB & 7)) return -1;      // 16-byte pieces
    if (a.b_packed < 0 || a.b_packed > 2) return -1;
    if (a.b_packed == 1 && ((a.K & 31) || a.batch != 1)) return -1;
    if (a.b_packed == 2 && (a.N & 7)) return -1;
    if (a.causal < 0 || a.causal > 4 || (a.causal == 4 && a.b_packed != 2)) return -1;
    if (a.batch_inner < 1 || a.batch % a.batch_inner || a.b_inner_div < 1 || (a.sA2 & 7) || (a.sB2 & 7)) return -1;
    if (a.tile < 0 || a.tile > 4) return -1;
    {   // one product's operands are addressed with 32-bit byte offsets
        const unsigned long long ea = ((unsigned long long)(a.M - 1) * a.lda + a.K) * 2;
        const unsigned long long eb = a.b_packed == 1 ? (unsigned long long)((a.N + 15) / 16) * (a.K / 32) * 1024
                                    : (a.b_packed == 2 ? ((unsigned long long)(a.K - 1) * a.ldb + a.N) * 2 : ((unsigned long long)(a.N - 1) * a.ldb + a.K) * 2);
        if (ea >= 0x7fffffffull || eb >= 0x7fffffffull) return -1;
    }
    if (a.ksplit < 0 || a.ksplit > 64 || (a.ksplit > 1 && (!a.out_f32 || a.epi || a.causal || a.batch != 1 || a.b_packed == 2 || a.tile == 1 || a.tile == 2 || a.sCk < (long long)a.M * a.ldc))) return -1;
    if (a.epi < 0 || a.epi > 2 || (a.epi && (a.out_f32 || a.batch != 1)) || (a.epi == MM_EPI_RESID && !a.resid) || (a.epi == MM_EPI_SILU_MUL && (a.N & 1))) return -1;
    if (dtype == KG_F16) { if (a.b_packed == 1) mm_launch<KG_F16, 1>(a, st); else if (a.b_packed == 2) mm_launch<KG_F16, 2>(a, st); else mm_launch<KG_F16, 0>(a, st); }
    else { if (a.b_packed == 1) mm_launch<KG_BF16, 1>(a, st); else if (a.b_packed == 2) mm_launch<KG_BF16, 2>(a, st); else mm_launch<KG_BF16, 0>(a, st); }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int DT, int P, bool F> static void mm_attr() {
    (void)hipFuncSetAttribute((const void*)kg_mm_kernel<DT, 128, 128, 2, 2, P, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * MM_BK * 2);
    (void)hipFuncSetAttribute((const void*)kg_mm_kernel<DT, 256, 256, 2, 4, P, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * MM_BK * 2);
    if (P != 2) (void)hipFuncSetAttribute((const void*)kg_mm8_kernel<DT, (P == 2 ? 0 : P), F>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (P != 2) (void)hipFuncSetAttribute((const void*)kg_mm8h_kernel<DT, (P == 2 ? 0 : P), F>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
}
void kg_mm_init() {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) g_mm_ncu = ncu;
    mm_attr<KG_F16, 0, false>(); mm_attr<KG_F16, 0, true>(); mm_attr<KG_F16, 1, false>(); mm_attr<KG_F16, 1, true>(); mm_attr<KG_F16, 2, false>(); mm_attr<KG_F16, 2, true>();
    mm_attr<KG_BF16, 0, false>(); mm_attr<KG_BF16, 0, true>(); mm_attr<KG_BF16, 1, false>(); mm_attr<KG_BF16, 1, true>(); mm_attr<KG_BF16, 2, false>(); mm_attr<KG_BF16, 2, true>();
    (void)hipGetLastError();
}

// out[b][c][r] = in[b][r][c] for 16-bit elements: 64 x 64 tiles through LDS, both sides in whole lines.  WIDE: both matrices in
// 16-byte pieces (R, Cn, the strides and the batch strides multiples of 8, bases 16-byte aligned): a thread loads 8 consecutive
// columns of a row and stores 8 consecutive rows of a column, gathered from LDS rows 65 words apart (conflict-free).
template <bool WIDE>
__global__ void __launch_bounds__(256) kg_transpose16_kernel(const unsigned short* in, unsigned short* out, int R, int Cn, long long ld_in, long long ld_out,
                                                             long long s_in, long long s_out) {
    __shared__ unsigned short t[64][130];                  // pitch 65 words
    const unsigned short* ib = in + (size_t)blockIdx.z * s_in;
    unsigned short* ob = out + (size_t)blockIdx.z * s_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    if (WIDE) {
        const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = r0 + rr + 32 * p, c = c0 + 8 * ch;
            kg_u4 v = kg_u4{0u, 0u, 0u, 0u};
            if (r < R && c < Cn) v = *(const kg_u4*)(ib + (size_t)r * ld_in + c);
            const kg_us8 h = __builtin_bit_cast(kg_us8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[8 * ch + e][2 * (rr + 32 * p)] = h[e];        // t[column][row], one element per word
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = c0 + rr + 32 * p, r = r0 + 8 * ch;
            if (c < Cn && r < R) {
                kg_us8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = t[rr + 32 * p][2 * (8 * ch + e)];
                *(kg_u4*)(ob + (size_t)c * ld_out + r) = __builtin_bit_cast(kg_u4, o);
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        t[i][tx] = (r < R && c < Cn) ? ib[(size_t)r * ld_in + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cn && r < R) ob[(size_t)c * ld_out + r] = t[tx][i];
    }
}
int kg_launch_transpose16(const void* in, void* out, int R, int Cn, long long ld_in, long long ld_out, int batch, long long s_in, long long s_out, hipStream_t st) {
    if (!in || !out || R < 1 || Cn < 1 || batch < 1 || batch > 65535) return -1;
    const dim3 grid((Cn + 63) / 64, (R + 63) / 64, batch);
    const bool wide = ((R | Cn | ld_in | ld_out | s_in | s_out) & 7) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (wide) hipLaunchKernelGGL(kg_transpose16_kernel<true>, grid, dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, R, Cn, ld_in, ld_out, s_in, s_out);
    else hipLaunchKernelGGL(kg_transpose16_kernel<false>, grid, dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, R, Cn, ld_in, ld_out, s_in, s_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
