// Host-visible kernel launchers (internal to the shared library).
#pragma once
#include "kg_common.h"

enum { PRO_NONE = 0, PRO_RMSNORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_ROPE_KV = 2, EPI_SILU_MUL = 3, EPI_HEAD = 4, EPI_STORE_F32 = 5 };

struct KgLinArgs {
    const void* W;              // packed weights, ceil16(N) x K
    int N, K;
    const KgDyn* dyn;
    // input rows [n_rows, K], element stride x_stride
    const void* x; int x_stride;
    int x_last_row;             // VEC mode: read row dyn->n_rows-1 of x instead of row 0
    const void* norm_w; float eps;          // PRO_RMSNORM
    // EPI_STORE / EPI_RESID / EPI_SILU_MUL
    void* out; int out_stride;
    const void* resid; int resid_stride;
    // EPI_ROPE_KV
    void* q_out; void* k_cache; void* v_cache;
    const float* rope_cos; const float* rope_sin;   // [max_pos, head_dim/2]
    int head_dim, q_dim, kv_dim, t_max;
    int v_transposed;                               // v_cache layout [Hkv, hd, t_max] instead of [Hkv, t_max, hd]
    // EPI_HEAD
    float* head_part; int head_part_stride;         // [rows][head_part_stride][4]
    void* logits_out; int logits_stride;            // optional
    int gx_slot_rows;                               // set by the launcher: rows of a wave's X slot (large-K MFMA mode)
    int side_bytes;                                 // set by the launcher: LDS bytes of staged epilogue inputs
    // prefill GEMM, EPI_RESID with few row groups: K cut into ksplit ranges on grid.y, fp32 partials
    // [ksplit][KG_GEMM_ROWS][N] summed in range order by a second kernel (0 / NULL: one range)
    float* kpart; int ksplit;
    // prefill GEMM, EPI_RESID: RMSNorm (gamma out_norm_w, a.eps) of the finished output rows for the next
    // projection, written to out_norm [rows][N] (NULL: none)
    const void* out_norm_w; void* out_norm;
    // prefill GEMM, EPI_RESID with kpart: 1 = stop at the fp32 sum of the K ranges, written to out as float
    // [rows][out_stride] with no residual (a tensor-parallel shard's partial sums)
    int part_only;
};

size_t kg_linear_smem(int mfma, int max_rows, int K, int nwaves);
int kg_launch_linear(const KgLinArgs& a, int dtype, int max_rows, int pro, int epi, int nblocks, int nwaves, hipStream_t st);
void kg_linear_init();
// prefill passes: up to KG_GEMM_ROWS rows at consecutive positions dyn->pos[0] + m (see gemm.hip); EPI_HEAD is not offered
int kg_launch_gemm(const KgLinArgs& a, int dtype, int pro, int epi, hipStream_t st);
void kg_gemm_init();

struct KgAttnArgs {
    const KgDyn* dyn;
    const void* q;              // [n_rows, H*hd]
    const void* k_cache;        // [Hkv, t_max, hd]
    const void* v_cache;
    int H, Hkv, hd, t_max;
    int v_transposed;           // 1: v_cache is [Hkv, hd, t_max] and the MFMA kernel runs (hd 32/64/128, t_max % 32 == 0)
    int nsplit;                 // number of fixed key chunks the cache is cut into
    int blocks_per_head;        // MFMA kernel: blocks that share the chunks of one head (launch geometry only)
    int teams;                  // MFMA kernel: 2 = two 4-wave teams per block work on different chunks at once (geometry only)
    float* part_o;              // [nsplit][rows][H][hd]
    float* part_ml;             // [nsplit][rows][H][2]
    unsigned* counters;         // [4][H] arrival counters of the split merge; zero between launches
    void* out;                  // [n_rows, H*hd]
    int max_rows;               // upper bound of dyn->n_rows: <= 16, or <= KG_GEMM_ROWS for the MFMA kernel (16-row blocks on grid.z)
    // MFMA kernel, optional: packed weights of the projection launched next (pf_groups row groups of pf_group_bytes,
    // walked by pf_grid blocks), of which the blocks without keys pull the first pf_bytes_per_group bytes of every row
    // group into the L2 of the XCD that will read them.  pf_w == NULL: nothing is fetched.
    const void* pf_w;
    int pf_groups, pf_group_bytes, pf_grid, pf_bytes_per_group;
    int pf_delay;               // the fetching blocks first sleep this many times 64 cycles
    int pf_min_chunks;          // fetch only when at least this many chunks hold keys: a shorter pass ends before the fetch does
};
int kg_launch_attn(const KgAttnArgs& a, int dtype, hipStream_t st);
// causal attention of `rows` prompt rows at positions pos0 .. in one launch (head_dim 32 / 64 / 128, V cached transposed)
int kg_launch_attn_rows(const void* q, const void* k_cache, const void* v_cache, void* out, int rows, int pos0, int H, int Hkv, int hd, int t_max, int dtype, hipStream_t st);
void kg_attn_init();

// persistent pass kernel: a list of phases walked by resident blocks (see persist.hip)
enum { KG_PH_QKV = 0, KG_PH_ATTN = 1, KG_PH_O = 2, KG_PH_GATEUP = 3, KG_PH_DOWN = 4, KG_PH_DOWN_GX = 5, KG_PH_HEAD = 6 };
struct KgPhase {
    int kind;
    int nwaves;                 // waves of a block that stream weights in a linear phase (<= 8)
    KgLinArgs lin;
    KgAttnArgs attn;
};
size_t kg_persist_plan_linear(KgPhase& p, int max_rows, int nblocks);
size_t kg_persist_attn_smem(int hd);
void kg_persist_init();
int kg_persist_max_blocks(int dtype, size_t smem);
// tlog: optional device buffer of 2 * n_phases + 1 timestamps (100 MHz ticks) written by block 0
int kg_launch_persist(const KgPhase* phases_dev, int n_phases, unsigned* bar, int dtype, int nblocks, size_t smem, unsigned long long* tlog, hipStream_t st);

// small control / glue kernels
int kg_launch_set_dyn(KgDyn* dyn, int active, int n_rows, int pos0, hipStream_t st);
int kg_launch_zero_u32(unsigned* p, int n, hipStream_t st);
int kg_launch_set_dyn_from(KgDyn* dyn, int n_rows, const int* pos0_p, int t_max, hipStream_t st);
int kg_launch_embed(const KgDyn* dyn, const int* tokens, const void* embed, int d, void* out, int out_stride, hipStream_t st);
int kg_launch_rmsnorm(const KgDyn* dyn, const void* x, int x_stride, const void* w, float eps, int K, void* out, int out_stride, int dtype, hipStream_t st);
int kg_launch_head_finalize(const KgDyn* dyn, const float* head_part, int head_part_stride, int nblocks, int max_rows,
                            int* tok_out, float* pmax_out, hipStream_t st);

int kg_launch_logit_records(const void* logits, int vocab, int rows, int nblocks, float* head_part, int stride, int dtype, hipStream_t st);

struct KgLoopBufs {
    int* state;                 // KS_* layout
    int* tokens;                // [max_length]
    KgDyn* dyn_shallow; KgDyn* dyn_adapter; KgDyn* dyn_verify;
    void* exited;               // [steps+1][d]
    void* prev_exit;            // [d]
    void* adapter_in;           // [2][d]
    int* head_tok;              // [KG_MAX_ROWS] argmax per row of the last head pass
    float* head_pmax;           // [KG_MAX_ROWS]
    int d; int elt;
    const void* embed;          // [vocab][d] embedding table
    void* h;                    // [d] input row of the next shallow pass (the control kernels gather it)
};
int kg_launch_round_begin(const KgLoopBufs& b, hipStream_t st);
int kg_launch_step_mid(const KgLoopBufs& b, int step, hipStream_t st);        // after the shallow pass: adapter rows
int kg_launch_step_end(const KgLoopBufs& b, int step, const float* head_part, int nblocks, hipStream_t st);  // after the draft head
// rank / top_k: top-k verification (top_k == 1: greedy, rank unused; 0: top-p table; -1: typical table)
int kg_launch_accept(const KgLoopBufs& b, const float* head_part, int stride, int nblocks, const int* rank, int top_k, hipStream_t st);
int kg_launch_topp_member(const KgLoopBufs& b, const void* logits, int vocab, float temperature, float top_p, int* member, int dtype, hipStream_t st);
int kg_launch_typical(const KgLoopBufs& b, const void* logits, int vocab, float temperature, float epsilon, float delta, int* typ, int dtype, hipStream_t st);
int kg_launch_topk_rank(const KgLoopBufs& b, const void* logits, int vocab, int* rank, int dtype, hipStream_t st);
int kg_launch_vanilla_begin(const KgLoopBufs& b, hipStream_t st);
int kg_launch_vanilla_end(const KgLoopBufs& b, const float* head_part, int nblocks, hipStream_t st);

// ---- dense matrix product (mm.hip): C[M][N] = alpha * A[M][K] . B[N][K]^T, fp32 accumulation ----
struct KgMmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    long long lda, ldb, ldc;            // row strides in elements (ldb unused for a packed B)
    long long sA, sB, sC; int batch;    // grid.z independent products z = zo * batch_inner + zi: element offsets zo * sX + zi * sX2
    int batch_inner; long long sA2, sB2, sC2;   // (B: (zi / b_inner_div) * sB2, query heads that share a key / value head); batch_inner 1: z = zo
    int b_inner_div;
    int b_packed;                       // layout of B: 0 [N][K] rows, 1 the packed 16 x 32 tile layout of the decode weights, 2 [K][N] rows (K-major)
    int out_f32;                        // C is float instead of the 16-bit input type
    int causal;                         // 1: tiles of C entirely above the diagonal are skipped (not written); 2: A is lower
                                        //    triangular (row m has no entry beyond column m): K is walked up to the tile's last row;
                                        // 3: A is upper triangular: K is walked from the tile's first row on;
                                        // 4: a K-major B whose row k ends at column k: K is walked from the tile's first column on
    float alpha;
    int tile;                           // 0: chosen by size; measurements: 1: 128 x 128 tiles; 2: 256 x 256, staged through registers;
                                        //    3 / 4: 256 x 256 / 128 x 256, loads write LDS
    // 16-bit output only.  epi 1: C = round16(resid + round16(A.B^T)) (the residual add of a decoder layer, resid rows ldr
    // elements apart); epi 2: columns (2i, 2i+1) hold (gate_i, up_i) and C[m][i] = silu(gate) * up with the reference's
    // roundings (N / 2 output columns, ldc counts those)
    int epi; const void* resid; long long ldr;
    // > 1: K cut into ksplit ranges of whole 64-steps (grid.y), range r writes its fp32 partial product at C + r * sCk elements;
    // needs out_f32, no epilogue, no causal mode, batch 1, a row-major or packed B
    int ksplit; long long sCk;
};
enum { MM_EPI_STORE = 0, MM_EPI_RESID = 1, MM_EPI_SILU_MUL = 2 };
// rows of the prompt pass that runs layer by layer over the whole prompt (prefill.hip)
int kg_launch_rows_rmsnorm(const void* x, int ld, const void* w, float eps, int K, void* out, int ld_out, int rows, int dtype, hipStream_t st);
// h_out[m] = round16(resid[m] + round16(part[0][m] + part[1][m] + ...)) (ranges added in order); with w also xn[m] = the RMSNorm of
// h_out[m] times w, as kg_launch_rows_rmsnorm computes it.  h_out may alias resid.
int kg_launch_rows_reduce(const float* part, int ksplit, long long split_stride, int N, const void* resid, int ldr, void* h_out, int ldh,
                          const void* w, float eps, void* xn, int ldx, int rows, int dtype, hipStream_t st);
int kg_launch_rows_rope_kv(const void* qkv, int rows, int pos0, int head_dim, int q_dim, int kv_dim, const float* rope_cos, const float* rope_sin,
                           void* q_out, void* k_cache, void* v_cache, int t_max, int v_transposed, int dtype, hipStream_t st);
int kg_launch_rows_embed(const int* tokens, int pos0, int rows, const void* embed, int d, void* out, hipStream_t st);
int kg_launch_mm(const KgMmArgs& a, int dtype, hipStream_t st);
void kg_mm_init();
int kg_launch_transpose16(const void* in, void* out, int R, int Cn, long long ld_in, long long ld_out, int batch, long long s_in, long long s_out, hipStream_t st);

// ---- tensor-parallel exchange (tp.hip) ----
#define KG_TP_MAX 8
#define KG_TP_FLAG_BLOCKS 1024          // flags per (parity, source rank): one per block of the largest reduce launch
#define KG_TP_ROWS_CAP KG_GEMM_ROWS     // rows a slot of the exchange buffer holds
struct KgTpArgs {
    const KgDyn* dyn;
    float* xbuf;                        // this GPU's exchange buffer: [2][tp][KG_TP_ROWS_CAP][d] floats, flags [2][tp][KG_TP_FLAG_BLOCKS], control words
    float* peer[KG_TP_MAX];             // by rank: that rank's exchange buffer as mapped here (unused for local ranks)
    int tp, d;
    unsigned local_mask;                // bit r: rank r's partial is produced on this GPU
    int e, exch_per_pass;               // exchange index inside a pass (its parity picks the buffer half); exchanges per pass
    const void* resid; int resid_stride;
    void* out; int out_stride;
};
__host__ __device__ static inline size_t kg_tp_slot_floats(int d) { return (size_t)KG_TP_ROWS_CAP * d; }
__host__ __device__ static inline size_t kg_tp_data_floats(int tp, int d) { return 2 * (size_t)tp * kg_tp_slot_floats(d); }
__host__ __device__ static inline size_t kg_tp_total_bytes(int tp, int d) { return (kg_tp_data_floats(tp, d) + 2 * (size_t)tp * KG_TP_FLAG_BLOCKS + 64) * 4; }
// out[m][:] = round16(resid[m][:] + round16(sum over ranks, in rank order, of slot[r][m][:])) after the one-shot exchange
int kg_launch_tp_reduce(const KgTpArgs& a, int max_rows, int dtype, hipStream_t st);
// end of a pass over the sharded layers (one thread): the pass counter the exchange flags are numbered from moves on
int kg_launch_tp_pass_end(const KgDyn* dyn, float* xbuf, int tp, int d, hipStream_t st);

void kg_set_error(const char* what, hipError_t e);
