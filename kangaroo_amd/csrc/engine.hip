// Host side of libkangaroo_hip.so: C ABI, pass composition, graph capture and the
// generate loops.  All device buffers are caller-owned (see include/kangaroo_hip.h).
#include "../../include/kangaroo_hip.h"
#include "kg_common.h"
#include "kg_kernels.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>

static thread_local std::string g_err;
void kg_set_error(const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
}
static int fail(int code, const char* msg) { g_err = msg; return code; }
#define KG_TRY(x) do { int rc_ = (x); if (rc_ != 0) { if (g_err.empty()) g_err = #x; return rc_; } } while (0)

extern "C" int kg_version(void) { return 1; }
extern "C" const char* kg_last_error(void) { return g_err.c_str(); }

static bool g_init = false;
static int g_ncu = 256;
static void kg_init_once() {
    if (g_init) return;
    kg_linear_init();
    kg_gemm_init();
    kg_attn_init();
    kg_mm_init();
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_ncu = p.multiProcessorCount;
    g_init = true;
}

static const char* kg_op_names[5] = {"qkv", "o", "gateup", "down", "head"};
enum { OP_QKV = 0, OP_O = 1, OP_GATEUP = 2, OP_DOWN = 3, OP_HEAD = 4 };

struct kg_session {
    kg_shape sh;
    kg_weights w;
    std::vector<kg_layer_weights> layers;
    kg_buffers buf;
    int elt;
    int q_dim, kv_dim;
    // workspace carve-up
    int* state; int* tokens;
    KgDyn *dyn_shallow, *dyn_adapter, *dyn_verify, *dyn_call;
    char *h, *h2, *xq, *attn_out, *act, *exited, *prev_exit, *adapter_in, *adapter_h, *hs_early;
    float *part_o, *part_ml, *head_part;
    unsigned* attn_counters;
    int* head_tok; float* head_pmax;
    char* vlogits; int* topk_rank;     // verify logits rows and rank of each drafted token (top-k verification)
    int accept_top_k;                  // 1: greedy acceptance
    float accept_top_p, temperature;   // accept_top_p > 0: top-p verification on logits / temperature
    int accept_typical; float typical_epsilon, typical_delta;   // typical verification at `temperature`
    int head_part_stride;
    int prefill_rows;
    // tuning
    // launch geometry per (mode: 0 GEMV / 1 MFMA, op: qkv, o, gateup, down, head)
    int waves[2][5], bpc[2][5];
    int nblk[2][5];                    // > 0: block count of the op, overriding blocks-per-CU
    int attn_nsplit, attn_blocks, attn_teams;
    int attn_prefetch_delay;           // the fetching blocks first sleep this many times 64 cycles, so the pass's own K/V requests are ahead of theirs
    int attn_prefetch_min;             // chunks that must hold keys before the fetch pays (a shorter attention pass ends before the fetch does)
    int attn_prefetch;                 // eighths of every o_proj row group that idle attention blocks pull into L2 (0: off)
    int v_transposed;
    // persistent pass kernel: phase programs [0] verify pass, [1] deep layers of plain decoding
    int persist;                       // 1: multi-row passes of the loop run as one persistent kernel
    int persist_waves[5];              // weight-streaming waves per linear phase (qkv, o, gateup, down, head)
    KgPhase* prog_dev[3]; int prog_n[3]; size_t prog_smem[3]; int prog_cap;     // [2]: ad-hoc passes of kg_forward_layers
    unsigned* bar;                     // [0..255] barrier counters, [256] abort flag
    char* rot; size_t rot_layer;       // per-layer activation buffers of the persistent passes
    char *pf_h, *pf_xn, *pf_xq, *pf_ao, *pf_act;   // activations of a prefill pass (KG_GEMM_ROWS rows)
    float* pf_part;                    // K-range partial sums of the d-row prefill projections [4][KG_GEMM_ROWS][d]
    // prompt pass that runs layer by layer over the whole prompt (t_max rows each): hidden rows, normalised rows, stored
    // q/k/v rows, q in head order, attention output, MLP activations
    char *pm_h, *pm_xn, *pm_qkv, *pm_q, *pm_ao, *pm_act;
    float* pm_part;                     // fp32 partial products of the projections whose K is cut into ranges: 2 * t_max rows of d
    int rows_ksplit;                    // tuning: 0 chosen by size, 1 never, n > 1 always n ranges (as far as the buffer allows)
    int rows_prefill_min;              // prompts of at least this many tokens take that pass (0: never)
    int gemm_prefill;                  // 1: prompts go through the 128-row GEMM passes
    int vanilla_gemv;                  // 1: plain decoding runs every layer and the head through the 1-row GEMV kernels
    int bench_layers;                  // kg_bench_op cycles through this many layers' weights (1: cache-warm)
    unsigned long long* tlog;          // phase timeline of the last persistent launch (KG_PERSIST_TLOG=1), else null
    std::vector<int> prog_kinds[3];
    // tensor-parallel remaining layers (kg_session_set_tp); size == 0: the session holds whole layers
    struct {
        int size, n_local, rank[KG_TP_MAX], hl, hkvl, fl;
        std::vector<kg_layer_weights> layers[KG_TP_MAX];
        void* kc[KG_TP_MAX]; void* vc[KG_TP_MAX];
        float* xbuf; float* peer[KG_TP_MAX];
        unsigned local_mask;
    } tp;
    // graphs
    hipStream_t own_stream;
    hipGraphExec_t round_graph; int round_graph_steps;
    hipGraphExec_t vanilla_graph;
    int* host_state; hipEvent_t ev[2]; hipEvent_t ev_in, ev_t0, ev_t1, ev_t2;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
    char* base; size_t off;
    template <class T> T* take(size_t n) { off = align_up(off, 256); T* p = base ? (T*)(base + off) : nullptr; off += n * sizeof(T); return p; }
};

static int prefill_rows_for(const kg_shape* sh) {
    // X rows of one chunk must fit LDS beside the partial-sum slots and the largest staged epilogue input
    // of a pass with an RMSNorm prologue (the RoPE rows of the qkv launch), exactly as kg_launch_linear adds them up
    for (int rows = 16; rows > 1; rows >>= 1) {
        const size_t side = ((size_t)rows * sh->head_dim * 4 + 255) / 256 * 256 + 512;
        if (side + kg_linear_smem(1, rows, sh->hidden_size, 8) <= (size_t)160 * 1024) return rows;
    }
    return 1;
}

static void carve(kg_session* s, Carver& c, const kg_shape* sh, int t_max) {
    const size_t d = sh->hidden_size, F = sh->intermediate_size, qd = (size_t)sh->num_heads * sh->head_dim;
    const size_t e = 2, R = KG_MAX_ROWS;
    const int max_rounds = t_max + 8;
    int* state = c.take<int>(KS_HDR + 2 * (size_t)max_rounds);
    int* tokens = c.take<int>((size_t)t_max + 64);
    KgDyn* dyn = c.take<KgDyn>(4);
    char* h = c.take<char>(R * d * e);
    char* h2 = c.take<char>(R * d * e);
    char* xq = c.take<char>(R * qd * e);
    char* attn_out = c.take<char>(R * qd * e);
    char* act = c.take<char>(R * F * e);
    char* exited = c.take<char>(R * d * e);
    char* prev_exit = c.take<char>(d * e);
    char* adapter_in = c.take<char>(2 * d * e);
    char* adapter_h = c.take<char>(R * d * e);
    char* hs_early = c.take<char>((size_t)t_max * d * e);
    const int nsplit_max = 32;
    float* part_o = c.take<float>((size_t)nsplit_max * KG_GEMM_ROWS * qd);
    float* part_ml = c.take<float>((size_t)nsplit_max * KG_GEMM_ROWS * sh->num_heads * 2);
    unsigned* attn_counters = c.take<unsigned>((KG_GEMM_ROWS / 16) * (size_t)sh->num_heads);
    const int hp_stride = (sh->vocab_size + 15) / 16;
    float* head_part = c.take<float>((size_t)R * hp_stride * 4);
    int* head_tok = c.take<int>(R);
    float* head_pmax = c.take<float>(R);
    char* vlogits = c.take<char>(R * (size_t)sh->vocab_size * e);
    int* topk_rank = c.take<int>(R + R * R);      // ranks, then the top-p membership table
    const int prog_cap = 5 * sh->num_layers + 2;
    KgPhase* prog0 = c.take<KgPhase>(prog_cap);
    KgPhase* prog1 = c.take<KgPhase>(prog_cap);
    KgPhase* prog2 = c.take<KgPhase>(prog_cap);
    unsigned* bar = c.take<unsigned>(512);
    // per-layer activation buffers of the persistent passes: h after attention, h after the MLP, q, attention output, MLP activations
    const size_t rot_layer = R * (2 * d + 2 * qd + F) * e;
    char* rot = c.take<char>(rot_layer * sh->num_layers);
    const size_t PR = KG_GEMM_ROWS;
    char* pf_h = c.take<char>(PR * d * e);
    char* pf_xn = c.take<char>(PR * d * e);
    char* pf_xq = c.take<char>(PR * qd * e);
    char* pf_ao = c.take<char>(PR * qd * e);
    char* pf_act = c.take<char>(PR * F * e);
    float* pf_part = c.take<float>(4 * PR * d);
    const size_t TR = (size_t)t_max, kvd2 = 2 * (size_t)sh->num_kv_heads * sh->head_dim;
    char* pm_h = c.take<char>(TR * d * e);
    char* pm_xn = c.take<char>(TR * d * e);
    char* pm_qkv = c.take<char>(TR * (qd + kvd2) * e);
    char* pm_q = c.take<char>(TR * qd * e);
    char* pm_ao = c.take<char>(TR * qd * e);
    char* pm_act = c.take<char>(TR * F * e);
    float* pm_part = c.take<float>(2 * TR * d);
    if (s) {
        s->pm_h = pm_h; s->pm_xn = pm_xn; s->pm_qkv = pm_qkv; s->pm_q = pm_q; s->pm_ao = pm_ao; s->pm_act = pm_act; s->pm_part = pm_part;
        s->prog_dev[0] = prog0; s->prog_dev[1] = prog1; s->prog_dev[2] = prog2; s->prog_cap = prog_cap; s->bar = bar; s->rot = rot; s->rot_layer = rot_layer;
        s->pf_h = pf_h; s->pf_xn = pf_xn; s->pf_xq = pf_xq; s->pf_ao = pf_ao; s->pf_act = pf_act; s->pf_part = pf_part;
        s->state = state; s->tokens = tokens;
        s->dyn_shallow = dyn; s->dyn_adapter = dyn + 1; s->dyn_verify = dyn + 2; s->dyn_call = dyn + 3;
        s->h = h; s->h2 = h2; s->xq = xq; s->attn_out = attn_out; s->act = act; s->exited = exited;
        s->prev_exit = prev_exit; s->adapter_in = adapter_in; s->adapter_h = adapter_h; s->hs_early = hs_early;
        s->part_o = part_o; s->part_ml = part_ml; s->attn_counters = attn_counters; s->head_part = head_part; s->head_part_stride = hp_stride;
        s->head_tok = head_tok; s->head_pmax = head_pmax; s->vlogits = vlogits; s->topk_rank = topk_rank;
    }
}

extern "C" size_t kg_workspace_bytes(const kg_shape* shape, int t_max) {
    Carver c{nullptr, 0};
    carve(nullptr, c, shape, t_max);
    return align_up(c.off, 256);
}

extern "C" int kg_session_create(const kg_shape* shape, const kg_weights* weights, const kg_buffers* buffers, kg_session** out) {
    g_err.clear();
    if (!shape || !weights || !buffers || !out) return fail(-1, "null argument");
    if (shape->hidden_size % 32 || shape->intermediate_size % 32) return fail(-1, "hidden_size and intermediate_size must be multiples of 32");
    if (shape->head_dim != 16 && shape->head_dim != 32 && shape->head_dim != 64 && shape->head_dim != 128) return fail(-1, "head_dim must be 16, 32, 64 or 128");
    if (shape->num_heads % shape->num_kv_heads) return fail(-1, "num_heads must be a multiple of num_kv_heads");
    // exit_layer == num_layers: a draft-only session (shallow layers, adapter, head) whose remaining layers run elsewhere
    // (tensor-parallel shards); kg_generate refuses it
    if (shape->exit_layer < 1 || shape->exit_layer > shape->num_layers) return fail(-1, "exit_layer must be in [1, num_layers]");
    if (buffers->t_max > shape->max_positions) return fail(-1, "t_max exceeds the RoPE table");
    if (buffers->workspace_bytes < kg_workspace_bytes(shape, buffers->t_max)) return fail(-1, "workspace too small");
    kg_init_once();
    kg_persist_init();
    kg_session* s = new kg_session();
    s->sh = *shape; s->w = *weights; s->buf = *buffers;
    s->layers.assign(weights->layers, weights->layers + shape->num_layers);
    s->w.layers = s->layers.data();
    s->elt = 2;
    s->tp.size = 0; s->tp.n_local = 0; s->tp.xbuf = nullptr; s->tp.local_mask = 0;
    s->q_dim = shape->num_heads * shape->head_dim;
    s->kv_dim = shape->num_kv_heads * shape->head_dim;
    Carver c{(char*)buffers->workspace, 0};
    carve(s, c, shape, buffers->t_max);
    s->prefill_rows = prefill_rows_for(shape);
    if (hipMemset(s->attn_counters, 0, (KG_GEMM_ROWS / 16) * (size_t)shape->num_heads * sizeof(unsigned)) != hipSuccess) { delete s; return fail(-2, "hipMemset failed"); }
    // one block per CU in both modes: an 8-wave block fills a CU, a second one only queues behind it
    for (int op = 0; op < 5; ++op) { s->waves[0][op] = 8; s->bpc[0][op] = 1; s->waves[1][op] = 8; s->bpc[1][op] = 1; s->nblk[0][op] = s->nblk[1][op] = 0; }
    s->waves[0][1] = 4; s->waves[0][3] = 4;      // o_proj, down_proj: one row group per block, 4 waves measured best
    s->waves[1][3] = 2;                          // down_proj in the large-K MFMA mode
    // o_proj, same kernel: the longer K, the fewer waves share a row group (measured at K = 4096 / 5120 / 6656:
    // 8.6 us with 8 waves, 13.3 with 4 against 15.3 with 8, 17.1 with 2 against 20.4 with 8)
    { const int Ko = shape->num_heads * shape->head_dim; s->waves[1][1] = Ko <= 4096 ? 8 : (Ko <= 5120 ? 4 : 2); }
    s->v_transposed = (shape->head_dim >= 32 && buffers->t_max % 32 == 0) ? 1 : 0;
    s->attn_nsplit = 16;
    s->accept_top_k = 1;
    s->accept_top_p = 0.f; s->temperature = 1.f;
    s->accept_typical = 0; s->typical_epsilon = 0.3f; s->typical_delta = 0.09f;
    s->gemm_prefill = s->v_transposed;           // the GEMM passes need the MFMA attention kernel
    // From 192 prompt tokens on the prompt goes through the layers one whole-prompt pass at a time (measured at the 7B
    // shape: 640 tokens 25.5 ms in 128-row passes); shorter prompts fill too few tiles of the large matrix products.
    s->rows_prefill_min = (s->v_transposed && shape->hidden_size % 8 == 0 && shape->intermediate_size % 8 == 0) ? 129 : 0;
    s->bench_layers = s->sh.num_layers;
    s->vanilla_gemv = 0;
    s->attn_teams = 1;
    s->attn_prefetch_min = 6; s->attn_prefetch_delay = 16;
    {   // about 24 MB of o_proj (3 MB per XCD, whose L2 holds 4 MB) in eighths of a row group (measured: 5/8 .. 8/8 alike at 7B, 4/8 .. 5/8 best at 13B, 2/8 at the 33B shape)
        const double ob = 2.0 * shape->hidden_size * s->q_dim;
        int e = (int)(8.0 * 24.0e6 / ob + 0.5);
        s->attn_prefetch = e < 1 ? 1 : (e > 8 ? 8 : e);
    }
    s->attn_blocks = 16;                         // one chunk per block: long contexts need many CUs pulling K/V (one CU sustains only ~60 GB/s)
    s->round_graph = nullptr; s->round_graph_steps = -1; s->vanilla_graph = nullptr;
    s->prog_n[0] = s->prog_n[1] = s->prog_n[2] = 0;
    for (int op = 0; op < 5; ++op) s->persist_waves[op] = 8;
    s->persist_waves[OP_DOWN] = 4;
    // Off by default: measured on MI355X at the 7B shape the software grid barrier (3-4.5 us per
    // phase as block 0 sees it, against ~1.6 us for a kernel boundary inside a graph) costs what
    // the uninterrupted weight stream wins; set "persist" = 2 to run the passes this way.
    s->persist = 0;
    s->tlog = nullptr;
    if (getenv("KG_PERSIST_TLOG") && hipMalloc((void**)&s->tlog, (2 * (size_t)s->prog_cap + 2) * sizeof(unsigned long long)) != hipSuccess) s->tlog = nullptr;
    if (hipMemset(s->bar, 0, 512 * sizeof(unsigned)) != hipSuccess) { delete s; return fail(-2, "hipMemset failed"); }
    if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess) { delete s; return fail(-2, "hipStreamCreate failed"); }
    if (hipHostMalloc((void**)&s->host_state, 4 * 64 * sizeof(int)) != hipSuccess) { delete s; return fail(-2, "hipHostMalloc failed"); }
    hipEventCreateWithFlags(&s->ev[0], hipEventDisableTiming);
    hipEventCreateWithFlags(&s->ev[1], hipEventDisableTiming);
    hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
    hipEventCreate(&s->ev_t0); hipEventCreate(&s->ev_t1); hipEventCreate(&s->ev_t2);
    *out = s;
    return 0;
}

extern "C" int kg_session_destroy(kg_session* s) {
    if (!s) return 0;
    hipStreamSynchronize(s->own_stream);
    if (s->round_graph) hipGraphExecDestroy(s->round_graph);
    if (s->vanilla_graph) hipGraphExecDestroy(s->vanilla_graph);
    hipEventDestroy(s->ev[0]); hipEventDestroy(s->ev[1]); hipEventDestroy(s->ev_in);
    hipEventDestroy(s->ev_t0); hipEventDestroy(s->ev_t1); hipEventDestroy(s->ev_t2);
    hipHostFree(s->host_state);
    hipStreamDestroy(s->own_stream);
    delete s;
    return 0;
}

extern "C" int kg_session_set(kg_session* s, const char* key, int value) {
    if (!s || !key) return fail(-1, "null argument");
    std::string k(key);
    if (value < 1) return fail(-1, "value must be >= 1");
    bool done = false;
    if (k == "attn_nsplit") { if (value > 32) return fail(-1, "nsplit <= 32"); s->attn_nsplit = value; done = true; }
    if (k == "gemm_prefill") { s->gemm_prefill = (value >= 2 && s->v_transposed) ? 1 : 0; done = true; }   // 1: off, 2: on
    if (k == "rows_prefill_min") { s->rows_prefill_min = (value >= 2 && s->v_transposed) ? value : 0; done = true; }   // 1: off
    if (k == "rows_ksplit") { if (value > 17) return fail(-1, "rows_ksplit <= 17"); s->rows_ksplit = value - 1; done = true; }   // 1: by size, 2: never, n + 1: n ranges
    if (k == "vanilla_gemv") { s->vanilla_gemv = value >= 2 ? 1 : 0; done = true; }   // 1: off, 2: on
    if (k == "bench_layers") { s->bench_layers = value < s->sh.num_layers ? value : s->sh.num_layers; done = true; }
    if (k == "accept_top_k") { if (value > s->sh.vocab_size) return fail(-1, "accept_top_k <= vocab_size"); s->accept_top_k = value; done = true; }
    if (k == "accept_typical") { s->accept_typical = value >= 2 ? 1 : 0; done = true; }   // 1: off, 2: on
    if (k == "attn_teams") { if (value > 2) return fail(-1, "attn_teams is 1 or 2"); s->attn_teams = value; done = true; }
    if (k == "attn_prefetch") { if (value > 9) return fail(-1, "attn_prefetch is 1 (off) .. 9 (whole row groups)"); s->attn_prefetch = value - 1; done = true; }
    if (k == "attn_prefetch_delay") { s->attn_prefetch_delay = value - 1; done = true; }
    if (k == "attn_prefetch_min") { s->attn_prefetch_min = value; done = true; }
    if (k == "attn_blocks") { if (value > 32) return fail(-1, "attn_blocks <= 32"); s->attn_blocks = value; done = true; }
    if (k == "persist") { s->persist = (value >= 2 && s->v_transposed && (g_ncu % 8) == 0 && s->tp.size == 0) ? 1 : 0; done = true; }   // 1: off, 2: on
    for (int op = 0; op < 5 && !done; ++op)
        if (k == std::string("persist.") + kg_op_names[op] + ".waves") { if (value > 8) return fail(-1, "waves <= 8"); s->persist_waves[op] = value; done = true; }
    for (int mode = 0; mode < 2 && !done; ++mode) {
        const std::string mname = mode == 0 ? "vec" : "mfma";
        if (k == mname + "_waves") { if (value > 8) return fail(-1, "waves <= 8"); for (int op = 0; op < 5; ++op) s->waves[mode][op] = value; done = true; }
        if (k == mname + "_blocks_per_cu") { for (int op = 0; op < 5; ++op) s->bpc[mode][op] = value; done = true; }
        for (int op = 0; op < 5 && !done; ++op) {
            if (k == mname + "." + kg_op_names[op] + ".waves") { if (value > 8) return fail(-1, "waves <= 8"); s->waves[mode][op] = value; done = true; }
            if (k == mname + "." + kg_op_names[op] + ".blocks_per_cu") { s->bpc[mode][op] = value; s->nblk[mode][op] = 0; done = true; }
            if (k == mname + "." + kg_op_names[op] + ".blocks") { s->nblk[mode][op] = value; done = true; }
        }
    }
    if (!done) return fail(-1, "unknown key");
    // graphs bake the launch geometry in
    if (s->round_graph) { hipGraphExecDestroy(s->round_graph); s->round_graph = nullptr; s->round_graph_steps = -1; }
    if (s->vanilla_graph) { hipGraphExecDestroy(s->vanilla_graph); s->vanilla_graph = nullptr; }
    s->prog_n[0] = s->prog_n[1] = s->prog_n[2] = 0;
    return 0;
}

extern "C" int kg_session_setf(kg_session* s, const char* key, float value) {
    if (!s || !key) return fail(-1, "null argument");
    std::string k(key);
    if (k == "accept_top_p") { if (!(value >= 0.f && value <= 1.f)) return fail(-1, "accept_top_p in [0, 1]"); s->accept_top_p = value; }
    else if (k == "temperature") { if (!(value > 0.f)) return fail(-1, "temperature > 0"); s->temperature = value; }
    else if (k == "typical_epsilon") s->typical_epsilon = value;
    else if (k == "typical_delta") s->typical_delta = value;
    else return fail(-1, "unknown key");
    if (s->round_graph) { hipGraphExecDestroy(s->round_graph); s->round_graph = nullptr; s->round_graph_steps = -1; }
    s->prog_n[0] = s->prog_n[1] = s->prog_n[2] = 0;
    return 0;
}

extern "C" float kg_session_getf(kg_session* s, const char* key) {
    if (!s || !key) return -1.f;
    std::string k(key);
    if (k == "accept_top_p") return s->accept_top_p;
    if (k == "temperature") return s->temperature;
    if (k == "typical_epsilon") return s->typical_epsilon;
    if (k == "typical_delta") return s->typical_delta;
    return -1.f;
}

extern "C" int kg_session_get(kg_session* s, const char* key) {
    if (!s || !key) return -1;
    std::string k(key);
    if (k == "v_transposed") return s->v_transposed;
    for (int mode = 0; mode < 2; ++mode) {
        const std::string mname = mode == 0 ? "vec" : "mfma";
        for (int op = 0; op < 5; ++op) {
            if (k == mname + "." + kg_op_names[op] + ".waves") return s->waves[mode][op];
            if (k == mname + "." + kg_op_names[op] + ".blocks_per_cu") return s->bpc[mode][op];
        }
    }
    if (k == "attn_nsplit") return s->attn_nsplit;
    if (k == "attn_blocks") return s->attn_blocks;
    if (k == "attn_teams") return s->attn_teams;
    if (k == "attn_prefetch") return s->attn_prefetch + 1;
    if (k == "attn_prefetch_min") return s->attn_prefetch_min;
    if (k == "attn_prefetch_delay") return s->attn_prefetch_delay + 1;
    if (k == "accept_top_k") return s->accept_top_k;
    if (k == "accept_typical") return s->accept_typical;
    if (k == "gemm_prefill") return s->gemm_prefill;
    if (k == "rows_prefill_min") return s->rows_prefill_min;
    if (k == "rows_ksplit") return s->rows_ksplit + 1;
    if (k == "vanilla_gemv") return s->vanilla_gemv;
    if (k == "persist") return s->persist;
    if (k == "prefill_rows") return s->prefill_rows;
    return -1;
}

extern "C" void* kg_session_ptr(kg_session* s, const char* name) {
    if (!s || !name) return nullptr;
    std::string k(name);
    if (k == "tokens") return s->tokens;
    if (k == "state") return s->state;
    if (k == "hs_early") return s->hs_early;
    if (k == "head_tok") return s->head_tok;
    if (k == "head_pmax") return s->head_pmax;
    return nullptr;
}

// ---------------------------------------------------------------------------------
// pass composition
// ---------------------------------------------------------------------------------
static int grid_for(const kg_session* s, int N, int max_rows, int op) {
    const int G = (N + 15) / 16;
    const int mode = max_rows == 1 ? 0 : 1;
    if (s->nblk[mode][op] > 0) return s->nblk[mode][op] < G ? s->nblk[mode][op] : G;
    // Balanced: with at most cap blocks a block walks k = ceil(G / cap) row groups; ceil(G / k) blocks do
    // that with every block within one group of k (1376 groups on 256 CUs: 230 blocks x 6 instead of
    // 96 blocks x 6 and 160 x 5; 320 groups: 160 x 2 instead of 64 x 2 and 192 x 1).
    // The projections that open with an RMSNorm of the X rows (qkv, gate/up, head) do best on about 200 of the 256
    // CUs (measured at the 7B and 13B shapes, 1 and 7 rows: qkv 192 blocks x 4 groups against 256 x 3 -0.3 us,
    // gate/up 197 x 7 against 230 x 6 -0.4 us, head 200 x 10 against 250 x 8 -0.5 us; 7B round -1.3 %): the weight
    // stream is saturated well below 256 CUs and every block fewer is one prologue fewer in the opening burst.
    // The residual projections (one row group per block at these shapes) want every CU.
    const bool norm_op = (op == OP_QKV || op == OP_GATEUP || op == OP_HEAD);
    const int cap = (norm_op ? g_ncu * 25 / 32 : g_ncu) * s->bpc[mode][op];
    if (cap >= G) return G;
    const int k = (G + cap - 1) / cap;
    return (G + k - 1) / k;
}
static int waves_for(const kg_session* s, int max_rows, int op) { return s->waves[max_rows == 1 ? 0 : 1][op]; }

static KgLinArgs lin_base(const KgDyn* dyn, const void* W, int N, int K, const void* x, int x_stride) {
    KgLinArgs a;
    memset(&a, 0, sizeof(a));
    a.W = W; a.N = N; a.K = K; a.dyn = dyn; a.x = x; a.x_stride = x_stride;
    return a;
}

// One decoder layer on the rows described by dyn.  lw.wgu == NULL: attention half only
// (the adapter layer, kangaroo/adapter.py:292-304).  h_in / h_out may alias.
static int run_layer(kg_session* s, const kg_layer_weights& lw, void* kc, void* vc, const KgDyn* dyn, int max_rows,
                     const void* h_in, void* h_out, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, F = sh.intermediate_size, dt = sh.dtype;
    {
        KgLinArgs a = lin_base(dyn, lw.wqkv, s->q_dim + 2 * s->kv_dim, d, h_in, d);
        a.norm_w = lw.norm1; a.eps = sh.rms_eps;
        a.q_out = s->xq; a.k_cache = kc; a.v_cache = vc;
        a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin;
        a.head_dim = sh.head_dim; a.q_dim = s->q_dim; a.kv_dim = s->kv_dim; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
        KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_RMSNORM, EPI_ROPE_KV, grid_for(s, a.N, max_rows, OP_QKV), waves_for(s, max_rows, OP_QKV), st));
    }
    {
        KgAttnArgs a;
        memset(&a, 0, sizeof(a));
        a.dyn = dyn; a.q = s->xq; a.k_cache = kc; a.v_cache = vc;
        a.H = sh.num_heads; a.Hkv = sh.num_kv_heads; a.hd = sh.head_dim; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
        a.nsplit = s->attn_nsplit; a.blocks_per_head = s->attn_blocks; a.teams = s->attn_teams; a.part_o = s->part_o; a.part_ml = s->part_ml; a.counters = s->attn_counters; a.out = s->attn_out; a.max_rows = max_rows;
        if (s->attn_prefetch > 0 && max_rows <= KG_MAX_ROWS) {
            a.pf_w = lw.wo; a.pf_groups = (d + 15) / 16; a.pf_group_bytes = (s->q_dim / 32) * 1024;
            a.pf_grid = grid_for(s, d, max_rows, OP_O); a.pf_bytes_per_group = a.pf_group_bytes / 8 * s->attn_prefetch; a.pf_min_chunks = s->attn_prefetch_min; a.pf_delay = s->attn_prefetch_delay;
        }
        KG_TRY(kg_launch_attn(a, dt, st));
    }
    const bool attn_only = (lw.wgu == nullptr);
    void* h_mid = h_out;
    {
        KgLinArgs a = lin_base(dyn, lw.wo, d, s->q_dim, s->attn_out, s->q_dim);
        a.out = h_mid; a.out_stride = d; a.resid = h_in; a.resid_stride = d;
        KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_NONE, EPI_RESID, grid_for(s, a.N, max_rows, OP_O), waves_for(s, max_rows, OP_O), st));
    }
    if (attn_only) return 0;
    {
        KgLinArgs a = lin_base(dyn, lw.wgu, 2 * F, d, h_mid, d);
        a.norm_w = lw.norm2; a.eps = sh.rms_eps;
        a.out = s->act; a.out_stride = F;
        KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_RMSNORM, EPI_SILU_MUL, grid_for(s, a.N, max_rows, OP_GATEUP), waves_for(s, max_rows, OP_GATEUP), st));
    }
    {
        KgLinArgs a = lin_base(dyn, lw.wdown, d, F, s->act, F);
        a.out = h_out; a.out_stride = d; a.resid = h_mid; a.resid_stride = d;
        KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_NONE, EPI_RESID, grid_for(s, a.N, max_rows, OP_DOWN), waves_for(s, max_rows, OP_DOWN), st));
    }
    return 0;
}

static bool tp_remote(const kg_session* s) { return s->tp.size > 1 && s->tp.local_mask != ((1u << s->tp.size) - 1u); }

static int tp_reduce(kg_session* s, const KgDyn* dyn, int max_rows, int e, const void* resid, void* out, hipStream_t st) {
    KgTpArgs a;
    memset(&a, 0, sizeof(a));
    a.dyn = dyn; a.xbuf = s->tp.xbuf;
    for (int r = 0; r < KG_TP_MAX; ++r) a.peer[r] = s->tp.peer[r];
    a.tp = s->tp.size; a.d = s->sh.hidden_size; a.local_mask = s->tp.local_mask;
    a.e = e; a.exch_per_pass = 2 * (s->sh.num_layers - s->sh.exit_layer);
    a.resid = resid; a.resid_stride = a.d; a.out = out; a.out_stride = a.d;
    return kg_launch_tp_reduce(a, max_rows, s->sh.dtype, st);
}
static float* tp_slot(kg_session* s, int e, int rank) {
    return s->tp.xbuf + ((size_t)(e & 1) * s->tp.size + rank) * kg_tp_slot_floats(s->sh.hidden_size);
}
// waves of a row-split projection's MFMA launch on a shard: K is a fraction of the unsplit K, so the X rows fit LDS
// and eight waves share a row group (the unsplit large-K mode streams X and does best with fewer)
static int tp_waves(const kg_session* s, int max_rows, int op, int K) {
    if (max_rows == 1) return s->waves[0][op];
    return ((size_t)max_rows * K * 2 <= 64 * 1024) ? 8 : s->waves[1][op];
}

// Layer l >= exit_layer of a session whose remaining layers are shards (kg_session_set_tp): every local shard runs
// qkv -> attention -> its columns of o_proj (fp32 partial sums into its slot of the exchange buffer), one exchange
// kernel adds the ranks' slots and the residual; the same for gate/up -> its columns of down_proj.  5 kernels per
// shard and 2 per layer.  h_in / h_out may alias.
static int run_layer_tp(kg_session* s, int l, const KgDyn* dyn, int max_rows, const void* h_in, void* h_out, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, dt = sh.dtype, hd = sh.head_dim, li = l - sh.exit_layer, e0 = 2 * li;
    const int qd = s->tp.hl * hd, kvd = s->tp.hkvl * hd, fl = s->tp.fl;
    const size_t cache_layer = (size_t)s->tp.hkvl * s->buf.t_max * hd * s->elt;
    for (int j = 0; j < s->tp.n_local; ++j) {
        const kg_layer_weights& lw = s->tp.layers[j][li];
        void* kc = (char*)s->tp.kc[j] + (size_t)li * cache_layer;
        void* vc = (char*)s->tp.vc[j] + (size_t)li * cache_layer;
        {
            KgLinArgs a = lin_base(dyn, lw.wqkv, qd + 2 * kvd, d, h_in, d);
            a.norm_w = lw.norm1; a.eps = sh.rms_eps;
            a.q_out = s->xq; a.k_cache = kc; a.v_cache = vc;
            a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin;
            a.head_dim = hd; a.q_dim = qd; a.kv_dim = kvd; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
            KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_RMSNORM, EPI_ROPE_KV, grid_for(s, a.N, max_rows, OP_QKV), waves_for(s, max_rows, OP_QKV), st));
        }
        {
            KgAttnArgs a;
            memset(&a, 0, sizeof(a));
            a.dyn = dyn; a.q = s->xq; a.k_cache = kc; a.v_cache = vc;
            a.H = s->tp.hl; a.Hkv = s->tp.hkvl; a.hd = hd; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
            a.nsplit = s->attn_nsplit; a.blocks_per_head = s->attn_blocks; a.teams = s->attn_teams; a.part_o = s->part_o; a.part_ml = s->part_ml; a.counters = s->attn_counters; a.out = s->attn_out; a.max_rows = max_rows;
            KG_TRY(kg_launch_attn(a, dt, st));
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wo, d, qd, s->attn_out, qd);
            a.out = tp_slot(s, e0, s->tp.rank[j]); a.out_stride = d;
            KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_NONE, EPI_STORE_F32, grid_for(s, a.N, max_rows, OP_O), tp_waves(s, max_rows, OP_O, qd), st));
        }
    }
    KG_TRY(tp_reduce(s, dyn, max_rows, e0, h_in, h_out, st));
    for (int j = 0; j < s->tp.n_local; ++j) {
        const kg_layer_weights& lw = s->tp.layers[j][li];
        {
            KgLinArgs a = lin_base(dyn, lw.wgu, 2 * fl, d, h_out, d);
            a.norm_w = lw.norm2; a.eps = sh.rms_eps;
            a.out = s->act; a.out_stride = fl;
            KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_RMSNORM, EPI_SILU_MUL, grid_for(s, a.N, max_rows, OP_GATEUP), waves_for(s, max_rows, OP_GATEUP), st));
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wdown, d, fl, s->act, fl);
            a.out = tp_slot(s, e0 + 1, s->tp.rank[j]); a.out_stride = d;
            KG_TRY(kg_launch_linear(a, dt, max_rows, PRO_NONE, EPI_STORE_F32, grid_for(s, a.N, max_rows, OP_DOWN), tp_waves(s, max_rows, OP_DOWN, fl), st));
        }
    }
    return tp_reduce(s, dyn, max_rows, e0 + 1, h_out, h_out, st);
}
// after the last sharded layer of a pass: the exchange flags of the next pass carry higher numbers
static int tp_pass_end(kg_session* s, const KgDyn* dyn, hipStream_t st) {
    if (!tp_remote(s)) return 0;
    return kg_launch_tp_pass_end(dyn, s->tp.xbuf, s->tp.size, s->sh.hidden_size, st);
}
static bool is_tp_layer(const kg_session* s, int l) { return s->tp.size > 1 && l >= s->sh.exit_layer; }

static void* layer_k(kg_session* s, int l) { return (char*)s->buf.k_cache + (size_t)l * s->sh.num_kv_heads * s->buf.t_max * s->sh.head_dim * s->elt; }
static void* layer_v(kg_session* s, int l) { return (char*)s->buf.v_cache + (size_t)l * s->sh.num_kv_heads * s->buf.t_max * s->sh.head_dim * s->elt; }
// base layer l on the rows of dyn, whole or sharded
static int run_layer_at(kg_session* s, int l, const KgDyn* dyn, int max_rows, const void* h_in, void* h_out, hipStream_t st) {
    if (is_tp_layer(s, l)) return run_layer_tp(s, l, dyn, max_rows, h_in, h_out, st);
    return run_layer(s, s->layers[l], layer_k(s, l), layer_v(s, l), dyn, max_rows, h_in, h_out, st);
}

// lm_head on rows of x; prologue RMSNorm with norm_w when given.  Leaves per-block
// (max, sum, argmax) records in head_part; returns the block count through *nblocks.
static int run_head(kg_session* s, const KgDyn* dyn, int max_rows, const void* x, int x_last_row, const void* norm_w,
                    void* logits_out, int* nblocks, hipStream_t st) {
    KgLinArgs a = lin_base(dyn, s->w.lm_head, s->sh.vocab_size, s->sh.hidden_size, x, s->sh.hidden_size);
    a.x_last_row = x_last_row;
    a.norm_w = norm_w; a.eps = s->sh.rms_eps;
    a.head_part = s->head_part; a.head_part_stride = s->head_part_stride;
    a.logits_out = logits_out; a.logits_stride = s->sh.vocab_size;
    int nb = grid_for(s, a.N, max_rows, OP_HEAD);
    *nblocks = nb;
    return kg_launch_linear(a, s->sh.dtype, max_rows, norm_w ? PRO_RMSNORM : PRO_NONE, EPI_HEAD, nb, waves_for(s, max_rows, OP_HEAD), st);
}

static KgLoopBufs loop_bufs(kg_session* s) {
    KgLoopBufs b;
    b.state = s->state; b.tokens = s->tokens;
    b.dyn_shallow = s->dyn_shallow; b.dyn_adapter = s->dyn_adapter; b.dyn_verify = s->dyn_verify;
    b.exited = s->exited; b.prev_exit = s->prev_exit; b.adapter_in = s->adapter_in;
    b.head_tok = s->head_tok; b.head_pmax = s->head_pmax;
    b.d = s->sh.hidden_size; b.elt = s->elt;
    b.embed = s->w.embed; b.h = s->h;
    return b;
}

// records the head phase of a persistent pass leaves: one per resident block that owns a row group
static int persist_head_records(const kg_session* s) { const int G = (s->sh.vocab_size + 15) / 16; return g_ncu < G ? g_ncu : G; }

static int build_program(kg_session* s, int which, const KgDyn* dyn, int max_rows, int l0, int l1, const void* x_in, void* hbuf, bool with_head);
static int run_program(kg_session* s, int which, hipStream_t st);

// host side of the exchange's time-out: a wait that gave up left the abort word set
static int tp_check_abort(kg_session* s, hipStream_t st) {
    if (!tp_remote(s)) return 0;
    unsigned aborted = 0;
    const size_t off = (kg_tp_data_floats(s->tp.size, s->sh.hidden_size) + 2 * (size_t)s->tp.size * KG_TP_FLAG_BLOCKS + 1) * 4;
    KG_CHECK_HIP(hipMemcpyAsync(&aborted, (char*)s->tp.xbuf + off, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    KG_CHECK_HIP(hipStreamSynchronize(st));
    if (aborted) return fail(-5, "a tensor-parallel peer did not deliver its partial sums in time; the session's exchange is switched off (results of this call are invalid)");
    return 0;
}

extern "C" size_t kg_tp_xbuf_bytes(int tp_size, int hidden_size) {
    if (tp_size < 2 || tp_size > KG_TP_MAX || hidden_size < 32) return 0;
    return align_up(kg_tp_total_bytes(tp_size, hidden_size), 4096);
}
extern "C" int kg_tp_xbuf_create(size_t bytes, void** out, void* ipc_handle_64) {
    g_err.clear();
    if (!out || bytes == 0) return fail(-1, "bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle is passed around as 64 bytes");
    void* p = nullptr;
    // uncached: every access goes to memory, so what a peer GPU stores here is seen without cache maintenance
    KG_CHECK_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && ipc_handle_64) e = hipIpcGetMemHandle((hipIpcMemHandle_t*)ipc_handle_64, p);
    if (e != hipSuccess) { (void)hipFree(p); kg_set_error("exchange buffer set-up", e); return -2; }
    *out = p;
    return 0;
}
extern "C" int kg_tp_xbuf_open(const void* ipc_handle_64, void** out) {
    g_err.clear();
    if (!ipc_handle_64 || !out) return fail(-1, "bad argument");
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle_64, sizeof(h));
    KG_CHECK_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
extern "C" int kg_tp_xbuf_close(void* mapped) { g_err.clear(); if (mapped) KG_CHECK_HIP(hipIpcCloseMemHandle(mapped)); return 0; }
extern "C" int kg_tp_xbuf_destroy(void* p) { g_err.clear(); if (p) KG_CHECK_HIP(hipFree(p)); return 0; }

extern "C" int kg_session_set_tp(kg_session* s, const kg_tp_config* c) {
    g_err.clear();
    if (!s || !c) return fail(-1, "null argument");
    const kg_shape& sh = s->sh;
    const int n_deep = sh.num_layers - sh.exit_layer;
    if (c->tp_size < 2 || c->tp_size > KG_TP_MAX) return fail(-1, "tp_size must be in [2, 8]");
    if (n_deep < 1) return fail(-1, "the session has no remaining layers to shard");
    if (c->n_local != 1 && c->n_local != c->tp_size) return fail(-1, "n_local must be 1 (one rank per GPU) or tp_size (all ranks rehearsed on one GPU)");
    if (c->heads_local < 1 || c->kv_heads_local < 1 || c->heads_local % c->kv_heads_local) return fail(-1, "bad local head counts");
    if (c->heads_local * c->tp_size < sh.num_heads || c->kv_heads_local * c->tp_size < sh.num_kv_heads) return fail(-1, "the shards do not cover every head");
    if (c->ffn_local % 32 || c->ffn_local * c->tp_size != sh.intermediate_size) return fail(-1, "ffn_local must be intermediate_size / tp_size and a multiple of 32");
    if (!c->xbuf) return fail(-1, "xbuf required");
    unsigned mask = 0;
    for (int j = 0; j < c->n_local; ++j) {
        const int r = c->local_rank[j];
        if (r < 0 || r >= c->tp_size || ((mask >> r) & 1u)) return fail(-1, "bad local_rank");
        mask |= 1u << r;
        if (!c->layers[j] || !c->k_cache[j] || !c->v_cache[j]) return fail(-1, "shard weights and caches required");
        for (int i = 0; i < n_deep; ++i) {
            const kg_layer_weights& lw = c->layers[j][i];
            if (!lw.wqkv || !lw.wo || !lw.wgu || !lw.wdown || !lw.norm1 || !lw.norm2) return fail(-1, "a shard layer misses a weight");
        }
    }
    for (int r = 0; r < c->tp_size; ++r)
        if (!((mask >> r) & 1u) && !c->peer_xbuf[r]) return fail(-1, "peer_xbuf required for every rank that is not local");
    s->tp.size = c->tp_size; s->tp.n_local = c->n_local; s->tp.hl = c->heads_local; s->tp.hkvl = c->kv_heads_local; s->tp.fl = c->ffn_local;
    s->tp.local_mask = mask; s->tp.xbuf = (float*)c->xbuf;
    for (int r = 0; r < KG_TP_MAX; ++r) s->tp.peer[r] = (r < c->tp_size && !((mask >> r) & 1u)) ? (float*)c->peer_xbuf[r] : nullptr;
    for (int j = 0; j < c->n_local; ++j) {
        s->tp.rank[j] = c->local_rank[j];
        s->tp.layers[j].assign(c->layers[j], c->layers[j] + n_deep);
        s->tp.kc[j] = c->k_cache[j]; s->tp.vc[j] = c->v_cache[j];
    }
    s->persist = 0;
    if (s->round_graph) { hipGraphExecDestroy(s->round_graph); s->round_graph = nullptr; s->round_graph_steps = -1; }
    if (s->vanilla_graph) { hipGraphExecDestroy(s->vanilla_graph); s->vanilla_graph = nullptr; }
    s->prog_n[0] = s->prog_n[1] = s->prog_n[2] = 0;
    return 0;
}

// ---------------------------------------------------------------------------------
// model-level entry points
// ---------------------------------------------------------------------------------
extern "C" int kg_set_tokens(kg_session* s, const int32_t* tokens_host, int pos0, int n, void* stream) {
    g_err.clear();
    if (!s || !tokens_host || pos0 < 0 || n < 0 || pos0 + n > s->buf.t_max) return fail(-1, "bad token range");
    for (int i = 0; i < n; ++i)
        if (tokens_host[i] < 0 || tokens_host[i] >= s->sh.vocab_size) return fail(-1, "token id out of range");
    KG_CHECK_HIP(hipMemcpyAsync(s->tokens + pos0, tokens_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int kg_forward_layers(kg_session* s, int layer_begin, int layer_end, const void* x_in, int pos0, int n_rows,
                                 void* h_out, void* stream) {
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    if (!s || !h_out) return fail(-1, "null argument");
    if (layer_begin < 0 || layer_end > s->sh.num_layers || layer_begin >= layer_end) return fail(-1, "bad layer range");
    if (n_rows < 1 || n_rows > KG_MAX_ROWS) return fail(-1, "n_rows must be in [1, 16]");
    if (pos0 < 0 || pos0 + n_rows > s->buf.t_max) return fail(-1, "positions exceed the cache");
    const int d = s->sh.hidden_size;
    const int max_rows = (n_rows == 1) ? 1 : n_rows;
    if (max_rows > 1 && kg_linear_smem(1, max_rows, d, 8) > 160 * 1024) return fail(-3, "rows x hidden_size exceed LDS; use fewer rows per call");
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n_rows, pos0, st));
    const void* cur = x_in;
    if (!x_in) {
        KG_TRY(kg_launch_embed(s->dyn_call, s->tokens, s->w.embed, d, s->h, d, st));
        cur = s->h;
    }
    if (s->persist && n_rows >= 2 && s->tp.size == 0) {
        // the whole pass as one persistent kernel (rows live in the session's h2 buffer meanwhile)
        KG_CHECK_HIP(hipStreamSynchronize(st));
        KG_TRY(build_program(s, 2, s->dyn_call, max_rows, layer_begin, layer_end, cur, s->h2, false));
        if (s->prog_n[2] > 0) {
            KG_CHECK_HIP(hipMemsetAsync(s->bar + 256, 0, sizeof(unsigned), st));
            KG_TRY(run_program(s, 2, st));
            KG_CHECK_HIP(hipMemcpyAsync(h_out, s->h2, (size_t)n_rows * d * s->elt, hipMemcpyDeviceToDevice, st));
            unsigned aborted = 0;
            KG_CHECK_HIP(hipMemcpyAsync(&aborted, s->bar + 256, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            KG_CHECK_HIP(hipStreamSynchronize(st));
            if (aborted) { s->persist = 0; return fail(-4, "persistent pass timed out at a grid barrier; persistent passes are now off for this session, call again"); }
            return 0;
        }
    }
    for (int l = layer_begin; l < layer_end; ++l) {
        void* dst = (l == layer_end - 1) ? h_out : (void*)s->h;
        KG_TRY(run_layer_at(s, l, s->dyn_call, max_rows, cur, dst, st));
        cur = dst;
    }
    if (is_tp_layer(s, layer_end - 1)) {
        KG_TRY(tp_pass_end(s, s->dyn_call, st));
        KG_TRY(tp_check_abort(s, st));
    }
    return 0;
}

extern "C" int kg_final_norm(kg_session* s, const void* x, int n_rows, void* out, void* stream) {
    g_err.clear();
    if (!s || !x || !out || n_rows < 1 || n_rows > KG_MAX_ROWS) return fail(-1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n_rows, 0, st));
    return kg_launch_rmsnorm(s->dyn_call, x, s->sh.hidden_size, s->w.final_norm, s->sh.rms_eps, s->sh.hidden_size, out, s->sh.hidden_size, s->sh.dtype, st);
}

extern "C" int kg_adapter_forward(kg_session* s, const void* x_in, int pos0, int n_rows, void* h_out, void* stream) {
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    if (!s || !x_in || !h_out) return fail(-1, "null argument");
    if (n_rows < 1 || n_rows > KG_MAX_ROWS) return fail(-1, "n_rows must be in [1, 16]");
    if (pos0 < 0 || pos0 + n_rows > s->buf.t_max) return fail(-1, "positions exceed the cache");
    const int d = s->sh.hidden_size;
    const int max_rows = (n_rows == 1) ? 1 : n_rows;
    if (max_rows > 1 && kg_linear_smem(1, max_rows, d, 8) > 160 * 1024) return fail(-3, "rows x hidden_size exceed LDS; use fewer rows per call");
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n_rows, pos0, st));
    KG_TRY(run_layer(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, s->dyn_call, max_rows, x_in, s->adapter_h, st));
    return kg_launch_rmsnorm(s->dyn_call, s->adapter_h, d, s->w.adapter_norm, s->sh.rms_eps, d, h_out, d, s->sh.dtype, st);
}

extern "C" int kg_head_logits(kg_session* s, const void* x, int n_rows, void* logits_out, int* argmax_out, float* pmax_out, void* stream) {
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    const int min_rows = (n_rows == -1) ? 2 : 1;          // -1: one row in a multi-row pass's arithmetic (see kg_linear)
    if (n_rows == -1) n_rows = 1;
    if (!s || !x || n_rows < 1 || n_rows > KG_MAX_ROWS) return fail(-1, "bad argument");
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n_rows, 0, st));
    int nb = 0;
    KG_TRY(run_head(s, s->dyn_call, n_rows < min_rows ? min_rows : n_rows, x, 0, nullptr, logits_out, &nb, st));
    if (argmax_out || pmax_out) {
        KG_TRY(kg_launch_head_finalize(s->dyn_call, s->head_part, s->head_part_stride, nb, n_rows,
                                       argmax_out ? argmax_out : s->head_tok, pmax_out ? pmax_out : s->head_pmax, st));
    }
    return 0;
}

// ---------------------------------------------------------------------------------
// persistent passes
// ---------------------------------------------------------------------------------
// Phase list of layers [l0, l1) + lm_head on the rows of dyn (<= max_rows): the same ops with
// the same arguments as run_layer / run_head issue kernel by kernel.  Returns 0 and leaves
// prog_n[which] = 0 when the pass cannot run persistently (the caller then launches kernels).
static int build_program(kg_session* s, int which, const KgDyn* dyn, int max_rows, int l0, int l1, const void* x_in, void* hbuf, bool with_head) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, F = sh.intermediate_size, nblk = g_ncu;
    s->prog_n[which] = 0;
    if (!s->persist || max_rows < 2 || 5 * (l1 - l0) + 1 > s->prog_cap) return 0;
    std::vector<KgPhase> prog;
    size_t smem = kg_persist_attn_smem(sh.head_dim);
    if (smem == 0) return 0;
    auto add_lin = [&](int kind, int op, const KgLinArgs& a) -> bool {
        KgPhase p;
        memset(&p, 0, sizeof(p));
        p.kind = kind; p.nwaves = s->persist_waves[op]; p.lin = a;
        const size_t need = kg_persist_plan_linear(p, max_rows, nblk);
        if (need == 0) return false;
        if (need > smem) smem = need;
        prog.push_back(p);
        return true;
    };
    // Every layer works in buffers of its own: within one launch a buffer is written once (by
    // one phase) and read by later phases only, which is what lets those reads be ordinary
    // cached loads (see KG_PERSIST_COH_LOADS).  The last layer writes the caller's hbuf.
    const size_t R = KG_MAX_ROWS, e = s->elt;
    const void* cur = x_in;
    for (int l = l0; l < l1; ++l) {
        const kg_layer_weights& lw = s->layers[l];
        char* base = s->rot + (size_t)l * s->rot_layer;
        void* hmid = base;
        void* hout = (l == l1 - 1) ? hbuf : (void*)(base + R * d * e);
        void* xq = base + 2 * R * d * e;
        void* ao = base + (2 * R * d + R * (size_t)s->q_dim) * e;
        void* act = base + (2 * R * d + 2 * R * (size_t)s->q_dim) * e;
        {
            KgLinArgs a = lin_base(dyn, lw.wqkv, s->q_dim + 2 * s->kv_dim, d, cur, d);
            a.norm_w = lw.norm1; a.eps = sh.rms_eps;
            a.q_out = xq; a.k_cache = layer_k(s, l); a.v_cache = layer_v(s, l);
            a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin;
            a.head_dim = sh.head_dim; a.q_dim = s->q_dim; a.kv_dim = s->kv_dim; a.t_max = s->buf.t_max; a.v_transposed = 1;
            if (!add_lin(KG_PH_QKV, OP_QKV, a)) return 0;
        }
        {
            KgPhase p;
            memset(&p, 0, sizeof(p));
            p.kind = KG_PH_ATTN; p.nwaves = 8;
            KgAttnArgs& a = p.attn;
            a.dyn = dyn; a.q = xq; a.k_cache = layer_k(s, l); a.v_cache = layer_v(s, l);
            a.H = sh.num_heads; a.Hkv = sh.num_kv_heads; a.hd = sh.head_dim; a.t_max = s->buf.t_max; a.v_transposed = 1;
            a.nsplit = s->attn_nsplit; a.blocks_per_head = s->attn_blocks; a.teams = s->attn_teams; a.part_o = s->part_o; a.part_ml = s->part_ml; a.counters = s->attn_counters; a.out = ao; a.max_rows = max_rows;
            prog.push_back(p);
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wo, d, s->q_dim, ao, s->q_dim);
            a.out = hmid; a.out_stride = d; a.resid = cur; a.resid_stride = d;
            if (!add_lin(KG_PH_O, OP_O, a)) return 0;
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wgu, 2 * F, d, hmid, d);
            a.norm_w = lw.norm2; a.eps = sh.rms_eps;
            a.out = act; a.out_stride = F;
            if (!add_lin(KG_PH_GATEUP, OP_GATEUP, a)) return 0;
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wdown, d, F, act, F);
            a.out = hout; a.out_stride = d; a.resid = hmid; a.resid_stride = d;
            if (!add_lin(KG_PH_DOWN, OP_DOWN, a)) return 0;
        }
        cur = hout;
    }
    if (with_head) {
        KgLinArgs a = lin_base(dyn, s->w.lm_head, sh.vocab_size, d, cur, d);
        a.norm_w = s->w.final_norm; a.eps = sh.rms_eps;
        a.head_part = s->head_part; a.head_part_stride = s->head_part_stride;
        a.logits_stride = sh.vocab_size;
        if (which == 0 && (s->accept_top_k > 1 || s->accept_top_p > 0.f || s->accept_typical)) a.logits_out = s->vlogits;
        if (!add_lin(KG_PH_HEAD, OP_HEAD, a)) return 0;
    }
    if (kg_persist_max_blocks(sh.dtype, smem) < nblk) return 0;
    KG_CHECK_HIP(hipMemcpy(s->prog_dev[which], prog.data(), prog.size() * sizeof(KgPhase), hipMemcpyHostToDevice));
    s->prog_n[which] = (int)prog.size();
    s->prog_kinds[which].clear();
    for (auto& p : prog) s->prog_kinds[which].push_back(p.kind);
    s->prog_smem[which] = smem;
    return 0;
}

static int run_program(kg_session* s, int which, hipStream_t st) {
    // cleared by a kernel: this runs inside the captured round graph, where a memset node does not reliably finish
    // before the kernel node that follows it
    KG_TRY(kg_launch_zero_u32(s->bar, 256, st));
    return kg_launch_persist(s->prog_dev[which], s->prog_n[which], s->bar, s->sh.dtype, g_ncu, s->prog_smem[which], s->tlog, st);
}

// ---------------------------------------------------------------------------------
// generate: prefill + graph-replayed rounds
// ---------------------------------------------------------------------------------
static int enqueue_round(kg_session* s, int steps, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, L = sh.exit_layer, N = sh.num_layers;
    const size_t rowb = (size_t)d * s->elt;
    KgLoopBufs b = loop_bufs(s);
    // the round's start state was left by the previous round's accept kernel (kg_generate: by a
    // kg_round_begin launch after the prefill)
    for (int k = 0; k <= steps; ++k) {
        // shallow pass on the newest token -> exited[k]; its embedding row was put into s->h by the control
        // kernel that produced the token (previous step's kg_step_end, previous round's kg_accept)
        const void* cur = s->h;
        for (int l = 0; l < L; ++l) {
            void* dst = (l == L - 1) ? (void*)(s->exited + k * rowb) : (void*)s->h;
            KG_TRY(run_layer(s, s->layers[l], layer_k(s, l), layer_v(s, l), s->dyn_shallow, 1, cur, dst, st));
            cur = dst;
        }
        if (k == steps) break;
        if (k == 0) KG_TRY(kg_launch_step_mid(b, k, st));
        if (k == 0) {
            KG_TRY(run_layer(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, s->dyn_adapter, 2, s->adapter_in, s->adapter_h, st));
        } else {
            KG_TRY(run_layer(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, s->dyn_adapter, 1, s->exited + k * rowb, s->adapter_h, st));
        }
        int nb = 0;
        KG_TRY(run_head(s, s->dyn_adapter, 1, s->adapter_h, 1, s->w.adapter_norm, nullptr, &nb, st));
        KG_TRY(kg_launch_step_end(b, k, s->head_part, nb, st));
    }
    const int vrows = steps + 1 > 1 ? steps + 1 : 2;
    int nb = 0;
    if (s->prog_n[0] > 0) {
        KG_TRY(run_program(s, 0, st));
        nb = persist_head_records(s);
    } else {
        const void* cur = s->exited;
        for (int l = L; l < N; ++l) {
            KG_TRY(run_layer_at(s, l, s->dyn_verify, vrows, cur, s->h2, st));
            cur = s->h2;
        }
        KG_TRY(tp_pass_end(s, s->dyn_verify, st));
        KG_TRY(run_head(s, s->dyn_verify, vrows, s->h2, 0, s->w.final_norm, (s->accept_top_k > 1 || s->accept_top_p > 0.f || s->accept_typical) ? s->vlogits : nullptr, &nb, st));
    }
    if (s->accept_typical) KG_TRY(kg_launch_typical(b, s->vlogits, sh.vocab_size, s->temperature, s->typical_epsilon, s->typical_delta, s->topk_rank, sh.dtype, st));
    else if (s->accept_top_p > 0.f) KG_TRY(kg_launch_topp_member(b, s->vlogits, sh.vocab_size, s->temperature, s->accept_top_p, s->topk_rank + KG_MAX_ROWS, sh.dtype, st));
    else if (s->accept_top_k > 1) KG_TRY(kg_launch_topk_rank(b, s->vlogits, sh.vocab_size, s->topk_rank, sh.dtype, st));
    KG_TRY(kg_launch_accept(b, s->head_part, s->head_part_stride, nb, s->topk_rank, s->accept_typical ? -1 : (s->accept_top_p > 0.f ? 0 : s->accept_top_k), st));
    return 0;
}

static int enqueue_vanilla(kg_session* s, hipStream_t st) {
    // Plain greedy decoding, one token per replay.  The shallow layers run the q=1 GEMV
    // kernels exactly as a draft step does; the remaining layers and the head run the
    // MFMA kernels with one valid row, i.e. the arithmetic of a verify row.  Token for
    // token the result is therefore bit-identical to the Kangaroo loop's output.
    const kg_shape& sh = s->sh;
    KgLoopBufs b = loop_bufs(s);
    // the step's state and input row were left by the previous step's kg_vanilla_end (kg_generate: by a
    // kg_vanilla_begin launch after the prefill)
    int nb = 0;
    if (s->vanilla_gemv) {
        // not the comparator of the losslessness tests: the remaining layers in GEMV arithmetic differ from
        // a verify row in summation order; this is the fastest plain decoding these kernels offer
        for (int l = 0; l < sh.num_layers; ++l)
            KG_TRY(run_layer_at(s, l, s->dyn_shallow, 1, s->h, s->h, st));
        KG_TRY(tp_pass_end(s, s->dyn_shallow, st));
        KG_TRY(run_head(s, s->dyn_shallow, 1, s->h, 0, s->w.final_norm, nullptr, &nb, st));
    } else if (s->prog_n[1] > 0) {
        for (int l = 0; l < sh.exit_layer; ++l)
            KG_TRY(run_layer(s, s->layers[l], layer_k(s, l), layer_v(s, l), s->dyn_shallow, 1, s->h, s->h, st));
        KG_TRY(run_program(s, 1, st));
        nb = persist_head_records(s);
    } else {
        for (int l = 0; l < sh.num_layers; ++l)
            KG_TRY(run_layer_at(s, l, s->dyn_shallow, l < sh.exit_layer ? 1 : 2, s->h, s->h, st));
        KG_TRY(tp_pass_end(s, s->dyn_shallow, st));
        KG_TRY(run_head(s, s->dyn_shallow, 2, s->h, 0, s->w.final_norm, nullptr, &nb, st));
    }
    KG_TRY(kg_launch_vanilla_end(b, s->head_part, nb, st));
    return 0;
}

static int capture(kg_session* s, int mode, int steps, hipGraphExec_t* out) {
    hipStream_t st = s->own_stream;
    hipGraph_t g = nullptr;
    KG_CHECK_HIP(hipStreamSynchronize(st));
    if (mode == 0) KG_TRY(build_program(s, 0, s->dyn_verify, steps + 1 > 1 ? steps + 1 : 2, s->sh.exit_layer, s->sh.num_layers, s->exited, s->h2, true));
    else KG_TRY(build_program(s, 1, s->dyn_shallow, 2, s->sh.exit_layer, s->sh.num_layers, s->h, s->h2, true));
    KG_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    int rc = (mode == 0) ? enqueue_round(s, steps, st) : enqueue_vanilla(s, st);
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != 0) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { kg_set_error("hipStreamEndCapture", e); return -2; }
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { kg_set_error("hipGraphInstantiate", e); return -2; }
    return 0;
}

// One decoder layer on the <= KG_GEMM_ROWS rows described by dyn, through the prefill GEMM
// kernels; in place on h.  lw.wgu == NULL: attention half only (the adapter layer).
// xn_ready: pf_xn already holds RMSNorm(h_in) with this layer's norm1 (left there by the previous layer's
// last projection); next_norm1: the gamma the layer's last projection should do the same with for the next layer.
static int run_layer_gemm(kg_session* s, const kg_layer_weights& lw, void* kc, void* vc, const KgDyn* dyn,
                          const void* h_in, void* h_out, hipStream_t st, bool xn_ready = false, const void* next_norm1 = nullptr) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, F = sh.intermediate_size, dt = sh.dtype;
    // The pass's rows are normalised once (not once per block that stages them), in the reference's order:
    // normalise, round, scale (adapter.py:254-259); where a projection's K ranges are summed by a kernel of
    // their own, that kernel also normalises the finished rows for the next projection.
    if (!xn_ready) KG_TRY(kg_launch_rmsnorm(dyn, h_in, d, lw.norm1, sh.rms_eps, d, s->pf_xn, d, dt, st));
    {
        KgLinArgs a = lin_base(dyn, lw.wqkv, s->q_dim + 2 * s->kv_dim, d, s->pf_xn, d);
        a.q_out = s->pf_xq; a.k_cache = kc; a.v_cache = vc;
        a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin;
        a.head_dim = sh.head_dim; a.q_dim = s->q_dim; a.kv_dim = s->kv_dim; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
        KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_ROPE_KV, st));
    }
    {
        KgAttnArgs a;
        memset(&a, 0, sizeof(a));
        a.dyn = dyn; a.q = s->pf_xq; a.k_cache = kc; a.v_cache = vc;
        a.H = sh.num_heads; a.Hkv = sh.num_kv_heads; a.hd = sh.head_dim; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
        a.nsplit = s->attn_nsplit; a.blocks_per_head = s->attn_blocks; a.teams = s->attn_teams; a.part_o = s->part_o; a.part_ml = s->part_ml; a.counters = s->attn_counters;
        a.out = s->pf_ao; a.max_rows = KG_GEMM_ROWS;
        KG_TRY(kg_launch_attn(a, dt, st));
    }
    {
        KgLinArgs a = lin_base(dyn, lw.wo, d, s->q_dim, s->pf_ao, s->q_dim);
        a.out = h_out; a.out_stride = d; a.resid = h_in; a.resid_stride = d; a.kpart = s->pf_part;
        if (lw.wgu != nullptr) { a.out_norm_w = lw.norm2; a.out_norm = s->pf_xn; a.eps = sh.rms_eps; }
        KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_RESID, st));
    }
    if (lw.wgu == nullptr) return 0;
    {
        KgLinArgs a = lin_base(dyn, lw.wgu, 2 * F, d, s->pf_xn, d);
        a.out = s->pf_act; a.out_stride = F;
        KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_SILU_MUL, st));
    }
    {
        KgLinArgs a = lin_base(dyn, lw.wdown, d, F, s->pf_act, F);
        a.out = h_out; a.out_stride = d; a.resid = h_out; a.resid_stride = d; a.kpart = s->pf_part;
        if (next_norm1) { a.out_norm_w = next_norm1; a.out_norm = s->pf_xn; a.eps = sh.rms_eps; }
        KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_RESID, st));
    }
    return 0;
}

// The same for a sharded layer (kg_session_set_tp): the row-split projections stop at the fp32 sum of their K ranges,
// written into the shard's slot of the exchange buffer; kg_tp_reduce_kernel adds the ranks and the residual.
static int run_layer_gemm_tp(kg_session* s, int l, const KgDyn* dyn, const void* h_in, void* h_out, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, dt = sh.dtype, hd = sh.head_dim, li = l - sh.exit_layer, e0 = 2 * li;
    const int qd = s->tp.hl * hd, kvd = s->tp.hkvl * hd, fl = s->tp.fl;
    const size_t cache_layer = (size_t)s->tp.hkvl * s->buf.t_max * hd * s->elt;
    KG_TRY(kg_launch_rmsnorm(dyn, h_in, d, s->tp.layers[0][li].norm1, sh.rms_eps, d, s->pf_xn, d, dt, st));
    for (int j = 0; j < s->tp.n_local; ++j) {
        const kg_layer_weights& lw = s->tp.layers[j][li];
        void* kc = (char*)s->tp.kc[j] + (size_t)li * cache_layer;
        void* vc = (char*)s->tp.vc[j] + (size_t)li * cache_layer;
        {
            KgLinArgs a = lin_base(dyn, lw.wqkv, qd + 2 * kvd, d, s->pf_xn, d);
            a.q_out = s->pf_xq; a.k_cache = kc; a.v_cache = vc;
            a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin;
            a.head_dim = hd; a.q_dim = qd; a.kv_dim = kvd; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
            KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_ROPE_KV, st));
        }
        {
            KgAttnArgs a;
            memset(&a, 0, sizeof(a));
            a.dyn = dyn; a.q = s->pf_xq; a.k_cache = kc; a.v_cache = vc;
            a.H = s->tp.hl; a.Hkv = s->tp.hkvl; a.hd = hd; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
            a.nsplit = s->attn_nsplit; a.blocks_per_head = s->attn_blocks; a.teams = s->attn_teams; a.part_o = s->part_o; a.part_ml = s->part_ml; a.counters = s->attn_counters;
            a.out = s->pf_ao; a.max_rows = KG_GEMM_ROWS;
            KG_TRY(kg_launch_attn(a, dt, st));
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wo, d, qd, s->pf_ao, qd);
            a.out = tp_slot(s, e0, s->tp.rank[j]); a.out_stride = d; a.kpart = s->pf_part; a.part_only = 1;
            KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_RESID, st));
        }
    }
    KG_TRY(tp_reduce(s, dyn, KG_GEMM_ROWS, e0, h_in, h_out, st));
    KG_TRY(kg_launch_rmsnorm(dyn, h_out, d, s->tp.layers[0][li].norm2, sh.rms_eps, d, s->pf_xn, d, dt, st));
    for (int j = 0; j < s->tp.n_local; ++j) {
        const kg_layer_weights& lw = s->tp.layers[j][li];
        {
            KgLinArgs a = lin_base(dyn, lw.wgu, 2 * fl, d, s->pf_xn, d);
            a.out = s->pf_act; a.out_stride = fl;
            KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_SILU_MUL, st));
        }
        {
            KgLinArgs a = lin_base(dyn, lw.wdown, d, fl, s->pf_act, fl);
            a.out = tp_slot(s, e0 + 1, s->tp.rank[j]); a.out_stride = d; a.kpart = s->pf_part; a.part_only = 1;
            KG_TRY(kg_launch_gemm(a, dt, PRO_NONE, EPI_RESID, st));
        }
    }
    return tp_reduce(s, dyn, KG_GEMM_ROWS, e0 + 1, h_out, h_out, st);
}

// Prefill: all layers over the prompt in passes of C rows (128-row GEMM passes, or prefill_rows
// rows through the decode kernels); keeps the input of the exit layer for the adapter's cache
// (inference_kangaroo.py:30-38).
// One decoder layer over `rows` prompt rows at positions 0 .., every projection ONE matrix product over all rows
// (csrc/mm.hip reading the packed weights), the row kernels of prefill.hip between them, attention in 128-row launches.
// lw.wgu == NULL: attention half only (the adapter layer).  h_in / h_out may alias.
static int run_layer_rows(kg_session* s, const kg_layer_weights& lw, void* kc, void* vc, int rows, const void* h_in, void* h_out, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, F = sh.intermediate_size, dt = sh.dtype, nqkv = s->q_dim + 2 * s->kv_dim;
    auto mm = [&](const void* A, int K, const void* Wp, int N, void* C, int ldc, int epi, const void* resid) -> int {
        KgMmArgs a;
        memset(&a, 0, sizeof(a));
        a.A = A; a.B = Wp; a.C = C; a.M = rows; a.N = N; a.K = K; a.lda = K; a.ldc = ldc; a.batch = 1; a.batch_inner = 1; a.b_inner_div = 1;
        a.b_packed = 1; a.alpha = 1.0f; a.epi = epi; a.resid = resid; a.ldr = d;
        return kg_launch_mm(a, dt, st);
    };
    // The d-column projections (o, down) have 16 column tiles at d = 4096: a short prompt leaves most CUs without a tile.
    // Their K is then cut into ranges (grid.y) that write fp32 partial products; the kernel that adds the ranges also adds
    // the residual and, after o, normalises the finished rows for gate/up (it replaces that norm kernel).
    // Cost of n ranges in steps of 64 k: rounds of blocks x steps per range, plus what the adding kernel reads.
    auto ranges = [&](int K) -> int {
        if (s->rows_ksplit == 1) return 1;
        const int tiles = ((rows + 127) / 128) * ((d + 255) / 256), steps = (K + 63) / 64, cap = (int)(2 * (size_t)s->buf.t_max / (size_t)rows);
        if (s->rows_ksplit > 1) return std::max(1, std::min(std::min(s->rows_ksplit, cap), steps));
        int best = 1; double best_c = 1e30;
        for (int n = 1; n <= 16 && n <= cap && n <= steps; ++n) {
            const double c = (double)((tiles * n + g_ncu - 1) / g_ncu) * ((steps + n - 1) / n) + (n > 1 ? 5.0 + 3.3 * n * ((double)rows * d / 2.6e6) : 0.0);
            if (c < best_c) { best_c = c; best = n; }
        }
        return best;
    };
    auto mm_ranges = [&](const void* A, int K, const void* Wp, int n) -> int {
        KgMmArgs a;
        memset(&a, 0, sizeof(a));
        a.A = A; a.B = Wp; a.C = s->pm_part; a.M = rows; a.N = d; a.K = K; a.lda = K; a.ldc = d; a.batch = 1; a.batch_inner = 1; a.b_inner_div = 1;
        a.b_packed = 1; a.alpha = 1.0f; a.out_f32 = 1; a.ksplit = n; a.sCk = (long long)rows * d; a.tile = 4;
        return kg_launch_mm(a, dt, st);
    };
    KG_TRY(kg_launch_rows_rmsnorm(h_in, d, lw.norm1, sh.rms_eps, d, s->pm_xn, d, rows, dt, st));
    KG_TRY(mm(s->pm_xn, d, lw.wqkv, nqkv, s->pm_qkv, nqkv, MM_EPI_STORE, nullptr));
    KG_TRY(kg_launch_rows_rope_kv(s->pm_qkv, rows, 0, sh.head_dim, s->q_dim, s->kv_dim, s->w.rope_cos, s->w.rope_sin, s->pm_q, kc, vc, s->buf.t_max, s->v_transposed, dt, st));
    KG_TRY(kg_launch_attn_rows(s->pm_q, kc, vc, s->pm_ao, rows, 0, sh.num_heads, sh.num_kv_heads, sh.head_dim, s->buf.t_max, dt, st));
    const int n_o = ranges(s->q_dim);
    if (n_o > 1) {
        KG_TRY(mm_ranges(s->pm_ao, s->q_dim, lw.wo, n_o));
        KG_TRY(kg_launch_rows_reduce(s->pm_part, n_o, (long long)rows * d, d, h_in, d, h_out, d, lw.wgu ? lw.norm2 : nullptr, sh.rms_eps, s->pm_xn, d, rows, dt, st));
        if (lw.wgu == nullptr) return 0;
    } else {
        KG_TRY(mm(s->pm_ao, s->q_dim, lw.wo, d, h_out, d, MM_EPI_RESID, h_in));
        if (lw.wgu == nullptr) return 0;
        KG_TRY(kg_launch_rows_rmsnorm(h_out, d, lw.norm2, sh.rms_eps, d, s->pm_xn, d, rows, dt, st));
    }
    KG_TRY(mm(s->pm_xn, d, lw.wgu, 2 * F, s->pm_act, F, MM_EPI_SILU_MUL, nullptr));
    const int n_d = ranges(F);
    if (n_d > 1) {
        KG_TRY(mm_ranges(s->pm_act, F, lw.wdown, n_d));
        return kg_launch_rows_reduce(s->pm_part, n_d, (long long)rows * d, d, h_out, d, h_out, d, nullptr, sh.rms_eps, nullptr, 0, rows, dt, st);
    }
    return mm(s->pm_act, F, lw.wdown, d, h_out, d, MM_EPI_RESID, h_out);
}

// The prompt pass layer by layer (weights streamed once per layer whatever the prompt length); see prefill().
static int prefill_rows_pass(kg_session* s, int T0, bool with_adapter, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, L = sh.exit_layer, N = sh.num_layers;
    const size_t rowb = (size_t)d * s->elt;
    KG_TRY(kg_launch_rows_embed(s->tokens, 0, T0, s->w.embed, d, s->pm_h, st));
    const void* cur = s->pm_h;
    for (int l = 0; l < N; ++l) {
        void* dst = (l == L - 1) ? (void*)s->hs_early : (void*)s->pm_h;          // the exit layer's input rows are kept for the adapter
        KG_TRY(run_layer_rows(s, s->layers[l], layer_k(s, l), layer_v(s, l), T0, cur, dst, st));
        cur = dst;
    }
    // first new token from the last row's logits (:33-34)
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, 1, T0 - 1, st));
    int nb = 0;
    KG_TRY(run_head(s, s->dyn_call, 1, (const char*)cur + (size_t)(T0 - 1) * rowb, 0, s->w.final_norm, nullptr, &nb, st));
    KG_TRY(kg_launch_vanilla_end(loop_bufs(s), s->head_part, nb, st));
    if (with_adapter) KG_TRY(run_layer_rows(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, T0, s->hs_early, s->pm_h, st));
    return 0;
}

// The whole-prompt pass behind the stepwise interface (the prefill call of evaluation/inference_kangaroo.py:30-35 made through
// the mirror classes): tokens [0, n) through all layers, the exit layer's input rows and the post-norm last hidden rows to the
// caller, logits of all rows (one matrix product with the head) when asked for.
extern "C" int kg_prefill_rows(kg_session* s, int n_tokens, void* early_out, void* final_out, void* logits_out, void* stream) {
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    if (!s || !early_out || !final_out) return fail(-1, "null argument");
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, L = sh.exit_layer, N = sh.num_layers, dt = sh.dtype;
    if (n_tokens < 1 || n_tokens > s->buf.t_max) return fail(-1, "token count exceeds the cache");
    if (s->rows_prefill_min <= 0 || s->tp.size != 0 || L >= N) return fail(-3, "the whole-prompt pass is not available for this session (head_dim < 32, a cache length that is no multiple of 32, sizes that are no multiples of 8, or tensor-parallel layers)");
    KG_TRY(kg_launch_rows_embed(s->tokens, 0, n_tokens, s->w.embed, d, s->pm_h, st));
    const void* cur = s->pm_h;
    for (int l = 0; l < N; ++l) {
        void* dst = (l == L - 1) ? (void*)s->hs_early : (void*)s->pm_h;
        KG_TRY(run_layer_rows(s, s->layers[l], layer_k(s, l), layer_v(s, l), n_tokens, cur, dst, st));
        cur = dst;
    }
    KG_CHECK_HIP(hipMemcpyAsync(early_out, s->hs_early, (size_t)n_tokens * d * s->elt, hipMemcpyDeviceToDevice, st));
    KG_TRY(kg_launch_rows_rmsnorm(cur, d, s->w.final_norm, sh.rms_eps, d, final_out, d, n_tokens, dt, st));
    if (logits_out) {
        KgMmArgs a;
        memset(&a, 0, sizeof(a));
        a.A = final_out; a.B = s->w.lm_head; a.C = logits_out; a.M = n_tokens; a.N = sh.vocab_size; a.K = d; a.lda = d; a.ldc = sh.vocab_size;
        a.batch = 1; a.batch_inner = 1; a.b_inner_div = 1; a.b_packed = 1; a.alpha = 1.0f;
        if (kg_launch_mm(a, dt, st)) return fail(-2, "kg_prefill_rows: the head product failed to launch");
    }
    return 0;
}

// The adapter over a whole prompt (positions 0 .. n_rows - 1, evaluation/inference_kangaroo.py:36-38): one matrix product per
// projection, the adapter's norm on every row.
extern "C" int kg_adapter_forward_rows(kg_session* s, const void* x_in, int n_rows, void* h_out, void* stream) {
    g_err.clear();
    hipStream_t st = (hipStream_t)stream;
    if (!s || !x_in || !h_out) return fail(-1, "null argument");
    const kg_shape& sh = s->sh;
    if (n_rows < 1 || n_rows > s->buf.t_max) return fail(-1, "row count exceeds the cache");
    if (s->rows_prefill_min <= 0) return fail(-3, "the whole-prompt pass is not available for this session");
    KG_TRY(run_layer_rows(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, n_rows, x_in, s->pm_h, st));
    return kg_launch_rows_rmsnorm(s->pm_h, sh.hidden_size, s->w.adapter_norm, sh.rms_eps, sh.hidden_size, h_out, sh.hidden_size, n_rows, sh.dtype, st);
}

static int prefill(kg_session* s, int T0, bool with_adapter, hipStream_t st) {
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, L = sh.exit_layer, N = sh.num_layers;
    if (s->rows_prefill_min > 0 && T0 >= s->rows_prefill_min && s->tp.size == 0 && L < N) return prefill_rows_pass(s, T0, with_adapter, st);
    const bool gemm = s->gemm_prefill && T0 > s->prefill_rows;
    const int C = gemm ? KG_GEMM_ROWS : s->prefill_rows;
    char* hb = gemm ? s->pf_h : s->h;
    const size_t rowb = (size_t)d * s->elt;
    for (int c0 = 0; c0 < T0; c0 += C) {
        const int n = (T0 - c0 < C) ? (T0 - c0) : C;
        KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n, c0, st));
        KG_TRY(kg_launch_embed(s->dyn_call, s->tokens, s->w.embed, d, hb, d, st));
        const void* cur = hb;
        for (int l = 0; l < N; ++l) {
            void* dst = (l == L - 1) ? (void*)(s->hs_early + (size_t)c0 * rowb) : (void*)hb;
            if (gemm && is_tp_layer(s, l)) KG_TRY(run_layer_gemm_tp(s, l, s->dyn_call, cur, dst, st));
            else if (gemm) KG_TRY(run_layer_gemm(s, s->layers[l], layer_k(s, l), layer_v(s, l), s->dyn_call, cur, dst, st, l > 0,
                                                 (l + 1 < N && !is_tp_layer(s, l + 1)) ? s->layers[l + 1].norm1 : nullptr));
            else KG_TRY(run_layer_at(s, l, s->dyn_call, C, cur, dst, st));
            cur = dst;
        }
        KG_TRY(tp_pass_end(s, s->dyn_call, st));
    }
    // first new token from the last row's logits (:33-34); dyn_call still describes the last pass
    int nb = 0;
    KG_TRY(run_head(s, s->dyn_call, 1, hb, 1, s->w.final_norm, nullptr, &nb, st));
    KG_TRY(kg_launch_vanilla_end(loop_bufs(s), s->head_part, nb, st));
    if (with_adapter) {
        for (int c0 = 0; c0 < T0; c0 += C) {
            const int n = (T0 - c0 < C) ? (T0 - c0) : C;
            KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, n, c0, st));
            if (gemm) KG_TRY(run_layer_gemm(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, s->dyn_call, s->hs_early + (size_t)c0 * rowb, s->pf_h, st));
            else KG_TRY(run_layer(s, s->w.adapter, s->buf.ak_cache, s->buf.av_cache, s->dyn_call, C, s->hs_early + (size_t)c0 * rowb, s->h2, st));
        }
    }
    return 0;
}

extern "C" int kg_generate(kg_session* s, int mode, const int32_t* prompt_host, int context_length, int max_new_tokens,
                           int max_length, int steps, double threshold, int eos_token_id,
                           int32_t* out_tokens_host, int32_t* accept_log_host, int accept_log_capacity,
                           kg_gen_result* result, void* stream) {
    g_err.clear();
    if (!s || !prompt_host || !out_tokens_host || !result) return fail(-1, "null argument");
    if (mode != 0 && mode != 1) return fail(-1, "mode must be 0 (kangaroo) or 1 (vanilla)");
    if (s->sh.exit_layer >= s->sh.num_layers) return fail(-1, "this session holds no remaining layers (draft-only): drive it pass by pass");
    if (context_length < 1 || max_length > s->buf.t_max || context_length >= max_length) return fail(-1, "context_length / max_length out of range");
    if (mode == 0 && (steps < 0 || steps + 1 > KG_MAX_ROWS - 1)) return fail(-1, "steps out of range");
    if (max_new_tokens < 1) return fail(-1, "max_new_tokens must be >= 1");
    for (int i = 0; i < context_length; ++i)
        if (prompt_host[i] < 0 || prompt_host[i] >= s->sh.vocab_size) return fail(-1, "token id out of range");
    hipStream_t st = s->own_stream;
    // order after whatever the caller queued on its stream
    KG_CHECK_HIP(hipEventRecord(s->ev_in, (hipStream_t)stream));
    KG_CHECK_HIP(hipStreamWaitEvent(st, s->ev_in, 0));

    if (mode == 0 && (s->round_graph == nullptr || s->round_graph_steps != steps)) {
        if (s->round_graph) { hipGraphExecDestroy(s->round_graph); s->round_graph = nullptr; }
        KG_TRY(capture(s, 0, steps, &s->round_graph));
        s->round_graph_steps = steps;
    }
    if (mode == 1 && s->vanilla_graph == nullptr) KG_TRY(capture(s, 1, 0, &s->vanilla_graph));

    const int T0 = context_length;
    const int max_infer = (max_length < T0 + max_new_tokens) ? max_length : (T0 + max_new_tokens);
    int* hs = s->host_state;          // [0..63] init block, [64..127], [128..191] read-back slots
    memset(hs, 0, 64 * sizeof(int));
    hs[KS_START] = T0 - 1;            // the prefill's head step moves it to T0
    hs[KS_END] = T0;
    hs[KS_ADAPTER_LEN] = T0;
    hs[KS_ALIVE] = 1;
    hs[KS_CTX] = T0;
    hs[KS_MAX_INFER] = max_infer;
    hs[KS_STEPS] = steps;
    hs[KS_EOS] = -1;                  // the prefill token never stops the loop (:34)
    unsigned long long tb; memcpy(&tb, &threshold, 8);
    hs[KS_THRESH_LO] = (int)(unsigned)(tb & 0xffffffffu);
    hs[KS_THRESH_HI] = (int)(unsigned)(tb >> 32);
    KG_CHECK_HIP(hipMemcpyAsync(s->state, hs, KS_HDR * sizeof(int), hipMemcpyHostToDevice, st));
    KG_CHECK_HIP(hipMemcpyAsync(s->tokens, prompt_host, (size_t)T0 * sizeof(int), hipMemcpyHostToDevice, st));
    // the rest of the token buffer starts as eos (:20); only the typical rule ever reads a slot it did not write
    if (mode == 0 && s->accept_typical) {
        if (eos_token_id < 0 || eos_token_id >= s->sh.vocab_size) return fail(-1, "the typical rule needs an eos id inside the vocabulary (unwritten token slots hold it)");
        KG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(s->tokens + T0), eos_token_id, (size_t)(s->buf.t_max + 64 - T0), st));
    }

    KG_CHECK_HIP(hipMemsetAsync(s->bar + 256, 0, sizeof(unsigned), st));
    KG_CHECK_HIP(hipEventRecord(s->ev_t0, st));
    KG_TRY(prefill(s, T0, mode == 0, st));
    // now arm the real eos / reset the round counter
    hs[32 + 0] = eos_token_id; hs[32 + 1] = 0;
    KG_CHECK_HIP(hipMemcpyAsync(s->state + KS_EOS, hs + 32, sizeof(int), hipMemcpyHostToDevice, st));
    KG_CHECK_HIP(hipMemcpyAsync(s->state + KS_ROUND, hs + 33, sizeof(int), hipMemcpyHostToDevice, st));
    KG_CHECK_HIP(hipMemcpyAsync(s->state + KS_STOP, hs + 33, sizeof(int), hipMemcpyHostToDevice, st));
    if (mode == 0) KG_TRY(kg_launch_round_begin(loop_bufs(s), st));
    else KG_TRY(kg_launch_vanilla_begin(loop_bufs(s), st));
    KG_CHECK_HIP(hipEventRecord(s->ev_t1, st));

    hipGraphExec_t ge = (mode == 0) ? s->round_graph : s->vanilla_graph;
    auto finished = [&](const int* v) {
        if (v[KS_STOP]) return true;
        if (mode == 0) return !(v[KS_START] < max_infer - 1 - steps);
        return !(v[KS_START] + 1 < max_infer);
    };
    int* slot[2] = {hs + 64, hs + 128};
    auto launch = [&](int i) -> int {
        KG_CHECK_HIP(hipGraphLaunch(ge, st));
        KG_CHECK_HIP(hipMemcpyAsync(slot[i], s->state, KS_HDR * sizeof(int), hipMemcpyDeviceToHost, st));
        KG_CHECK_HIP(hipEventRecord(s->ev[i], st));
        return 0;
    };
    // state right after the prefill decides whether any round runs at all
    KG_CHECK_HIP(hipMemcpyAsync(slot[0], s->state, KS_HDR * sizeof(int), hipMemcpyDeviceToHost, st));
    KG_CHECK_HIP(hipEventRecord(s->ev[0], st));
    KG_CHECK_HIP(hipEventSynchronize(s->ev[0]));
    int final_slot = 0;
    if (!finished(slot[0])) {
        // keep one replay queued ahead of the one being waited for
        int cur = 0;
        KG_TRY(launch(cur));
        const int guard = 2 * s->buf.t_max + 16;
        for (int it = 0; it < guard; ++it) {
            KG_TRY(launch(1 - cur));
            KG_CHECK_HIP(hipEventSynchronize(s->ev[cur]));
            if (finished(slot[cur])) { final_slot = cur; break; }
            cur = 1 - cur;
            final_slot = cur;
        }
    }
    KG_CHECK_HIP(hipEventRecord(s->ev_t2, st));
    KG_CHECK_HIP(hipStreamSynchronize(st));
    if (s->prog_n[mode] > 0 && s->tlog) {
        // timeline of the last persistent launch, as seen by block 0 (100 MHz ticks -> us)
        const int n = s->prog_n[mode];
        std::vector<unsigned long long> t(2 * n + 1);
        KG_CHECK_HIP(hipMemcpy(t.data(), s->tlog, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        static const char* names[7] = {"qkv", "attn", "o", "gateup", "down", "down_gx", "head"};
        double wait_us[7] = {0}, work_us[7] = {0}; int cnt[7] = {0};
        for (int i = 0; i < n; ++i) {
            const int k = s->prog_kinds[mode][i];
            wait_us[k] += (double)(t[2 * i + 1] - t[2 * i]) * 0.01;
            work_us[k] += (double)(t[2 * i + 2] - t[2 * i + 1]) * 0.01;
            cnt[k]++;
        }
        fprintf(stderr, "[persist timeline, block 0] total %.1f us over %d phases\n", (double)(t[2 * n] - t[0]) * 0.01, n);
        for (int k = 0; k < 7; ++k) if (cnt[k]) fprintf(stderr, "  %-8s x%3d  wait %6.2f us  work %6.2f us\n", names[k], cnt[k], wait_us[k] / cnt[k], work_us[k] / cnt[k]);
    }
    if (s->prog_n[mode] > 0) {
        unsigned aborted = 0;
        KG_CHECK_HIP(hipMemcpy(&aborted, s->bar + 256, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (aborted) {
            // a grid barrier gave up: the blocks were not all resident.  Fall back to kernel-by-kernel passes.
            s->persist = 0; s->prog_n[0] = s->prog_n[1] = 0;
            if (s->round_graph) { hipGraphExecDestroy(s->round_graph); s->round_graph = nullptr; s->round_graph_steps = -1; }
            if (s->vanilla_graph) { hipGraphExecDestroy(s->vanilla_graph); s->vanilla_graph = nullptr; }
            return fail(-4, "persistent pass timed out at a grid barrier; persistent passes are now off for this session, call again");
        }
    }
    KG_TRY(tp_check_abort(s, st));
    // the replay queued after the finishing one ran with the loop condition false: a no-op
    const int* fin = slot[final_slot];
    const int sidx = fin[KS_START];
    result->n_tokens = sidx + 1;
    result->new_tokens = sidx - T0 + 1;
    result->rounds = fin[KS_ROUND];
    result->drafted_total = fin[KS_NDRAFT_TOTAL];
    result->stopped_on_eos = fin[KS_STOP];
    hipEventElapsedTime(&result->prefill_ms, s->ev_t0, s->ev_t1);
    hipEventElapsedTime(&result->decode_ms, s->ev_t1, s->ev_t2);
    KG_CHECK_HIP(hipMemcpy(out_tokens_host, s->tokens, (size_t)(sidx + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (accept_log_host && mode == 0) {
        int n = fin[KS_ROUND];
        if (n > accept_log_capacity) n = accept_log_capacity;
        if (n > 0) KG_CHECK_HIP(hipMemcpy(accept_log_host, s->state + KS_HDR, (size_t)n * 2 * sizeof(int), hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int kg_bench_op(kg_session* s, int op, int rows, int iters, float* us_per_launch, size_t* weight_bytes_per_launch) {
    g_err.clear();
    if (!s || !us_per_launch || op < 0 || op > 4 || rows < 1 || rows > KG_MAX_ROWS || iters < 1 || iters > 4096) return fail(-1, "bad argument");
    hipStream_t st = s->own_stream;
    const kg_shape& sh = s->sh;
    const int d = sh.hidden_size, F = sh.intermediate_size, dt = sh.dtype;
    const int N = (s->tp.size > 1 && s->bench_layers > sh.exit_layer) ? sh.exit_layer : s->bench_layers;     // whole layers only
    KG_TRY(kg_launch_set_dyn(s->dyn_call, 1, rows, 0, st));
    KG_CHECK_HIP(hipStreamSynchronize(st));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    KG_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    int rc = 0;
    size_t wbytes = 0;
    for (int i = 0; i < iters && rc == 0; ++i) {
        const kg_layer_weights& lw = s->layers[i % N];
        if (op == 0) {
            KgLinArgs a = lin_base(s->dyn_call, lw.wqkv, s->q_dim + 2 * s->kv_dim, d, s->h, d);
            a.norm_w = lw.norm1; a.eps = sh.rms_eps; a.q_out = s->xq; a.k_cache = layer_k(s, i % N); a.v_cache = layer_v(s, i % N);
            a.rope_cos = s->w.rope_cos; a.rope_sin = s->w.rope_sin; a.head_dim = sh.head_dim; a.q_dim = s->q_dim; a.kv_dim = s->kv_dim; a.t_max = s->buf.t_max; a.v_transposed = s->v_transposed;
            wbytes = (size_t)a.N * d * 2;
            rc = kg_launch_linear(a, dt, rows, PRO_RMSNORM, EPI_ROPE_KV, grid_for(s, a.N, rows, OP_QKV), waves_for(s, rows, OP_QKV), st);
        } else if (op == 1) {
            KgLinArgs a = lin_base(s->dyn_call, lw.wo, d, s->q_dim, s->attn_out, s->q_dim);
            a.out = s->h2; a.out_stride = d; a.resid = s->h; a.resid_stride = d;
            wbytes = (size_t)d * s->q_dim * 2;
            rc = kg_launch_linear(a, dt, rows, PRO_NONE, EPI_RESID, grid_for(s, a.N, rows, OP_O), waves_for(s, rows, OP_O), st);
        } else if (op == 2) {
            KgLinArgs a = lin_base(s->dyn_call, lw.wgu, 2 * F, d, s->h, d);
            a.norm_w = lw.norm2; a.eps = sh.rms_eps; a.out = s->act; a.out_stride = F;
            wbytes = (size_t)2 * F * d * 2;
            rc = kg_launch_linear(a, dt, rows, PRO_RMSNORM, EPI_SILU_MUL, grid_for(s, a.N, rows, OP_GATEUP), waves_for(s, rows, OP_GATEUP), st);
        } else if (op == 3) {
            KgLinArgs a = lin_base(s->dyn_call, lw.wdown, d, F, s->act, F);
            a.out = s->h2; a.out_stride = d; a.resid = s->h; a.resid_stride = d;
            wbytes = (size_t)d * F * 2;
            rc = kg_launch_linear(a, dt, rows, PRO_NONE, EPI_RESID, grid_for(s, a.N, rows, OP_DOWN), waves_for(s, rows, OP_DOWN), st);
        } else {
            int nb = 0;
            wbytes = (size_t)sh.vocab_size * d * 2;
            rc = run_head(s, s->dyn_call, rows, s->h, 0, s->w.final_norm, nullptr, &nb, st);
        }
    }
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != 0) { if (g) hipGraphDestroy(g); return fail(rc, "launch refused"); }
    if (e != hipSuccess) { kg_set_error("hipStreamEndCapture", e); return -2; }
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { kg_set_error("hipGraphInstantiate", e); return -2; }
    KG_CHECK_HIP(hipGraphLaunch(ge, st));              // warm-up
    KG_CHECK_HIP(hipEventRecord(s->ev_t0, st));
    KG_CHECK_HIP(hipGraphLaunch(ge, st));
    KG_CHECK_HIP(hipEventRecord(s->ev_t1, st));
    KG_CHECK_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    hipEventElapsedTime(&ms, s->ev_t0, s->ev_t1);
    hipGraphExecDestroy(ge);
    *us_per_launch = ms * 1000.f / iters;
    if (weight_bytes_per_launch) *weight_bytes_per_launch = wbytes;
    return 0;
}

// ---------------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------------
extern "C" int kg_linear(int dtype, const void* w_packed, int N, int K, const void* x, int n_rows,
                         int prologue, const void* norm_w, float eps,
                         int epilogue, const void* resid, void* out, int out_stride,
                         void* scratch, int nblocks, int nwaves, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!w_packed || !x || !out || !scratch) return fail(-1, "null argument");
    // n_rows == -1: ONE row in the arithmetic of a multi-row pass (the MFMA kernel with one valid row), so that a row
    // computed alone and the same row inside a verify pass agree bit for bit
    const int min_rows = (n_rows == -1) ? 2 : 1;
    if (n_rows == -1) n_rows = 1;
    if (n_rows < 1 || n_rows > KG_GEMM_ROWS) return fail(-1, "n_rows must be in [1, 128] (or -1)");
    if (epilogue != EPI_STORE && epilogue != EPI_RESID && epilogue != EPI_SILU_MUL && epilogue != EPI_STORE_F32) return fail(-1, "epilogue must be 0, 1, 3 or 5");
    if (epilogue == EPI_STORE_F32 && (n_rows > KG_MAX_ROWS || prologue != PRO_NONE)) return fail(-1, "epilogue 5 (fp32 partial sums) takes at most 16 rows and no prologue");
    if (prologue == PRO_RMSNORM && !norm_w) return fail(-1, "norm_w required");
    if (epilogue == EPI_RESID && !resid) return fail(-1, "resid required");
    hipStream_t st = (hipStream_t)stream;
    KgDyn* dyn = (KgDyn*)scratch;
    KG_TRY(kg_launch_set_dyn(dyn, 1, n_rows, 0, st));
    KgLinArgs a = lin_base(dyn, w_packed, N, K, x, K);
    a.norm_w = norm_w; a.eps = eps; a.out = out; a.out_stride = out_stride; a.resid = resid; a.resid_stride = out_stride;
    if (n_rows > KG_MAX_ROWS && prologue == PRO_RMSNORM) {
        // prefill-sized passes normalise their rows once, into the scratch buffer (kg_linear_scratch_bytes)
        void* xn = (char*)scratch + 256;
        KG_TRY(kg_launch_rmsnorm(dyn, x, K, norm_w, eps, K, xn, K, dtype, st));
        a.x = xn; prologue = PRO_NONE;
    }
    int rc = (n_rows > KG_MAX_ROWS) ? kg_launch_gemm(a, dtype, prologue, epilogue, st)       // prefill-sized pass
                                    : kg_launch_linear(a, dtype, n_rows < min_rows ? min_rows : n_rows, prologue, epilogue, nblocks > 0 ? nblocks : g_ncu * 2, nwaves > 0 ? nwaves : 8, st);
    if (rc == -3) return fail(-3, "rows x K exceed LDS");
    if (rc) return fail(rc, "kg_launch_linear failed");
    return 0;
}

extern "C" size_t kg_linear_scratch_bytes(int n_rows, int K) {
    return 256 + (n_rows > KG_MAX_ROWS ? (size_t)n_rows * K * 2 : 0);
}

extern "C" int kg_qkv_rope_append(int dtype, const void* wqkv_packed, int d, int n_heads, int n_kv_heads, int head_dim,
                                  const void* x, int n_rows, int pos0, const void* norm_w, float eps,
                                  const float* rope_cos, const float* rope_sin,
                                  void* q_out, void* k_cache, void* v_cache, int t_max, int v_transposed,
                                  void* scratch, int nblocks, int nwaves, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!wqkv_packed || !x || !norm_w || !rope_cos || !rope_sin || !q_out || !k_cache || !v_cache || !scratch) return fail(-1, "null argument");
    const int min_rows = (n_rows == -1) ? 2 : 1;          // -1: one row in a multi-row pass's arithmetic (see kg_linear)
    if (n_rows == -1) n_rows = 1;
    const bool pos_dev = (pos0 == KG_POS_IN_SCRATCH);       // the first position is the int at scratch + 128, read on the device
    if (n_rows < 1 || n_rows > KG_GEMM_ROWS || (!pos_dev && (pos0 < 0 || pos0 + n_rows > t_max))) return fail(-1, "bad rows / positions");
    hipStream_t st = (hipStream_t)stream;
    KgDyn* dyn = (KgDyn*)scratch;
    if (pos_dev) KG_TRY(kg_launch_set_dyn_from(dyn, n_rows, (const int*)((const char*)scratch + 128), t_max, st));
    else KG_TRY(kg_launch_set_dyn(dyn, 1, n_rows, pos0, st));
    const int qd = n_heads * head_dim, kvd = n_kv_heads * head_dim;
    KgLinArgs a = lin_base(dyn, wqkv_packed, qd + 2 * kvd, d, x, d);
    a.norm_w = norm_w; a.eps = eps; a.q_out = q_out; a.k_cache = k_cache; a.v_cache = v_cache;
    a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.head_dim = head_dim; a.q_dim = qd; a.kv_dim = kvd; a.t_max = t_max; a.v_transposed = v_transposed ? 1 : 0;
    if (n_rows > KG_MAX_ROWS) {
        void* xn = (char*)scratch + 256;
        KG_TRY(kg_launch_rmsnorm(dyn, x, d, norm_w, eps, d, xn, d, dtype, st));
        a.x = xn;
    }
    int rc = (n_rows > KG_MAX_ROWS) ? kg_launch_gemm(a, dtype, PRO_NONE, EPI_ROPE_KV, st)
                                    : kg_launch_linear(a, dtype, n_rows < min_rows ? min_rows : n_rows, PRO_RMSNORM, EPI_ROPE_KV, nblocks > 0 ? nblocks : g_ncu * 2, nwaves > 0 ? nwaves : 8, st);
    if (rc) return fail(rc, "kg_launch_linear failed");
    return 0;
}

extern "C" size_t kg_attention_scratch_bytes(int n_heads, int head_dim, int nsplit) {
    return 256 + (size_t)nsplit * KG_GEMM_ROWS * n_heads * (head_dim + 2) * sizeof(float) + 256 + (KG_GEMM_ROWS / 16) * (size_t)n_heads * sizeof(unsigned);
}

extern "C" int kg_attention(int dtype, const void* q, int n_rows, int pos0, const void* k_cache, const void* v_cache,
                            int n_heads, int n_kv_heads, int head_dim, int t_max, int v_transposed, int nsplit,
                            void* out, void* scratch, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!q || !k_cache || !v_cache || !out || !scratch) return fail(-1, "null argument");
    const bool pos_dev = (pos0 == KG_POS_IN_SCRATCH);
    if (n_rows < 1 || n_rows > (v_transposed ? KG_GEMM_ROWS : KG_MAX_ROWS) || (!pos_dev && (pos0 < 0 || pos0 + n_rows > t_max)) || nsplit < 1 || nsplit > 32) return fail(-1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    KgDyn* dyn = (KgDyn*)scratch;
    if (pos_dev) KG_TRY(kg_launch_set_dyn_from(dyn, n_rows, (const int*)((const char*)scratch + 128), t_max, st));
    else KG_TRY(kg_launch_set_dyn(dyn, 1, n_rows, pos0, st));
    KgAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.dyn = dyn; a.q = q; a.k_cache = k_cache; a.v_cache = v_cache; a.H = n_heads; a.Hkv = n_kv_heads; a.hd = head_dim; a.t_max = t_max; a.v_transposed = v_transposed ? 1 : 0;
    a.nsplit = nsplit; a.blocks_per_head = nsplit; a.teams = 1; a.max_rows = n_rows;
    a.part_o = (float*)((char*)scratch + 256);
    a.part_ml = a.part_o + (size_t)nsplit * KG_GEMM_ROWS * n_heads * head_dim;
    a.counters = (unsigned*)(((uintptr_t)(a.part_ml + (size_t)nsplit * KG_GEMM_ROWS * n_heads * 2) + 255) & ~(uintptr_t)255);
    KG_TRY(kg_launch_zero_u32(a.counters, (KG_GEMM_ROWS / 16) * n_heads, st));
    // partial buffers are indexed with max_rows = n_rows, within the KG_GEMM_ROWS-sized scratch
    a.out = out;
    int rc = kg_launch_attn(a, dtype, st);
    if (rc) return fail(rc, "kg_launch_attn failed (head_dim 16/32/64/128; transposed V needs head_dim >= 32 and t_max % 32 == 0)");
    return 0;
}

extern "C" size_t kg_head_scratch_bytes(int vocab) {
    return 256 + (size_t)KG_MAX_ROWS * ((vocab + 15) / 16) * 4 * sizeof(float);
}

extern "C" int kg_head(int dtype, const void* w_packed, int vocab, int d, const void* x, int n_rows,
                       int prologue, const void* norm_w, float eps,
                       void* logits_out, int* argmax_out, float* pmax_out,
                       void* scratch, int nblocks, int nwaves, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!w_packed || !x || !scratch || !argmax_out || !pmax_out) return fail(-1, "null argument");
    if (n_rows < 1 || n_rows > KG_MAX_ROWS) return fail(-1, "n_rows must be in [1, 16]");
    if (prologue == PRO_RMSNORM && !norm_w) return fail(-1, "norm_w required");
    hipStream_t st = (hipStream_t)stream;
    KgDyn* dyn = (KgDyn*)scratch;
    KG_TRY(kg_launch_set_dyn(dyn, 1, n_rows, 0, st));
    KgLinArgs a = lin_base(dyn, w_packed, vocab, d, x, d);
    a.norm_w = norm_w; a.eps = eps;
    a.head_part = (float*)((char*)scratch + 256); a.head_part_stride = (vocab + 15) / 16;
    a.logits_out = logits_out; a.logits_stride = vocab;
    int nb = nblocks > 0 ? nblocks : g_ncu * 2;
    const int G = (vocab + 15) / 16;
    if (nb > G) nb = G;
    int rc = kg_launch_linear(a, dtype, n_rows, prologue, EPI_HEAD, nb, nwaves > 0 ? nwaves : 8, st);
    if (rc) return fail(rc, "kg_launch_linear failed");
    KG_TRY(kg_launch_head_finalize(dyn, a.head_part, a.head_part_stride, nb, n_rows, argmax_out, pmax_out, st));
    return 0;
}

extern "C" int kg_matmul_nt(int dtype, const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                            int batch, long long sA, long long sB, long long sC, int b_packed, int out_f32, int causal, float alpha, void* stream) {
    return kg_matmul_nt_heads(dtype, A, B, C, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, 1, 0, 0, 0, 1, b_packed, out_f32, causal, alpha, stream);
}

extern "C" int kg_matmul_nt_heads(int dtype, const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                  int batch, long long sA, long long sB, long long sC, int batch_inner, long long sA2, long long sB2, long long sC2, int b_inner_div,
                                  int b_packed, int out_f32, int causal, float alpha, void* stream) {
    g_err.clear();
    kg_init_once();
    if (dtype != KG_F16 && dtype != KG_BF16) return fail(-1, "dtype");
    KgMmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.batch = batch; a.sA = sA; a.sB = sB; a.sC = sC; a.b_packed = b_packed; a.out_f32 = out_f32; a.causal = causal; a.alpha = alpha;
    a.batch_inner = batch_inner; a.sA2 = sA2; a.sB2 = sB2; a.sC2 = sC2; a.b_inner_div = b_inner_div;
    { const char* t = getenv("KG_MM_TILE"); a.tile = t ? atoi(t) : 0; }      // measurements: 1 forces 128 x 128 tiles, 2 / 3 the two forms of 256 x 256
    const int rc = kg_launch_mm(a, dtype, (hipStream_t)stream);
    if (rc == -1) return fail(-1, "kg_matmul_nt: bad shape (K and the row / batch strides of A and B must be multiples of 8; a packed B needs K % 32 == 0 and batch 1; one product's A and B each span less than 2 GiB)");
    if (rc) return fail(rc, "kg_matmul_nt launch failed");
    return 0;
}

extern "C" int kg_matmul_nt_ranges(int dtype, const void* A, const void* B, float* parts, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                   int b_packed, int n_ranges, long long range_stride, void* stream) {
    g_err.clear();
    kg_init_once();
    if (dtype != KG_F16 && dtype != KG_BF16) return fail(-1, "dtype");
    if (n_ranges < 1 || n_ranges > 64 || b_packed == 2) return fail(-1, "kg_matmul_nt_ranges: 1..64 ranges, a row-major or packed B");
    KgMmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = B; a.C = parts; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.batch = 1; a.batch_inner = 1; a.b_inner_div = 1; a.b_packed = b_packed; a.out_f32 = 1; a.alpha = 1.0f;
    a.ksplit = n_ranges; a.sCk = range_stride;
    { const char* t = getenv("KG_MM_TILE"); a.tile = t ? atoi(t) : 0; if (a.tile != 3) a.tile = 4; }
    const int rc = kg_launch_mm(a, dtype, (hipStream_t)stream);
    if (rc == -1) return fail(-1, "kg_matmul_nt_ranges: bad shape (K and the row strides multiples of 8, range_stride >= M * ldc; a packed B needs K % 32 == 0)");
    if (rc) return fail(rc, "kg_matmul_nt_ranges launch failed");
    return 0;
}

extern "C" int kg_transpose16(const void* in, void* out, int rows, int cols, long long ld_in, long long ld_out, int batch, long long s_in, long long s_out, void* stream) {
    g_err.clear();
    const int rc = kg_launch_transpose16(in, out, rows, cols, ld_in, ld_out, batch, s_in, s_out, (hipStream_t)stream);
    if (rc) return fail(rc, "kg_transpose16: bad argument or launch failure");
    return 0;
}

// ---------------------------------------------------------------------------------
// control kernels of the loop as entry points of their own: a throw-away loop state is laid out in the
// caller's scratch buffer and the SAME kernels the round graph replays run on it
// ---------------------------------------------------------------------------------
namespace {
struct CtlScratch {
    int* state; int* tokens; KgDyn* dyn; int* head_tok; float* head_pmax; int* rank; float* head_part; int stride; int tok_cap;
};
const int KG_CTL_TOKENS = 4096;          // token slots of the scratch state
const int KG_CTL_RECORDS = 256;          // head records per row
size_t ctl_carve(CtlScratch* c, void* base) {
    Carver cv{(char*)base, 0};
    int* state = cv.take<int>(KS_HDR + 64);
    int* tokens = cv.take<int>(KG_CTL_TOKENS);
    KgDyn* dyn = cv.take<KgDyn>(3);
    int* head_tok = cv.take<int>(KG_MAX_ROWS);
    float* head_pmax = cv.take<float>(KG_MAX_ROWS);
    int* rank = cv.take<int>(KG_MAX_ROWS + KG_MAX_ROWS * KG_MAX_ROWS);
    float* head_part = cv.take<float>((size_t)KG_MAX_ROWS * KG_CTL_RECORDS * 4);
    if (c) { c->state = state; c->tokens = tokens; c->dyn = dyn; c->head_tok = head_tok; c->head_pmax = head_pmax; c->rank = rank;
             c->head_part = head_part; c->stride = KG_CTL_RECORDS; c->tok_cap = KG_CTL_TOKENS; }
    return align_up(cv.off, 256);
}
KgLoopBufs ctl_bufs(const CtlScratch& c) {
    KgLoopBufs b;
    memset(&b, 0, sizeof(b));
    b.state = c.state; b.tokens = c.tokens;
    b.dyn_shallow = c.dyn; b.dyn_adapter = c.dyn + 1; b.dyn_verify = c.dyn + 2;
    b.head_tok = c.head_tok; b.head_pmax = c.head_pmax;
    b.d = 0; b.elt = 2;                  // no hidden rows: the kernels' row copies and embedding gathers move nothing
    b.exited = c.state; b.prev_exit = c.state; b.adapter_in = c.state; b.embed = c.state; b.h = c.state;
    return b;
}
}  // namespace

extern "C" size_t kg_control_scratch_bytes(void) { return ctl_carve(nullptr, nullptr); }

extern "C" int kg_accept(int dtype, const void* logits, int vocab, int q, const int32_t* drafted_host, int start_index,
                         int eos_token_id, int rule, int top_k, float top_p, float temperature, float epsilon, float delta,
                         int steps, int max_infer, int adapter_len, int32_t* out_host, void* scratch, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!logits || !out_host || !scratch || (q > 1 && !drafted_host)) return fail(-1, "null argument");
    if (dtype != KG_F16 && dtype != KG_BF16) return fail(-1, "dtype");
    if (q < 1 || q > KG_MAX_ROWS - 1 || vocab < 1) return fail(-1, "q must be in [1, 15]");
    if (start_index < 0 || start_index + q + 1 > KG_CTL_TOKENS) return fail(-1, "start_index out of range for the scratch state");
    if (rule < 0 || rule > 3 || (rule == 1 && (top_k < 2 || top_k > vocab)) || (rule == 2 && (!(top_p > 0.f && top_p <= 1.f) || !(temperature > 0.f))) ||
        (rule == 3 && !(temperature > 0.f)))
        return fail(-1, "rule: 0 greedy, 1 top-k (2 <= k <= vocab), 2 top-p (0 < p <= 1, temperature > 0), 3 typical (temperature > 0)");
    if (start_index + 2 * q + 2 > KG_CTL_TOKENS) return fail(-1, "start_index out of range for the scratch state");
    hipStream_t st = (hipStream_t)stream;
    CtlScratch c;
    ctl_carve(&c, scratch);
    KgLoopBufs b = ctl_bufs(c);
    int hs[KS_HDR];
    memset(hs, 0, sizeof(hs));
    hs[KS_START] = start_index; hs[KS_END] = start_index + q; hs[KS_ADAPTER_LEN] = adapter_len; hs[KS_ALIVE] = 1;
    hs[KS_MAX_INFER] = max_infer; hs[KS_STEPS] = steps; hs[KS_EOS] = eos_token_id;
    { const double thr0 = 0.6; unsigned long long tb; memcpy(&tb, &thr0, 8);
      hs[KS_THRESH_LO] = (int)(unsigned)(tb & 0xffffffffu); hs[KS_THRESH_HI] = (int)(unsigned)(tb >> 32); }
    KG_CHECK_HIP(hipMemcpyAsync(c.state, hs, sizeof(hs), hipMemcpyHostToDevice, st));
    // token slots the round did not write hold eos, as the loop's buffer does from the start (:20)
    KG_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(c.tokens + start_index + 1), eos_token_id, (size_t)(2 * q + 1), st));
    if (q > 1) KG_CHECK_HIP(hipMemcpyAsync(c.tokens + start_index + 1, drafted_host, (size_t)(q - 1) * sizeof(int), hipMemcpyHostToDevice, st));
    const int G = (vocab + 15) / 16, nb = G < KG_CTL_RECORDS ? G : KG_CTL_RECORDS;
    KG_TRY(kg_launch_logit_records(logits, vocab, q, nb, c.head_part, c.stride, dtype, st));
    if (rule == 3) KG_TRY(kg_launch_typical(b, logits, vocab, temperature, epsilon, delta, c.rank, dtype, st));
    else if (rule == 2) KG_TRY(kg_launch_topp_member(b, logits, vocab, temperature, top_p, c.rank + KG_MAX_ROWS, dtype, st));
    else if (rule == 1) KG_TRY(kg_launch_topk_rank(b, logits, vocab, c.rank, dtype, st));
    KG_TRY(kg_launch_accept(b, c.head_part, c.stride, nb, c.rank, rule == 3 ? -1 : (rule == 2 ? 0 : (rule == 1 ? top_k : 1)), st));
    int hs2[KS_HDR + 2];
    KG_CHECK_HIP(hipMemcpyAsync(hs2, c.state, sizeof(hs2), hipMemcpyDeviceToHost, st));
    int tok = 0;
    KG_CHECK_HIP(hipStreamSynchronize(st));
    const int snew = hs2[KS_START];
    KG_CHECK_HIP(hipMemcpy(&tok, c.tokens + snew, sizeof(int), hipMemcpyDeviceToHost));
    out_host[0] = hs2[KS_HDR];           // accept length (:104)
    out_host[1] = tok;                   // token written at the new start index (:98)
    out_host[2] = hs2[KS_STOP];          // eos produced (:100)
    out_host[3] = snew;                  // new start index (:99)
    out_host[4] = hs2[KS_ADAPTER_LEN];   // adapter cache length after the rollback (:114-118)
    out_host[5] = hs2[KS_ALIVE];         // the next round runs (:46, :123)
    out_host[6] = hs2[KS_HDR + 1];       // drafted tokens of the round
    out_host[7] = hs2[KS_ROUND];
    out_host[8] = hs2[KS_THRESH_LO];     // the draft loop's confidence threshold after the round, as a double (0.6 going in;
    out_host[9] = hs2[KS_THRESH_HI];     //  only the typical rule changes it, MTBench :301)
    return 0;
}

extern "C" int kg_draft_step(int dtype, const void* logits, int vocab, int step, int steps, double threshold,
                             int start_index, int end_index, int adapter_len, int32_t* out_host, void* scratch, void* stream) {
    g_err.clear();
    kg_init_once();
    if (!out_host || !scratch) return fail(-1, "null argument");
    if (dtype != KG_F16 && dtype != KG_BF16) return fail(-1, "dtype");
    if (step < 0 || step >= steps || steps > KG_MAX_ROWS - 2) return fail(-1, "step must be in [0, steps)");
    if (start_index < 1 || end_index <= start_index || end_index + 1 >= KG_CTL_TOKENS) return fail(-1, "indices out of range for the scratch state");
    hipStream_t st = (hipStream_t)stream;
    CtlScratch c;
    ctl_carve(&c, scratch);
    KgLoopBufs b = ctl_bufs(c);
    int hs[KS_HDR];
    memset(hs, 0, sizeof(hs));
    hs[KS_START] = start_index; hs[KS_END] = end_index; hs[KS_ADAPTER_LEN] = adapter_len; hs[KS_ALIVE] = 1; hs[KS_DRAFT_ON] = 1;
    hs[KS_STEPS] = steps; hs[KS_DRAFTED] = end_index - start_index - 1;
    unsigned long long tb; memcpy(&tb, &threshold, 8);
    hs[KS_THRESH_LO] = (int)(unsigned)(tb & 0xffffffffu); hs[KS_THRESH_HI] = (int)(unsigned)(tb >> 32);
    KG_CHECK_HIP(hipMemcpyAsync(c.state, hs, sizeof(hs), hipMemcpyHostToDevice, st));
    KG_CHECK_HIP(hipMemsetAsync(c.dyn, 0, 3 * sizeof(KgDyn), st));
    // adapter rows of the step (:55-61): only step 0 can find the adapter cache one row short
    KG_TRY(kg_launch_step_mid(b, step, st));
    KgDyn dyn_mid;
    KG_CHECK_HIP(hipMemcpyAsync(&dyn_mid, c.dyn + 1, sizeof(KgDyn), hipMemcpyDeviceToHost, st));
    int nb = 0;
    if (logits) {
        const int G = (vocab + 15) / 16;
        nb = G < KG_CTL_RECORDS ? G : KG_CTL_RECORDS;
        KG_TRY(kg_launch_logit_records(logits, vocab, 1, nb, c.head_part, c.stride, dtype, st));
        KG_TRY(kg_launch_step_end(b, step, c.head_part, nb, st));
    }
    int hs2[KS_HDR]; KgDyn dyn[3]; int tok = -1;
    KG_CHECK_HIP(hipMemcpyAsync(hs2, c.state, sizeof(hs2), hipMemcpyDeviceToHost, st));
    KG_CHECK_HIP(hipMemcpyAsync(dyn, c.dyn, sizeof(dyn), hipMemcpyDeviceToHost, st));
    KG_CHECK_HIP(hipMemcpyAsync(&tok, c.tokens + end_index, sizeof(int), hipMemcpyDeviceToHost, st));
    KG_CHECK_HIP(hipStreamSynchronize(st));
    out_host[0] = dyn_mid.n_rows;        // adapter rows of this step (2: the missing-row patch)
    out_host[1] = dyn_mid.pos[0];        // their first position
    out_host[2] = logits ? tok : -1;     // drafted token (:80)
    out_host[3] = hs2[KS_SCORE];         // float bits of the confidence (:83)
    out_host[4] = hs2[KS_END];           // end_index after the step (:82)
    out_host[5] = hs2[KS_DRAFT_ON];      // the next step's adapter + head run (:74)
    out_host[6] = hs2[KS_ADAPTER_LEN];
    out_host[7] = dyn[2].n_rows;         // rows the verify pass will carry
    return 0;
}

// ---------------------------------------------------------------------------------
// kg_head for callers that may not allocate: the partial records and the pass parameters live in a
// file-scope device array instead of a caller-provided scratch buffer
// ---------------------------------------------------------------------------------
#define KG_NOALLOC_MAX_VOCAB 65536
#define KG_NOALLOC_BYTES (256 + (size_t)KG_MAX_ROWS * (KG_NOALLOC_MAX_VOCAB / 16) * 4 * sizeof(float))
static __device__ __align__(256) unsigned char kg_noalloc_scratch[KG_NOALLOC_BYTES];
static void* g_noalloc_ptr[64];          // per device: the array's device address (every GPU holds its own copy)

extern "C" size_t kg_head_noalloc_max_vocab(void) { return KG_NOALLOC_MAX_VOCAB; }

extern "C" int kg_head_noalloc_init(void) {
    g_err.clear();
    int dev = 0;
    KG_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(-1, "device ordinal beyond 63");
    if (!g_noalloc_ptr[dev]) {
        void* p = nullptr;
        KG_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(kg_noalloc_scratch)));
        g_noalloc_ptr[dev] = p;
    }
    kg_init_once();
    return 0;
}

extern "C" int kg_head_noalloc(int dtype, const void* w_packed, int vocab, int d, const void* x, int n_rows,
                               int prologue, const void* norm_w, float eps,
                               void* logits_out, int* argmax_out, float* pmax_out, int nblocks, int nwaves, void* stream) {
    g_err.clear();
    if (vocab < 1 || vocab > KG_NOALLOC_MAX_VOCAB) return fail(-1, "kg_head_noalloc: vocab exceeds the built-in scratch (65536 rows at most)");
    int dev = 0;
    KG_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(-1, "device ordinal beyond 63");
    if (!g_noalloc_ptr[dev]) KG_TRY(kg_head_noalloc_init());
    return kg_head(dtype, w_packed, vocab, d, x, n_rows, prologue, norm_w, eps, logits_out, argmax_out, pmax_out,
                   g_noalloc_ptr[dev], nblocks, nwaves, stream);
}
