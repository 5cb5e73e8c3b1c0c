// Shared declarations of the mpnhip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mpnhip.h"

namespace mpnhip {

void set_error(const char* fmt, ...);

#define MPN_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            mpnhip::set_error(__VA_ARGS__); \
            return MPNHIP_ERR_ARG;        \
        }                                 \
    } while (0)

#define MPN_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            mpnhip::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MPNHIP_ERR_HIP;                                                         \
        }                                                                                  \
    } while (0)

#define MPN_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            mpnhip::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return MPNHIP_ERR_HIP;                                                           \
        }                                                                                    \
    } while (0)

#define MPN_TRY(call)            \
    do {                         \
        int _r = (call);         \
        if (_r != MPNHIP_OK) return _r; \
    } while (0)

// ------------------------------------------------------------------------------------ GEMM
// C[m, n] = epilogue( sum_k A[m, k] * B[k, n] ) for up to two row groups that share N and K but have
// their own row range (read from device memory so no host sync is needed), weights and tables.
//
// A layouts:  A_KCONTIG  A[m][k] row-major (leading dim lda), optionally rows through a_idx and a
//                         second K segment [ksplit, K) taken from A2 (the reference's torch.cat of
//                         initial and current features, mpn.py:369-373, without the copy);
//             A_MCONTIG  A stored [k][m] (m contiguous, leading dim lda): C = A^T-style products.
// B layouts:  B_KCONTIG  B given as W[n][k] (nn.Linear weight, leading dim ldb);
//             B_NCONTIG  B stored [k][n] (n contiguous, leading dim ldb).
enum { A_KCONTIG = 0, A_MCONTIG = 1 };
enum { B_KCONTIG = 0, B_NCONTIG = 1 };

struct GemmGroup {
    const float* A;
    const float* A2;       // second K segment (k >= ksplit) or nullptr
    const int* a_idx;      // optional row indirection for A (A_KCONTIG only)
    const float* B;
    const float* bias;     // [N] or nullptr
    const float* G1;       // gather-add epilogue: + G1[g1_idx[m]][n]
    const int* g1_idx;     // nullptr -> identity (row m - row_begin + g_row0)
    const float* G2;
    const int* g2_idx;
    const float* mask;     // ReLU-backward mask, applied last: value = mask[m][n] > 0 ? value : 0 (ld = ldmask)
    float* C;
    const int* c_idx;      // optional row scatter for C
    const int* row_begin;  // device int: first row of this group (nullptr -> 0)
    const int* row_end;    // device int: one past the last row (nullptr -> m_static)
    int64_t lda, lda2, ldb, ldg1, ldg2, ldmask, ldc;
    int64_t m_static;      // row count when row_end == nullptr
    // bf16-operand mode, tiled kernel (gemm_bf16.hip) only: the operand is bf16 ROWS in memory -- A / A2 (a16) or B (b16) point at
    // unsigned shorts, their leading dims count them; C16: a bf16 mirror of the result, [rows][ldc16] (row m, no scatter)
    int a16, b16;
    unsigned short* C16;
    int64_t ldc16;
};

struct GemmArgs {
    GemmGroup g[2];
    int ngroups;
    int N, K, ksplit;      // ksplit == K when A2 is unused
    int relu;              // max(.,0) at the end
    int accumulate;        // C += result (after relu, before the mask)
    int64_t m_upper;       // host-side upper bound of the total row count (sizes the grid)
    int epi_vec;           // set by launch_gemm: epilogue operands allow 16-byte vector access
    int small_tiles;       // 1: take the 64 x 64 configuration whatever m_upper says (two K halves as two groups: m_upper = 2 M)
};

int launch_gemm(const GemmArgs& args, int a_layout, int b_layout, hipStream_t stream);
// MPNHIP_PREC_BF16 at large row counts / with bf16 rows in memory (gemm_bf16.hip); false: not a shape of that kernel
bool launch_gemm_bf16_tiled(const GemmArgs& args, hipStream_t stream, int* status);
// operand precision of the calling thread's GEMMs (mpnhip_model.precision): 0 fp32, 1 bf16 operands / fp32 accumulate
void set_gemm_precision(int p);
int gemm_precision();

// ------------------------------------------------------------------------------------ weight gradients
// dW[o, c] += sum_m dZ[m, o] * H[m, c], db[o] += sum_m dZ[m, o] over a (device-resident) row range.
struct TnGroup {
    const float* dZ;       // [rows, n_out], leading dim ldz
    const int* dz_idx;     // optional row gather
    const float* H;        // [rows, csplit] first column segment of the layer input
    const float* H2;       // columns >= csplit (nullptr when csplit == k_in)
    const int* h_idx;      // optional row gather
    const int* row_begin;  // device ints (nullptr -> 0 / m_static)
    const int* row_end;
    float* slab;           // split partials, tn_slab_floats() floats
    float* grad_w;         // += ; leading dim ldw (may point into a wider matrix)
    float* grad_b;         // += ; may be nullptr
    int64_t ldz, ldh, ldh2, ldw, m_static;
    // batched form (all L message-passing steps in one product): batch b reads dZ + b * z_bstride etc.
    // (a stride of 0 = the same block every step, e.g. the re-attached initial features)
    int64_t z_bstride, h_bstride, h2_bstride;
};
struct TnArgs {
    TnGroup g[2];
    int ngroups;
    int n_out, k_in, csplit;
    int64_t m_upper;       // upper bound of the rows of ONE batch
    double flops;          // algorithmic flops of the launch (2 rows n_out k_in over all batches / groups): profile hooks only
    int nbatch;            // >= 1
    int chunk, nsplit;     // filled by tn_plan: rows per chunk, chunks per batch
    int red_ny, red_stage; // slab reduction in two stages (launch_gemm_tn): ny partial sums first, then their sum
    int xcd_map;           // 1: XCD-aware block -> (tile, chunk) mapping
    int tiles, ny8;        // filled by launch_gemm_tn: output tiles, row chunks x batches padded to a multiple of 8 (XCD mapping)
};
void tn_plan(TnArgs& a);
// pitch of a slab image's rows: k_in weight columns + the bias column, padded to whole 16-byte columns
__host__ __device__ static inline int tn_kpad(int k_in) { return (k_in + 4) & ~3; }
size_t tn_slab_floats(int n_out, int k_in, int64_t m_upper, int nbatch);
int launch_gemm_tn(const TnArgs& args, hipStream_t stream);

// ---- three-piece operand form, row-panel blocks, all products of a group of steps in one launch (wgrad_panel.hip) ----
struct WpJob {             // one product (one direction group of it), as the kernels see it
    const float* dZ;       // [rows, n_out], leading dim ldz; batch b at dZ + b * z_bstride
    const float* H;        // [rows, k_in], leading dim ldh (columns [0, csplit) when H2 is given)
    const float* H2;       // columns [csplit, k_in) (row-panel variants only) or nullptr
    const int* dz_idx;     // optional row gather of dZ (wp_block_vec, wp_block_small)
    const int* h_idx;      // optional row gather of H (wp_block_small)
    const int* row_begin;  // device ints (nullptr -> 0 / m_static)
    const int* row_end;
    float* slab;           // [nbatch * nsplit][n_out][tn_kpad(k_in)]
    float* grad_w;         // += ; leading dim ldw
    float* grad_b;         // += ; may be nullptr
    int64_t ldz, ldh, z_bstride, h_bstride, ldw, m_static, ldh2, h2_bstride;
    int n_out, k_in, nbatch, csplit;
    int pieces;            // 3: fp32 operands as three bf16 pieces, six products (MPNHIP_PREC_FP32_SPLIT); 1: operands rounded to
                           // bf16, one product (MPNHIP_PREC_BF16)
    int lds2;              // 1: the launch provides two stage images (the pipelined split loop of the wide variants)
    int src16;             // 1: the operands are bf16 rows in memory (dZ / H point at unsigned shorts, leading dims and batch strides
                           // count them); the [1 x k] form: H only (its dZ is the fp32 logit gradient)
    int chunk, nsplit;     // rows per chunk, chunks per batch
    int variant;           // block tile shape (wgrad_panel.hip kVariants)
    int tiles_o, tiles_c;  // output tiles of that shape
    int block0;            // first block of the job in the product launch
    int red_block0;        // ... in the slab-sum launch
};
constexpr int WP_MAX_JOBS = 16;
struct WpTable { WpJob job[WP_MAX_JOBS]; int njobs; int debug; };   // debug: timing ablations (MPNHIP_WP_DEBUG; results wrong)
struct WpProduct {         // host-side description of one product
    const float* dZ; int64_t ldz, z_bstride;
    const float* H; int64_t ldh, h_bstride;
    const int* row_begin; const int* row_end;   // device-side row range (a direction group) or nullptr
    int64_t rows;          // rows of one batch (upper bound when ranged)
    int nbatch, n_out, k_in;
    float* grad_w; int64_t ldw; float* grad_b;
    const int* dz_idx;     // optional row gathers (narrow products and the [1 x k] form only)
    const int* h_idx;
    const float* H2; int64_t ldh2, h2_bstride; int csplit;   // second column segment of H (nullptr: none)
    int pieces;            // 3 (fp32 from three bf16 pieces) or 1 (bf16-rounded operands); 0 = 3
    int src16;             // bf16 source rows (WpJob::src16); needs pieces == 1
};
// (nblocks2 / bytes2: the jobs of wgrad_rows16.hip's kernel -- variant >= 16, their block0 counts inside that launch)
struct WpBatch { WpTable tab; float* slab; size_t slab_floats, used; double flops, bytes, bytes2; int nblocks, nblocks2, nred; bool batched;
                 hipStream_t stream; bool has_stream; };   // has_stream: the stream every flush of this batch goes to is known (wp_batch_roll)
bool wp_eligible(const WpProduct& p);
// wgrad_rows16.hip: the one-pass LDS-DMA kernel for bf16 rows at the 256-d widths
int r16_variant(int n_out, int k_in, int* tiles_o, int* tiles_c);
int launch_wgrad_rows16(const WpTable& tab, int nblocks, hipStream_t s);
// (batched: the job shares its launch with the other products of a group of steps -- fewer row chunks per job)
size_t wp_slab_floats(int n_out, int k_in, int64_t rows, int nbatch, bool ranged, bool batched, bool src16 = false);
void wp_batch_begin(WpBatch* b, float* slab, size_t slab_floats, bool batched);   // opens b for the calling thread
void wp_batch_set_stream(hipStream_t s);     // the open batch will be flushed on `s` (allows wp_batch_roll)
// the open batch is full (job table or slab space) for these eligible products: run what it holds on its stream and reopen it
// empty -- launches on one stream are serial, so the slab region is reused; false: not possible (no stream set / nothing recorded)
bool wp_batch_roll(int* status);
bool wp_batch_open();
bool wp_batch_add(const WpProduct& p);       // true: recorded (a batch is open, the product is eligible, table and slab space suffice)
bool wp_batch_add(const WpProduct* ps, int n);   // all n (the direction groups of one product) or none
int wp_batch_flush(hipStream_t stream);      // product launch + slab-sum launch; closes the batch
void wp_batch_abort();
struct WpBatchGuard { ~WpBatchGuard() { wp_batch_abort(); } };

// few rows, long K (node encoder at the reference's graph sizes): split-K into `scratch`, then a fixed-order sum (gemm.hip)
// `next` (optional): the Linear layer that follows ([next->n x n] weights, output next->y): evaluated in the summing launch when it is
// narrow enough (next->done is set); otherwise untouched
struct SplitkNext { const float* w; const float* b; int n; int relu; float* y; int64_t ldy; bool done; };
bool linear_splitk(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int64_t m, int n, int k, int relu,
                   float* scratch, size_t scratch_floats, hipStream_t stream, int* status, SplitkNext* next = nullptr);
size_t linear_splitk_scratch_floats(int64_t m, int n, int k);
// Convenience: y = act(x W^T + b)
int linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int64_t m, int n, int k,
           int relu, hipStream_t stream);

// ------------------------------------------------------------------------------------ graph
// Layout of the prepared graph buffer (all int32 arrays; offsets in bytes from the buffer start are
// a pure function of (N, E), see graph_layout()).
struct GraphView {
    int N;
    int64_t E;
    int* header;    // [8]: {error_flag, E_out, E_in, E_self, E_out (dup, as group-0 row_end), E_out+E_in, 0, 0}
    int* perm;      // [E]   sorted position -> original edge id
    int* srow;      // [E]   row of the edge at sorted position
    int* scol;      // [E]   col ...
    int* seg_ptr;   // [3N+1] CSR over keys dir*N + row (dir 0 out, 1 in, 2 self)
    int* cperm;     // [E]   positions (in sorted order) re-sorted stably by key dir*N + col
    int* cseg_ptr;  // [3N+1] CSR over keys dir*N + col, indexing cperm
    int* rperm;     // [E]   positions sorted by row only (all directions): row-gradient segments
    int* rseg_ptr;  // [N+1]
    int* cperm_all; // [E]   positions sorted by col only (all directions)
    int* cseg_all;  // [N+1]
};
size_t graph_layout(int N, int64_t E, GraphView* view, void* base);

// ------------------------------------------------------------------------------------ segments
// out[n][0:dim] = AGG over in-segment(n), out[n][dim:2dim] = AGG over out-segment(n)
// (torch.cat((flow_in, flow_out)), mpn.py:97); src rows in sorted edge order.
int aggregate(const GraphView& g, const float* src, int dim, int agg, float* out, int* argmax, hipStream_t stream);
// dn = 32: aggregate -> node update -> next step's projections (P = P0 + x' Wx^T; P = nullptr: none) in one launch; agg_out:
// the aggregated messages [N, 64] for the backward pass (training) or nullptr
int node_step32(const GraphView& g, const float* msg, int agg, const float* Wu, const float* bu, float* x_new, float* agg_out,
                const float* Wx, int64_t ldwx, const float* P0, float* P, int pw, hipStream_t stream);
// generic: out[s][:] = AGG_{j in [ptr[s], ptr[s+1])} src[list ? list[j] : j][:]; out leading dim ldo
int segment_reduce_csr(const float* src, int64_t lds, const int* list, const int* ptr, int nseg, int dim, int agg,
                       float* out, int64_t ldo, int* argmax, int accumulate, hipStream_t stream);

// total_rows (optional hint): rows summed over all segments; long average segments use a block-per-segment kernel
int segment_reduce_csr2(const float* src, int64_t lds, const int* list, const int* ptr, int nseg, int dim, float* out,
                        int64_t ldo, int nmod, int off0, int off1, hipStream_t stream, int64_t total_rows = 0, int runs = 1,
                        int run_stride = 0);
// dn = 32 backward: dX = dP Wx, dZn = dX (.) [x_prev > 0], dAGG = dZn Wu in one launch (segment.hip)
int node_step32_bwd(const float* dP, int N, int pw, const float* Wx, int64_t ldwx, const float* x_prev, const float* Wu, float* dZn,
                    float* dAGG, hipStream_t stream);
// dn = 64 / 128: aggregation -> node update -> the next step's projections in one launch, split operands (node_chain.hip)
struct NodeChainArgs {
    int N, dn, pw, agg;
    const int* seg_ptr;             // graph CSR over keys dir * N + row
    const float* M;                 // [E, dn] messages, sorted edge order
    const unsigned short* wu_img;   // packed units (pack_node_chain)
    const float* bu;                // [dn]
    const unsigned short* wx_img;
    const float* P0;                // [N, pw]
    float* P_next;                  // [N, pw] or nullptr (last step)
    float* x_new;                   // [N, dn]
    float* agg_out;                 // [N, 2 dn] or nullptr
    int debug;                      // ablation bits, read only by a build with -DMPNHIP_NODE_FWD_DEBUG (tools/node_chain_ablate.sh)
};
bool node_chain_supported(int dn, int pw, int kx);
size_t node_chain_image_shorts(int dn, int pw, size_t* off_wx);
int pack_node_chain(const float* Wu, const float* Wnode, int dn, int pw, int kx, unsigned short* img, hipStream_t s);
int launch_node_chain(const NodeChainArgs& a, hipStream_t s);
// ... and its backward mirror: dX = dP Wx, dZn = dX (.) [x_prev > 0], dAGG = dZn Wu in one launch
struct NodeChainBwdArgs {
    int N, dn, pw;
    const float* dP;                 // [N, pw] gradient of this step's projections
    const unsigned short* wxT_img;   // packed units (pack_node_chain_bwd)
    const float* x_prev;             // [N, dn] output of the previous step's node update (its ReLU mask)
    const unsigned short* wuT_img;
    float* dZn;                      // [N, dn] out: gradient at the previous step's node-update pre-activation
    float* dAGG;                     // [N, 2 dn] out
#ifdef MPNHIP_NODE_BWD_DEBUG
    int debug;                       // ablation build (make EXTRA=-DMPNHIP_NODE_BWD_DEBUG): 1 no phase-1 MFMAs, 2 no phase 3, 4 no phase 1
#endif
};
bool node_chain_bwd_supported(int dn, int pw, int kx);
size_t node_chain_bwd_image_shorts(int dn, int pw, size_t* off_wu);
int pack_node_chain_bwd(const float* Wu, const float* Wnode, int dn, int pw, int kx, unsigned short* img, hipStream_t s);
int launch_node_chain_bwd(const NodeChainBwdArgs& a, hipStream_t s);
// the whole step loop of an inference forward at the reference's widths in one launch (persist32.hip)
struct Persist32Args {
    int N, L, agg, pw, he, hn, hc, nodes_per_block;
    int64_t E;
    const int* seg_ptr; const int* srow; const int* scol; const int* perm;
    const float* e0;               // [E, 16] encoder output, sorted order
    float* e;                      // [E, 16] current edge features (written by step 1 on; the final e' on return)
    const float* P0;               // [N, pw]
    float* P[2];                   // P[0]: the first step's projections on entry
    const float* W1; int64_t ld_w1; int col_w1;           // edge layer 0 [he, in_dim], its [e0 | e] columns start at col_w1
    const float* W2;               // edge layer 1 [16, he]
    const float* Wc1;              // classifier layer 0 [hc, 16]
    const float* Wf1[2]; int64_t ld_wf1; int col_wf1;     // flow layer 0 [hn, in_dim], e' columns from col_wf1 (0: flow_out, 1: flow_in)
    const float* Wf2[2];           // flow layer 1 [32, hn]
    const float* Wu;               // node update [32, 64]
    const float* Wnode;            // packed projections [pw, 64]
    const float* b2; const float* bc1; const float* wc2; const float* bc2; const float* bf2[2]; const float* bu;
    float* logits;                 // [L, E] original edge order
    float* x_out;                  // [N, 32] the last step's node features
    unsigned* barrier;             // 4 words, zeroed by the launcher: [0] arrivals, [1] set when a wait gave up
    int debug;                     // timing experiments only (MPNHIP_PERSIST_DEBUG): 1 no barrier, 2 no edge tiles, 4 no node update
};
bool persist32_supported(int dn, int de, int he, int hn, int hc, int pw, int kx, int64_t N, int64_t E);
int launch_persist32(Persist32Args a, hipStream_t s);
// one segment_reduce_csr2 call as data; segment_reduce_csr2_x3: three of them, in one launch where the block kernel applies
struct SegReduce2 {
    const float* src; int64_t lds; const int* list; const int* ptr; int nseg; int dim; float* out; int64_t ldo; int nmod; int off0; int off1;
    int runs; int run_stride;   // (segment_reduce_csr2's `runs` form: a segment as the union of `runs` CSR runs; 0 / 1 = plain)
    unsigned short* out16; int64_t ldo16;   // (segment_reduce_csr2_x3_bf16 only) the same sums rounded to bf16 rows as well, or nullptr
};
int segment_reduce_csr2_x3(const SegReduce2 c[3], int64_t total_rows, hipStream_t stream);
// the same over bf16 source rows (c[i].src points at unsigned shorts, c[i].lds counts them): short-segment kernel, one launch
int segment_reduce_csr2_x3_bf16(const SegReduce2 c[3], hipStream_t stream);

// in-stream event timing of two designated kernels (see mpnhip_profile_enable)
enum { PROF_GEMM = 0, PROF_AGG = 1, PROF_CHAIN_BWD = 2, PROF_TN = 3, PROF_KINDS = 4 };
// `work`: algorithmic work of the bracketed launch (flops or bytes), summed for mpnhip_profile_read_kind
void prof_begin(int kind, hipStream_t s, double work = 0.0);
void prof_end(int kind, hipStream_t s);
// true (once per bracket) while a bracket is open: the events to attach to the designated kernel's dispatch
bool prof_launch_events(hipEvent_t* start, hipEvent_t* stop);
// launch `kernel` normally, or with the open bracket's events attached to the dispatch (hipExtLaunchKernelGGL)
#define MPN_LAUNCH_PROFILED(kernel, grid, block, stream, ...)                                              \
    do {                                                                                                  \
        hipEvent_t _e0, _e1;                                                                              \
        if (mpnhip::prof_launch_events(&_e0, &_e1))                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, _e0, _e1, 0, __VA_ARGS__);              \
        else                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                              \
    } while (0)

// Deferred weight packing: the ~15-25 tiny copy / pad / transpose launches that build the packed weight images of a forward (and the
// 8 of a backward) are RECORDED while a batch is open and run as ONE launch at the flush (k_pack_multi, mpn.hip) -- at the
// reference's graph sizes those launches, ~5 us apart, were 6 % of a training step.
struct PackOp {
    const float* src;   // nullptr: zeros
    float* dst;
    int64_t lds;
    int c0, rows, cols, rows_pad, cols_pad, ldd, dst_c0;
    int transposed;     // 0: dst[r * ldd + dst_c0 + c] = src[r * lds + c0 + c] (r < rows, c < cols; else 0) over rows_pad x cols_pad
                        // 1: dst[k * cols_pad + n] = src[n * lds + c0 + k] (k < rows, n < cols; else 0) over rows_pad (k) x cols_pad (n)
};
struct PackBatch {
    static constexpr int MAX = 32;
    PackOp op[MAX];
    int n;
};
void pack_batch_begin(PackBatch* b);            // opens `b` for the calling thread (n = 0)
bool pack_batch_add(const PackOp& op);          // true: recorded (a batch is open and has room)
int pack_batch_flush(hipStream_t stream);       // launches what was recorded, closes the batch
void pack_batch_abort();                        // closes an open batch without launching it
struct PackBatchGuard { ~PackBatchGuard() { pack_batch_abort(); } };   // declare beside the batch (error returns before the flush)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Host-side counters of the kernel variants launched (mpnhip_debug_counters): test instrumentation that lets a parity test
// assert WHICH code path produced the result it checked.  One relaxed atomic add per launch site.
enum PathCounter {
    PC_CHAIN_FWD = 0, PC_CHAIN_FWD_SPLIT, PC_CHAIN_BWD, PC_CHAIN_BWD_SPLIT, PC_AGGREGATE, PC_AGGREGATE_BLOCK, PC_NODE_STEP32,
    PC_NODE_STEP32_BWD, PC_SEG_SHORT, PC_SEG_BLOCK, PC_SEG_BLOCK3, PC_EDGE_ENCODER, PC_EDGE_ENCODER_BWD, PC_TN_MFMA, PC_TN_SMALL,
    PC_TN_GENERIC, PC_GEMM_FP32, PC_GEMM_SPLIT, PC_GEMM_BF16, PC_WEIGHT_PACK, PC_SEG_SHORT3, PC_GEMM_SPLITK, PC_CHAIN_FWD_BF16, PC_TN_PANEL, PC_TN_PANEL_LAUNCH, PC_NODE_CHAIN, PC_PERSIST32, PC_TN_PANEL_FALLBACK, PC_CHAIN_BWD_BF16, PC_NODE_CHAIN_BWD, PC_GEMM_BF16_TILED, PC_GEMM_BF16_RING, PC_TN_ROWS16, PC_TN_ROWS16_LAUNCH, PC_TN_PANEL_NARROW, PC_COUNT
};
void count_path(int id);

}  // namespace mpnhip
