// Fused per-edge chain kernel of one message-passing step (see edge_chain.hip).
#pragma once
#include "common.h"

namespace mpnhip {

struct EdgeChainArgs {
    int E;                 // edges (sorted order)
    int N;                 // nodes (rows of P): only checked against the kernel's 32-bit row offsets
    int split;             // 1: weight images are three-piece bf16 split images (pack_split), products from six bf16 MFMAs
    int he, de, hn, dn, hc;  // real widths; the kernel pads them to multiples of 32 (zero-padded weight images)
    const int* header;     // graph header: [1] = E_out, [2] = E_in
    const int* srow;
    const int* scol;
    const int* perm;
    const float* xa;       // first-layer input segments [e0 | e]: xa [E, k1a] (ld ldxa), xb [E, k1b] or nullptr
    const float* xb;
    int64_t ldxa, ldxb;
    int k1a, k1b;
    const float* Q0;       // [E, he] or nullptr: per-edge C-in of the first layer (the re-attached e0's share, computed once
                           // per forward: e0 does not change from step to step); then xa / w1T cover the remaining columns only
    const float* P;        // [N, pw] per-node projections: [Pr (he) | Pc (he) | Pf_out (hn) | Pf_in (hn)]
    int pw;
    // pre-transposed, zero-padded weight images WT[k][n] (capital = width rounded up to 32) and biases
    const float* w1T;      // [k1a + k1b][HE]   edge layer 0, e-part columns
    const float* w2T;      // [HE][DE]          edge layer 1
    const float* b2;       // [de]
    const float* wc1T;     // [DE][32]          classifier layer 0
    const float* bc1;      // [hc]
    const float* wc2;      // [hc]              classifier layer 1 (out dim 1)
    const float* bc2;      // [1]
    const float* wf1T_out; // [DE][HN]          flow_out layer 0, e'-part columns
    const float* wf1T_in;
    const float* wf2T_out; // [HN][DN]          flow_out layer 1
    const float* wf2T_in;
    const float* bf2_out;  // [dn]
    const float* bf2_in;
    // outputs (sorted edge order)
    float* e_new;          // [E, de]
    float* msg;            // [E, dn]
    float* logits;         // [E] in ORIGINAL order (through perm) or nullptr
    float* save_h1;        // [E, he] / nullptr  (training: activations kept for the backward pass)
    float* save_hc;        // [E, hc] / nullptr
    float* save_hf;        // [E, hn] / nullptr
    unsigned* save_mask;   // [chain_mask_ints(E, ...)] / nullptr: ReLU masks as bits, in the kernels' private lane layout
#ifdef MPNHIP_CHAIN_TS
    long long* ts;         // debug build: 16 cycle stamps per wave
#endif
};

// Backward of the same chain for one step (edge_chain.hip, edge_chain_bwd_kernel): all activation-gradient
// products of the per-edge modules, with the saved activations as ReLU masks.
struct EdgeChainBwdArgs {
    int E, N, agg, first_step, cat_two;
    int split;             // 1: the weight images are split images (pack_split)
    int skip_e0;           // 1: B6 leaves out the column pass(es) of the re-attached e0 half: their gradient is ONE product with the
                           // sum of the steps' dZ1 after the step loop (backward.hip).  Needs whole passes: 2 DE a multiple of 128.
    int he, de, hn, dn, hc;  // real widths
    const int* header;
    const int* srow;
    const int* perm;
    const int* seg_ptr;
    const float* dAGG;     // [N, 2dn] gradient of the aggregated messages [flow_in | flow_out]
    const unsigned* mask;  // ReLU masks written by the forward kernel of this step (save_mask)
    const int* ARG;        // [N, 2dn] arg max (max aggregation) or nullptr
    const float* dlog;     // [E] gradient of this step's logits, ORIGINAL edge order
    float* dE_io;          // [E, de] in: gradient w.r.t. e_s from the later step; out: dZ of the last edge layer
    float* dZM;            // [E, dn] out
    float* dZF;            // [E, hn] out
    float* dZc;            // [E, hc] out
    float* dZ1;            // [E, he] out
    float* dE0;            // [E, de] accumulated gradient of the re-attached initial edge features
    float* dEprev;         // [E, de] out: gradient w.r.t. e_{s-1} (unused at the first step)
    // zero-padded copies of the weights in their native nn.Linear orientation W[n][k] (capital = padded to 32)
    const float* wf2_out; const float* wf2_in;  // [DN][HN]
    const float* wfe_out; const float* wfe_in;  // [HN][DE]   e'-part columns of flow layer 0
    const float* wc1;      // [32][DE]
    const float* wc2;      // [hc]   (the model's tensor)
    const float* w2;       // [DE][HE]
    const float* w1e;      // one image [HE][ncol] per pass of <= 64 padded columns of [e0 | e_{s-1}]
#ifdef MPNHIP_CHAIN_TS
    long long* ts;
#endif
};
int launch_edge_chain_bwd(const EdgeChainBwdArgs& a, hipStream_t s);

// ReLU-mask words per lane (two 32-feature tiles per 32-bit word, one word range per section H1 | e' | HC | HF | M) and
// the size of one step's mask buffer: every wave tile (4 per block of 128 edges, + the launchers' 3 spare blocks)
// holds NW x 64 words
static inline int chain_mask_words(int he, int de, int hn, int dn) {
    auto w = [](int v) { return ((v + 31) / 32 + 1) / 2; };
    return w(he) + w(de) + 1 + w(hn) + w(dn);
}
static inline size_t chain_mask_ints(int64_t E, int he, int de, int hn, int dn) {
    return (size_t)((E + 127) / 128 + 3) * 4 * 64 * (size_t)chain_mask_words(he, de, hn, dn);
}
bool edge_chain_supported(int he, int de, int hn, int dn, int hc, int k1a, int k1b);
int launch_edge_chain(const EdgeChainArgs& a, hipStream_t s);
// the three-layer edge encoder at the wider models' dims in one launch (edge_chain.hip: k_edge_encoder_mfma); 1 launched / 0 not its shape / < 0 error
int launch_edge_encoder_mfma(const float* x, const int* idx, int64_t rows, int in_dim, const float* const w[3], const float* const b[3],
                             const int dims[3], float* h1_out, float* h2_out, float* y, hipStream_t s);
// Split image of the logical operand A[k][n] = src[k * sk + n * sn] (zero beyond K x N), padded to Kp x Np (multiples of
// 16 / 32): Kp Np 6 bytes at dst, in the unit order edge_chain.hip documents.  ntr_image / t0: the Np / 32 column tiles are
// tiles t0 .. of an image with ntr_image tiles per k block (default: the whole image).
int pack_split(const float* src, int64_t sk, int64_t sn, int K, int N, int Kp, int Np, float* dst, hipStream_t s,
               int ntr_image = 0, int t0 = 0);
// The split images of one forward / backward as ONE launch: pack_split() calls between split_batch_begin and split_batch_flush
// are recorded (up to 16) and run together (13 + 7 launches of ~4.5 us per cfg-B training step otherwise).
struct SplitOp { const float* src; int64_t sk, sn; int K, N, Kp, Np; unsigned short* dst; int ntr_image, t0; };
struct SplitBatch { static constexpr int MAX = 16; SplitOp op[MAX]; int n; };
void split_batch_begin(SplitBatch* b);
int split_batch_flush(hipStream_t s);
void split_batch_abort();                                              // closes an open batch without launching it
struct SplitBatchGuard { ~SplitBatchGuard() { split_batch_abort(); } }; // declare beside the batch: an error return between
                                                                       // begin and flush must not leave the thread's batch
                                                                       // pointer on a dead stack frame
int transpose_padded(const float* W, int64_t ldw, int k0, int n_rows, int k_cols, float* WT, int n_pad, int k_pad, hipStream_t s);
int pack_padded(const float* src, int64_t lds, int c0, int rows, int cols, float* dst, int rows_pad, int cols_pad, int ldd,
                int dst_c0, hipStream_t s);


// ---- bf16-operand forward chain (edge_chain_bf16.hip): N-tiled hidden layers, widths up to BASELINE.json configs[4] (256-d)
struct EdgeChainBf16Args {
    int E, N;
    int he, de, hn, dn, hc;   // real widths (multiples of 4; hc any)
    const int* header;        // graph header: [1] = E_out, [2] = E_in
    const int* srow;
    const int* scol;
    const int* perm;
    const float* xa;          // first-layer input segments [e0 | e]: xa [E, de] (ld ldxa), xb [E, de] (ld ldxb)
    const float* xb;
    int64_t ldxa, ldxb;
    const float* P;           // [N, pw] per-node projections: [Pr (he) | Pc (he) | Pf_out (hn) | Pf_in (hn)]
    int pw;
    const void* img_edge;     // pair images (pack_chain_bf16)
    const void* img_cls;
    const void* img_flow[2];  // 0: flow_out, 1: flow_in
    const float* b2;          // [de]
    const float* bc1;         // [hc]
    const float* wc2;         // [hc]
    const float* bc2;         // [1]
    const float* bf2_out;     // [dn]
    const float* bf2_in;
    float* e_new;             // [E, de]  sorted edge order
    float* msg;               // [E, dn]
    float* logits;            // [E] ORIGINAL order (through perm)
    // fused aggregation (agg_out != nullptr: msg is not written): the kernel sums / averages / maximises the messages of every
    // (direction, row) segment itself -- whole segments into agg_out [N, 2 dn] = [flow_in | flow_out] (zero-initialised by the
    // caller: empty segments), pieces of segments that cross wave tiles into `piece`, added up by k_agg_fixup
    const int* seg_ptr;       // graph CSR over keys dir * N + row
    float* agg_out;
    float* piece;             // chain_bf16_agg_scratch_floats()
    int* start_row;
    int agg;                  // MPNHIP_AGG_*
    int plain_barriers;       // 1: __syncthreads() at the weight-chunk ends instead of the counted waits (A-B switch)
    // training (all five or none): bf16 row-major copies of the hidden activations and of e_new for the weight-gradient
    // products, and the ReLU decisions of H1 | e' | HC | HF | M as bits for the backward chain kernel (edge_chain_bf16_bwd.hip)
    unsigned short* save_h1;  // [E, he] bf16
    unsigned short* save_hc;  // [E, hc]
    unsigned short* save_hf;  // [E, hn]
    unsigned short* save_eb;  // [E, de]  (= e16_out)
    unsigned* save_mask;      // chain_bf16_mask_ints(E, ...) words
    // bf16 copies of the edge features between the steps (round 4): the kernel rounds its first-layer input to bf16 anyway, so reading
    // [e0 | e] as bf16 rows gives IDENTICAL results for half the bytes; e16_out: e_new as bf16 rows for the next step (and, in
    // training, the weight-gradient products); e_new (fp32) may then be NULL (every step but the one whose features are returned)
    const unsigned short* xa16;   // [E, de] or nullptr (then xa / ldxa, fp32)
    const unsigned short* xb16;
    unsigned short* e16_out;      // [E, de] or nullptr
#ifdef MPNHIP_CHAIN_TS
    long long* ts;                // debug build: 48 cycle stamps per wave
#endif
    int debug_skip;           // MPNHIP_CHAIN_BF16_DEBUG_SKIP, timing ablations: 1 no row saves, 2 no mask words (results wrong); 4: plain
                              // instead of non-temporal row stores (A-B)
};
// mask words per lane of one 32-edge wave tile (two 32-feature tiles per word; sections H1 | e' | HC | HF | M) and the size of one
// step's mask buffer (every wave tile of the launch incl. its 3 spare blocks; `epb`-independent: 32-edge tiles)
static inline int chain_bf16_mask_words(int he, int de, int hn, int dn, int hc) {
    auto w = [](int v) { return ((v + 31) / 32 + 1) / 2; };
    return w(he) + w(de) + w(hc) + w(hn) + w(dn);
}
static inline size_t chain_bf16_mask_ints(int64_t E, int he, int de, int hn, int dn, int hc) {
    return (size_t)((E + 255) / 256 + 4) * 8 * 64 * (size_t)chain_bf16_mask_words(he, de, hn, dn, hc);
}
// bit of element r (0..15) of a tile with parity p inside its mask word
__host__ __device__ static inline int chain_bf16_mask_bit(int p, int r) { return 8 * p + (r >> 1) + 16 * (r & 1); }
// floats of the fused aggregation's scratch: piece [tiles][2][pad32(dn)] followed by start_row [tiles] (ints)
size_t chain_bf16_agg_scratch_floats(int64_t E, int dn, size_t* off_start_row);
bool edge_chain_bf16_supported(int he, int de, int hn, int dn, int hc, int ef);
// bytes of the four pair images (edge | classifier | flow_out | flow_in) and the offsets of the last three
size_t chain_bf16_image_bytes(int he, int de, int hn, int dn, int hc, int ef, size_t* off_cls, size_t* off_flow0, size_t* off_flow1);
// w_edge0: edge MLP layer 0 [he][ld_edge0], its e columns start at col0_edge (ef segments of de); w_edge1 [de][he];
// w_cls0 [hc][de]; w_flow0[q] [hn][ld_flow0], e' columns from col0_flow; w_flow1[q] [dn][hn]
int pack_chain_bf16(const float* w_edge0, int ld_edge0, int col0_edge, int ef, const float* w_edge1, const float* w_cls0,
                    const float* const w_flow0[2], int ld_flow0, int col0_flow, const float* const w_flow1[2],
                    int he, int de, int hn, int dn, int hc, void* image, hipStream_t s);
int launch_edge_chain_bf16(const EdgeChainBf16Args& a, hipStream_t s);
// dst[i] = bf16(src[i]) (round to nearest even), n a multiple of 4, 16-byte aligned src
int to_bf16_rows(const float* src, unsigned short* dst, int64_t n, hipStream_t s);
// Test instrumentation (mpnhip_debug_saved): rows of a bf16 [E, width] save in ORIGINAL edge order as floats, and the ReLU
// decisions of one mask section (0 H1, 1 e', 2 HC, 3 HF, 4 M) as 1.0 / 0.0 floats [E, width] in original order
int chain_bf16_debug_rows(const unsigned short* src, const int* perm, int64_t E, int width, float* out, hipStream_t s);
int chain_bf16_debug_mask(const unsigned* mask, int section, const int* header, const int* perm, int64_t E, int he, int de, int hn, int dn,
                          int hc, float* out, hipStream_t s);
// One pair image: nsec sections [KA first-layer units | 2 x TO second-layer units] of 1 KiB (edge_chain_bf16.hip); element
// strides make the transposed (backward) images the same kernel
int pack_pair_bf16_general(const float* Wa, int64_t sa_n, int64_t sa_k, int a_col0, int seg_real, int seg_pad, int nseg, int H,
                           const float* Wb, int64_t sb_o, int64_t sb_k, int O, int KA, int TO, int nsec, void* dst, hipStream_t s,
                           int a_natural = 0);

// ---- backward of the bf16-operand chain (edge_chain_bf16_bwd.hip) ----
// the tile counts / exactness launch_edge_chain_bf16_bwd has an instantiation for (its dispatch table, stated once: the forward's
// plan asks this before it saves in the bf16-row form, so that a width the backward cannot take trains on the unfused path)
static inline bool edge_chain_bf16_bwd_supported(int he, int de, int hn, int dn, int hc) {
    const bool exact = he % 32 == 0 && de % 32 == 0 && hn % 32 == 0 && dn % 32 == 0 && hc % 32 == 0;
    const int t1 = (he + 31) / 32, t2 = (de + 31) / 32, tf = (hn + 31) / 32, td = (dn + 31) / 32, tc = (hc + 31) / 32;
    if (t1 == 20 && t2 == 4 && tf == 14 && td == 8 && tc == 2) return exact;
    return (t1 == 10 && t2 == 2 && tf == 7 && td == 4 && tc == 1) || (t1 == 5 && t2 == 1 && tf == 4 && td == 2 && tc == 1) ||
           (t1 == 3 && t2 == 1 && tf == 2 && td == 1 && tc == 1);
}
struct EdgeChainBf16BwdArgs {
    int E, N, agg, first_step;
    int he, de, hn, dn, hc;   // real widths
    int plain_barriers;       // 1: __syncthreads() at the chunk ends (A-B switch)
    const int* header;
    const int* srow;
    const int* perm;
    const int* seg_ptr;
    const float* dAGG;        // [N, 2 dn] gradient of the aggregated messages [flow_in | flow_out]
    const unsigned* mask;     // ReLU decisions written by the forward kernel's SAVE variant of this step
    const int* ARG;           // [N, 2 dn] arg max (max aggregation) or nullptr
    const float* dlog;        // [E] gradient of this step's logits, ORIGINAL edge order
    const float* dE_in;       // [E, de] gradient w.r.t. e_s arriving from the later step
    unsigned short* dZM;      // outputs: bf16 rows [E, dn]
    unsigned short* dZF;      // [E, hn]
    unsigned short* dZc;      // [E, hc]
    unsigned short* dZ2;      // [E, de]  dZ of the edge MLP's last layer
    unsigned short* dZ1;      // [E, he]
    float* dE0;               // [E, de] running gradient of the re-attached initial edge features (first step: += here)
    float* dEprev;            // [E, de] out: gradient w.r.t. e_{s-1} (later steps)
    const void* img_edge;     // backward pair images (pack_chain_bf16_bwd)
    const void* img_cls;
    const void* img_flow[2];
    const float* wc2;         // [hc] (the model's tensor)
    int debug_skip;           // MPNHIP_CHAIN_BF16_DEBUG_SKIP, timing ablations: 1 no dZ row stores (results wrong); 4: non-temporal row stores (A-B)
};
size_t chain_bf16_bwd_image_bytes(int he, int de, int hn, int dn, int hc, size_t* off_cls, size_t* off_flow0, size_t* off_flow1);
// w_edge0 [he][ld_edge0] with the e_{s-1} columns starting at col_e; w_edge1 [de][he]; w_cls0 [hc][de]; w_flow0[q] [hn][ld_flow0], e'
// columns from col0_flow; w_flow1[q] [dn][hn]
int pack_chain_bf16_bwd(const float* w_edge0, int ld_edge0, int col_e, const float* w_edge1, const float* w_cls0,
                        const float* const w_flow0[2], int ld_flow0, int col0_flow, const float* const w_flow1[2],
                        int he, int de, int hn, int dn, int hc, void* image, hipStream_t s);
int launch_edge_chain_bf16_bwd(const EdgeChainBf16BwdArgs& a, hipStream_t s);
// edges per block / waves per block of the launch for these widths (forward and backward kernels share the mapping)
void chain_bf16_geometry(int he, int de, int hn, int dn, int hc, int* epb, int* nw);

}  // namespace mpnhip
