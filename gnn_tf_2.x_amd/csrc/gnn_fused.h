// Host/device interface of the fused iteration kernel (gnn_fused_kernel.h).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int GNN_FUSED_MAXL = 3;
constexpr int GNN_FUSED_WAVES = 8;                 // waves per (persistent) workgroup; w and w + 4 share a SIMD
constexpr int GNN_FUSED_THREADS = 64 * GNN_FUSED_WAVES;

struct GnnFusedArgs {
    // graph
    int64_t n_rows, row_begin;
    const int32_t *indptr, *adj_src;
    const float *adj_w;
    const float *inv;        // [n_rows, IW] = [nodes | aggregated nodes | aggregated arcs] (label columns of the concat)
    // state
    const float *state_cur;  // [N_pad, Ds] all nodes
    float *state_nxt;        // owned rows
    int64_t state_bytes;     // size of the replica state_cur points to
    // shapes
    // in_s: columns of the LDS tile in use (concat width + alignment hole); c_aggs: first column of the aggregated-state block
    int Ds, NLc, AL, IW, in_s, c_aggs, KP, lpr, lpr_log2, vec, kk0;
    // layers: packed weights [kk][lane][tiles of the layer], biases padded to whole tiles
    const float *Wp[GNN_FUSED_MAXL];
    const float *bias[GNN_FUSED_MAXL];
    const float *bn_scale, *bn_shift;   // padded to whole tiles, or nullptr
    // control
    float thr;
    const int *gate;
    int *flag_out;
    int world;
    int certify;             // 1 (split arithmetic): also raise the "robust" / "borderline" words of the certified gate (gnn_flag_raise_certified)
    int *tile_ctr;           // device-wide tile counter of this iteration (zeroed at the start of gnn_loop_run)
    int wstride;             // 1 normally; 0 (GNN_FUSED_DEBUG=1, timing experiments only) makes every K-step re-read step 0
    int stagger;             // s_sleep(127) rounds the second half of the waves waits before its first tile
    // split arithmetic (impl 2): per layer the bf16-piece weight image [chunk][out tile][piece][lane][8 bf16], and the number
    // of K = 16 chunks of layer 0
    const int *Ws[GNN_FUSED_MAXL];
    int chunks0;
    // the same image through one buffer descriptor: base pointer, size and the byte offset of every layer, so that the weight
    // loads of the unrolled layers are buffer_load(rsrc, lane * 16, scalar offset) without any per-load vector address arithmetic
    const int *Ws_base;
    int ws_bytes, ws_off[GNN_FUSED_MAXL];
    int tile_base;           // first tile of this launch (tickets count from it)
    int full_tiles;          // 1: state width 64 - launch the full-tile specialisation of the kernel (a partial last tile takes its masked branch)
    int variant;             // tuning switches (bit 0: raised wave priority during the gather); fixed in the shipped build
    // feature-sliced exchange: aggregated states of the owned rows [n_rows, Ds], computed outside the kernel (no gather), else nullptr
    const float *agg_in;
    int threads;             // threads per workgroup of the launch (0: GNN_FUSED_THREADS)
    int single_ticket;       // 1: the launch has no more tiles than waves - a wave draws ONE ticket at start (no look-ahead tile)
    // diagnostics only (GNN_FUSED_STAMPS=<file>): s_memtime stamps per wave at the phase boundaries, else nullptr
    unsigned long long *stamps;
    // diagnostics only (GNN_POISON=1, diagnostic build): floats of the launch's dynamic LDS allocation that every workgroup fills with NaN
    // before its first tile (a read of a never-written LDS word then shows as a NaN instead of a stale value), else 0
    int lds_floats;
};

// control block of the persistent small-graph loop (gnn_small.hip)
struct GnnSmallCtl {
    float *state0, *state1;  // the two state replicas (ping-pong), all rows
    const float *init;       // initial state of the owned rows [n_rows, Ds] (injected / drawn state, or the node labels for D == 0)
    int *kfinal;             // receives the number of executed bodies
    int *flags;              // word [b]: barrier + gate of body b (low half arrivals, high half movers), zeroed before the launch
    int *host_result;        // pinned host memory (zero-copy): [k, status]; status: set to 1 by a workgroup whose barrier spin gave up, never cleared by the
                             // kernel (sticky), zeroed by the host before the launch
    float *xs;               // padded exchange rows [2][tiles * 32][DP] (gnn_small.hip, small_gather_padded): the state between bodies
    int DP;                  // 16 (Ds <= 16) or 32 floats per exchange row
    int rnd;                 // arcs per gather round for DP == 16 (4, or 8 when some row has more than 8 arcs)
    // 16-node-tile form (gnn_small16.hip): the Keras-layout kernels W[din][dout] (its A operands are read from them directly) and the
    // row stride of its LDS tile
    const float *Wraw[GNN_FUSED_MAXL];
    int din[GNN_FUSED_MAXL], dout[GNN_FUSED_MAXL];
    int KP16;
    int *zero_words;         // the OTHER run's gate words (double-buffered by run parity): zeroed here for the next run
    int n_words;
    int max_iter;
    // output stage folded into the launch (apply_filters + a one-layer net_output, GNN.py:275-279; as k_out1), or out == nullptr
    float *out;              // [n_masked, T]
    const uint8_t *mask;     // [n_rows]
    const int32_t *mask_pos; // [n_rows] position of a masked row among the masked rows
    const float *nodes_own;  // node labels of the owned rows [n_rows, NL]
    const float *ow, *ob, *obn_scale, *obn_shift;   // net_output: W [wf, T], b [T], BatchNormalization scale / shift or nullptr
    int NL, NLc, T, oact;
    // graph readout folded into the launch (NodeGraph^T . out, GNN.py:331-332; as k_readout), or ng_ip == nullptr: CSR over graphs of
    // (node, weight), result [G, T] written to pinned host memory by workgroup 0 after one more grid barrier (word ro_word of `flags`)
    const int32_t *ng_ip, *ng_node;
    const float *ng_w;
    float *ng_host;
    int G, ro_word;
    int ecache;              // arcs of a tile whose ids / weights may be kept in LDS (GNN_SMALL_ECACHE; 0: none)
};
bool gnn_small_launch(int layers, int act, int kk0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes,
                      hipStream_t st);

bool gnn_small16_launch(int layers, int act, int s0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st);
size_t gnn_small16_lds_bytes(int kp16);

// one per translation unit gnn_fused_l{1,2,3}.hip; false = no instantiation for (act, nt, ntl)
bool gnn_fused_launch_l1(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
// split-arithmetic instantiations: gnn_fused_s{1,2,3}.hip
bool gnn_fused_launch_s1(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
bool gnn_fused_launch_s2(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
bool gnn_fused_launch_s3(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
// wave-pair form (gnn_fused_pair_kernel.h: split arithmetic, state width 64, 128-wide hidden layers, 9 layer-0 chunks): gnn_fused_p{2,3}.hip
bool gnn_fused_launch_p2(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
bool gnn_fused_launch_p3(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
bool gnn_fused_launch_l2(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
bool gnn_fused_launch_l3(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st);
