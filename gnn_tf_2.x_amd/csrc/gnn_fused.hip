// Host side of the fused iteration kernel (device code: gnn_fused_kernel.h): which nets it covers, the packed weight
// image, the loop-invariant label block, and the per-iteration launch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gnn_common.h"
#include "gnn_fused.h"

namespace {

constexpr int MAXL = GNN_FUSED_MAXL;
constexpr int K_GROUP = 12;     // layer-0 K-steps are consumed in pipelined groups of 3 x 4 (layer_from_lds)
constexpr int K_SLACK = 8;      // zero K-steps after the layer-0 block: the pipeline prefetches two groups past the end

struct FusedPlan {
    int layers = 0, NT = 0, NTL = 0, KP = 0, kk0 = 0, act = 0;
    // split arithmetic, state width 64 (the tuned shape): the LDS tile is laid out for 16-byte accesses - rows 16-byte aligned
    // (KPs a multiple of 4 with KPs / 4 odd: ds_read_b128 down a column stays bank-conflict free) and the aggregated-state block
    // starting on a multiple of 4 columns, i.e. after a hole of `pad` zero columns behind [state | nodes]
    int pad = 0, KPs = 0;
    int nt[MAXL] = {0, 0, 0};       // tiles of each layer's output
    int kk[MAXL] = {0, 0, 0};       // K-steps of each layer
    size_t w_off[MAXL] = {0, 0, 0}, b_off[MAXL] = {0, 0, 0}, bn_off = 0, total = 0;
    // split arithmetic (impl 2): K = 16 chunks per layer and the dword offsets of the bf16-piece images
    int chunks[MAXL] = {0, 0, 0};
    size_t s_off[MAXL] = {0, 0, 0}, s_total = 0;
};

constexpr int GNN_FUSED_VARIANT_DEFAULT = 1;      // bit 0: raised wave priority during the gather (measured: -1 %)
// start-up spread: every wave waits 0 .. n x 8k cycles before its first tile.  Round 2 / 3 (two tickets per wave drawn at kernel start, which
// already spread the waves by up to 46 us): 20 rounds, 12 for grids with one to four tiles per wave.  Round 4 (static first two rounds of
// tiles, no atomics at start; profiles/r04_midsize.txt): 8 and 6.
constexpr int GNN_FUSED_SPREAD_DEFAULT = 8;
constexpr int GNN_FUSED_SPREAD_SMALL_DEFAULT = 6;
constexpr int S_SLACK = 2;      // zero chunks after the layer-0 block of the split image (layer0_split looks two chunks ahead)

int round_tiles(int width) { return width <= 32 ? 1 : (width <= 64 ? 2 : 4); }

bool make_plan(const gnn_mlp *m, int nlc, FusedPlan &p)
{
    if (m->n_layers < 1 || m->n_layers > MAXL) return false;
    p.layers = m->n_layers;
    p.act = m->acts[0];
    int hid = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->acts[l] != p.act || m->acts[l] == GNN_ACT_SOFTMAX) return false;   // one activation for all layers
        if (m->dims[l + 1] > 128) return false;
        if (l < m->n_layers - 1) hid = std::max(hid, m->dims[l + 1]);
    }
    p.NTL = round_tiles(m->dims.back());
    if (p.layers == 1) {
        p.NT = p.NTL;
    } else {
        p.NT = std::max(round_tiles(hid), p.NTL);
        if (p.NT > 1 && p.NTL == 1) p.NTL = 2;          // instantiated pairs: (1,1) (2,2) (4,2) (4,4)
    }
    p.kk0 = ((m->dims[0] + 1) / 2 + K_GROUP - 1) / K_GROUP * K_GROUP;
    p.KP = std::max(2 * p.kk0, (m->dims[0] + 15) / 16 * 16) + 1;      // odd: conflict-free column reads
    size_t off = 0;
    for (int l = 0; l < p.layers; ++l) {
        p.nt[l] = l == p.layers - 1 ? p.NTL : p.NT;
        p.kk[l] = l == 0 ? p.kk0 : 16 * p.NT;
        p.w_off[l] = off;
        off += (size_t)(p.kk[l] + (l == 0 ? K_SLACK : 0)) * 64 * p.nt[l];
    }
    for (int l = 0; l < p.layers; ++l) { p.b_off[l] = off; off += 32 * (size_t)p.nt[l]; }
    p.bn_off = off;
    off += 2 * 32 * (size_t)p.NTL;
    p.total = off;
    const int ds = m->dims.back();
    p.pad = ds == 64 ? (4 - (ds + nlc) % 4) % 4 : 0;
    p.KPs = p.KP;
    if (ds == 64) {
        p.KPs = (std::max(2 * p.kk0, (m->dims[0] + p.pad + 15) / 16 * 16) + 3) / 4 * 4;
        if ((p.KPs / 4) % 2 == 0) p.KPs += 4;
    }
    size_t soff = 0;
    for (int l = 0; l < p.layers; ++l) {
        p.chunks[l] = l == 0 ? (m->dims[0] + p.pad + 15) / 16 : 2 * p.NT;
        p.s_off[l] = soff;
        soff += (size_t)(p.chunks[l] + (l == 0 ? S_SLACK : 0)) * p.nt[l] * 3 * 256;
    }
    p.s_total = soff;
    return true;
}

// GNN_FUSED_WAVES wave tiles [32][KP], 128 B of slack (the layer-0 pipeline reads two groups past the last tile), GNN_FUSED_WAVES x 36 row pointers
size_t lds_bytes(const FusedPlan &p)
{
    // ... and the last layer's bias / BatchNormalization scale / shift (3 x 32 NTL floats) and the hidden biases (2 x 32 NT)
    return (size_t)GNN_FUSED_WAVES * 32 * std::max(p.KP, p.KPs) * sizeof(float) + 128 + GNN_FUSED_WAVES * 36 * sizeof(int) + (3 + 2) * 32 * 4 * sizeof(float) + 16;
}

// the wave-pair form (gnn_fused_pair_kernel.h) covers the tuned shape family only: split arithmetic, state width 64, two or three layers, 128-wide
// hidden layers, a concat of nine K = 16 chunks
constexpr int PAIR_CH0 = 9;
bool pair_covers(const FusedPlan &p, int ds) { return ds == 64 && p.NTL == 2 && p.NT == 4 && (p.layers == 2 || p.layers == 3) && p.chunks[0] == PAIR_CH0; }
int pair_xs(const FusedPlan &p)       // row stride of a pair's gather tile X' (the LDS columns behind the own state), a multiple of 4 with XS / 4 odd
{
    int xs = 16 * p.chunks[0] - 64 + 4;
    if ((xs / 4) % 2 == 0) xs += 4;
    return xs;
}
// four pairs x (X'[32][XS] + P[CH0][3][64] x 16 B), 16 control words, 8 x 20 row pointers, staged vectors, slack
size_t pair_lds_bytes(const FusedPlan &p)
{
    return sizeof(float) * ((size_t)4 * (32 * pair_xs(p) + p.chunks[0] * 768) + 16 + GNN_FUSED_WAVES * 20 + 3 * 32 * 2 + 2 * 32 * 4) + 128;
}
int device_cus(int device)
{
    static int n_cu_dev[64] = {0};
    if (device < 0 || device >= 64) return 256;
    if (!n_cu_dev[device]) {
        hipDeviceProp_t prop;
        n_cu_dev[device] = hipGetDeviceProperties(&prop, device) == hipSuccess ? std::max(1, prop.multiProcessorCount) : 256;
    }
    return n_cu_dev[device];
}

}   // namespace

// The library's choice between the two forms of the default path's kernel (gnn_loop_set_tile_form(l, 0)): the wave pair while no pair of the
// launch gets a second tile (tiles <= 4 x CUs: the launch is one tile latency long and a pair's tile takes about half as long as a wave's:
// N = 4 k .. 32 k: 7 - 11 % less time per iteration), one wave per tile beyond (N = 41 k: +13 %, BASELINE size: 0.78 against 0.68 ms -
// the seven meetings of a pair per tile cost more than its shorter matrix phases return; profiles/r05_midsize_forms.txt, r05_pair_stamps.txt).
bool gnn_fused_pair_selected(const gnn_loop *l)
{
    FusedPlan p;
    if (l->impl_req != 2 || l->slice_mode || !make_plan(l->st, l->NLc, p)) return false;
    if (!pair_covers(p, l->Ds) || pair_lds_bytes(p) > 160 * 1024) return false;
    if (l->tile_form) return l->tile_form == 2;
    return (l->g->n_rows + 31) / 32 <= (int64_t)4 * device_cus(l->device);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
int gnn_fused_pack(gnn_mlp *m, int nlc)
{
    FusedPlan p;
    if (!make_plan(m, nlc, p)) { gnn_fused_release(m); return GNN_OK; }
    m->pack_nlc = nlc;
    const int lab = m->dims.back() + nlc;      // [state | nodes] columns in front of the alignment hole
    std::vector<float> img(p.total, 0.0f);
    std::vector<uint32_t> simg(p.s_total, 0u);
    std::vector<float> W, b;
    for (int l = 0; l < m->n_layers; ++l) {
        const int n_in = m->dims[l], n_out = m->dims[l + 1];
        W.resize((size_t)n_in * n_out);
        b.resize(n_out);
        HIPCHK(hipMemcpy(W.data(), m->W[l], W.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), m->b[l], b.size() * sizeof(float), hipMemcpyDeviceToHost));
        float *wp = img.data() + p.w_off[l];
        const int nt = p.nt[l];
        for (int kk = 0; kk < p.kk[l]; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int jt = 0; jt < nt; ++jt) {
                    const int k = 2 * kk + (lane >> 5), j = 32 * jt + (lane & 31);
                    wp[((size_t)kk * 64 + lane) * nt + jt] = (k < n_in && j < n_out) ? W[(size_t)k * n_out + j] : 0.0f;
                }
        for (int j = 0; j < n_out; ++j) img[p.b_off[l] + j] = b[j];
        // split image: [chunk][out tile][piece][lane][8 bf16]; element i of lane (m, h) is k(h, i) of gnn_fused_kernel.h
        // Folded SELU between the dense layers of the split path (gnn_fused_kernel.h, GNN_S1_E): a layer whose OUTPUT feeds the
        // folded activation is scaled by log2(e) (so is its bias, at staging), a layer whose INPUT comes from it by scale / log2(e).
        float fold = 1.0f;
        if (p.act == GNN_ACT_SELU && m->n_layers > 1) {
            const double LOG2E = 1.44269504088896341, SCALE = 1.0507009873554805;
            const bool in_folded = l > 0, out_folded = l < m->n_layers - 1;
            fold = (float)((in_folded ? SCALE / LOG2E : 1.0) * (out_folded ? LOG2E : 1.0));
        }
        uint32_t *sp = simg.data() + p.s_off[l];
        for (int c = 0; c < p.chunks[l]; ++c)
            for (int jt = 0; jt < nt; ++jt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int h = lane >> 5, r = 8 * (c & 1) + i;
                        int k = l == 0 ? 16 * c + 8 * h + i : 32 * (c >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (l == 0 && p.pad) k = k < lab ? k : (k < lab + p.pad ? n_in : k - p.pad);   // LDS column -> concat column (hole: zero)
                        const int j = 32 * jt + (lane & 31);
                        float v = (k < n_in && j < n_out) ? W[(size_t)k * n_out + j] : 0.0f;
                        v *= fold;
                        for (int pc = 0; pc < 3; ++pc) {          // truncation split: v == p0 + p1 + p2 exactly
                            uint32_t bits;
                            memcpy(&bits, &v, 4);
                            const uint32_t hi = bits & 0xffff0000u;
                            float piece;
                            memcpy(&piece, &hi, 4);
                            v = v - piece;
                            uint32_t &d = sp[((((size_t)c * nt + jt) * 3 + pc) * 64 + lane) * 4 + i / 2];
                            d |= (i & 1) ? hi : (hi >> 16);
                        }
                    }
    }
    if (m->has_bn) {
        const int f = m->dims.back();
        HIPCHK(hipMemcpy(img.data() + p.bn_off, m->bn_scale, f * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(img.data() + p.bn_off + 32 * p.NTL, m->bn_shift, f * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (m->packed_floats != p.total) {
        gnn_fused_release(m);
        HIPCHK(gnn_dev_malloc((void **)&m->packed, p.total * sizeof(float)));
        m->packed_floats = p.total;
    }
    HIPCHK(hipMemcpy(m->packed, img.data(), p.total * sizeof(float), hipMemcpyHostToDevice));
    if (m->packed_split_dwords != p.s_total) {
        if (m->packed_split) (void)hipFree(m->packed_split);
        m->packed_split = nullptr;
        HIPCHK(gnn_dev_malloc((void **)&m->packed_split, p.s_total * sizeof(uint32_t)));
        m->packed_split_dwords = p.s_total;
    }
    HIPCHK(hipMemcpy(m->packed_split, simg.data(), p.s_total * sizeof(uint32_t), hipMemcpyHostToDevice));
    return GNN_OK;
}

void gnn_fused_release(gnn_mlp *m)
{
    if (m->packed) (void)hipFree(m->packed);
    if (m->packed_split) (void)hipFree(m->packed_split);
    m->packed = nullptr;
    m->packed_split = nullptr;
    m->packed_floats = 0;
    m->packed_split_dwords = 0;
}

bool gnn_fused_supported(const gnn_loop *l)
{
    FusedPlan p;
    if (!make_plan(l->st, l->NLc, p)) return false;
    if (l->st->pack_dirty || l->st->pack_nlc != l->NLc) {            // weights (or the concat layout) changed since the images were built
        if (gnn_fused_pack(l->st, l->NLc) != GNN_OK) return false;
        l->st->pack_dirty = false;
    }
    if (!l->st->packed) return false;
    if (lds_bytes(p) > 160 * 1024) return false;                            // one 8-wave workgroup per CU
    const int Ds = l->Ds;
    if (!((Ds % 4 == 0 && Ds <= 256) || Ds <= 64)) return false;           // one column chunk per lane in the gather
    if ((int64_t)l->N_pad * Ds * (int64_t)sizeof(float) >= ((int64_t)1 << 31)) return false;   // 32-bit row offsets
    return l->g->n_rows > 0;
}

int gnn_fused_prepare(gnn_loop *l)
{
    const gnn_graph *g = l->g;
    const int IW = 2 * l->NLc + g->AL;
    if (!l->inv) {
        // (rows padded to whole 32-node tiles and zeroed: the full-tile kernel reads the label columns of a partial last tile unguarded)
        const size_t inv_floats = std::max<size_t>(1, (size_t)((g->n_rows + 31) / 32 * 32) * IW);
        HIPCHK(gnn_dev_malloc((void **)&l->inv, inv_floats * sizeof(float)));
        HIPCHK(hipMemsetAsync(l->inv, 0, inv_floats * sizeof(float), l->stream));
    }
    if (IW == 0) return GNN_OK;
    // [nodes | Adjacency^T . nodes | ArcNode^T . arc labels]  (GNN.py:263, :259): loop-invariant, and unchanged from run to run
    // unless gnn_graph_update_labels rewrote the labels in between (LGNN stacks)
    if (l->inv_version == g->label_version) return GNN_OK;
    l->inv_version = g->label_version;
    int rc = gnn_launch_spmm(l->stream, g->n_rows, g->sh->indptr, nullptr, g->sh->arc_w, gnn_graph_arc_labels(g), g->AL, g->AL,
                             l->inv + 2 * l->NLc, IW, nullptr, 1);
    if (rc) return rc;
    if (l->NLc) {
        rc = gnn_launch_spmm(l->stream, g->n_rows, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, g->nodes, g->NL, g->NL,
                             l->inv + l->NLc, IW, nullptr, 1);
        if (rc) return rc;
        rc = gnn_launch_copy_cols(l->stream, g->n_rows, g->NL, g->nodes + (size_t)g->own_off * g->NL, g->NL, l->inv, IW, nullptr, 1);
        if (rc) return rc;
    }
    return GNN_OK;
}

// everything of the kernel arguments that does not depend on the launch geometry; split: arithmetic mode / tile layout
static int fused_args(gnn_loop *l, int k, bool split, FusedPlan &p, GnnFusedArgs &a)
{
    const gnn_graph *g = l->g;
    const gnn_mlp *m = l->st;
    if (!make_plan(m, l->NLc, p)) return gnn_fail(GNN_ERR_UNSUPPORTED, "fused path does not cover this net_state");
    if (m->pack_nlc != l->NLc) return gnn_fail(GNN_ERR_STATE, "weight image laid out for another label width");
    const int cur = k & 1, nxt = cur ^ 1, P = l->world;
    a = GnnFusedArgs{};
    a.n_rows = g->n_rows; a.row_begin = l->own_off;     // replica row of the first owned row
    a.indptr = g->sh->indptr; a.adj_src = g->sh->adj_src; a.adj_w = g->sh->adj_w;
    a.inv = l->inv;
    a.state_cur = l->state[cur];
    a.state_bytes = (int64_t)l->N_pad * l->Ds * (int64_t)sizeof(float);
    a.state_nxt = l->state[nxt] + (size_t)l->own_off * l->Ds;
    const int pad = split ? p.pad : 0;
    a.Ds = l->Ds; a.NLc = l->NLc; a.AL = g->AL; a.IW = 2 * l->NLc + g->AL; a.in_s = l->in_s + pad; a.c_aggs = l->Ds + l->NLc + pad;
    a.KP = split ? p.KPs : p.KP; a.kk0 = p.kk0;
    a.vec = (l->Ds % 4 == 0) ? 4 : 1;
    int lpr = 1, lg = 0;
    while ((lpr * a.vec < l->Ds || lpr < 2) && lpr < 64) { lpr <<= 1; ++lg; }
    a.lpr = lpr; a.lpr_log2 = lg;
    for (int i = 0; i < p.layers; ++i) {
        a.Wp[i] = m->packed + p.w_off[i];
        a.bias[i] = m->packed + p.b_off[i];
    }
    for (int i = 0; i < p.layers; ++i) a.Ws[i] = m->packed_split + p.s_off[i];
    a.chunks0 = p.chunks[0];
    a.Ws_base = m->packed_split;
    a.ws_bytes = (int)(p.s_total * sizeof(uint32_t));
    for (int i = 0; i < p.layers; ++i) a.ws_off[i] = (int)(p.s_off[i] * sizeof(uint32_t));
    a.variant = GNN_FUSED_VARIANT_DEFAULT;
    a.bn_scale = m->has_bn ? m->packed + p.bn_off : nullptr;
    a.bn_shift = m->has_bn ? m->packed + p.bn_off + 32 * p.NTL : nullptr;
    a.thr = l->thr;
    a.gate = l->flags + (size_t)k * P * GNN_FLAG_WORDS;
    a.flag_out = l->flags + ((size_t)(k + 1) * P + l->rank) * GNN_FLAG_WORDS;
    a.world = P;
    a.certify = split ? 1 : 0;
    a.stamps = nullptr;
    a.agg_in = l->slice_mode ? l->agg_own : nullptr;
    a.threads = 0;
    a.wstride = 1;
    a.tile_base = 0;
    a.full_tiles = 0;
    return GNN_OK;
}

int gnn_fused_iteration(gnn_loop *l, int k)
{
    const gnn_graph *g = l->g;
    FusedPlan p;
    GnnFusedArgs a;
    const bool split = l->impl_req == 2;
    int rc = fused_args(l, k, split, p, a);
    if (rc) return rc;
    const size_t n_tiles = (size_t)((g->n_rows + 31) / 32);
#ifdef GNN_DIAG
    static const int variant_env = getenv("GNN_FUSED_VARIANT") ? atoi(getenv("GNN_FUSED_VARIANT")) : GNN_FUSED_VARIANT_DEFAULT;
    a.variant = variant_env;
#endif
#ifdef GNN_DIAG   // diagnostic build only (make DIAG=1): timing experiments and per-wave phase stamps; never in the shipped library
    static const int debug = getenv("GNN_FUSED_DEBUG") ? atoi(getenv("GNN_FUSED_DEBUG")) : 0;
    a.wstride = (debug & 1) ? 0 : 1;                                  // 0: every K-step re-reads step 0 (results meaningless)
    static const char *stamp_file = getenv("GNN_FUSED_STAMPS");       // dump per-wave phase stamps of body 1
    static unsigned long long *stamp_buf = nullptr;
    const size_t n_waves = n_tiles;
    if (stamp_file && k == 1) {
        if (!stamp_buf) HIPCHK(gnn_dev_malloc((void **)&stamp_buf, n_waves * 16 * sizeof(unsigned long long)));      // (8 slots per tile: k_fused; 16: k_fused_pair)
        HIPCHK(hipMemsetAsync(stamp_buf, 0, n_waves * 16 * sizeof(unsigned long long), l->stream));
        a.stamps = stamp_buf;
    }
#endif
    if (l->device < 0 || l->device >= 64) return gnn_fail(GNN_ERR_ARG, "device %d out of range", l->device);
    const int n_cu = device_cus(l->device);
    // one workgroup per CU; small graphs spread their tiles over as many CUs as they have tiles (a tile alone on a CU runs
    // faster than eight sharing its L1 / LDS / SIMDs; the waves without a tile leave at once)
    const unsigned grid = (unsigned)std::min<size_t>((size_t)n_cu, n_tiles);
    a.tile_ctr = l->tile_ctr + k;
    int stagger_rounds = GNN_FUSED_SPREAD_DEFAULT;
#ifdef GNN_DIAG
    static const int stagger_env = getenv("GNN_FUSED_STAGGER") ? atoi(getenv("GNN_FUSED_STAGGER")) : GNN_FUSED_SPREAD_DEFAULT;   // tuning experiments
    stagger_rounds = stagger_env;
#endif
    // start-up spread (k_fused): the full amount when every wave has four or more tiles; the small one between one and four tiles per wave
    // (tools/bench_midsize.py, profiles/r04_midsize.txt); none when no wave has a second tile
    a.stagger = n_tiles >= (size_t)4 * GNN_FUSED_WAVES * grid ? stagger_rounds : (n_tiles > (size_t)GNN_FUSED_WAVES * grid ? GNN_FUSED_SPREAD_SMALL_DEFAULT : 0);
#ifdef GNN_DIAG      // experiment: small grids (1 - 4 tiles per wave) with the two-cluster offset (variant bit 2): waves 4-7 start GNN_FUSED_STAGGER_SMALL x 8k cycles late
    static const int stagger_small = getenv("GNN_FUSED_STAGGER_SMALL") ? atoi(getenv("GNN_FUSED_STAGGER_SMALL")) : 0;
    const bool small_grid = n_tiles > (size_t)GNN_FUSED_WAVES * grid && n_tiles < (size_t)4 * GNN_FUSED_WAVES * grid;
    if (small_grid && stagger_small > 0) { a.stagger = stagger_small; a.variant |= 4; }
    static const int spread_small = getenv("GNN_FUSED_SPREAD_SMALL") ? atoi(getenv("GNN_FUSED_SPREAD_SMALL")) : -1;      // ... or the hashed spread 0 .. n rounds
    if (small_grid && spread_small >= 0) a.stagger = spread_small;
#endif
    // fewer tiles than waves: a wave that drew two tickets at start would run two tiles one after the other while another wave of the
    // launch gets none (N = 31k: 488 of 2,048 waves did all the work) - the look-ahead ticket is only drawn when every wave has a tile
    a.single_ticket = n_tiles <= (size_t)GNN_FUSED_WAVES * grid ? 1 : 0;
    const size_t lds = lds_bytes(p);
    a.lds_floats = gnn_poison_enabled() ? (int)(lds / sizeof(float)) : 0;
    auto go = [&](const GnnFusedArgs &aa, unsigned gr) -> bool {
        if (split) {
            if (p.layers == 1) return gnn_fused_launch_s1(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
            if (p.layers == 2) return gnn_fused_launch_s2(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
            return gnn_fused_launch_s3(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
        }
        if (p.layers == 1) return gnn_fused_launch_l1(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
        if (p.layers == 2) return gnn_fused_launch_l2(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
        return gnn_fused_launch_l3(p.act, p.NT, p.NTL, aa, gr, lds, l->stream);
    };
    bool ok = false;
    const int64_t n_tiles64 = (g->n_rows + 31) / 32;
    bool pair = false;
    if (split && gnn_fused_pair_selected(l)) {
        // wave-pair form: four pairs per workgroup, one workgroup per CU; the tile counters, gates and flags are k_fused's
        GnnFusedArgs ap = a;
        ap.KP = pair_xs(p);
        ap.full_tiles = 1; ap.tile_base = 0;
        const unsigned grid_p = (unsigned)std::min<int64_t>((int64_t)n_cu, (n_tiles64 + 3) / 4);
        ap.stagger = n_tiles64 >= (int64_t)4 * 4 * grid_p ? stagger_rounds : (n_tiles64 > (int64_t)4 * grid_p ? GNN_FUSED_SPREAD_SMALL_DEFAULT : 0);
        const size_t lds_p = pair_lds_bytes(p);
        ap.lds_floats = gnn_poison_enabled() ? (int)(lds_p / sizeof(float)) : 0;
        pair = p.layers == 2 ? gnn_fused_launch_p2(p.act, ap, grid_p, lds_p, l->stream) : gnn_fused_launch_p3(p.act, ap, grid_p, lds_p, l->stream);
    }
    if (pair) ok = true;
    else if (l->Ds == 64 && p.NTL == 2 && n_tiles64 >= 1) {
        // the full-tile specialisation (no generic paths compiled in) on every tile; a partial last tile takes a wave-uniform
        // branch with masked row stores / condition votes (the row buffers are padded to whole tiles, rows past n_rows have no arcs)
        GnnFusedArgs af = a;
        af.full_tiles = 1; af.tile_base = 0;
        ok = go(af, (unsigned)std::min<size_t>((size_t)n_cu, (size_t)n_tiles64));
    } else
        ok = go(a, grid);
    if (!ok) return gnn_fail(GNN_ERR_UNSUPPORTED, "no fused instantiation for %d layers, tiles (%d,%d), activation %d", p.layers, p.NT, p.NTL, p.act);
#ifdef GNN_DIAG
    if (a.stamps) {
        std::vector<unsigned long long> host(n_waves * (pair ? 16 : 8));
        HIPCHK(hipStreamSynchronize(l->stream));
        HIPCHK(hipMemcpy(host.data(), stamp_buf, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_file, "wb")) { fwrite(host.data(), sizeof(unsigned long long), host.size(), f); fclose(f); }
    }
#endif
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// persistent small-graph loop (device code: gnn_small.hip)
// ---------------------------------------------------------------------------------------------------------------------
bool gnn_small_supported(const gnn_loop *l)
{
    if (l->world != 1 || l->impl_req < 1 || l->small_disabled || l->profiling) return false;
    if (!gnn_fused_supported(l)) return false;
    FusedPlan p;
    if (!make_plan(l->st, l->NLc, p)) return false;
    if (p.NT != 1 || p.NTL != 1) return false;                      // nets no wider than 32 (one 32-feature tile per layer)
    const int64_t n_tiles = (l->g->n_rows + 31) / 32;
    if (p.kk0 > 48) return false;                                   // layer-0 weights are kept in registers
    return n_tiles >= 1 && n_tiles <= 256;                          // every tile resident at once (one wave each), with a wide margin
}

int gnn_small_run(gnn_loop *l, bool *output_done)
{
    FusedPlan p;
    GnnFusedArgs a;
    int rc = fused_args(l, 0, false, p, a);                          // exact f32-MFMA arithmetic, unpadded tile layout
    if (rc) return rc;
    const gnn_graph *g = l->g;
    a.gate = nullptr; a.flag_out = nullptr; a.tile_ctr = nullptr; a.stagger = 0;
    GnnSmallCtl c{};
    c.state0 = l->state[0]; c.state1 = l->state[1];
    c.init = l->D ? l->state_init : g->nodes + (size_t)g->own_off * g->NL;      // D == 0: NL == Ds (GNN.py:265)
    c.kfinal = l->kfinal_dev;
    c.host_result = l->kfinal_host;                                   // pinned, device-visible: no copy back
    // 16-node tiles (gnn_small16.hip) while twice the workgroups are still resident at once: a body is a chain of latencies, and a
    // 16-node tile's dense layers and activations are half as long
    bool tile16 = g->n_rows <= 16 * 256 && a.in_s <= 96;
#ifdef GNN_DIAG
    static const int tile_env = getenv("GNN_SMALL_TILE") ? atoi(getenv("GNN_SMALL_TILE")) : 0;
    if (tile_env == 32) tile16 = false;
#endif
    const int rows_per_tile = tile16 ? 16 : 32;
    const unsigned grid = (unsigned)((g->n_rows + rows_per_tile - 1) / rows_per_tile);
    c.DP = l->Ds <= 16 ? 16 : 32;
    {   // padded exchange rows: allocated with the first persistent run of the loop, never initialised (every row is written before it is read)
        const size_t need = (size_t)2 * grid * rows_per_tile * c.DP;
        if (l->small_xs_floats < need) {
            if (l->small_xs) (void)hipFree(l->small_xs);
            l->small_xs = nullptr; l->small_xs_floats = 0;
            HIPCHK(gnn_dev_malloc((void **)&l->small_xs, need * sizeof(float)));
            l->small_xs_floats = need;
        }
        c.xs = l->small_xs;
    }
    c.max_iter = l->max_iter;
    c.ecache = 1024;                                                  // = GNN_SMALL_ECACHE (gnn_small.hip)
#ifdef GNN_DIAG
    static const int ecache_env = getenv("GNN_SMALL_ECACHE") ? atoi(getenv("GNN_SMALL_ECACHE")) : 1024;
    c.ecache = std::min(1024, ecache_env);
#endif
    // Gate words: one per body, double-buffered by run parity at the start of the flag block.  This launch polls its own half
    // and zeroes the other half for the next run, so a run costs no memset; both halves are cleared by the host only after
    // something else (a per-body run) has used the block.
    const size_t n_words = ((size_t)l->max_iter + 3 + 3) & ~(size_t)3;      // gate of every body, + 1, + the barrier in front of the folded graph readout
    l->kfinal_host[1] = 0;                                            // status: cleared HERE, only ever set by the kernel (sticky)
    if (!l->small_words_clean) {
        HIPCHK(hipMemsetAsync(l->flags, 0, sizeof(int) * 2 * n_words, l->stream));
        l->small_words_clean = true;
        l->small_runs = 0;
    }
    c.flags = l->flags + (l->small_runs & 1) * n_words;
    c.zero_words = l->flags + ((l->small_runs & 1) ^ 1) * n_words;
    c.n_words = (int)n_words;
    ++l->small_runs;
    // output stage inside the launch when it is the usual one-layer head (same condition as k_out1)
    *output_done = false;
    const gnn_mlp *ou = l->ou;
    if (!l->edge_mode && g->n_masked && ou->n_layers == 1 && l->T <= 8 && l->Ds + l->NLc <= 64 && l->Ds <= 32 && g->NL <= 32) {
        c.out = l->out; c.mask = g->sh->mask; c.mask_pos = g->sh->masked_rows + g->n_masked;
        c.nodes_own = g->nodes + (size_t)g->own_off * g->NL;
        c.ow = ou->W[0]; c.ob = ou->b[0];
        c.obn_scale = ou->has_bn ? ou->bn_scale : nullptr; c.obn_shift = ou->has_bn ? ou->bn_shift : nullptr;
        c.NL = g->NL; c.NLc = l->NLc; c.T = l->T; c.oact = ou->acts[0];
        *output_done = true;
        // graph readout in the same launch when a NodeGraph is already cached with the loop (gnn_loop_readout uploaded it after an earlier run)
        l->ng_inlaunch = false;
        if (l->ng_ip && l->ng_G > 0 && g->n_masked == g->n_rows) {
            if (l->ng_host_floats < l->ng_G * l->T) {
                if (l->ng_host) (void)hipHostFree(l->ng_host);
                l->ng_host = nullptr; l->ng_host_floats = 0;
                if (hipHostMalloc((void **)&l->ng_host, sizeof(float) * (size_t)l->ng_G * l->T) == hipSuccess) l->ng_host_floats = l->ng_G * l->T;
            }
            if (l->ng_host) {
                c.ng_ip = l->ng_ip; c.ng_node = l->ng_node; c.ng_w = l->ng_w; c.ng_host = l->ng_host; c.G = l->ng_G; c.ro_word = l->max_iter + 1;
                l->ng_inlaunch = true;                            // (cleared again by run_loops if the launch gives up)
                l->ng_inlaunch_run = l->out_runs;                 // ... and valid for THIS run's outputs only
            }
        }
    }
    const size_t lds = sizeof(float) * ((size_t)32 * p.KP + 32 + 36 + 160 + 544 + 2048 + 2 * 1024 + 4);    // tile, row pointers, epilogue vectors, head, scratch, arc cache (GNN_SMALL_ECACHE)
    c.rnd = g->sh->max_degree > 8 ? 8 : 4;                       // entries per gather round
    // K-steps of layer 0 the kernel keeps in registers: the smallest instantiated count that covers the concat width (the packed
    // image has p.kk0 >= that many; the steps dropped are zero rows of the image)
    int kk_small = p.kk0;
    for (int cand : {8, 12, 16, 24, 32, 36, 40, 48})
        if (2 * cand >= a.in_s && cand <= p.kk0) { kk_small = cand; break; }
#ifdef GNN_DIAG
    static const char *small_stamp_file = getenv("GNN_SMALL_STAMPS");
    static unsigned long long *small_stamp_buf = nullptr;
    if (small_stamp_file) {
        if (!small_stamp_buf) HIPCHK(gnn_dev_malloc((void **)&small_stamp_buf, 256 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(small_stamp_buf, 0, 256 * sizeof(unsigned long long), l->stream));
        a.stamps = small_stamp_buf;
    }
#endif
    bool launched;
    if (tile16) {
        const gnn_mlp *m = l->st;
        for (int q = 0; q < p.layers; ++q) { c.Wraw[q] = m->W[q]; c.din[q] = m->dims[q]; c.dout[q] = m->dims[q + 1]; }
        int s0 = (a.in_s + 3) / 4;
        s0 = (s0 + 3) / 4 * 4;                                       // instantiated: 4, 8, ..., 24 K-steps of 4
        c.KP16 = std::max((a.in_s + 3) / 4 * 4, 4 * s0);
        if (c.KP16 % 8 == 0) c.KP16 += 4;                            // rows 16 bytes apart in the banks: the B-operand column reads do not conflict
        a.lds_floats = gnn_poison_enabled() ? (int)(gnn_small16_lds_bytes(c.KP16) / sizeof(float)) : 0;
        launched = gnn_small16_launch(p.layers, p.act, s0, a, c, grid, gnn_small16_lds_bytes(c.KP16), l->stream);
    } else {
        a.lds_floats = gnn_poison_enabled() ? (int)(lds / sizeof(float)) : 0;
        launched = gnn_small_launch(p.layers, p.act, kk_small, a, c, grid, lds, l->stream);
    }
    if (!launched)
        return gnn_fail(GNN_ERR_UNSUPPORTED, "no persistent-loop instantiation for %d layers, activation %d", p.layers, p.act);
    HIPCHK(hipGetLastError());
#ifdef GNN_DIAG
    if (small_stamp_file) {
        unsigned long long host[256];
        HIPCHK(hipStreamSynchronize(l->stream));
        HIPCHK(hipMemcpy(host, small_stamp_buf, sizeof(host), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(small_stamp_file, "wb")) { fwrite(host, sizeof(unsigned long long), 256, f); fclose(f); }
    }
#endif
    return GNN_OK;      // k and the status word are written straight into the pinned host words
}
