// xeve_amd/csrc/tree.hip -- the mode decision of a batch of I-picture CTUs on the device: mode_analyze_lcu -> mode_coding_tree
// (src_base/xeve_mode.c:2007-2375, 2518-2610 = ctx->fn_mode_analyze_lcu in an I slice).  Baseline quad-tree, no delta QP, rdo_dbk_switch 0.
//
// A CHAIN is one CTU of one picture.  Inside a chain everything is serial: a CU's predictors are its neighbours' reconstruction, its bit counts start from
// the coder state its predecessor's winning mode left.  So the width of the launch is the number of chains (pictures), and the chains advance in LOCKSTEP
// through the full quad-tree in the reference's visiting order; what is data dependent (a node outside the picture, the early-termination rule of
// I pictures) becomes a per-chain flag, and a chain whose node is off runs the node's batched intra analysis on a harmless position and discards the result.
//
// Per tree node (size 2^(L+2)) the schedule is
//     ENTER(L)        the node's entry: coder state from the parent / the previous sibling, split_cu_flag = 0 priced, clear_map_scu (:1129), the intra job
//     [xeve_hip_pintra_analyze_cu_jobs at this size over all chains]                                         (intra.hip; none above max_cu_intra)
//     LEAF(L)         copy_to_cu_data (:868) of the intra CU, mode_cpy_rec_to_ref (:797), the early termination (:2174-2187), split_cu_flag = 1 priced
//     4 x { the quadrant's subtree; CHILD_DONE(L): its cost added, copy_cu_data (:430) into the parent, update_map_scu (:1036) }
//     EXIT(L)         the cheaper alternative kept (a split must win by more than 0.0001), picture + split mode + coder state of the winner
// and consecutive tree operations between two analyses are ONE launch (k_tree_ops, one workgroup per chain).
// P / B slices (mode_coding_unit, :1310-1350): between ENTER and LEAF the node runs [candidates from the maps, xeve_hip_inter_candidates] [the whole inter
// analysis, xeve_hip_pinter_analyze_cu_jobs] [SATD of the inter winner's luma prediction] MID(L) [the intra analysis, cut against that SATD]; LEAF keeps the intra
// result only where the inter winner has a residual and the intra cost is smaller (mode_check_intra, :1226-1308); a skipped CU at depth >= ecu_depth is not split.
// All costs are doubles built with the reference's operations in the reference's order (-ffp-contract=off), compared as the reference compares them.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "xh_common.h"
#include "cu_lane.h"

#define MAX_COST 1.7e+308
typedef xeve_hip_ctu_data CtuData;
typedef xeve_hip_sbac     SbacState;

struct Node { // one per (level, chain)
    int    active, x0, y0, leaf, do_split, best_split, dist_cu, cu_mode, try_intra, next_split; // next_split: a side node's early-termination verdict (op_leaf_side -> op_exit)
    double cost_best, cost_temp, unit_cost, cost_split;                                         // cost_split: a side node's split alternative (cost_temp is then the unsplit one's alone)
};

#define XT_STREAMS 5 // the caller's stream + one side stream per level 1 .. 4 (8x8 .. 64x64 nodes)
struct TreeK {
    int    nchains, log2_ctu, pic_w, pic_h, w_scu, h_scu, max_cu, min_cu, min_cuwh, idc, ws, hs, slice_qp, slice_num, s_mod_l, s_mod_c;
    long   mod_pic_l, mod_pic_c, map_pic;
    double lambda0;
    // per call (device)
    pel                        *mod[3];
    uint32_t                   *map_scu, *map_cu_mode;
    int8_t                     *map_ipm;
    const SbacState            *states;
    const xeve_hip_ctu_job     *jobs;
    CtuData                    *out;
    SbacState                  *out_next;
    double                     *out_cost;
    // workspace (device)
    Node                       *node;                          // [5][nchains]
    SbacState                  *curr, *next, *before, *tdepth; // [5][nchains] each: core->s_curr_best / s_next_best [L][L], s_temp_prev_comp_best, s_temp_depth
    CtuData                    *best, *temp;                   // [5][nchains]: core->cu_data_best / cu_data_temp [L][L]
    // P / B slices
    int                         inter, ecu_depth, s_org_l, vh; // vh: P / B chains of several pictures -- the batch as one tall picture for the inter analysis (xh_common.h)
    int16_t                   (*map_mv)[2][2];
    int8_t                    (*map_refi)[2];
    // THE SIDE STREAM (round 6).  The analysis of a node that has children needs nothing its children produce and they need nothing of it -- both start from the node's
    // entry state and from neighbours outside the node (xeve_mode.c:2061-2262: s_curr_before_split; the maps inside the node are cleared for either) -- until op_exit
    // compares the two costs.  A call with a side stream runs the analyses of every such node there while the main stream walks on into the children: two launch
    // chains side by side instead of one (setting 2: every level of such nodes on a side stream of its own, up to four -- measured slower: the device's kernels then
    // contend four ways).  Every array an analysis reads or writes exists once per stream (ac[0] main); side_of[L] says which a level uses.
    struct Ac {
        xeve_hip_intra_job          *ijobs; // [5][nchains]: the job arrays per LEVEL -- the main stream enters the next side node while the side stream may still be reading an outer one's
        const xeve_hip_intra_result *ires;  // [nchains]
        const int16_t               *icoef; // dense blocks of the node's analysis: Y of all chains, then U, then V
        const pel                   *irec;
        SbacState                   *sbest; // [nchains]: core->s_temp_best of the node's intra analysis
        xeve_hip_inter_job          *ejobs; // [5][nchains]
        xeve_hip_job                *sjobs; // [5][nchains]: SATD(original, inter winner's luma prediction)
        const xeve_hip_inter_result *eres;
        const int16_t               *ecoef;   // Y of all chains, then U, then V
        const pel                   *erec[3]; // [nchains][block] per component
        const int32_t               *esatd;
        const SbacState             *enext;   // [nchains]: core->s_next_best of the inter analysis
    } ac[XT_STREAMS];
    unsigned char side_of[8]; // per level L: 0 = the node's analyses run in line, s > 0 = on side stream s (the node has a CU of its size AND children): a stream per level
    SbacState    *csplit;     // [5][nchains]: the state the FIRST child of a node starts from (the node's entry state + split_cu_flag = 1)
    CtuData      *tsplit;     // [5][nchains]: a side node's cu_data_temp of the split alternative (temp stays the staging block of its own analysis)
};

enum { OP_ENTER = 0, OP_LEAF = 1, OP_CHILD_DONE = 2, OP_EXIT = 3, OP_ROOT_DONE = 4, OP_MID = 5, OP_SPLIT_PREP = 6, OP_LEAF_SIDE = 7 };
#define MAX_OPS 12
struct OpList {
    int           n;
    unsigned char op[MAX_OPS], lvl[MAX_OPS];
    signed char   part[MAX_OPS];
};

// ---- one context-coded bin with the bit counter's bookkeeping: SBAC_LOAD + xeve_sbac_bit_reset (xeve_mode.c:39-49) + xeve_sbac_encode_bin (xeve_eco.c:521-575, the
// byte output only advancing counters, :392-453) + xeve_get_bit_number (xeve_mode.c:51-55) ----------------------------------------------------------------------
__device__ static void bc_byte(SbacState &s, unsigned b)
{
    if(s.is_pending_byte) {
        if(s.pending_byte == 0) s.stacked_zero++;
        else s.bitcounter += 8 * s.stacked_zero + 8, s.stacked_zero = 0;
    }
    s.pending_byte = b & 0xFF, s.is_pending_byte = 1;
}
__device__ static void bc_shift(SbacState &s)
{
    s.code <<= 1;
    if(--s.code_bits) return;
    const unsigned out = s.code >> 17;
    s.code &= (1u << 17) - 1;
    if(out < 0xFF) {
        for(; s.stacked_ff; s.stacked_ff--) bc_byte(s, 0xFF);
        bc_byte(s, out);
    }
    else if(out > 0xFF) {
        s.pending_byte++;
        for(; s.stacked_ff; s.stacked_ff--) bc_byte(s, 0);
        bc_byte(s, out);
    }
    else s.stacked_ff++;
    s.code_bits = 8;
}
__device__ static unsigned split_flag_bits(const SbacState &from, SbacState &to, int split)
{
    SbacState s = from;
    s.code &= 0x7FFFF, s.code_bits = 11;
    s.pending_byte = s.is_pending_byte = s.stacked_ff = s.stacked_zero = s.bitcounter = s.bin_counter = 0;
    unsigned state = s.ctx[XEVE_HIP_CTX_SPLIT_CU] >> 1, mps = s.ctx[XEVE_HIP_CTX_SPLIT_CU] & 1;
    unsigned lps = (state * s.range) >> 9;
    if(lps < 437) lps = 437;
    s.bin_counter++;
    s.range -= lps;
    if((unsigned)(split != 0) != mps) {
        if(s.range >= lps) s.code += s.range, s.range = lps;
        state = state + ((512 - state + 16) >> 5);
        if(state > 256) mps = 1 - mps, state = 512 - state;
    }
    else state = state - ((state + 16) >> 5);
    s.ctx[XEVE_HIP_CTX_SPLIT_CU] = (uint16_t)((state << 1) + mps);
    while(s.range < 8192) s.range <<= 1, bc_shift(s);
    to = s;
    return s.bitcounter + 8 * (s.stacked_zero + s.stacked_ff) + 8 * (s.is_pending_byte ? 1 : 0) + 8 - s.code_bits + 3;
}

// the same on the state where it lies: only the coder's registers and the one model change (round 6: the operations below move a state with all lanes of the block and
// thread 0 touches these ten words -- a 180-byte copy in, the bin, a 180-byte copy out by one lane were three dependent round trips to HBM per call)
__device__ static unsigned split_flag_inplace(SbacState *st, int split)
{
    SbacState s;
    s.range = st->range, s.code = st->code & 0x7FFFF, s.code_bits = 11;
    s.pending_byte = s.is_pending_byte = s.stacked_ff = s.stacked_zero = s.bitcounter = s.bin_counter = 0;
    const unsigned m = st->ctx[XEVE_HIP_CTX_SPLIT_CU];
    unsigned state = m >> 1, mps = m & 1;
    unsigned lps = (state * s.range) >> 9;
    if(lps < 437) lps = 437;
    s.bin_counter++;
    s.range -= lps;
    if((unsigned)(split != 0) != mps) {
        if(s.range >= lps) s.code += s.range, s.range = lps;
        state = state + ((512 - state + 16) >> 5);
        if(state > 256) mps = 1 - mps, state = 512 - state;
    }
    else state = state - ((state + 16) >> 5);
    while(s.range < 8192) s.range <<= 1, bc_shift(s);
    st->range = s.range, st->code = s.code, st->code_bits = s.code_bits, st->stacked_ff = s.stacked_ff, st->stacked_zero = s.stacked_zero, st->pending_byte = s.pending_byte;
    st->is_pending_byte = s.is_pending_byte, st->bitcounter = s.bitcounter, st->bin_counter = s.bin_counter;
    st->ctx[XEVE_HIP_CTX_SPLIT_CU] = (uint16_t)((state << 1) + mps);
    return s.bitcounter + 8 * (s.stacked_zero + s.stacked_ff) + 8 * (s.is_pending_byte ? 1 : 0) + 8 - s.code_bits + 3;
}

// ---- block-cooperative pieces (every thread of the workgroup calls them with the same arguments) ---------------------------------------------------------------
static_assert(sizeof(SbacState) % 4 == 0, "coder states are moved word by word");
// a coder state moved by all lanes (word i by the lane(s) with i = threadIdx.x mod blockDim.x); two destinations at once where a state is kept twice
__device__ static void st_copy(SbacState *dst, const SbacState *src, SbacState *dst2 = nullptr)
{
    const unsigned *a = (const unsigned *)src;
    unsigned       *b = (unsigned *)dst, *b2 = (unsigned *)dst2;
    for(int i = threadIdx.x; i < (int)(sizeof(SbacState) / 4); i += blockDim.x) {
        const unsigned v = a[i];
        b[i] = v;
        if(b2) b2[i] = v;
    }
}
__device__ static void st_zero(SbacState *dst)
{
    unsigned *b = (unsigned *)dst;
    for(int i = threadIdx.x; i < (int)(sizeof(SbacState) / 4); i += blockDim.x) b[i] = 0;
}
__device__ static void cud_init(CtuData *d, int log2)
{   // init_cu_data (:374-428): what the I-slice walk reads back -- split modes and luma / chroma modes cleared
    const int n = 1 << (2 * (log2 - 2));
    for(int u = threadIdx.x; u < n; u += blockDim.x) {
        for(int k = 0; k < XEVE_HIP_CU_DEPTHS; k++) d->split_mode[k][u] = 0;
        d->ipm[0][u] = 0, d->ipm[1][u] = 0;
    }
}
// copy_cu_data (:430-620): the sub-block (x, y; log2) of dst (pitch 1 << log2_cus) <- all of src, split modes from depth cud on
__device__ static void cud_copy(CtuData *dst, const CtuData *src, int x, int y, int log2, int log2_cus, int cud, int idc, int ws, int hs)
{
    const int n = 1 << (log2 - 2), cus = 1 << (log2_cus - 2), cw = 1 << log2, cs = 1 << log2_cus;
    for(int u = threadIdx.x; u < n * n; u += blockDim.x) {
        const int j = u / n, i = u - j * n, di = ((y >> 2) + j) * cus + (x >> 2) + i, si = u;
        for(int k = cud; k < XEVE_HIP_CU_DEPTHS; k++) dst->split_mode[k][di] = src->split_mode[k][si];
        dst->pred_mode[di] = src->pred_mode[si], dst->ipm[0][di] = src->ipm[0][si], dst->ipm[1][di] = src->ipm[1][si], dst->depth[di] = src->depth[si];
        dst->map_scu[di] = src->map_scu[si], dst->map_cu_mode[di] = src->map_cu_mode[si];
        for(int c = 0; c < 3; c++) dst->nnz[c][di] = src->nnz[c][si];
        for(int k = 0; k < 4; k++) (&dst->mv[di][0][0])[k] = (&src->mv[si][0][0])[k], (&dst->mvd[di][0][0])[k] = (&src->mvd[si][0][0])[k];
        for(int k = 0; k < 2; k++) dst->refi[di][k] = src->refi[si][k], dst->mvp_idx[di][k] = src->mvp_idx[si][k];
    }
    for(int t = threadIdx.x; t < cw * cw; t += blockDim.x) {
        const int j = t >> log2, i = t & (cw - 1), d = (y + j) * cs + x + i;
        dst->coef[0][d] = src->coef[0][t], dst->reco[0][d] = src->reco[0][t];
    }
    if(idc) {
        const int wc = cw >> ws, hc = cw >> hs, sc = cs >> ws;
        for(int t = threadIdx.x; t < wc * hc; t += blockDim.x) {
            const int j = t / wc, i = t - j * wc, d = ((y >> hs) + j) * sc + (x >> ws) + i;
            dst->coef[1][d] = src->coef[1][t], dst->reco[1][d] = src->reco[1][t];
            dst->coef[2][d] = src->coef[2][t], dst->reco[2][d] = src->reco[2][t];
        }
    }
}
__device__ static void clear_map(const TreeK &K, int pic, int x, int y, int cu)
{   // clear_map_scu (:1129-1155)
    const int w = (x + cu > K.pic_w ? K.pic_w - x : cu) >> 2, h = (y + cu > K.pic_h ? K.pic_h - y : cu) >> 2;
    uint32_t *ms = K.map_scu + (long)pic * K.map_pic, *mc = K.map_cu_mode + (long)pic * K.map_pic;
    for(int t = threadIdx.x; t < w * h; t += blockDim.x) {
        const int j = t / w, i = t - j * w, g = ((y >> 2) + j) * K.w_scu + (x >> 2) + i;
        ms[g] = 0, mc[g] = 0;
    }
}
__device__ static void update_map(const TreeK &K, int pic, const CtuData *d, int x, int y, int cu)
{   // update_map_scu (:1036-1127) + the intra part of update_to_ctx_map (:2445-2516): the maps the intra analysis of later CUs reads
    const int w = (x + cu > K.pic_w ? K.pic_w - x : cu) >> 2, h = (y + cu > K.pic_h ? K.pic_h - y : cu) >> 2, n = cu >> 2;
    uint32_t *ms = K.map_scu + (long)pic * K.map_pic, *mc = K.map_cu_mode + (long)pic * K.map_pic;
    int8_t   *mi = K.map_ipm + (long)pic * K.map_pic;
    for(int t = threadIdx.x; t < w * h; t += blockDim.x) {
        const int j = t / w, i = t - j * w, g = ((y >> 2) + j) * K.w_scu + (x >> 2) + i, u = j * n + i;
        ms[g] = d->map_scu[u], mc[g] = d->map_cu_mode[u], mi[g] = d->ipm[0][u];
        if(K.inter) {
            const long gm = (long)pic * K.map_pic + g;
            for(int k = 0; k < 4; k++) (&K.map_mv[gm][0][0])[k] = (&d->mv[u][0][0])[k];
            K.map_refi[gm][0] = d->refi[u][0], K.map_refi[gm][1] = d->refi[u][1];
        }
    }
}
__device__ static void rec_to_pic(const TreeK &K, int pic, const CtuData *d, int x, int y, int cu)
{   // mode_cpy_rec_to_ref (:797-866)
    const int w = x + cu > K.pic_w ? K.pic_w - x : cu, h = y + cu > K.pic_h ? K.pic_h - y : cu;
    pel *m = K.mod[0] + (long)pic * K.mod_pic_l;
    for(int t = threadIdx.x; t < w * h; t += blockDim.x) {
        const int j = t / w, i = t - j * w;
        m[(long)(y + j) * K.s_mod_l + x + i] = d->reco[0][j * cu + i];
    }
    if(K.idc) {
        const int wc = w >> K.ws, hc = h >> K.hs, sc = cu >> K.ws;
        pel *mu = K.mod[1] + (long)pic * K.mod_pic_c, *mv = K.mod[2] + (long)pic * K.mod_pic_c;
        for(int t = threadIdx.x; t < wc * hc; t += blockDim.x) {
            const int  j = t / wc, i = t - j * wc;
            const long g = (long)((y >> K.hs) + j) * K.s_mod_c + (x >> K.ws) + i;
            mu[g] = d->reco[1][j * sc + i], mv[g] = d->reco[2][j * sc + i];
        }
    }
}

// ---- the tree operations -------------------------------------------------------------------------------------------------------------------------------------
#define AT(arr, L) ((arr) + (long)(L) * K.nchains + c)

__device__ static void op_enter(const TreeK &K, int c, int L, int part, int *sh)
{
    const xeve_hip_ctu_job J = K.jobs[c];
    Node *nd = AT(K.node, L);
    const int log2 = L + 2, cu = 1 << log2;
    if(threadIdx.x == 0) {
        int active, x0, y0;
        if(part < 0) active = 1, x0 = J.x, y0 = J.y;
        else {
            const Node *p = AT(K.node, L + 1);
            x0 = p->x0 + (part & 1) * cu, y0 = p->y0 + (part >> 1) * cu;
            active = p->active && p->do_split && x0 < K.pic_w && y0 < K.pic_h;
        }
        const int boundary = active && !(x0 + cu <= K.pic_w && y0 + cu <= K.pic_h), leaf = active && !boundary && cu <= K.max_cu;
        nd->active = active, nd->x0 = x0, nd->y0 = y0, nd->leaf = leaf;
        // a chain whose node is off analyses a CU of this size that lies inside the picture (the schedule runs no analysis of a size the picture cannot hold)
        const int jx = leaf ? x0 : max(0, min(J.x, K.pic_w - cu)), jy = leaf ? y0 : max(0, min(J.y, K.pic_h - cu));
        xeve_hip_intra_job ij;
        memset(&ij, 0, sizeof(ij));
        ij.x = jx, ij.y = jy, ij.inter_satd = 0xFFFFFFFFu, ij.sbac = c, ij.pic = J.pic;
        const TreeK::Ac &A = K.ac[K.side_of[L]];
        *AT(A.ijobs, L) = ij;
        if(K.inter) {
            xeve_hip_inter_job ej;
            memset(&ej, 0, sizeof(ej));
            ej.x = jx, ej.y = jy + J.pic * K.vh, ej.sbac = c; // ctx_skip / ctx_pred_mode: 0 without sps_cm_init_flag (xeve_get_ctx_some_flags, xeve_util.c:1181-1288)
            *AT(A.ejobs, L) = ej;
            *AT(A.sjobs, L) = xh_make_job((long)jy + (long)J.pic * K.vh, K.s_org_l, jx, c * cu * cu);
            nd->try_intra = 0, nd->cu_mode = 0, nd->unit_cost = MAX_COST;
        }
        sh[0] = active, sh[1] = leaf, sh[2] = boundary, sh[3] = x0, sh[4] = y0;
    }
    __syncthreads();
    const int active = sh[0], leaf = sh[1], boundary = sh[2], x0 = sh[3], y0 = sh[4];
    if(active) { // the node's entry state: the caller's (root), the one behind the parent's split flag (first quadrant), what the previous quadrant's winner left (:2248-2262);
                 // kept twice (s_curr_best / s_curr_before_split, :2061); s_temp_depth cleared (:2031)
        const SbacState *src = part < 0 ? K.states + J.sbac : part == 0 ? AT(K.csplit, L + 1) : AT(K.next, L);
        st_copy(AT(K.curr, L), src, AT(K.before, L));
        st_zero(AT(K.tdepth, L));
    }
    __syncthreads();
    if(threadIdx.x == 0 && active) {
        nd->cost_best = MAX_COST, nd->best_split = 0, nd->do_split = 0, nd->dist_cu = 0;
        double cost_temp = 0.0;
        if(!boundary) {
            if(leaf) {
                if(cu > K.min_cuwh) cost_temp += (double)(int)split_flag_inplace(AT(K.curr, L), 0) * K.lambda0; // split_cu_flag = 0 (:2079-2091)
            }
            else cost_temp = MAX_COST;
        }
        nd->cost_temp = cost_temp;
    }
    // the node's own CU is staged straight into cu_data_best (round 6: it went through cu_data_temp and was copied: the unsplit alternative is always the first one stored)
    if(active && !boundary) cud_init(AT(K.best, L), log2);
    if(leaf) clear_map(K, J.pic, x0, y0, cu);
}

// copy_to_cu_data (:868-1034) of the node's CU into its cu_data_temp: the unit fields, then the dense blocks
__device__ static void unit_to_temp(const TreeK &K, CtuData *t, int log2, int cud, int n, int cu_mode, const int8_t *ipm, const int32_t *nnz, const xeve_hip_inter_result *R,
                                    const int16_t *cy, const int16_t *cu_, const int16_t *cv, const pel *ry, const pel *ru, const pel *rv, int n0, int n1)
{
    const uint32_t scu = ((uint32_t)K.slice_num & 0x7F) | ((uint32_t)K.slice_qp << 16) | (1u << 31) | (cu_mode == 0 ? 1u << 15 : 0) | (cu_mode == 2 ? 1u << 23 : 0); // _SN, _QP, _COD, _IF, _SF
    const uint32_t cum = ((uint32_t)log2 << 24) | ((uint32_t)log2 << 28);                                                                                            // MCU_SET_LOGW / LOGH
    for(int u = threadIdx.x; u < n; u += blockDim.x) {
        t->pred_mode[u] = (uint8_t)cu_mode, t->depth[u] = (int8_t)cud;
        if(cu_mode == 0) t->ipm[0][u] = ipm[0], t->ipm[1][u] = K.idc ? ipm[1] : 0;
        t->nnz[0][u] = nnz[0], t->nnz[1][u] = K.idc ? nnz[1] : 0, t->nnz[2][u] = K.idc ? nnz[2] : 0;
        t->map_scu[u] = scu, t->map_cu_mode[u] = cum;
        for(int k = 0; k < 4; k++) (&t->mv[u][0][0])[k] = R ? (&R->mv[0][0])[k] : 0, (&t->mvd[u][0][0])[k] = R ? (&R->mvd[0][0])[k] : 0;
        for(int k = 0; k < 2; k++) t->refi[u][k] = R ? R->refi[k] : -1, t->mvp_idx[u][k] = R ? R->mvp_idx[k] : 0;
    }
    for(int i = threadIdx.x; i < n0; i += blockDim.x) t->coef[0][i] = cy[i], t->reco[0][i] = ry[i];
    if(K.idc)
        for(int i = threadIdx.x; i < n1; i += blockDim.x) t->coef[1][i] = cu_[i], t->reco[1][i] = ru[i], t->coef[2][i] = cv[i], t->reco[2][i] = rv[i];
}

// P / B slices, after the inter analysis: mode_check_inter's store (:1199-1218) and what mode_check_intra needs (:1245-1262)
__device__ static void op_mid(const TreeK &K, int c, int L)
{
    Node *nd = AT(K.node, L);
    if(!nd->leaf) return;
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (K.log2_ctu - log2), n = 1 << (2 * L), n0 = cu * cu, n1 = K.idc ? n0 >> (K.ws + K.hs) : 0;
    const TreeK::Ac &A = K.ac[K.side_of[L]];
    const xeve_hip_inter_result R = A.eres[c];
    unit_to_temp(K, AT(K.best, L), log2, cud, n, R.cu_mode, nullptr, R.nnz, &A.eres[c], A.ecoef + (long)c * n0, A.ecoef + (long)K.nchains * n0 + (long)c * n1,
                 A.ecoef + (long)K.nchains * (n0 + n1) + (long)c * n1, A.erec[0] + (long)c * n0, A.erec[1] + (long)c * n1, A.erec[2] + (long)c * n1, n0, n1);
    if(threadIdx.x == 0) {
        nd->unit_cost = R.cost, nd->cu_mode = R.cu_mode;
        nd->try_intra = R.nnz[0] != 0 || R.nnz[1] != 0 || R.nnz[2] != 0;
        if(nd->try_intra) AT(A.ijobs, L)->inter_satd = (uint32_t)A.esatd[c]; // core->inter_satd (a chain that does not try intra keeps the job: its result is dropped)
    }
}

// mode_coding_unit's end (:1310-1350) and the store of the unsplit alternative (:2116-2137): the intra analysis becomes the CU's mode in an I slice, and in a P / B slice
// where it is cheaper than the inter winner.  to_pic: mode_cpy_rec_to_ref now (a side node leaves it to op_exit: its children are writing the same samples meanwhile,
// nothing reads the node's own reconstruction before op_exit puts the winner's there anyway)
__device__ static void leaf_decide(const TreeK &K, int c, int L, int *sh, bool to_pic)
{
    const xeve_hip_ctu_job J = K.jobs[c];
    Node *nd = AT(K.node, L);
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (K.log2_ctu - log2), n = 1 << (2 * L), n0 = cu * cu, n1 = K.idc ? n0 >> (K.ws + K.hs) : 0;
    CtuData *b = AT(K.best, L);
    const TreeK::Ac &A = K.ac[K.side_of[L]];
    const xeve_hip_intra_result R = A.ires[c];
    const int intra_wins = !K.inter || (nd->try_intra && R.cost < nd->unit_cost);
    if(intra_wins) // (over the inter winner op_mid left there)
        unit_to_temp(K, b, log2, cud, n, 0, R.ipm, R.nnz, nullptr, A.icoef + (long)c * n0, A.icoef + (long)K.nchains * n0 + (long)c * n1,
                     A.icoef + (long)K.nchains * (n0 + n1) + (long)c * n1, A.irec + (long)c * n0, A.irec + (long)K.nchains * n0 + (long)c * n1,
                     A.irec + (long)K.nchains * (n0 + n1) + (long)c * n1, n0, n1);
    __syncthreads(); // (every wave has read nd->unit_cost for intra_wins above before thread 0 moves it below: a block of four waves, levels 32x32 and 64x64)
    if(threadIdx.x == 0) {
        if(intra_wins) nd->unit_cost = R.cost, nd->cu_mode = 0, nd->dist_cu = R.dist_cu;
        else nd->dist_cu = 0x7FFFFFFF;
        const double cost_temp = nd->cost_temp + nd->unit_cost;
        sh[0] = nd->cost_best > cost_temp;
        if(sh[0]) nd->cost_best = cost_temp, nd->best_split = 0; // (:2116-2135)
        nd->cost_temp = nd->cost_best;
    }
    __syncthreads(); // (also: the CU's blocks above are in cu_data_best for every thread)
    const int better = sh[0];
    if(better) {
        st_copy(AT(K.tdepth, L), intra_wins ? A.sbest + c : A.enext + c);
        if(to_pic) rec_to_pic(K, J.pic, b, nd->x0, nd->y0, cu);
    }
    __syncthreads();
}
// the early terminations behind the unsplit alternative (:2162-2187): 0 = the node is not split whatever its children would cost
__device__ static int leaf_next_split(const TreeK &K, const Node *nd, int L)
{
    const int log2 = L + 2, cud = 2 * (K.log2_ctu - log2);
    int next_split = 1;
    if(nd->active && nd->cost_best != MAX_COST && K.inter && cud >= K.ecu_depth && nd->cu_mode == 2 /* MODE_SKIP */) next_split = 0; // early CU termination (:2162-2172)
    if(nd->active && nd->cost_best != MAX_COST && !K.inter) { // early termination in I pictures (:2174-2187)
        const int th = 1 << (2 * log2 + 7);
        if(nd->dist_cu < th) {
            const int bits_inc = (2 * log2 >= 6 ? 2 : 0) + 8;
            if(nd->dist_cu < K.lambda0 * bits_inc) next_split = 0;
        }
    }
    return next_split;
}

__device__ static void op_leaf(const TreeK &K, int c, int L, int *sh)
{
    const xeve_hip_ctu_job J = K.jobs[c];
    Node *nd = AT(K.node, L);
    const int log2 = L + 2, cu = 1 << log2;
    const int active = nd->active, x0 = nd->x0, y0 = nd->y0;
    CtuData *t = AT(K.temp, L);
    if(nd->leaf) leaf_decide(K, c, L, sh, true);
    if(threadIdx.x == 0) {
        const int do_split = active && cu > 4 && leaf_next_split(K, nd, L) && cu > K.min_cu && cu > K.min_cuwh;
        nd->do_split = do_split;
        if(do_split) { // SPLIT_QUAD (:2189-2329): split_cu_flag = 1 from the node's entry state
            SbacState run;
            nd->cost_temp = (double)(int)split_flag_bits(*AT(K.before, L), run, 1) * K.lambda0;
            *AT(K.curr, L) = run, *AT(K.csplit, L) = run;
        }
        sh[0] = do_split;
    }
    __syncthreads();
    const int do_split = sh[0];
    __syncthreads();
    if(do_split) {
        cud_init(t, log2);
        clear_map(K, J.pic, x0, y0, cu);
    }
}

// ---- a node whose analyses run on the side stream ------------------------------------------------------------------------------------------------------------
// main stream, right behind op_enter: the split alternative is set up WITHOUT the node's own verdict -- whether the early terminations (leaf_next_split) forbid the
// split is only known when the side stream is through, so the children are walked in any case (in lockstep they are analysed in any case: a chain whose node is
// off rides along) and op_exit drops them where the verdict says so.  What the children leave in the picture and in the maps meanwhile lies inside the node: op_exit
// puts the winner's reconstruction there and the parent's op_child_done / op_root_done the winner's map entries, before anything reads either.
__device__ static void op_split_prep(const TreeK &K, int c, int L, int *sh)
{
    const xeve_hip_ctu_job J = K.jobs[c];
    Node *nd = AT(K.node, L);
    const int log2 = L + 2, cu = 1 << log2;
    if(threadIdx.x == 0) {
        const int do_split = nd->active && cu > 4 && cu > K.min_cu && cu > K.min_cuwh;
        nd->do_split = do_split, nd->next_split = 1;
        if(do_split) {
            SbacState run;
            nd->cost_split = (double)(int)split_flag_bits(*AT(K.before, L), run, 1) * K.lambda0;
            *AT(K.csplit, L) = run;
        }
        sh[0] = do_split;
    }
    __syncthreads();
    const int do_split = sh[0];
    __syncthreads();
    if(do_split) {
        cud_init(AT(K.tsplit, L), log2);
        clear_map(K, J.pic, nd->x0, nd->y0, cu);
    }
}
// side stream, behind the node's analyses: the unsplit alternative and the verdict
__device__ static void op_leaf_side(const TreeK &K, int c, int L, int *sh)
{
    Node *nd = AT(K.node, L);
    if(nd->leaf) leaf_decide(K, c, L, sh, false);
    if(threadIdx.x == 0) nd->next_split = leaf_next_split(K, nd, L);
}

__device__ static void op_child_done(const TreeK &K, int c, int L, int part)
{   // L = the parent's level; the quadrant just left is node (L - 1)
    const xeve_hip_ctu_job J = K.jobs[c];
    Node       *p  = AT(K.node, L);
    const Node *ch = AT(K.node, L - 1);
    if(!ch->active) return;
    const int log2 = L + 2, cud = 2 * (K.log2_ctu - log2), half = 1 << (log2 - 1);
    const int side = K.side_of[L];
    if(threadIdx.x == 0) (side ? p->cost_split : p->cost_temp) += ch->cost_best;
    cud_copy(side ? AT(K.tsplit, L) : AT(K.temp, L), AT(K.best, L - 1), ch->x0 - p->x0, ch->y0 - p->y0, log2 - 1, log2, cud, K.idc, K.ws, K.hs);
    update_map(K, J.pic, AT(K.best, L - 1), ch->x0, ch->y0, half);
    (void)part;
}

__device__ static void op_exit(const TreeK &K, int c, int L, int *sh)
{
    const xeve_hip_ctu_job J = K.jobs[c];
    Node *nd = AT(K.node, L);
    if(!nd->active) return;
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (K.log2_ctu - log2);
    const int side = K.side_of[L];
    if(threadIdx.x == 0) {
        const double cost_split = side ? nd->cost_split : nd->cost_temp;
        sh[0] = nd->do_split && (!side || nd->next_split) && nd->cost_best - 0.0001 > cost_split;
        if(sh[0]) nd->cost_best = cost_split, nd->best_split = 5 /* SPLIT_QUAD */;
    }
    __syncthreads();
    const int split_wins = sh[0];
    // s_temp_depth = the last quadrant's exit state where the split wins; s_next_best of the node = s_temp_depth (a lane re-reads the word it wrote itself)
    if(split_wins) st_copy(AT(K.tdepth, L), AT(K.next, L - 1));
    st_copy(AT(K.next, L), AT(K.tdepth, L));
    __syncthreads();
    CtuData *b = AT(K.best, L);
    if(split_wins) {
        cud_copy(b, side ? AT(K.tsplit, L) : AT(K.temp, L), 0, 0, log2, log2, cud, K.idc, K.ws, K.hs);
        __syncthreads();
    }
    rec_to_pic(K, J.pic, b, nd->x0, nd->y0, cu);
    if(cu >= 8 && threadIdx.x == 0) b->split_mode[cud][((cu >> 1) >> 2) * (cu >> 2) + ((cu >> 1) >> 2)] = (int8_t)nd->best_split; // xeve_set_split_mode (xeve_util.c:1148-1161)
}

__device__ static void op_root_done(const TreeK &K, int c, int L)
{   // update_to_ctx_map + update_map_scu (:2455-2516), then the products the caller takes
    const xeve_hip_ctu_job J = K.jobs[c];
    const Node    *nd = AT(K.node, L);
    const CtuData *b  = AT(K.best, L);
    update_map(K, J.pic, b, nd->x0, nd->y0, 1 << (L + 2));
    const uint32_t *s = (const uint32_t *)b;
    uint32_t       *d = (uint32_t *)(K.out + c);
    for(int i = threadIdx.x; i < (int)(sizeof(CtuData) / 4); i += blockDim.x) d[i] = s[i];
    if(threadIdx.x == 0) K.out_next[c] = *AT(K.next, L), K.out_cost[c] = nd->cost_best;
}

__global__ void __launch_bounds__(256) k_tree_ops(TreeK K, OpList ops)
{
    __shared__ int sh[8];
    const int c = blockIdx.x;
    for(int i = 0; i < ops.n; i++) {
        const int L = ops.lvl[i], part = ops.part[i];
        switch(ops.op[i]) {
        case OP_ENTER: op_enter(K, c, L, part, sh); break;
        case OP_LEAF: op_leaf(K, c, L, sh); break;
        case OP_CHILD_DONE: op_child_done(K, c, L, part); break;
        case OP_EXIT: op_exit(K, c, L, sh); break;
        case OP_MID: op_mid(K, c, L); break;
        case OP_SPLIT_PREP: op_split_prep(K, c, L, sh); break;
        case OP_LEAF_SIDE: op_leaf_side(K, c, L, sh); break;
        default: op_root_done(K, c, L); break;
        }
        __syncthreads(); // (a workgroup-scope release / acquire: the next operation reads what this one wrote to global memory)
    }
}

// ---- the intra analysis of a 4x4 / 8x8 node: one LANE per chain (cu_lane.h) ------------------------------------------------------------------------------------------
// 320 of the 341 nodes of an I-picture CTU.  Same inputs and outputs as the batched composite (xeve_hip_pintra_analyze_cu_jobs): the node's job, the chain's entry
// coder state; result record, dense levels and reconstruction (Y of all chains, then U, then V), core->s_temp_best -- so the tree operations around it do not change.
template <int LOG2> __global__ void __launch_bounds__(64) k_intra_lane(TreeK K, xl::Params P, const pel *org_y, const pel *org_u, const pel *org_v, const uint8_t *map_tidx,
                                                                        long org_pic_l, long org_pic_c)
{
    // one chain per WAVE, one lane working: the chains of a wave took every branch of this long scalar code one after the other (the writer kernels of encode.hip
    // met the same wall: 16 chains per wave ran 13 x slower than a wave each)
    const int c = blockIdx.x, L = LOG2 - 2;
    if(c >= K.nchains || threadIdx.x != 0) return;
    const Node *nd = AT(K.node, L);
    if(!nd->leaf || (K.inter && !nd->try_intra)) return; // (nothing reads the result of a chain whose node is off)
    const TreeK::Ac &A = K.ac[K.side_of[L]];
    const xeve_hip_intra_job J = *AT(A.ijobs, L);
    const int  n0 = 1 << (2 * LOG2), n1 = K.idc ? n0 >> (K.ws + K.hs) : 0;
    const pel *org[3] = {org_y + J.pic * org_pic_l, org_u ? org_u + J.pic * org_pic_c : nullptr, org_v ? org_v + J.pic * org_pic_c : nullptr};
    const pel *mod[3] = {K.mod[0] + J.pic * K.mod_pic_l, K.mod[1] ? K.mod[1] + J.pic * K.mod_pic_c : nullptr, K.mod[2] ? K.mod[2] + J.pic * K.mod_pic_c : nullptr};
    int16_t *coef = const_cast<int16_t *>(A.icoef);
    pel     *rec = const_cast<pel *>(A.irec);
    const long oy = (long)c * n0, ou = (long)K.nchains * n0 + (long)c * n1, ov = (long)K.nchains * (n0 + n1) + (long)c * n1;
    xeve_hip_intra_result R;
    xl::intra_cu<LOG2>(P, org, mod, K.map_scu + J.pic * K.map_pic, K.map_ipm + J.pic * K.map_pic, map_tidx + J.pic * K.map_pic, *AT(K.curr, L), J, R, coef + oy, coef + ou, coef + ov,
                       rec + oy, rec + ou, rec + ov, A.sbest[c]);
    const_cast<xeve_hip_intra_result *>(A.ires)[c] = R;
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------------------
extern "C" int xeve_hip_satd_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs, const int32_t *cand_off, int ncand, int w, int h,
                                  int bit_depth, int32_t *out, void *stream);
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
struct TreeLayout {
    size_t node, curr, next, before, tdepth, csplit, best, temp, tsplit, zero32, total, zero_from, zero_bytes;
    struct Ac {
        size_t sbest, ijobs, ires, icoef, irec, iws, iws_bytes, est;
        size_t ejobs, sjobs, eres, ecoef, erec[3], epred, esatd, enext, ews, ews_bytes; // P / B slices
    } ac[XT_STREAMS]; // the arrays of the analyses, per stream (TreeK::Ac)
    unsigned char side_of[8];
};
static bool tree_params_ok(const xeve_hip_tree_params *p)
{   // everything the analyses of the walk would refuse is refused here, before the first launch (a walk that stops half way leaves half-written maps behind)
    if(!(p && p->log2_ctu >= 3 && p->log2_ctu <= 6 && p->pic_w > 0 && p->pic_h > 0 && (p->pic_w & 3) == 0 && (p->pic_h & 3) == 0 && xh_pow2(p->max_cu) &&
         xh_pow2(p->min_cu) && p->min_cu >= 4 && p->max_cu >= p->min_cu && p->min_cuwh >= 4 && xh_pow2(p->min_cuwh) && p->ip.w_scu == (p->pic_w + 3) >> 2 &&
         p->ip.h_scu == (p->pic_h + 3) >> 2 && (p->rdo_dbk == 0 || p->rdo_dbk == 1)))
        return false;
    const xeve_hip_intra_params &ip = p->ip;
    if(!(ip.tool_iqt == 0 && ip.bit_depth >= 8 && ip.bit_depth <= 14 && (ip.chroma_format_idc == 0 || ip.chroma_format_idc == 1 || ip.chroma_format_idc == 3) &&
         ip.slice_type >= 0 && ip.slice_type <= 2 && p->slice_qp >= 0 && p->slice_qp <= 127))
        return false;
    for(int k = 0; k < (ip.chroma_format_idc ? 3 : 1); k++)
        if(ip.qp[k] < 0 || ip.qp[k] > 51 + 6 * (ip.bit_depth - 8)) return false;
    return true;
}
static bool tree_inter_ok(const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I)
{   // (the composed inter analysis covers square CUs 8 .. 64 -- every inter CU of the Baseline quad-tree at the presets that keep min_cu_inter at 8; 4x4 inter CUs
    // -- preset placebo -- are the fused walk's: xh_walk_only)
    return I && (p->ip.slice_type == 0 || p->ip.slice_type == 1) && p->min_cu >= 4 && I->refp && I->map_mv && I->map_refi && I->col_mv0 && I->coef_l && I->coef_c &&
           (p->ip.slice_type == 1 || I->col_mv1) && I->ipar.rdo.slice_type == p->ip.slice_type && I->ipar.rdo.pic_w == p->pic_w && I->ipar.rdo.pic_h == p->pic_h &&
           I->ipar.rdo.chroma_format_idc == p->ip.chroma_format_idc && I->ipar.rdo.bit_depth == p->ip.bit_depth &&
           // what xeve_hip_pinter_analyze_cu_jobs requires of its parameters (inter.hip inter_params_ok), checked HERE so that both walks refuse the same calls: the fused
           // kernel indexes per-candidate and per-list arrays with these and divides by max_cand
           I->ipar.max_cand >= 1 && I->ipar.max_cand <= 4 && I->ipar.rdo.tool_iqt == 0 && I->ipar.rdo.num_refp[0] >= 1 && I->ipar.rdo.num_refp[0] <= XEVE_HIP_MAX_REFP &&
           (p->ip.slice_type == 1 || (I->ipar.rdo.num_refp[1] >= 1 && I->ipar.rdo.num_refp[1] <= I->ipar.rdo.num_refp[0])) && I->ipar.me.hpel_cnt >= 0 &&
           I->ipar.me.hpel_cnt <= 8 && I->ipar.me.qpel_cnt >= 0 && I->ipar.me.qpel_cnt <= 8;
}
static xeve_hip_intra_params level_params(const xeve_hip_tree_params *p, int log2)
{
    xeve_hip_intra_params ip = p->ip;
    ip.log2_cuw = ip.log2_cuh = log2;
    return ip;
}
static xeve_hip_inter_params level_inter_params(const xeve_hip_tree_inter *I, int log2)
{
    xeve_hip_inter_params ep = I->ipar;
    ep.rdo.log2_cuw = ep.rdo.log2_cuh = log2;
    return ep;
}
static xl::Params lane_params(const xeve_hip_tree_params *p, int log2, int s_org_l, int s_org_c, int s_mod_l, int s_mod_c)
{
    static const int q_scale[6] = {26214, 23302, 20560, 18396, 16384, 14764}, dq_scale[6] = {40, 45, 51, 57, 64, 71}; // xeve_quant_scale[0] (xeve_tq.c:37), xeve_tbl_dq_scale_b (xeve_tbl.c:237)
    xl::Params P;
    const xeve_hip_intra_params &ip = p->ip;
    const int idc = ip.chroma_format_idc, bd = ip.bit_depth, lc = log2 - (idc <= 2 ? 1 : 0);
    P.idc = idc, P.bd = bd, P.slice_type = ip.slice_type, P.cip = ip.constrained_intra_pred != 0, P.w_scu = ip.w_scu, P.h_scu = ip.h_scu;
    P.s_org_l = s_org_l, P.s_org_c = s_org_c, P.s_mod_l = s_mod_l, P.s_mod_c = s_mod_c;
    for(int c = 0; c < 3; c++) {
        const int q = ip.qp[c], log2_size = c ? lc : log2, tr_shift = 15 - bd - log2_size;
        P.qp[c] = q, P.q_scale[c] = q_scale[q % 6], P.dq_scale[c] = dq_scale[q % 6] << (q / 6), P.lambda[c] = ip.lambda[c];
        double e = (double)(1 << 15) * pow(2.0, -tr_shift); // ctx->err_scale[qp % 6][log2_size - 1], xeve_init_err_scale (xeve_tq.c:406-423)
        e = e / q_scale[q % 6] / (1 << (bd - 8));
        P.err_scale[c] = (int64_t)(e * (double)(1 << 20));
    }
    P.sqrt_lambda0 = ip.sqrt_lambda0, P.wgt[0] = ip.dist_chroma_weight[0], P.wgt[1] = ip.dist_chroma_weight[1];
    P.entropy = xh_entropy_table();
    return P;
}
// a node of this size can be a CU at all: within max_cu and no larger than the picture (the analyses of a size the picture cannot hold are left out of the schedule)
static bool level_has_cu(const xeve_hip_tree_params *p, int log2) { return (1 << log2) <= p->max_cu && (1 << log2) <= p->pic_w && (1 << log2) <= p->pic_h; }

// a node of this size has children in the walk (op_leaf's static part of do_split)
static bool level_has_kids(const xeve_hip_tree_params *p, int log2) { return (1 << log2) > 4 && (1 << log2) > p->min_cu && (1 << log2) > p->min_cuwh; }
// the side stream (TreeK): on unless switched off (XEVE_HIP_TREE_SIDE=0 / xeve_hip_walk_side(0): the one-stream walk of rounds 2-5, kept for A / B measurements and pinned by
// the GPU suite beside the default)
static std::atomic<int> g_tree_side{getenv("XEVE_HIP_TREE_SIDE") ? std::min(2, std::max(0, atoi(getenv("XEVE_HIP_TREE_SIDE")))) : 1};
extern "C" int xeve_hip_walk_side(int on)
{
    const int before = g_tree_side.load();
    if(on >= 0 && on <= 2) g_tree_side.store(on);
    return before;
}

static TreeLayout tree_layout(int nchains, const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int s_org_l, int s_org_c, int side)
{
    TreeLayout L;
    memset(&L, 0, sizeof(L));
    const size_t N = (size_t)nchains;
    const int    idc = p->ip.chroma_format_idc;
    for(int log2 = 2; log2 <= p->log2_ctu; log2++) L.side_of[log2 - 2] = side && level_has_cu(p, log2) && level_has_kids(p, log2) ? (side == 2 ? log2 - 2 : 1) : 0; // (2: a side stream per LEVEL, a measurement setting: 5 % slower than one side stream at the bench's width, profiles/r06_side_stream.md)
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    L.zero_from = o;
    L.node = take(5 * N * sizeof(Node)), L.curr = take(5 * N * sizeof(SbacState)), L.next = take(5 * N * sizeof(SbacState)), L.before = take(5 * N * sizeof(SbacState));
    L.tdepth = take(5 * N * sizeof(SbacState)), L.csplit = take(5 * N * sizeof(SbacState)), L.best = take(5 * N * sizeof(CtuData)), L.temp = take(5 * N * sizeof(CtuData));
    L.tsplit = side ? take(5 * N * sizeof(CtuData)) : L.temp;
    for(int a = 0; a < XT_STREAMS; a++) L.ac[a].sbest = take(N * sizeof(SbacState));
    L.zero32 = take(64);
    L.zero_bytes = o - L.zero_from;
    for(int a = 0; a < XT_STREAMS; a++) { // every array of the analyses at the size of the largest CU its stream analyses (one stream: all of them in ac[0])
        TreeLayout::Ac &A = L.ac[a];
        int top = 0;
        for(int log2 = 2; log2 <= p->log2_ctu; log2++)
            if(level_has_cu(p, log2) && L.side_of[log2 - 2] == a) top = 1 << log2;
        if(!top) continue;
        const int n0 = top * top, n1 = idc ? n0 >> ((idc <= 2) + (idc <= 1)) : 0;
        A.ijobs = take(5 * N * sizeof(xeve_hip_intra_job)), A.ires = take(N * sizeof(xeve_hip_intra_result)), A.est = take(N * sizeof(xeve_hip_rdoq_est_full));
        A.icoef = take(N * ((size_t)n0 + 2 * (size_t)n1) * 2 + 64), A.irec = take(N * ((size_t)n0 + 2 * (size_t)n1) * sizeof(pel) + 64);
        if(I) {
            A.ejobs = take(5 * N * sizeof(xeve_hip_inter_job)), A.sjobs = take(5 * N * sizeof(xeve_hip_job)), A.eres = take(N * sizeof(xeve_hip_inter_result));
            A.ecoef = take(N * ((size_t)n0 + 2 * (size_t)n1) * 2 + 64);
            for(int c = 0; c < 3; c++) A.erec[c] = take(N * (size_t)(c ? n1 : n0) * sizeof(pel) + 64);
            A.epred = take(N * (size_t)n0 * sizeof(pel) + 64), A.esatd = take(N * 4), A.enext = take(N * sizeof(SbacState));
        }
        for(int log2 = 2; log2 <= p->log2_ctu; log2++) {
            if(!level_has_cu(p, log2) || L.side_of[log2 - 2] != a) continue;
            const xeve_hip_intra_params ip = level_params(p, log2);
            A.iws_bytes = std::max(A.iws_bytes, xeve_hip_pintra_analyze_cu_workspace(nchains, nchains, &ip));
            if(I) {
                const xeve_hip_inter_params ep = level_inter_params(I, log2);
                A.ews_bytes = std::max(A.ews_bytes, xeve_hip_pinter_analyze_cu_workspace(nchains, nchains, &ep, s_org_l, s_org_c));
            }
        }
        A.iws = take(A.iws_bytes), A.ews = take(A.ews_bytes);
    }
    L.total = o;
    return L;
}
// what a workspace must hold whichever way the call then runs (the side stream can be switched between the query and the call)
static size_t tree_workspace(int nchains, const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int s_org_l, int s_org_c)
{
    return std::max(std::max(tree_layout(nchains, p, I, s_org_l, s_org_c, 0).total, tree_layout(nchains, p, I, s_org_l, s_org_c, 1).total),
                    tree_layout(nchains, p, I, s_org_l, s_org_c, 2).total);
}

// the fused walk (walk.hip): one launch per call
bool   xh_walk_supported(const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int nchains);
size_t xh_walk_workspace(int nchains);
int    xh_walk_run(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                   const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems, const xeve_hip_sbac *states, const xeve_hip_tree_params *p,
                   const xeve_hip_tree_inter *I, const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, void *workspace,
                   size_t workspace_bytes, int vh, hipStream_t st);

extern "C" size_t xeve_hip_mode_analyze_ctu_workspace(int nchains, const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int s_org_l, int s_org_c)
{
    if(!tree_params_ok(p) || nchains <= 0 || (p->ip.slice_type != 2 && !tree_inter_ok(p, I))) return 0;
    if(p->ip.slice_type == 2) I = nullptr;
    if(xh_walk_supported(p, I, nchains)) return xh_walk_workspace(nchains);
    if(xh_walk_only(p)) { xh_set_error("rdo_dbk_switch and 4x4 inter CUs (presets slow, placebo) run on the fused walk only: it is switched off (XEVE_HIP_WALK=0 / xeve_hip_walk_select) or does not take these parameters"); return 0; }
    return tree_workspace(nchains, p, I, s_org_l, s_org_c);
}
extern "C" size_t xeve_hip_mode_analyze_ctu_intra_workspace(int nchains, const xeve_hip_tree_params *p)
{
    if(!tree_params_ok(p) || nchains <= 0 || p->ip.slice_type != 2) return 0;
    if(xh_walk_supported(p, nullptr, nchains)) return xh_walk_workspace(nchains);
    if(xh_walk_only(p)) return 0;
    return tree_workspace(nchains, p, nullptr, 0, 0);
}

namespace {
struct TreeGraph {
    std::vector<char> key;
    int               seen  = 0;
    hipGraph_t        graph = nullptr;
    hipGraphExec_t    exec  = nullptr;
};
struct TreeGraphs { // per thread; dropped when the library is re-bound
    uint32_t               gen = 0;
    std::vector<TreeGraph> v;
    void drop()
    {
        if(!v.empty()) (void)hipDeviceSynchronize(); // (a replay may still be in flight)
        for(auto &g : v) {
            if(g.exec) (void)hipGraphExecDestroy(g.exec);
            if(g.graph) (void)hipGraphDestroy(g.graph);
        }
        v.clear();
    }
    TreeGraph *find(const std::vector<char> &key)
    {
        if(gen != xh_generation()) drop(), gen = xh_generation();
        for(auto &g : v)
            if(g.key == key) return &g;
        return nullptr;
    }
    void add(const std::vector<char> &key)
    {
        if(v.size() >= 64) drop(); // (a batch encoder's walk: I and P / B pictures x the chain counts of a wavefront's ramp = a few dozen distinct calls)
        TreeGraph g;
        g.key = key;
        v.push_back(std::move(g));
    }
    ~TreeGraphs() { drop(); }
};
enum { AN_NONE = 0, AN_INTRA = 1, AN_INTER = 2 };
// one entry of the schedule: [wait for an event] [tree operations] [an analysis] [record an event], all on one of the call's two streams
struct Item {
    int    st;               // 0 the caller's stream, 1 the side stream
    OpList ops;              // run before the analysis
    int    kind, size;       // the analysis: AN_*, log2 of the CU size
    int    wait_ev, rec_ev;  // -1: none
};
inline int ev_enter(int L) { return 2 * L; }    // main stream: the side node of level L is entered (its jobs and entry state stand)
inline int ev_done(int L) { return 2 * L + 1; } // side stream: its unsplit alternative is decided
struct Walk { // the static schedule of one CTU: every node of the full quad-tree in the reference's order; operations between two analyses fused
    const xeve_hip_tree_params *p;
    bool                        inter, side;
    const unsigned char        *side_of = nullptr; // TreeLayout::side_of: the stream of a side node's level
    std::vector<Item>           items;
    std::vector<int>            pending; // side nodes entered whose analyses are not yet queued (innermost last)
    OpList                      cur;
    int                         wait_next = -1; // the next main-stream entry starts with this wait
    void flush(int k, int log2, int rec = -1)
    {
        Item it;
        it.st = 0, it.ops = cur, it.kind = k, it.size = log2, it.wait_ev = wait_next, it.rec_ev = rec;
        items.push_back(it), cur.n = 0, wait_next = -1;
    }
    void add(int op, int L, int part)
    {
        if(cur.n == MAX_OPS) flush(AN_NONE, 0);
        cur.op[cur.n] = (unsigned char)op, cur.lvl[cur.n] = (unsigned char)L, cur.part[cur.n] = (signed char)part, cur.n++;
    }
    // The side stream's work of the nodes entered so far, INNERMOST FIRST: the main stream needs the innermost node's verdict soonest (after four children of the next
    // level down), the outer ones' only when whole subtrees are through.  Called before the main stream's next analysis is queued.
    void queue_pending()
    {
        for(; !pending.empty(); pending.pop_back()) {
            const int L = pending.back();
            Item it;
            it.st = side_of[L], it.ops.n = 0, it.kind = inter ? AN_INTER : AN_INTRA, it.size = L + 2, it.wait_ev = ev_enter(L), it.rec_ev = -1;
            items.push_back(it);
            if(inter) {
                it.ops.n = 1, it.ops.op[0] = OP_MID, it.ops.lvl[0] = (unsigned char)L, it.ops.part[0] = 0, it.kind = AN_INTRA, it.wait_ev = -1;
                items.push_back(it);
            }
            it.ops.n = 1, it.ops.op[0] = OP_LEAF_SIDE, it.ops.lvl[0] = (unsigned char)L, it.ops.part[0] = 0, it.kind = AN_NONE, it.size = 0, it.wait_ev = -1, it.rec_ev = ev_done(L);
            items.push_back(it);
        }
    }
    void node(int L, int part)
    {
        const int  cu = 1 << (L + 2);
        const bool has = level_has_cu(p, L + 2), kids = level_has_kids(p, L + 2);
        add(OP_ENTER, L, part);
        if(side && has && kids) { // analyses on the side stream, the children on this one, joined in front of op_exit
            add(OP_SPLIT_PREP, L, 0);
            flush(AN_NONE, 0, ev_enter(L));
            pending.push_back(L);
            for(int q = 0; q < 4; q++) {
                node(L - 1, q);
                add(OP_CHILD_DONE, L, q);
            }
            queue_pending(); // (nothing is left by now: the first node without children below queued it)
            if(cur.n) flush(AN_NONE, 0);
            wait_next = ev_done(L);
            add(OP_EXIT, L, 0);
            return;
        }
        if(has) {
            queue_pending();
            if(inter) {
                flush(AN_INTER, L + 2);
                add(OP_MID, L, 0);
            }
            flush(AN_INTRA, L + 2);
        }
        add(OP_LEAF, L, 0);
        if(kids)
            for(int q = 0; q < 4; q++) {
                node(L - 1, q);
                add(OP_CHILD_DONE, L, q);
            }
        add(OP_EXIT, L, 0);
        (void)cu;
    }
};
// the side stream and the events of the fork / join, per host thread (an encoder walks from one thread; the bench's batches have a thread each)
struct TreeSide {
    uint32_t    gen = 0;
    hipStream_t st[XT_STREAMS] = {}; // [1 ..]: the side stream of level 1 ..
    hipEvent_t  ev[10] = {};
    void drop()
    {
        for(auto &q : st)
            if(q) (void)hipStreamSynchronize(q), (void)hipStreamDestroy(q), q = nullptr;
        for(auto &e : ev)
            if(e) (void)hipEventDestroy(e), e = nullptr;
    }
    bool ready()
    {
        if(gen != xh_generation()) drop(), gen = xh_generation();
        if(st[1]) return true;
        // (stream priorities -- the side stream at the device's lowest, the caller's at its highest -- change nothing: measured, profiles/r06_side_stream.md)
        for(int a = 1; a < XT_STREAMS; a++)
            if(hipStreamCreateWithFlags(&st[a], hipStreamNonBlocking) != hipSuccess) { st[a] = nullptr, drop(); return false; }
        for(auto &e : ev)
            if(hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { drop(); return false; }
        return true;
    }
    ~TreeSide() { drop(); }
};
} // namespace

extern "C" int xeve_hip_mode_analyze_ctu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                              uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems,
                                              const xeve_hip_sbac *states, int nstates, const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I,
                                              const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost,
                                              void *workspace, size_t workspace_bytes, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(org && mod && map_scu && map_ipm && map_tidx && map_cu_mode && states && nstates > 0 && jobs && nchains >= 0 && out && next_best && cost && workspace);
    XH_REQUIRE(tree_params_ok(p));
    XH_REQUIRE(org[0] && mod[0] && (p->ip.chroma_format_idc == 0 || (org[1] && org[2] && mod[1] && mod[2])));
    if(p->ip.slice_type == 2) I = nullptr;
    else XH_REQUIRE(tree_inter_ok(p, I));
    // P / B chains of several pictures: the inter analysis addresses the batch as one tall picture (xh_common.h), so the pictures must be stacked vertically -- every
    // plane and map of picture p exactly p * vh luma rows below picture 0's, vh a multiple of 64 that covers the padded reference pictures
    int vh = 0;
    if(I && pic_elems) {
        const int ws_ = p->ip.chroma_format_idc <= 2, hs_ = p->ip.chroma_format_idc <= 1;
        XH_REQUIRE(s_org_l > 0 && pic_elems[0] > 0 && pic_elems[0] % s_org_l == 0);
        vh = (int)(pic_elems[0] / s_org_l);
        XH_REQUIRE(vh % 64 == 0 && vh >= p->pic_h && pic_elems[2] == (int64_t)vh * s_mod_l && pic_elems[4] == (int64_t)(vh >> 2) * p->ip.w_scu);
        XH_REQUIRE(s_mod_l == I->s_ref_l && s_mod_c == I->s_ref_c);
        if(p->ip.chroma_format_idc) XH_REQUIRE(pic_elems[1] == (int64_t)(vh >> hs_) * s_org_c && pic_elems[3] == (int64_t)(vh >> hs_) * s_mod_c);
        (void)ws_;
    }
    if(nchains == 0) return XEVE_HIP_OK;
    // which walk: by the WIDTH OF THE BATCH the call belongs to (the caller's state records: every chain of every GOP), not by the chains this step carries -- the ramp
    // steps of a wide batch's pictures stay on the composed walk (a long-running fused launch between the other streams' short kernels cost 7 %, profiles/r04_walks.md)
    // rdo_dbk_switch and 4x4 inter CUs (presets slow, placebo) live in the fused walk alone (walk_dbk.h, walk_inter.h): the composed walk's stages do neither
    XH_REQUIRE(!xh_walk_only(p) || xh_walk_supported(p, I, std::max(nchains, nstates)));
    if(xh_walk_supported(p, I, std::max(nchains, nstates))) // the fused walk: the whole schedule inside one kernel (walk.hip)
        return xh_walk_run(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, pic_elems, states, p, I, jobs, nchains, out, next_best, cost,
                           workspace, workspace_bytes, vh, (hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    // the side stream: unless switched off, the caller is capturing its stream into a graph, or the walk replays from a graph of its own
    static const int use_graph = getenv("XEVE_HIP_TREE_GRAPH") ? atoi(getenv("XEVE_HIP_TREE_GRAPH")) : 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if(st) (void)hipStreamIsCapturing(st, &cap);
    static thread_local TreeSide side_res;
    const int side = !use_graph && cap == hipStreamCaptureStatusNone && side_res.ready() ? g_tree_side.load(std::memory_order_relaxed) : 0;
    const TreeLayout L = tree_layout(nchains, p, I, s_org_l, s_org_c, side);
    XH_REQUIRE(workspace_bytes >= L.total);
    char       *W = (char *)workspace;
    const int   idc = p->ip.chroma_format_idc;
    TreeK K;
    memset(&K, 0, sizeof(K));
    K.nchains = nchains, K.log2_ctu = p->log2_ctu, K.pic_w = p->pic_w, K.pic_h = p->pic_h, K.w_scu = p->ip.w_scu, K.h_scu = p->ip.h_scu, K.max_cu = p->max_cu;
    K.min_cu = p->min_cu, K.min_cuwh = p->min_cuwh, K.idc = idc, K.ws = idc <= 2, K.hs = idc <= 1, K.slice_qp = p->slice_qp, K.slice_num = p->slice_num;
    K.s_mod_l = s_mod_l, K.s_mod_c = s_mod_c, K.mod_pic_l = pic_elems ? pic_elems[2] : 0, K.mod_pic_c = pic_elems ? pic_elems[3] : 0, K.map_pic = pic_elems ? pic_elems[4] : 0;
    K.lambda0 = p->ip.lambda[0];
    K.mod[0] = mod[0], K.mod[1] = mod[1], K.mod[2] = mod[2], K.map_scu = map_scu, K.map_cu_mode = map_cu_mode, K.map_ipm = map_ipm, K.states = states, K.jobs = jobs;
    K.out = out, K.out_next = next_best, K.out_cost = cost;
    K.node = (Node *)(W + L.node), K.curr = (SbacState *)(W + L.curr), K.next = (SbacState *)(W + L.next), K.before = (SbacState *)(W + L.before);
    K.tdepth = (SbacState *)(W + L.tdepth), K.csplit = (SbacState *)(W + L.csplit), K.best = (CtuData *)(W + L.best), K.temp = (CtuData *)(W + L.temp);
    K.tsplit = (CtuData *)(W + L.tsplit);
    memcpy(K.side_of, L.side_of, sizeof(K.side_of));
    if(I) K.inter = 1, K.ecu_depth = I->ecu_depth, K.s_org_l = s_org_l, K.vh = vh, K.map_mv = (int16_t(*)[2][2])I->map_mv, K.map_refi = (int8_t(*)[2])I->map_refi;
    for(int a = 0; a < XT_STREAMS; a++) {
        TreeK::Ac &A = K.ac[a];
        const TreeLayout::Ac &LA = L.ac[a];
        A.sbest = (SbacState *)(W + LA.sbest);
        A.ijobs = (xeve_hip_intra_job *)(W + LA.ijobs), A.ires = (xeve_hip_intra_result *)(W + LA.ires), A.icoef = (int16_t *)(W + LA.icoef), A.irec = (pel *)(W + LA.irec);
        if(I) {
            A.ejobs = (xeve_hip_inter_job *)(W + LA.ejobs), A.sjobs = (xeve_hip_job *)(W + LA.sjobs), A.eres = (xeve_hip_inter_result *)(W + LA.eres), A.ecoef = (int16_t *)(W + LA.ecoef);
            for(int c = 0; c < 3; c++) A.erec[c] = (pel *)(W + LA.erec[c]);
            A.esatd = (int32_t *)(W + LA.esatd), A.enext = (SbacState *)(W + LA.enext);
        }
    }
    Walk wk;
    wk.p = p, wk.inter = I != nullptr, wk.side = side != 0, wk.side_of = L.side_of, wk.cur.n = 0;
    wk.node(p->log2_ctu - 2, -1);
    wk.add(OP_ROOT_DONE, p->log2_ctu - 2, 0);
    wk.flush(AN_NONE, 0);
    const pel *const modc[3] = {mod[0], mod[1], mod[2]};
    // XEVE_HIP_TREE_LANE=1: the 4x4 / 8x8 nodes decided by one lane per chain (cu_lane.h, k_intra_lane) instead of the batched composite.  MEASURED
    // (profiles/r02_tree_lane.log): 3 launches per node instead of 29 and 3.5 ms of host time per CTU step instead of 56, but the serial lane code runs ~ 560 us per
    // node (a single wave issues one dependent instruction every ~ 9 cycles and the chains of a wave diverge): 179 ms per CTU step for one chain against 68 ms, 583 ms
    // against 217 ms for 8192 chains.  It only pays with >= 8 waves per SIMD, i.e. >= 65 000 chains in flight.  OFF by default; kept because it is the bit-exact,
    // CPU-tested (tests/test_cu_lane.py) starting point of a wave-per-chain node kernel.
    static const int use_lane = getenv("XEVE_HIP_TREE_LANE") ? atoi(getenv("XEVE_HIP_TREE_LANE")) : 0;
    XH_REQUIRE(xh_entropy_table() != nullptr);
    auto enqueue = [&]() -> int { // the whole walk: on `st`, and on the side stream between the events
        XH_HIP(hipMemsetAsync(W + L.zero_from, 0, L.zero_bytes, st)); // the walk's own state starts from zero (a node the picture cuts leaves its outside part untouched)
        hipStream_t sts[XT_STREAMS] = {st, st, st, st, st};
        if(side)
            for(int a = 1; a < XT_STREAMS; a++) sts[a] = side_res.st[a];
        bool recorded[10] = {};
        for(const Item &it : wk.items) {
            hipStream_t s = sts[it.st];
            void *sv = (void *)s;
            if(it.wait_ev >= 0) {
                XH_REQUIRE(side && recorded[it.wait_ev]); // (a wait on an event not yet recorded would not wait)
                XH_HIP(hipStreamWaitEvent(s, side_res.ev[it.wait_ev], 0));
            }
            if(it.ops.n) {
                // one wave per chain where every operation of the list moves a CU of at most 16x16 (256 luma samples, 16 units): the operations are a dozen phases between
                // barriers with a global-memory round trip each, so a launch lasts (phases x latency) x (blocks / blocks resident at once) -- 5344 blocks of four waves
                // are 2.6 rounds on 256 CUs, of one wave a single round (224 -> see profiles/r06_tree_ops.md); the large levels keep four waves for their 4096-sample copies
                int top = 0;
                for(int i = 0; i < it.ops.n; i++) top = std::max(top, (int)it.ops.lvl[i]);
                k_tree_ops<<<nchains, top <= 2 ? 64 : 256, 0, s>>>(K, it.ops);
            }
            const int log2 = it.size, cu = 1 << log2;
            const TreeK::Ac       &A = K.ac[it.st];
            const TreeLayout::Ac &LA = L.ac[it.st];
            int rc = XEVE_HIP_OK;
            // core->rdoq_est_* of the node's entry states (xeve_mode.c:792): once per node, for its inter analysis' two pinter_residue_rdo batches and its intra analysis
            // (round 6: each of the three made its own)
            auto *est = (xeve_hip_rdoq_est_full *)(W + LA.est);
            if(it.kind == AN_INTER || (it.kind == AN_INTRA && !I && !(log2 <= 3 && use_lane))) {
                rc = xeve_hip_rdoq_bit_est(K.curr + (size_t)(log2 - 2) * nchains, nchains, est, sv);
                if(rc != XEVE_HIP_OK) return rc;
            }
            if(it.kind == AN_INTRA && log2 <= 3 && use_lane) { // one lane per chain decides the node (cu_lane.h)
                const xl::Params LP = lane_params(p, log2, s_org_l, s_org_c, s_mod_l, s_mod_c);
                const int grid = nchains;
                if(log2 == 2) k_intra_lane<2><<<grid, 64, 0, s>>>(K, LP, org[0], org[1], org[2], map_tidx, pic_elems ? pic_elems[0] : 0, pic_elems ? pic_elems[1] : 0);
                else k_intra_lane<3><<<grid, 64, 0, s>>>(K, LP, org[0], org[1], org[2], map_tidx, pic_elems ? pic_elems[0] : 0, pic_elems ? pic_elems[1] : 0);
            }
            else if(it.kind == AN_INTRA) {
                const xeve_hip_intra_params ip = level_params(p, log2);
                rc = xh_pintra_analyze_cu_jobs_x(org, s_org_l, s_org_c, modc, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, pic_elems, K.curr + (size_t)(log2 - 2) * nchains,
                                                 nchains, &ip, A.ijobs + (size_t)(log2 - 2) * nchains, nchains, (xeve_hip_intra_result *)(W + LA.ires), (int16_t *)(W + LA.icoef), (pel *)(W + LA.irec), A.sbest,
                                                 W + LA.iws, LA.iws_bytes, sv, est);
            }
            else if(it.kind == AN_INTER) {
                const xeve_hip_inter_params ep = level_inter_params(I, log2);
                XhVhScope tall(vh);
                // (the CUs' merge / MVP candidates from the per-unit maps -- xeve_hip_inter_candidates -- by the analysis' first kernel)
                const XhInterCand C = {map_scu, map_tidx, I->map_mv, I->col_mv0, I->col_mv1 ? I->col_mv1 : I->col_mv0, p->ip.w_scu, 1 << (log2 - 2), 1 << (log2 - 2),
                                       p->ip.slice_type == 0, vh};
                rc = xh_pinter_analyze_cu_jobs_x(org, s_org_l, s_org_c, I->refp, I->s_ref_l, I->s_ref_c, K.curr + (size_t)(log2 - 2) * nchains, nchains, &ep, A.ejobs + (size_t)(log2 - 2) * nchains,
                                                 nchains, I->coef_l, I->coef_c, (xeve_hip_inter_result *)(W + LA.eres), (int16_t *)(W + LA.ecoef), (pel *)(W + LA.erec[0]),
                                                 (pel *)(W + LA.erec[1]), (pel *)(W + LA.erec[2]), (pel *)(W + LA.epred), (xeve_hip_sbac *)(W + LA.enext), W + LA.ews,
                                                 LA.ews_bytes, sv, &C, est);
                if(rc == XEVE_HIP_OK) // core->inter_satd = xeve_satd_16b(original, mi->pred_y_best) (mode_check_intra, :1250-1262)
                    rc = xeve_hip_satd_jobs(org[0], s_org_l, (const pel *)(W + LA.epred), cu, A.sjobs + (size_t)(log2 - 2) * nchains, nchains, (const int32_t *)(W + L.zero32), 1, cu, cu, p->ip.bit_depth,
                                            (int32_t *)(W + LA.esatd), sv);
            }
            if(rc != XEVE_HIP_OK) return rc;
            if(it.rec_ev >= 0) {
                XH_REQUIRE(side);
                XH_HIP(hipEventRecord(side_res.ev[it.rec_ev], s));
                recorded[it.rec_ev] = true;
            }
        }
        return XEVE_HIP_OK;
    };
    // The schedule is static (10 000 launches per I-picture CTU, 15 600 per P / B CTU) and every operand sits at an address the caller chose, so a call whose
    // arguments repeat can be captured into a HIP graph and replayed.  MEASURED (profiles/r02_tree_graph.log): the replay frees the host -- 2.8 ms instead of
    // 56 .. 100 ms of launch calls per CTU step -- but the GPU runs the same step 8 .. 10 ms SLOWER (dependent kernel nodes of a graph dispatch no faster than
    // stream launches here), so it is OFF unless XEVE_HIP_TREE_GRAPH=1: for a caller that needs its host thread, not for speed.
    if(use_graph && st && cap == hipStreamCaptureStatusNone && !xh_prof_on(XH_PROF_SEARCH) && !xh_prof_on(XH_PROF_CU_BITS)) {
        static thread_local TreeGraphs G;
        std::vector<char> key;
        auto put = [&](const void *a, size_t n) { key.insert(key.end(), (const char *)a, (const char *)a + n); };
        const void *ptrs[] = {org[0], org[1], org[2], mod[0], mod[1], mod[2], map_scu, map_ipm, map_tidx, map_cu_mode, states, jobs, out, next_best, cost, workspace, stream};
        const long  ints[] = {s_org_l, s_org_c, s_mod_l, s_mod_c, nstates, nchains, (long)workspace_bytes, pic_elems ? 1 : 0};
        put(ptrs, sizeof(ptrs)), put(ints, sizeof(ints)), put(p, sizeof(*p));
        if(pic_elems) put(pic_elems, 5 * sizeof(int64_t));
        if(I) put(I, sizeof(*I)), put(I->refp, sizeof(xeve_hip_refpic) * 2 * (size_t)std::max(I->ipar.rdo.num_refp[0], I->ipar.rdo.num_refp[1]));
        TreeGraph *g = G.find(key);
        if(g && g->exec) {
            XH_HIP(hipGraphLaunch(g->exec, st));
            return XEVE_HIP_OK;
        }
        if(g && ++g->seen >= 2) {
            XH_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int        rc = enqueue();
            const hipError_t ec = hipStreamEndCapture(st, &g->graph);
            if(rc != XEVE_HIP_OK) return rc;
            XH_HIP(ec);
            XH_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
            XH_HIP(hipGraphLaunch(g->exec, st));
            return XEVE_HIP_OK;
        }
        if(!g) G.add(key);
    }
    const int rc = enqueue();
    if(rc != XEVE_HIP_OK) return rc;
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_mode_analyze_ctu_intra_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                                    uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems,
                                                    const xeve_hip_sbac *states, int nstates, const xeve_hip_tree_params *p, const xeve_hip_ctu_job *jobs,
                                                    int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, void *workspace,
                                                    size_t workspace_bytes, void *stream)
{
    XH_REQUIRE(p && p->ip.slice_type == 2);
    return xeve_hip_mode_analyze_ctu_jobs(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, pic_elems, states, nstates, p, nullptr, jobs,
                                          nchains, out, next_best, cost, workspace, workspace_bytes, stream);
}

// ---- host-memory form of ONE call of ctx->fn_mode_analyze_lcu in an I slice (stage, launch, synchronise: one exchange per CTU) --------------------------------
// Every pointer is HOST memory: org / mod = sample (0, 0) of the original picture's planes and of the picture being reconstructed, the maps = ctx->map_scu / map_ipm /
// map_tidx / map_cu_mode.  What the walk reads is moved as a small LOCAL PICTURE: the CTU, one unit to its left and above (none on the picture's edge), and the
// CTU's width again to the right (the samples up-right of its CUs); the picture's own right / bottom edge stays an edge, so a CTU the picture cuts is cut the
// same way.  The CTU's part of the reconstruction and of the maps is written back.
extern "C" int xeve_hip_mode_analyze_ctu_intra_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                                    uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xeve_hip_sbac *entry,
                                                    const xeve_hip_tree_params *p, int x0, int y0, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost)
{
    XH_ENTER();
    XH_REQUIRE(org && mod && map_scu && map_ipm && map_tidx && map_cu_mode && entry && out && next_best && cost && tree_params_ok(p));
    const int idc = p->ip.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, ncomp = idc ? 3 : 1, ctu = 1 << p->log2_ctu, n = ctu >> 2;
    XH_REQUIRE(org[0] && mod[0] && (!idc || (org[1] && org[2] && mod[1] && mod[2])));
    XH_REQUIRE(x0 >= 0 && y0 >= 0 && x0 < p->pic_w && y0 < p->pic_h && (x0 & (ctu - 1)) == 0 && (y0 & (ctu - 1)) == 0);
    const int x_scu = x0 >> 2, y_scu = y0 >> 2, lx = x_scu > 0, ly = y_scu > 0, nw = std::min(2 * n, p->ip.w_scu - x_scu), nh = std::min(n, p->ip.h_scu - y_scu);
    const int Wl = lx + nw, Hl = ly + nh, cw = std::min(n, p->ip.w_scu - x_scu); // cw x nh units: the CTU's part inside the picture
    const size_t nmap = (size_t)Wl * Hl;
    const int    pw[3] = {Wl * 4, Wl * (4 >> ws), Wl * (4 >> ws)}, ph[3] = {Hl * 4, Hl * (4 >> hs), Hl * (4 >> hs)};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 63) & ~(size_t)63; return at; };
    xeve_hip_ctu_job jl;
    jl.x = 4 * lx, jl.y = 4 * ly, jl.sbac = 0, jl.pic = 0;
    const size_t o_job = take(sizeof(jl)), o_st = take(sizeof(*entry)), o_tidx = take(nmap);
    size_t o_org[3] = {0, 0, 0};
    for(int c = 0; c < ncomp; c++) o_org[c] = take((size_t)pw[c] * ph[c] * 2);
    const size_t o_scu = take(nmap * 4), o_ipm = take(nmap), o_cum = take(nmap * 4); // from here on the buffers come back too
    size_t o_mod[3] = {0, 0, 0};
    for(int c = 0; c < ncomp; c++) o_mod[c] = take((size_t)pw[c] * ph[c] * 2);
    const size_t in_bytes = o;
    const size_t o_out = take(sizeof(*out)), o_next = take(sizeof(*next_best)), o_cost = take(sizeof(double));
    const size_t io_bytes = o;
    xeve_hip_tree_params pl = *p;
    pl.ip.w_scu = Wl, pl.ip.h_scu = Hl, pl.pic_w = Wl * 4, pl.pic_h = Hl * 4;
    const size_t wsb = xeve_hip_mode_analyze_ctu_intra_workspace(1, &pl);
    XH_REQUIRE(wsb > 0);
    static thread_local XhHostArena C;
    int rc = C.ensure(io_bytes, wsb);
    if(rc != XEVE_HIP_OK) return rc;
    char *H = C.pin, *D = C.dev;
    memcpy(H + o_job, &jl, sizeof(jl)), memcpy(H + o_st, entry, sizeof(*entry));
    const int gu0 = (y_scu - ly) * p->ip.w_scu + x_scu - lx; // the window's first unit in the picture's maps
    for(int j = 0; j < Hl; j++) {
        const size_t g = (size_t)gu0 + (size_t)j * p->ip.w_scu, l = (size_t)j * Wl;
        memcpy(H + o_scu + 4 * l, map_scu + g, 4 * (size_t)Wl), memcpy(H + o_cum + 4 * l, map_cu_mode + g, 4 * (size_t)Wl);
        memcpy(H + o_ipm + l, map_ipm + g, Wl), memcpy(H + o_tidx + l, map_tidx + g, Wl);
    }
    for(int c = 0; c < ncomp; c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, so = c ? s_org_c : s_org_l, sm = c ? s_mod_c : s_mod_l;
        const int xw = (x0 >> sx) - lx * (4 >> sx), yw = (y0 >> sy) - ly * (4 >> sy); // the window's first sample in the plane
        pel *lo = (pel *)(H + o_org[c]), *lm = (pel *)(H + o_mod[c]);
        for(int r = 0; r < ph[c]; r++) {
            memcpy(lo + (size_t)r * pw[c], org[c] + (size_t)(yw + r) * so + xw, sizeof(pel) * pw[c]);
            memcpy(lm + (size_t)r * pw[c], mod[c] + (size_t)(yw + r) * sm + xw, sizeof(pel) * pw[c]);
        }
    }
    XH_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, C.st));
    const pel *d_org[3] = {(const pel *)(D + o_org[0]), idc ? (const pel *)(D + o_org[1]) : nullptr, idc ? (const pel *)(D + o_org[2]) : nullptr};
    pel       *d_mod[3] = {(pel *)(D + o_mod[0]), idc ? (pel *)(D + o_mod[1]) : nullptr, idc ? (pel *)(D + o_mod[2]) : nullptr};
    char *d_ws = D + ((io_bytes + 255) & ~(size_t)255);
    rc = xeve_hip_mode_analyze_ctu_intra_jobs(d_org, pw[0], pw[1], d_mod, pw[0], pw[1], (uint32_t *)(D + o_scu), (int8_t *)(D + o_ipm), (const uint8_t *)(D + o_tidx),
                                              (uint32_t *)(D + o_cum), nullptr, (const xeve_hip_sbac *)(D + o_st), 1, &pl, (const xeve_hip_ctu_job *)(D + o_job), 1,
                                              (xeve_hip_ctu_data *)(D + o_out), (xeve_hip_sbac *)(D + o_next), (double *)(D + o_cost), d_ws,
                                              C.dev_bytes - (size_t)(d_ws - D), C.st);
    if(rc != XEVE_HIP_OK) return rc;
    XH_HIP(hipMemcpyAsync(H + o_scu, D + o_scu, io_bytes - o_scu, hipMemcpyDeviceToHost, C.st));
    XH_HIP(hipStreamSynchronize(C.st));
    memcpy(out, H + o_out, sizeof(*out)), memcpy(next_best, H + o_next, sizeof(*next_best)), memcpy(cost, H + o_cost, sizeof(double));
    for(int j = 0; j < nh; j++) { // the CTU's units
        const size_t g = (size_t)(y_scu + j) * p->ip.w_scu + x_scu, l = (size_t)(ly + j) * Wl + lx;
        memcpy(map_scu + g, H + o_scu + 4 * l, 4 * (size_t)cw), memcpy(map_cu_mode + g, H + o_cum + 4 * l, 4 * (size_t)cw), memcpy(map_ipm + g, H + o_ipm + l, cw);
    }
    for(int c = 0; c < ncomp; c++) { // the CTU's samples
        const int sx = c ? ws : 0, sy = c ? hs : 0, sm = c ? s_mod_c : s_mod_l, bw = (cw * 4) >> sx, bh = (nh * 4) >> sy, xl = lx * (4 >> sx), yl = ly * (4 >> sy);
        const pel *lm = (const pel *)(H + o_mod[c]);
        for(int r = 0; r < bh; r++) memcpy(mod[c] + (size_t)((y0 >> sy) + r) * sm + (x0 >> sx), lm + (size_t)(yl + r) * pw[c] + xl, sizeof(pel) * bw);
    }
    return XEVE_HIP_OK;
}

// ---- host-memory form for every slice type (P / B: needs resident pictures, xeve_hip_picture_begin) -------------------------------------------------------------
// The inter analysis addresses the reference pictures with the CU's position in the picture, so a P / B CTU is walked in PICTURE coordinates: the original, the
// reference pictures, the collocated motion maps and the tile map are resident copies (one upload per picture); the picture being reconstructed and the maps the
// walk updates live in per-thread device buffers of picture size, of which only the CTU's neighbourhood is current -- it is uploaded before the walk (the CTU, one
// unit to its left and above, the CTU's width again to the right) and the CTU's own part is read back after it.
namespace {
struct TreePicCtx {
    uint32_t    gen = 0;
    hipStream_t st  = nullptr;
    char       *buf[12] = {};
    size_t      cap[12] = {};
    void release()
    {
        if(st) (void)hipStreamDestroy(st);
        for(int i = 0; i < 12; i++) {
            if(buf[i]) (void)hipFree(buf[i]);
            buf[i] = nullptr, cap[i] = 0;
        }
        st = nullptr;
    }
    int ensure(int i, size_t bytes)
    {
        if(cap[i] >= bytes) return XEVE_HIP_OK;
        if(buf[i]) {
            XH_HIP(hipStreamSynchronize(st));
            (void)hipFree(buf[i]);
            buf[i] = nullptr, cap[i] = 0;
        }
        XH_HIP(hipMalloc((void **)&buf[i], bytes + (bytes >> 3)));
        XH_HIP(hipMemsetAsync(buf[i], 0, bytes + (bytes >> 3), st));
        cap[i] = bytes + (bytes >> 3);
        return XEVE_HIP_OK;
    }
    int begin()
    {
        if(gen != xh_generation()) release(), gen = xh_generation();
        if(!st) XH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return XEVE_HIP_OK;
    }
    ~TreePicCtx() { release(); }
};
enum { B_MOD0 = 0, B_MOD1 = 1, B_MOD2 = 2, B_SCU = 3, B_IPM = 4, B_CUM = 5, B_MV = 6, B_REFI = 7, B_IO = 8, B_WS = 9 };
} // namespace

extern "C" int xeve_hip_mode_analyze_ctu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                              uint32_t *map_scu, int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xeve_hip_sbac *entry,
                                              const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int pad_l, int pad_c, int x0, int y0,
                                              xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost)
{
    XH_ENTER();
    XH_REQUIRE(p);
    if(p->ip.slice_type == 2)
        return xeve_hip_mode_analyze_ctu_intra_host(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, entry, p, x0, y0, out, next_best, cost);
    XH_REQUIRE(org && mod && map_scu && map_ipm && map_tidx && map_cu_mode && entry && out && next_best && cost && tree_params_ok(p) && tree_inter_ok(p, I) && pad_l >= 0 && pad_c >= 0);
    if(!xh_resident_on()) {
        xh_set_error("xeve_hip_mode_analyze_ctu_host: a P / B CTU needs resident pictures (announce each picture with xeve_hip_picture_begin)");
        return XEVE_HIP_ERR_ARG;
    }
    const int idc = p->ip.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, ncomp = idc ? 3 : 1, ctu = 1 << p->log2_ctu, n = ctu >> 2, isb = p->ip.slice_type == 0;
    XH_REQUIRE(org[0] && mod[0] && (!idc || (org[1] && org[2] && mod[1] && mod[2])));
    XH_REQUIRE(x0 >= 0 && y0 >= 0 && x0 < p->pic_w && y0 < p->pic_h && (x0 & (ctu - 1)) == 0 && (y0 & (ctu - 1)) == 0);
    const int w_scu = p->ip.w_scu, h_scu = p->ip.h_scu, nscu = w_scu * h_scu, hc = p->pic_h >> hs;
    const int x_scu = x0 >> 2, y_scu = y0 >> 2, lx = x_scu > 0, ly = y_scu > 0, nw = std::min(2 * n, w_scu - x_scu), nh = std::min(n, h_scu - y_scu), cw = std::min(n, w_scu - x_scu);
    static thread_local TreePicCtx C;
    int rc = C.begin();
    if(rc != XEVE_HIP_OK) return rc;
    // resident: the original, the reference pictures of both lists, the collocated maps, the tile map
    const size_t eo[3] = {(size_t)s_org_l * p->pic_h, (size_t)s_org_c * hc, (size_t)s_org_c * hc};
    const size_t er[3] = {(size_t)I->s_ref_l * (p->pic_h + 2 * pad_l), (size_t)I->s_ref_c * (hc + 2 * pad_c), (size_t)I->s_ref_c * (hc + 2 * pad_c)};
    const size_t orr[3] = {(size_t)pad_l * I->s_ref_l + pad_l, (size_t)pad_c * I->s_ref_c + pad_c, (size_t)pad_c * I->s_ref_c + pad_c};
    const pel *dorg[3] = {nullptr, nullptr, nullptr};
    for(int c = 0; c < ncomp; c++)
        if(!(dorg[c] = (const pel *)xh_resident(org[c], eo[c] * sizeof(pel)))) return XEVE_HIP_ERR_DEVICE;
    xeve_hip_refpic tab[2 * XEVE_HIP_MAX_REFP];
    memset(tab, 0, sizeof(tab));
    const int nr[2] = {I->ipar.rdo.num_refp[0], isb ? I->ipar.rdo.num_refp[1] : 0};
    XH_REQUIRE(nr[0] >= 1 && nr[0] <= XEVE_HIP_MAX_REFP && nr[1] <= XEVE_HIP_MAX_REFP);
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            const xeve_hip_refpic &e = I->refp[r * 2 + l];
            XH_REQUIRE(e.y && (idc == 0 || (e.u && e.v)));
            const xeve_hip_pel *hp[3] = {e.y, e.u, e.v};
            const pel *dp[3] = {nullptr, nullptr, nullptr};
            for(int c = 0; c < ncomp; c++) {
                const pel *d = (const pel *)xh_resident(hp[c] - orr[c], er[c] * sizeof(pel));
                if(!d) return XEVE_HIP_ERR_DEVICE;
                dp[c] = d + orr[c];
            }
            tab[r * 2 + l].y = dp[0], tab[r * 2 + l].u = dp[1], tab[r * 2 + l].v = dp[2], tab[r * 2 + l].poc = e.poc;
        }
    if(!isb) tab[0 * 2 + 1] = tab[0 * 2 + 0]; // (P slices never read list 1; keep the table addressable)
    const int16_t *dcol0 = (const int16_t *)xh_resident(I->col_mv0, (size_t)nscu * 8), *dcol1 = isb ? (const int16_t *)xh_resident(I->col_mv1, (size_t)nscu * 8) : dcol0;
    const uint8_t *dtidx = (const uint8_t *)xh_resident(map_tidx, (size_t)nscu);
    if(!dcol0 || !dcol1 || !dtidx) return XEVE_HIP_ERR_DEVICE;
    // per-thread picture-sized buffers + the call's small records + the workspace
    xeve_hip_tree_inter Id = *I;
    const size_t wsb = xeve_hip_mode_analyze_ctu_workspace(1, p, I, s_org_l, s_org_c);
    XH_REQUIRE(wsb > 0);
    const size_t o_job = 0, o_st = 256, o_out = 512, o_next = o_out + al(sizeof(*out)), o_cost = o_next + 256, io_bytes = o_cost + 256;
    const size_t need[10] = {(size_t)s_mod_l * p->pic_h * 2, (size_t)s_mod_c * hc * 2, (size_t)s_mod_c * hc * 2, (size_t)nscu * 4, (size_t)nscu, (size_t)nscu * 4, (size_t)nscu * 8,
                             (size_t)nscu * 2, io_bytes, wsb};
    for(int i = 0; i < 10; i++)
        if((i == B_MOD1 || i == B_MOD2) && !idc) continue;
        else if((rc = C.ensure(i, need[i])) != XEVE_HIP_OK) return rc;
    pel *dmod[3] = {(pel *)C.buf[B_MOD0], (pel *)C.buf[B_MOD1], (pel *)C.buf[B_MOD2]};
    Id.refp = tab, Id.map_mv = (int16_t *)C.buf[B_MV], Id.map_refi = (int8_t *)C.buf[B_REFI], Id.col_mv0 = dcol0, Id.col_mv1 = dcol1;
    // the neighbourhood in: rows [y_scu - ly, y_scu + nh) x columns [x_scu - lx, x_scu + nw) of every map, the same window of the planes
    const size_t u0 = (size_t)(y_scu - ly) * w_scu + (x_scu - lx);
    const int    Wl = lx + nw, Hl = ly + nh;
    auto copy2d = [&](void *dst, const void *src, size_t pitch, size_t width, size_t height, hipMemcpyKind kind) {
        return width && height ? hipMemcpy2DAsync(dst, pitch, src, pitch, width, height, kind, C.st) : hipSuccess;
    };
    XH_HIP(copy2d(C.buf[B_SCU] + u0 * 4, map_scu + u0, (size_t)w_scu * 4, (size_t)Wl * 4, Hl, hipMemcpyHostToDevice));
    if(y_scu + nh < h_scu) { // the units BELOW the CTU and its left margin: the left neighbours of a CU reach as far below it as it is high, i.e. out of the CTU's bottom row
                             // into units that are not coded yet -- their flags must say so here too (the buffers keep what earlier pictures left)
        const size_t b0 = (size_t)(y_scu + nh) * w_scu + (x_scu - lx);
        XH_HIP(copy2d(C.buf[B_SCU] + b0 * 4, map_scu + b0, (size_t)w_scu * 4, (size_t)(lx + cw) * 4, std::min(n, h_scu - (y_scu + nh)), hipMemcpyHostToDevice));
    }
    XH_HIP(copy2d(C.buf[B_CUM] + u0 * 4, map_cu_mode + u0, (size_t)w_scu * 4, (size_t)Wl * 4, Hl, hipMemcpyHostToDevice));
    XH_HIP(copy2d(C.buf[B_IPM] + u0, map_ipm + u0, (size_t)w_scu, (size_t)Wl, Hl, hipMemcpyHostToDevice));
    XH_HIP(copy2d(C.buf[B_MV] + u0 * 8, I->map_mv + u0 * 4, (size_t)w_scu * 8, (size_t)Wl * 8, Hl, hipMemcpyHostToDevice));
    XH_HIP(copy2d(C.buf[B_REFI] + u0 * 2, I->map_refi + u0 * 2, (size_t)w_scu * 2, (size_t)Wl * 2, Hl, hipMemcpyHostToDevice));
    for(int c = 0; c < ncomp; c++) {
        const int    sx = c ? ws : 0, sy = c ? hs : 0, sm = c ? s_mod_c : s_mod_l;
        const size_t s0 = (size_t)((y0 >> sy) - ly * (4 >> sy)) * sm + ((x0 >> sx) - lx * (4 >> sx));
        XH_HIP(copy2d(dmod[c] + s0, mod[c] + s0, (size_t)sm * 2, (size_t)(Wl * (4 >> sx)) * 2, (size_t)Hl * (4 >> sy), hipMemcpyHostToDevice));
    }
    xeve_hip_ctu_job jl;
    jl.x = x0, jl.y = y0, jl.sbac = 0, jl.pic = 0;
    XH_HIP(hipMemcpyAsync(C.buf[B_IO] + o_job, &jl, sizeof(jl), hipMemcpyHostToDevice, C.st));
    XH_HIP(hipMemcpyAsync(C.buf[B_IO] + o_st, entry, sizeof(*entry), hipMemcpyHostToDevice, C.st));
    rc = xeve_hip_mode_analyze_ctu_jobs(dorg, s_org_l, s_org_c, dmod, s_mod_l, s_mod_c, (uint32_t *)C.buf[B_SCU], (int8_t *)C.buf[B_IPM], dtidx, (uint32_t *)C.buf[B_CUM], nullptr,
                                        (const xeve_hip_sbac *)(C.buf[B_IO] + o_st), 1, p, &Id, (const xeve_hip_ctu_job *)(C.buf[B_IO] + o_job), 1,
                                        (xeve_hip_ctu_data *)(C.buf[B_IO] + o_out), (xeve_hip_sbac *)(C.buf[B_IO] + o_next), (double *)(C.buf[B_IO] + o_cost), C.buf[B_WS],
                                        C.cap[B_WS], C.st);
    if(rc != XEVE_HIP_OK) return rc;
    // the CTU's part out
    const size_t c0 = (size_t)y_scu * w_scu + x_scu;
    XH_HIP(hipMemcpyAsync(out, C.buf[B_IO] + o_out, sizeof(*out), hipMemcpyDeviceToHost, C.st));
    XH_HIP(hipMemcpyAsync(next_best, C.buf[B_IO] + o_next, sizeof(*next_best), hipMemcpyDeviceToHost, C.st));
    XH_HIP(hipMemcpyAsync(cost, C.buf[B_IO] + o_cost, sizeof(double), hipMemcpyDeviceToHost, C.st));
    XH_HIP(copy2d(map_scu + c0, C.buf[B_SCU] + c0 * 4, (size_t)w_scu * 4, (size_t)cw * 4, nh, hipMemcpyDeviceToHost));
    XH_HIP(copy2d(map_cu_mode + c0, C.buf[B_CUM] + c0 * 4, (size_t)w_scu * 4, (size_t)cw * 4, nh, hipMemcpyDeviceToHost));
    XH_HIP(copy2d(map_ipm + c0, C.buf[B_IPM] + c0, (size_t)w_scu, (size_t)cw, nh, hipMemcpyDeviceToHost));
    XH_HIP(copy2d(I->map_mv + c0 * 4, C.buf[B_MV] + c0 * 8, (size_t)w_scu * 8, (size_t)cw * 8, nh, hipMemcpyDeviceToHost));
    XH_HIP(copy2d(I->map_refi + c0 * 2, C.buf[B_REFI] + c0 * 2, (size_t)w_scu * 2, (size_t)cw * 2, nh, hipMemcpyDeviceToHost));
    for(int c = 0; c < ncomp; c++) {
        const int    sx = c ? ws : 0, sy = c ? hs : 0, sm = c ? s_mod_c : s_mod_l;
        const size_t s0 = (size_t)(y0 >> sy) * sm + (x0 >> sx);
        XH_HIP(copy2d(mod[c] + s0, dmod[c] + s0, (size_t)sm * 2, (size_t)((cw * 4) >> sx) * 2, (size_t)((nh * 4) >> sy), hipMemcpyDeviceToHost));
    }
    XH_HIP(hipStreamSynchronize(C.st));
    return XEVE_HIP_OK;
}
