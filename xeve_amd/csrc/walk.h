// xeve_amd/csrc/walk.h -- the CTU mode decision (mode_analyze_lcu -> mode_coding_tree -> mode_coding_unit, src_base/xeve_mode.c:1310-1350, 2007-2610) of a GROUP of
// chains executed by ONE TEAM of threads from the first tree node to the last, without leaving the kernel.
//
// Why this form.  The composed walk (tree.hip) is 10 000 .. 15 600 dependent launches per CTU step, every one a load -> compute -> store of a few waves, all chains of
// the batch in lockstep: its wall time is the host's issue time.  What is serial inside a chain cannot be made parallel (a CU's predictors are its neighbours'
// reconstruction, its bit counts start from the coder state its predecessor's winner left, an arithmetic coder is a recurrence per bin), so the form that removes the
// launches without giving up the only parallel axis the coder has -- independent bit-count jobs -- is:
//   * a team = one workgroup, 256 threads; it owns C chains (C <= 8) and walks them in TEAM-LOCAL lockstep through the static quad-tree schedule; where a launch
//     boundary was there is a workgroup barrier, what passed through HBM between two launches passes through the team's slice of the workspace (L2) or LDS;
//   * every stage spreads the union of its chains' work over all lanes: samples / transform outputs / search candidates one item per lane, the serial automata
//     (RDOQ, the CABAC counter) one JOB per lane with the jobs of the C chains side by side, so that a wave of the counter carries C x (candidates) busy lanes
//     instead of the 3 .. 5 a single chain has;
//   * teams are independent of each other: no grid-wide lockstep, a team that is done with its CTUs ends.
// Everything is __host__ __device__ and written against Tm {thread, team size} + sync(): libxeve_hip.so instantiates the device side (walk.hip: k_walk), the test
// harness (tests/native/walk_host.cpp) builds the SAME functions for the host as a team of one thread, where every stage degenerates to a plain loop -- that is how the
// CPU suite holds this file bit for bit against the pinned oracle without a GPU.
//
// Costs are doubles built with the reference's operations in the reference's order (-ffp-contract=off).
#pragma once
#include <stdint.h>
#include <string.h>
#include "../../include/xeve_hip.h"

#if !defined(__HIPCC__) && !defined(__host__) // (a plain C++ compiler builds the host side: the test harnesses under tests/native and oracle/)
#define __host__
#define __device__
#endif
#ifndef XW
#define XW __host__ __device__ static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define XW_DEVICE 1
#else
#define XW_DEVICE 0
#endif

namespace xw {
typedef int16_t            pel;
typedef xeve_hip_sbac      Sbac;
typedef xeve_hip_ctu_data  CtuData;
typedef unsigned long long u64;

#define XW_MAXC 8   // chains per team
#define XW_NT 256   // threads per team on the device
#define XW_CODL 96  // lanes of a team that can run the coder at once (their models live in LDS)
#define XW_MAXR XEVE_HIP_MAX_REFP
#define XW_MAX_COST 1.7e+308
#define XW_NB 136   // one neighbour line: [0] = the corner sample, then up to 2 * 64 samples (+ slack)
#define XW_NCTX XEVE_HIP_SBAC_NCTX

struct Tm {
    int tid, n;
};
#if !XW_DEVICE
// A host team of SEVERAL threads (tests/native/walk_host.cpp, threads > 1: the device's lane mapping with real threads -- the results must not depend on the team's size, and
// under ThreadSanitizer every access two lanes share without a barrier between them shows up): the harness registers the team's barrier with each of its threads; without
// one, sync() is free (a team of one thread).
struct HostTeam {
    void (*barrier)(void *);
    void *arg;
};
inline HostTeam &host_team()
{
    static thread_local HostTeam t = {nullptr, nullptr};
    return t;
}
#endif
XW void sync(const Tm &)
{
#if XW_DEVICE
    __syncthreads();
#else
    HostTeam &t = host_team();
    if(t.barrier) t.barrier(t.arg);
#endif
}
XW void aadd(int *p, int v)
{
#if XW_DEVICE
    atomicAdd(p, v);
#else
    __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
#endif
}
XW void aadd64(u64 *p, u64 v)
{
#if XW_DEVICE
    atomicAdd(p, v);
#else
    __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
#endif
}
XW void aor(int *p, int v)
{
#if XW_DEVICE
    atomicOr(p, v);
#else
    __atomic_fetch_or(p, v, __ATOMIC_RELAXED);
#endif
}
// stage classes of the in-kernel profiler (XEVE_HIP_WALK_PROF=1: thread 0 of team 0 adds the cycles between two marks to its class)
enum { PR_CLEAR = 0, PR_ENTER, PR_LEAF, PR_CHILD, PR_EXIT, PR_ROOT, PR_MID, PR_I_SETUP, PR_I_NBR, PR_I_PRED, PR_I_SATD, PR_I_LIST, PR_I_BITS, PR_I_PICK, PR_I_CPRED, PR_I_FINAL,
       PR_B_DIFF, PR_B_T0, PR_B_T1, PR_B_RDOQ, PR_B_DQ, PR_B_T2, PR_B_T3, PR_B_REC, PR_E_CAND, PR_E_SKIP, PR_E_ME, PR_E_SPEL, PR_E_MC, PR_E_BITS, PR_E_GLUE, PR_E_FINAL, PR_M_BITS, PR_M_SAD, PR_M_SEL, PR_Q_A, PR_Q_B, PR_N };
XW int iabs(int v) { return v < 0 ? -v : v; }
XW unsigned mul24(unsigned a, unsigned b)
{
#if XW_DEVICE
    return __umul24(a, b);
#else
    return a * b;
#endif
}
XW int imin(int a, int b) { return a < b ? a : b; }
XW int imax(int a, int b) { return a > b ? a : b; }
XW int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
XW int ilog2(unsigned v) { return v ? 31 - __builtin_clz(v) : 0; }

// ---- records ---------------------------------------------------------------------------------------------------------------------------------------------------
struct Node { // one per (level, chain)
    int    active, x0, y0, leaf, do_split, best_split, dist_cu, cu_mode, try_intra, pad_;
    double cost_best, cost_temp, unit_cost;
};
// scratch of one transform block: everything the residual chain of a candidate leaves behind
struct Slot {
    int16_t  coef[4096]; // residual -> DCT coefficients -> (after RDOQ) dequantised levels -> inverse transform output
    int32_t  tb[4096];   // between the two passes of a transform
    int16_t  lev[4096];  // quantised levels (what is coded)
    pel      rec[4096];  // reconstruction
    uint32_t ev[4096];   // the levels as (zero run, |level| - 1, sign, at-the-end) events in scan order
    int64_t  rq[2][4096]; // RDOQ: per scan position, cost of the best level minus cost of no level, after a non-zero / after a zero (the run state)
};
struct IntraRes {
    double  cost;
    int32_t dist_cu, nnz[3], pred_cnt, ipm, slot, on;
};
struct InterRes {
    double  cost, cost_inter[5];
    int32_t cu_mode, best_idx, nnz[3], slot; // slot: first of the winner's three Slots (Y, U, V); -1: skipped (no coefficients)
    int16_t mv[2][2], mvd[2][2];
    int8_t  refi[2];
    uint8_t mvp_idx[2];
    uint32_t satd;                           // core->inter_satd
};
#define XW_NSLOT 12
struct Cw { // the workspace of one chain
    Node     node[5];
    Sbac     curr[5], next[5], before[5], tdepth[5], sbest, enext;
    IntraRes ires;
    InterRes eres;
    pel      nb[3][2][XW_NB];
    pel      ipred[5][4096];     // the five luma predictors of the intra analysis
    pel      cpred[2][4096];     // chroma predictors of its winner
    pel      epred[4][3][4096];  // inter: prediction of the candidates DIR / L0 / L1 / BI
    pel      upred[2][4][3][4096]; // inter: the uni-directional predictions of the merge candidates [list][index][component]
    pel      spred[3][4096];     // inter: the skip winner's prediction
    pel      bpred[4096];        // inter: the fixed list's luma prediction of a bi round
    u64      sk_ssd[16][3];      // inter: SSD of every merge candidate pair, per component
    u64      sk_dbk[16][3];      // inter, rdo_dbk_switch: the loop filter's share of it (two's complement)
    Sbac     cst[4][3];          // inter: exit coder states of a candidate's counts: all-zero CU, CU as quantised, the chosen combination
    pel      wrec[3][4096];      // inter: the winner's reconstruction
    int16_t  org_bi[4096];
    Slot     slot[XW_NSLOT];
    CtuData  best[5], temp[5];
};

enum { OP_ENTER = 0, OP_LEAF = 1, OP_CHILD_DONE = 2, OP_EXIT = 3, OP_ROOT_DONE = 4, OP_MID = 5, OP_INTRA = 6, OP_INTER = 7 };
struct Op {
    unsigned char op, lvl;
    signed char   part, pad_;
};
struct RefPic {
    const pel *y, *u, *v;
    int32_t    poc, pad_;
};

struct P { // one call
    int    nchains, C, full, rdo_dbk;
    int    log2_ctu, pic_w, pic_h, w_scu, h_scu, max_cu, min_cu, min_cuwh, idc, ws, hs, bd, slice_type, slice_qp, slice_num, cip;
    int    s_org_l, s_org_c, s_mod_l, s_mod_c;
    long   org_pic_l, org_pic_c, mod_pic_l, mod_pic_c, map_pic;
    int    qp[3], q_scale[3], dq_scale[3];
    int64_t err_scale[3][7]; // [component][log2 of the block size]
    double lambda[3], sqrt_lambda0, wgt[2];
    const pel *org[3];
    pel       *mod[3];
    uint32_t  *map_scu, *map_cu_mode;
    int8_t    *map_ipm;
    const uint8_t *map_tidx;
    const Sbac *states;
    const xeve_hip_ctu_job *jobs;
    CtuData *out;
    Sbac    *out_next;
    double  *out_cost;
    const int32_t  *entropy; // entropy_bits[1024] of xeve_init_bits_est
    const int8_t   *dct;     // the six DCT-II matrices [k][x], then the six transposed ones [x][k]
    const uint16_t *scan;    // zig-zag scans of the square blocks 2 .. 64
    const Op       *ops;
    int             nops;
    Cw             *cw;
    // P / B slices
    int    inter, isb, ecu_depth, vh, nref[2], max_cand, poc, col_list_poc0, s_ref_l, s_ref_c;
    int    refi_bits[2][XW_MAXR], range_recentre[2][XW_MAXR];
    xeve_hip_epzs_params me;
    double skip_th;
    RefPic refp[2 * XW_MAXR]; // [refi * 2 + list]
    int16_t (*map_mv)[2][2];
    int8_t  (*map_refi)[2];
    const int16_t (*col0)[2][2], (*col1)[2][2];
    const int16_t *mc_l, *mc_c; // [16][8], [32][4]
    u64 *prof;                  // [PR_N] cycles + [PR_N] marks, or null
    int  deal;                  // how a coder stage deals its jobs to the lanes: 0 round-robin over the team's waves, 1 packed into as few waves as hold them
    int  dbg;                   // debugging: the inter analysis stops after stage `dbg` (0: runs whole)
    u64 *sad_units;             // the walk's kernel-class timer is on: sample pairs the motion search compared are added here (XH_PROF_STRIPES words), else null
};
XW int dct_off(int log2n) { return ((1 << (2 * log2n)) - 4) / 3; }       // 0, 4, 20, 84, 340, 1364 (log2n 1 .. 6)
#define XW_DCT_ELEMS (4 + 16 + 64 + 256 + 1024 + 4096)
XW const int8_t *dct_m(const P &p, int log2n) { return p.dct + dct_off(log2n); }                  // [k][x]
XW const int8_t *dct_t(const P &p, int log2n) { return p.dct + XW_DCT_ELEMS + dct_off(log2n); }   // [x][k]
XW const uint16_t *scan_of(const P &p, int log2n) { return p.scan + dct_off(log2n); }

struct Blk { // one transform block of a stage; its scratch arrays (a Slot of the chain's workspace, or -- small blocks -- the team's LDS arena)
    const pel *org, *pred;
    int16_t   *coef, *lev; // residual -> DCT coefficients -> dequantised levels -> inverse transform output; the quantised levels
    int32_t   *tb;         // between the two passes of a transform
    pel       *rec;        // reconstruction (RDOQ's scan-ordered buffer before that)
    uint32_t  *ev;         // the levels as events
    int64_t   *rq;         // RDOQ scratch: [2][n]
    int        s_org, comp, on, nnz, nev, k, is_intra, any;
    int        sum_all, best_last;
    u64        ssd[2]; // SSD(prediction, original), SSD(reconstruction, original)
    u64        unc;    // RDOQ: the block's distortion with every level zero
};

XW void blk_slot(Blk &B, Slot *s) { B.coef = s->coef, B.lev = s->lev, B.tb = s->tb, B.rec = s->rec, B.ev = s->ev, B.rq = &s->rq[0][0]; }

enum { M_L0 = 0, M_L1 = 1, M_BI = 2, M_SKIP = 3, M_DIR = 4, M_NUM = 5 }; // PRED_* (xeve_def.h:461-469)
struct ISt { // the inter analysis of one chain's CU, between the stages
    int      on, go, x, y, pic;
    int16_t  mvp[2][4][2], mv_col[2];
    int16_t  mv[M_NUM][2][2], mvd[M_NUM][2][2];
    int8_t   refi[M_NUM][2];
    uint8_t  mvpi[M_NUM][2];
    int16_t  mv_scale[2][XW_MAXR][2];
    int32_t  mot_bits[2];
    double   cost_inter[M_NUM];
    int32_t  nnz[M_NUM][3];
    // skip / merge
    int32_t  dup[2];             // bit i: candidate i of the list repeats an earlier one
    // analyze_bi
    int32_t  lidx_ref, active, refi_best, changed;
    uint32_t best_mecost, best_mecost_l[2];
    int8_t   rf[2];
    int8_t   best, cu_mode;
    int8_t   csel[M_NUM];        // which of a candidate's counts gave its cost (index into Cw::cst[..])
};
#define XW_MEJ 16   // searches of one pass
#define XW_MEC 128  // candidates of one search round
enum { PH_D1 = 0, PH_RASTER, PH_RASTER_REF, PH_DREF, PH_IREF, PH_HPEL, PH_QPEL, PH_DONE };
enum { CT_NONE = 0, CT_DENSE, CT_LIST, CT_GRID, CT_SPEL, CT_RINGS };
struct MeJob { // one pinter_me_epzs call (xeve_pinter.c:699-869) as a state machine: rounds of candidates, evaluated by all lanes
    int        on, k, l, r, bi, x, y, so;
    const pel *ref, *org;
    int        gmvp[2], mvp[2], refi_bits, extra_bits, range_rc, range[4];
    int        phase, tmpstep, beststep, mv[2], mot_bits;
    unsigned   cost_best;
    int        step, not_found, faststep, bx, by, ix, iy, d_beststep, d_bits, d_run; // the running me_ipel_diamond
    unsigned   d_cost;
    int        r_mv[2], r_bits, r_pos, r_total, r_nx, r_stp, r_ss, r_cx, r_cy;       // me_raster
    unsigned   r_cost;
    int        s_mv[2], s_bits;                                                     // me_spel_pattern / me_ipel_refinement
    unsigned   s_cost;
    int        ctype, nc, c0, c1, c2;                                               // the round under evaluation
    short      cx[96], cy[96];
    short      ring_n, ring_end[8], ring_step[8];                                   // CT_RINGS: every remaining ring of the diamond in one round
};

struct Lds { // the team's shared memory
    union { // (the motion search and the residual / coder stages never run at the same time: they share their space)
        struct {
            uint16_t ctx[XW_NCTX * XW_CODL]; // [model][coder lane]
            Blk      blk[XW_MAXC * 9];
        };
        struct {
            MeJob    mej[XW_MEJ];
            unsigned mcost[XW_MEJ * XW_MEC];
            short    mbits[XW_MEJ * XW_MEC];
        };
    };
    ISt      ist[XW_MAXC];
    int      sh[XW_MAXC][24];        // per-chain scalars of the running stage
    int      acc[XW_MAXC * 40];      // integer sums of a stage (SATD per mode, SAD per candidate)
    int32_t  est[XW_MAXC][28];       // the rate tables RDOQ reads, of each chain's entry state
    int      flag[4];
    const Sbac *jsrc[XW_CODL]; // the coder lanes' entry states / where their exit states go (null: nowhere)
    Sbac       *jdst[XW_CODL];
    long long t0;
    int deal;
};
// a mark: everything since the previous mark belongs to class `cls` (call right after a sync)
XW void mark(const Tm &tm, const P &p, Lds &S, int cls)
{
#if XW_DEVICE
    if(p.prof && tm.tid == 0) {
        const long long t = clock64();
        if(blockIdx.x == 0) p.prof[cls] += (u64)(t - S.t0), p.prof[PR_N + cls] += 1;
        S.t0 = t;
    }
#else
    (void)tm, (void)p, (void)S, (void)cls;
#endif
}

// ---- tables ------------------------------------------------------------------------------------------------------------------------------------------------------
// xeve_tbl_mpm (xeve_tbl.c:40-48): [left mode + 1 | 0][up mode + 1 | 0] -> rank of every mode
XW int mpm_rank(int row, int m)
{
    constexpr uint8_t t[36][5] = {{0, 2, 3, 1, 4}, {0, 2, 1, 3, 4}, {0, 2, 1, 3, 4}, {1, 2, 0, 3, 4}, {0, 2, 1, 3, 4}, {0, 1, 2, 3, 4}, {1, 0, 2, 3, 4}, {0, 1, 2, 3, 4}, {0, 1, 2, 3, 4},
                                  {1, 2, 0, 3, 4}, {0, 1, 3, 2, 4}, {0, 2, 1, 4, 3}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {2, 0, 1, 3, 4}, {1, 0, 3, 2, 4}, {0, 1, 2, 4, 3},
                                  {1, 0, 2, 3, 4}, {0, 2, 1, 3, 4}, {1, 0, 2, 3, 4}, {1, 2, 0, 3, 4}, {0, 1, 2, 3, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 3, 2, 1, 4}, {1, 0, 2, 3, 4},
                                  {1, 2, 0, 3, 4}, {1, 2, 3, 0, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}, {0, 1, 2, 4, 3}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}};
    return t[row][m];
}

// ---- the arithmetic coder in bit-count mode (xeve_eco.c:392-575, xeve_mode.c:39-55) --------------------------------------------------------------------------------
// FULL: every field of XEVE_SBAC kept exactly (the C-ABI's callers compare exit states field for field).  !FULL: what a later COUNT depends on -- the range and the
// context models -- plus the number of renormalisation shifts, which IS xeve_get_bit_number's result: every shift decrements code_bits, every 8th moves a byte out of
// the code register, and bitcounter + 8 * (stacked + pending bytes) + 8 - code_bits + 3 counts exactly the shifts since xeve_sbac_bit_reset set code_bits to 11.
struct Cod {
    uint32_t  range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter, shifts;
    uint16_t *m; // the lane's models: model i at m[i * ms]
    int       ms;
};
#define XW_M(c, i) (c).m[(i) * (c).ms]
XW void cod_load(Cod &c, const Sbac &s, uint16_t *m, int ms)
{
    c.range = s.range, c.code = s.code, c.code_bits = s.code_bits, c.stacked_ff = s.stacked_ff, c.stacked_zero = s.stacked_zero, c.pending_byte = s.pending_byte;
    c.is_pending_byte = s.is_pending_byte, c.bitcounter = s.bitcounter, c.bin_counter = s.bin_counter, c.shifts = 0, c.m = m, c.ms = ms;
    for(int i = 0; i < XW_NCTX; i++) m[i * ms] = s.ctx[i];
}
XW void cod_reset(Cod &c)
{ // xeve_sbac_bit_reset
    c.code &= 0x7FFFF, c.code_bits = 11;
    c.pending_byte = c.is_pending_byte = c.stacked_ff = c.stacked_zero = c.bitcounter = c.bin_counter = 0, c.shifts = 0;
}
template <bool FULL> XW void cod_store(const Cod &c, Sbac &s)
{
    if(FULL) {
        s.range = c.range, s.code = c.code, s.code_bits = c.code_bits, s.stacked_ff = c.stacked_ff, s.stacked_zero = c.stacked_zero, s.pending_byte = c.pending_byte;
        s.is_pending_byte = c.is_pending_byte, s.bitcounter = c.bitcounter, s.bin_counter = c.bin_counter;
    }
    else {
        s.range = c.range, s.code = 0, s.code_bits = 11, s.stacked_ff = s.stacked_zero = s.pending_byte = s.is_pending_byte = s.bitcounter = s.bin_counter = 0;
    }
    for(int i = 0; i < XW_NCTX; i++) s.ctx[i] = c.m[i * c.ms];
}
template <bool FULL> XW unsigned cod_bits(const Cod &c)
{
    if(FULL) return c.bitcounter + 8 * (c.stacked_zero + c.stacked_ff) + 8 * (c.is_pending_byte ? 1 : 0) + 8 - c.code_bits + 3;
    return c.shifts;
}
XW void cod_byte(Cod &s, unsigned b)
{
    if(s.is_pending_byte) {
        if(s.pending_byte == 0) s.stacked_zero++;
        else s.bitcounter += 8 * s.stacked_zero + 8, s.stacked_zero = 0;
    }
    s.pending_byte = b & 0xFF, s.is_pending_byte = 1;
}
XW void cod_shift1(Cod &s)
{
    s.code <<= 1;
    if(--s.code_bits) return;
    const unsigned out = s.code >> 17;
    s.code &= (1u << 17) - 1;
    if(out < 0xFF) {
        for(; s.stacked_ff; s.stacked_ff--) cod_byte(s, 0xFF);
        cod_byte(s, out);
    }
    else if(out > 0xFF) {
        s.pending_byte++;
        for(; s.stacked_ff; s.stacked_ff--) cod_byte(s, 0);
        cod_byte(s, out);
    }
    else s.stacked_ff++;
    s.code_bits = 8;
}
template <bool FULL> XW void cod_shift(Cod &s, int n)
{ // n renormalisation shifts (a context-coded bin renormalises by 0 .. 5 bits, so at most one byte leaves)
    if(!FULL) {
        s.shifts += n;
        return;
    }
    while(n >= (int)s.code_bits) {
        n -= (int)s.code_bits;
        s.code <<= s.code_bits - 1, s.code_bits = 1;
        cod_shift1(s);
    }
    s.code <<= n, s.code_bits -= n;
}
// one context-coded bin on a model the caller holds in a register (xeve_sbac_encode_bin, xeve_eco.c:521-575)
template <bool FULL> XW void cod_bin_m(Cod &s, unsigned &model, unsigned bin)
{
    unsigned state = (model >> 1) & 511u, mps = model & 1;
    unsigned lps = mul24(state, s.range & 0xFFFFu) >> 9;
    if(lps < 437) lps = 437;
    if(FULL) s.bin_counter++;
    s.range -= lps;
    if((bin != 0) != (mps != 0)) {
        if(s.range >= lps) {
            if(FULL) s.code += s.range;
            s.range = lps;
        }
        state = state + ((512 - state + 16) >> 5);
        if(state > 256) mps = 1 - mps, state = 512 - state;
    }
    else state = state - ((state + 16) >> 5);
    model = (state << 1) + mps;
    if(s.range < 8192) {
        const int n = __builtin_clz(s.range) - 18;
        s.range <<= n;
        cod_shift<FULL>(s, n);
    }
}
template <bool FULL> XW void cod_bin(Cod &s, int ci, unsigned bin)
{
    unsigned model = XW_M(s, ci);
    cod_bin_m<FULL>(s, model, bin);
    XW_M(s, ci) = (uint16_t)model;
}
template <bool FULL> XW void cod_ep(Cod &s, unsigned bin)
{ // sbac_encode_bin_ep (xeve_eco.c:455-472): the range loses its LSB
    if(FULL) {
        s.bin_counter++;
        s.range >>= 1;
        if(bin) s.code += s.range;
        s.range <<= 1;
        cod_shift1(s);
    }
    else s.range &= ~1u, s.shifts++;
}
template <bool FULL> XW void cod_unary_m(Cod &s, unsigned &m0, unsigned &m1, unsigned sym)
{ // sbac_write_unary_sym with two models (xeve_eco.c:474-490)
    cod_bin_m<FULL>(s, m0, sym ? 1 : 0);
    while(sym) {
        sym--;
        cod_bin_m<FULL>(s, m1, sym ? 1 : 0);
    }
}
template <bool FULL> XW void cod_unary2(Cod &s, unsigned sym, int ci)
{
    unsigned m0 = XW_M(s, ci), m1 = XW_M(s, ci + 1);
    cod_unary_m<FULL>(s, m0, m1, sym);
    XW_M(s, ci) = (uint16_t)m0, XW_M(s, ci + 1) = (uint16_t)m1;
}
// xeve_eco_run_length_cc (xeve_eco.c:707-771), Baseline contexts (sps_cm_init_flag 0), from the block's event list
XW uint32_t ev_pack(int v, int run, int at_end)
{
    const unsigned a = (unsigned)(v < 0 ? -v : v) & 0xFFFFu;
    return ((a - 1) & 0x7FFFu) | ((unsigned)(v < 0) << 15) | ((unsigned)run << 16) | ((unsigned)at_end << 28);
}
// count-only form of one context-coded bin: branch-free, the model as (state, mps) in registers.  11 dependent operations from range to range.
struct Mdl {
    unsigned s, m;
};
XW Mdl mdl_of(unsigned v) { Mdl a; a.s = (v >> 1) & 511u, a.m = v & 1u; return a; }
XW unsigned mdl_pack(const Mdl &a) { return (a.s << 1) + a.m; }
XW void cnt_bin(unsigned &range, unsigned &shifts, Mdl &a, unsigned bin)
{
    unsigned lps = mul24(a.s, range) >> 9; // (range <= 2^14)
    lps = lps < 437u ? 437u : lps;
    const unsigned r2 = range - lps, is_lps = (bin ^ a.m) & 1u, msk = 0u - is_lps;
    const unsigned sl = a.s + ((528u - a.s) >> 5), flip = sl > 256u ? 1u : 0u, sl2 = flip ? 512u - sl : sl, sm = a.s - ((a.s + 16u) >> 5);
    const unsigned rl = r2 < lps ? r2 : lps, r = is_lps ? rl : r2;
    a.s = sm ^ ((sm ^ sl2) & msk), a.m = a.m ^ (flip & is_lps);
    const unsigned n = (unsigned)__builtin_clz(r) - 18u; // 437 <= r < 2^14: no shift from 2^13 on
    range = r << n, shifts += n;
}
#if defined(XW_CODER_PAIRS) && XW_CODER_PAIRS
// EXPERIMENT SWITCH (off in every shipped build; tools/gpu/r05_variants.sh measures it): two bins of a unary code per loop trip, the second one predicated -- half the
// loop branches (a branch on a vector condition costs a VALU -> SALU round trip per trip) for one wasted bin's arithmetic on odd lengths
XW void cnt_bin_if(unsigned &range, unsigned &shifts, Mdl &a, unsigned bin, bool on)
{
    unsigned r = range, sh = shifts;
    Mdl      b = a;
    cnt_bin(r, sh, b, bin);
    range = on ? r : range, shifts = on ? sh : shifts, a.s = on ? b.s : a.s, a.m = on ? b.m : a.m;
}
XW void cnt_unary(unsigned &range, unsigned &shifts, Mdl &m0, Mdl &m1, unsigned sym)
{
    cnt_bin(range, shifts, m0, sym ? 1u : 0u);
    while(sym) {
        const bool two = sym >= 2;
        cnt_bin(range, shifts, m1, sym > 1 ? 1u : 0u);
        cnt_bin_if(range, shifts, m1, sym > 2 ? 1u : 0u, two);
        sym -= two ? 2u : 1u;
    }
}
#else
XW void cnt_unary(unsigned &range, unsigned &shifts, Mdl &m0, Mdl &m1, unsigned sym)
{
    cnt_bin(range, shifts, m0, sym ? 1u : 0u);
    while(sym) {
        sym--;
        cnt_bin(range, shifts, m1, sym ? 1u : 0u);
    }
}
#endif
// the count-only event loop of a coefficient block: range, shift count and the five models it touches, in and out.  XW_NOINLINE_CODER=1 (walk.hip sets it) builds it as a
// real function -- a register allocation of its own instead of the enclosing stage's, where the coder's range travelled through a spill slot every event: 32 VGPRs, no
// scratch access inside the loops, -5 % per step on the device.
struct CntState {
    unsigned range, shifts;
    Mdl      r0, r1, l0, l1, la;
};
#if defined(XW_NOINLINE_CODER) && XW_NOINLINE_CODER
#define XW_CNT __host__ __device__ static __attribute__((noinline))
#else
#define XW_CNT XW
#endif
XW_CNT void cnt_events(CntState &st, const uint32_t *ev, int nev)
{
    Mdl      r0 = st.r0, r1 = st.r1, l0 = st.l0, l1 = st.l1, la = st.la;
    unsigned range = st.range, shifts = st.shifts;
    // (four events are on their way while one is coded: an event is 3 .. 10 bins of ~40 instructions, a load from the chain's workspace ~1 us)
    uint32_t e = nev > 0 ? ev[0] : 0u, e1 = nev > 1 ? ev[1] : 0u, e2 = nev > 2 ? ev[2] : 0u, e3 = nev > 3 ? ev[3] : 0u;
    for(int i = 0; i < nev; i++) {
        const uint32_t en = i + 4 < nev ? ev[i + 4] : 0u;
        cnt_unary(range, shifts, r0, r1, (e >> 16) & 0xFFFu);
        cnt_unary(range, shifts, l0, l1, e & 0x7FFFu);
        range &= ~1u, shifts++; // the sign, bypass coded
        if((e >> 28) & 1u) break;
        cnt_bin(range, shifts, la, i == nev - 1 ? 1u : 0u);
        e = e1, e1 = e2, e2 = e3, e3 = en;
    }
    st.r0 = r0, st.r1 = r1, st.l0 = l0, st.l1 = l1, st.la = la, st.range = range, st.shifts = shifts;
}
template <bool FULL> XW void cod_events(Cod &s, const uint32_t *ev, int nev, int ch)
{
    const int t0 = ch ? 2 : 0;
    if(!FULL) {
        CntState st;
        st.r0 = mdl_of(XW_M(s, XEVE_HIP_CTX_RUN + t0)), st.r1 = mdl_of(XW_M(s, XEVE_HIP_CTX_RUN + t0 + 1)), st.l0 = mdl_of(XW_M(s, XEVE_HIP_CTX_LEVEL + t0));
        st.l1 = mdl_of(XW_M(s, XEVE_HIP_CTX_LEVEL + t0 + 1)), st.la = mdl_of(XW_M(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0)));
        st.range = s.range, st.shifts = s.shifts;
        cnt_events(st, ev, nev);
        s.range = st.range, s.shifts = st.shifts;
        XW_M(s, XEVE_HIP_CTX_RUN + t0) = (uint16_t)mdl_pack(st.r0), XW_M(s, XEVE_HIP_CTX_RUN + t0 + 1) = (uint16_t)mdl_pack(st.r1);
        XW_M(s, XEVE_HIP_CTX_LEVEL + t0) = (uint16_t)mdl_pack(st.l0), XW_M(s, XEVE_HIP_CTX_LEVEL + t0 + 1) = (uint16_t)mdl_pack(st.l1);
        XW_M(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0)) = (uint16_t)mdl_pack(st.la);
        return;
    }
    unsigned r0 = XW_M(s, XEVE_HIP_CTX_RUN + t0), r1 = XW_M(s, XEVE_HIP_CTX_RUN + t0 + 1), l0 = XW_M(s, XEVE_HIP_CTX_LEVEL + t0), l1 = XW_M(s, XEVE_HIP_CTX_LEVEL + t0 + 1);
    unsigned la = XW_M(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0));
    for(int i = 0; i < nev; i++) {
        const uint32_t e = ev[i];
        cod_unary_m<FULL>(s, r0, r1, (e >> 16) & 0xFFFu);
        cod_unary_m<FULL>(s, l0, l1, e & 0x7FFFu);
        cod_ep<FULL>(s, (e >> 15) & 1u);
        if((e >> 28) & 1u) break; // the last scan position: no flag follows
        cod_bin_m<FULL>(s, la, i == nev - 1);
    }
    XW_M(s, XEVE_HIP_CTX_RUN + t0) = (uint16_t)r0, XW_M(s, XEVE_HIP_CTX_RUN + t0 + 1) = (uint16_t)r1;
    XW_M(s, XEVE_HIP_CTX_LEVEL + t0) = (uint16_t)l0, XW_M(s, XEVE_HIP_CTX_LEVEL + t0 + 1) = (uint16_t)l1, XW_M(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0)) = (uint16_t)la;
}
// xeve_eco_abs_mvd + sign (xeve_eco.c:1205-1270)
template <bool FULL> XW void cod_mvd1(Cod &s, int v)
{
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    uint32_t nn = (a + 1) >> 1;
    int len = 0;
    for(; len < 16 && nn; len++) nn >>= 1;
    const uint32_t code = (1u << len) | ((a + 1 - (1u << len)) & ((1u << len) - 1));
    const int nbin = 2 * len + 1;
    for(int i = 0; i < nbin; i++) {
        const uint32_t b = (code >> (nbin - 1 - i)) & 1;
        if(i <= 1) cod_bin<FULL>(s, XEVE_HIP_CTX_MVD, b);
        else cod_ep<FULL>(s, b);
    }
    if(a) cod_ep<FULL>(s, v < 0);
}
template <bool FULL> XW void cod_mvp_idx(Cod &s, int idx)
{ // sbac_write_truncate_unary_sym(idx, 3, 4) (xeve_eco.c:492-511, 1190-1203)
    for(int i = 0; i < 3; i++) {
        const int sym = i == idx ? 0 : 1;
        cod_bin<FULL>(s, XEVE_HIP_CTX_MVP_IDX + i, sym);
        if(!sym) break;
    }
}
template <bool FULL> XW void cod_refi(Cod &s, int num_refp, int refi)
{ // xeve_eco_refi (xeve_eco.c:1158-1188)
    if(num_refp <= 1) return;
    if(refi == 0) {
        cod_bin<FULL>(s, XEVE_HIP_CTX_REFI, 0);
        return;
    }
    cod_bin<FULL>(s, XEVE_HIP_CTX_REFI, 1);
    for(int i = 2; i < num_refp; i++) {
        const int bin = i == refi + 1 ? 0 : 1;
        if(i == 2) cod_bin<FULL>(s, XEVE_HIP_CTX_REFI + 1, bin);
        else cod_ep<FULL>(s, bin);
        if(!bin) break;
    }
}
// the coefficient part of a CU (xeve_eco_coef -> xeve_eco_cbf + xeve_eco_run_length_cc, xeve_eco.c:793-894, 1067-1089): nnz / events of Y, U, V; run = the components
// this call covers (bit 0 Y, 1 U, 2 V)
struct CoefSet {
    const uint32_t *ev[3];
    int             nev[3], nnz[3];
};
template <bool FULL> XW void cod_coef(Cod &s, int idc, const CoefSet &q, int run, int is_intra)
{
    const int cbf[3] = {q.nnz[0] != 0, q.nnz[1] != 0, q.nnz[2] != 0}, r0 = run & 1, r1 = (run >> 1) & 1, r2 = (run >> 2) & 1;
    const int cbf_all = (r0 && cbf[0]) + (r1 && cbf[1]) + (r2 && cbf[2]);
    if(!is_intra) {
        if(r0 + r1 + r2 == 3) {
            cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_ALL, cbf_all != 0);
            if(!cbf_all) return;
        }
        if(r1 && idc) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_CB, cbf[1]);
        if(r2 && idc) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_CR, cbf[2]);
        if(r0 && cbf[1] + cbf[2] != 0) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_LUMA, cbf[0]);
    }
    else {
        if(r1 && idc) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_CB, cbf[1]);
        if(r2 && idc) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_CR, cbf[2]);
        if(r0) cod_bin<FULL>(s, XEVE_HIP_CTX_CBF_LUMA, cbf[0]);
    }
    if(r0 && cbf[0]) cod_events<FULL>(s, q.ev[0], q.nev[0], 0);
    if(r1 && cbf[1]) cod_events<FULL>(s, q.ev[1], q.nev[1], 1);
    if(r2 && cbf[2]) cod_events<FULL>(s, q.ev[2], q.nev[2], 1);
}
// the split_cu_flag of a node (xeve_eco_split_mode, Baseline: one bin) counted from `from`; the state after it into `to` (the one model it touches is updated in place)
template <bool FULL> XW unsigned split_flag_bits(const Sbac &from, Sbac &to, int split)
{
    to = from;
    Cod c;
    c.range = from.range, c.code = from.code, c.code_bits = from.code_bits, c.shifts = 0, c.m = to.ctx, c.ms = 1;
    cod_reset(c);
    cod_bin<FULL>(c, XEVE_HIP_CTX_SPLIT_CU, split != 0);
    if(FULL) {
        to.range = c.range, to.code = c.code, to.code_bits = c.code_bits, to.stacked_ff = c.stacked_ff, to.stacked_zero = c.stacked_zero, to.pending_byte = c.pending_byte;
        to.is_pending_byte = c.is_pending_byte, to.bitcounter = c.bitcounter, to.bin_counter = c.bin_counter;
    }
    else to.range = c.range, to.code = 0, to.code_bits = 11, to.stacked_ff = to.stacked_zero = to.pending_byte = to.is_pending_byte = to.bitcounter = to.bin_counter = 0;
    return cod_bits<FULL>(c);
}

// ---- rate tables of RDOQ (xeve_rdoq_bit_est, xeve_mode.c:315-372): the Baseline run / level syntax touches contexts 0 .. 3 only ------------------------------------
enum { E_CBF_L = 0, E_CBF_CB = 2, E_CBF_CR = 4, E_CBF_ALL = 6, E_RUN = 8, E_LEVEL = 16, E_LAST = 24 }; // est[...][bin]
XW int32_t no_bits(const P &p, int symbol, uint16_t cm)
{
    const unsigned mps = cm & 1;
    unsigned state = cm >> 1;
    state = ((unsigned)(symbol != 0) != mps) ? state : 512 - state;
    return p.entropy[state << 1];
}
XW void est_entry(const P &p, const Sbac &s, int i, int32_t *e)
{ // entry i of the 28
    const int b = i & 1, g = i >> 1;
    int ci;
    if(g == 0) ci = XEVE_HIP_CTX_CBF_LUMA;
    else if(g == 1) ci = XEVE_HIP_CTX_CBF_CB;
    else if(g == 2) ci = XEVE_HIP_CTX_CBF_CR;
    else if(g == 3) ci = XEVE_HIP_CTX_CBF_ALL;
    else if(g < 8) ci = XEVE_HIP_CTX_RUN + (g - 4);
    else if(g < 12) ci = XEVE_HIP_CTX_LEVEL + (g - 8);
    else ci = XEVE_HIP_CTX_LAST + (g - 12);
    e[i] = no_bits(p, b, s.ctx[ci]);
}

// ---- a stage of bit-count jobs: job j on coder lane j (further rounds when there are more jobs than lanes) --------------------------------------------------------------
// src(j, in, out) -> is the job on, its entry state, where its exit state goes (null: nowhere); run(j, coder) codes the job's syntax and takes the bits.  The context
// models travel between the states (HBM / L2) and the lanes' LDS rows cooperatively, a word per lane.
XW int coder_lanes(const Tm &tm) { return imin(tm.n, XW_CODL); }
template <bool FULL, class Src, class Run> XW void coder_stage(const Tm &tm, Lds &S, int njobs, Src src, Run run)
{
    const int cl = coder_lanes(tm);
    for(int base = 0; base < njobs; base += cl) {
        const int cnt = imin(cl, njobs - base);
        for(int l = tm.tid; l < cnt; l += tm.n) {
            const Sbac *in = nullptr;
            Sbac       *out = nullptr;
            const bool  on = src(base + l, in, out);
            S.jsrc[l] = on ? in : nullptr, S.jdst[l] = on ? out : nullptr;
        }
        sync(tm);
        for(int i = tm.tid; i < cnt * (XW_NCTX / 2); i += tm.n) {
            const int l = i / (XW_NCTX / 2), w = i - l * (XW_NCTX / 2);
            const Sbac *in = S.jsrc[l];
            if(!in) continue;
            const uint32_t v = ((const uint32_t *)in->ctx)[w];
            S.ctx[(2 * w) * XW_CODL + l] = (uint16_t)v, S.ctx[(2 * w + 1) * XW_CODL + l] = (uint16_t)(v >> 16);
        }
        sync(tm);
        // job slot of this thread: the jobs are dealt round-robin over the team's waves (a wave runs as long as its longest lane per symbol: the fewer lanes of one
        // wave are busy, the less they wait for each other -- and the other waves sit on SIMDs that have nothing else to do in a coder stage)
        const int nw = tm.n >= 64 ? tm.n / 64 : 1, slot = tm.n >= 64 && !S.deal ? (tm.tid & 63) * nw + (tm.tid >> 6) : tm.tid;
        if(slot < cnt && S.jsrc[slot]) {
            const Sbac &in = *S.jsrc[slot];
            Cod c;
            c.range = in.range, c.code = in.code, c.code_bits = in.code_bits, c.stacked_ff = in.stacked_ff, c.stacked_zero = in.stacked_zero, c.pending_byte = in.pending_byte;
            c.is_pending_byte = in.is_pending_byte, c.bitcounter = in.bitcounter, c.bin_counter = in.bin_counter, c.shifts = 0, c.m = S.ctx + slot, c.ms = XW_CODL;
            cod_reset(c);
            run(base + slot, c);
            Sbac *o = S.jdst[slot];
            if(o) {
                if(FULL) {
                    o->range = c.range, o->code = c.code, o->code_bits = c.code_bits, o->stacked_ff = c.stacked_ff, o->stacked_zero = c.stacked_zero, o->pending_byte = c.pending_byte;
                    o->is_pending_byte = c.is_pending_byte, o->bitcounter = c.bitcounter, o->bin_counter = c.bin_counter;
                }
                else o->range = c.range, o->code = 0, o->code_bits = 11, o->stacked_ff = o->stacked_zero = o->pending_byte = o->is_pending_byte = o->bitcounter = o->bin_counter = 0;
            }
        }
        sync(tm);
        for(int i = tm.tid; i < cnt * (XW_NCTX / 2); i += tm.n) {
            const int l = i / (XW_NCTX / 2), w = i - l * (XW_NCTX / 2);
            Sbac *o = S.jdst[l];
            if(!o) continue;
            ((uint32_t *)o->ctx)[w] = (uint32_t)S.ctx[(2 * w) * XW_CODL + l] | ((uint32_t)S.ctx[(2 * w + 1) * XW_CODL + l] << 16);
        }
        sync(tm);
    }
}

// EXPERIMENT SWITCH (off in every shipped build; tools/gpu/r05_variants.sh measures it): XW_NOINLINE_STAGES=1 builds the block stages and the motion search's rounds as real
// functions -- a register allocation each instead of the one 128-VGPR allocation of the whole walk (static: 3966 -> 3528 scratch accesses in the kernel)
#if defined(XW_NOINLINE_STAGES) && XW_NOINLINE_STAGES
#define XW_ST __host__ __device__ static __attribute__((noinline))
#else
#define XW_ST XW
#endif
// ---- stages over transform blocks ------------------------------------------------------------------------------------------------------------------------------------
// residual = original - prediction into coef; optionally SSD(prediction, original) (the shift per sample, xeve_ssd_16b)
XW_ST void st_diff(const Tm &tm, Blk *b, int nb, int log2n, int want_ssd, int bd)
{
    const int N = 1 << log2n, per = N * N / (N >= 4 ? 4 : 2), g = N >= 4 ? 4 : 2, sh = (bd - 8) * 2;
    for(int i = tm.tid; i < nb * per; i += tm.n) {
        const int bi = i / per, e = (i - bi * per) * g, y = e >> log2n, x = e & (N - 1);
        Blk &B = b[bi];
        if(!B.on) continue;
        const pel *o = B.org + (long)y * B.s_org + x, *q = B.pred + e;
        int16_t   *d = B.coef + e;
        u64 acc = 0;
        for(int t = 0; t < g; t++) {
            const int v = (int)o[t] - (int)q[t];
            d[t] = (int16_t)v;
            acc += (u64)((v * v) >> sh);
        }
        if(want_ssd) aadd64(&B.ssd[0], acc);
    }
}
// four (G) outputs of a transform pass: acc[t] = sum_i m[i * ms + t] * v[i * vs], i < n.  The loads of eight terms are issued together (a pass is a chain of
// memory round trips otherwise); an s32 source is split at 14 bits so that both partial sums stay in s32 (|v| < 2^29, n <= 64, |m| <= 90) -- exact.
template <class T> XW void dot4(const int8_t *m, int ms, const T *v, int vs, int n, int G, int64_t acc[4])
{
    constexpr bool wide = sizeof(T) == 4;
    int32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    int i = 0;
    if(G == 4)
        for(; i + 8 <= n; i += 8) {
            int32_t  x[8];
            uint32_t w[8];
#pragma unroll
            for(int t = 0; t < 8; t++) x[t] = (int32_t)v[(i + t) * vs], w[t] = *reinterpret_cast<const uint32_t *>(m + (i + t) * ms);
#pragma unroll
            for(int t = 0; t < 8; t++) {
                const int32_t xl = wide ? (x[t] & 16383) : x[t], xh = wide ? (x[t] >> 14) : 0;
#pragma unroll
                for(int q = 0; q < 4; q++) {
                    const int32_t c = (int8_t)(w[t] >> (8 * q));
                    lo[q] += c * xl;
                    if(wide) hi[q] += c * xh;
                }
            }
        }
    for(; i < n; i++) {
        const int32_t x = (int32_t)v[i * vs], xl = wide ? (x & 16383) : x, xh = wide ? (x >> 14) : 0;
        for(int q = 0; q < G; q++) {
            const int32_t c = m[i * ms + q];
            lo[q] += c * xl;
            if(wide) hi[q] += c * xh;
        }
    }
    for(int q = 0; q < 4; q++) acc[q] = wide ? ((int64_t)hi[q] << 14) + lo[q] : (int64_t)lo[q];
}
// one pass of a transform over every block: pass 0 / 1 = xeve_trans rows / columns (xeve_tq.c:396-404), 2 / 3 = xeve_itrans (xeve_itdq.c:435-440)
XW_ST void st_tpass(const Tm &tm, const P &p, Blk *b, int nb, int log2n, int pass)
{
    const int N = 1 << log2n, G = N >= 4 ? 4 : N, per = N * (N / G);
    const int8_t *M = dct_m(p, log2n), *Mt = dct_t(p, log2n);
    const int fshift = (log2n - 1 + p.bd - 8) + (log2n + 6), ishift = 7 + (12 - (p.bd - 8));
    for(int i = tm.tid; i < nb * per; i += tm.n) {
        const int bi = i / per, e = i - bi * per, j = e / (N / G), g0 = (e - j * (N / G)) * G;
        Blk &B = b[bi];
        if(!B.on || (pass >= 2 && !B.nnz)) continue;
        int64_t a[4];
        if(pass == 0) { // tb[k * N + j] = sum_x M[k][x] * coef[j * N + x]
            dot4(Mt + g0, N, B.coef + j * N, 1, N, G, a);
            for(int t = 0; t < G; t++) B.tb[(g0 + t) * N + j] = (N == 64 && g0 + t >= 32) ? 0 : (int32_t)a[t];
        }
        else if(pass == 1) { // coef[k * N + j] = (sum_x M[k][x] * tb[j * N + x] + add) >> shift
            dot4(Mt + g0, N, B.tb + j * N, 1, N, G, a);
            const int64_t add = (int64_t)1 << (fshift - 1);
            for(int t = 0; t < G; t++) B.coef[(g0 + t) * N + j] = (N == 64 && g0 + t >= 32) ? (int16_t)0 : (int16_t)((a[t] + add) >> fshift);
        }
        else if(pass == 2) { // tb[j * N + x] = clip32(sum_k M[k][x] * coef[k * N + j]); (the 64-point inverse reads the 32 rows a forward transform can leave)
            dot4(M + g0, N, B.coef + j, N, N == 64 ? 32 : N, G, a);
            for(int t = 0; t < G; t++) B.tb[j * N + g0 + t] = (int32_t)(a[t] < INT32_MIN ? INT32_MIN : a[t] > INT32_MAX ? INT32_MAX : a[t]);
        }
        else { // coef[j * N + x] = clip16((sum_k M[k][x] * tb[k * N + j] + add) >> shift)
            dot4(M + g0, N, B.tb + j, N, N == 64 ? 32 : N, G, a);
            const int64_t add = (int64_t)1 << (ishift - 1);
            for(int t = 0; t < G; t++) {
                const int64_t v = (a[t] + add) >> ishift;
                B.coef[j * N + g0 + t] = (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
            }
        }
    }
}
// get_ic_rate_cost_rl (xeve_tq.c:425-456): s32 rate arithmetic as the reference; c = 0 luma / 2 chroma
XW int64_t rl_cost(unsigned abs_level, int run_nonzero, int c, int64_t lambda, const int32_t *e)
{
    uint32_t rate;
    if(abs_level == 0) rate = (uint32_t)e[E_RUN + 2 * (c + run_nonzero) + 1];
    else {
        rate = 32768u + (uint32_t)e[E_RUN + 2 * (c + run_nonzero)];
        if(abs_level == 1) rate += (uint32_t)e[E_LEVEL + 2 * c];
        else rate += (uint32_t)e[E_LEVEL + 2 * c + 1] + (uint32_t)e[E_LEVEL + 2 * (c + 1) + 1] * (abs_level - 2) + (uint32_t)e[E_LEVEL + 2 * (c + 1)];
    }
    return (int64_t)(int32_t)rate * lambda;
}
// The zero-block pre-test (xeve_tq.c:666-699) + xeve_rdoq_run_length_cc (:497-649) of square blocks + their event lists, in four steps:
//   a (lane per position)  the coefficients into scan order (rec serves as the buffer: the reconstruction is written later), the pre-test, the sums over the block
//   b (lane per block)     the run-length automaton over the scan: the level kept per position written back in place, best_last
//   c (lane per position)  the levels into the block, zero from best_last on;   d (lane per block)  the event list and the number of levels
XW void rdoq_level(int v, int q_value, int q_bits, int64_t cap, int64_t &ld, uint32_t &m)
{
    const int64_t t = (int64_t)iabs(v) * q_value;
    ld = (int)(t < cap ? t : cap);
    m = (uint32_t)(ld >> q_bits);
    if(!((ld - ((int64_t)m << q_bits)) < ((int64_t)1 << (q_bits - 1)))) m++;
}
XW_ST void st_rdoq(const Tm &tm, const P &p, Lds &S, Blk *b, int nb, int log2n)
{
    const int N = 1 << log2n, nn = N * N, bd = p.bd;
    const uint16_t *scan = scan_of(p, log2n);
    for(int i = tm.tid; i < nb; i += tm.n) b[i].any = 0, b[i].sum_all = 0, b[i].unc = 0, b[i].best_last = 0, b[i].nnz = 0, b[i].nev = 0;
    sync(tm);
    for(int i = tm.tid; i < nb * nn; i += tm.n) { // a
        const int bi = i >> (2 * log2n), q = i & (nn - 1);
        Blk &B = b[bi];
        if(!B.on) continue;
        const int comp = B.comp, qp = p.qp[comp], q_value = p.q_scale[comp], q_bits = 14 + (15 - bd - log2n) + qp / 6;
        const int v = B.coef[scan[q]];
        B.rec[q] = (pel)v;
        if(!v) continue;
        const int64_t zthr = ((int64_t)1 << q_bits) - ((int64_t)(p.slice_type == 2 ? 201 : 153) << (q_bits - 9));
        if((int64_t)iabs(v) * q_value >= zthr) aor(&B.any, 1);
        int64_t  ld;
        uint32_t m;
        rdoq_level(v, q_value, q_bits, (int64_t)INT32_MAX - ((int64_t)1 << (q_bits - 1)), ld, m);
        const int64_t es = p.err_scale[comp][log2n], err = (ld * es) >> 20, uncoded = err * err;
        aadd64(&B.unc, (u64)uncoded);
        if(m) aadd(&B.sum_all, (int)m);
        // get_coded_level_rl (xeve_tq.c:458-490) for both run states: the best level and what it costs beyond coding nothing
        const int32_t *est = S.est[B.k];
        const int c = comp ? 2 : 0;
        const int64_t lambda = (int64_t)(p.lambda[comp] * (double)(1 << 15) + 0.5);
        const int16_t  mx = (int16_t)(v > 0 ? (int16_t)m : -(int16_t)m);
        const uint32_t max_abs = (uint32_t)iabs(mx), lo = max_abs > 1 ? max_abs - 1 : 1;
        uint32_t packed = 0;
        for(int rs = 0; rs < 2; rs++) {
            int64_t  coded = uncoded + rl_cost(0, rs, c, lambda, est);
            uint32_t best = 0;
            for(uint32_t a = max_abs; a >= lo; a--) {
                const int64_t d = ld - ((int64_t)a << q_bits), e2 = (d * es) >> 20, cost = e2 * e2 + rl_cost(a, rs, c, lambda, est);
                if(cost < coded) best = a, coded = cost;
            }
            B.rq[rs * nn + q] = coded - uncoded, packed |= (best & 0xFFFFu) << (16 * rs);
        }
        B.tb[q] = (int32_t)packed;
    }
    sync(tm), mark(tm, p, S, PR_Q_A);
    for(int i = tm.tid; i < nb; i += tm.n) { // b
        Blk &B = b[i];
        if(!B.on || !B.any || !B.sum_all) continue;
        const int comp = B.comp, c = comp ? 2 : 0, ctx_last = comp ? 1 : 0;
        const int32_t *est = S.est[B.k];
        const int64_t lambda = (int64_t)(p.lambda[comp] * (double)(1 << 15) + 0.5);
        const int32_t *cbf = est + (B.is_intra ? (comp == 0 ? E_CBF_L : comp == 1 ? E_CBF_CB : E_CBF_CR) : (comp == 0 ? E_CBF_ALL : comp == 1 ? E_CBF_CB : E_CBF_CR));
        const int64_t z0 = rl_cost(0, 0, c, lambda, est), z1 = rl_cost(0, 1, c, lambda, est), last0 = (int64_t)est[E_LAST + 2 * ctx_last] * lambda;
        const int64_t last1 = (int64_t)est[E_LAST + 2 * ctx_last + 1] * lambda;
        int64_t  best_cost = (int64_t)B.unc + (int64_t)cbf[0] * lambda, base_cost = (int64_t)B.unc + (int64_t)cbf[1] * lambda;
        uint32_t run = 0, best_last = 0;
        pel *sc = B.rec;
        // (eight positions at a time: their operands are fetched together -- the scan is a chain of memory round trips otherwise)
        for(int q0 = 0; q0 < nn; q0 += 8) {
            const int cnt = nn - q0 < 8 ? nn - q0 : 8;
            int      v8[8];
            int32_t  t8[8];
            int64_t  r0[8], r1[8];
            int      any = 0;
#pragma unroll
            for(int t = 0; t < 8; t++) v8[t] = t < cnt ? sc[q0 + t] : 0, any |= v8[t];
            if(!any) { // eight zeros: the first continues the run it finds, the rest are zeros after a zero
                base_cost += (run ? z1 : z0) + (int64_t)(cnt - 1) * z1, run += cnt;
                continue;
            }
            {
#pragma unroll
                for(int t = 0; t < 8; t++)
                    if(t < cnt) t8[t] = B.tb[q0 + t], r0[t] = B.rq[q0 + t], r1[t] = B.rq[nn + q0 + t];
            }
#pragma unroll
            for(int t = 0; t < 8; t++) {
                if(t >= cnt) break;
                const int v = v8[t];
                if(!v) { // a zero: the rate of one more zero of the run
                    base_cost += run ? z1 : z0;
                    run++;
                    continue;
                }
                const int rs = run != 0;
                const uint32_t best = ((uint32_t)t8[t] >> (16 * rs)) & 0xFFFFu;
                sc[q0 + t] = (pel)(v < 0 ? -(int32_t)best : (int32_t)best);
                base_cost += rs ? r1[t] : r0[t];
                if(best) {
                    const int64_t cur_is_last = base_cost + last1;
                    base_cost += last0;
                    if(cur_is_last < best_cost) best_cost = cur_is_last, best_last = (uint32_t)(q0 + t) + 1;
                    run = 0;
                }
                else run++;
            }
        }
        B.best_last = (int)best_last;
    }
    sync(tm), mark(tm, p, S, PR_Q_B);
    for(int i = tm.tid; i < nb * nn; i += tm.n) { // c
        const int bi = i >> (2 * log2n), q = i & (nn - 1);
        const Blk &B = b[bi];
        if(B.on) B.lev[scan[q]] = q < B.best_last ? B.rec[q] : (pel)0;
    }
    for(int i = tm.tid; i < nb; i += tm.n) { // d
        Blk &B = b[i];
        if(!B.on) continue;
        const pel *sc = B.rec;
        int run = 0, nnz = 0;
        for(int q0 = 0; q0 < B.best_last; q0 += 8) {
            const int cnt = B.best_last - q0 < 8 ? B.best_last - q0 : 8;
            int v8[8];
#pragma unroll
            for(int t = 0; t < 8; t++) v8[t] = t < cnt ? sc[q0 + t] : 0;
#pragma unroll
            for(int t = 0; t < 8; t++) {
                if(t >= cnt) break;
                if(!v8[t]) {
                    run++;
                    continue;
                }
                B.ev[nnz++] = ev_pack(v8[t], run, q0 + t == nn - 1);
                run = 0;
            }
        }
        B.nnz = nnz, B.nev = nnz;
    }
}
// xeve_dquant (xeve_itdq.c:442-475) of the levels into coef, for the blocks that have any
XW_ST void st_dquant(const Tm &tm, const P &p, Blk *b, int nb, int log2n)
{
    const int nn = 1 << (2 * log2n), shift = (uint8_t)(20 - 14 - (15 - p.bd - log2n));
    const int32_t offset = shift == 0 ? 0 : 1 << (shift - 1);
    for(int i = tm.tid; i < nb * nn; i += tm.n) {
        const int bi = i >> (2 * log2n), e = i & (nn - 1);
        Blk &B = b[bi];
        if(!B.on || !B.nnz) continue;
        const int64_t l = ((int64_t)B.lev[e] * p.dq_scale[B.comp] + offset) >> shift;
        B.coef[e] = (int16_t)(l < -32768 ? -32768 : l > 32767 ? 32767 : l);
    }
}
// xeve_recon_blk (xeve_recon.c:34-57: the sum wraps to s16 before the clip) + SSD(reconstruction, original)
XW_ST void st_recon(const Tm &tm, const P &p, Blk *b, int nb, int log2n)
{
    const int N = 1 << log2n, g = N >= 4 ? 4 : 2, per = N * N / g, sh = (p.bd - 8) * 2, maxv = (1 << p.bd) - 1;
    for(int i = tm.tid; i < nb * per; i += tm.n) {
        const int bi = i / per, e = (i - bi * per) * g, y = e >> log2n, x = e & (N - 1);
        Blk &B = b[bi];
        if(!B.on) continue;
        const pel *o = B.org + (long)y * B.s_org + x, *q = B.pred + e;
        const int16_t *r = B.coef + e;
        pel *d = B.rec + e;
        u64 acc = 0;
        for(int t = 0; t < g; t++) {
            const int16_t w = B.nnz ? (int16_t)(r[t] + q[t]) : q[t];
            const int v = w < 0 ? 0 : w > maxv ? maxv : w, df = v - (int)o[t];
            d[t] = (pel)v;
            acc += (u64)((df * df) >> sh);
        }
        aadd64(&B.ssd[1], acc);
    }
}
// residual .. reconstruction of the blocks (all of one size): what pintra_residue_rdo / pinter_residue_rdo do between the prediction and the bit count
XW void blocks_chain(const Tm &tm, const P &p, Lds &S, Blk *b, int nb, int log2n, int want_pred_ssd)
{
    st_diff(tm, b, nb, log2n, want_pred_ssd, p.bd);
    sync(tm), mark(tm, p, S, PR_B_DIFF);
    st_tpass(tm, p, b, nb, log2n, 0);
    sync(tm), mark(tm, p, S, PR_B_T0);
    st_tpass(tm, p, b, nb, log2n, 1);
    sync(tm), mark(tm, p, S, PR_B_T1);
    st_rdoq(tm, p, S, b, nb, log2n);
    sync(tm), mark(tm, p, S, PR_B_RDOQ);
    st_dquant(tm, p, b, nb, log2n);
    sync(tm), mark(tm, p, S, PR_B_DQ);
    st_tpass(tm, p, b, nb, log2n, 2);
    sync(tm), mark(tm, p, S, PR_B_T2);
    st_tpass(tm, p, b, nb, log2n, 3);
    sync(tm), mark(tm, p, S, PR_B_T3);
    st_recon(tm, p, b, nb, log2n);
    sync(tm), mark(tm, p, S, PR_B_REC);
}

// xeve_had of one 8x8 or 4x4 tile of (org - cur) (xeve_sad.c:419-602): unnormalised Hadamard, the DC term >> 2, rounding per size.  Fully unrolled: the tile
// lives in registers.
template <int n> XW int had_tile_n(const pel *org, int s_org, const pel *cur, int s_cur)
{
    int t[n * n];
#pragma unroll
    for(int y = 0; y < n; y++)
#pragma unroll
        for(int x = 0; x < n; x++) t[y * n + x] = (int)org[y * s_org + x] - (int)cur[y * s_cur + x];
#pragma unroll
    for(int pass = 0; pass < 2; pass++) {
        const int st = pass ? n : 1, line = pass ? 1 : n;
#pragma unroll
        for(int r = 0; r < n; r++)
#pragma unroll
            for(int len = 1; len < n; len <<= 1)
#pragma unroll
                for(int base = 0; base < n; base += 2 * len)
#pragma unroll
                    for(int i = base; i < base + len; i++) {
                        const int a = t[r * line + i * st], c = t[r * line + (i + len) * st];
                        t[r * line + i * st] = a + c, t[r * line + (i + len) * st] = a - c;
                    }
    }
    int sum = iabs(t[0]) >> 2;
#pragma unroll
    for(int i = 1; i < n * n; i++) sum += iabs(t[i]);
    return n == 4 ? (sum + 1) >> 1 : (sum + 2) >> 2;
}
XW int had_tile(const pel *org, int s_org, const pel *cur, int s_cur, int n)
{
    return n == 4 ? had_tile_n<4>(org, s_org, cur, s_cur) : had_tile_n<8>(org, s_org, cur, s_cur);
}

} // namespace xw
#include "walk_dbk.h"
#include "walk_intra.h"
#include "walk_inter.h"
#include "walk_tree.h"
