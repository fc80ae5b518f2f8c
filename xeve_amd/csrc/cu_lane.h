// xeve_amd/csrc/cu_lane.h -- the intra analysis of ONE small CU (4x4, 8x8) decided by ONE lane: pintra_analyze_cu (src_base/xeve_pintra.c:544-698) start to end.
//
// Why a lane: inside a chain of the CTU walk (tree.hip) the CUs are strictly serial, and 320 of the 341 nodes of an I-picture CTU are 4x4 / 8x8 CUs whose
// arithmetic is a few thousand operations.  Composed from the batched kernels such a node is 29 dependent launches (~ 200 us of launch latency for ~ 2 us of
// work); decided by one lane it is one launch for all chains, and the parallel axis is the only one the problem has: the chains (64 per wave).  The code is
// therefore plain scalar C++ per lane -- neighbours, the five predictors, SATD, the candidate list, per candidate the residual, DCT, RDOQ, the intra luma syntax
// through the arithmetic coder, reconstruction and SSD; chroma with the winner's mode; the CU's cost from the whole syntax and the exit coder state.
// Every function is __host__ __device__: libxeve_hip.so instantiates the device side only (k_intra_lane in tree.hip); tests/native builds the same functions for
// the host so that `pytest -m "not gpu"` checks them bit for bit against the oracle without a GPU.
#pragma once
#include <stdint.h>
#include "../../include/xeve_hip.h"

// A translation unit may define XL (e.g. with always_inline) and XL_CTX (where the coder's context models live) before including this header: the bitstream
// writer's kernels (encode.hip) keep the models in LDS and the coder core in registers that way.  The defaults are the plain record.
#ifndef XL
#define XL __host__ __device__ static inline
#endif
#ifndef XL_CTX
#define XL_CTX(s, ci) (s).ctx[ci]
#endif
// XL_SINK(o): does this coder write bytes?  A writer-only translation unit defines it as true: a comparison of the sink's (private) address with null is one the
// compiler does not fold, and it keeps the sink's fields in scratch memory instead of registers.
#ifndef XL_SINK
#define XL_SINK(o) ((o) != nullptr)
#endif

namespace xl {
typedef int16_t pel;
typedef xeve_hip_sbac Sbac;

struct Params { // one level of one call
    int     idc, bd, slice_type, cip, w_scu, h_scu, s_org_l, s_org_c, s_mod_l, s_mod_c;
    int     qp[3], q_scale[3], dq_scale[3]; // xeve_quant_scale[0][qp % 6]; xeve_tbl_dq_scale_b[qp % 6] << (qp / 6)
    int64_t err_scale[3];                   // get_err_scale (xeve_tq.c:406-423) of each component's block size and QP
    double  lambda[3], sqrt_lambda0, wgt[2];
    const int32_t *entropy;                 // entropy_bits[1024] of xeve_init_bits_est (xeve_mode.c:304-313)
};
struct Est { // the rate tables RDOQ reads (xeve_rdoq_bit_est, xeve_mode.c:326-372): the Baseline run / level syntax touches contexts 0 .. 3 only
    int32_t cbf[3][2], run[4][2], level[4][2], last[2][2];
};

// ---- tables ---------------------------------------------------------------------------------------------------------------------------------------------------
// DCT-II matrices of xeve_tbl_tm2 / tm4 / tm8 from the closed form round(64 * sqrt(2) * cos(...)) (first row 64); zig-zag scans of xeve_tbl_scan (xeve_util.c:1301-1325)
XL int dct_m(int n, int k, int x)
{
    static constexpr int8_t t2[4]  = {64, 64, 64, -64};
    static constexpr int8_t t4[16] = {64, 64, 64, 64, 84, 35, -35, -84, 64, -64, -64, 64, 35, -84, 84, -35};
    static constexpr int8_t t8[64] = {64, 64,  64,  64,  64,  64,  64,  64,  89, 75,  50,  18,  -18, -50, -75, -89, 84, 35,  -35, -84, -84, -35, 35,  84,  75, -18, -89, -50, 50,  89,  18,  -75,
                           64, -64, -64, 64,  64,  -64, -64, 64,  50, -89, 18,  75,  -75, -18, 89,  -50, 35, -84, 84,  -35, -35, 84,  -84, 35,  18, -50, 75,  -89, 89,  -75, 50,  -18};
    return n == 2 ? t2[k * 2 + x] : n == 4 ? t4[k * 4 + x] : t8[k * 8 + x];
}
XL int zigzag(int n, int p)
{
    static constexpr uint8_t z4[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
    static constexpr uint8_t z8[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return n == 2 ? p : n == 4 ? z4[p] : z8[p];
}
// xeve_tbl_mpm (xeve_tbl.c:40-48): [left mode + 1 | 0][up mode + 1 | 0] -> rank of every mode
XL int mpm_rank(int l, int u, int m)
{
    static constexpr uint8_t t[36][5] = {{0, 2, 3, 1, 4}, {0, 2, 1, 3, 4}, {0, 2, 1, 3, 4}, {1, 2, 0, 3, 4}, {0, 2, 1, 3, 4}, {0, 1, 2, 3, 4}, {1, 0, 2, 3, 4}, {0, 1, 2, 3, 4}, {0, 1, 2, 3, 4},
                              {1, 2, 0, 3, 4}, {0, 1, 3, 2, 4}, {0, 2, 1, 4, 3}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {2, 0, 1, 3, 4}, {1, 0, 3, 2, 4}, {0, 1, 2, 4, 3},
                              {1, 0, 2, 3, 4}, {0, 2, 1, 3, 4}, {1, 0, 2, 3, 4}, {1, 2, 0, 3, 4}, {0, 1, 2, 3, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 3, 2, 1, 4}, {1, 0, 2, 3, 4},
                              {1, 2, 0, 3, 4}, {1, 2, 3, 0, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}, {0, 1, 2, 4, 3}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}};
    return t[l * 6 + u][m];
}

// ---- the arithmetic coder in bit-count mode (xeve_eco.c:392-575, xeve_mode.c:39-55): every field of XEVE_SBAC kept exactly ---------------------------------------
// write mode (the bitstream writer's coder, is_bitcount clear): the bytes go to a sink instead of advancing the counter
struct Sink {
    uint8_t *p;
    int      cap, n;
};
XL void sb_byte(Sbac &s, unsigned b, Sink *o = nullptr)
{
    if(s.is_pending_byte) {
        if(s.pending_byte == 0) s.stacked_zero++;
        else if(XL_SINK(o)) {
            for(; s.stacked_zero; s.stacked_zero--) {
                if(o->n < o->cap) o->p[o->n] = 0;
                o->n++;
            }
            if(o->n < o->cap) o->p[o->n] = (uint8_t)s.pending_byte;
            o->n++;
        }
        else s.bitcounter += 8 * s.stacked_zero + 8, s.stacked_zero = 0;
    }
    s.pending_byte = b & 0xFF, s.is_pending_byte = 1;
}
XL void sb_shift(Sbac &s, Sink *o = nullptr)
{
    s.code <<= 1;
    if(--s.code_bits) return;
    const unsigned out = s.code >> 17;
    s.code &= (1u << 17) - 1;
    if(out < 0xFF) {
        for(; s.stacked_ff; s.stacked_ff--) sb_byte(s, 0xFF, o);
        sb_byte(s, out, o);
    }
    else if(out > 0xFF) {
        s.pending_byte++;
        for(; s.stacked_ff; s.stacked_ff--) sb_byte(s, 0, o);
        sb_byte(s, out, o);
    }
    else s.stacked_ff++;
    s.code_bits = 8;
}
// n sb_shift steps at once (a context-coded bin renormalises by 0 .. 5 bits, so at most one byte leaves): the code register reaches the byte boundary with the same
// value as bit by bit, the byte leaves there, the rest of the shift follows
XL void sb_shift_n(Sbac &s, int n, Sink *o = nullptr)
{
    while(n >= (int)s.code_bits) {
        n -= (int)s.code_bits;
        s.code <<= s.code_bits - 1, s.code_bits = 1;
        sb_shift(s, o); // the boundary step itself: byte out, code_bits = 8
    }
    s.code <<= n, s.code_bits -= n;
}
// one context-coded bin on a model the caller holds in a register (a run of bins on one model -- the tail of a unary symbol -- loads and stores it once)
XL void sb_bin_m(Sbac &s, unsigned &model, unsigned bin, Sink *o = nullptr)
{
    unsigned state = (model >> 1) & 511u, mps = model & 1;
    unsigned lps = (state * (s.range & 0xFFFFu)) >> 9; // (the masks change nothing -- 9-bit state, 16-bit range -- and let the device use its full-rate 24-bit multiply)
    if(lps < 437) lps = 437;
    s.bin_counter++;
    s.range -= lps;
    if((bin != 0) != (mps != 0)) {
        if(s.range >= lps) s.code += s.range, s.range = lps;
        state = state + ((512 - state + 16) >> 5);
        if(state > 256) mps = 1 - mps, state = 512 - state;
    }
    else state = state - ((state + 16) >> 5);
    model = (state << 1) + mps;
    if(s.range < 8192) { // (xeve_sbac_encode_bin's renormalisation loop, :559-575, in one step)
        const int n = __builtin_clz(s.range) - 18;
        s.range <<= n;
        sb_shift_n(s, n, o);
    }
}
XL void sb_bin(Sbac &s, int ci, unsigned bin, Sink *o = nullptr)
{
    unsigned model = XL_CTX(s, ci);
    sb_bin_m(s, model, bin, o);
    XL_CTX(s, ci) = (uint16_t)model;
}
XL void sb_bin_ep(Sbac &s, unsigned bin, Sink *o = nullptr)
{   // (the range loses its LSB, xeve_eco.c:455-472)
    s.bin_counter++;
    s.range >>= 1;
    if(bin) s.code += s.range;
    s.range <<= 1;
    sb_shift(s, o);
}
XL void sb_bit_reset(Sbac &s)
{
    s.code &= 0x7FFFF, s.code_bits = 11;
    s.pending_byte = s.is_pending_byte = s.stacked_ff = s.stacked_zero = s.bitcounter = s.bin_counter = 0;
}
XL unsigned sb_bits(const Sbac &s) { return s.bitcounter + 8 * (s.stacked_zero + s.stacked_ff) + 8 * (s.is_pending_byte ? 1 : 0) + 8 - s.code_bits + 3; }
XL void sb_unary2(Sbac &s, unsigned sym, int ci, Sink *o = nullptr)
{   // sbac_write_unary_sym with two models (xeve_eco.c:474-490)
    sb_bin(s, ci, sym ? 1 : 0, o);
    if(!sym) return;
    unsigned model = XL_CTX(s, ci + 1);
    while(sym) {
        sym--;
        sb_bin_m(s, model, sym ? 1 : 0, o);
    }
    XL_CTX(s, ci + 1) = (uint16_t)model;
}
// xeve_eco_run_length_cc (xeve_eco.c:707-771), Baseline contexts (sps_cm_init_flag 0); n: 2, 4, 8 through the tables, larger blocks through the scan the caller passes
XL void sb_run_length(Sbac &s, const int16_t *coef, int n, int num_sig, int ch, Sink *o = nullptr, const uint16_t *scan = nullptr)
{
    unsigned run = 0;
    const int t0 = ch ? 2 : 0, nn = n * n;
    for(int pos = 0; pos < nn; pos++) {
        const int c = coef[scan ? scan[pos] : zigzag(n, pos)];
        if(!c) {
            run++;
            continue;
        }
        const unsigned level = (unsigned)(c < 0 ? -c : c) & 0xFFFF;
        sb_unary2(s, run, XEVE_HIP_CTX_RUN + t0, o);
        sb_unary2(s, level - 1, XEVE_HIP_CTX_LEVEL + t0, o);
        sb_bin_ep(s, c < 0, o);
        if(pos == nn - 1) break;
        run = 0, num_sig--;
        sb_bin(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0), num_sig == 0, o);
        if(num_sig == 0) break;
    }
}
// the head of an intra CU (xeve_rdo_bit_cnt_cu_intra*, xeve_mode.c:81-175): skip flag and pred_mode outside I slices, the mode as its rank among the most probable
XL void sb_intra_head(Sbac &s, const Params &P, const xeve_hip_intra_job &J, int rank)
{
    if(P.slice_type != 2) {
        sb_bin(s, XEVE_HIP_CTX_SKIP_FLAG + J.ctx_skip, 0);
        sb_bin(s, XEVE_HIP_CTX_PRED_MODE + J.ctx_pred_mode, 1);
    }
    sb_unary2(s, (unsigned)rank, XEVE_HIP_CTX_INTRA_DIR);
}
// xeve_eco_coef of an intra CU (xeve_eco.c:864-890, 1067-1089): one cbf per coded component (chroma first), then the levels
XL void sb_intra_coef(Sbac &s, const Params &P, const int nnz[3], const int16_t *cy, const int16_t *cu, const int16_t *cv, int run_y, int run_c, int n, int nc)
{
    if(run_c && P.idc) sb_bin(s, XEVE_HIP_CTX_CBF_CB, nnz[1] != 0), sb_bin(s, XEVE_HIP_CTX_CBF_CR, nnz[2] != 0);
    if(run_y) sb_bin(s, XEVE_HIP_CTX_CBF_LUMA, nnz[0] != 0);
    if(run_y && nnz[0]) sb_run_length(s, cy, n, nnz[0], 0);
    if(run_c && nnz[1]) sb_run_length(s, cu, nc, nnz[1], 1);
    if(run_c && nnz[2]) sb_run_length(s, cv, nc, nnz[2], 1);
}

// ---- rate tables of RDOQ (biari_no_bits, xeve_mode.c:315-324) ---------------------------------------------------------------------------------------------------
XL int32_t no_bits(const Params &P, int symbol, uint16_t cm)
{
    const unsigned mps = cm & 1;
    unsigned state = cm >> 1;
    state = ((unsigned)(symbol != 0) != mps) ? state : 512 - state;
    return P.entropy[state << 1];
}
XL void bit_est(const Params &P, const Sbac &s, Est &e)
{
    for(int b = 0; b < 2; b++) {
        e.cbf[0][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_CBF_LUMA]), e.cbf[1][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_CBF_CB]), e.cbf[2][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_CBF_CR]);
        for(int c = 0; c < 4; c++) e.run[c][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_RUN + c]), e.level[c][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_LEVEL + c]);
        for(int c = 0; c < 2; c++) e.last[c][b] = no_bits(P, b, s.ctx[XEVE_HIP_CTX_LAST + c]);
    }
}

// ---- transforms, quantisation, reconstruction (square n x n, n = 2, 4, 8) ------------------------------------------------------------------------------------------
// xeve_trans (xeve_tq.c:396-404): rows then columns through the partial-butterfly matrices, s32 between the passes, truncating store
XL void trans(int16_t *coef, int n, int log2n, int bd)
{
    int32_t tb[64];
    for(int j = 0; j < n; j++)
        for(int k = 0; k < n; k++) {
            int64_t acc = 0;
            for(int x = 0; x < n; x++) acc += (int64_t)dct_m(n, k, x) * coef[j * n + x];
            tb[k * n + j] = (int32_t)acc;
        }
    const int     shift = (log2n - 1 + bd - 8) + (log2n + 6);
    const int64_t add = (int64_t)1 << (shift - 1);
    for(int j = 0; j < n; j++)
        for(int k = 0; k < n; k++) {
            int64_t acc = 0;
            for(int x = 0; x < n; x++) acc += (int64_t)dct_m(n, k, x) * tb[j * n + x];
            coef[k * n + j] = (int16_t)((acc + add) >> shift);
        }
}
// xeve_itrans (xeve_itdq.c:435-440): columns (clipped to s32), then rows with the rounding shift (clipped to s16)
XL void itrans(int16_t *coef, int n, int bd)
{
    int32_t tb[64];
    for(int j = 0; j < n; j++)
        for(int x = 0; x < n; x++) {
            int64_t acc = 0;
            for(int k = 0; k < n; k++) acc += (int64_t)dct_m(n, k, x) * coef[k * n + j];
            acc = acc < INT32_MIN ? INT32_MIN : acc > INT32_MAX ? INT32_MAX : acc;
            tb[j * n + x] = (int32_t)acc;
        }
    const int     shift = 7 + (12 - (bd - 8));
    const int64_t add = (int64_t)1 << (shift - 1);
    for(int j = 0; j < n; j++)
        for(int x = 0; x < n; x++) {
            int64_t acc = 0;
            for(int k = 0; k < n; k++) acc += (int64_t)dct_m(n, k, x) * tb[k * n + j];
            acc = (acc + add) >> shift;
            coef[j * n + x] = (int16_t)(acc < -32768 ? -32768 : acc > 32767 ? 32767 : acc);
        }
}
XL int iabs(int v) { return v < 0 ? -v : v; }
// the zero-block pre-test of RDOQ (xeve_tq.c:666-699), square blocks
XL int zero_test(const int16_t *coef, int n, int log2n, int qp, int scale, int is_intra_slice, int bd)
{
    const int     shift = 14 + (15 - bd - log2n) + qp / 6;
    const int64_t thr = ((int64_t)1 << shift) - ((int64_t)(is_intra_slice ? 201 : 153) << (shift - 9));
    for(int i = 0; i < n * n; i++)
        if((int64_t)iabs(coef[i]) * scale >= thr) return 1;
    return 0;
}
// get_ic_rate_cost_rl (xeve_tq.c:425-456): s32 rate arithmetic as the reference; c = 0 luma / 2 chroma
XL int64_t rl_cost(unsigned abs_level, int run_nonzero, int c, int64_t lambda, const Est &e)
{
    uint32_t rate;
    if(abs_level == 0) rate = (uint32_t)e.run[c + run_nonzero][1];
    else {
        rate = 32768u + (uint32_t)e.run[c + run_nonzero][0];
        if(abs_level == 1) rate += (uint32_t)e.level[c][0];
        else rate += (uint32_t)e.level[c][1] + (uint32_t)e.level[c + 1][1] * (abs_level - 2) + (uint32_t)e.level[c + 1][0];
    }
    return (int64_t)(int32_t)rate * lambda;
}
// xeve_rdoq_run_length_cc (xeve_tq.c:497-649), square n x n; comp 0 Y / 1 U / 2 V selects the cbf rates of an intra CU; returns the number of levels kept
XL int rdoq(int16_t *coef, int n, int log2n, int qp, int q_value, double d_lambda, int comp, int bd, int64_t es, const Est &est)
{
    const int     q_bits = 14 + (15 - bd - log2n) + qp / 6, nn = n * n, c = comp ? 2 : 0, ctx_last = comp ? 1 : 0;
    const int64_t lambda = (int64_t)(d_lambda * (double)(1 << 15) + 0.5);
    int64_t ld[64];
    int16_t mx[64], out[64];
    int64_t block_uncoded = 0;
    int     sum_all = 0, nnz = 0;
    for(int p = 0; p < nn; p++) {
        const int     v = coef[zigzag(n, p)];
        const int64_t t = (int64_t)iabs(v) * q_value, cap = (int64_t)INT32_MAX - ((int64_t)1 << (q_bits - 1));
        const int64_t level_double = (int)(t < cap ? t : cap);
        uint32_t m = (uint32_t)(level_double >> q_bits);
        if(!((level_double - ((int64_t)m << q_bits)) < ((int64_t)1 << (q_bits - 1)))) m++;
        const int64_t err = (level_double * es) >> 20;
        block_uncoded += err * err;
        ld[p] = level_double, mx[p] = (int16_t)(v > 0 ? (int16_t)m : -(int16_t)m);
        sum_all += (int)m;
        out[p] = 0;
    }
    if(sum_all != 0) {
        int64_t  best_cost = block_uncoded + (int64_t)est.cbf[comp][0] * lambda, base_cost = block_uncoded + (int64_t)est.cbf[comp][1] * lambda;
        uint32_t run = 0, best_last = 0;
        for(int p = 0; p < nn; p++) {
            const uint32_t max_abs = (uint32_t)iabs(mx[p]);
            const int64_t  e1 = (ld[p] * es) >> 20, uncoded = e1 * e1;
            int64_t  coded = uncoded + rl_cost(0, run != 0, c, lambda, est);
            uint32_t best = 0;
            const uint32_t lo = max_abs > 1 ? max_abs - 1 : 1;
            for(uint32_t a = max_abs; a >= lo; a--) { // get_coded_level_rl (xeve_tq.c:458-490)
                const int64_t d = ld[p] - ((int64_t)a << q_bits), e2 = (d * es) >> 20, cost = e2 * e2 + rl_cost(a, run != 0, c, lambda, est);
                if(cost < coded) best = a, coded = cost;
            }
            out[p] = (int16_t)(mx[p] < 0 ? -(int32_t)best : (int32_t)best);
            base_cost += coded - uncoded;
            if(best) {
                const int64_t cur_is_last = base_cost + (int64_t)est.last[ctx_last][1] * lambda;
                base_cost += (int64_t)est.last[ctx_last][0] * lambda;
                if(cur_is_last < best_cost) best_cost = cur_is_last, best_last = (uint32_t)p + 1;
                run = 0;
            }
            else run++;
        }
        for(int p = 0; p < nn; p++) {
            if((uint32_t)p < best_last) nnz += out[p] != 0;
            else out[p] = 0;
        }
    }
    for(int p = 0; p < nn; p++) coef[zigzag(n, p)] = out[p];
    return nnz;
}
// xeve_dquant (xeve_itdq.c:442-475), square blocks
XL void dquant(int16_t *coef, int n, int log2n, int scale, int bd)
{
    const int     shift = (uint8_t)(20 - 14 - (15 - bd - log2n));
    const int32_t offset = shift == 0 ? 0 : 1 << (shift - 1);
    for(int i = 0; i < n * n; i++) {
        int64_t lev = ((int64_t)coef[i] * scale + offset) >> shift;
        coef[i] = (int16_t)(lev < -32768 ? -32768 : lev > 32767 ? 32767 : lev);
    }
}
// one block through the residual chain: residual, DCT, zero pre-test, RDOQ, dequantisation, inverse DCT, reconstruction (the sum wraps to s16 before the clip,
// xeve_recon.c:34-57); coef receives the levels, rec the reconstruction; returns the number of levels
XL int chain(const Params &P, const Est &est, const pel *org, int s_org, const pel *pred, int n, int log2n, int comp, int16_t *coef, pel *rec)
{
    const int nn = n * n, maxv = (1 << P.bd) - 1;
    int16_t   tmp[64];
    for(int y = 0; y < n; y++)
        for(int x = 0; x < n; x++) coef[y * n + x] = (int16_t)((int)org[y * s_org + x] - (int)pred[y * n + x]);
    trans(coef, n, log2n, P.bd);
    int nnz = 0;
    if(zero_test(coef, n, log2n, P.qp[comp], P.q_scale[comp], P.slice_type == 2, P.bd))
        nnz = rdoq(coef, n, log2n, P.qp[comp], P.q_scale[comp], P.lambda[comp], comp, P.bd, P.err_scale[comp], est);
    else
        for(int i = 0; i < nn; i++) coef[i] = 0;
    if(nnz) {
        for(int i = 0; i < nn; i++) tmp[i] = coef[i];
        dquant(tmp, n, log2n, P.dq_scale[comp], P.bd);
        itrans(tmp, n, P.bd);
    }
    for(int i = 0; i < nn; i++) {
        const int16_t t = nnz ? (int16_t)(tmp[i] + pred[i]) : pred[i];
        rec[i] = (pel)(t < 0 ? 0 : t > maxv ? maxv : t);
    }
    return nnz;
}
XL int64_t ssd(const pel *a, const pel *b, int s_b, int n, int bd)
{   // xeve_ssd_16b (xeve_sad.c:275-297): the shift per sample
    const int sh = (bd - 8) * 2;
    int64_t   acc = 0;
    for(int y = 0; y < n; y++)
        for(int x = 0; x < n; x++) {
            const int d = (int)a[y * n + x] - (int)b[y * s_b + x];
            acc += (d * d) >> sh;
        }
    return acc;
}
// xeve_had of a 4x4 or 8x8 block (xeve_sad.c:419-602, 1051-1135): unnormalised Hadamard of the difference, the DC term >> 2, rounding per size, >> (bd - 8)
XL int satd(const pel *org, int s_org, const pel *cur, int n, int bd)
{
    int t[64];
    for(int y = 0; y < n; y++)
        for(int x = 0; x < n; x++) t[y * n + x] = (int)org[y * s_org + x] - (int)cur[y * n + x];
    for(int pass = 0; pass < 2; pass++) {
        const int st = pass ? n : 1, line = pass ? 1 : n;
        for(int r = 0; r < n; r++)
            for(int len = 1; len < n; len <<= 1)
                for(int base = 0; base < n; base += 2 * len)
                    for(int i = base; i < base + len; i++) {
                        const int a = t[r * line + i * st], b = t[r * line + (i + len) * st];
                        t[r * line + i * st] = a + b, t[r * line + (i + len) * st] = a - b;
                    }
    }
    int sum = iabs(t[0]) >> 2;
    for(int i = 1; i < n * n; i++) sum += iabs(t[i]);
    sum = n == 4 ? (sum + 1) >> 1 : (sum + 2) >> 2;
    return sum >> (bd - 8);
}

// ---- neighbours and predictors ----------------------------------------------------------------------------------------------------------------------------------
#define XL_COD(m) (((m) >> 31) & 1u)
#define XL_IF(m)  (((m) >> 15) & 1u)
// xeve_get_nbr (xeve_ipred.c:32-105) for one component; up / left point at element 0 of arrays with one element in front (the corner) and cw + ch behind
XL void get_nbr(const Params &P, const uint32_t *map_scu, const uint8_t *map_tidx, const pel *src, int s, int x, int y, int c, int cw, int ch, pel *left, pel *up)
{
    const int ws = P.idc <= 2, hs = P.idc <= 1;
    int scuw = c == 0 ? cw >> 2 : cw >> (2 - ws), scuh = c == 0 ? ch >> 2 : ch >> (2 - hs), unit = c == 0 ? 4 : 2;
    if(c && P.idc == 2) scuh *= 2;
    if(c && P.idc == 3) unit *= 2;
    const int x_scu = x >> 2, y_scu = y >> 2, scup = y_scu * P.w_scu + x_scu; // (x, y: the CU's LUMA position)
    const pel grey = (pel)(1 << (P.bd - 1));
    auto usable = [&](int u) { return XL_COD(map_scu[u]) && (!P.cip || XL_IF(map_scu[u])) && map_tidx[scup] == map_tidx[u]; };
    up[-1] = (x_scu > 0 && y_scu > 0 && usable(scup - P.w_scu - 1)) ? src[-s - 1] : grey;
    left[-1] = up[-1];
    for(int i = 0; i < scuw + scuh; i++) {
        const bool ok = y_scu > 0 && x_scu + i < P.w_scu && usable(scup - P.w_scu + i);
        for(int k = 0; k < unit; k++) up[i * unit + k] = ok ? src[-s + i * unit + k] : grey;
    }
    for(int i = 0; i < scuh + scuw; i++) {
        const bool ok = x_scu > 0 && y_scu + i < P.h_scu && usable(scup - 1 + i * P.w_scu);
        for(int k = 0; k < unit; k++) left[i * unit + k] = ok ? src[(long)(i * unit + k) * s - 1] : grey;
    }
}
// xeve_ipred / xeve_ipred_uv (xeve_ipred.c:107-227): DC 0, horizontal 1, vertical 2, up-left diagonal 3, up-right average 4; dst dense n x n
XL void ipred(const pel *left, const pel *up, pel *dst, int ipm, int n, int log2n)
{
    if(ipm == 0) {
        int dc = 0;
        for(int i = 0; i < n; i++) dc += left[i] + up[i];
        dc = (dc + n) >> (log2n + 1);
        for(int i = 0; i < n * n; i++) dst[i] = (pel)dc;
        return;
    }
    for(int i = 0; i < n; i++)
        for(int j = 0; j < n; j++) {
            int v;
            if(ipm == 1) v = left[i];
            else if(ipm == 2) v = up[j];
            else if(ipm == 3) v = i > j ? left[i - j - 1] : (i == j ? up[-1] : up[j - i - 1]);
            else v = (up[i + j + 1] + left[i + j + 1]) >> 1;
            dst[i * n + j] = (pel)v;
        }
}

// ---- pintra_analyze_cu (xeve_pintra.c:544-698) of one CU of size 2^LOG2 (LOG2 = 2, 3) -----------------------------------------------------------------------------
// org / mod: the picture's planes at sample (0, 0); coef / rec: dense blocks (Y n*n, U, V nc*nc each); best = core->s_temp_best.  Baseline, rdo_dbk_switch 0, no delta QP.
template <int LOG2>
XL void intra_cu(const Params &P, const pel *const org[3], const pel *const mod[3], const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, const Sbac &entry,
                 const xeve_hip_intra_job &J, xeve_hip_intra_result &R, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, pel *rec_y, pel *rec_u, pel *rec_v, Sbac &best)
{
    constexpr int N = 1 << LOG2, N0 = N * N;
    const int idc = P.idc, ws = idc <= 2, hs = idc <= 1, nc = idc ? N >> ws : 0, lc = LOG2 - ws, x = J.x, y = J.y; // (4:2:0 and 4:4:4: square chroma blocks)
    pel nb[3][2][2 * N + 2];
    get_nbr(P, map_scu, map_tidx, mod[0] + (long)y * P.s_mod_l + x, P.s_mod_l, x, y, 0, N, N, nb[0][0] + 1, nb[0][1] + 1);
    for(int c = 1; c < 3 && idc; c++)
        get_nbr(P, map_scu, map_tidx, mod[c] + (long)(y >> hs) * P.s_mod_c + (x >> ws), P.s_mod_c, x, y, c, nc, N >> hs, nb[c][0] + 1, nb[c][1] + 1);
    int l = 0, u = 0; // xeve_get_mpm (xeve_ipred.c:229-252)
    {
        const int x_scu = x >> 2, y_scu = y >> 2, scup = y_scu * P.w_scu + x_scu;
        if(x_scu > 0 && XL_IF(map_scu[scup - 1]) && XL_COD(map_scu[scup - 1]) && map_tidx[scup] == map_tidx[scup - 1]) l = map_ipm[scup - 1] + 1;
        if(y_scu > 0 && XL_IF(map_scu[scup - P.w_scu]) && XL_COD(map_scu[scup - P.w_scu]) && map_tidx[scup] == map_tidx[scup - P.w_scu]) u = map_ipm[scup - P.w_scu] + 1;
    }
    const pel *oy = org[0] + (long)y * P.s_org_l + x;
    // make_ipred_list (:308-374): SATD + sqrt(lambda) * bits of the mode, the insertion-sorted list of the five cheapest, cut from the tail while the SATD alone
    // exceeds 1.2 x the SATD of the best inter prediction
    pel      pred[5][N0];
    int      list[5];
    double   cand_cost[5];
    uint32_t cand_satd[5];
    for(int i = 0; i < 5; i++) list[i] = 0, cand_cost[i] = 1.7e+308, cand_satd[i] = 0xFFFFFFFFu;
    for(int m = 0; m < 5; m++) {
        ipred(nb[0][0] + 1, nb[0][1] + 1, pred[m], m, N, LOG2);
        const uint32_t sa = (uint32_t)satd(oy, P.s_org_l, pred[m], N, P.bd);
        Sbac s = entry;
        sb_bit_reset(s);
        sb_unary2(s, (unsigned)mpm_rank(l, u, m), XEVE_HIP_CTX_INTRA_DIR); // xeve_rdo_bit_cnt_intra_dir (xeve_mode.c:136-139)
        const double cost = (double)sa + (double)(int)sb_bits(s) * P.sqrt_lambda0;
        int shift = 0;
        while(shift < 5 && cost < cand_cost[4 - shift]) shift++;
        if(shift) {
            for(int j = 1; j < shift; j++) list[5 - j] = list[4 - j], cand_cost[5 - j] = cand_cost[4 - j], cand_satd[5 - j] = cand_satd[4 - j];
            list[5 - shift] = m, cand_cost[5 - shift] = cost, cand_satd[5 - shift] = sa;
        }
    }
    int pred_cnt = 5;
    for(int i = 4; i >= 1; i--) {
        if((double)cand_satd[i] > (double)J.inter_satd * (1.2)) pred_cnt--;
        else break;
    }
    // the luma RDO of the list (:604-637; pintra_residue_rdo mode 0): every candidate starts from the entry coder state
    Est est;
    bit_est(P, entry, est); // core->rdoq_est_* of mode_coding_unit (xeve_mode.c:792)
    double  cost_best = 1.7e+308;
    int     best_ipd = -1, nnz_best[3] = {0, 0, 0};
    int32_t best_dist_y = 0, best_dist_c = 0;
    int16_t ct[N0];
    pel     rt[N0];
    for(int j = 0; j < pred_cnt; j++) {
        const int m = list[j];
        const int nnz = chain(P, est, oy, P.s_org_l, pred[m], N, LOG2, 0, ct, rt);
        Sbac s = entry;
        sb_bit_reset(s);
        sb_intra_head(s, P, J, mpm_rank(l, u, m));
        const int nz[3] = {nnz, 0, 0};
        sb_intra_coef(s, P, nz, ct, nullptr, nullptr, 1, 0, N, 0); // xeve_rdo_bit_cnt_cu_intra_luma (xeve_mode.c:81-117)
        double cost = 0;
        cost += (double)ssd(rt, oy, P.s_org_l, N, P.bd);
        const int32_t dist = (int32_t)cost;
        cost += (double)(int)sb_bits(s) * P.lambda[0];
        if(cost < cost_best) {
            cost_best = cost, best_dist_y = dist, best_ipd = m, nnz_best[0] = nnz;
            for(int i = 0; i < N0; i++) coef_y[i] = ct[i], rec_y[i] = rt[i];
        }
    }
    // chroma with the luma winner's mode (:639-658; mode 1): the bits of that call never reach an output (cost_t is discarded, :643-646)
    if(idc) {
        double cost = 0;
        pel    pc[N0];
        for(int c = 1; c < 3; c++) {
            const pel *oc = org[c] + (long)(y >> hs) * P.s_org_c + (x >> ws);
            int16_t *cf = c == 1 ? coef_u : coef_v;
            pel     *rc = c == 1 ? rec_u : rec_v;
            ipred(nb[c][0] + 1, nb[c][1] + 1, pc, best_ipd, nc, lc);
            nnz_best[c] = chain(P, est, oc, P.s_org_c, pc, nc, lc, c, cf, rc);
            cost += P.wgt[c - 1] * (double)ssd(rc, oc, P.s_org_c, nc, P.bd);
        }
        best_dist_c = (int32_t)cost;
    }
    // the CU's cost (:679-695): the whole syntax from the entry state; its exit state is core->s_temp_best
    Sbac s = entry;
    sb_bit_reset(s);
    sb_intra_head(s, P, J, mpm_rank(l, u, best_ipd));
    sb_intra_coef(s, P, nnz_best, coef_y, coef_u, coef_v, 1, 1, N, nc);
    double cost = (double)(int)sb_bits(s) * P.lambda[0];
    cost += best_dist_y;
    if(idc) cost += best_dist_c;
    best = s;
    R.cost = cost, R.dist_cu = best_dist_y + (idc ? best_dist_c : 0), R.nnz[0] = nnz_best[0], R.nnz[1] = nnz_best[1], R.nnz[2] = nnz_best[2], R.pred_cnt = pred_cnt;
    R.ipm[0] = (int8_t)best_ipd, R.ipm[1] = (int8_t)(idc ? best_ipd : 0), R.pad_[0] = R.pad_[1] = 0;
}
} // namespace xl
