/* oracle/ref_intra_driver.c -- TEST INFRASTRUCTURE ONLY: runs the reference's own intra analysis of one CU -- the static pintra_analyze_cu
 * (src_base/xeve_pintra.c:544-698, = ctx->fn_pintra_analyze_cu) with everything it calls (xeve_get_nbr, xeve_ipred, xeve_get_mpm, make_ipred_list,
 * pintra_residue_rdo, xeve_rdo_bit_cnt_cu_intra*, xeve_sub_block_tq, xeve_itdq, xeve_recon) -- on flat inputs, so that oracle/xeve_oracle.c's
 * xo_pintra_analyze_cu can be pinned against it.  The reference's files are compiled IN PLACE (the static functions are reachable only that way);
 * nothing of them is copied here.  The flat structs mirror oracle/xeve_oracle.h. */
#include <stdlib.h>
#include <string.h>
#include "xeve_pintra.c"
#include "xeve_mode.c" /* xeve_rdoq_bit_est / xeve_init_bits_est are static there */
#include "xeve_eco.h"
#include "xeve_tq.h"
#include "xeve_itdq.h"

enum { C_SKIP = 0, C_PRED_MODE = 2, C_DIRECT = 5, C_INTER_DIR = 6, C_REFI = 8, C_MVP_IDX = 10, C_MVD = 13, C_CBF_ALL = 14,
       C_CBF_LUMA = 15, C_CBF_CB = 16, C_CBF_CR = 17, C_RUN = 18, C_LAST = 42, C_LEVEL = 44, C_INTRA_DIR = 68, C_SPLIT_CU = 70, C_DELTA_QP = 71, C_N = 72 };
typedef struct { u32 range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter; u16 ctx[C_N]; } drv_sbac;
#define MAP(F)                                                                          \
    F(skip_flag, C_SKIP, 2) F(pred_mode, C_PRED_MODE, 3) F(direct_mode_flag, C_DIRECT, 1) \
    F(inter_dir, C_INTER_DIR, 2) F(refi, C_REFI, 2) F(mvp_idx, C_MVP_IDX, 3) F(mvd, C_MVD, 1) \
    F(cbf_all, C_CBF_ALL, 1) F(cbf_luma, C_CBF_LUMA, 1) F(cbf_cb, C_CBF_CB, 1) F(cbf_cr, C_CBF_CR, 1) \
    F(run, C_RUN, 24) F(last, C_LAST, 2) F(level, C_LEVEL, 24) \
    F(intra_dir, C_INTRA_DIR, 2) F(split_cu_flag, C_SPLIT_CU, 1) F(delta_qp, C_DELTA_QP, 1)
static void to_ref(XEVE_SBAC *d, const drv_sbac *s)
{
    xeve_sbac_reset(d, 0, 0, 0);
    d->range = s->range, d->code = s->code, d->code_bits = s->code_bits, d->stacked_ff = s->stacked_ff, d->stacked_zero = s->stacked_zero;
    d->pending_byte = s->pending_byte, d->is_pending_byte = s->is_pending_byte, d->bitcounter = s->bitcounter, d->bin_counter = s->bin_counter;
    d->is_bitcount = 1;
#define F(name, at, n) memcpy(d->ctx.name, s->ctx + at, 2 * n);
    MAP(F)
#undef F
}
static void from_ref(drv_sbac *d, const XEVE_SBAC *s)
{
    d->range = s->range, d->code = s->code, d->code_bits = s->code_bits, d->stacked_ff = s->stacked_ff, d->stacked_zero = s->stacked_zero;
    d->pending_byte = s->pending_byte, d->is_pending_byte = s->is_pending_byte, d->bitcounter = s->bitcounter, d->bin_counter = s->bin_counter;
#define F(name, at, n) memcpy(d->ctx + at, s->ctx.name, 2 * n);
    MAP(F)
#undef F
}

typedef struct { int log2_cuw, log2_cuh, w_scu, h_scu, slice_type, chroma_format_idc, bit_depth, tool_iqt, constrained_intra_pred, qp[3]; double lambda[3], sqrt_lambda0, dist_chroma_weight[2]; } drv_intra_params;
typedef struct { int x, y; u32 inter_satd; int sbac, pic; u8 ctx_skip, ctx_pred_mode, pad_[2]; } drv_intra_job;
typedef struct { double cost; int dist_cu, nnz[3], pred_cnt; s8 ipm[2], pad_[2]; } drv_intra_result;

void refdrv_pintra_analyze_cu(pel *org_y, pel *org_u, pel *org_v, int s_org_l, int s_org_c, pel *mod_y, pel *mod_u, pel *mod_v, int s_mod_l, int s_mod_c,
                              u32 *map_scu, s8 *map_ipm, u8 *map_tidx, const drv_sbac *states, const drv_intra_params *p, const drv_intra_job *job,
                              drv_intra_result *res, s16 *coef_y, s16 *coef_u, s16 *coef_v, pel *rec_y, pel *rec_u, pel *rec_v, drv_sbac *best)
{
    static XEVE_CTX  *ctx;
    static XEVE_CORE *core;
    static XEVE_SH    sh;
    static XEVE_MODE  mi;
    static s16(*coef)[MAX_CU_DIM];
    if(!ctx) ctx = calloc(1, sizeof(*ctx)), core = calloc(1, sizeof(*core)), coef = calloc(N_C, sizeof(*coef)), xeve_init_bits_est();
    XEVE_PINTRA *pi = &ctx->pintra[0];
    const int ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    const int lw = p->log2_cuw, lh = p->log2_cuh;
    /* the plain-C dispatch tables (xeve_platform_init_func, xeve_enc.c:722-825) */
    xeve_func_sad = xeve_tbl_sad_16b, xeve_func_ssd = xeve_tbl_ssd_16b, xeve_func_diff = xeve_tbl_diff_16b, xeve_func_satd = xeve_tbl_satd_16b;
    xeve_func_txb = &xeve_tbl_txb, ctx->fn_itxb = &xeve_tbl_itxb;
    ctx->fn_tq = xeve_sub_block_tq, ctx->fn_itdp = xeve_itdq, ctx->fn_recon = xeve_recon, ctx->fn_eco_coef = xeve_eco_coef;
    ctx->fn_rdoq_set_ctx_cc = xeve_rdoq_set_ctx_cc, ctx->fn_mode_rdo_bit_cnt_intra_dir = xeve_rdo_bit_cnt_intra_dir;
    ctx->fn_rdo_intra_ext = NULL, ctx->fn_rdo_intra_ext_c = NULL;
    ctx->param.tool_iqt = p->tool_iqt, ctx->param.codec_bit_depth = p->bit_depth, ctx->param.rdoq = 1, ctx->param.rdo_dbk_switch = 0;
    ctx->param.cs_w_shift = ws, ctx->param.cs_h_shift = hs;
    xeve_init_err_scale(ctx);
    ctx->sps.bit_depth_luma_minus8 = ctx->sps.bit_depth_chroma_minus8 = p->bit_depth - 8;
    ctx->sps.chroma_format_idc = p->chroma_format_idc, ctx->sps.tool_admvp = 0, ctx->pps.cu_qp_delta_enabled_flag = 0;
    ctx->pps.constrained_intra_pred_flag = p->constrained_intra_pred;
    ctx->w_scu = p->w_scu, ctx->h_scu = p->h_scu, ctx->w = p->w_scu << MIN_CU_LOG2, ctx->h = p->h_scu << MIN_CU_LOG2;
    ctx->map_scu = map_scu, ctx->map_ipm = map_ipm, ctx->map_tidx = map_tidx;
    ctx->sh = &sh, sh.slice_type = p->slice_type, ctx->slice_type = p->slice_type;
    pi->o[Y_C] = org_y, pi->o[U_C] = org_u, pi->o[V_C] = org_v, pi->s_o[Y_C] = s_org_l, pi->s_o[U_C] = pi->s_o[V_C] = s_org_c;
    pi->m[Y_C] = mod_y, pi->m[U_C] = mod_u, pi->m[V_C] = mod_v, pi->s_m[Y_C] = s_mod_l, pi->s_m[U_C] = pi->s_m[V_C] = s_mod_c;
    pi->slice_type = p->slice_type;
    /* the core as mode_cu_init / mode_check_intra leave it (xeve_mode.c:760-800, 1244-1276) */
    core->ctx = ctx, core->thread_cnt = 0, core->log2_cuw = lw, core->log2_cuh = lh, core->cuw = 1 << lw, core->cuh = 1 << lh;
    core->x_scu = PEL2SCU(job->x), core->y_scu = PEL2SCU(job->y), core->scup = core->y_scu * ctx->w_scu + core->x_scu;
    core->qp_y = p->qp[0], core->qp_u = p->qp[1], core->qp_v = p->qp[2];
    for(int c = 0; c < 3; c++) core->lambda[c] = p->lambda[c];
    core->sqrt_lambda[0] = p->sqrt_lambda0;
    core->dist_chroma_weight[0] = p->dist_chroma_weight[0], core->dist_chroma_weight[1] = p->dist_chroma_weight[1];
    core->tree_cons.changed = 0, core->tree_cons.tree_type = TREE_LC, core->tree_cons.mode_cons = eAll;
    core->ctx_flags[CNID_SKIP_FLAG] = job->ctx_skip, core->ctx_flags[CNID_PRED_MODE] = job->ctx_pred_mode;
    core->bs_temp.pdata[1] = &core->s_temp_run;
    core->inter_satd = job->inter_satd;
    core->avail_lr   = 0;
    core->avail_cu   = xeve_get_avail_intra(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, lw, lh, ctx->map_scu, ctx->map_tidx);
    to_ref(&core->s_curr_best[lw - 2][lh - 2], &states[job->sbac]);
    xeve_rdoq_bit_est(&core->s_curr_best[lw - 2][lh - 2], core); /* xeve_mode.c:792 */

    pel *rec[N_C];
    int  s_rec[N_C];
    res->cost = pintra_analyze_cu(ctx, core, job->x, job->y, lw, lh, &mi, coef, rec, s_rec);
    res->dist_cu = core->dist_cu, res->ipm[0] = core->ipm[0], res->ipm[1] = p->chroma_format_idc ? core->ipm[1] : 0, res->pad_[0] = res->pad_[1] = 0;
    for(int c = 0; c < N_C; c++) res->nnz[c] = core->nnz[c];
    res->pred_cnt = 0; /* a local of the reference function: not observable */
    memcpy(coef_y, coef[Y_C], sizeof(s16) << (lw + lh)), memcpy(rec_y, rec[Y_C], sizeof(pel) << (lw + lh));
    if(p->chroma_format_idc) {
        memcpy(coef_u, coef[U_C], sizeof(s16) << (lw + lh - ws - hs)), memcpy(coef_v, coef[V_C], sizeof(s16) << (lw + lh - ws - hs));
        memcpy(rec_u, rec[U_C], sizeof(pel) << (lw + lh - ws - hs)), memcpy(rec_v, rec[V_C], sizeof(pel) << (lw + lh - ws - hs));
    }
    from_ref(best, &core->s_temp_best);
}

/* the bare pieces, for unit pins */
void refdrv_get_nbr(int x, int y, int cuw, int cuh, pel *src, int s_src, u32 *map_scu, u8 *map_tidx, int w_scu, int h_scu, int ch, int constrained, int bit_depth,
                    int chroma_format_idc, int log2_cuw_l, int log2_cuh_l, pel *left, pel *up, int n)
{
    static pel nb[N_C][N_REF][MAX_CU_SIZE * 3];
    const int  ws = XEVE_GET_CHROMA_W_SHIFT(chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(chroma_format_idc);
    const int  x_scu = PEL2SCU(ch ? x << ws : x), y_scu = PEL2SCU(ch ? y << hs : y), scup = y_scu * w_scu + x_scu;
    const u16  avail = xeve_get_avail_intra(x_scu, y_scu, w_scu, h_scu, scup, log2_cuw_l, log2_cuh_l, map_scu, map_tidx);
    xeve_get_nbr(x, y, cuw, cuh, src, s_src, avail, nb, scup, map_scu, w_scu, h_scu, ch, constrained, map_tidx, bit_depth, chroma_format_idc);
    memcpy(left - 1, nb[ch][0] + 1, sizeof(pel) * (n + 1)), memcpy(up - 1, nb[ch][1] + cuh - 1, sizeof(pel) * (n + 1));
}
void refdrv_ipred(pel *left, pel *up, pel *dst, int ipm, int w, int h, int chroma)
{
    if(chroma) xeve_ipred_uv(left, up, NULL, 0, dst, ipm, ipm, w, h);
    else xeve_ipred(left, up, NULL, 0, dst, ipm, w, h);
}
