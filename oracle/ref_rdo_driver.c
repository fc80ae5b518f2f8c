/*
 * oracle/ref_rdo_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_rdo.so).
 * Calls the reference's static pinter_residue_rdo (src_base/xeve_pinter.c:906-1336) -- prediction, residual, transform + RDOQ,
 * reconstruction, CABAC bit counting and the coded-block-flag decision of ONE inter CU candidate -- on caller-supplied pictures
 * and coder state.  xeve_pinter.c and xeve_mode.c are compiled in place (their static functions pinter_residue_rdo, pinter_mc,
 * xeve_rdoq_bit_est are needed); everything else (xeve_mc, xeve_sub_block_tq, xeve_itdq, xeve_recon, xeve_eco_coef, the SAD /
 * MC / transform tables) is linked from oracle/_ref/libxeveb_ref.so.  The context is set up the way xeve_platform_init_func,
 * xeve_pic_prepare and mode_coding_unit leave it for a Baseline encoder with rdoq = 1, rdo_dbk_switch = 0, no delta QP.
 */
#include "xeve_pinter.c"
#include "xeve_mode.c"
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

typedef struct { pel *y, *u, *v; int poc, pad_; } drv_refpic;
typedef struct { int log2_cuw, log2_cuh, pic_w, pic_h, slice_type, num_refp[2], chroma_format_idc, bit_depth, tool_iqt, qp[3], pad_; double lambda[3], dist_chroma_weight[2]; } drv_rdo_params;
typedef struct { int x, y; s16 mv[2][2], mvd[2][2]; s8 refi[2]; u8 mvp_idx[2]; u8 dir_flag, ctx_skip, ctx_pred_mode, pad_; int sbac; } drv_rdo_job;
typedef struct { double cost; int nnz[3], pad_; s64 dist[2][3]; } drv_rdo_result;

static XEVE_CTX  *g_ctx;
static XEVE_CORE *g_core;
static int        g_simd;
void refdrv_set_simd(int on) { g_simd = on; }

void refdrv_residue_rdo(pel *org_y, pel *org_u, pel *org_v, int s_org_l, int s_org_c, const drv_refpic *refs, int s_l, int s_c, const drv_sbac *states,
                        const drv_rdo_params *p, const drv_rdo_job *job, drv_rdo_result *res, s16 *coef_y, s16 *coef_u, s16 *coef_v, drv_sbac *best)
{
    static XEVE_PIC   pic_o, pics[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static XEVE_REFP  refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static XEVE_SH    sh;
    if(!g_ctx) g_ctx = calloc(1, sizeof(*g_ctx)), g_core = calloc(1, sizeof(*g_core)), xeve_init_bits_est();
    XEVE_CTX  *ctx  = g_ctx;
    XEVE_CORE *core = g_core;
    XEVE_PINTER *pi = &ctx->pinter[0];
    const int ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    const int lw = p->log2_cuw, lh = p->log2_cuh;
    /* dispatch tables (xeve_platform_init_func, xeve_enc.c:722-825): the plain-C ones; refdrv_set_simd(1) lets the reference pick the tables for this
     * CPU instead (same results; used when the reference is timed as the CPU baseline) */
    if(g_simd) xeve_platform_init_func(ctx);
    else {
        xeve_func_sad = xeve_tbl_sad_16b, xeve_func_ssd = xeve_tbl_ssd_16b, xeve_func_diff = xeve_tbl_diff_16b, xeve_func_satd = xeve_tbl_satd_16b;
        xeve_func_mc_l = xeve_tbl_mc_l, xeve_func_mc_c = xeve_tbl_mc_c, xeve_func_average_no_clip = &xeve_average_16b_no_clip;
        xeve_func_txb = &xeve_tbl_txb, ctx->fn_itxb = &xeve_tbl_itxb;
    }
    ctx->fn_tq = xeve_sub_block_tq, ctx->fn_itdp = xeve_itdq, ctx->fn_recon = xeve_recon, ctx->fn_eco_coef = xeve_eco_coef;
    ctx->fn_rdoq_set_ctx_cc = xeve_rdoq_set_ctx_cc;
    ctx->param.tool_iqt = p->tool_iqt, ctx->param.codec_bit_depth = p->bit_depth, ctx->param.rdoq = 1, ctx->param.rdo_dbk_switch = 0;
    ctx->param.cs_w_shift = ws, ctx->param.cs_h_shift = hs;
    xeve_init_err_scale(ctx);
    ctx->sps.bit_depth_luma_minus8 = ctx->sps.bit_depth_chroma_minus8 = p->bit_depth - 8;
    ctx->sps.chroma_format_idc = p->chroma_format_idc, ctx->sps.tool_admvp = 0, ctx->pps.cu_qp_delta_enabled_flag = 0;
    ctx->w = p->pic_w, ctx->h = p->pic_h;
    ctx->rpm.num_refp[0] = p->num_refp[0], ctx->rpm.num_refp[1] = p->num_refp[1];
    ctx->sh = &sh, sh.slice_type = p->slice_type;
    /* pictures */
    pic_o.y = org_y, pic_o.u = org_u, pic_o.v = org_v, pic_o.s_l = s_org_l, pic_o.s_c = s_org_c;
    int nref = p->num_refp[0] > p->num_refp[1] ? p->num_refp[0] : p->num_refp[1];
    for(int r = 0; r < nref; r++)
        for(int l = 0; l < REFP_NUM; l++) {
            const drv_refpic *s = &refs[r * 2 + l];
            pics[r][l].y = s->y, pics[r][l].u = s->u, pics[r][l].v = s->v, pics[r][l].s_l = s_l, pics[r][l].s_c = s_c, pics[r][l].poc = s->poc;
            refp[r][l].pic = &pics[r][l], refp[r][l].poc = s->poc;
        }
    pi->pic_o = &pic_o, pi->o[Y_C] = org_y, pi->o[U_C] = org_u, pi->o[V_C] = org_v, pi->s_o[Y_C] = s_org_l, pi->s_o[U_C] = pi->s_o[V_C] = s_org_c;
    pi->refp = refp, pi->slice_type = p->slice_type, pi->fn_mc = pinter_mc;
    /* the candidate */
    const int pidx = job->dir_flag ? PRED_DIR : (job->refi[0] >= 0 ? (job->refi[1] >= 0 ? PRED_BI : PRED_L0) : PRED_L1);
    for(int l = 0; l < REFP_NUM; l++) {
        pi->refi[pidx][l] = job->refi[l], pi->mvp_idx[pidx][l] = job->mvp_idx[l];
        for(int d = 0; d < MV_D; d++) pi->mv[pidx][l][d] = job->mv[l][d], pi->mvd[pidx][l][d] = job->mvd[l][d];
    }
    /* the core as mode_coding_unit leaves it (xeve_mode.c:760-800) */
    core->ctx = ctx, core->thread_cnt = 0, core->log2_cuw = lw, core->log2_cuh = lh, core->cuw = 1 << lw, core->cuh = 1 << lh;
    core->qp_y = p->qp[0], core->qp_u = p->qp[1], core->qp_v = p->qp[2];
    for(int c = 0; c < 3; c++) core->lambda[c] = p->lambda[c];
    core->dist_chroma_weight[0] = p->dist_chroma_weight[0], core->dist_chroma_weight[1] = p->dist_chroma_weight[1];
    core->tree_cons.changed = 0, core->tree_cons.tree_type = TREE_LC, core->tree_cons.mode_cons = eAll;
    core->ctx_flags[CNID_SKIP_FLAG] = job->ctx_skip, core->ctx_flags[CNID_PRED_MODE] = job->ctx_pred_mode;
    core->bs_temp.pdata[1] = &core->s_temp_run;
    core->cost_best = MAX_COST;
    to_ref(&core->s_curr_best[lw - 2][lh - 2], &states[job->sbac]);
    xeve_rdoq_bit_est(&core->s_curr_best[lw - 2][lh - 2], core); /* xeve_mode.c:792 */

    res->cost = pinter_residue_rdo(ctx, core, job->x, job->y, lw, lh, pi->pred[pidx], pi->coef[pidx], pidx, pi->mvp_idx[pidx]);
    for(int c = 0; c < N_C; c++) res->nnz[c] = core->nnz[c];
    memset(res->dist, 0, sizeof(res->dist)); /* locals of the reference function: not observable */
    memcpy(coef_y, pi->coef[pidx][Y_C], sizeof(s16) << (lw + lh));
    if(p->chroma_format_idc) {
        memcpy(coef_u, pi->coef[pidx][U_C], sizeof(s16) << (lw + lh - ws - hs));
        memcpy(coef_v, pi->coef[pidx][V_C], sizeof(s16) << (lw + lh - ws - hs));
    }
    from_ref(best, &core->s_temp_best);
}


/* xeve_analyze_skip (static, xeve_pinter.c:1337-1530).  The merge candidates come from the reference's own xeve_get_motion (xeve_util.c:526-573),
 * untouched: the neighbour maps it reads (left, up, up-right unit of ctx->map_mv; the collocated unit of refp[0][list].map_mv) are loaded with the
 * caller's four vectors per list, all neighbours available -- Baseline always pairs them with reference index 0. */
typedef struct { int x, y; s16 mvp[2][4][2]; s8 refi_pred[2][4]; int ncand, sbac; u8 ctx_skip, pad_[3]; } drv_skip_job;
typedef struct { double cost; s64 best_ssd; int idx0, idx1; s16 mv[2][2]; s8 refi[2]; s8 pad_[6]; } drv_skip_result;

void refdrv_analyze_skip(pel *org_y, pel *org_u, pel *org_v, int s_org_l, int s_org_c, const drv_refpic *refs, int s_l, int s_c, const drv_sbac *states,
                         const drv_rdo_params *p, const drv_skip_job *job, drv_skip_result *res, pel *pred_y, pel *pred_u, pel *pred_v, drv_sbac *best)
{
    /* reuse the context set-up of refdrv_residue_rdo through a dummy candidate, then run the skip analysis */
    drv_rdo_job dj;
    drv_rdo_result dr;
    static s16 c0[MAX_CU_DIM], c1[MAX_CU_DIM], c2[MAX_CU_DIM];
    drv_sbac   tmp;
    memset(&dj, 0, sizeof(dj));
    dj.x = job->x, dj.y = job->y, dj.refi[0] = 0, dj.refi[1] = -1, dj.sbac = job->sbac, dj.ctx_skip = job->ctx_skip;
    refdrv_residue_rdo(org_y, org_u, org_v, s_org_l, s_org_c, refs, s_l, s_c, states, p, &dj, &dr, c0, c1, c2, &tmp);
    XEVE_CTX  *ctx  = g_ctx;
    XEVE_CORE *core = g_core;
    XEVE_PINTER *pi = &ctx->pinter[0];
    const int lw = p->log2_cuw, lh = p->log2_cuh, ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    ctx->slice_type = p->slice_type, pi->skip_merge_cand_num = job->ncand;
    core->cost_best = MAX_COST, core->scup = 0, core->avail_cu = 0;
    core->ctx_flags[CNID_SKIP_FLAG] = job->ctx_skip;
    to_ref(&core->s_curr_best[lw - 2][lh - 2], &states[job->sbac]);
    memset(&core->s_temp_best, 0, sizeof(core->s_temp_best));
    {
        enum { W_SCU = 32, SCUP = W_SCU + 1 };
        static s16 map_mv[2 * W_SCU + 32][REFP_NUM][MV_D], col_mv[REFP_NUM][1024][REFP_NUM][MV_D];
        static s8  map_refi[2 * W_SCU + 32][REFP_NUM];
        const int  cuw_scu = (1 << lw) >> MIN_CU_LOG2;
        ctx->map_mv = map_mv, ctx->map_refi = map_refi, ctx->w_scu = W_SCU;
        core->scup = SCUP, core->avail_cu = AVAIL_LE | AVAIL_UP | AVAIL_UP_RI;
        for(int l = 0; l < REFP_NUM; l++) {
            for(int d = 0; d < MV_D; d++) {
                map_mv[SCUP - 1][l][d] = job->mvp[l][0][d], map_mv[SCUP - W_SCU][l][d] = job->mvp[l][1][d];
                map_mv[SCUP - W_SCU + cuw_scu][l][d] = job->mvp[l][2][d], col_mv[l][SCUP][0][d] = job->mvp[l][3][d];
            }
            pi->refp[0][l].map_mv = col_mv[l];
        }
    }
    res->cost = xeve_analyze_skip(ctx, core, job->x, job->y, lw, lh);
    res->best_ssd = pi->best_ssd, res->idx0 = pi->mvp_idx[PRED_SKIP][REFP_0], res->idx1 = pi->mvp_idx[PRED_SKIP][REFP_1];
    for(int l = 0; l < 2; l++) res->mv[l][0] = pi->mv[PRED_SKIP][l][MV_X], res->mv[l][1] = pi->mv[PRED_SKIP][l][MV_Y], res->refi[l] = pi->refi[PRED_SKIP][l];
    memset(res->pad_, 0, sizeof(res->pad_));
    memcpy(pred_y, pi->pred[PRED_SKIP][0][Y_C], sizeof(pel) << (lw + lh));
    if(p->chroma_format_idc) memcpy(pred_u, pi->pred[PRED_SKIP][0][U_C], sizeof(pel) << (lw + lh - ws - hs)), memcpy(pred_v, pi->pred[PRED_SKIP][0][V_C], sizeof(pel) << (lw + lh - ws - hs));
    from_ref(best, &core->s_temp_best);
}


/* xeve_pinter_analyze_cu (xeve_pinter.c:1839-2047) = ctx->fn_pinter_analyze_cu: the whole inter analysis of one CU, with the reference's own
 * xeve_get_motion / xeve_get_mv_dir reading neighbour maps loaded with the caller's vectors (as in refdrv_analyze_skip), pinter_me_epzs as
 * pi->fn_me (me_complexity 1, the caller's sub-pel pattern sizes), get_range_ipel deriving the ranges from gop_size and the POCs. */
typedef struct { unsigned lambda_mv; int refi_bits, extra_bits, bi, faststep, max_search_range, range_recentre, min_clip[2], max_clip[2], reserved; } drv_me_params;
typedef struct { unsigned lambda_mv; int refi_bits, extra_bits, bi, hpel_cnt, qpel_cnt; } drv_spel_params;
typedef struct { drv_rdo_params rdo; drv_me_params me; drv_spel_params spel; int refi_bits[2][8], range_recentre[2][8], max_cand, poc, col_list_poc0, pad_; double skip_th; } drv_inter_params;
typedef struct { int x, y; s16 mvp[2][4][2]; s16 mv_col[2]; int sbac; u8 ctx_skip, ctx_pred_mode, pad_[2]; } drv_inter_job;
typedef struct { double cost, cost_inter[5]; int cu_mode, best_idx; s16 mv[2][2], mvd[2][2]; s8 refi[2]; u8 mvp_idx[2]; int nnz[3], pad_[2]; } drv_inter_result;

void refdrv_pinter_analyze_cu(pel *org_y, pel *org_u, pel *org_v, int s_org_l, int s_org_c, const drv_refpic *refs, int s_l, int s_c, const drv_sbac *states,
                              const drv_inter_params *P, int gop_size, const drv_inter_job *job, drv_inter_result *res, s16 *coef_y, s16 *coef_u, s16 *coef_v,
                              pel *rec_y, pel *rec_u, pel *rec_v, drv_sbac *next_best)
{
    const drv_rdo_params *p = &P->rdo;
    /* context, pictures, neighbour maps: as for the skip analysis */
    drv_skip_job    sj;
    drv_skip_result sr;
    drv_sbac        tmp;
    static pel      t0[MAX_CU_DIM], t1[MAX_CU_DIM], t2[MAX_CU_DIM];
    memset(&sj, 0, sizeof(sj));
    sj.x = job->x, sj.y = job->y, memcpy(sj.mvp, job->mvp, sizeof(sj.mvp)), sj.ncand = P->max_cand, sj.sbac = job->sbac, sj.ctx_skip = job->ctx_skip;
    refdrv_analyze_skip(org_y, org_u, org_v, s_org_l, s_org_c, refs, s_l, s_c, states, p, &sj, &sr, t0, t1, t2, &tmp);
    XEVE_CTX    *ctx  = g_ctx;
    XEVE_CORE   *core = g_core;
    XEVE_PINTER *pi   = &ctx->pinter[0];
    const int lw = p->log2_cuw, lh = p->log2_cuh, ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    /* motion search set-up (pinter_init_lcu / pinter_set_complexity, :1716-1771, :2049-2140) */
    pi->fn_me = pinter_me_epzs;
    pi->min_clip[MV_X] = P->me.min_clip[0], pi->min_clip[MV_Y] = P->me.min_clip[1], pi->max_clip[MV_X] = P->me.max_clip[0], pi->max_clip[MV_Y] = P->me.max_clip[1];
    pi->lambda_mv = P->me.lambda_mv, pi->max_search_range = P->me.max_search_range, pi->gop_size = gop_size, pi->poc = P->poc;
    pi->me_complexity = (P->me.reserved & 1) ? 2 : 1; /* bit 0: me_raster on */
    pi->search_pattern_hpel = tbl_search_pattern_hpel_partial, pi->search_pattern_hpel_cnt = P->spel.hpel_cnt;
    pi->search_pattern_qpel = tbl_search_pattern_qpel_8point, pi->search_pattern_qpel_cnt = P->spel.qpel_cnt;
    pi->me_level = P->spel.hpel_cnt == 0 ? ME_LEV_IPEL : (P->spel.qpel_cnt > 0 ? ME_LEV_QPEL : ME_LEV_HPEL), pi->mc_l_coeff = xeve_tbl_mc_l_coeff;
    memset(pi->mot_bits, 0, sizeof(pi->mot_bits));
    /* temporal direct (xeve_get_mv_dir): POCs and the collocated vector at the CU's bottom-right unit */
    ctx->poc.poc_val = P->poc, ctx->h_scu = 64, ctx->param.skip_th = P->skip_th;
    static u32 list_poc[XEVE_MAX_NUM_REF_PICS];
    list_poc[0] = P->col_list_poc0, pi->refp[0][REFP_1].list_poc = list_poc;
    {
        const int corner = core->scup + ((1 << (lw - MIN_CU_LOG2)) - 1) + ((1 << (lh - MIN_CU_LOG2)) - 1) * ctx->w_scu;
        pi->refp[0][REFP_1].map_mv[corner][0][MV_X] = job->mv_col[0], pi->refp[0][REFP_1].map_mv[corner][0][MV_Y] = job->mv_col[1];
    }
    core->ctx_flags[CNID_PRED_MODE] = job->ctx_pred_mode, core->cu_mode = MODE_INTRA, core->cost_best = MAX_COST;
    memset(&core->s_next_best[lw - 2][lh - 2], 0, sizeof(XEVE_SBAC));
    static XEVE_MODE mi;
    static s16       coef[N_C][MAX_CU_DIM];
    pel             *rec[N_C];
    int              s_rec[N_C];
    memset(&mi, 0, sizeof(mi)), memset(coef, 0, sizeof(coef));
    res->cost = xeve_pinter_analyze_cu(ctx, core, job->x, job->y, lw, lh, &mi, coef, rec, s_rec);
    for(int m = 0; m < 5; m++) res->cost_inter[m] = 0; /* locals of the reference function: not observable */
    res->cu_mode = core->cu_mode, res->best_idx = -1;
    for(int l = 0; l < 2; l++) {
        res->refi[l] = mi.refi[l], res->mvp_idx[l] = mi.mvp_idx[l];
        for(int d = 0; d < 2; d++) res->mv[l][d] = mi.mv[l][d], res->mvd[l][d] = mi.mvd[l][d];
    }
    for(int c = 0; c < N_C; c++) res->nnz[c] = core->nnz[c];
    res->pad_[0] = res->pad_[1] = 0;
    const int n0 = 1 << (lw + lh), n1 = n0 >> (ws + hs);
    memcpy(coef_y, coef[Y_C], sizeof(s16) * n0), memcpy(rec_y, rec[Y_C], sizeof(pel) * n0);
    if(p->chroma_format_idc) {
        memcpy(coef_u, coef[U_C], sizeof(s16) * n1), memcpy(coef_v, coef[V_C], sizeof(s16) * n1);
        memcpy(rec_u, rec[U_C], sizeof(pel) * n1), memcpy(rec_v, rec[V_C], sizeof(pel) * n1);
    }
    from_ref(next_best, &core->s_next_best[lw - 2][lh - 2]);
}


/* the candidates of a drv_inter_job from the encoder's per-unit maps, by the reference's exported xeve_get_avail_inter + xeve_get_motion, and the
 * collocated vector xeve_get_mv_dir would read */
void refdrv_inter_candidates(u32 *map_scu, u8 *map_tidx, s16 (*map_mv)[REFP_NUM][MV_D], s16 (*col0)[REFP_NUM][MV_D], s16 (*col1)[REFP_NUM][MV_D], int w_scu, int h_scu,
                             int log2_cuw, int log2_cuh, int slice_type, drv_inter_job *job)
{
    static XEVE_REFP refp[1][REFP_NUM];
    static s8        map_refi[1][REFP_NUM]; /* not read by the Baseline derivation */
    const int x_scu = job->x >> MIN_CU_LOG2, y_scu = job->y >> MIN_CU_LOG2, scup = y_scu * w_scu + x_scu, cuw = 1 << log2_cuw, cuh = 1 << log2_cuh;
    refp[0][REFP_0].map_mv = col0, refp[0][REFP_1].map_mv = col1;
    const u16 avail = xeve_get_avail_inter(x_scu, y_scu, w_scu, h_scu, scup, cuw, cuh, map_scu, map_tidx);
    memset(job->mvp, 0, sizeof(job->mvp)), job->mv_col[0] = job->mv_col[1] = 0;
    for(int l = 0; l <= (slice_type == SLICE_B ? 1 : 0); l++) {
        s8 refi[MAX_NUM_MVP];
        xeve_get_motion(scup, l, map_refi, map_mv, refp, cuw, cuh, w_scu, avail, refi, job->mvp[l]);
    }
    if(slice_type == SLICE_B) {
        const int corner = scup + ((1 << (log2_cuw - MIN_CU_LOG2)) - 1) + ((1 << (log2_cuh - MIN_CU_LOG2)) - 1) * w_scu; /* xeve_pinter.c:1543 */
        job->mv_col[0] = col1[corner][0][MV_X], job->mv_col[1] = col1[corner][0][MV_Y];
    }
}
