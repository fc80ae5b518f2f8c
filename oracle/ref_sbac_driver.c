/*
 * oracle/ref_sbac_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_sbac.so).
 * Flat wrapper around the reference's CABAC bit counting for an inter CU: SBAC_LOAD + xeve_sbac_bit_reset +
 * xeve_rdo_bit_cnt_cu_inter / _cu_inter_comp / _cu_skip + xeve_get_bit_number (src_base/xeve_mode.c:39-295), exactly as
 * pinter_residue_rdo strings them together (src_base/xeve_pinter.c:1112-1131).  XEVE_CTX / XEVE_CORE / XEVE_SBAC come
 * from the reference's own headers; the flat structs below mirror oracle/xeve_oracle.h (xo_sbac, xo_cu_bits_*).
 * Also wraps the static xeve_rdoq_bit_est (xeve_mode.c:326-372) and the entropy_bits table (xeve_init_bits_est, :304-313).
 */
#include <stdlib.h>
#include <string.h>
#include "xeve_mode.c" /* the reference's file, compiled in place: xeve_rdoq_bit_est and entropy_bits are static there */
#include "xeve_eco.h"

enum { C_SKIP = 0, C_PRED_MODE = 2, C_DIRECT = 5, C_INTER_DIR = 6, C_REFI = 8, C_MVP_IDX = 10, C_MVD = 13, C_CBF_ALL = 14,
       C_CBF_LUMA = 15, C_CBF_CB = 16, C_CBF_CR = 17, C_RUN = 18, C_LAST = 42, C_LEVEL = 44, C_INTRA_DIR = 68, C_SPLIT_CU = 70, C_DELTA_QP = 71, C_N = 72 };
typedef struct { u32 range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter; u16 ctx[C_N]; } drv_sbac;
typedef struct { int log2_cuw, log2_cuh, slice_type, num_refp[2], cm_init, chroma_format_idc; } drv_params;
typedef struct { int coef_off[3], nnz[3], sbac; s16 mvd[2][2]; s8 refi[2]; u8 mvp_idx[2]; u8 mode, dir_flag, ctx_skip, ctx_pred_mode; } drv_job;

#define MAP(F)                                                                          \
    F(skip_flag, C_SKIP, 2) F(pred_mode, C_PRED_MODE, 3) F(direct_mode_flag, C_DIRECT, 1) \
    F(inter_dir, C_INTER_DIR, 2) F(refi, C_REFI, 2) F(mvp_idx, C_MVP_IDX, 3) F(mvd, C_MVD, 1) \
    F(cbf_all, C_CBF_ALL, 1) F(cbf_luma, C_CBF_LUMA, 1) F(cbf_cb, C_CBF_CB, 1) F(cbf_cr, C_CBF_CR, 1) \
    F(run, C_RUN, 24) F(last, C_LAST, 2) F(level, C_LEVEL, 24) \
    F(intra_dir, C_INTRA_DIR, 2) F(split_cu_flag, C_SPLIT_CU, 1) F(delta_qp, C_DELTA_QP, 1)

static void to_ref(XEVE_SBAC *d, const drv_sbac *s, int cm_init)
{
    xeve_sbac_reset(d, 0, 0, cm_init); /* every other model = PROB_INIT; sets sps_cm_init_flag */
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

unsigned refdrv_cu_bits(const drv_sbac *in, drv_sbac *out, const drv_params *p, const drv_job *j, const s16 *coef)
{
    static __thread XEVE_CTX  *ctx; /* per thread: oracle/cpu_bench.c calls this from its worker threads */
    static __thread XEVE_CORE *core;
    static __thread s16(*cbuf)[MAX_CU_DIM];
    if(!ctx) ctx = calloc(1, sizeof(*ctx)), core = calloc(1, sizeof(*core)), cbuf = calloc(N_C, sizeof(*cbuf));
    int ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    ctx->fn_eco_coef = xeve_eco_coef;
    ctx->sps.tool_admvp = 0, ctx->sps.chroma_format_idc = p->chroma_format_idc;
    ctx->pps.cu_qp_delta_enabled_flag = 0;
    ctx->param.cs_w_shift = ws, ctx->param.cs_h_shift = hs;
    ctx->rpm.num_refp[0] = p->num_refp[0], ctx->rpm.num_refp[1] = p->num_refp[1];
    core->ctx = ctx, core->thread_cnt = 0;
    core->log2_cuw = p->log2_cuw, core->log2_cuh = p->log2_cuh;
    core->tree_cons.changed = 0, core->tree_cons.tree_type = TREE_LC, core->tree_cons.mode_cons = eAll;
    core->ctx_flags[CNID_SKIP_FLAG] = j->ctx_skip, core->ctx_flags[CNID_PRED_MODE] = j->ctx_pred_mode;
    core->bs_temp.pdata[1] = &core->s_temp_run;
    for(int c = 0; c < N_C; c++) {
        int n = 1 << (p->log2_cuw + p->log2_cuh - (c ? ws + hs : 0));
        memset(core->nnz_sub[c], 0, sizeof(core->nnz_sub[c]));
        core->nnz_sub[c][0] = core->nnz[c] = j->nnz[c];
        memcpy(cbuf[c], coef + j->coef_off[c], n * sizeof(s16));
    }
    to_ref(&core->s_temp_run, in + j->sbac, p->cm_init); /* SBAC_LOAD */
    if(j->mode == 5) { /* ctx->fn_eco_coef alone; flags in dir_flag: 1 intra, 2 b_no_cbf, 4/8/16 run Y/U/V, 32 no bit_reset */
        const int f = j->dir_flag;
        if(!(f & 32)) xeve_sbac_bit_reset(&core->s_temp_run);
        xeve_eco_coef(ctx, core, &core->bs_temp, cbuf, (f & 1) ? MODE_INTRA : MODE_INTER, 0, (f & 2) ? 1 : 0, (f >> 2) & 7);
        if(out) from_ref(out, &core->s_temp_run);
        return xeve_get_bit_number(&core->s_temp_run);
    }
    xeve_sbac_bit_reset(&core->s_temp_run);
    s8  refi[REFP_NUM] = {j->refi[0], j->refi[1]};
    s16 mvd[REFP_NUM][MV_D] = {{j->mvd[0][0], j->mvd[0][1]}, {j->mvd[1][0], j->mvd[1][1]}};
    u8  mvp_idx[REFP_NUM] = {j->mvp_idx[0], j->mvp_idx[1]};
    if(j->mode >= 7 && j->mode <= 9) { /* intra CU: job.mvp_idx[0] = the unary index mpm[ipm]; realised as mode 0 of a list whose entry 0 is that index */
        static __thread u8 mpm[IPD_CNT_B];
        mpm[0] = j->mvp_idx[0], core->mpm_b_list = mpm, core->ipm[0] = 0;
        ctx->fn_mode_rdo_bit_cnt_intra_dir = xeve_rdo_bit_cnt_intra_dir, ctx->fn_rdo_intra_ext = NULL, ctx->fn_rdo_intra_ext_c = NULL;
        if(j->mode == 7) xeve_rdo_bit_cnt_cu_intra(ctx, core, p->slice_type, 0, cbuf);
        else if(j->mode == 8) {
            core->nnz_sub[U_C][0] = core->nnz_sub[V_C][0] = 0; /* (xeve_sub_block_tq run for luma alone leaves them cleared) */
            xeve_rdo_bit_cnt_cu_intra_luma(ctx, core, p->slice_type, 0, cbuf);
        }
        else xeve_rdo_bit_cnt_intra_dir(ctx, core, 0);
    }
    else if(j->mode == 4) xeve_rdo_bit_cnt_cu_skip(ctx, core, p->slice_type, 0, j->mvp_idx[0], j->mvp_idx[1], 0, 0);
    else if(j->mode == 0)
        xeve_rdo_bit_cnt_cu_inter(ctx, core, p->slice_type, 0, refi, mvd, cbuf, j->dir_flag ? PRED_DIR : (refi[0] >= 0 ? (refi[1] >= 0 ? PRED_BI : PRED_L0) : PRED_L1),
                                  mvp_idx, 0, 0, NULL);
    else xeve_rdo_bit_cnt_cu_inter_comp(core, cbuf, j->mode - 1, 0, ctx, core->tree_cons);
    if(out) from_ref(out, &core->s_temp_run);
    return xeve_get_bit_number(&core->s_temp_run);
}

/* bare pieces, for unit pins */
void refdrv_sbac_bin(drv_sbac *s, int ci, unsigned bin, int ep)
{
    XEVE_SBAC r;
    XEVE_BSW  bs;
    memset(&bs, 0, sizeof(bs));
    to_ref(&r, s, 0);
    bs.pdata[1] = &r;
    u16 *m = (u16 *)&r.ctx; /* only the mapped fields matter: pick the model through the same MAP */
    (void)m;
    SBAC_CTX_MODEL *model = NULL;
#define F(name, at, n) if(ci >= at && ci < at + n) model = r.ctx.name + (ci - at);
    MAP(F)
#undef F
    if(ep) sbac_encode_bin_ep(bin, &r, &bs);
    else xeve_sbac_encode_bin(bin, &r, model, &bs);
    from_ref(s, &r);
}
void refdrv_run_length_cc(drv_sbac *s, const s16 *coef, int log2w, int log2h, int num_sig, int ch, int cm_init)
{
    XEVE_SBAC r;
    XEVE_BSW  bs;
    s16       tmp[MAX_TR_DIM];
    memset(&bs, 0, sizeof(bs));
    to_ref(&r, s, cm_init);
    bs.pdata[1] = &r;
    memcpy(tmp, coef, sizeof(s16) << (log2w + log2h));
    xeve_eco_run_length_cc(NULL, &bs, tmp, log2w, log2h, num_sig, ch);
    from_ref(s, &r);
}

/* xeve_rdoq_bit_est on a flat state; out = cbf_all[2], cbf_luma[2], cbf_cb[2], cbf_cr[2], run[24][2], level[24][2], last[2][2] */
void refdrv_rdoq_bit_est(const drv_sbac *s, int *out)
{
    static __thread XEVE_CORE *core;
    XEVE_SBAC r;
    if(!core) core = calloc(1, sizeof(*core)), xeve_init_bits_est();
    to_ref(&r, s, 0);
    xeve_rdoq_bit_est(&r, core);
    memcpy(out, core->rdoq_est_cbf_all, 8), memcpy(out + 2, core->rdoq_est_cbf_luma, 8);
    memcpy(out + 4, core->rdoq_est_cbf_cb, 8), memcpy(out + 6, core->rdoq_est_cbf_cr, 8);
    memcpy(out + 8, core->rdoq_est_run, sizeof(core->rdoq_est_run));
    memcpy(out + 56, core->rdoq_est_level, sizeof(core->rdoq_est_level));
    memcpy(out + 104, core->rdoq_est_last, sizeof(core->rdoq_est_last));
}
void refdrv_entropy_bits(int *out)
{
    xeve_init_bits_est();
    memcpy(out, entropy_bits, sizeof(entropy_bits));
}
