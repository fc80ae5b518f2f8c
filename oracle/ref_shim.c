/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE (integration parity test only; built only where
 * /root/reference exists, output oracle/_ref/libxeve_hip_shim.so).
 *
 * An LD_PRELOAD interposer that does, without editing the reference, exactly what the ~10-line
 * "if (gpu)" branch INTEGRATION.md proposes for xeve_platform_init_func would do
 * (reference: src_base/xeve_enc.c:722-779): after the reference has installed its SIMD tables it
 * overwrites them with the HIP dispatch tables of libxeve_hip.so, and routes ctx->fn_recon
 * (xeve_enc.c:822, xeve_type.h:978) to xeve_recon_blk_hip.  The reference's callers
 * (xeve_pinter.c, xeve_mode.c, xeve_pintra.c, ...) then run UNCHANGED on top of the HIP kernels.
 *
 * It is compiled against the reference's own headers (for XEVE_CTX's layout) and resolves
 * libxeve_hip.so with dlopen at run time: path in $XEVE_HIP_LIB, device ordinal in $XEVE_HIP_DEVICE.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "xeve_type.h"
#include "xeve_mc.h"

static void (*hip_recon_blk)(s16 *, pel *, int, int, int, int, pel *, int);
static unsigned long long (*hip_table_calls)(void);

static void shim_recon(XEVE_CTX *ctx, XEVE_CORE *core, s16 *coef, pel *pred, int is_coef, int cuw, int cuh, int s_rec, pel *rec, int bit_depth)
{
    hip_recon_blk(coef, pred, is_coef, cuw, cuh, s_rec, rec, bit_depth);
}

/* XEVE_HIP_SHIM_DF=1: also the in-loop deblocking and the reference-picture padding run on the GPU -- ctx->fn_loop_filter (xeve_loop_filter,
 * xeve_enc.c:2355-2415) and ctx->fn_picbuf_expand (xeve_pic_expand, xeve_enc.c:808,1274) are routed to xeve_hip_deblock_host /
 * xeve_hip_picbuf_expand_host.  The adapter hands over what xeve_loop_filter would read from the context and leaves the side effects
 * the reference's filter leaves (COD bits set, map_unrefined_mv = map_mv, slice offsets copied into the picture). */
typedef struct { int w, h, w_scu, h_scu, log2_max_cuwh, bit_depth_luma, bit_depth_chroma, chroma_format_idc, qp_u_offset, qp_v_offset, qp_chroma[2][100]; } hip_df_params;
static int (*hip_deblock_host)(pel *, pel *, pel *, int, int, int, int, const u32 *, const u32 *, const u8 *, const s8 *, const s16 *, const hip_df_params *);
static int (*hip_expand_host)(pel *, pel *, pel *, int, int, int, int, int, int, int, int, int);
static const char *(*hip_err)(void);
static unsigned long long df_calls, pad_calls;

static int shim_loop_filter(XEVE_CTX *ctx, XEVE_CORE *core)
{
    if(!ctx->sh->deblocking_filter_on) return XEVE_OK;
    XEVE_PIC *pic = PIC_MODE(ctx);
    hip_df_params p;
    const int bc = ctx->sps.bit_depth_chroma_minus8;
    p.w = ctx->w, p.h = ctx->h, p.w_scu = ctx->w_scu, p.h_scu = ctx->h_scu, p.log2_max_cuwh = ctx->log2_max_cuwh;
    p.bit_depth_luma = ctx->sps.bit_depth_luma_minus8 + 8, p.bit_depth_chroma = bc + 8, p.chroma_format_idc = ctx->sps.chroma_format_idc;
    p.qp_u_offset = ctx->sh->qp_u_offset, p.qp_v_offset = ctx->sh->qp_v_offset;
    for(int c = 0; c < 2; c++)
        for(int i = 0; i < 100; i++) p.qp_chroma[c][i] = i <= 57 + 6 * bc ? ctx->qp_chroma_dynamic_ext[c][i] : 0;
    for(u32 i = 0; i < ctx->f_scu; i++) /* xeve_deblock (xeve_df.c:545-556) */
        if(!MCU_GET_DMVRF(ctx->map_scu[i])) memcpy(ctx->map_unrefined_mv[i], ctx->map_mv[i], sizeof(ctx->map_mv[i]));
    if(hip_deblock_host(pic->y, pic->u, pic->v, pic->s_l, pic->s_c, pic->pad_l, pic->pad_c, ctx->map_scu, ctx->map_cu_mode, ctx->map_tidx, (const s8 *)ctx->map_refi,
                        (const s16 *)ctx->map_unrefined_mv, &p) != 0) {
        fprintf(stderr, "[xeve_hip_shim] deblock: %s\n", hip_err());
        abort();
    }
    for(u32 i = 0; i < ctx->f_scu; i++) MCU_SET_COD(ctx->map_scu[i]);
    pic->pic_deblock_alpha_offset = ctx->sh->sh_deblock_alpha_offset, pic->pic_deblock_beta_offset = ctx->sh->sh_deblock_beta_offset;
    pic->pic_qp_u_offset = ctx->sh->qp_u_offset, pic->pic_qp_v_offset = ctx->sh->qp_v_offset;
    df_calls++;
    return XEVE_OK;
}

static void shim_pic_expand(XEVE_CTX *ctx, XEVE_PIC *pic)
{
    if(hip_expand_host(pic->y, pic->u, pic->v, pic->s_l, pic->s_c, pic->w_l, pic->h_l, pic->w_c, pic->h_c, pic->pad_l, pic->pad_c, ctx->sps.chroma_format_idc) != 0) {
        fprintf(stderr, "[xeve_hip_shim] picbuf_expand: %s\n", hip_err());
        abort();
    }
    pad_calls++;
}

/* XEVE_HIP_SHIM_MC=1: the CU motion-compensation driver -- pi->fn_mc (pinter_mc -> xeve_mc, xeve_pinter.c:2058-2085,2106) -- runs on the GPU as a
 * whole (clip, per-list interpolation of Y / U / V, identical-motion shortcut, bi-prediction average): xeve_hip_mc_cu_host. */
typedef struct { const pel *y, *u, *v; int poc, pad_; } hip_refpic;
typedef struct { int x, y; s16 mv[2][2]; s8 refi[2]; s8 pad_[2]; } hip_mc_job;
static int (*hip_mc_cu_host)(const hip_refpic *, int, int, int, int, int, int, int, int, const hip_mc_job *, int, int, int, int, int, const void *, const void *, pel *, pel *, pel *);
static unsigned long long mc_calls;

static void shim_mc(XEVE_CTX *ctx, XEVE_CORE *core, int x, int y, int w, int h, s8 refi[REFP_NUM], s16 (*mv)[MV_D], XEVE_REFP (*refp)[REFP_NUM],
                    pel pred[REFP_NUM][N_C][MAX_CU_DIM], int poc_c, int apply_dmvr, s16 dmvr_mv[MAX_CU_CNT_IN_LCU][REFP_NUM][MV_D])
{
    hip_refpic tab[2 * XEVE_MAX_NUM_REF_PICS];
    hip_mc_job j;
    const int n0 = ctx->rpm.num_refp[REFP_0], n1 = ctx->rpm.num_refp[REFP_1], nmax = n0 > n1 ? n0 : n1;
    XEVE_PIC *any = NULL;
    memset(tab, 0, sizeof(tab));
    for(int r = 0; r < nmax; r++)
        for(int l = 0; l < REFP_NUM; l++) {
            XEVE_PIC *p = r < ctx->rpm.num_refp[l] ? refp[r][l].pic : NULL;
            if(!p) continue;
            tab[r * 2 + l].y = p->y, tab[r * 2 + l].u = p->u, tab[r * 2 + l].v = p->v, tab[r * 2 + l].poc = p->poc;
            any = p;
        }
    j.x = x, j.y = y, j.refi[0] = refi[REFP_0], j.refi[1] = refi[REFP_1], j.pad_[0] = j.pad_[1] = 0;
    for(int l = 0; l < REFP_NUM; l++) j.mv[l][0] = mv[l][MV_X], j.mv[l][1] = mv[l][MV_Y];
    if(hip_mc_cu_host(tab, n0, n1, any->s_l, any->s_c, any->pad_l, any->pad_c, ctx->w, ctx->h, &j, w, h, ctx->sps.bit_depth_luma_minus8 + 8,
                      ctx->sps.bit_depth_chroma_minus8 + 8, ctx->sps.chroma_format_idc, xeve_tbl_mc_l_coeff, xeve_tbl_mc_c_coeff, pred[0][Y_C], pred[0][U_C],
                      pred[0][V_C]) != 0) {
        fprintf(stderr, "[xeve_hip_shim] mc: %s\n", hip_err());
        abort();
    }
    mc_calls++;
}

/* XEVE_HIP_SHIM_ME=1: the per-list motion search runs on the GPU -- pi->fn_me (pinter_me_epzs, xeve_pinter.c:699-869,2104) is routed to
 * xeve_hip_me_epzs_host.  The adapter passes what pinter_me_epzs reads from XEVE_PINTER (original and reference luma planes, org_bi,
 * lambda_mv, clip window, search ranges as get_range_ipel derives them (:122-129), sub-pel pattern sizes, the other list's mot_bits) and
 * applies its side effect on pi->mot_bits[lidx]. */
typedef struct { unsigned lambda_mv; int refi_bits, extra_bits, bi, faststep, max_search_range, range_recentre, min_clip[2], max_clip[2], reserved; int hpel_cnt, qpel_cnt; } hip_epzs_params;
typedef struct { int x, y, org_off; s16 mvp[2], mv_start[2]; } hip_epzs_job;
typedef struct { s16 mv[2]; unsigned cost; int beststep, best_mv_bits; } hip_me_result;
static int (*hip_me_epzs_host)(const pel *, int, const pel *, const pel *, int, int, int, const hip_epzs_job *, int, int, int, const void *, const hip_epzs_params *,
                               hip_me_result *);
static unsigned long long me_calls;

static u32 shim_me(XEVE_PINTER *pi, int x, int y, int log2_cuw, int log2_cuh, s8 *refi, int lidx, s16 mvp[MV_D], s16 mv[MV_D], int bi, int bit_depth_luma)
{
    const int ri = *refi, lidx_r = lidx == REFP_0 ? REFP_1 : REFP_0;
    XEVE_PIC *rp = pi->refp[ri][lidx].pic;
    hip_epzs_params p;
    hip_epzs_job    j;
    hip_me_result   r;
    const int offset = pi->gop_size >> 1; /* get_range_ipel (xeve_pinter.c:122-129) */
    p.lambda_mv = pi->lambda_mv, p.refi_bits = xeve_tbl_refi_bits[pi->num_refp][ri], p.extra_bits = bi ? pi->mot_bits[lidx_r] : 0, p.bi = bi, p.faststep = 3;
    p.reserved = (pi->me_complexity > 1 ? 1 : 0) | (ri << 8); /* me_raster on; its step scales with refi + 1 */
    p.max_search_range = pi->max_search_range;
    p.range_recentre = XEVE_CLIP3(pi->max_search_range >> 2, pi->max_search_range,
                                  (pi->max_search_range * XEVE_ABS(pi->poc - (int)pi->refp[ri][lidx].poc) + offset) / pi->gop_size);
    p.min_clip[0] = pi->min_clip[MV_X], p.min_clip[1] = pi->min_clip[MV_Y], p.max_clip[0] = pi->max_clip[MV_X], p.max_clip[1] = pi->max_clip[MV_Y];
    p.hpel_cnt = pi->me_level > ME_LEV_IPEL ? pi->search_pattern_hpel_cnt : 0, p.qpel_cnt = pi->me_level > ME_LEV_HPEL ? pi->search_pattern_qpel_cnt : 0;
    j.x = x, j.y = y, j.org_off = 0, j.mvp[0] = mvp[MV_X], j.mvp[1] = mvp[MV_Y], j.mv_start[0] = mv[MV_X], j.mv_start[1] = mv[MV_Y];
    if(hip_me_epzs_host(pi->o[Y_C], pi->s_o[Y_C], bi ? (const pel *)pi->org_bi : NULL, rp->y, rp->s_l, rp->pad_l, rp->h_l, &j, log2_cuw, log2_cuh, bit_depth_luma,
                        pi->mc_l_coeff, &p, &r) != 0) {
        fprintf(stderr, "[xeve_hip_shim] me: %s\n", hip_err());
        abort();
    }
    mv[MV_X] = r.mv[0], mv[MV_Y] = r.mv[1];
    if(r.best_mv_bits > 0) pi->mot_bits[lidx] = r.best_mv_bits;
    me_calls++;
    return r.cost;
}

/* XEVE_HIP_SHIM_TQ=1: transform + quantisation (zero pre-test + RDOQ, the quantiser every preset configures) and dequantisation + inverse
 * transform of every transform block run on the GPU -- ctx->fn_tq (xeve_sub_block_tq, xeve_tq.c:750-864) and ctx->fn_itdp (xeve_itdq,
 * xeve_itdq.c:499-580) are replaced by the same per-component loops around xeve_hip_tq_nnz_host / xeve_hip_itdq_host, with core->rdoq_est_*
 * (filled by the reference's xeve_rdoq_bit_est) handed over as the estimate record.  Baseline: one transform block per component (CU <= 64). */
typedef struct { int cbf_all[2], cbf_luma[2], cbf_cb[2], cbf_cr[2], run[24][2], level[24][2], last[2][2]; } hip_est_full;
static int (*hip_tq_nnz_host)(s16 *, int, int, int, double, int, int, int, int, int, const hip_est_full *, int, int *);
static int (*hip_itdq_host)(s16 *, int, int, int, int);
static unsigned long long tq_calls, itdq_calls;

static int shim_tq(XEVE_CTX *ctx, XEVE_CORE *core, s16 coef[N_C][MAX_CU_DIM], int log2_cuw, int log2_cuh, int slice_type, int nnz[N_C], int is_intra, int run_stats)
{
    int run[N_C] = {run_stats & 1, (run_stats >> 1) & 1, (run_stats >> 2) & 1};
    const int ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift;
    const u8  qp[N_C] = {core->qp_y, core->qp_u, core->qp_v};
    hip_est_full e;
    if(log2_cuw > MAX_TR_LOG2 || log2_cuh > MAX_TR_LOG2) { fprintf(stderr, "[xeve_hip_shim] CU larger than one transform block\n"); abort(); }
    memcpy(e.cbf_all, core->rdoq_est_cbf_all, 8), memcpy(e.cbf_luma, core->rdoq_est_cbf_luma, 8), memcpy(e.cbf_cb, core->rdoq_est_cbf_cb, 8), memcpy(e.cbf_cr, core->rdoq_est_cbf_cr, 8);
    memcpy(e.run, core->rdoq_est_run, sizeof(e.run)), memcpy(e.level, core->rdoq_est_level, sizeof(e.level)), memcpy(e.last, core->rdoq_est_last, sizeof(e.last));
    xeve_mset(core->nnz_sub, 0, sizeof(int) * N_C * MAX_SUB_TB_NUM);
    if(!ctx->sps.chroma_format_idc) run[1] = run[2] = 0;
    for(int c = 0; c < N_C; c++) {
        nnz[c] = 0;
        if(!run[c]) continue;
        int n = 0;
        if(hip_tq_nnz_host(coef[c], log2_cuw - (c ? ws : 0), log2_cuh - (c ? hs : 0), qp[c], core->lambda[c], c, is_intra, slice_type == SLICE_I,
                           ctx->sps.bit_depth_luma_minus8 + 8, ctx->param.tool_iqt, &e, ctx->param.rdoq, &n) != 0) {
            fprintf(stderr, "[xeve_hip_shim] tq: %s\n", hip_err());
            abort();
        }
        core->nnz_sub[c][0] = nnz[c] = n;
        tq_calls++;
    }
    return nnz[Y_C] + nnz[U_C] + nnz[V_C];
}

static void shim_itdq(XEVE_CTX *ctx, XEVE_CORE *core, s16 coef[N_C][MAX_CU_DIM], int nnz_sub[N_C][MAX_SUB_TB_NUM])
{
    const int ws = XEVE_GET_CHROMA_W_SHIFT(ctx->sps.chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(ctx->sps.chroma_format_idc);
    const u8  qp[N_C] = {core->qp_y, core->qp_u, core->qp_v};
    if(core->log2_cuw > MAX_TR_LOG2 || core->log2_cuh > MAX_TR_LOG2) { fprintf(stderr, "[xeve_hip_shim] CU larger than one transform block\n"); abort(); }
    for(int c = 0; c < N_C; c++) {
        if((c && !ctx->sps.chroma_format_idc) || !nnz_sub[c][0]) continue;
        if(hip_itdq_host(coef[c], core->log2_cuw - (c ? ws : 0), core->log2_cuh - (c ? hs : 0), qp[c], ctx->sps.bit_depth_luma_minus8 + 8) != 0) {
            fprintf(stderr, "[xeve_hip_shim] itdq: %s\n", hip_err());
            abort();
        }
        itdq_calls++;
    }
}

/* XEVE_HIP_SHIM_ECO=1: while the encoder counts bits (sbac->is_bitcount, i.e. inside the RDO), the coefficient syntax of every CU -- cbf flags
 * and run / level / sign / last bins through the adaptive arithmetic coder -- runs on the GPU: ctx->fn_eco_coef (xeve_eco_coef, xeve_eco.c:1067-1089)
 * hands the live XEVE_SBAC over, field by field, and takes it back as xeve_hip_eco_coef_host leaves it.  Writing the real bitstream stays with the
 * reference's function. */
typedef struct { u32 range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter; u16 ctx[72]; } hip_sbac;
static int (*hip_eco_coef_host)(hip_sbac *, const s16 *, const s16 *, const s16 *, int, int, const int *, int, int, int);
static int (*orig_eco_coef)(XEVE_CTX *, XEVE_CORE *, XEVE_BSW *, s16 coef[N_C][MAX_CU_DIM], u8, int, int, int);
static unsigned long long eco_calls;
#define SBAC_MAP(F)                                                                                                             \
    F(skip_flag, 0, 2) F(pred_mode, 2, 3) F(direct_mode_flag, 5, 1) F(inter_dir, 6, 2) F(refi, 8, 2) F(mvp_idx, 10, 3) F(mvd, 13, 1)  \
    F(cbf_all, 14, 1) F(cbf_luma, 15, 1) F(cbf_cb, 16, 1) F(cbf_cr, 17, 1) F(run, 18, 24) F(last, 42, 2) F(level, 44, 24) \
    F(intra_dir, 68, 2) F(split_cu_flag, 70, 1) F(delta_qp, 71, 1)

static int shim_eco_coef(XEVE_CTX *ctx, XEVE_CORE *core, XEVE_BSW *bs, s16 coef[N_C][MAX_CU_DIM], u8 pred_mode, int enc_dqp, int b_no_cbf, int run_stats)
{
    XEVE_SBAC *sbac = (XEVE_SBAC *)bs->pdata[1];
    if(!sbac->is_bitcount || ctx->pps.cu_qp_delta_enabled_flag || core->log2_cuw > MAX_TR_LOG2 || core->log2_cuh > MAX_TR_LOG2 ||
       core->tree_cons.tree_type != TREE_LC || core->tree_cons.mode_cons != eAll)
        return orig_eco_coef(ctx, core, bs, coef, pred_mode, enc_dqp, b_no_cbf, run_stats);
    hip_sbac h;
    h.range = sbac->range, h.code = sbac->code, h.code_bits = sbac->code_bits, h.stacked_ff = sbac->stacked_ff, h.stacked_zero = sbac->stacked_zero;
    h.pending_byte = sbac->pending_byte, h.is_pending_byte = sbac->is_pending_byte, h.bitcounter = sbac->bitcounter, h.bin_counter = sbac->bin_counter;
#define F(name, at, n) memcpy(h.ctx + at, sbac->ctx.name, 2 * n);
    SBAC_MAP(F)
#undef F
    const int nnz[3] = {core->nnz_sub[Y_C][0], core->nnz_sub[U_C][0], core->nnz_sub[V_C][0]};
    const int flags = (pred_mode == MODE_INTRA ? 1 : 0) | (b_no_cbf == 1 ? 2 : 0) | ((run_stats & 7) << 2);
    if(hip_eco_coef_host(&h, coef[Y_C], coef[U_C], coef[V_C], core->log2_cuw, core->log2_cuh, nnz, flags, ctx->sps.chroma_format_idc, sbac->ctx.sps_cm_init_flag) != 0) {
        fprintf(stderr, "[xeve_hip_shim] eco_coef: %s\n", hip_err());
        abort();
    }
    sbac->range = h.range, sbac->code = h.code, sbac->code_bits = h.code_bits, sbac->stacked_ff = h.stacked_ff, sbac->stacked_zero = h.stacked_zero;
    sbac->pending_byte = h.pending_byte, sbac->is_pending_byte = h.is_pending_byte, sbac->bitcounter = h.bitcounter, sbac->bin_counter = h.bin_counter;
#define F(name, at, n) memcpy(sbac->ctx.name, h.ctx + at, 2 * n);
    SBAC_MAP(F)
#undef F
    eco_calls++;
    return XEVE_OK;
}

/* XEVE_HIP_SHIM_INTER=1: the WHOLE inter analysis of a CU -- ctx->fn_pinter_analyze_cu (xeve_pinter_analyze_cu, xeve_pinter.c:1839-2047): skip / merge
 * analysis, temporal direct, per-list motion search + check_best_mvp + pinter_residue_rdo, the iterated bi-prediction search, the mode decision and
 * the reconstruction -- runs on the GPU (xeve_hip_pinter_analyze_cu_host).  The adapter derives the merge / MVP candidates with the reference's own
 * xeve_get_motion from the live maps, hands the entry coder state over field by field, and leaves what the reference's function leaves for
 * mode_check_inter / copy_to_cu_data: core->cu_mode, core->nnz / nnz_sub, mi->*, the coefficient and reconstruction buffers, core->s_next_best,
 * core->cost_best.  Square CUs 8..64 (every inter CU of the Baseline quad-tree at these presets); anything else goes to the reference's function. */
typedef struct { int log2_cuw, log2_cuh, pic_w, pic_h, slice_type, num_refp[2], chroma_format_idc, bit_depth, tool_iqt, qp[3], pad_; double lambda[3], dist_chroma_weight[2]; } hip_rdo_params;
typedef struct { hip_rdo_params rdo; hip_epzs_params me; int refi_bits[2][8], range_recentre[2][8], max_cand, poc, col_list_poc0, pad_; double skip_th; } hip_inter_params;
typedef struct { int x, y; s16 mvp[2][4][2]; s16 mv_col[2]; int sbac; u8 ctx_skip, ctx_pred_mode, pad_[2]; } hip_inter_job;
typedef struct { double cost, cost_inter[5]; int cu_mode, best_idx; s16 mv[2][2], mvd[2][2]; s8 refi[2]; u8 mvp_idx[2]; int nnz[3], pad_[2]; } hip_inter_result;
static int (*hip_inter_host)(const pel *const *, int, int, const hip_refpic *, int, int, int, int, const hip_sbac *, const hip_inter_params *, const hip_inter_job *,
                             const void *, const void *, hip_inter_result *, s16 *, s16 *, s16 *, pel *, pel *, pel *, pel *, hip_sbac *);
static double (*orig_pinter_analyze_cu)(XEVE_CTX *, XEVE_CORE *, int, int, int, int, XEVE_MODE *, s16 coef[N_C][MAX_CU_DIM], pel *rec[N_C], int s_rec[N_C]);
static unsigned long long inter_calls, inter_fallbacks;
static double inter_seconds; /* wall time inside xeve_hip_pinter_analyze_cu_host, all threads */

static double shim_pinter_analyze_cu(XEVE_CTX *ctx, XEVE_CORE *core, int x, int y, int log2_cuw, int log2_cuh, XEVE_MODE *mi, s16 coef[N_C][MAX_CU_DIM], pel *rec[N_C],
                                     int s_rec[N_C])
{
    XEVE_PINTER *pi = &ctx->pinter[core->thread_cnt];
    const int isb = pi->slice_type == SLICE_B, idc = ctx->sps.chroma_format_idc, ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift;
    const int nr[2] = {ctx->rpm.num_refp[REFP_0], isb ? ctx->rpm.num_refp[REFP_1] : 0};
    if(log2_cuw != log2_cuh || log2_cuw < 3 || log2_cuw > 6 || ctx->pps.cu_qp_delta_enabled_flag || ctx->param.rdo_dbk_switch || nr[0] > 8 || nr[1] > nr[0] || nr[0] < 1 || (isb && nr[1] < 1) || core->tree_cons.tree_type != TREE_LC || core->tree_cons.mode_cons != eAll) {
        inter_fallbacks++;
        return orig_pinter_analyze_cu(ctx, core, x, y, log2_cuw, log2_cuh, mi, coef, rec, s_rec);
    }
    hip_inter_params P;
    memset(&P, 0, sizeof(P));
    P.rdo.log2_cuw = log2_cuw, P.rdo.log2_cuh = log2_cuh, P.rdo.pic_w = ctx->w, P.rdo.pic_h = ctx->h, P.rdo.slice_type = pi->slice_type;
    P.rdo.num_refp[0] = nr[0], P.rdo.num_refp[1] = nr[1], P.rdo.chroma_format_idc = idc, P.rdo.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8, P.rdo.tool_iqt = ctx->param.tool_iqt;
    P.rdo.qp[0] = core->qp_y, P.rdo.qp[1] = core->qp_u, P.rdo.qp[2] = core->qp_v;
    for(int c = 0; c < 3; c++) P.rdo.lambda[c] = core->lambda[c];
    P.rdo.dist_chroma_weight[0] = core->dist_chroma_weight[0], P.rdo.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    P.me.lambda_mv = pi->lambda_mv, P.me.faststep = 3, P.me.max_search_range = pi->max_search_range;
    P.me.min_clip[0] = pi->min_clip[MV_X], P.me.min_clip[1] = pi->min_clip[MV_Y], P.me.max_clip[0] = pi->max_clip[MV_X], P.me.max_clip[1] = pi->max_clip[MV_Y];
    P.me.hpel_cnt = pi->me_level > ME_LEV_IPEL ? pi->search_pattern_hpel_cnt : 0, P.me.qpel_cnt = pi->me_level > ME_LEV_HPEL ? pi->search_pattern_qpel_cnt : 0;
    P.me.reserved = pi->me_complexity > 1 ? 1 : 0;
    hip_refpic tab[16];
    memset(tab, 0, sizeof(tab));
    XEVE_PIC *any = pi->refp[0][REFP_0].pic;
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            XEVE_PIC *rp = pi->refp[r][l].pic;
            tab[r * 2 + l].y = rp->y, tab[r * 2 + l].u = rp->u, tab[r * 2 + l].v = rp->v, tab[r * 2 + l].poc = pi->refp[r][l].poc;
            P.refi_bits[l][r] = xeve_tbl_refi_bits[nr[l]][r];
            P.range_recentre[l][r] = XEVE_CLIP3(pi->max_search_range >> 2, pi->max_search_range, /* get_range_ipel (xeve_pinter.c:122-129) */
                                                (pi->max_search_range * XEVE_ABS(pi->poc - (int)pi->refp[r][l].poc) + (pi->gop_size >> 1)) / pi->gop_size);
        }
    P.max_cand = pi->skip_merge_cand_num, P.poc = ctx->poc.poc_val, P.col_list_poc0 = isb ? (int)pi->refp[0][REFP_1].list_poc[0] : 0, P.skip_th = ctx->param.skip_th;
    hip_inter_job J;
    memset(&J, 0, sizeof(J));
    J.x = x, J.y = y, J.ctx_skip = core->ctx_flags[CNID_SKIP_FLAG], J.ctx_pred_mode = core->ctx_flags[CNID_PRED_MODE];
    for(int l = 0; l <= isb; l++) { /* the candidates of xeve_analyze_skip and of the per-list search: the reference's own derivation (xeve_util.c:526-573) */
        s8 refi_tmp[MAX_NUM_MVP];
        xeve_get_motion(core->scup, l, ctx->map_refi, ctx->map_mv, pi->refp, core->cuw, core->cuh, ctx->w_scu, core->avail_cu, refi_tmp, J.mvp[l]);
    }
    if(isb) {
        const int corner = core->scup + ((1 << (log2_cuw - MIN_CU_LOG2)) - 1) + ((1 << (log2_cuh - MIN_CU_LOG2)) - 1) * ctx->w_scu; /* xeve_get_mv_dir's scup (:1543) */
        J.mv_col[0] = pi->refp[0][REFP_1].map_mv[corner][0][MV_X], J.mv_col[1] = pi->refp[0][REFP_1].map_mv[corner][0][MV_Y];
    }
    const XEVE_SBAC *sb = &core->s_curr_best[log2_cuw - 2][log2_cuh - 2];
    hip_sbac h, nb;
    h.range = sb->range, h.code = sb->code, h.code_bits = sb->code_bits, h.stacked_ff = sb->stacked_ff, h.stacked_zero = sb->stacked_zero;
    h.pending_byte = sb->pending_byte, h.is_pending_byte = sb->is_pending_byte, h.bitcounter = sb->bitcounter, h.bin_counter = sb->bin_counter;
#define F(name, at, n) memcpy(h.ctx + at, sb->ctx.name, 2 * n);
    SBAC_MAP(F)
#undef F
    hip_inter_result R;
    static __thread s16 cf[N_C][MAX_CU_DIM];
    static __thread pel rc[N_C][MAX_CU_DIM], py[MAX_CU_DIM];
    const pel *org[3] = {pi->o[Y_C], pi->o[U_C], pi->o[V_C]};
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if(hip_inter_host(org, pi->s_o[Y_C], pi->s_o[U_C], tab, any->s_l, any->s_c, any->pad_l, any->pad_c, &h, &P, &J, pi->mc_l_coeff, pi->mc_c_coeff, &R, cf[Y_C], cf[U_C],
                      cf[V_C], rc[Y_C], rc[U_C], rc[V_C], py, &nb) != 0) {
        fprintf(stderr, "[xeve_hip_shim] inter analysis: %s\n", hip_err());
        abort();
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    inter_seconds += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec); /* (a statistic: races between encoder threads only blur it) */
    /* what xeve_pinter_analyze_cu leaves behind (:2004-2046) */
    const int best = R.best_idx, n0 = 1 << (log2_cuw + log2_cuh), n1 = n0 >> (ws + hs);
    core->cu_mode = R.cu_mode;
    for(int c = 0; c < N_C; c++) {
        if(c && !idc) continue;
        const int n = c ? n1 : n0;
        memcpy(coef[c], cf[c], sizeof(s16) * n);
        memcpy(pi->rec[best][c], rc[c], sizeof(pel) * n);
        rec[c] = pi->rec[best][c], s_rec[c] = c ? (1 << log2_cuw) >> ws : 1 << log2_cuw;
        core->nnz[c] = R.nnz[c];
        memset(core->nnz_sub[c], 0, sizeof(int) * MAX_SUB_TB_NUM);
        core->nnz_sub[c][0] = R.nnz[c];
    }
    memcpy(pi->pred[best][0][Y_C], py, sizeof(pel) * n0);
    mi->pred_y_best = pi->pred[best][0][Y_C];
    for(int l = 0; l < REFP_NUM; l++) {
        mi->refi[l] = R.refi[l], mi->mvp_idx[l] = R.mvp_idx[l];
        mi->mv[l][MV_X] = R.mv[l][0], mi->mv[l][MV_Y] = R.mv[l][1], mi->mvd[l][MV_X] = R.mvd[l][0], mi->mvd[l][MV_Y] = R.mvd[l][1];
    }
    XEVE_SBAC *out = &core->s_next_best[log2_cuw - 2][log2_cuh - 2];
    *out = *sb; /* fields the analysis does not touch (is_bitcount, the context models of other syntax) */
    out->range = nb.range, out->code = nb.code, out->code_bits = nb.code_bits, out->stacked_ff = nb.stacked_ff, out->stacked_zero = nb.stacked_zero;
    out->pending_byte = nb.pending_byte, out->is_pending_byte = nb.is_pending_byte, out->bitcounter = nb.bitcounter, out->bin_counter = nb.bin_counter;
#define F(name, at, n) memcpy(out->ctx.name, nb.ctx + at, 2 * n);
    SBAC_MAP(F)
#undef F
    core->dqp_next_best[log2_cuw - 2][log2_cuh - 2] = core->dqp_curr_best[log2_cuw - 2][log2_cuh - 2];
    if(R.cost < core->cost_best) core->cost_best = R.cost;
    inter_calls++;
    return R.cost;
}

/* XEVE_HIP_SHIM_RESIDENT=1 (with XEVE_HIP_SHIM_INTER): resident pictures.  ctx->fn_mode_analyze_frame is the reference's once-per-picture hook (called
 * by xeve_pic before the CTU loop, xeve_enc.c:275; mode_analyze_frame itself does nothing, xeve_mode.c:2441): it announces the new picture to the
 * library, which from then on uploads each plane the inter analysis is handed once per picture instead of once per CU. */
static int (*hip_picture_begin)(void);
static int (*hip_resident_stats)(unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *);
static int (*orig_analyze_frame)(XEVE_CTX *);
static int shim_analyze_frame(XEVE_CTX *ctx)
{
    if(hip_picture_begin() != 0) {
        fprintf(stderr, "[xeve_hip_shim] picture_begin: %s\n", hip_err());
        abort();
    }
    return orig_analyze_frame ? orig_analyze_frame(ctx) : XEVE_OK;
}

/* XEVE_HIP_SHIM_INTRA=1: ctx->fn_pintra_analyze_cu (the static pintra_analyze_cu, xeve_pintra.c:544-698) -> xeve_hip_pintra_analyze_cu_host: the neighbour
 * samples from the mode picture, the five predictors, the SATD + mode-bits candidate list, the luma RDO of the list, the chroma RDO of its winner, the CU's cost and
 * exit coder state.  The adapter hands over the planes, the 4x4-unit maps and the entry coder state as they stand and leaves behind what the reference's function
 * leaves for mode_check_intra / copy_to_cu_data: coef, rec / s_rec (pi->rec), core->nnz / nnz_sub, core->ipm, core->dist_cu, core->s_temp_best, core->dqp_temp_best.
 * Square CUs 4..64 of the Baseline quad-tree; anything else goes to the reference's function. */
typedef struct { int log2_cuw, log2_cuh, w_scu, h_scu, slice_type, chroma_format_idc, bit_depth, tool_iqt, constrained_intra_pred, qp[3]; double lambda[3], sqrt_lambda0, dist_chroma_weight[2]; } hip_intra_params;
typedef struct { int x, y; u32 inter_satd; int sbac, pic; u8 ctx_skip, ctx_pred_mode, pad_[2]; } hip_intra_job;
typedef struct { double cost; int dist_cu, nnz[3], pred_cnt; s8 ipm[2], pad_[2]; } hip_intra_result;
static int (*hip_intra_host)(const pel *const *, int, int, const pel *const *, int, int, const u32 *, const s8 *, const u8 *, const hip_sbac *, const hip_intra_params *,
                             const hip_intra_job *, hip_intra_result *, s16 *, s16 *, s16 *, pel *, pel *, pel *, hip_sbac *);
static double (*orig_pintra_analyze_cu)(XEVE_CTX *, XEVE_CORE *, int, int, int, int, XEVE_MODE *, s16 coef[N_C][MAX_CU_DIM], pel *rec[N_C], int s_rec[N_C]);
static unsigned long long intra_calls, intra_fallbacks;

static double shim_pintra_analyze_cu(XEVE_CTX *ctx, XEVE_CORE *core, int x, int y, int log2_cuw, int log2_cuh, XEVE_MODE *mi, s16 coef[N_C][MAX_CU_DIM], pel *rec[N_C],
                                     int s_rec[N_C])
{
    XEVE_PINTRA *pi = &ctx->pintra[core->thread_cnt];
    const int idc = ctx->sps.chroma_format_idc, ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift;
    if(log2_cuw != log2_cuh || log2_cuw < 2 || log2_cuw > 6 || ctx->pps.cu_qp_delta_enabled_flag || ctx->param.rdo_dbk_switch || ctx->param.tool_iqt || idc == 2 ||
       core->tree_cons.tree_type != TREE_LC || core->tree_cons.mode_cons != eAll || ctx->fn_rdo_intra_ext || ctx->fn_rdo_intra_ext_c || ctx->sps.tool_admvp) {
        intra_fallbacks++;
        return orig_pintra_analyze_cu(ctx, core, x, y, log2_cuw, log2_cuh, mi, coef, rec, s_rec);
    }
    hip_intra_params P;
    memset(&P, 0, sizeof(P));
    P.log2_cuw = log2_cuw, P.log2_cuh = log2_cuh, P.w_scu = ctx->w_scu, P.h_scu = ctx->h_scu, P.slice_type = ctx->sh->slice_type, P.chroma_format_idc = idc;
    P.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8, P.tool_iqt = ctx->param.tool_iqt, P.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;
    P.qp[0] = core->qp_y, P.qp[1] = core->qp_u, P.qp[2] = core->qp_v;
    for(int c = 0; c < 3; c++) P.lambda[c] = core->lambda[c];
    P.sqrt_lambda0 = core->sqrt_lambda[0], P.dist_chroma_weight[0] = core->dist_chroma_weight[0], P.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    hip_intra_job J;
    memset(&J, 0, sizeof(J));
    J.x = x, J.y = y, J.inter_satd = core->inter_satd, J.ctx_skip = core->ctx_flags[CNID_SKIP_FLAG], J.ctx_pred_mode = core->ctx_flags[CNID_PRED_MODE];
    const XEVE_SBAC *sb = &core->s_curr_best[log2_cuw - 2][log2_cuh - 2];
    hip_sbac h, nb;
    h.range = sb->range, h.code = sb->code, h.code_bits = sb->code_bits, h.stacked_ff = sb->stacked_ff, h.stacked_zero = sb->stacked_zero;
    h.pending_byte = sb->pending_byte, h.is_pending_byte = sb->is_pending_byte, h.bitcounter = sb->bitcounter, h.bin_counter = sb->bin_counter;
#define F(name, at, n) memcpy(h.ctx + at, sb->ctx.name, 2 * n);
    SBAC_MAP(F)
#undef F
    hip_intra_result R;
    static __thread s16 cf[N_C][MAX_CU_DIM];
    const pel *org[3] = {pi->o[Y_C], pi->o[U_C], pi->o[V_C]}, *mod[3] = {pi->m[Y_C], pi->m[U_C], pi->m[V_C]};
    if(hip_intra_host(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pi->s_m[Y_C], pi->s_m[U_C], ctx->map_scu, ctx->map_ipm, ctx->map_tidx, &h, &P, &J, &R, cf[Y_C], cf[U_C], cf[V_C],
                      pi->rec[Y_C], pi->rec[U_C], pi->rec[V_C], &nb) != 0) {
        fprintf(stderr, "[xeve_hip_shim] intra analysis: %s\n", hip_err());
        abort();
    }
    __atomic_fetch_add(&intra_calls, 1, __ATOMIC_RELAXED);
    /* what pintra_analyze_cu leaves behind (:660-697) */
    const int n0 = 1 << (log2_cuw + log2_cuh), n1 = n0 >> (ws + hs);
    for(int c = 0; c < N_C; c++) {
        if(c && !idc) continue;
        memcpy(coef[c], cf[c], sizeof(s16) * (c ? n1 : n0));
        rec[c] = pi->rec[c], s_rec[c] = c ? (1 << log2_cuw) >> ws : 1 << log2_cuw;
        core->nnz[c] = R.nnz[c];
        memset(core->nnz_sub[c], 0, sizeof(int) * MAX_SUB_TB_NUM);
        core->nnz_sub[c][0] = R.nnz[c];
    }
    core->ipm[0] = R.ipm[0];
    if(idc) core->ipm[1] = R.ipm[1];
    core->dist_cu = R.dist_cu;
    xeve_get_mpm(core->x_scu, core->y_scu, 1 << log2_cuw, 1 << log2_cuh, ctx->map_scu, ctx->map_ipm, core->scup, ctx->w_scu, &core->mpm_b_list, ctx->map_tidx); /* (pintra_get_mpm, :376-388) */
    XEVE_SBAC *out = &core->s_temp_best;
    *out = *sb; /* fields the analysis does not touch (is_bitcount, the context models of other syntax) */
    out->range = nb.range, out->code = nb.code, out->code_bits = nb.code_bits, out->stacked_ff = nb.stacked_ff, out->stacked_zero = nb.stacked_zero;
    out->pending_byte = nb.pending_byte, out->is_pending_byte = nb.is_pending_byte, out->bitcounter = nb.bitcounter, out->bin_counter = nb.bin_counter;
#define F(name, at, n) memcpy(out->ctx.name, nb.ctx + at, 2 * n);
    SBAC_MAP(F)
#undef F
    core->dqp_temp_run = core->dqp_curr_best[log2_cuw - 2][log2_cuh - 2], core->dqp_temp_best = core->dqp_temp_run; /* (:681, :696) */
    return R.cost;
}

/* XEVE_SHIM_SHADOW_TREE=<path of libxeve_oracle.so>: SHADOW MODE for the CTU mode decision of I pictures (ctx->fn_mode_analyze_lcu = mode_analyze_lcu ->
 * mode_coding_tree, xeve_mode.c:2007-2610).  Before the reference analyses a CTU the adapter snapshots what the walk reads (the picture being reconstructed, the
 * unit maps, the entry coder state); after the reference has run it lets the oracle's restatement (xo_mode_analyze_ctu_intra) walk the same CTU on the snapshot
 * and compares everything the walk produces: split modes, prediction modes, coded-block counts, unit maps, coefficients, reconstruction, exit coder state.  The
 * encoder continues with the reference's own results; mismatches are counted and reported at exit.  CPU only: this pins the ORACLE against the live encoder. */
#include "xeve_oracle.h"
static double (*xo_tree)(const xo_pel *const *, int, int, xo_pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                         const xo_tree_params *, int, int, xo_ctu_data *, xo_sbac *);
static double (*xo_tree_any)(const xo_pel *const *, int, int, xo_pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                             const xo_tree_params *, const xo_tree_inter *, int, int, xo_ctu_data *, xo_sbac *);
static int (*orig_mode_analyze_lcu)(XEVE_CTX *, XEVE_CORE *);
static unsigned long long shadow_ctus, shadow_bad, shadow_skipped, shadow_inter_ctus;

/* what the inter side of the walk is handed (the same derivation as shim_pinter_analyze_cu above, once per CTU); tab: 16 entries */
static int tree_inter_setup(XEVE_CTX *ctx, XEVE_CORE *core, xo_tree_inter *I, xo_refpic *tab, int16_t (*map_mv)[2][2], int8_t (*map_refi)[2])
{
    XEVE_PINTER *pi = &ctx->pinter[core->thread_cnt];
    const int isb = ctx->sh->slice_type == SLICE_B, idc = ctx->sps.chroma_format_idc;
    const int nr[2] = {ctx->rpm.num_refp[REFP_0], isb ? ctx->rpm.num_refp[REFP_1] : 0};
    if(nr[0] > 8 || nr[1] > nr[0] || nr[0] < 1 || (isb && nr[1] < 1) || ctx->param.min_cu_inter < 8) return -1;
    memset(I, 0, sizeof(*I)), memset(tab, 0, 16 * sizeof(*tab));
    xo_inter_params *P = &I->ipar;
    P->rdo.pic_w = ctx->w, P->rdo.pic_h = ctx->h, P->rdo.slice_type = ctx->sh->slice_type;
    P->rdo.num_refp[0] = nr[0], P->rdo.num_refp[1] = nr[1], P->rdo.chroma_format_idc = idc, P->rdo.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8, P->rdo.tool_iqt = 0;
    P->rdo.qp[0] = core->qp_y, P->rdo.qp[1] = core->qp_u, P->rdo.qp[2] = core->qp_v;
    for(int c = 0; c < 3; c++) P->rdo.lambda[c] = core->lambda[c];
    P->rdo.dist_chroma_weight[0] = core->dist_chroma_weight[0], P->rdo.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    P->me.me.lambda_mv = pi->lambda_mv, P->me.me.faststep = 3, P->me.me.max_search_range = pi->max_search_range;
    P->me.me.min_clip[0] = pi->min_clip[MV_X], P->me.me.min_clip[1] = pi->min_clip[MV_Y], P->me.me.max_clip[0] = pi->max_clip[MV_X], P->me.me.max_clip[1] = pi->max_clip[MV_Y];
    P->me.spel.lambda_mv = pi->lambda_mv;
    P->me.spel.hpel_cnt = pi->me_level > ME_LEV_IPEL ? pi->search_pattern_hpel_cnt : 0, P->me.spel.qpel_cnt = pi->me_level > ME_LEV_HPEL ? pi->search_pattern_qpel_cnt : 0;
    P->me.me.reserved = pi->me_complexity > 1 ? 1 : 0;
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            XEVE_PIC *rp = pi->refp[r][l].pic;
            tab[r * 2 + l].y = rp->y, tab[r * 2 + l].u = rp->u, tab[r * 2 + l].v = rp->v, tab[r * 2 + l].poc = pi->refp[r][l].poc;
            P->refi_bits[l][r] = xeve_tbl_refi_bits[nr[l]][r];
            P->range_recentre[l][r] = XEVE_CLIP3(pi->max_search_range >> 2, pi->max_search_range, /* get_range_ipel (xeve_pinter.c:122-129) */
                                                 (pi->max_search_range * XEVE_ABS((int)ctx->poc.poc_val - (int)pi->refp[r][l].poc) + (ctx->param.gop_size >> 1)) / ctx->param.gop_size);
        }
    P->max_cand = pi->skip_merge_cand_num, P->poc = ctx->poc.poc_val, P->col_list_poc0 = isb ? (int)pi->refp[0][REFP_1].list_poc[0] : 0, P->skip_th = ctx->param.skip_th;
    XEVE_PIC *any = pi->refp[0][REFP_0].pic;
    I->refp = tab, I->s_ref_l = any->s_l, I->s_ref_c = any->s_c, I->map_mv = map_mv, I->map_refi = map_refi;
    I->col0 = (const int16_t(*)[2][2])pi->refp[0][REFP_0].map_mv, I->col1 = isb ? (const int16_t(*)[2][2])pi->refp[0][REFP_1].map_mv : I->col0;
    I->ecu_depth = (ctx->poc.poc_val % 2) ? ENC_ECU_DEPTH_B - 2 : ENC_ECU_DEPTH_B; /* ENC_ECU_ADAPTIVE (xeve_mode.c:2162-2166) */
    return 0;
}

static void sbac_to_flat(xo_sbac *h, const XEVE_SBAC *sb)
{
    memset(h, 0, sizeof(*h));
    h->range = sb->range, h->code = sb->code, h->code_bits = sb->code_bits, h->stacked_ff = sb->stacked_ff, h->stacked_zero = sb->stacked_zero;
    h->pending_byte = sb->pending_byte, h->is_pending_byte = sb->is_pending_byte, h->bitcounter = sb->bitcounter, h->bin_counter = sb->bin_counter;
#define F(name, at, n) memcpy(h->ctx + at, sb->ctx.name, 2 * n);
    SBAC_MAP(F)
#undef F
}

/* the writer's side in shadow mode: when CTU n + 1 enters, the reference has written CTU n with xeve_eco_tree; the oracle's xo_eco_ctu writes the same CTU from the
 * state CTU n entered with, on the maps as the decision left them, and must arrive at the state CTU n + 1 enters with (it is loaded from the writer, xeve_enc.c:139),
 * at the same bytes in the bitstream buffer and at the same unit flags */
static int (*xo_eco)(xo_sbac *, const xo_ctu_data *, const xo_tree_params *, const int *, uint32_t *, const int8_t *, const uint8_t *, uint32_t *, int, int, uint8_t *, int);
static unsigned long long eco_ctus, eco_bad, eco_bytes;
static struct {
    int            valid, lcu, x0, y0, num_refp[2];
    long           byte_pos;
    const void    *pic;
    xo_sbac        entry;
    xo_ctu_data    out;
    xo_tree_params P;
    uint32_t      *scu, *cum;
    int8_t        *ipm;
} W;
static long bsw_pos(const XEVE_BSW *bs) { return (long)(bs->cur - bs->beg) + ((32 - bs->leftbits) >> 3); }
static int  bsw_byte(const XEVE_BSW *bs, long pos)
{
    const long flushed = (long)(bs->cur - bs->beg);
    return pos < flushed ? bs->beg[pos] : (int)((bs->code >> (24 - 8 * (pos - flushed))) & 0xFF);
}
static void gblob(FILE *f, const char *name, const void *data, size_t n);
static int  golden_wanted(int lcu);
static void shadow_writer_check(XEVE_CTX *ctx, XEVE_CORE *core, const xo_sbac *now)
{
    const XEVE_BSW *bs = &ctx->bs[core->thread_cnt];
    if(W.valid && W.pic == (const void *)PIC_MODE(ctx) && core->lcu_num == W.lcu + 1 && xo_eco) {
        static uint8_t bytes[1 << 16];
        xo_sbac s = W.entry;
        const int n = xo_eco(&s, &W.out, &W.P, W.num_refp, W.scu, W.ipm, ctx->map_tidx, W.cum, W.x0, W.y0, bytes, (int)sizeof(bytes));
        int bad = 0;
        if(s.range != now->range || s.code != now->code || s.code_bits != now->code_bits || s.stacked_ff != now->stacked_ff || s.stacked_zero != now->stacked_zero ||
           s.pending_byte != now->pending_byte || s.is_pending_byte != now->is_pending_byte || memcmp(s.ctx, now->ctx, sizeof(s.ctx))) {
            if(eco_bad < 6) fprintf(stderr, "[shadow writer] CTU %d: coder state differs (range %u vs %u, code %u vs %u, bits %u vs %u)\n", W.lcu, s.range, now->range, s.code, now->code, s.code_bits, now->code_bits);
            bad = 1;
        }
        const long p1 = bsw_pos(bs);
        if(p1 - W.byte_pos != n) { if(eco_bad < 6) fprintf(stderr, "[shadow writer] CTU %d: %d bytes vs %ld in the bitstream\n", W.lcu, n, p1 - W.byte_pos); bad = 1; }
        else
            for(int i = 0; i < n && i < (int)sizeof(bytes); i++)
                if(bytes[i] != bsw_byte(bs, W.byte_pos + i)) { if(eco_bad < 6) fprintf(stderr, "[shadow writer] CTU %d: byte %d differs\n", W.lcu, i); bad = 1; break; }
        const int nu = 1 << (ctx->log2_max_cuwh - 2), wu = XEVE_MIN(nu, ctx->w_scu - (W.x0 >> 2)), hu = XEVE_MIN(nu, ctx->h_scu - (W.y0 >> 2));
        for(int j = 0; j < hu && !bad; j++)
            for(int i = 0; i < wu; i++) {
                const int g = ((W.y0 >> 2) + j) * ctx->w_scu + (W.x0 >> 2) + i;
                if(W.scu[g] != ctx->map_scu[g] || W.cum[g] != ctx->map_cu_mode[g]) { if(eco_bad < 6) fprintf(stderr, "[shadow writer] CTU %d: unit %d flags %08x / %08x vs %08x / %08x\n", W.lcu, g, W.scu[g], W.cum[g], ctx->map_scu[g], ctx->map_cu_mode[g]); bad = 1; break; }
            }
        eco_ctus++, eco_bad += bad, eco_bytes += (unsigned long long)n;
        if(getenv("XEVE_SHIM_TREE_GOLDEN") && golden_wanted(W.lcu)) { /* the reference writer's side of a recorded CTU: the state it left (= this CTU's entry state), its bytes */
            FILE *f = fopen(getenv("XEVE_SHIM_TREE_GOLDEN"), "ab");
            if(f) {
                const int32_t hd[8] = {(int32_t)ctx->poc.poc_val, ctx->sh->slice_type, W.x0, W.y0, W.lcu, W.num_refp[0], W.num_refp[1], (int32_t)(p1 - W.byte_pos)};
                uint8_t *rb = malloc((size_t)(p1 - W.byte_pos) + 1);
                for(long i = 0; i < p1 - W.byte_pos; i++) rb[i] = (uint8_t)bsw_byte(bs, W.byte_pos + i);
                gblob(f, "wr_head", hd, sizeof(hd)), gblob(f, "wr_state", now, sizeof(*now)), gblob(f, "wr_bytes", rb, (size_t)(p1 - W.byte_pos));
                gblob(f, "wr_scu", ctx->map_scu, 4 * (size_t)(ctx->w_scu * ctx->h_scu)), gblob(f, "wr_cu_mode", ctx->map_cu_mode, 4 * (size_t)(ctx->w_scu * ctx->h_scu));
                gblob(f, "end", NULL, 0);
                free(rb), fclose(f);
            }
        }
    }
    W.valid = 0;
}

/* XEVE_SHIM_TREE_GOLDEN=<file> (shadow mode): for the CTUs listed in XEVE_SHIM_TREE_GOLDEN_CTUS (default "0") of every picture, append a record of what the CTU mode
 * decision was handed and of what THE REFERENCE made of it -- the raw material of tests/golden/tree_v1.npz (tests/golden/make_tree_golden.py turns the records into
 * arrays).  A record is a sequence of named blobs: name[16], int64 size, data; "end" closes it. */
static void gblob(FILE *f, const char *name, const void *data, size_t n)
{
    char    nm[16] = {0};
    int64_t sz = (int64_t)n;
    strncpy(nm, name, 15);
    fwrite(nm, 1, 16, f), fwrite(&sz, 8, 1, f);
    if(n) fwrite(data, 1, n, f);
}
static void gplane(FILE *f, const char *name, const pel *p, int stride, int w, int h)
{   /* the picture area, rows packed */
    pel *t = malloc(sizeof(pel) * (size_t)w * h);
    for(int y = 0; y < h; y++) memcpy(t + (size_t)y * w, p + (size_t)y * stride, sizeof(pel) * w);
    gblob(f, name, t, sizeof(pel) * (size_t)w * h);
    free(t);
}
static int golden_wanted(int lcu)
{
    const char *l = getenv("XEVE_SHIM_TREE_GOLDEN_CTUS");
    if(!l) return lcu == 0;
    for(const char *q = l; *q;) {
        if(atoi(q) == lcu) return 1;
        while(*q && *q != ',') q++;
        if(*q) q++;
    }
    return 0;
}
static void golden_dump(XEVE_CTX *ctx, XEVE_CORE *core, const xo_tree_params *P, const xo_tree_inter *TI, xo_pel *const mod_before[3], const uint32_t *scu, const int8_t *ipm,
                        const uint32_t *cum, const int16_t (*mv)[2][2], const int8_t (*refi)[2], const xo_sbac *entry, int x0, int y0, int lcu)
{
    FILE *f = fopen(getenv("XEVE_SHIM_TREE_GOLDEN"), "ab");
    if(!f) return;
    XEVE_PIC    *pm = PIC_MODE(ctx);
    XEVE_PINTRA *pi = &ctx->pintra[core->thread_cnt];
    const int idc = ctx->sps.chroma_format_idc, ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift, w = ctx->w, h = ctx->h, wc = idc ? w >> ws : 0, hc = idc ? h >> hs : 0;
    const int nscu = ctx->w_scu * ctx->h_scu, L = ctx->log2_max_cuwh - 2;
    const int32_t hd[8] = {(int32_t)ctx->poc.poc_val, ctx->sh->slice_type, x0, y0, lcu, TI ? TI->s_ref_l : 0, TI ? TI->s_ref_c : 0, TI ? TI->ecu_depth : 0};
    gblob(f, "head", hd, sizeof(hd)), gblob(f, "params", P, sizeof(*P)), gblob(f, "entry", entry, sizeof(*entry));
    gplane(f, "org_y", pi->o[Y_C], pi->s_o[Y_C], w, h), gplane(f, "mod_y", mod_before[0], pm->s_l, w, h);
    if(idc) {
        gplane(f, "org_u", pi->o[U_C], pi->s_o[U_C], wc, hc), gplane(f, "org_v", pi->o[V_C], pi->s_o[U_C], wc, hc);
        gplane(f, "mod_u", mod_before[1], pm->s_c, wc, hc), gplane(f, "mod_v", mod_before[2], pm->s_c, wc, hc);
    }
    gblob(f, "map_scu", scu, 4 * (size_t)nscu), gblob(f, "map_ipm", ipm, (size_t)nscu), gblob(f, "map_tidx", ctx->map_tidx, (size_t)nscu), gblob(f, "map_cu_mode", cum, 4 * (size_t)nscu);
    if(TI) {
        XEVE_PINTER *pin = &ctx->pinter[core->thread_cnt];
        XEVE_PIC    *any = pin->refp[0][REFP_0].pic;
        const int32_t rh[6] = {TI->ipar.rdo.num_refp[0], TI->ipar.rdo.num_refp[1], any->pad_l, any->pad_c, any->s_l, any->s_c};
        gblob(f, "ref_head", rh, sizeof(rh)), gblob(f, "inter_params", &TI->ipar, sizeof(TI->ipar));
        gblob(f, "map_mv", mv, sizeof(*mv) * (size_t)nscu), gblob(f, "map_refi", refi, sizeof(*refi) * (size_t)nscu);
        gblob(f, "col0", TI->col0, 8 * (size_t)nscu), gblob(f, "col1", TI->col1, 8 * (size_t)nscu);
        for(int l = 0; l < 2; l++)
            for(int r = 0; r < rh[l]; r++) { /* whole padded planes: the searches and the interpolation read around the picture */
                XEVE_PIC *rp = pin->refp[r][l].pic;
                char nm[16];
                const int32_t poc = (int32_t)pin->refp[r][l].poc;
                snprintf(nm, sizeof(nm), "ref%d_%d_poc", r, l), gblob(f, nm, &poc, 4);
                snprintf(nm, sizeof(nm), "ref%d_%d_y", r, l), gblob(f, nm, rp->y - rp->pad_l * rp->s_l - rp->pad_l, sizeof(pel) * (size_t)rp->s_l * (h + 2 * rp->pad_l));
                if(idc) {
                    snprintf(nm, sizeof(nm), "ref%d_%d_u", r, l), gblob(f, nm, rp->u - rp->pad_c * rp->s_c - rp->pad_c, sizeof(pel) * (size_t)rp->s_c * (hc + 2 * rp->pad_c));
                    snprintf(nm, sizeof(nm), "ref%d_%d_v", r, l), gblob(f, nm, rp->v - rp->pad_c * rp->s_c - rp->pad_c, sizeof(pel) * (size_t)rp->s_c * (hc + 2 * rp->pad_c));
                }
            }
    }
    /* what the reference made of it: the CTU's XEVE_CU_DATA in the oracle's record layout (units inside the picture; split modes of all units), the maps and the
     * picture after, core->s_next_best */
    static xo_ctu_data g;
    memset(&g, 0, sizeof(g));
    const XEVE_CU_DATA *cd = &ctx->map_cu_data[lcu];
    const int nu = 1 << L, ctu = 1 << ctx->log2_max_cuwh, wu = XEVE_MIN(nu, ctx->w_scu - (x0 >> 2)), hu = XEVE_MIN(nu, ctx->h_scu - (y0 >> 2));
    for(int u = 0; u < nu * nu; u++)
        for(int d = 0; d < XO_CU_DEPTHS; d++) g.split_mode[d][u] = cd->split_mode[d][SQUARE][u];
    for(int j = 0; j < hu; j++)
        for(int i = 0; i < wu; i++) {
            const int u = j * nu + i;
            g.pred_mode[u] = cd->pred_mode[u], g.ipm[0][u] = cd->ipm[0][u], g.ipm[1][u] = cd->ipm[1][u], g.depth[u] = cd->depth[u];
            for(int c = 0; c < 3; c++) g.nnz[c][u] = cd->nnz[c][u];
            g.map_scu[u] = cd->map_scu[u], g.map_cu_mode[u] = cd->map_cu_mode[u];
            memcpy(g.mv[u], cd->mv[u], sizeof(g.mv[u])), memcpy(g.mvd[u], cd->mvd[u], sizeof(g.mvd[u]));
            g.refi[u][0] = cd->refi[u][0], g.refi[u][1] = cd->refi[u][1], g.mvp_idx[u][0] = cd->mvp_idx[u][0], g.mvp_idx[u][1] = cd->mvp_idx[u][1];
        }
    for(int c = 0; c < (idc ? 3 : 1); c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, cs = ctu >> sx, ww = (wu * 4) >> sx, hh = (hu * 4) >> sy;
        for(int yy = 0; yy < hh; yy++) memcpy(g.coef[c] + yy * cs, cd->coef[c] + yy * cs, sizeof(s16) * ww), memcpy(g.reco[c] + yy * cs, cd->reco[c] + yy * cs, sizeof(pel) * ww);
    }
    xo_sbac nb;
    sbac_to_flat(&nb, &core->s_next_best[L][L]);
    gblob(f, "ref_ctu", &g, sizeof(g)), gblob(f, "ref_next", &nb, sizeof(nb));
    gblob(f, "ref_scu", ctx->map_scu, 4 * (size_t)nscu), gblob(f, "ref_ipm", ctx->map_ipm, (size_t)nscu), gblob(f, "ref_cu_mode", ctx->map_cu_mode, 4 * (size_t)nscu);
    if(TI) gblob(f, "ref_mv", ctx->map_mv, sizeof(*ctx->map_mv) * (size_t)nscu), gblob(f, "ref_refi", ctx->map_refi, sizeof(*ctx->map_refi) * (size_t)nscu);
    gplane(f, "ref_mod_y", pm->y, pm->s_l, w, h);
    if(idc) gplane(f, "ref_mod_u", pm->u, pm->s_c, wc, hc), gplane(f, "ref_mod_v", pm->v, pm->s_c, wc, hc);
    gblob(f, "end", NULL, 0);
    fclose(f);
}

/* the whole picture in shadow mode: at a picture's first CTU the oracle decides AND writes every CTU of the picture on its own -- private copies of the picture being
 * reconstructed and of the maps, every CTU entering with the state the oracle's own writer left (the closed chain), the tile end and xeve_sbac_finish at the end -- and
 * when the reference has written the picture (ctx->fn_loop_filter is called right after its CTU loop) the bytes in its bitstream buffer must be the oracle's */
static int (*xo_tile_end)(xo_sbac *, uint8_t *, int);
static int (*orig_shadow_loop_filter)(XEVE_CTX *, XEVE_CORE *);
static unsigned long long ap_pics, ap_bad, ap_bytes;
static struct {
    int      valid, n, n_ctus, cap; /* n: all of the slice data; n_ctus: the part before the tile's end */
    long     pos0;
    uint8_t *bytes;
} AP;
static void build_tree_params(XEVE_CTX *ctx, XEVE_CORE *core, xo_tree_params *P)
{
    const int is_i = ctx->sh->slice_type == SLICE_I;
    memset(P, 0, sizeof(*P));
    P->ip.w_scu = ctx->w_scu, P->ip.h_scu = ctx->h_scu, P->ip.slice_type = ctx->sh->slice_type, P->ip.chroma_format_idc = ctx->sps.chroma_format_idc;
    P->ip.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8, P->ip.tool_iqt = 0, P->ip.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;
    P->ip.qp[0] = core->qp_y, P->ip.qp[1] = core->qp_u, P->ip.qp[2] = core->qp_v;
    for(int c = 0; c < 3; c++) P->ip.lambda[c] = core->lambda[c];
    P->ip.sqrt_lambda0 = core->sqrt_lambda[0], P->ip.dist_chroma_weight[0] = core->dist_chroma_weight[0], P->ip.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    P->pic_w = ctx->w, P->pic_h = ctx->h, P->log2_ctu = ctx->log2_max_cuwh, P->min_cuwh = ctx->min_cuwh;
    P->max_cu = is_i ? ctx->param.max_cu_intra : ctx->param.max_cu_inter, P->min_cu = is_i ? ctx->param.min_cu_intra : ctx->param.min_cu_inter;
    P->slice_qp = ctx->tile[core->tile_idx].qp, P->slice_num = ctx->slice_num;
}
static void shadow_whole_picture(XEVE_CTX *ctx, XEVE_CORE *core, const xo_sbac *entry)
{
    AP.valid = 0;
    if(!xo_eco || !xo_tile_end || !xo_tree_any || ctx->tile_cnt != 1 || getenv("XEVE_SHIM_SHADOW_NO_PICTURE")) return;
    const int is_i = ctx->sh->slice_type == SLICE_I, hs = ctx->param.cs_h_shift, nscu = ctx->w_scu * ctx->h_scu, hc = ctx->h >> hs, ctu = ctx->max_cuwh;
    XEVE_PIC    *pm = PIC_MODE(ctx);
    XEVE_PINTRA *pi = &ctx->pintra[core->thread_cnt];
    /* (mode_cu_init's QPs and the slice's lambdas: set_lambda has run for this slice in xeve_pic, the QPs follow from the tile QP) */
    core->qp = ctx->tile[core->tile_idx].qp, core->qp_y = GET_LUMA_QP(core->qp, ctx->sps.bit_depth_luma_minus8);
    core->qp_u = ctx->qp_chroma_dynamic[0][XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_u_offset)] + 6 * ctx->sps.bit_depth_chroma_minus8;
    core->qp_v = ctx->qp_chroma_dynamic[1][XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_v_offset)] + 6 * ctx->sps.bit_depth_chroma_minus8;
    xo_tree_params P;
    build_tree_params(ctx, core, &P);
    xo_pel   *mod[3] = {malloc(sizeof(pel) * pm->s_l * (ctx->h + 1)), malloc(sizeof(pel) * pm->s_c * (hc + 1)), malloc(sizeof(pel) * pm->s_c * (hc + 1))};
    uint32_t *scu = malloc(4 * nscu), *cum = malloc(4 * nscu);
    int8_t   *ipm = malloc(nscu), (*refi)[2] = malloc(sizeof(*refi) * nscu);
    int16_t (*mv)[2][2] = malloc(sizeof(*mv) * nscu);
    memcpy(mod[0], pm->y, sizeof(pel) * pm->s_l * ctx->h), memcpy(mod[1], pm->u, sizeof(pel) * pm->s_c * hc), memcpy(mod[2], pm->v, sizeof(pel) * pm->s_c * hc);
    memcpy(scu, ctx->map_scu, 4 * nscu), memcpy(cum, ctx->map_cu_mode, 4 * nscu), memcpy(ipm, ctx->map_ipm, nscu);
    memcpy(mv, ctx->map_mv, sizeof(*mv) * nscu), memcpy(refi, ctx->map_refi, sizeof(*refi) * nscu);
    xo_tree_inter TI;
    xo_refpic     tab[16];
    int ok = 1;
    if(!is_i) ok = tree_inter_setup(ctx, core, &TI, tab, mv, refi) == 0;
    if(ok) {
        static xo_ctu_data out;
        xo_sbac   state = *entry, next;
        const int num_refp[2] = {ctx->rpm.num_refp[REFP_0], ctx->rpm.num_refp[REFP_1]};
        const xo_pel *org[3] = {pi->o[Y_C], pi->o[U_C], pi->o[V_C]};
        if(AP.cap < (1 << 24)) AP.bytes = realloc(AP.bytes, 1 << 24), AP.cap = 1 << 24;
        AP.n = 0;
        for(int y0 = 0; y0 < ctx->h; y0 += ctu)
            for(int x0 = 0; x0 < ctx->w; x0 += ctu) {
                (void)xo_tree_any(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, scu, ipm, ctx->map_tidx, cum, &state, &P, is_i ? NULL : &TI, x0, y0, &out, &next);
                for(int j = 0; j < XEVE_MIN(ctu, ctx->h - y0) >> 2; j++) /* mode_analyze_lcu's tail: the CTU's coded flags reset */
                    for(int i = 0; i < XEVE_MIN(ctu, ctx->w - x0) >> 2; i++) scu[((y0 >> 2) + j) * ctx->w_scu + (x0 >> 2) + i] &= 0x7FFFFFFFu;
                AP.n += xo_eco(&state, &out, &P, num_refp, scu, ipm, ctx->map_tidx, cum, x0, y0, AP.bytes + AP.n, AP.cap - AP.n);
            }
        AP.n_ctus = AP.n;
        AP.n += xo_tile_end(&state, AP.bytes + AP.n, AP.cap - AP.n);
        AP.pos0 = bsw_pos(&ctx->bs[core->thread_cnt]), AP.valid = 1;
        if(getenv("XEVE_SHIM_SHADOW_DUMP")) { /* [poc, n, bytes] per picture: the test looks for them at the end of the slice NAL units of the bitstream file */
            FILE *f = fopen(getenv("XEVE_SHIM_SHADOW_DUMP"), "ab");
            const int32_t hd[2] = {(int32_t)ctx->poc.poc_val, AP.n};
            if(f) fwrite(hd, 4, 2, f), fwrite(AP.bytes, 1, (size_t)AP.n, f), fclose(f);
        }
    }
    free(mod[0]), free(mod[1]), free(mod[2]), free(scu), free(cum), free(ipm), free(mv), free(refi);
}
static int shim_shadow_loop_filter(XEVE_CTX *ctx, XEVE_CORE *core)
{
    if(AP.valid) { /* the reference's CTU loop has written the CTUs once (the pass that feeds the mode decision; xeve_pic writes them again behind it and a third time, after
                    * the loop filter, into the output): the first pass's bytes must be the oracle's, up to what the coder still holds at the tile's end */
        const XEVE_BSW *bs = &ctx->bs[0];
        const long n = bsw_pos(bs) - AP.pos0;
        int  bad = n < AP.n_ctus;
        long k = 0;
        while(!bad && k < AP.n_ctus && AP.bytes[k] == bsw_byte(bs, AP.pos0 + k)) k++;
        bad |= k < AP.n_ctus;
        if(bad && ap_bad < 4) fprintf(stderr, "[shadow picture] poc %d: the oracle's slice data (%d bytes before the tile's end) differs from the reference's at byte %ld\n", (int)ctx->poc.poc_val, AP.n_ctus, k);
        ap_pics++, ap_bad += bad, ap_bytes += (unsigned long long)AP.n_ctus;
        AP.valid = 0;
    }
    return orig_shadow_loop_filter(ctx, core);
}

static int shim_mode_analyze_lcu(XEVE_CTX *ctx, XEVE_CORE *core)
{
    if(ctx->param.threads == 1 && xo_eco && core->lcu_num == 0 && !ctx->pps.cu_qp_delta_enabled_flag && !ctx->param.rdo_dbk_switch && !ctx->param.tool_iqt && !ctx->sps.tool_admvp &&
       ctx->log2_max_cuwh == 6 && ctx->sps.chroma_format_idc != 2) {
        xo_sbac first;
        sbac_to_flat(&first, &core->s_curr_best[ctx->log2_max_cuwh - 2][ctx->log2_max_cuwh - 2]);
        shadow_whole_picture(ctx, core, &first);
    }
    if(ctx->param.threads == 1 && xo_eco) { /* (the entry state of this CTU = the writer's state after the previous one) */
        xo_sbac now;
        sbac_to_flat(&now, &core->s_curr_best[ctx->log2_max_cuwh - 2][ctx->log2_max_cuwh - 2]);
        shadow_writer_check(ctx, core, &now);
    }
    const int L = ctx->log2_max_cuwh - 2, idc = ctx->sps.chroma_format_idc, ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift;
    const int is_i = ctx->sh->slice_type == SLICE_I;
    xo_tree_inter  TI;
    xo_refpic      tab[16];
    const int nscu0 = ctx->w_scu * ctx->h_scu;
    int16_t (*m_mv)[2][2] = NULL;
    int8_t  (*m_refi)[2]  = NULL;
    if(ctx->pps.cu_qp_delta_enabled_flag || ctx->param.rdo_dbk_switch || ctx->param.tool_iqt || ctx->sps.tool_admvp || ctx->log2_max_cuwh != 6 || idc == 2 ||
       ctx->param.threads != 1 || (!is_i && (!xo_tree_any || getenv("XEVE_SHIM_SHADOW_I_ONLY")))) {
        shadow_skipped++;
        return orig_mode_analyze_lcu(ctx, core);
    }
    if(!is_i) {
        /* (mode_cu_init's QPs: the inter parameters are read before the reference has run on this CTU) */
        core->qp = ctx->tile[core->tile_idx].qp, core->qp_y = GET_LUMA_QP(core->qp, ctx->sps.bit_depth_luma_minus8);
        core->qp_u = ctx->qp_chroma_dynamic[0][XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_u_offset)] + 6 * ctx->sps.bit_depth_chroma_minus8;
        core->qp_v = ctx->qp_chroma_dynamic[1][XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_v_offset)] + 6 * ctx->sps.bit_depth_chroma_minus8;
        m_mv = malloc(sizeof(*m_mv) * nscu0), m_refi = malloc(sizeof(*m_refi) * nscu0);
        memcpy(m_mv, ctx->map_mv, sizeof(*m_mv) * nscu0), memcpy(m_refi, ctx->map_refi, sizeof(*m_refi) * nscu0);
        if(tree_inter_setup(ctx, core, &TI, tab, m_mv, m_refi) != 0) {
            free(m_mv), free(m_refi);
            shadow_skipped++;
            return orig_mode_analyze_lcu(ctx, core);
        }
    }
    XEVE_PIC *pm = PIC_MODE(ctx);
    XEVE_PINTRA *pi = &ctx->pintra[core->thread_cnt];
    const int nscu = ctx->w_scu * ctx->h_scu, hl = ctx->h, hc = ctx->h >> hs;
    /* snapshot */
    xo_pel *mod[3] = {malloc(sizeof(pel) * pm->s_l * (hl + 1)), malloc(sizeof(pel) * pm->s_c * (hc + 1)), malloc(sizeof(pel) * pm->s_c * (hc + 1))};
    memcpy(mod[0], pm->y, sizeof(pel) * pm->s_l * hl), memcpy(mod[1], pm->u, sizeof(pel) * pm->s_c * hc), memcpy(mod[2], pm->v, sizeof(pel) * pm->s_c * hc);
    uint32_t *m_scu = malloc(4 * nscu), *m_cum = malloc(4 * nscu);
    int8_t   *m_ipm = malloc(nscu);
    memcpy(m_scu, ctx->map_scu, 4 * nscu), memcpy(m_cum, ctx->map_cu_mode, 4 * nscu), memcpy(m_ipm, ctx->map_ipm, nscu);
    xo_sbac entry, next, ref_next;
    sbac_to_flat(&entry, &core->s_curr_best[L][L]);
    const int x0 = core->x_pel, y0 = core->y_pel, lcu = core->lcu_num;

    const int rc = orig_mode_analyze_lcu(ctx, core); /* the reference decides; its results stay */

    xo_tree_params P;
    memset(&P, 0, sizeof(P));
    P.ip.w_scu = ctx->w_scu, P.ip.h_scu = ctx->h_scu, P.ip.slice_type = 2, P.ip.chroma_format_idc = idc, P.ip.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8;
    P.ip.tool_iqt = 0, P.ip.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;
    P.ip.qp[0] = core->qp_y, P.ip.qp[1] = core->qp_u, P.ip.qp[2] = core->qp_v; /* (mode_cu_init derives them from the tile QP: the same for every CU without delta QP) */
    for(int c = 0; c < 3; c++) P.ip.lambda[c] = core->lambda[c];
    P.ip.sqrt_lambda0 = core->sqrt_lambda[0], P.ip.dist_chroma_weight[0] = core->dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    P.pic_w = ctx->w, P.pic_h = ctx->h, P.log2_ctu = ctx->log2_max_cuwh, P.min_cuwh = ctx->min_cuwh;
    P.max_cu = is_i ? ctx->param.max_cu_intra : ctx->param.max_cu_inter, P.min_cu = is_i ? ctx->param.min_cu_intra : ctx->param.min_cu_inter;
    P.slice_qp = ctx->tile[core->tile_idx].qp, P.slice_num = ctx->slice_num;
    static __thread xo_ctu_data out;
    const xo_pel *org[3] = {pi->o[Y_C], pi->o[U_C], pi->o[V_C]};
    if(!is_i) P.ip.slice_type = ctx->sh->slice_type;
    if(getenv("XEVE_SHIM_TREE_GOLDEN") && golden_wanted(lcu)) golden_dump(ctx, core, &P, is_i ? NULL : &TI, mod, m_scu, m_ipm, m_cum, (const int16_t(*)[2][2])m_mv, (const int8_t(*)[2])m_refi, &entry, x0, y0, lcu);
    if(is_i) (void)xo_tree(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, m_scu, m_ipm, ctx->map_tidx, m_cum, &entry, &P, x0, y0, &out, &next);
    else {
        (void)xo_tree_any(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, m_scu, m_ipm, ctx->map_tidx, m_cum, &entry, &P, &TI, x0, y0, &out, &next);
        shadow_inter_ctus++;
    }

    /* compare */
    int bad = 0;
    const XEVE_CU_DATA *cd = &ctx->map_cu_data[lcu];
    const int nu = 16, wu = XEVE_MIN(nu, ctx->w_scu - (x0 >> 2)), hu = XEVE_MIN(nu, ctx->h_scu - (y0 >> 2));
#define BAD(what, ...) do { if(bad++ < 6 && shadow_bad < 6) fprintf(stderr, "[shadow] CTU %d (%d,%d): " what "\n", lcu, x0, y0, __VA_ARGS__); } while(0)
    /* split modes: every unit of the CTU -- the flag of a node the picture edge cuts sits at the node's centre, which may lie outside the picture, and the
     * bitstream writer reads it there (xeve_get_split_mode, xeve_util.c:1125-1144) */
    for(int u = 0; u < nu * nu; u++)
        for(int d = 0; d < XO_CU_DEPTHS; d++)
            if(cd->split_mode[d][SQUARE][u] != out.split_mode[d][u]) BAD("split_mode[%d][%d] %d vs %d", d, u, cd->split_mode[d][SQUARE][u], out.split_mode[d][u]);
    for(int j = 0; j < hu; j++)
        for(int i = 0; i < wu; i++) {
            const int u = j * nu + i;
            if(cd->pred_mode[u] != out.pred_mode[u]) BAD("pred_mode[%d] %d vs %d", u, cd->pred_mode[u], out.pred_mode[u]);
            if(cd->ipm[0][u] != out.ipm[0][u] || (idc && cd->ipm[1][u] != out.ipm[1][u])) BAD("ipm[%d] %d,%d vs %d,%d", u, cd->ipm[0][u], cd->ipm[1][u], out.ipm[0][u], out.ipm[1][u]);
            if(cd->depth[u] != out.depth[u]) BAD("depth[%d] %d vs %d", u, cd->depth[u], out.depth[u]);
            for(int c = 0; c < (idc ? 3 : 1); c++)
                if(cd->nnz[c][u] != out.nnz[c][u]) BAD("nnz[%d][%d] %d vs %d", c, u, cd->nnz[c][u], out.nnz[c][u]);
            if(cd->map_scu[u] != out.map_scu[u]) BAD("map_scu[%d] %08x vs %08x", u, cd->map_scu[u], out.map_scu[u]);
            if(cd->map_cu_mode[u] != out.map_cu_mode[u]) BAD("map_cu_mode[%d] %08x vs %08x", u, cd->map_cu_mode[u], out.map_cu_mode[u]);
            const int g = ((y0 >> 2) + j) * ctx->w_scu + (x0 >> 2) + i;
            if(!is_i) {
                if(cd->skip_flag[u] != (out.pred_mode[u] == MODE_SKIP)) BAD("skip_flag[%d] %d (mode %d)", u, cd->skip_flag[u], out.pred_mode[u]);
                for(int l = 0; l < 2; l++) {
                    if(cd->refi[u][l] != out.refi[u][l]) BAD("refi[%d][%d] %d vs %d", u, l, cd->refi[u][l], out.refi[u][l]);
                    if(cd->mv[u][l][0] != out.mv[u][l][0] || cd->mv[u][l][1] != out.mv[u][l][1]) BAD("mv[%d][%d] (%d,%d) vs (%d,%d)", u, l, cd->mv[u][l][0], cd->mv[u][l][1], out.mv[u][l][0], out.mv[u][l][1]);
                    if(out.pred_mode[u] != MODE_INTRA && out.refi[u][l] >= 0 && out.pred_mode[u] != MODE_DIR) {
                        if(cd->mvp_idx[u][l] != out.mvp_idx[u][l]) BAD("mvp_idx[%d][%d] %d vs %d", u, l, cd->mvp_idx[u][l], out.mvp_idx[u][l]);
                        if(out.pred_mode[u] == MODE_INTER && (cd->mvd[u][l][0] != out.mvd[u][l][0] || cd->mvd[u][l][1] != out.mvd[u][l][1])) BAD("mvd[%d][%d] (%d,%d) vs (%d,%d)", u, l, cd->mvd[u][l][0], cd->mvd[u][l][1], out.mvd[u][l][0], out.mvd[u][l][1]);
                    }
                    if(ctx->map_refi[g][l] != m_refi[g][l]) BAD("ctx->map_refi[%d][%d] %d vs %d", g, l, ctx->map_refi[g][l], m_refi[g][l]);
                    if(ctx->map_mv[g][l][0] != m_mv[g][l][0] || ctx->map_mv[g][l][1] != m_mv[g][l][1]) BAD("ctx->map_mv[%d][%d] (%d,%d) vs (%d,%d)", g, l, ctx->map_mv[g][l][0], ctx->map_mv[g][l][1], m_mv[g][l][0], m_mv[g][l][1]);
                }
            }
            if((ctx->map_scu[g] | (1u << 31)) != m_scu[g]) BAD("ctx->map_scu[%d] %08x vs %08x", g, ctx->map_scu[g], m_scu[g]);
            if(ctx->map_ipm[g] != m_ipm[g]) BAD("ctx->map_ipm[%d] %d vs %d", g, ctx->map_ipm[g], m_ipm[g]);
        }
    for(int c = 0; c < (idc ? 3 : 1); c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, cs = 64 >> sx, w = (wu * 4) >> sx, h = (hu * 4) >> sy, s = c ? pm->s_c : pm->s_l;
        const pel *pr = (c == 0 ? pm->y : c == 1 ? pm->u : pm->v) + (y0 >> sy) * s + (x0 >> sx);
        const xo_pel *po = mod[c] + (y0 >> sy) * s + (x0 >> sx);
        for(int yy = 0; yy < h; yy++)
            for(int xx = 0; xx < w; xx++) {
                if(cd->nnz[c][(((yy << sy) >> 2) * nu) + ((xx << sx) >> 2)] /* (a CU without coded levels keeps stale ones in the reference) */ &&
                   cd->coef[c][yy * cs + xx] != out.coef[c][yy * cs + xx]) BAD("coef[%d] (%d,%d) %d vs %d", c, xx, yy, cd->coef[c][yy * cs + xx], out.coef[c][yy * cs + xx]);
                if(cd->reco[c][yy * cs + xx] != out.reco[c][yy * cs + xx]) BAD("reco[%d] (%d,%d) %d vs %d", c, xx, yy, cd->reco[c][yy * cs + xx], out.reco[c][yy * cs + xx]);
                if(pr[yy * s + xx] != po[yy * s + xx]) BAD("picture[%d] (%d,%d) %d vs %d", c, xx, yy, pr[yy * s + xx], po[yy * s + xx]);
            }
    }
    sbac_to_flat(&ref_next, &core->s_next_best[L][L]);
    if(memcmp(&ref_next, &next, sizeof(next))) BAD("exit coder state differs (range %u vs %u)", ref_next.range, next.range);
    shadow_ctus++;
    if(bad) shadow_bad++;
    if(xo_eco && !bad) { /* what the writer is about to write: kept for the check at the next CTU's entry */
        const int nscu2 = ctx->w_scu * ctx->h_scu;
        W.scu = realloc(W.scu, 4 * nscu2), W.cum = realloc(W.cum, 4 * nscu2), W.ipm = realloc(W.ipm, nscu2);
        memcpy(W.scu, ctx->map_scu, 4 * nscu2), memcpy(W.cum, ctx->map_cu_mode, 4 * nscu2), memcpy(W.ipm, ctx->map_ipm, nscu2);
        W.valid = 1, W.lcu = lcu, W.x0 = x0, W.y0 = y0, W.pic = (const void *)PIC_MODE(ctx), W.entry = entry, W.out = out, W.P = P;
        W.num_refp[0] = ctx->rpm.num_refp[REFP_0], W.num_refp[1] = ctx->rpm.num_refp[REFP_1];
        W.byte_pos = bsw_pos(&ctx->bs[core->thread_cnt]);
    }
    free(mod[0]), free(mod[1]), free(mod[2]), free(m_scu), free(m_cum), free(m_ipm), free(m_mv), free(m_refi);
    return rc;
}


/* XEVE_HIP_SHIM_TREE=1 (with XEVE_HIP_LIB): ROUTE MODE for the CTU mode decision of I pictures -- ctx->fn_mode_analyze_lcu served by the device-side tree walk
 * (xeve_hip_mode_analyze_ctu_intra_host): ONE host<->device exchange per CTU instead of one per CU.  The adapter hands over what the walk reads (the original,
 * the picture being reconstructed, the unit maps, the entry coder state) and stores what mode_analyze_lcu leaves behind (xeve_mode.c:2518-2610): the CTU's
 * XEVE_CU_DATA in ctx->map_cu_data (copy_to_cu_data's fields for an intra CU, :868-1034), the context maps (update_to_ctx_map :2445-2516 + update_map_scu
 * :1036-1127), the reconstruction in PIC_MODE, the coded flags reset (:2591-2607).  XEVE_SHIM_TREE_ORACLE=<path of libxeve_oracle.so> runs the same adapter with
 * the oracle's restatement as the engine: CPU only, to test the adapter's stores without a GPU. */
typedef int (*hip_tree_host_fn)(const pel *const *, int, int, pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                                const xo_tree_params *, int, int, xo_ctu_data *, xo_sbac *, double *);
static hip_tree_host_fn hip_tree_host;
typedef struct { const void *refp; int s_ref_l, s_ref_c; hip_inter_params ipar; int16_t *map_mv; int8_t *map_refi; const int16_t *col0, *col1; const void *coef_l, *coef_c; int ecu_depth, pad_; } hip_tree_inter;
typedef int (*hip_tree_any_host_fn)(const pel *const *, int, int, pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                                    const xo_tree_params *, const hip_tree_inter *, int, int, int, int, xo_ctu_data *, xo_sbac *, double *);
static hip_tree_any_host_fn hip_tree_any_host;
static int               tree_engine_oracle;
static unsigned long long tree_calls, tree_fallbacks, tree_check_ctus, tree_check_bad;
static int                tree_check;
static double             tree_seconds;

static int shim_route_mode_analyze_lcu(XEVE_CTX *ctx, XEVE_CORE *core)
{
    const int L = ctx->log2_max_cuwh - 2, idc = ctx->sps.chroma_format_idc, ws = ctx->param.cs_w_shift, hs = ctx->param.cs_h_shift;
    const int is_i = ctx->sh->slice_type == SLICE_I;
    if(ctx->pps.cu_qp_delta_enabled_flag || ctx->param.rdo_dbk_switch || ctx->param.tool_iqt || ctx->sps.tool_admvp || ctx->log2_max_cuwh > 6 || ctx->log2_max_cuwh < 3 ||
       idc == 2 || (!is_i && (ctx->log2_max_cuwh != 6 || !(tree_engine_oracle ? (void *)xo_tree_any : (void *)hip_tree_any_host)))) {
        __sync_fetch_and_add(&tree_fallbacks, 1);
        return orig_mode_analyze_lcu(ctx, core);
    }
    XEVE_PIC    *pm = PIC_MODE(ctx);
    XEVE_PINTRA *pi = &ctx->pintra[core->thread_cnt];
    XEVE_MODE   *mi = &ctx->mode[core->thread_cnt];
    memset(mi->mvp_idx, 0, sizeof(u8) * REFP_NUM), memset(mi->mvd, 0, sizeof(s16) * REFP_NUM * MV_D);
    /* what mode_cu_init (:1157-1220) derives for every CU of the slice when there is no delta QP */
    core->qp = ctx->tile[core->tile_idx].qp;
    core->qp_y = GET_LUMA_QP(core->qp, ctx->sps.bit_depth_luma_minus8);
    {
        const int qp_i_cb = XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_u_offset);
        const int qp_i_cr = XEVE_CLIP3(-6 * ctx->sps.bit_depth_chroma_minus8, 57, core->qp + ctx->sh->qp_v_offset);
        core->qp_u = ctx->qp_chroma_dynamic[0][qp_i_cb] + 6 * ctx->sps.bit_depth_chroma_minus8;
        core->qp_v = ctx->qp_chroma_dynamic[1][qp_i_cr] + 6 * ctx->sps.bit_depth_chroma_minus8;
    }
    xo_tree_params P;
    memset(&P, 0, sizeof(P));
    P.ip.w_scu = ctx->w_scu, P.ip.h_scu = ctx->h_scu, P.ip.slice_type = 2, P.ip.chroma_format_idc = idc, P.ip.bit_depth = ctx->sps.bit_depth_luma_minus8 + 8;
    P.ip.tool_iqt = 0, P.ip.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;
    P.ip.qp[0] = core->qp_y, P.ip.qp[1] = core->qp_u, P.ip.qp[2] = core->qp_v;
    for(int c = 0; c < 3; c++) P.ip.lambda[c] = core->lambda[c];
    P.ip.sqrt_lambda0 = core->sqrt_lambda[0], P.ip.dist_chroma_weight[0] = core->dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = core->dist_chroma_weight[1];
    P.pic_w = ctx->w, P.pic_h = ctx->h, P.log2_ctu = ctx->log2_max_cuwh, P.min_cuwh = ctx->min_cuwh;
    P.max_cu = is_i ? ctx->param.max_cu_intra : ctx->param.max_cu_inter, P.min_cu = is_i ? ctx->param.min_cu_intra : ctx->param.min_cu_inter;
    P.ip.slice_type = ctx->sh->slice_type;
    P.slice_qp = ctx->tile[core->tile_idx].qp, P.slice_num = ctx->slice_num;
    static __thread xo_ctu_data out;
    xo_sbac entry, next;
    sbac_to_flat(&entry, &core->s_curr_best[L][L]);
    const int  x0 = core->x_pel, y0 = core->y_pel;
    const pel *org[3] = {pi->o[Y_C], pi->o[U_C], pi->o[V_C]};
    pel       *mod[3] = {pm->y, pm->u, pm->v};
    double     cost = 0;
    struct timespec t0, t1;
    xo_tree_inter TI;
    xo_refpic     tab[16];
    if(!is_i && tree_inter_setup(ctx, core, &TI, tab, (int16_t(*)[2][2])ctx->map_mv, (int8_t(*)[2])ctx->map_refi) != 0) {
        __sync_fetch_and_add(&tree_fallbacks, 1);
        return orig_mode_analyze_lcu(ctx, core);
    }
    /* XEVE_SHIM_TREE_CHECK=<libxeve_oracle.so> (with the GPU as the engine): the oracle walks a snapshot of the same inputs first and the two results are compared per CTU
     * -- locates a deviation of the device walk inside a long encode (which CTU, which field) */
    static __thread xo_ctu_data chk;
    xo_sbac   chk_next;
    xo_pel   *cm[3] = {NULL, NULL, NULL};
    uint32_t *c_scu = NULL, *c_cum = NULL;
    int8_t   *c_ipm = NULL, (*c_refi)[2] = NULL;
    int16_t (*c_mv)[2][2] = NULL;
    const int do_check = tree_check && !tree_engine_oracle;
    if(do_check) {
        const int nscu = ctx->w_scu * ctx->h_scu, hc = ctx->h >> hs;
        cm[0] = malloc(sizeof(pel) * pm->s_l * (ctx->h + 1)), cm[1] = malloc(sizeof(pel) * pm->s_c * (hc + 1)), cm[2] = malloc(sizeof(pel) * pm->s_c * (hc + 1));
        memcpy(cm[0], pm->y, sizeof(pel) * pm->s_l * ctx->h), memcpy(cm[1], pm->u, sizeof(pel) * pm->s_c * hc), memcpy(cm[2], pm->v, sizeof(pel) * pm->s_c * hc);
        c_scu = malloc(4 * nscu), c_cum = malloc(4 * nscu), c_ipm = malloc(nscu), c_mv = malloc(sizeof(*c_mv) * nscu), c_refi = malloc(sizeof(*c_refi) * nscu);
        memcpy(c_scu, ctx->map_scu, 4 * nscu), memcpy(c_cum, ctx->map_cu_mode, 4 * nscu), memcpy(c_ipm, ctx->map_ipm, nscu);
        memcpy(c_mv, ctx->map_mv, sizeof(*c_mv) * nscu), memcpy(c_refi, ctx->map_refi, sizeof(*c_refi) * nscu);
        if(is_i) (void)xo_tree((const xo_pel *const *)org, pi->s_o[Y_C], pi->s_o[U_C], cm, pm->s_l, pm->s_c, c_scu, c_ipm, ctx->map_tidx, c_cum, &entry, &P, x0, y0, &chk, &chk_next);
        else {
            xo_tree_inter TC = TI;
            TC.map_mv = c_mv, TC.map_refi = c_refi;
            (void)xo_tree_any((const xo_pel *const *)org, pi->s_o[Y_C], pi->s_o[U_C], cm, pm->s_l, pm->s_c, c_scu, c_ipm, ctx->map_tidx, c_cum, &entry, &P, &TC, x0, y0, &chk, &chk_next);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if(!is_i) {
        if(tree_engine_oracle) cost = xo_tree_any((const xo_pel *const *)org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, ctx->map_scu, ctx->map_ipm, ctx->map_tidx, ctx->map_cu_mode, &entry, &P, &TI, x0, y0, &out, &next);
        else {
            XEVE_PINTER *pin = &ctx->pinter[core->thread_cnt];
            XEVE_PIC    *any = pin->refp[0][REFP_0].pic;
            hip_tree_inter HI;
            memset(&HI, 0, sizeof(HI));
            HI.refp = tab, HI.s_ref_l = TI.s_ref_l, HI.s_ref_c = TI.s_ref_c, HI.map_mv = (int16_t *)ctx->map_mv, HI.map_refi = (int8_t *)ctx->map_refi;
            HI.col0 = (const int16_t *)TI.col0, HI.col1 = (const int16_t *)TI.col1, HI.coef_l = pin->mc_l_coeff, HI.coef_c = pin->mc_c_coeff, HI.ecu_depth = TI.ecu_depth;
            memcpy(&HI.ipar.rdo, &TI.ipar.rdo, sizeof(HI.ipar.rdo)), memcpy(&HI.ipar.me, &TI.ipar.me.me, sizeof(TI.ipar.me.me));
            HI.ipar.me.hpel_cnt = TI.ipar.me.spel.hpel_cnt, HI.ipar.me.qpel_cnt = TI.ipar.me.spel.qpel_cnt;
            memcpy(HI.ipar.refi_bits, TI.ipar.refi_bits, sizeof(HI.ipar.refi_bits)), memcpy(HI.ipar.range_recentre, TI.ipar.range_recentre, sizeof(HI.ipar.range_recentre));
            HI.ipar.max_cand = TI.ipar.max_cand, HI.ipar.poc = TI.ipar.poc, HI.ipar.col_list_poc0 = TI.ipar.col_list_poc0, HI.ipar.skip_th = TI.ipar.skip_th;
            if(hip_tree_any_host(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, ctx->map_scu, ctx->map_ipm, ctx->map_tidx, ctx->map_cu_mode, &entry, &P, &HI, any->pad_l,
                                 any->pad_c, x0, y0, &out, &next, &cost) != 0) {
                fprintf(stderr, "[xeve_hip_shim] xeve_hip_mode_analyze_ctu_host: %s\n", hip_err ? hip_err() : "?");
                abort();
            }
        }
    }
    else if(tree_engine_oracle) cost = xo_tree((const xo_pel *const *)org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, ctx->map_scu, ctx->map_ipm, ctx->map_tidx, ctx->map_cu_mode, &entry, &P, x0, y0, &out, &next);
    else if(hip_tree_host(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, ctx->map_scu, ctx->map_ipm, ctx->map_tidx, ctx->map_cu_mode, &entry, &P, x0, y0, &out, &next, &cost) != 0) {
        fprintf(stderr, "[xeve_hip_shim] xeve_hip_mode_analyze_ctu_intra_host: %s\n", hip_err ? hip_err() : "?");
        abort();
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)cost;
    if(do_check) {
        int bad = 0;
#define CHK(field) do { if(memcmp(out.field, chk.field, sizeof(out.field))) { if(tree_check_bad < 8) { size_t k_ = 0; while(((const char *)out.field)[k_] == ((const char *)chk.field)[k_]) k_++; \
            fprintf(stderr, "[tree check] CTU %d (%d,%d) slice %d: %s differs at byte %zu\n", core->lcu_num, x0, y0, ctx->sh->slice_type, #field, k_); } bad = 1; } } while(0)
        CHK(split_mode); CHK(pred_mode); CHK(ipm); CHK(depth); CHK(nnz); CHK(map_scu); CHK(map_cu_mode); CHK(coef); CHK(reco); CHK(mv); CHK(mvd); CHK(refi); CHK(mvp_idx);
#undef CHK
        if(memcmp(&next, &chk_next, sizeof(next))) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU %d (%d,%d): exit coder state differs\n", core->lcu_num, x0, y0); bad = 1; }
        const int nscu = ctx->w_scu * ctx->h_scu;
        if(memcmp(c_scu, ctx->map_scu, 4 * nscu) || memcmp(c_ipm, ctx->map_ipm, nscu) || memcmp(c_cum, ctx->map_cu_mode, 4 * nscu)) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU %d (%d,%d): unit maps differ\n", core->lcu_num, x0, y0); bad = 1; }
        if(!is_i && (memcmp(c_mv, ctx->map_mv, sizeof(*c_mv) * nscu) || memcmp(c_refi, ctx->map_refi, sizeof(*c_refi) * nscu))) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU %d (%d,%d): motion maps differ\n", core->lcu_num, x0, y0); bad = 1; }
        if(memcmp(cm[0], pm->y, sizeof(pel) * pm->s_l * ctx->h)) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU %d (%d,%d): luma picture differs\n", core->lcu_num, x0, y0); bad = 1; }
        tree_check_ctus++, tree_check_bad += bad;
        free(cm[0]), free(cm[1]), free(cm[2]), free(c_scu), free(c_cum), free(c_ipm), free(c_mv), free(c_refi);
    }
    /* the CTU's data for the entropy coder (xeve_eco_tree -> xeve_eco_unit reads ctx->map_cu_data[lcu_num]) */
    XEVE_CU_DATA *cd = &ctx->map_cu_data[core->lcu_num];
    const int nu = 1 << L, ctu = 1 << ctx->log2_max_cuwh, wu = XEVE_MIN(nu, ctx->w_scu - (x0 >> 2)), hu = XEVE_MIN(nu, ctx->h_scu - (y0 >> 2));
    for(int u = 0; u < nu * nu; u++) /* (every unit: the flag of a node the picture edge cuts sits at the node's centre, possibly outside the picture) */
        for(int d = 0; d < XO_CU_DEPTHS; d++) cd->split_mode[d][SQUARE][u] = out.split_mode[d][u];
    for(int j = 0; j < hu; j++)
        for(int i = 0; i < wu; i++) {
            const int u = j * nu + i, g = ((y0 >> 2) + j) * ctx->w_scu + (x0 >> 2) + i;
            cd->pred_mode[u] = out.pred_mode[u], cd->pred_mode_chroma[u] = out.pred_mode[u], cd->skip_flag[u] = out.pred_mode[u] == MODE_SKIP, cd->mmvd_flag[u] = 0, cd->affine_flag[u] = 0;
            cd->dmvr_flag[u] = 0, cd->mvr_idx[u] = 0, cd->bi_idx[u] = 0;
            cd->ipm[0][u] = out.ipm[0][u], cd->ipm[1][u] = out.ipm[1][u], cd->depth[u] = out.depth[u];
            cd->qp_y[u] = core->qp_y, cd->qp_u[u] = core->qp_u, cd->qp_v[u] = core->qp_v;
            for(int c = 0; c < N_C; c++) {
                cd->nnz[c][u] = out.nnz[c][u];
                cd->nnz_sub[c][0][u] = out.nnz[c][u], cd->nnz_sub[c][1][u] = cd->nnz_sub[c][2][u] = cd->nnz_sub[c][3][u] = 0; /* one transform block per CU up to 64x64 */
            }
            cd->map_scu[u] = out.map_scu[u], cd->map_cu_mode[u] = out.map_cu_mode[u];
            memcpy(cd->mv[u], out.mv[u], sizeof(cd->mv[u])), memcpy(cd->unrefined_mv[u], out.mv[u], sizeof(cd->unrefined_mv[u])), memcpy(cd->mvd[u], out.mvd[u], sizeof(cd->mvd[u]));
            cd->refi[u][REFP_0] = out.refi[u][0], cd->refi[u][REFP_1] = out.refi[u][1], cd->mvp_idx[u][REFP_0] = out.mvp_idx[u][0], cd->mvp_idx[u][REFP_1] = out.mvp_idx[u][1];
            /* the context maps: motion (the walk keeps them in P / B slices; an I slice's units are intra: zero vectors, no reference), unrefined motion = motion
             * without DMVR, depth (update_map_scu); then the coded flag reset */
            memcpy(ctx->map_mv[g], out.mv[u], sizeof(ctx->map_mv[g])), memcpy(ctx->map_unrefined_mv[g], out.mv[u], sizeof(ctx->map_unrefined_mv[g]));
            ctx->map_refi[g][REFP_0] = out.refi[u][0], ctx->map_refi[g][REFP_1] = out.refi[u][1];
            ctx->map_depth[g] = out.depth[u];
            MCU_CLR_COD(ctx->map_scu[g]);
        }
    for(int c = 0; c < (idc ? 3 : 1); c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, cs = ctu >> sx, w = (wu * 4) >> sx, h = (hu * 4) >> sy;
        for(int yy = 0; yy < h; yy++) {
            memcpy(cd->coef[c] + yy * cs, out.coef[c] + yy * cs, sizeof(s16) * w);
            memcpy(cd->reco[c] + yy * cs, out.reco[c] + yy * cs, sizeof(pel) * w);
        }
    }
    /* core->s_next_best[L][L]: the coder state of the winner (the next CTU starts from the bitstream writer's own state, xeve_enc.c:139, not from this one) */
    XEVE_SBAC *nb = &core->s_next_best[L][L];
    *nb = core->s_curr_best[L][L];
    nb->range = next.range, nb->code = next.code, nb->code_bits = next.code_bits, nb->stacked_ff = next.stacked_ff, nb->stacked_zero = next.stacked_zero;
    nb->pending_byte = next.pending_byte, nb->is_pending_byte = next.is_pending_byte, nb->bitcounter = next.bitcounter, nb->bin_counter = next.bin_counter;
#define F(name, at, n) memcpy(nb->ctx.name, next.ctx + at, 2 * n);
    SBAC_MAP(F)
#undef F
    __sync_fetch_and_add(&tree_calls, 1);
    tree_seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec); /* (per-thread sums would be exact; this is a report line) */
    return XEVE_OK;
}

static void report(void)
{
    if(hip_resident_stats) {
        unsigned long long pics = 0, up = 0, bytes = 0, hits = 0;
        hip_resident_stats(&pics, &up, &bytes, &hits);
        fprintf(stderr, "[xeve_hip_shim] resident pictures: %llu pictures announced, %llu planes uploaded (%llu bytes), %llu plane look-ups served from HBM\n", pics, up, bytes, hits);
    }
    if(inter_calls || inter_fallbacks) fprintf(stderr, "[xeve_hip_shim] CUs whose whole inter analysis ran on the GPU: %llu (left to the reference: %llu)\n", inter_calls, inter_fallbacks);
    if(intra_calls || intra_fallbacks) fprintf(stderr, "[xeve_hip_shim] CUs whose intra analysis ran on the GPU: %llu (left to the reference: %llu)\n", intra_calls, intra_fallbacks);
    if(tree_calls || tree_fallbacks) fprintf(stderr, "[xeve_hip_shim] CTUs whose whole mode decision ran on the %s: %llu (left to the reference: %llu), %.1f ms per CTU\n", tree_engine_oracle ? "oracle (CPU)" : "GPU", tree_calls, tree_fallbacks, tree_calls ? 1e3 * tree_seconds / (double)tree_calls : 0.0);
    if(ap_pics) fprintf(stderr, "[xeve_hip_shim] shadow pictures: %llu pictures decided and written by the oracle on its own (%llu bytes of slice data), %llu differ from the reference's\n", ap_pics, ap_bytes, ap_bad);
    if(eco_ctus) fprintf(stderr, "[xeve_hip_shim] shadow writer: %llu CTUs written by the oracle beside xeve_eco_tree (%llu bytes of bitstream compared), %llu differ\n", eco_ctus, eco_bytes, eco_bad);
    if(tree_check) fprintf(stderr, "[xeve_hip_shim] device walk checked against the oracle per CTU: %llu CTUs, %llu differ\n", tree_check_ctus, tree_check_bad);
    if(shadow_ctus || shadow_skipped) fprintf(stderr, "[xeve_hip_shim] shadow tree walk: %llu CTUs compared, %llu differ (%llu not covered), %llu of them in P / B pictures\n", shadow_ctus, shadow_bad, shadow_skipped, shadow_inter_ctus);
    if(inter_calls) fprintf(stderr, "[xeve_hip_shim] time inside the GPU calls: %.2f s = %.0f us per CU\n", inter_seconds, 1e6 * inter_seconds / (double)inter_calls);
    if(hip_table_calls) fprintf(stderr, "[xeve_hip_shim] dispatch-table calls served by HIP: %llu\n", hip_table_calls());
    if(eco_calls) fprintf(stderr, "[xeve_hip_shim] CUs whose coefficient bits were counted on the GPU: %llu\n", eco_calls);
    if(tq_calls) fprintf(stderr, "[xeve_hip_shim] transform blocks quantised (RDOQ) on the GPU: %llu, dequantised + inverse transformed: %llu\n", tq_calls, itdq_calls);
    if(mc_calls) fprintf(stderr, "[xeve_hip_shim] CU predictions (xeve_mc) made on the GPU: %llu\n", mc_calls);
    if(me_calls) fprintf(stderr, "[xeve_hip_shim] motion searches (pinter_me_epzs) served by the GPU: %llu\n", me_calls);
    if(df_calls || pad_calls) fprintf(stderr, "[xeve_hip_shim] pictures deblocked on the GPU: %llu, padded on the GPU: %llu\n", df_calls, pad_calls);
}

void xeve_platform_init_func(XEVE_CTX *ctx)
{
    void (*orig)(XEVE_CTX *) = (void (*)(XEVE_CTX *))dlsym(RTLD_NEXT, "xeve_platform_init_func");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim] reference xeve_platform_init_func not found\n"); abort(); }
    orig(ctx);
    /* XEVE_SHIM_ME_COMPLEXITY / XEVE_SHIM_ME_LEVEL: settings of the motion search the app has no option for (pi->me_complexity = param.me_algo: 2 adds
     * me_raster; pi->me_level = param.me_sub: 1 = integer refinement instead of the sub-pel pattern) -- applied to plain and GPU runs alike, so that the
     * other branches of pinter_me_epzs can be compared inside the encoder too */
    {
        const char *mc = getenv("XEVE_SHIM_ME_COMPLEXITY"), *ml = getenv("XEVE_SHIM_ME_LEVEL");
        for(int i = 0; i < ctx->param.threads && (mc || ml); i++) {
            if(mc) ctx->pinter[i].me_complexity = atoi(mc);
            if(ml) ctx->pinter[i].me_level = atoi(ml);
        }
    }
    if(getenv("XEVE_SHIM_SHADOW_TREE") && ctx->fn_mode_analyze_lcu && ctx->fn_mode_analyze_lcu != shim_mode_analyze_lcu) {
        void *oh = dlopen(getenv("XEVE_SHIM_SHADOW_TREE"), RTLD_NOW | RTLD_LOCAL);
        if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra"))) { fprintf(stderr, "[xeve_hip_shim] shadow tree: %s\n", dlerror()); abort(); }
        xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu");
        xo_eco = dlsym(oh, "xo_eco_ctu"), xo_tile_end = dlsym(oh, "xo_eco_tile_end");
        orig_mode_analyze_lcu = ctx->fn_mode_analyze_lcu, ctx->fn_mode_analyze_lcu = shim_mode_analyze_lcu;
        if(ctx->fn_loop_filter != shim_shadow_loop_filter) orig_shadow_loop_filter = ctx->fn_loop_filter, ctx->fn_loop_filter = shim_shadow_loop_filter;
        fprintf(stderr, "[xeve_hip_shim] shadow mode: the oracle walks and writes every CTU beside the reference\n");
        atexit(report);
    }
    if(getenv("XEVE_SHIM_TREE_ORACLE") && ctx->fn_mode_analyze_lcu && ctx->fn_mode_analyze_lcu != shim_route_mode_analyze_lcu) {
        void *oh = dlopen(getenv("XEVE_SHIM_TREE_ORACLE"), RTLD_NOW | RTLD_LOCAL);
        if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra"))) { fprintf(stderr, "[xeve_hip_shim] tree route (oracle engine): %s\n", dlerror()); abort(); }
        if(!getenv("XEVE_SHIM_TREE_I_ONLY")) xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu");
        orig_mode_analyze_lcu = ctx->fn_mode_analyze_lcu, ctx->fn_mode_analyze_lcu = shim_route_mode_analyze_lcu, tree_engine_oracle = 1;
        fprintf(stderr, "[xeve_hip_shim] CTU mode decision of I pictures served by the ORACLE through the route adapter (CPU test of the adapter)\n");
        atexit(report);
    }
    const char *lib = getenv("XEVE_HIP_LIB");
    if(!lib) return; /* plain reference run */
    void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if(!h) { fprintf(stderr, "[xeve_hip_shim] %s\n", dlerror()); abort(); }
    int (*init)(int)        = (int (*)(int))dlsym(h, "xeve_hip_init");
    int (*install)(void *)  = (int (*)(void *))dlsym(h, "xeve_hip_install_tables");
    const char *(*err)(void) = (const char *(*)(void))dlsym(h, "xeve_hip_last_error");
    hip_recon_blk            = (void (*)(s16 *, pel *, int, int, int, int, pel *, int))dlsym(h, "xeve_recon_blk_hip");
    hip_table_calls          = (unsigned long long (*)(void))dlsym(h, "xeve_hip_table_calls");
    const char *dev          = getenv("XEVE_HIP_DEVICE");
    if(init(dev ? atoi(dev) : 0) != 0) { fprintf(stderr, "[xeve_hip_shim] init: %s\n", err()); abort(); }
    /* XEVE_HIP_SHIM_TABLES=0: leave the per-call dispatch tables (and fn_recon) with the reference -- for runs that route the coarse entry points only (the
     * real-size encodes: the table layer's launch + sync per 128-byte block would dominate their wall time without adding coverage the table tests lack) */
    const int tables = !(getenv("XEVE_HIP_SHIM_TABLES") && atoi(getenv("XEVE_HIP_SHIM_TABLES")) == 0);
    int n = 0;
    if(tables) {
        n = install(&ctx->fn_itxb);
        if(n != 9) { fprintf(stderr, "[xeve_hip_shim] install: %d (%s)\n", n, err()); abort(); }
        ctx->fn_recon = shim_recon;
    }
    if(getenv("XEVE_HIP_SHIM_DF") && atoi(getenv("XEVE_HIP_SHIM_DF"))) {
        hip_deblock_host = dlsym(h, "xeve_hip_deblock_host"), hip_expand_host = dlsym(h, "xeve_hip_picbuf_expand_host"), hip_err = err;
        if(!hip_deblock_host || !hip_expand_host) { fprintf(stderr, "[xeve_hip_shim] deblock / expand entry points missing\n"); abort(); }
        ctx->fn_loop_filter = shim_loop_filter, ctx->fn_picbuf_expand = shim_pic_expand;
        fprintf(stderr, "[xeve_hip_shim] loop filter and picture padding routed to the GPU\n");
    }
    if(getenv("XEVE_HIP_SHIM_ECO") && atoi(getenv("XEVE_HIP_SHIM_ECO"))) {
        hip_eco_coef_host = dlsym(h, "xeve_hip_eco_coef_host"), hip_err = err;
        if(!hip_eco_coef_host) { fprintf(stderr, "[xeve_hip_shim] eco_coef entry point missing\n"); abort(); }
        orig_eco_coef = ctx->fn_eco_coef, ctx->fn_eco_coef = shim_eco_coef;
        fprintf(stderr, "[xeve_hip_shim] CABAC bit counting of the coefficient syntax routed to the GPU\n");
    }
    if(getenv("XEVE_HIP_SHIM_TQ") && atoi(getenv("XEVE_HIP_SHIM_TQ"))) {
        hip_tq_nnz_host = dlsym(h, "xeve_hip_tq_nnz_host"), hip_itdq_host = dlsym(h, "xeve_hip_itdq_host"), hip_err = err;
        if(!hip_tq_nnz_host || !hip_itdq_host) { fprintf(stderr, "[xeve_hip_shim] tq / itdq entry points missing\n"); abort(); }
        ctx->fn_tq = shim_tq, ctx->fn_itdp = shim_itdq;
        fprintf(stderr, "[xeve_hip_shim] transform + RDOQ and dequantisation + inverse transform routed to the GPU\n");
    }
    if(getenv("XEVE_HIP_SHIM_MC") && atoi(getenv("XEVE_HIP_SHIM_MC"))) {
        hip_mc_cu_host = dlsym(h, "xeve_hip_mc_cu_host"), hip_err = err;
        if(!hip_mc_cu_host) { fprintf(stderr, "[xeve_hip_shim] mc entry point missing\n"); abort(); }
        for(int i = 0; i < ctx->param.threads; i++) ctx->pinter[i].fn_mc = shim_mc;
        fprintf(stderr, "[xeve_hip_shim] CU motion compensation routed to the GPU\n");
    }
    if(getenv("XEVE_HIP_SHIM_INTER") && atoi(getenv("XEVE_HIP_SHIM_INTER")) && ctx->fn_pinter_analyze_cu) {
        hip_inter_host = dlsym(h, "xeve_hip_pinter_analyze_cu_host"), hip_err = err;
        if(!hip_inter_host) { fprintf(stderr, "[xeve_hip_shim] inter-analysis entry point missing\n"); abort(); }
        orig_pinter_analyze_cu = ctx->fn_pinter_analyze_cu, ctx->fn_pinter_analyze_cu = shim_pinter_analyze_cu;
        fprintf(stderr, "[xeve_hip_shim] whole inter analysis of a CU routed to the GPU\n");
        if(getenv("XEVE_HIP_SHIM_RESIDENT") && atoi(getenv("XEVE_HIP_SHIM_RESIDENT"))) {
            hip_picture_begin = dlsym(h, "xeve_hip_picture_begin"), hip_resident_stats = dlsym(h, "xeve_hip_resident_stats");
            if(!hip_picture_begin || !hip_resident_stats) { fprintf(stderr, "[xeve_hip_shim] resident-picture entry points missing\n"); abort(); }
            orig_analyze_frame = ctx->fn_mode_analyze_frame, ctx->fn_mode_analyze_frame = shim_analyze_frame;
            fprintf(stderr, "[xeve_hip_shim] pictures resident in HBM (one upload per plane and picture)\n");
        }
    }
    if(getenv("XEVE_HIP_SHIM_INTRA") && atoi(getenv("XEVE_HIP_SHIM_INTRA")) && ctx->fn_pintra_analyze_cu) {
        hip_intra_host = dlsym(h, "xeve_hip_pintra_analyze_cu_host"), hip_err = err;
        if(!hip_intra_host) { fprintf(stderr, "[xeve_hip_shim] intra-analysis entry point missing\n"); abort(); }
        orig_pintra_analyze_cu = ctx->fn_pintra_analyze_cu, ctx->fn_pintra_analyze_cu = shim_pintra_analyze_cu;
        fprintf(stderr, "[xeve_hip_shim] intra analysis of a CU routed to the GPU\n");
    }
    if(getenv("XEVE_HIP_SHIM_TREE") && atoi(getenv("XEVE_HIP_SHIM_TREE")) && ctx->fn_mode_analyze_lcu) {
        hip_tree_host = (hip_tree_host_fn)dlsym(h, "xeve_hip_mode_analyze_ctu_intra_host"), hip_err = err;
        if(!hip_tree_host) { fprintf(stderr, "[xeve_hip_shim] CTU tree-walk entry point missing\n"); abort(); }
        orig_mode_analyze_lcu = ctx->fn_mode_analyze_lcu, ctx->fn_mode_analyze_lcu = shim_route_mode_analyze_lcu;
        if(atoi(getenv("XEVE_HIP_SHIM_TREE")) > 1) { /* 2: P and B pictures too (needs resident pictures: one upload per plane and picture) */
            hip_tree_any_host = (hip_tree_any_host_fn)dlsym(h, "xeve_hip_mode_analyze_ctu_host");
            hip_picture_begin = dlsym(h, "xeve_hip_picture_begin"), hip_resident_stats = dlsym(h, "xeve_hip_resident_stats");
            if(!hip_tree_any_host || !hip_picture_begin || !hip_resident_stats) { fprintf(stderr, "[xeve_hip_shim] CTU tree-walk (P / B) entry points missing\n"); abort(); }
            if(ctx->fn_mode_analyze_frame != shim_analyze_frame) orig_analyze_frame = ctx->fn_mode_analyze_frame, ctx->fn_mode_analyze_frame = shim_analyze_frame;
        }
        fprintf(stderr, "[xeve_hip_shim] CTU mode decision of %s pictures routed to the GPU (one exchange per CTU)\n", hip_tree_any_host ? "I, P and B" : "I");
        if(getenv("XEVE_SHIM_TREE_CHECK")) {
            void *oh = dlopen(getenv("XEVE_SHIM_TREE_CHECK"), RTLD_NOW | RTLD_LOCAL);
            if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra")) || !(xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu"))) { fprintf(stderr, "[xeve_hip_shim] tree check: %s\n", dlerror()); abort(); }
            tree_check = 1;
        }
    }
    if(getenv("XEVE_HIP_SHIM_ME") && atoi(getenv("XEVE_HIP_SHIM_ME"))) {
        hip_me_epzs_host = dlsym(h, "xeve_hip_me_epzs_host"), hip_err = err;
        if(!hip_me_epzs_host) { fprintf(stderr, "[xeve_hip_shim] motion-search entry point missing\n"); abort(); }
        for(int i = 0; i < ctx->param.threads; i++) ctx->pinter[i].fn_me = shim_me;
        fprintf(stderr, "[xeve_hip_shim] motion search routed to the GPU\n");
    }
    atexit(report);
    if(tables) fprintf(stderr, "[xeve_hip_shim] HIP dispatch tables installed (%d pointers + fn_recon)\n", n);
    else fprintf(stderr, "[xeve_hip_shim] dispatch tables left with the reference (XEVE_HIP_SHIM_TABLES=0)\n");
}
