/*
 * oracle/xeve_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the arithmetic of XEVE's inter-prediction / RDO hot
 * path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (xeve_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the unmodified reference compiled in place (oracle/_ref/libxeveb_ref.so, see
 * oracle/Makefile) by tests/test_oracle_vs_ref.py, and against the committed
 * golden vectors in tests/golden/ (generated from that same reference build by
 * tests/golden/make_golden.py).
 *
 * Conventions (reference: src_base/xeve_port.h:54):
 *   pel = int16_t; all strides are in ELEMENTS; bit depth is the codec-internal
 *   depth (10 for every BASELINE config).
 */
#ifndef XEVE_ORACLE_H
#define XEVE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t xo_pel;

/* ---- block distortion (reference: src_base/xeve_sad.c) ------------------- */
/* a1: sad_16b            xeve_sad.c:40-61   */
int     xo_sad(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int bit_depth);
/* a2: ssd_16b            xeve_sad.c:275-297 */
int64_t xo_ssd(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int bit_depth);
/* a3: diff_16b           xeve_sad.c:160-178 */
void    xo_diff(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int st_diff, int16_t *diff);
/* a4: xeve_had           xeve_sad.c:394-1140 */
int     xo_satd(int w, int h, const xo_pel *org, const xo_pel *cur, int s_org, int s_cur, int bit_depth);

/* ---- motion compensation (reference: src_base/xeve_mc.c) ------------------ */
/* Baseline interpolation coefficient tables, xeve_mc.c:39-93 */
extern const int16_t xo_mc_l_coeff[16][8];
extern const int16_t xo_mc_c_coeff[32][4];
/* a5: xeve_mc_l_{00,n0,0n,nn}  xeve_mc.c:99-254.
 * frac_x / frac_y select the table entry exactly as the reference macro
 * xeve_mc_l does (xeve_mc.h:96-99): index [frac_x != 0][frac_y != 0]. */
void xo_mc_l(int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred,
             xo_pel *pred, int w, int h, int bit_depth, const int16_t (*coef)[8]);
/* a6: xeve_mc_c_{00,n0,0n,nn}  xeve_mc.c:259-381 */
void xo_mc_c(int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred,
             xo_pel *pred, int w, int h, int bit_depth, const int16_t (*coef)[4]);
/* a7: xeve_average_16b_no_clip xeve_mc.c:449-463 */
void xo_avg(const int16_t *src, const int16_t *ref, int16_t *dst, int s_src, int s_ref, int s_dst, int w, int h);

/* ---- Main profile, first slice (reference: src_main/xevem_mc.c, xevem_tq.c, xevem_itdq.c) ---- */
extern const int16_t xom_mc_l_coeff[16][8]; /* xevem_tbl_mc_l_coeff, xevem_mc.c:48-66  */
extern const int16_t xom_mc_c_coeff[32][4]; /* xevem_tbl_mc_c_coeff, xevem_mc.c:68-104 */
/* xevem_tbl_dmvr_mc_l (kind 0), xevem_tbl_dmvr_mc_c (kind 1), xevem_tbl_bl_mc_l (kind 2); entry [frac_x != 0][frac_y != 0]
 * (xevem_mc.c:167-463).  The Main 1-D transforms tx_pb / itx_pb are xo_tx / xo_itx with step == 2. */
void xo_mc_main(int kind, int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, xo_pel *pred, int w, int h,
                int bit_depth);

/* the rest of the Main dispatch list: inverse ATS (xeve_func_itrans), the affine gradient search's Sobel filters and normal equations */
/* xeve_tbl_intra_pred_ang[group][right] (xevem_ipred.c:456-815); le / up / ri point at element 0 of lines indexed -1 .. w + h - 1 */
void xo_ipred_ang(int group, int right, const xo_pel *le, const xo_pel *up, const xo_pel *ri, xo_pel *dst, int w, int h, int ipm, int bit_depth);
void xo_ats_matrix(int type, int log2n, int8_t *m);
void xo_itrans_ats(int type, int log2n, const int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2);
/* the forward passes, xeve_trans_map_tbl[type][log2 N - 1] (xevem_tq.c:53-56, 336-680): coef[j * line + i] = (sum_k M[j][k] block[i * N + k] + rnd) >> shift (no clip, 16-bit
 * store) for i < line - skip_line and j < N - skip_line_2 (the 4-point forms ignore skip_line_2), zero elsewhere */
void xo_trans_ats(int type, int log2n, const int16_t *block, int16_t *coef, int shift, int line, int skip_line, int skip_line_2);
void xo_sobel(int vertical, const xo_pel *pred, int s_pred, int32_t *der, int s_der, int w, int h);
void xo_equal_coeff(const xo_pel *residue, const int32_t *d0, const int32_t *d1, int s_der, int64_t (*eq)[7], int w, int h, int vertex_num);

/* ---- transforms (reference: src_base/xeve_tq.c, xeve_itdq.c, xeve_tbl.c) -- */
/* DCT-II integer matrix of size n x n (n = 2..64), xeve_tbl.c:83-236. */
void xo_dct_matrix(int n, int8_t *m /* n*n, row-major [k][x] */);
/* a9: tx_pb{2..64}b  xeve_tq.c:40-392   (1-D forward, transposing write) */
void xo_tx(int log2n, const void *src, void *dst, int shift, int line, int step);
/* a12: xeve_itx_pb{2..64}b xeve_itdq.c:34-430 (1-D inverse) */
void xo_itx(int log2n, const void *src, void *dst, int shift, int line, int step);
/* xeve_trans  xeve_tq.c:396-404  (2-D forward, in place on dense coef[w*h]) */
void xo_trans(int16_t *coef, int log2w, int log2h, int bit_depth);
/* xeve_itrans xeve_itdq.c:435-440 (2-D inverse, in place) */
void xo_itrans(int16_t *coef, int log2w, int log2h, int bit_depth);
/* a10: plain-quant branch of xeve_quant_nnz xeve_tq.c:704-727; returns nnz.
 * scale = xeve_quant_scale[tool_iqt][qp % 6] (xeve_tq.c:37-38). */
extern const int xo_quant_scale[2][6];
int  xo_quant(int16_t *coef, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth);
/* RDOQ all-zero pre-test of xeve_quant_nnz xeve_tq.c:666-699: returns 1 when
 * some coefficient may quantise to non-zero (block is "coded"), 0 otherwise. */
int  xo_rdoq_zero_test(const int16_t *coef, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth);
/* a13: xeve_dquant + shift/offset derivation of itdq_cu xeve_itdq.c:442-475.
 * scale = xeve_tbl_dq_scale_b[qp % 6] << (qp / 6) (xeve_tbl.c:237, xeve_itdq.c:549). */
extern const int xo_dq_scale[6];
void xo_dquant(int16_t *coef, int log2w, int log2h, int scale, int bit_depth);
/* a15: xeve_recon_blk xeve_recon.c:34-57 */
void xo_recon(const int16_t *coef, const xo_pel *pred, int is_coef, int cuw, int cuh, int s_rec, xo_pel *rec, int bit_depth);

/* ---- RDOQ (a11; reference: xeve_rdoq_run_length_cc, src_base/xeve_tq.c:497-649) -------------------------------- */
/* zig-zag scan of a w x h block, xeve_tbl_scan[log2w-1][log2h-1] (xeve_tbl.c:625; generator xeve_util.c:1289-1327) */
void xo_zigzag(int log2w, int log2h, uint16_t *scan);
/* ctx->err_scale[qp % 6][log2_size - 1]  (xeve_init_err_scale, xeve_tq.c:406-423) */
int64_t xo_err_scale(int qp_rem, int log2_size, int bit_depth, int tool_iqt);
/* CABAC-derived bit estimates the reference keeps in XEVE_CORE (xeve_type.h:737-747, filled by xeve_rdoq_bit_est,
 * xeve_mode.c:326-372); cbf = the pair the reference picks for this component / slice type (xeve_tq.c:565-583) */
typedef struct xo_rdoq_est {
    int32_t cbf[2];
    int32_t run[24][2], level[24][2], last[2][2];
} xo_rdoq_est;
/* in place on coef (dense w*h); returns nnz.  lambda as the reference receives it (double). */
int xo_rdoq(int16_t *coef, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt, const xo_rdoq_est *est);

/* ---- integer-pel motion search (reference: src_base/xeve_pinter.c) ------------ */
/* get_mv_bits (xeve_pinter.c:74-120) without the reference-index term; table xeve_tbl_mv_bits (xeve_tbl.c:286-496)
 * restated in closed form */
int xo_mv_bits(int mvd_x, int mvd_y);

typedef struct xo_me_params {
    uint32_t lambda_mv;        /* pi->lambda_mv                          (xeve_pinter.c:47,1763) */
    int32_t  refi_bits;        /* xeve_tbl_refi_bits[num_refp][refi]     (xeve_pinter.c:118)     */
    int32_t  extra_bits;       /* bi ? pi->mot_bits[other list] : 0      (xeve_pinter.c:429)     */
    int32_t  bi;               /* 0 = BI_NON, 1 = BI_NORMAL (11x11 grid, one round, SAD >> 1), 2 = BI_FL0/FL1-like (normal rounds, SAD >> 1) */
    int32_t  faststep;         /* MAX_FIRST_SEARCH_STEP 3 / MAX_REFINE_SEARCH_STEP 2 (xeve_pred.h:63-69) */
    int32_t  max_search_range; /* pi->max_search_range                   (xeve_pinter.c:535)     */
    int32_t  range_recentre;   /* the search range get_range_ipel derives for this reference picture (xeve_pinter.c:124-129) */
    int32_t  min_clip[2], max_clip[2]; /* pi->min_clip / max_clip        (xeve_pinter.c:132-136) */
    int32_t  reserved;         /* pinter_me_epzs only: bit 0 = raster search on (pi->me_complexity > 1), bits 8..15 = refi (MULTI_REF_ME_STEP) */
} xo_me_params;

typedef struct xo_me_job {
    int32_t x, y;       /* block position in the picture (integer pel) */
    int32_t org_off;    /* bi != 0: element offset of this job's dense org_bi block (stride = w); else unused */
    int16_t range[4];   /* min x, min y, max x, max y (absolute integer-pel positions, xeve_pinter.c:122-140) */
    int16_t gmvp[2];    /* MVP in picture coordinates, quarter pel */
    int16_t mvi[2];     /* initial MV in picture coordinates, quarter pel */
    int32_t beststep_in; /* *beststep on entry (the reference threads `tmpstep` through successive calls) */
} xo_me_job;

typedef struct xo_me_result {
    int16_t  mv[2];     /* best MV relative to the block, quarter-pel units (integer positions) */
    uint32_t cost;      /* cost_best */
    int32_t  beststep;  /* *beststep on exit */
    int32_t  best_mv_bits;
} xo_me_result;

/* me_ipel_diamond (xeve_pinter.c:363-551).  org0 / ref0 point at sample (0,0) of the picture inside its padded plane. */
void xo_me_ipel_diamond(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_me_job *job,
                        int log2w, int log2h, int bit_depth, const xo_me_params *p, xo_me_result *res);

/* me_spel_pattern (xeve_pinter.c:553-697): half-pel points around mvi, then (qpel_cnt > 0) quarter-pel points around
 * the half-pel winner; each candidate = xeve_mc_l + SAD + MV_COST.  mvi / result mv are relative to the block. */
typedef struct xo_spel_params {
    uint32_t lambda_mv;
    int32_t  refi_bits, extra_bits, bi, hpel_cnt, qpel_cnt;
} xo_spel_params;
typedef struct xo_spel_job {
    int32_t x, y, org_off;
    int16_t gmvp[2], mvi[2];
} xo_spel_job;
void xo_me_spel_pattern(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_spel_job *job,
                        int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_spel_params *p, xo_me_result *res);

/* me_raster (xeve_pinter.c:158-268): grid of step max(5, S / 2) * (refi + 1) over the range, then 3x3 grids of halving step around the best;
 * me_ipel_refinement (:270-361): the nine integer positions around mvi.  Both: cost, mv (relative, quarter pel), best_mv_bits in *res. */
void xo_me_raster(const xo_pel *org0, int s_org, const xo_pel *ref0, int s_ref, int x, int y, const int16_t range[4], const int16_t gmvp[2], int log2w, int log2h,
                  int bit_depth, const xo_me_params *p, int refi, xo_me_result *res);
void xo_me_ipel_refinement(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t range[4],
                           const int16_t gmvp[2], const int16_t mvi[2], int log2w, int log2h, int bit_depth, const xo_me_params *p, xo_me_result *res);
/* pinter_me_epzs (xeve_pinter.c:699-869); p->me.reserved bit 0 adds the raster search (me_complexity > 1), p->spel.hpel_cnt == 0 replaces the
 * sub-pel stage by me_ipel_refinement (me_level <= ME_LEV_IPEL).  With neither: first
 * diamond search from the MVP (faststep 3), refinement searches from the running best while beststep > 0 (faststep 2),
 * then me_spel_pattern.  mvp / mv are relative to the block (quarter pel); for bi == 1 `mv` is also the starting point. */
typedef struct xo_epzs_params {
    xo_me_params   me;   /* faststep is ignored (3 / 2 as in the reference) */
    xo_spel_params spel; /* lambda_mv, refi_bits, extra_bits, bi are taken from `me` */
} xo_epzs_params;
uint32_t xo_me_epzs(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t mvp[2],
                    int16_t mv[2], int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_epzs_params *p);
/* the same, also reporting what the searches leave in pi->mot_bits[lidx] (*mot_bits is only written when they change it) */
uint32_t xo_me_epzs_mot(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t mvp[2],
                        int16_t mv[2], int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_epzs_params *p, int *mot_bits);

/* ---- CABAC (SBAC) bit counting for the inter RDO (reference: src_base/xeve_eco.c, xeve_mode.c) -- */
/* The fields of XEVE_SBAC (xeve_type.h:527-540) plus the context models of XEVE_SBAC_CTX (xeve_def.h:736-790) that the
 * inter-CU syntax touches, in the order below (ctx[XO_CTX_x + i] = sbac->ctx.x[i]). */
enum {
    XO_CTX_SKIP_FLAG = 0,  /* [2]  */
    XO_CTX_PRED_MODE = 2,  /* [3]  */
    XO_CTX_DIRECT    = 5,  /* [1]  direct_mode_flag */
    XO_CTX_INTER_DIR = 6,  /* [2]  */
    XO_CTX_REFI      = 8,  /* [2]  */
    XO_CTX_MVP_IDX   = 10, /* [3]  */
    XO_CTX_MVD       = 13, /* [1]  */
    XO_CTX_CBF_ALL   = 14, XO_CTX_CBF_LUMA = 15, XO_CTX_CBF_CB = 16, XO_CTX_CBF_CR = 17,
    XO_CTX_RUN       = 18, /* [24] */
    XO_CTX_LAST      = 42, /* [2]  */
    XO_CTX_LEVEL     = 44, /* [24] */
    XO_CTX_INTRA_DIR = 68, /* [2]  intra_dir (xeve_eco_intra_dir, xeve_eco.c:1104-1121) */
    XO_CTX_SPLIT_CU  = 70, /* [1]  split_cu_flag */
    XO_CTX_DELTA_QP  = 71, /* [1]  delta_qp */
    XO_SBAC_NCTX     = 72
};
typedef struct xo_sbac {
    uint32_t range, code, code_bits, stacked_ff, stacked_zero, pending_byte, is_pending_byte, bitcounter, bin_counter;
    uint16_t ctx[XO_SBAC_NCTX];
} xo_sbac;
/* xeve_sbac_reset (xeve_eco.c:597-620): every model = PROB_INIT */
void     xo_sbac_reset(xo_sbac *s);
/* xeve_sbac_bit_reset / xeve_get_bit_number (xeve_mode.c:39-55) */
void     xo_sbac_bit_reset(xo_sbac *s);
uint32_t xo_sbac_bits(const xo_sbac *s);
/* xeve_sbac_encode_bin (xeve_eco.c:521-575) / sbac_encode_bin_ep (:455-472) in bit-count mode (is_bitcount = 1) */
void     xo_sbac_bin(xo_sbac *s, int ctx, uint32_t bin);
void     xo_sbac_bin_ep(xo_sbac *s, uint32_t bin);
/* xeve_eco_run_length_cc (xeve_eco.c:707-771); ch = 0 luma / 1 chroma; cm_init = sbac->ctx.sps_cm_init_flag */
void     xo_eco_run_length_cc(xo_sbac *s, const int16_t *coef, int log2w, int log2h, int num_sig, int ch, int cm_init);

/* xeve_rdoq_bit_est (xeve_mode.c:326-372): the bit estimates RDOQ reads, derived from the live coder state with the table of
 * xeve_init_bits_est (xeve_mode.c:304-313); the fields mirror core->rdoq_est_* (xeve_type.h:737-747) for the run-level syntax */
typedef struct xo_rdoq_est_full {
    int32_t cbf_all[2], cbf_luma[2], cbf_cb[2], cbf_cr[2];
    int32_t run[24][2], level[24][2], last[2][2];
} xo_rdoq_est_full;
int32_t  xo_entropy_bits(int i); /* entropy_bits[i], i = 0..1023 */
void     xo_rdoq_bit_est(const xo_sbac *s, xo_rdoq_est_full *e);
/* the cbf pair xeve_rdoq_run_length_cc picks (xeve_tq.c:565-583); ch_type 0 Y / 1 U / 2 V */
void     xo_rdoq_est_select(const xo_rdoq_est_full *f, int ch_type, int is_intra, xo_rdoq_est *e);

typedef struct xo_cu_bits_params {
    int32_t log2_cuw, log2_cuh;
    int32_t slice_type;          /* XEVE_ST_B 0 / XEVE_ST_P 1 / XEVE_ST_I 2 (inc/xeve.h:170-172)           */
    int32_t num_refp[2];         /* ctx->rpm.num_refp                                                     */
    int32_t cm_init;             /* sps_cm_init_flag (0 in Baseline)                                      */
    int32_t chroma_format_idc;   /* 0 = 4:0:0 ... 3 = 4:4:4; w/h shift as XEVE_GET_CHROMA_{W,H}_SHIFT     */
} xo_cu_bits_params;
enum { XO_BITS_CU_INTER = 0, XO_BITS_COMP_Y = 1, XO_BITS_COMP_U = 2, XO_BITS_COMP_V = 3, XO_BITS_CU_SKIP = 4, XO_BITS_ECO_COEF = 5,
       XO_BITS_MVP = 6 /* xeve_rdo_bit_cnt_mvp (xeve_mode.c:57-79): mvp_idx + mvd of every used list */,
       /* intra CU (Baseline): job.mvp_idx[0] holds the unary index mpm[ipm] of the luma mode (xeve_eco_intra_dir) */
       XO_BITS_CU_INTRA = 7 /* xeve_rdo_bit_cnt_cu_intra (xeve_mode.c:141-175) */, XO_BITS_INTRA_LUMA = 8 /* ..._intra_luma (:81-117) */,
       XO_BITS_INTRA_DIR = 9 /* xeve_rdo_bit_cnt_intra_dir (:136-139) */ };
/* XO_BITS_ECO_COEF: xeve_eco_coef (cbf flags + coefficients) on its own; job.dir_flag then holds these flags */
enum { XO_ECO_INTRA = 1, XO_ECO_NO_CBF = 2, XO_ECO_RUN_Y = 4, XO_ECO_RUN_U = 8, XO_ECO_RUN_V = 16, XO_ECO_NO_RESET = 32 /* continue the coder where the state stands */ };
typedef struct xo_cu_bits_job {
    int32_t coef_off[3];   /* element offsets of the dense Y / U / V coefficient blocks            */
    int32_t nnz[3];        /* core->nnz_sub[c][0] (0 = cbf 0, the block is not coded)              */
    int32_t sbac;          /* index of the entry state                                             */
    int16_t mvd[2][2];
    int8_t  refi[2];
    uint8_t mvp_idx[2];
    uint8_t mode;          /* XO_BITS_*                                                            */
    uint8_t dir_flag;      /* pidx == PRED_DIR (direct mode: no inter_pred_idc / refi / mvp / mvd) */
    uint8_t ctx_skip, ctx_pred_mode; /* core->ctx_flags[CNID_SKIP_FLAG], [CNID_PRED_MODE]          */
} xo_cu_bits_job;
/* SBAC_LOAD + xeve_sbac_bit_reset + { xeve_rdo_bit_cnt_cu_inter (xeve_mode.c:201-274) | xeve_rdo_bit_cnt_cu_inter_comp
 * (:177-199) | xeve_rdo_bit_cnt_cu_skip (:276-295) } + xeve_get_bit_number, as pinter_residue_rdo strings them together
 * (xeve_pinter.c:1112-1131 ...); Baseline tool set (tool_admvp 0, no delta QP, CU <= 64x64 so one transform block per
 * component).  *out = the state SBAC_STORE would keep. */
uint32_t xo_cu_bits(const xo_sbac *in, xo_sbac *out, const xo_cu_bits_params *p, const xo_cu_bits_job *job, const int16_t *coef);

/* ---- in-loop deblocking + reference-picture padding (reference: src_base/xeve_df.c, xeve_util.c) -- */
/* xeve_tbl_df_st (xeve_tbl.c:239-257) */
extern const uint8_t xo_df_st[4][52];
typedef struct xo_deblock_params {
    int32_t w, h;                    /* picture size in luma samples (pic->w_l, pic->h_l) */
    int32_t w_scu, h_scu;            /* ctx->w_scu, ctx->h_scu (4x4 units) */
    int32_t log2_max_cuwh;           /* CTU size (6 in Baseline) */
    int32_t bit_depth_luma, bit_depth_chroma, chroma_format_idc;
    int32_t qp_u_offset, qp_v_offset; /* pic->pic_qp_u_offset / pic_qp_v_offset (slice header) */
    int32_t qp_chroma[2][100];       /* ctx->qp_chroma_dynamic[c][q] at index q + 6 * (bit_depth_chroma - 8), q = -6 * (bd - 8) .. 57 */
} xo_deblock_params;
/* xeve_loop_filter (xeve_enc.c:2355-2415) = for both edge directions (vertical edges first): xeve_deblock (xeve_df.c:522-573)
 * walking every CTU's quad-tree (xeve_deblock_tree :575-639) into xeve_deblock_cu_ver / _cu_hor (:253-471); one tile, one slice.
 * y / u / v point at sample (0, 0) of the planes; map_scu (MCU_* bit fields, xeve_def.h:585-640), map_cu_mode (CU size in
 * bits 24-31, xeve_def.h:679-683), map_refi [f_scu][2], map_mv [f_scu][2][2] are per 4x4 unit.  map_scu's COD bits are
 * scratch, as in the reference. */
void xo_deblock_picture(xo_pel *y, xo_pel *u, xo_pel *v, int s_l, int s_c, uint32_t *map_scu, const uint32_t *map_cu_mode,
                        const int8_t *map_refi, const int16_t *map_mv, const xo_deblock_params *p);
void xo_deblock_picture_tiles(xo_pel *y, xo_pel *u, xo_pel *v, int s_l, int s_c, uint32_t *map_scu, const uint32_t *map_cu_mode, const uint8_t *map_tidx,
                              const int8_t *map_refi, const int16_t *map_mv, const xo_deblock_params *p);
/* xeve_picbuf_expand (xeve_util.c:190-248) on one plane: a = sample (0, 0) */
/* rdo_dbk_switch = 1 (preset slow): calc_delta_dist_filter_boundary (xeve_mode.c:1534-2005).  D = what it reads besides its arguments: the reconstruction so far
 * (PIC_MODE, unfiltered) and the unit maps; dp: the picture's deblock parameters (its qp offsets are NOT used: the scratch picture's are zero); qp: ctx->tile[].qp.
 * xo_rdo_dbk_begin / _end switch it on for the analyses of the calling thread (xo_mode_analyze_ctu and everything below it). */
typedef struct xo_dbk_ctx {
    const xo_pel   *mod[3];
    int32_t         s_mod_l, s_mod_c, qp, on;
    const uint32_t *map_scu;
    const int8_t   *map_refi;   /* [unit][list] */
    const int16_t  *map_mv;     /* [unit][list][x, y] */
    const uint8_t  *map_tidx;   /* or NULL */
    const xo_deblock_params *dp;
} xo_dbk_ctx;
void xo_rdo_dbk_begin(const xo_dbk_ctx *c);
void xo_rdo_dbk_end(void);
int  xo_rdo_dbk_on(void);
void xo_delta_dist(const xo_dbk_ctx *D, const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_pel *const src[3], int x, int y, int cuw, int cuh,
                   int intra_flag, int cbf_l, const int8_t refi[2], const int16_t mv[2][2], int64_t delta[3]);
void xo_picbuf_expand(xo_pel *a, int s, int w, int h, int exp);

/* ---- a8: the motion-compensation driver of one CU (reference: xeve_mc, src_base/xeve_mc.c:465-610, with xeve_mv_clip :401-447) -- */
typedef struct xo_refpic {
    const xo_pel *y, *u, *v; /* sample (0, 0) of the three planes of refp[refi][list].pic */
    int32_t       poc;       /* refp[refi][list].pic->poc (identical-motion test, xeve_mc.c:546-551) */
    int32_t       pad_;
} xo_refpic;
typedef struct xo_cu_mc_job {
    int32_t x, y;      /* CU position, luma samples */
    int16_t mv[2][2];  /* quarter pel, relative to the CU */
    int8_t  refi[2];   /* < 0: list unused */
    int8_t  pad_[2];
} xo_cu_mc_job;
/* refp[refi * 2 + list]; pred_* receive what the reference leaves in pred[0][Y_C / U_C / V_C] (dense, stride w resp. w >> w_shift) */
void xo_mc_cu(const xo_refpic *refp, int s_l, int s_c, int pic_w, int pic_h, const xo_cu_mc_job *job, int w, int h, int bit_depth_luma,
              int bit_depth_chroma, int chroma_format_idc, xo_pel *pred_y, xo_pel *pred_u, xo_pel *pred_v);

/* ---- the whole of pinter_residue_rdo (reference: src_base/xeve_pinter.c:906-1336) for one CU candidate ------------------------ */
/* prediction (xeve_mc), residual, SSD, transform + RDOQ with the estimates of the entry coder state, reconstruction, and the
 * rate-distortion decision about the coded-block flags: all-zero alternative, as quantised, each component with / without its
 * coefficients (coder state threaded from component to component), final combination.  Presets with rdo_dbk_switch = 0
 * (fast, medium), no delta QP, CU <= 64x64. */
typedef struct xo_rdo_params {
    int32_t log2_cuw, log2_cuh, pic_w, pic_h;
    int32_t slice_type, num_refp[2], chroma_format_idc, bit_depth, tool_iqt;
    int32_t qp[3];                 /* core->qp_y / qp_u / qp_v (already offset by the bit depth, xeve_def.h:52) */
    int32_t pad_;
    double  lambda[3];             /* core->lambda */
    double  dist_chroma_weight[2]; /* core->dist_chroma_weight */
} xo_rdo_params;
typedef struct xo_rdo_job {
    int32_t x, y;
    int16_t mv[2][2], mvd[2][2];   /* pi->mv[pidx], pi->mvd[pidx] */
    int8_t  refi[2];
    uint8_t mvp_idx[2];
    uint8_t dir_flag;              /* pidx == PRED_DIR: direct mode (no all-zero test, no motion syntax) */
    uint8_t ctx_skip, ctx_pred_mode, pad_;
    int32_t sbac;                  /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] in the state array */
} xo_rdo_job;
typedef struct xo_rdo_result {
    double   cost;                 /* the return value */
    int32_t  nnz[3];               /* core->nnz on exit */
    int32_t  pad_;
    int64_t  dist[2][3];           /* [0] no residual, [1] as quantised */
} xo_rdo_result;
/* org[c] / refp planes point at sample (0, 0); coef_y/u/v receive pi->coef[pidx] on exit (dense); best = core->s_temp_best */
void xo_residue_rdo(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                    const xo_rdo_params *p, const xo_rdo_job *job, xo_rdo_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                    xo_sbac *best);

/* ---- xeve_analyze_skip (reference: src_base/xeve_pinter.c:1337-1530): the skip / merge analysis of one CU -------------------------- */
/* every (idx0, idx1) pair of the two lists' merge candidates (duplicates of an earlier candidate pruned): prediction (xeve_mc), SSD of
 * Y / U / V against the original, cost = distortion + lambda * bits of the skip syntax; the first strictly smaller cost wins.  The candidate
 * lists are what xeve_get_motion (xeve_util.c:526-573) derived from the neighbouring CUs -- an input here.  Uses xo_rdo_params (log2_cuw/h,
 * pic_w/h, slice_type, num_refp, chroma_format_idc, bit_depth, lambda[0], dist_chroma_weight). */
typedef struct xo_skip_job {
    int32_t x, y;
    int16_t mvp[2][4][2];   /* pi->mvp[list][idx] */
    int8_t  refi_pred[2][4]; /* pi->refi_pred[list][idx] */
    int32_t ncand;          /* pi->skip_merge_cand_num (<= 4) */
    int32_t sbac;           /* index of the entry coder state */
    uint8_t ctx_skip, pad_[3];
} xo_skip_job;
typedef struct xo_skip_result {
    double  cost;           /* the return value (MAX_COST 1.7e308 when no candidate is usable) */
    int64_t best_ssd;       /* pi->best_ssd */
    int32_t idx0, idx1;     /* pi->mvp_idx[PRED_SKIP] */
    int16_t mv[2][2];       /* pi->mv[PRED_SKIP] */
    int8_t  refi[2];        /* pi->refi[PRED_SKIP] */
    int8_t  pad_[6];
} xo_skip_result;
/* pred_* receive pi->pred[PRED_SKIP][0] (untouched when no candidate is usable); best = core->s_temp_best (ditto) */
void xo_analyze_skip(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                     const xo_rdo_params *p, const xo_skip_job *job, xo_skip_result *res, xo_pel *pred_y, xo_pel *pred_u, xo_pel *pred_v, xo_sbac *best);


/* ---- the whole inter analysis of one CU: xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047) = ctx->fn_pinter_analyze_cu ----
 * Baseline (tool_admvp 0): skip / merge analysis; unless the skip residual is below the skip_th threshold: temporal direct (B), per list the
 * motion search over every reference picture (pinter_me_epzs) + check_best_mvp + pinter_residue_rdo, the iterated bi-prediction search (B)
 * + pinter_residue_rdo; the cheapest mode's coefficients, reconstruction, motion data and coder state. */
#define XO_MAX_REFP 8
typedef struct xo_inter_params {
    xo_rdo_params  rdo;
    xo_epzs_params me;                             /* lambda_mv, max_search_range, clips, hpel / qpel counts; the four below are set per search */
    int32_t        refi_bits[2][XO_MAX_REFP];      /* xeve_tbl_refi_bits[num_refp[l]][refi] */
    int32_t        range_recentre[2][XO_MAX_REFP]; /* get_range_ipel's POC-distance scaled range of refp[refi][l] */
    int32_t        max_cand;                       /* pi->skip_merge_cand_num */
    int32_t        poc, col_list_poc0;             /* ctx->poc.poc_val; refp[0][REFP_1].list_poc[0] (temporal direct) */
    int32_t        pad_;
    double         skip_th;                        /* ctx->param.skip_th */
} xo_inter_params;
typedef struct xo_inter_job {
    int32_t x, y;
    int16_t mvp[2][4][2]; /* xeve_get_motion's candidates per list (left, up, up-right, collocated); reference index 0 each */
    int16_t mv_col[2];    /* refp[0][REFP_1].map_mv[bottom-right unit of the CU][0] (xeve_get_mv_dir) */
    int32_t sbac;
    uint8_t ctx_skip, ctx_pred_mode, pad_[2];
} xo_inter_job;
typedef struct xo_inter_result {
    double  cost;          /* the return value: cost_inter[best_idx] */
    double  cost_inter[5]; /* PRED_L0, PRED_L1, PRED_BI, PRED_SKIP, PRED_DIR (1.7e308 where not evaluated; locals of the reference function) */
    int32_t cu_mode;       /* MODE_INTER 1 / MODE_SKIP 2 / MODE_DIR 3 */
    int32_t best_idx;
    int16_t mv[2][2], mvd[2][2]; /* mi->mv, mi->mvd; entries of an unused list are 0 here (stale in the reference) */
    int8_t  refi[2];
    uint8_t mvp_idx[2];
    int32_t nnz[3];
    int32_t pad_[2];
} xo_inter_result;
/* coef_*: the `coef` argument (zero for MODE_SKIP, where the reference leaves stale data behind nnz = 0); rec_*: pi->rec[best_idx] (dense);
 * next_best: core->s_next_best[log2_cuw - 2][log2_cuh - 2] */
void xo_pinter_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                          const xo_inter_params *P, const xo_inter_job *job, xo_inter_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                          xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *next_best);
/* The candidates an xo_inter_job carries, from the maps the encoder keeps per 4x4 unit: xeve_get_avail_inter (xeve_util.c:652-714; only the
 * left / up / up-right bits matter here) + xeve_get_motion (xeve_util.c:526-573) per list + the collocated vector xeve_get_mv_dir reads
 * (xeve_util.c:631-632, unit = the CU's bottom-right one).  map_mv / col0 / col1: [unit][list][x, y]; col0 / col1 = refp[0][REFP_0 / REFP_1].map_mv.
 * job->x / y are read; job->mvp / mv_col are written (list 1 and mv_col stay 0 in P slices). */
void xo_inter_candidates(const uint32_t *map_scu, const uint8_t *map_tidx, const int16_t (*map_mv)[2][2], const int16_t (*col0)[2][2],
                         const int16_t (*col1)[2][2], int w_scu, int h_scu, int log2_cuw, int log2_cuh, int slice_type, xo_inter_job *job);
/* check_best_mvp (xeve_pinter.c:1773-1837): returns the chosen index; mvd is recomputed against it */
int xo_check_best_mvp(const xo_sbac *entry, int slice_type, const int8_t refi[2], int lidx, const int16_t mvp[4][2], const int16_t mv[2], int mvp_idx,
                      double lambda0, int16_t mvd[2]);

/* ---- the intra analysis of one CU: pintra_analyze_cu (src_base/xeve_pintra.c:544-698) = ctx->fn_pintra_analyze_cu, Baseline ---- */
extern const uint8_t xo_tbl_mpm[6][6][5];
void xo_get_nbr(int x, int y, int cuw, int cuh, const xo_pel *src, int s_src, const uint32_t *map_scu, const uint8_t *map_tidx, int w_scu, int h_scu, int ch,
                int constrained_intra_pred, int bit_depth, int chroma_format_idc, xo_pel *left, xo_pel *up);
void xo_ipred(const xo_pel *left, const xo_pel *up, xo_pel *dst, int ipm, int w, int h);
const uint8_t *xo_get_mpm(int x_scu, int y_scu, const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, int w_scu);
typedef struct xo_intra_params {
    int32_t log2_cuw, log2_cuh, w_scu, h_scu;   /* picture size in 4x4 units (ctx->w_scu, ctx->h_scu) */
    int32_t slice_type, chroma_format_idc, bit_depth, tool_iqt;
    int32_t constrained_intra_pred, qp[3];      /* pps.constrained_intra_pred_flag; core->qp_y / qp_u / qp_v */
    double  lambda[3];                          /* core->lambda */
    double  sqrt_lambda0;                       /* core->sqrt_lambda[0] */
    double  dist_chroma_weight[2];
} xo_intra_params;
typedef struct xo_intra_job {
    int32_t  x, y;
    uint32_t inter_satd;   /* core->inter_satd: SATD of the best inter prediction, 0xFFFFFFFF when there is none (mode_check_intra, xeve_mode.c:1250-1262) */
    int32_t  sbac;         /* index of core->s_curr_best[log2_cuw - 2][log2_cuh - 2] */
    int32_t  pic;          /* which picture of a multi-picture batch the CU belongs to (the HIP batched form; not read here: the caller passes that picture) */
    uint8_t  ctx_skip, ctx_pred_mode, pad_[2];
} xo_intra_job;
typedef struct xo_intra_result {
    double  cost;          /* the return value */
    int32_t dist_cu;       /* core->dist_cu */
    int32_t nnz[3];        /* core->nnz */
    int32_t pred_cnt;      /* candidates that went through the luma RDO (local of the reference function) */
    int8_t  ipm[2];        /* core->ipm */
    int8_t  pad_[2];
} xo_intra_result;
/* org: original planes; mod: the planes of the picture being reconstructed (pi->m = PIC_MODE(ctx)), whose samples left of and above the CU are the
 * predictors' input; map_scu / map_ipm / map_tidx: ctx->map_scu, ctx->map_ipm, ctx->map_tidx.  coef_*: the `coef` argument; rec_*: pi->rec (dense);
 * best: core->s_temp_best. */
void xo_pintra_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_pel *const mod[3], int s_mod_l, int s_mod_c, const uint32_t *map_scu,
                          const int8_t *map_ipm, const uint8_t *map_tidx, const xo_sbac *states, const xo_intra_params *p, const xo_intra_job *job,
                          xo_intra_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *best);

/* ---- the mode decision of one CTU of an I picture: mode_analyze_lcu -> mode_coding_tree (src_base/xeve_mode.c:2007-2375, 2518-2610), Baseline ---- */
#define XO_CU_DEPTHS 10 /* cud = 0, 2, 4, 6, 8 for 64 .. 4 (a quad split counts as two levels): the rows of XEVE_CU_DATA.split_mode[][SQUARE][] the walk can touch */
typedef struct xo_tree_params {
    xo_intra_params ip;     /* what every CU's intra analysis gets (log2_cuw / log2_cuh are set per node) */
    int32_t pic_w, pic_h;   /* ctx->w, ctx->h */
    int32_t log2_ctu;       /* ctx->log2_max_cuwh */
    int32_t max_cu, min_cu; /* ctx->param.max_cu_intra, min_cu_intra (samples) */
    int32_t min_cuwh;       /* ctx->min_cuwh */
    int32_t slice_qp, slice_num; /* ctx->tile[].qp (the QP field of map_scu), ctx->slice_num */
    int32_t rdo_dbk; /* rdo_dbk_switch: the callers of xo_mode_analyze_ctu switch it on with xo_rdo_dbk_begin (the context it needs is theirs) */
} xo_tree_params;
typedef struct xo_ctu_data { /* the fields of XEVE_CU_DATA (xeve_type.h:573-617) an I-slice CTU carries; 4x4 units in raster order, pitch = the block's width in units */
    int8_t   split_mode[XO_CU_DEPTHS][256]; /* [depth][unit]: shape SQUARE */
    uint8_t  pred_mode[256];
    int8_t   ipm[2][256], depth[256];
    int32_t  nnz[3][256];
    uint32_t map_scu[256], map_cu_mode[256];
    int16_t  coef[3][64 * 64];
    xo_pel   reco[3][64 * 64];
    int16_t  mv[256][2][2], mvd[256][2][2]; /* P / B slices: [unit][list][x, y]; zero for an intra unit and for a list the CU does not use */
    int8_t   refi[256][2];
    uint8_t  mvp_idx[256][2];
} xo_ctu_data;
/* the inter side of the walk (P / B slices) */
typedef struct xo_tree_inter {
    const xo_refpic *refp;          /* [refi * 2 + list], as for xo_pinter_analyze_cu */
    int32_t          s_ref_l, s_ref_c;
    xo_inter_params  ipar;          /* rdo.log2_cuw / log2_cuh are set per node */
    int16_t        (*map_mv)[2][2]; /* ctx->map_mv: read for the candidates, updated with every decided CU */
    int8_t         (*map_refi)[2];  /* ctx->map_refi: updated */
    const int16_t  (*col0)[2][2], (*col1)[2][2]; /* refp[0][REFP_0 / REFP_1].map_mv */
    int32_t          ecu_depth;     /* the depth from which a skipped CU ends the split: ENC_ECU_DEPTH_B 4, minus 2 on odd POCs (xeve_mode.c:2162-2172) */
    int32_t          pad_;
} xo_tree_inter;
/* mod (the picture being reconstructed), map_scu, map_ipm, map_cu_mode are read AND updated as the reference's walk updates them; on return they hold the CTU's
 * decision (as after update_to_ctx_map + update_map_scu; the caller's reset of the coded flags, xeve_mode.c:2591-2607, is not applied).  Returns the CTU's cost. */
/* every slice type: P->ip.slice_type 0 B / 1 P with `inter` (max_cu / min_cu = ctx->param.max_cu_inter / min_cu_inter), 2 I as below (inter ignored) */
double xo_mode_analyze_ctu(const xo_pel *const org[3], int s_org_l, int s_org_c, xo_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                           const uint8_t *map_tidx, uint32_t *map_cu_mode, const xo_sbac *entry, const xo_tree_params *P, const xo_tree_inter *inter, int x0, int y0,
                           xo_ctu_data *out, xo_sbac *next_best);
/* The bitstream writer's side of a decided CTU: xeve_eco_tree (xeve_enc.c:35-100) with the writer's coder `s` (continued, never reset inside a tile): split flags and the
 * syntax of every CU as xeve_eco_unit writes it (xeve_eco.c:1431-1640; not the rate estimate's syntax: no direct_mode_flag / inter_pred_idc in P slices).  map_scu /
 * map_cu_mode (the CTU's units: coded flags cleared on entry) receive what xeve_eco_unit stores.  Returns the number of bytes the coder emitted; the first bytes_cap of
 * them are stored (bytes may be NULL). */
int xo_eco_ctu(xo_sbac *s, const xo_ctu_data *d, const xo_tree_params *P, const int num_refp[2], uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx,
               uint32_t *map_cu_mode, int x0, int y0, uint8_t *bytes, int bytes_cap);
/* the end of a tile: the terminating bin (1) and xeve_sbac_finish (xeve_eco.c:577-595, 622-672); returns the bytes that come out */
int xo_eco_tile_end(xo_sbac *s, uint8_t *bytes, int cap);
double xo_mode_analyze_ctu_intra(const xo_pel *const org[3], int s_org_l, int s_org_c, xo_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu,
                                 int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xo_sbac *entry, const xo_tree_params *P, int x0, int y0,
                                 xo_ctu_data *out, xo_sbac *next_best);

/* ---- Main profile: the adaptive loop filter's sample kernels (reference: src_main/xevem_alf.c) ------------------------------------------------------------------ */
typedef struct xo_alf_area { int32_t x, y, w, h; } xo_alf_area; /* AREA (xevem_alf.h:65-71) */
/* alf_copy_and_extend / alf_copy_and_extend_tile (:91-168): rec -> tmp (both at sample (0, 0) of the w x h area), then m samples of edge replication on every side */
void xo_alf_copy_and_extend(xo_pel *tmp, int s_tmp, const xo_pel *rec, int s_rec, int w, int h, int m);
/* alf_derive_classification (:463-486) = alf_derive_classification_blk (:488-654) over 32x32 pieces of the area: classifier[(y + i) * s_cls + x + j] of every sample =
 * (class << 2) | transpose index of its 4x4 block.  src at sample (0, 0) of the picture the area's coordinates count in; reads 3 samples around the area. */
void xo_alf_classify(uint8_t *classifier, int s_cls, const xo_pel *src, int s_src, const xo_alf_area *blk, int bit_depth);
/* alf_filter_blk_7 (:656-787) / alf_filter_blk_5 (:789-882): dst / src at the area's first sample; the classifier is indexed with the area's own coordinates (7-tap
 * form); filter_set: 25 x 13 coefficients (7-tap) or 7 (5-tap) */
void xo_alf_filter7(const uint8_t *classifier, int s_cls, xo_pel *dst, int s_dst, const xo_pel *src, int s_src, const xo_alf_area *blk, const int16_t *filter_set,
                    int clip_min, int clip_max);
void xo_alf_filter5(xo_pel *dst, int s_dst, const xo_pel *src, int s_src, const xo_alf_area *blk, const int16_t *filter_set, int clip_min, int clip_max);
/* xeve_alf_get_blk_stats (:3836-3888) + xeve_alf_clac_covariance (:3890-3952): per class the auto-correlation E[13][13] (the leading ncoef x ncoef part, symmetric), the
 * cross-correlation y[13] and the energy of (org - rec) over the w x h samples at (x, y).  taps 5 | 7 (ncoef 7 | 13).  classifier NULL (chroma): class 0, no
 * transposition.  E / y / pix are ADDED to (the reference accumulates into the CTU's record), nclasses = 25 with a classifier, else 1. */
void xo_alf_blk_stats(int taps, const uint8_t *classifier, int s_cls, const xo_pel *org, int s_org, const xo_pel *rec, int s_rec, int x, int y, int w, int h,
                      double *E /* [nclasses][13][13] */, double *yv /* [nclasses][13] */, double *pix /* [nclasses] */);

/* ---- Main profile: affine motion compensation of one CU (reference: xeve_affine_mc, src_main/xevem_mc.c:2236-2339) ---------------------------------------------------- */
typedef struct xo_affine_job {
    int32_t x, y;         /* CU position, luma samples */
    int16_t mv[2][3][2];  /* per list the control-point vectors (top-left, top-right, bottom-left), quarter pel */
    int8_t  refi[2];      /* < 0: list unused */
    int8_t  vertex_num;   /* 2 (four-parameter model: the third vector is not read) | 3 */
    int8_t  pad_;
} xo_affine_job;
/* derive_affine_subblock_size_bi (xevem_util.c:1203-1272, with check_eif_applicability_bi :1451-1480), then per used list xeve_affine_mc_lc (:1671-1915: blocks of
 * sub_w x sub_h through the Main 8- / 4-tap filters at the block centre's vector, or -- sub-blocks below 8 -- the enhanced interpolation filter, xeve_eif_mc :2123-2234:
 * a bilinear sample per position at the position's own vector, then the 3-tap {-1, 10, -1} filter in both directions), then the average of two lists.  refp[refi * 2 + list];
 * pred_* dense (w, w / 2); 4:2:0.  path (may be NULL): sub_w, sub_h, the memory-bandwidth condition. */
void xo_affine_mc(const xo_refpic *refp, int s_l, int s_c, int pic_w, int pic_h, const xo_affine_job *job, int w, int h, int bit_depth, xo_pel *pred_y, xo_pel *pred_u,
                  xo_pel *pred_v, int *path);

/* ---- Main profile: the affine gradient search of one CU on one reference picture (reference: pinter_affine_me_gradient, src_main/xevem_pinter.c:4290-4501, with solve_equal
 * :4213-4255, get_affine_mv_bits :4257-4288 and the luma-only compensation xeve_affine_mc_l, xevem_mc.c:1532-1669) -------------------------------------------------------- */
typedef struct xo_affine_me_job {
    int32_t  x, y;           /* CU position, luma samples */
    int16_t  mvp[3][2];      /* the predictor's control points (bits of the difference) */
    int16_t  mv[3][2];       /* in: the start vectors; out: the best ones found */
    int8_t   refi, list;     /* the reference picture: refp[refi * 2 + list] */
    int8_t   bi;             /* 1: the original is the job's org_bi block (2 * org - the other list's prediction), SATD >> 1, + mot_bits_other */
    int8_t   vertex_num;     /* 2 | 3 */
    int32_t  mot_bits_other; /* pi->mot_bits[1 - list] */
    uint32_t cost;           /* out: cost_best - MV_COST(best_bits) */
} xo_affine_me_job;
/* org: the picture's original luma plane (sample (0, 0)), pitch s_org -- or, job->bi, the job's dense w x h block of 16-bit values (pitch w).  Per round: error = org - pred,
 * Sobel derivatives of the prediction, the normal equations (64-bit sums), solve_equal in double, the control points moved by the rounded solution, compensation + SATD + vector
 * bits; 7 / 5 rounds (uni / bi), two fewer with three control points; stops when the update is zero. */
void xo_affine_me_gradient(const xo_refpic *refp, int s_l, int pic_w, int pic_h, const int16_t *org, int s_org, xo_affine_me_job *job, int w, int h, int bit_depth,
                           uint32_t lambda_mv, int num_refp);
/* the luma prediction alone (xeve_affine_mc_l): pred dense w x h */
void xo_affine_mc_l(const xo_pel *ref_y, int s_l, int pic_w, int pic_h, int x, int y, const int16_t mv[3][2], int vertex_num, int w, int h, int bit_depth, xo_pel *pred);
/* solve_equal (xevem_pinter.c:4213-4255): Gaussian elimination with row pivoting on eq[1 .. order][0 .. order], row 0 as scratch */
void xo_affine_solve(double (*eq)[7], int order, double *para);

#ifdef __cplusplus
}
#endif
#endif
