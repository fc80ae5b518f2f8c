/*
 * oracle/ref_me_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_me.so).
 *
 * me_ipel_diamond is `static` in the reference (src_base/xeve_pinter.c:363).  To pin the oracle's restatement of
 * it against the REAL function, this driver compiles the reference's xeve_pinter.c, unmodified and where it lies,
 * into its own translation unit (#include of the .c file) and exports a flat C wrapper around the static function.
 * All other symbols (tables, xeve_func_sad, ...) come from oracle/_ref/libxeveb_ref.so.
 */
#include "xeve_pinter.c"

typedef struct { unsigned lambda_mv; int refi_bits_unused, extra_bits, bi, faststep, max_search_range, range_recentre_unused;
                 int min_clip[2], max_clip[2], beststep_in; } drv_params;

/* returns cost_best; out[0..1] = mv, out[2] = beststep, out[3] = mot_bits[lidx] after the call */
unsigned refdrv_me_ipel_diamond(pel *org0, int s_org, const s16 *org_bi, pel *ref0, int s_ref, int x, int y, int log2w, int log2h,
                                int bit_depth, const s16 range_in[4], const s16 gmvp_in[2], const s16 mvi_in[2], int bi, int faststep,
                                unsigned lambda_mv, int num_refp, int refi, int mot_bits_other, int max_search_range, int gop_size, int poc,
                                int ref_poc, const int min_clip[2], const int max_clip[2], int beststep_in, int out[4])
{
    static XEVE_PINTER *pi;
    static XEVE_PIC     pic;
    static XEVE_REFP    refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    if(!pi) pi = calloc(1, sizeof(*pi));
    xeve_func_sad = xeve_tbl_sad_16b; /* plain-C table of the reference */
    const int lidx = REFP_0, lidx_r = REFP_1;
    pi->o[Y_C] = org0, pi->s_o[Y_C] = s_org;
    pic.y = ref0, pic.s_l = s_ref;
    refp[refi][lidx].pic = &pic, refp[refi][lidx].poc = ref_poc;
    pi->refp = refp;
    if(org_bi) memcpy(pi->org_bi, org_bi, sizeof(s16) << (log2w + log2h));
    pi->min_clip[MV_X] = min_clip[0], pi->min_clip[MV_Y] = min_clip[1];
    pi->max_clip[MV_X] = max_clip[0], pi->max_clip[MV_Y] = max_clip[1];
    pi->num_refp = num_refp, pi->lambda_mv = lambda_mv;
    pi->mot_bits[lidx_r] = mot_bits_other, pi->mot_bits[lidx] = -1;
    pi->max_search_range = max_search_range, pi->gop_size = gop_size, pi->poc = poc;
    s16 range[MV_RANGE_DIM][MV_D] = {{range_in[0], range_in[1]}, {range_in[2], range_in[3]}};
    s16 gmvp[MV_D] = {gmvp_in[0], gmvp_in[1]}, mvi[MV_D] = {mvi_in[0], mvi_in[1]}, mv[MV_D];
    int beststep = beststep_in;
    unsigned cost = me_ipel_diamond(pi, x, y, log2w, log2h, (s8)refi, lidx, range, gmvp, mvi, mv, bi, &beststep, faststep, bit_depth);
    out[0] = mv[MV_X], out[1] = mv[MV_Y], out[2] = beststep, out[3] = pi->mot_bits[lidx];
    return cost;
}

int refdrv_mv_bits(int mvd_x, int mvd_y, int num_refp, int refi) { return get_mv_bits(mvd_x, mvd_y, num_refp, refi); }
int refdrv_refi_bits(int num_refp, int refi) { return xeve_tbl_refi_bits[num_refp][refi]; }


/* me_spel_pattern (static, xeve_pinter.c:553): returns cost; out[0..1] = mv, out[2] = mot_bits[lidx] after the call */
unsigned refdrv_me_spel_pattern(pel *org0, int s_org, const s16 *org_bi, pel *ref0, int s_ref, int x, int y, int log2w, int log2h,
                                int bit_depth, const s16 gmvp_in[2], const s16 mvi_in[2], int bi, unsigned lambda_mv, int num_refp, int refi,
                                int mot_bits_other, int hpel_cnt, int qpel_cnt, int out[3])
{
    static XEVE_PINTER *pi;
    static XEVE_PIC     pic;
    static XEVE_REFP    refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    if(!pi) pi = calloc(1, sizeof(*pi));
    xeve_func_sad  = xeve_tbl_sad_16b;
    xeve_func_mc_l = xeve_tbl_mc_l;
    const int lidx = REFP_0, lidx_r = REFP_1;
    pi->o[Y_C] = org0, pi->s_o[Y_C] = s_org;
    pic.y = ref0, pic.s_l = s_ref;
    refp[refi][lidx].pic = &pic;
    pi->refp = refp;
    if(org_bi) memcpy(pi->org_bi, org_bi, sizeof(s16) << (log2w + log2h));
    pi->num_refp = num_refp, pi->lambda_mv = lambda_mv;
    pi->mot_bits[lidx_r] = mot_bits_other, pi->mot_bits[lidx] = -1;
    pi->search_pattern_hpel = tbl_search_pattern_hpel_partial, pi->search_pattern_hpel_cnt = hpel_cnt;
    pi->search_pattern_qpel = tbl_search_pattern_qpel_8point, pi->search_pattern_qpel_cnt = qpel_cnt;
    pi->me_level = hpel_cnt == 0 ? ME_LEV_IPEL : (qpel_cnt > 0 ? ME_LEV_QPEL : ME_LEV_HPEL);
    pi->mc_l_coeff = xeve_tbl_mc_l_coeff;
    s16 gmvp[MV_D] = {gmvp_in[0], gmvp_in[1]}, mvi[MV_D] = {mvi_in[0], mvi_in[1]}, mv[MV_D];
    unsigned cost = me_spel_pattern(pi, x, y, log2w, log2h, (s8)refi, lidx, gmvp, mvi, mv, bi, bit_depth);
    out[0] = mv[MV_X], out[1] = mv[MV_Y], out[2] = pi->mot_bits[lidx];
    return cost;
}


static int g_last_mot_bits;
unsigned refdrv_me_epzs_x(pel *org0, int s_org, const s16 *org_bi, pel *ref0, int s_ref, int x, int y, int log2w, int log2h, int bit_depth,
                          const s16 mvp_in[2], s16 mv_io[2], int bi, unsigned lambda_mv, int num_refp, int refi_in, int mot_bits_other,
                          int max_search_range, int gop_size, int poc, int ref_poc, const int min_clip[2], const int max_clip[2], int hpel_cnt,
                          int qpel_cnt, int me_complexity);
/* pinter_me_epzs (static, xeve_pinter.c:699) with me_complexity 1 (no raster); mv_io: in = start for bi == BI_NORMAL, out = result */
unsigned refdrv_me_epzs(pel *org0, int s_org, const s16 *org_bi, pel *ref0, int s_ref, int x, int y, int log2w, int log2h, int bit_depth,
                        const s16 mvp_in[2], s16 mv_io[2], int bi, unsigned lambda_mv, int num_refp, int refi_in, int mot_bits_other,
                        int max_search_range, int gop_size, int poc, int ref_poc, const int min_clip[2], const int max_clip[2], int hpel_cnt,
                        int qpel_cnt)
{
    return refdrv_me_epzs_x(org0, s_org, org_bi, ref0, s_ref, x, y, log2w, log2h, bit_depth, mvp_in, mv_io, bi, lambda_mv, num_refp, refi_in, mot_bits_other,
                            max_search_range, gop_size, poc, ref_poc, min_clip, max_clip, hpel_cnt, qpel_cnt, 1);
}

/* the same with me_complexity (pi->me_complexity = param.me_algo: 2 adds me_raster) as an argument; hpel_cnt == 0 selects me_level = ME_LEV_IPEL */
unsigned refdrv_me_epzs_x(pel *org0, int s_org, const s16 *org_bi, pel *ref0, int s_ref, int x, int y, int log2w, int log2h, int bit_depth,
                          const s16 mvp_in[2], s16 mv_io[2], int bi, unsigned lambda_mv, int num_refp, int refi_in, int mot_bits_other,
                          int max_search_range, int gop_size, int poc, int ref_poc, const int min_clip[2], const int max_clip[2], int hpel_cnt,
                          int qpel_cnt, int me_complexity)
{
    static XEVE_PINTER *pi;
    static XEVE_PIC     pic;
    static XEVE_REFP    refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    if(!pi) pi = calloc(1, sizeof(*pi));
    xeve_func_sad  = xeve_tbl_sad_16b;
    xeve_func_mc_l = xeve_tbl_mc_l;
    const int lidx = REFP_0, lidx_r = REFP_1;
    pi->o[Y_C] = org0, pi->s_o[Y_C] = s_org;
    pic.y = ref0, pic.s_l = s_ref;
    refp[refi_in][lidx].pic = &pic, refp[refi_in][lidx].poc = ref_poc;
    pi->refp = refp;
    if(org_bi) memcpy(pi->org_bi, org_bi, sizeof(s16) << (log2w + log2h));
    pi->min_clip[MV_X] = min_clip[0], pi->min_clip[MV_Y] = min_clip[1];
    pi->max_clip[MV_X] = max_clip[0], pi->max_clip[MV_Y] = max_clip[1];
    pi->num_refp = num_refp, pi->lambda_mv = lambda_mv;
    pi->mot_bits[lidx_r] = mot_bits_other, pi->mot_bits[lidx] = -1;
    pi->max_search_range = max_search_range, pi->gop_size = gop_size, pi->poc = poc;
    pi->me_complexity = me_complexity;
    pi->search_pattern_hpel = tbl_search_pattern_hpel_partial, pi->search_pattern_hpel_cnt = hpel_cnt;
    pi->search_pattern_qpel = tbl_search_pattern_qpel_8point, pi->search_pattern_qpel_cnt = qpel_cnt;
    pi->me_level = hpel_cnt == 0 ? ME_LEV_IPEL : (qpel_cnt > 0 ? ME_LEV_QPEL : ME_LEV_HPEL);
    pi->mc_l_coeff = xeve_tbl_mc_l_coeff;
    s16 mvp[MV_D] = {mvp_in[0], mvp_in[1]}, mv[MV_D] = {mv_io[0], mv_io[1]};
    s8  refi = (s8)refi_in;
    unsigned cost = pinter_me_epzs(pi, x, y, log2w, log2h, &refi, lidx, mvp, mv, bi, bit_depth);
    mv_io[0] = mv[MV_X], mv_io[1] = mv[MV_Y];
    g_last_mot_bits = pi->mot_bits[lidx];
    return cost;
}
/* pi->mot_bits[lidx] after the last refdrv_me_epzs call (-1: the searches left it untouched) */
int refdrv_me_epzs_mot_bits(void) { return g_last_mot_bits; }
