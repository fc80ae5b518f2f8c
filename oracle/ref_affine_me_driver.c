/*
 * oracle/ref_affine_me_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_affine_me.so, linked against the Main-profile library).
 * pinter_affine_me_gradient is `static` in the reference (src_main/xevem_pinter.c:4290).  To pin the oracle's restatement (xo_affine_me_gradient) against the REAL function this
 * driver compiles the reference's xevem_pinter.c, unmodified and where it lies, into its own translation unit (#include of the .c file) and exports a flat C wrapper around the
 * static function: it fills the fields of XEVE_PINTER the function reads (the reference picture, the original or the bi-prediction target, lambda, the other list's bits).
 * All other symbols (xeve_affine_mc_l, the SATD / Sobel / equation kernels' tables) come from oracle/_ref/libxevem_ref.so.
 */
#include "xevem_pinter.c"

/* returns the function's value (cost_best - the cost of its vector bits); mv_io: in the start vectors, out the best ones ([3][2]).  org: bi ? dense w x h block : the picture's
 * plane (sample (0, 0)) with pitch s_org.  simd: 0 = the plain C kernels (xevem_util.c:3958-3960), 1 = the SSE ones the application's build runs (:3932-3934) */
unsigned refdrv_affine_me_gradient(pel *ref_y, int s_l, int pic_w, int pic_h, pel *org, int s_org, int x, int y, int log2w, int log2h, int refi, int lidx, const s16 mvp_in[3][2],
                                   s16 mv_io[3][2], int bi, int vertex_num, int bit_depth, unsigned lambda_mv, int num_refp, int mot_bits_other, int simd)
{
    static XEVE_PINTER *pi;
    static XEVE_PIC     pic;
    static XEVE_REFP    refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static pel         *tmp;
    if(!pi) pi = calloc(1, sizeof(*pi)), tmp = malloc(sizeof(pel) * (MAX_CU_SIZE + 2) * (MAX_CU_SIZE + 2) * 2);
    xeve_func_mc_l = xeve_tbl_mc_l, xeve_func_satd = xeve_tbl_satd_16b, xeve_func_diff = xeve_tbl_diff_16b;
    if(simd) {
        xevem_func_aff_h_sobel_flt = &xevem_scaled_horizontal_sobel_filter_sse, xevem_func_aff_v_sobel_flt = &xevem_scaled_vertical_sobel_filter_sse;
        xevem_func_aff_eq_coef_comp = &xevem_equal_coeff_computer_sse;
        xeve_func_satd = xeve_tbl_satd_16b_sse, xeve_func_diff = xeve_tbl_diff_16b_sse;
    }
    else {
        xevem_func_aff_h_sobel_flt = &xevem_scaled_horizontal_sobel_filter, xevem_func_aff_v_sobel_flt = &xevem_scaled_vertical_sobel_filter;
        xevem_func_aff_eq_coef_comp = &xevem_equal_coeff_computer;
    }
    memset(&pic, 0, sizeof(pic));
    pic.y = ref_y, pic.s_l = s_l, pic.w_l = pic_w, pic.h_l = pic_h;
    refp[refi][lidx].pic = &pic;
    pi->refp = refp;
    if(bi) memcpy(pi->org_bi, org, sizeof(s16) << (log2w + log2h));
    else pi->o[Y_C] = org, pi->s_o[Y_C] = s_org;
    pi->num_refp = (u8)num_refp, pi->lambda_mv = lambda_mv;
    pi->mot_bits[1 - lidx] = mot_bits_other;
    s16 mvp[VER_NUM][MV_D], mv[VER_NUM][MV_D];
    memset(mvp, 0, sizeof(mvp)), memset(mv, 0, sizeof(mv));
    for(int v = 0; v < 3; v++) mvp[v][MV_X] = mvp_in[v][0], mvp[v][MV_Y] = mvp_in[v][1], mv[v][MV_X] = mv_io[v][0], mv[v][MV_Y] = mv_io[v][1];
    s8 ri = (s8)refi;
    const unsigned r = pinter_affine_me_gradient(pi, x, y, log2w, log2h, &ri, lidx, mvp, mv, bi, vertex_num, tmp, bit_depth, bit_depth, 1);
    for(int v = 0; v < 3; v++) mv_io[v][0] = mv[v][MV_X], mv_io[v][1] = mv[v][MV_Y];
    return r;
}
