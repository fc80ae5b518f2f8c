/*
 * oracle/ref_affine_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_affine.so, linked against the Main-profile library).
 * Calls the reference's own xeve_affine_mc (src_main/xevem_mc.c:2236-2339: sub-block size, per list xeve_affine_mc_lc -- sub-block interpolation or the enhanced
 * interpolation filter --, the bi-prediction average) on caller-supplied planes: builds the XEVE_PIC / XEVE_REFP records the function reads and copies its three
 * prediction planes out.  Also exposes derive_affine_subblock_size_bi's answer (the path a case takes).
 */
#include <stdlib.h>
#include <string.h>
#include "xevem_type.h"
#include "xevem_mc.h"

/* planes[(refi * 2 + list) * 3 + c]: sample (0, 0) of component c of reference picture refi of the list; out_y / out_u / out_v dense w x h (w/2 x h/2) */
int refdrv_affine_mc(int x, int y, int pic_w, int pic_h, int w, int h, const s8 refi[2], const s16 mv[2][3][2], pel *const *planes, int s_l, int s_c, int vertex_num,
                     int bit_depth, pel *out_y, pel *out_u, pel *out_v, int *path)
{
    static XEVE_REFP refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static XEVE_PIC  pics[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    pel(*pred)[N_C][MAX_CU_DIM] = malloc(sizeof(pel) * 2 * N_C * MAX_CU_DIM);
    pel *tmp                    = malloc(sizeof(pel) * (MAX_CU_SIZE + 2) * (MAX_CU_SIZE + 2) * 2);
    s16  ac[REFP_NUM][VER_NUM][MV_D];
    s8   rf[REFP_NUM] = {refi[0], refi[1]};
    memset(ac, 0, sizeof(ac));
    xeve_func_mc_l = xeve_tbl_mc_l, xeve_func_mc_c = xeve_tbl_mc_c; /* (what xeve_platform_init_func would choose a variant of: the plain C tables, xeve_mc.c:383-399) */
    for(int l = 0; l < 2; l++)
        for(int v = 0; v < 3; v++) ac[l][v][MV_X] = mv[l][v][0], ac[l][v][MV_Y] = mv[l][v][1];
    for(int l = 0; l < 2; l++)
        if(refi[l] >= 0) {
            XEVE_PIC *p = &pics[refi[l]][l];
            memset(p, 0, sizeof(*p));
            p->y = planes[(refi[l] * 2 + l) * 3 + 0], p->u = planes[(refi[l] * 2 + l) * 3 + 1], p->v = planes[(refi[l] * 2 + l) * 3 + 2];
            p->s_l = s_l, p->s_c = s_c, p->w_l = pic_w, p->h_l = pic_h, p->w_c = pic_w >> 1, p->h_c = pic_h >> 1;
            refp[refi[l]][l].pic = p;
        }
    if(path) {
        int  sw = 0, sh = 0;
        BOOL mem = FALSE;
        derive_affine_subblock_size_bi(ac, rf, w, h, &sw, &sh, vertex_num, &mem);
        path[0] = sw, path[1] = sh, path[2] = mem;
    }
    xeve_affine_mc(x, y, pic_w, pic_h, w, h, rf, ac, refp, pred, vertex_num, tmp, bit_depth, bit_depth, 1);
    memcpy(out_y, pred[0][Y_C], sizeof(pel) * w * h);
    memcpy(out_u, pred[0][U_C], sizeof(pel) * (w >> 1) * (h >> 1));
    memcpy(out_v, pred[0][V_C], sizeof(pel) * (w >> 1) * (h >> 1));
    free(pred), free(tmp);
    return 0;
}
