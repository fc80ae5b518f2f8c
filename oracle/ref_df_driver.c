/*
 * oracle/ref_df_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_df.so).
 * Runs the reference's own in-loop deblocking driver on caller-supplied planes and maps: the loop of xeve_loop_filter
 * (src_base/xeve_enc.c:2355-2415: both edge directions, vertical edges first, COD bits cleared before each) around the
 * real xeve_deblock -> xeve_deblock_tree -> xeve_deblock_unit -> xeve_deblock_cu_ver / _cu_hor (src_base/xeve_df.c).
 * The CTU split_mode arrays xeve_deblock_tree reads are filled with the reference's own xeve_set_split_mode /
 * xeve_split_get_part_structure from the CU sizes recorded in map_cu_mode.  Also wraps xeve_picbuf_expand.
 */
#include <stdlib.h>
#include <string.h>
#include "xeve_type.h"
#include "xeve_df.h"
#include "xeve_mc.h"

typedef struct {
    int w, h, w_scu, h_scu, log2_max_cuwh, bit_depth_luma, bit_depth_chroma, chroma_format_idc, qp_u_offset, qp_v_offset;
    int qp_chroma[2][100];
} drv_df_params;

static void set_tree(XEVE_CTX *ctx, int lcu, int x, int y, int cuw, int cud, int cup)
{
    int t = (x >> MIN_CU_LOG2) + (y >> MIN_CU_LOG2) * ctx->w_scu;
    int split = (1 << MCU_GET_LOGW(ctx->map_cu_mode[t])) < cuw;
    xeve_set_split_mode(split ? SPLIT_QUAD : NO_SPLIT, cud, cup, cuw, cuw, ctx->max_cuwh, ctx->map_cu_data[lcu].split_mode);
    if(split) {
        XEVE_SPLIT_STRUCT ss;
        xeve_split_get_part_structure(SPLIT_QUAD, x, y, cuw, cuw, cup, cud, ctx->log2_culine, &ss);
        for(int k = 0; k < ss.part_count; k++)
            if(ss.x_pos[k] < ctx->w && ss.y_pos[k] < ctx->h) set_tree(ctx, lcu, ss.x_pos[k], ss.y_pos[k], ss.width[k], ss.cud[k], ss.cup[k]);
    }
}

int refdrv_deblock_picture_tiles(pel *y, pel *u, pel *v, int s_l, int s_c, u32 *map_scu, u32 *map_cu_mode, s8 *map_refi, s16 *map_mv,
                                 const drv_df_params *p, int split_x_lcu, int split_y_lcu, u8 *map_tidx_out);
int refdrv_deblock_picture(pel *y, pel *u, pel *v, int s_l, int s_c, u32 *map_scu, u32 *map_cu_mode, s8 *map_refi, s16 *map_mv,
                           const drv_df_params *p)
{
    return refdrv_deblock_picture_tiles(y, u, v, s_l, s_c, map_scu, map_cu_mode, map_refi, map_mv, p, 0, 0, NULL);
}

/* split_x_lcu / split_y_lcu > 0: a tile boundary after that many CTU columns / rows (up to 2 x 2 tiles, numbered in raster order as xeve_set_tile_info
 * does); the loop of xeve_loop_filter then runs xeve_deblock once per tile and direction.  map_tidx_out (f_scu bytes or NULL) receives the tile of every unit. */
int refdrv_deblock_picture_tiles(pel *y, pel *u, pel *v, int s_l, int s_c, u32 *map_scu, u32 *map_cu_mode, s8 *map_refi, s16 *map_mv,
                                 const drv_df_params *p, int split_x_lcu, int split_y_lcu, u8 *map_tidx_out)
{
    XEVE_CTX  *ctx  = calloc(1, sizeof(*ctx));
    XEVE_CORE *core = calloc(1, sizeof(*core));
    XEVE_PIC   pic;
    XEVE_SH    sh;
    XEVE_TILE  tile[4];
    int        ntile = 0;
    int        ws = XEVE_GET_CHROMA_W_SHIFT(p->chroma_format_idc), hs = XEVE_GET_CHROMA_H_SHIFT(p->chroma_format_idc);
    memset(&pic, 0, sizeof(pic)), memset(&sh, 0, sizeof(sh)), memset(tile, 0, sizeof(tile));
    ctx->w = p->w, ctx->h = p->h, ctx->w_scu = p->w_scu, ctx->h_scu = p->h_scu, ctx->f_scu = p->w_scu * p->h_scu;
    ctx->log2_max_cuwh = p->log2_max_cuwh, ctx->max_cuwh = 1 << p->log2_max_cuwh;
    ctx->log2_culine = p->log2_max_cuwh - MIN_CU_LOG2;
    ctx->w_lcu = (p->w + ctx->max_cuwh - 1) >> p->log2_max_cuwh, ctx->h_lcu = (p->h + ctx->max_cuwh - 1) >> p->log2_max_cuwh;
    ctx->f_lcu = ctx->w_lcu * ctx->h_lcu;
    ctx->map_scu = map_scu, ctx->map_cu_mode = map_cu_mode, ctx->map_refi = (void *)map_refi, ctx->map_mv = (void *)map_mv;
    ctx->map_unrefined_mv = calloc(ctx->f_scu, sizeof(s16) * REFP_NUM * MV_D);
    ctx->map_tidx = calloc(ctx->f_scu, 1);
    ctx->map_cu_data = calloc(ctx->f_lcu, sizeof(XEVE_CU_DATA));
    ctx->tile = tile;
    {
        const int sx = split_x_lcu > 0 && split_x_lcu < ctx->w_lcu ? split_x_lcu : ctx->w_lcu, sy = split_y_lcu > 0 && split_y_lcu < ctx->h_lcu ? split_y_lcu : ctx->h_lcu;
        const int x0[2] = {0, sx}, x1[2] = {sx, ctx->w_lcu}, y0[2] = {0, sy}, y1[2] = {sy, ctx->h_lcu};
        for(int ty = 0; ty < 2; ty++)
            for(int tx = 0; tx < 2; tx++) {
                if(x0[tx] >= x1[tx] || y0[ty] >= y1[ty]) continue;
                tile[ntile].ctba_rs_first = y0[ty] * ctx->w_lcu + x0[tx], tile[ntile].w_ctb = x1[tx] - x0[tx], tile[ntile].h_ctb = y1[ty] - y0[ty];
                const int per = ctx->max_cuwh >> MIN_CU_LOG2;
                for(int j = y0[ty] * per; j < y1[ty] * per && j < ctx->h_scu; j++)
                    for(int i = x0[tx] * per; i < x1[tx] * per && i < ctx->w_scu; i++) ctx->map_tidx[j * ctx->w_scu + i] = (u8)ntile;
                ntile++;
            }
        if(map_tidx_out) memcpy(map_tidx_out, ctx->map_tidx, ctx->f_scu);
    }
    ctx->sh = &sh, sh.qp_u_offset = p->qp_u_offset, sh.qp_v_offset = p->qp_v_offset;
    ctx->sps.bit_depth_luma_minus8 = p->bit_depth_luma - 8, ctx->sps.bit_depth_chroma_minus8 = p->bit_depth_chroma - 8;
    ctx->sps.chroma_format_idc = p->chroma_format_idc;
    ctx->param.codec_bit_depth = p->bit_depth_chroma;
    for(int c = 0; c < 2; c++) {
        memcpy(ctx->qp_chroma_dynamic_ext[c], p->qp_chroma[c], sizeof(int) * 100);
        ctx->qp_chroma_dynamic[c] = &ctx->qp_chroma_dynamic_ext[c][6 * (p->bit_depth_chroma - 8)]; /* xeve_util.c:1845-1846 */
    }
    ctx->fn_deblock_tree = xeve_deblock_tree, ctx->fn_deblock_unit = xeve_deblock_unit;
    pic.y = y, pic.u = u, pic.v = v, pic.s_l = s_l, pic.s_c = s_c, pic.w_l = p->w, pic.h_l = p->h, pic.w_c = p->w >> ws, pic.h_c = p->h >> hs;
    for(int ly = 0; ly < ctx->h_lcu; ly++)
        for(int lx = 0; lx < ctx->w_lcu; lx++) set_tree(ctx, ly * ctx->w_lcu + lx, lx << p->log2_max_cuwh, ly << p->log2_max_cuwh, ctx->max_cuwh, 0, 0);
    core->ctx = ctx;
    for(int is_hor_edge = 0; is_hor_edge <= 1; is_hor_edge++) { /* xeve_loop_filter, one slice, tile after tile */
        for(u32 i = 0; i < ctx->f_scu; i++) MCU_CLR_COD(ctx->map_scu[i]);
        core->deblock_is_hor = is_hor_edge;
        for(int t = 0; t < ntile; t++) xeve_deblock(ctx, &pic, t, 0, core);
    }
    free(ctx->map_unrefined_mv), free(ctx->map_tidx), free(ctx->map_cu_data), free(ctx), free(core);
    return 0;
}

void refdrv_picbuf_expand(pel *y, pel *u, pel *v, int s_l, int s_c, int w_l, int h_l, int w_c, int h_c, int exp_l, int exp_c, int chroma_format_idc)
{
    XEVE_PIC pic;
    memset(&pic, 0, sizeof(pic));
    pic.y = y, pic.u = u, pic.v = v, pic.s_l = s_l, pic.s_c = s_c, pic.w_l = w_l, pic.h_l = h_l, pic.w_c = w_c, pic.h_c = h_c;
    xeve_picbuf_expand(&pic, exp_l, exp_c, chroma_format_idc);
}

/* xeve_mc (src_base/xeve_mc.c:465-610) on caller-supplied planes: refs[refi * 2 + list] = {y, u, v (sample (0,0)), poc}.  The
 * dispatch pointers are process globals that the encoder sets in xeve_platform_init_func; here: the plain-C tables. */
typedef struct { pel *y, *u, *v; int poc, pad_; } drv_refpic;
typedef struct { int x, y; s16 mv[2][2]; s8 refi[2]; s8 pad_[2]; } drv_mc_job;
void refdrv_mc_cu(const drv_refpic *refs, int nref, int s_l, int s_c, int pic_w, int pic_h, const drv_mc_job *job, int w, int h, int bd_l, int bd_c,
                  int chroma_format_idc, pel *pred_y, pel *pred_u, pel *pred_v)
{
    static __thread XEVE_PIC  pics[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static __thread XEVE_REFP refp[XEVE_MAX_NUM_REF_PICS][REFP_NUM];
    static __thread pel(*pred)[N_C][MAX_CU_DIM];
    if(!pred) pred = calloc(REFP_NUM, sizeof(*pred));
    xeve_func_mc_l = xeve_tbl_mc_l, xeve_func_mc_c = xeve_tbl_mc_c, xeve_func_average_no_clip = &xeve_average_16b_no_clip;
    for(int r = 0; r < nref; r++)
        for(int l = 0; l < REFP_NUM; l++) {
            const drv_refpic *s = &refs[r * 2 + l];
            pics[r][l].y = s->y, pics[r][l].u = s->u, pics[r][l].v = s->v, pics[r][l].s_l = s_l, pics[r][l].s_c = s_c, pics[r][l].poc = s->poc;
            refp[r][l].pic = &pics[r][l], refp[r][l].poc = s->poc;
        }
    s8  refi[REFP_NUM] = {job->refi[0], job->refi[1]};
    s16 mv[REFP_NUM][MV_D] = {{job->mv[0][0], job->mv[0][1]}, {job->mv[1][0], job->mv[1][1]}};
    xeve_mc(job->x, job->y, pic_w, pic_h, w, h, refi, mv, refp, pred, bd_l, bd_c, chroma_format_idc);
    int cw = w >> XEVE_GET_CHROMA_W_SHIFT(chroma_format_idc), ch = h >> XEVE_GET_CHROMA_H_SHIFT(chroma_format_idc);
    memcpy(pred_y, pred[0][Y_C], sizeof(pel) * w * h);
    if(chroma_format_idc) memcpy(pred_u, pred[0][U_C], sizeof(pel) * cw * ch), memcpy(pred_v, pred[0][V_C], sizeof(pel) * cw * ch);
}
