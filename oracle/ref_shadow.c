/*
 * oracle/ref_shadow.c -- TEST INFRASTRUCTURE (built only where /root/reference exists, output oracle/_ref/libxeve_ref_shadow.so).
 *
 * The checker's side of the in-encoder tests, split from the product's binding (shim/xeve_hip_shim.c, which this file EMBEDS so that its routes can be wrapped):
 *   XEVE_SHIM_SHADOW_TREE=<libxeve_oracle.so>   shadow mode: the oracle walks and writes every CTU beside the live reference encoder and every product is compared
 *                                               (pins the ORACLE; also the golden recorder of tests/golden/make_tree_golden.py)
 *   XEVE_SHIM_TREE_ORACLE=<libxeve_oracle.so>   the shim's CTU route adapter with the oracle as its engine (CPU test of the adapter's stores)
 *   XEVE_SHIM_TREE_CHECK=<libxeve_oracle.so>    with the GPU routes on: the oracle walks a snapshot of every CTU's inputs beside the device and the results are compared
 * The product shim knows none of these switches and opens nothing but $XEVE_HIP_LIB.
 */
#define XEVE_SHIM_EMBEDDED 1
#include "../shim/xeve_hip_shim.c"

/* SHADOW MODE for the CTU mode decision (ctx->fn_mode_analyze_lcu = mode_analyze_lcu -> mode_coding_tree, xeve_mode.c:2007-2610).  Before the reference analyses a
 * CTU the adapter snapshots what the walk reads (the picture being reconstructed, the unit maps, the entry coder state); after the reference has run it lets the
 * oracle's restatement walk the same CTU on the snapshot and compares everything the walk produces: split modes, prediction modes, coded-block counts, unit maps,
 * coefficients, reconstruction, exit coder state.  The encoder continues with the reference's own results; mismatches are counted and reported at exit.  CPU only:
 * this pins the ORACLE against the live encoder. */
#include "xeve_oracle.h"
static double (*xo_tree)(const xo_pel *const *, int, int, xo_pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                         const xo_tree_params *, int, int, xo_ctu_data *, xo_sbac *);
static double (*xo_tree_any)(const xo_pel *const *, int, int, xo_pel *const *, int, int, uint32_t *, int8_t *, const uint8_t *, uint32_t *, const xo_sbac *,
                             const xo_tree_params *, const xo_tree_inter *, int, int, xo_ctu_data *, xo_sbac *);
static unsigned long long shadow_ctus, shadow_bad, shadow_skipped, shadow_inter_ctus;
static void (*xo_dbk_begin)(const xo_dbk_ctx *);
static void (*xo_dbk_end)(void);

/* what the inter side of the walk is handed (the same derivation as shim_pinter_analyze_cu above, once per CTU); tab: 16 entries */
static int shadow_inter_setup(XEVE_CTX *ctx, XEVE_CORE *core, xo_tree_inter *I, xo_refpic *tab, int16_t (*map_mv)[2][2], int8_t (*map_refi)[2])
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

#define sbac_to_flat(h, sb) sbac_to_flat((xeve_hip_sbac *)(h), sb) /* (the shim's: xo_sbac and xeve_hip_sbac are the same record) */

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
    if(!is_i) ok = shadow_inter_setup(ctx, core, &TI, tab, mv, refi) == 0;
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
    if(ctx->pps.cu_qp_delta_enabled_flag || (ctx->param.rdo_dbk_switch && !xo_dbk_begin) || ctx->param.tool_iqt || ctx->sps.tool_admvp || ctx->log2_max_cuwh != 6 || idc == 2 ||
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
        if(shadow_inter_setup(ctx, core, &TI, tab, m_mv, m_refi) != 0) {
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
    xo_deblock_params dbk_dp;
    int8_t  *z_refi = NULL;
    int16_t *z_mv = NULL;
    if(ctx->param.rdo_dbk_switch) { /* preset slow: the oracle's candidates get the loop filter's share too, read off the SNAPSHOT (reconstruction so far, unit maps) */
        const int bc = ctx->sps.bit_depth_chroma_minus8;
        memset(&dbk_dp, 0, sizeof(dbk_dp));
        dbk_dp.w = ctx->w, dbk_dp.h = ctx->h, dbk_dp.w_scu = ctx->w_scu, dbk_dp.h_scu = ctx->h_scu, dbk_dp.log2_max_cuwh = ctx->log2_max_cuwh;
        dbk_dp.bit_depth_luma = ctx->sps.bit_depth_luma_minus8 + 8, dbk_dp.bit_depth_chroma = bc + 8, dbk_dp.chroma_format_idc = idc;
        for(int c = 0; c < 2; c++)
            for(int i = 0; i < 100; i++) dbk_dp.qp_chroma[c][i] = i <= 57 + 6 * bc ? ctx->qp_chroma_dynamic_ext[c][i] : 0;
        if(!m_refi) z_refi = malloc(2 * nscu), z_mv = calloc(4 * nscu, sizeof(int16_t)), memset(z_refi, -1, 2 * nscu);
        xo_dbk_ctx D;
        memset(&D, 0, sizeof(D));
        D.mod[0] = mod[0], D.mod[1] = mod[1], D.mod[2] = mod[2], D.s_mod_l = pm->s_l, D.s_mod_c = pm->s_c, D.qp = ctx->tile[core->tile_idx].qp;
        D.map_scu = m_scu, D.map_refi = m_refi ? (const int8_t *)m_refi : z_refi, D.map_mv = m_mv ? (const int16_t *)m_mv : z_mv, D.map_tidx = ctx->map_tidx, D.dp = &dbk_dp;
        P.rdo_dbk = 1;
        xo_dbk_begin(&D);
    }
    if(is_i) (void)xo_tree(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, m_scu, m_ipm, ctx->map_tidx, m_cum, &entry, &P, x0, y0, &out, &next);
    else {
        (void)xo_tree_any(org, pi->s_o[Y_C], pi->s_o[U_C], mod, pm->s_l, pm->s_c, m_scu, m_ipm, ctx->map_tidx, m_cum, &entry, &P, &TI, x0, y0, &out, &next);
        shadow_inter_ctus++;
    }
    if(ctx->param.rdo_dbk_switch) xo_dbk_end(), free(z_refi), free(z_mv);

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



/* XEVE_SHIM_DBK_LOG=<file>: every call of the reference's calc_delta_dist_filter_boundary with what it returned -- held against the oracle's own log (XO_DBK_LOG) when a
 * preset-slow bitstream differs */
void calc_delta_dist_filter_boundary(XEVE_CTX *ctx, XEVE_PIC *pic_rec, XEVE_PIC *pic_org, int cuw, int cuh, pel (*src)[MAX_CU_DIM], int s_src, int x, int y, u16 avail_lr,
                                     u8 intra_flag, u8 cbf_l, s8 *refi, s16 (*mv)[MV_D], u8 is_mv_from_mvf, XEVE_CORE *core)
{
    static void (*real)(XEVE_CTX *, XEVE_PIC *, XEVE_PIC *, int, int, pel (*)[MAX_CU_DIM], int, int, int, u16, u8, u8, s8 *, s16 (*)[MV_D], u8, XEVE_CORE *);
    if(!real) real = dlsym(RTLD_NEXT, "calc_delta_dist_filter_boundary");
    real(ctx, pic_rec, pic_org, cuw, cuh, src, s_src, x, y, avail_lr, intra_flag, cbf_l, refi, mv, is_mv_from_mvf, core);
    const char *f = getenv("XEVE_SHIM_DBK_LOG");
    if(f) {
        FILE *o = fopen(f, "a");
        fprintf(o, "poc %d x %d y %d cu %d intra %d cbf %d refi %d %d mv %d %d %d %d lr %d -> %lld %lld %lld\n", (int)ctx->poc.poc_val, x, y, cuw, intra_flag, cbf_l, refi ? refi[0] : -9,
                refi ? refi[1] : -9, mv ? mv[0][0] : 0, mv ? mv[0][1] : 0, mv ? mv[1][0] : 0, mv ? mv[1][1] : 0, avail_lr, (long long)core->delta_dist[0], (long long)core->delta_dist[1],
                (long long)core->delta_dist[2]);
        fclose(o);
    }
}

/* ---- the shim's CTU route adapter with the ORACLE as its engine (XEVE_SHIM_TREE_ORACLE) -------------------------------------------------------------------------- */
static void inter_to_oracle(xo_tree_inter *T, xo_refpic *tab, const xeve_hip_tree_inter *H)
{
    memset(T, 0, sizeof(*T));
    memcpy(tab, H->refp, 16 * sizeof(*tab)); /* (same record) */
    T->refp = tab, T->s_ref_l = H->s_ref_l, T->s_ref_c = H->s_ref_c, T->map_mv = (int16_t(*)[2][2])H->map_mv, T->map_refi = (int8_t(*)[2])H->map_refi;
    T->col0 = (const int16_t(*)[2][2])H->col_mv0, T->col1 = (const int16_t(*)[2][2])H->col_mv1, T->ecu_depth = H->ecu_depth;
    memcpy(&T->ipar.rdo, &H->ipar.rdo, sizeof(T->ipar.rdo)), memcpy(&T->ipar.me.me, &H->ipar.me.me, sizeof(T->ipar.me.me));
    T->ipar.me.spel.lambda_mv = H->ipar.me.me.lambda_mv, T->ipar.me.spel.hpel_cnt = H->ipar.me.hpel_cnt, T->ipar.me.spel.qpel_cnt = H->ipar.me.qpel_cnt;
    memcpy(T->ipar.refi_bits, H->ipar.refi_bits, sizeof(T->ipar.refi_bits)), memcpy(T->ipar.range_recentre, H->ipar.range_recentre, sizeof(T->ipar.range_recentre));
    T->ipar.max_cand = H->ipar.max_cand, T->ipar.poc = H->ipar.poc, T->ipar.col_list_poc0 = H->ipar.col_list_poc0, T->ipar.skip_th = H->ipar.skip_th;
}
static int oracle_tree_host(const pel *const *org, int so_l, int so_c, pel *const *mod, int sm_l, int sm_c, uint32_t *scu, int8_t *ipm, const uint8_t *tidx, uint32_t *cum,
                            const xeve_hip_sbac *entry, const xeve_hip_tree_params *P, int x0, int y0, xeve_hip_ctu_data *out, xeve_hip_sbac *next, double *cost)
{
    *cost = xo_tree((const xo_pel *const *)org, so_l, so_c, (xo_pel *const *)mod, sm_l, sm_c, scu, ipm, tidx, cum, (const xo_sbac *)entry, (const xo_tree_params *)P, x0, y0,
                    (xo_ctu_data *)out, (xo_sbac *)next);
    return 0;
}
static int oracle_tree_any_host(const pel *const *org, int so_l, int so_c, pel *const *mod, int sm_l, int sm_c, uint32_t *scu, int8_t *ipm, const uint8_t *tidx, uint32_t *cum,
                                const xeve_hip_sbac *entry, const xeve_hip_tree_params *P, const xeve_hip_tree_inter *H, int pad_l, int pad_c, int x0, int y0,
                                xeve_hip_ctu_data *out, xeve_hip_sbac *next, double *cost)
{
    xo_tree_inter T;
    xo_refpic     tab[16];
    (void)pad_l, (void)pad_c;
    inter_to_oracle(&T, tab, H);
    *cost = xo_tree_any((const xo_pel *const *)org, so_l, so_c, (xo_pel *const *)mod, sm_l, sm_c, scu, ipm, tidx, cum, (const xo_sbac *)entry, (const xo_tree_params *)P, &T, x0,
                        y0, (xo_ctu_data *)out, (xo_sbac *)next);
    return 0;
}

/* ---- XEVE_SHIM_TREE_CHECK: the device's engine wrapped -- the oracle walks a snapshot of the same inputs first, the two results are compared per CTU (locates a
 * deviation of the device walk inside a long encode: which CTU, which field).  The picture's geometry comes from the parameters the adapter hands over. ----------- */
static hip_tree_host_fn     gpu_tree_host;
static hip_tree_any_host_fn gpu_tree_any_host;
static unsigned long long   tree_check_ctus, tree_check_bad;
static int checked_walk(const pel *const *org, int so_l, int so_c, pel *const *mod, int sm_l, int sm_c, uint32_t *scu, int8_t *ipm, const uint8_t *tidx, uint32_t *cum,
                        const xeve_hip_sbac *entry, const xeve_hip_tree_params *P, const xeve_hip_tree_inter *H, int pad_l, int pad_c, int x0, int y0, xeve_hip_ctu_data *out,
                        xeve_hip_sbac *next, double *cost)
{
    static __thread xo_ctu_data chk;
    xo_sbac   chk_next;
    const int nscu = P->ip.w_scu * P->ip.h_scu, hs = P->ip.chroma_format_idc <= 1, hc = P->pic_h >> hs, is_i = H == NULL;
    xo_pel   *cm[3] = {malloc(sizeof(pel) * sm_l * (P->pic_h + 1)), malloc(sizeof(pel) * sm_c * (hc + 1)), malloc(sizeof(pel) * sm_c * (hc + 1))};
    uint32_t *c_scu = malloc(4 * nscu), *c_cum = malloc(4 * nscu);
    int8_t   *c_ipm = malloc(nscu), (*c_refi)[2] = malloc(sizeof(*c_refi) * nscu);
    int16_t (*c_mv)[2][2] = malloc(sizeof(*c_mv) * nscu);
    memcpy(cm[0], mod[0], sizeof(pel) * sm_l * P->pic_h), memcpy(cm[1], mod[1], sizeof(pel) * sm_c * hc), memcpy(cm[2], mod[2], sizeof(pel) * sm_c * hc);
    memcpy(c_scu, scu, 4 * nscu), memcpy(c_cum, cum, 4 * nscu), memcpy(c_ipm, ipm, nscu);
    if(is_i) (void)xo_tree((const xo_pel *const *)org, so_l, so_c, cm, sm_l, sm_c, c_scu, c_ipm, tidx, c_cum, (const xo_sbac *)entry, (const xo_tree_params *)P, x0, y0, &chk, &chk_next);
    else {
        xo_tree_inter T;
        xo_refpic     tab[16];
        inter_to_oracle(&T, tab, H);
        memcpy(c_mv, H->map_mv, sizeof(*c_mv) * nscu), memcpy(c_refi, H->map_refi, sizeof(*c_refi) * nscu);
        T.map_mv = c_mv, T.map_refi = c_refi;
        (void)xo_tree_any((const xo_pel *const *)org, so_l, so_c, cm, sm_l, sm_c, c_scu, c_ipm, tidx, c_cum, (const xo_sbac *)entry, (const xo_tree_params *)P, &T, x0, y0, &chk, &chk_next);
    }
    const int rc = is_i ? gpu_tree_host(org, so_l, so_c, mod, sm_l, sm_c, scu, ipm, tidx, cum, entry, P, x0, y0, out, next, cost)
                        : gpu_tree_any_host(org, so_l, so_c, mod, sm_l, sm_c, scu, ipm, tidx, cum, entry, P, H, pad_l, pad_c, x0, y0, out, next, cost);
    if(rc == 0) {
        const xo_ctu_data *o = (const xo_ctu_data *)out;
        int bad = 0;
#define CHK(field) do { if(memcmp(o->field, chk.field, sizeof(o->field))) { if(tree_check_bad < 8) { size_t k_ = 0; while(((const char *)o->field)[k_] == ((const char *)chk.field)[k_]) k_++; \
            fprintf(stderr, "[tree check] CTU (%d,%d) slice %d: %s differs at byte %zu\n", x0, y0, P->ip.slice_type, #field, k_); } bad = 1; } } while(0)
        CHK(split_mode); CHK(pred_mode); CHK(ipm); CHK(depth); CHK(nnz); CHK(map_scu); CHK(map_cu_mode); CHK(coef); CHK(reco); CHK(mv); CHK(mvd); CHK(refi); CHK(mvp_idx);
#undef CHK
        if(memcmp(next, &chk_next, sizeof(chk_next))) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU (%d,%d): exit coder state differs\n", x0, y0); bad = 1; }
        if(memcmp(c_scu, scu, 4 * nscu) || memcmp(c_ipm, ipm, nscu) || memcmp(c_cum, cum, 4 * nscu)) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU (%d,%d): unit maps differ\n", x0, y0); bad = 1; }
        if(!is_i && (memcmp(c_mv, H->map_mv, sizeof(*c_mv) * nscu) || memcmp(c_refi, H->map_refi, sizeof(*c_refi) * nscu))) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU (%d,%d): motion maps differ\n", x0, y0); bad = 1; }
        if(memcmp(cm[0], mod[0], sizeof(pel) * sm_l * P->pic_h)) { if(tree_check_bad < 8) fprintf(stderr, "[tree check] CTU (%d,%d): luma picture differs\n", x0, y0); bad = 1; }
        __sync_fetch_and_add(&tree_check_ctus, 1), __sync_fetch_and_add(&tree_check_bad, bad);
    }
    free(cm[0]), free(cm[1]), free(cm[2]), free(c_scu), free(c_cum), free(c_ipm), free(c_mv), free(c_refi);
    return rc;
}
static int checked_tree_host(const pel *const *org, int so_l, int so_c, pel *const *mod, int sm_l, int sm_c, uint32_t *scu, int8_t *ipm, const uint8_t *tidx, uint32_t *cum,
                             const xeve_hip_sbac *entry, const xeve_hip_tree_params *P, int x0, int y0, xeve_hip_ctu_data *out, xeve_hip_sbac *next, double *cost)
{
    return checked_walk(org, so_l, so_c, mod, sm_l, sm_c, scu, ipm, tidx, cum, entry, P, NULL, 0, 0, x0, y0, out, next, cost);
}

static void shadow_report(void)
{
    if(ap_pics) fprintf(stderr, "[xeve_hip_shim] shadow pictures: %llu pictures decided and written by the oracle on its own (%llu bytes of slice data), %llu differ from the reference's\n", ap_pics, ap_bytes, ap_bad);
    if(eco_ctus) fprintf(stderr, "[xeve_hip_shim] shadow writer: %llu CTUs written by the oracle beside xeve_eco_tree (%llu bytes of bitstream compared), %llu differ\n", eco_ctus, eco_bytes, eco_bad);
    if(gpu_tree_host) fprintf(stderr, "[xeve_hip_shim] device walk checked against the oracle per CTU: %llu CTUs, %llu differ\n", tree_check_ctus, tree_check_bad);
    if(shadow_ctus || shadow_skipped) fprintf(stderr, "[xeve_hip_shim] shadow tree walk: %llu CTUs compared, %llu differ (%llu not covered), %llu of them in P / B pictures\n", shadow_ctus, shadow_bad, shadow_skipped, shadow_inter_ctus);
}

void xeve_platform_init_func(XEVE_CTX *ctx)
{
    void (*orig)(XEVE_CTX *) = (void (*)(XEVE_CTX *))dlsym(RTLD_NEXT, "xeve_platform_init_func");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim] reference xeve_platform_init_func not found\n"); abort(); }
    orig(ctx);
    if(getenv("XEVE_SHIM_SHADOW_TREE") && ctx->fn_mode_analyze_lcu && ctx->fn_mode_analyze_lcu != shim_mode_analyze_lcu) {
        void *oh = dlopen(getenv("XEVE_SHIM_SHADOW_TREE"), RTLD_NOW | RTLD_LOCAL);
        if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra"))) { fprintf(stderr, "[xeve_hip_shim] shadow tree: %s\n", dlerror()); abort(); }
        xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu");
        xo_dbk_begin = dlsym(oh, "xo_rdo_dbk_begin"), xo_dbk_end = dlsym(oh, "xo_rdo_dbk_end");
        xo_eco = dlsym(oh, "xo_eco_ctu"), xo_tile_end = dlsym(oh, "xo_eco_tile_end");
        orig_mode_analyze_lcu = ctx->fn_mode_analyze_lcu, ctx->fn_mode_analyze_lcu = shim_mode_analyze_lcu;
        if(ctx->fn_loop_filter != shim_shadow_loop_filter) orig_shadow_loop_filter = ctx->fn_loop_filter, ctx->fn_loop_filter = shim_shadow_loop_filter;
        fprintf(stderr, "[xeve_hip_shim] shadow mode: the oracle walks and writes every CTU beside the reference\n");
        atexit(shadow_report);
    }
    if(getenv("XEVE_SHIM_TREE_ORACLE") && ctx->fn_mode_analyze_lcu && ctx->fn_mode_analyze_lcu != shim_route_mode_analyze_lcu) {
        void *oh = dlopen(getenv("XEVE_SHIM_TREE_ORACLE"), RTLD_NOW | RTLD_LOCAL);
        if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra"))) { fprintf(stderr, "[xeve_hip_shim] tree route (oracle engine): %s\n", dlerror()); abort(); }
        hip_tree_host = oracle_tree_host;
        if(!getenv("XEVE_SHIM_TREE_I_ONLY") && (xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu"))) hip_tree_any_host = oracle_tree_any_host;
        orig_mode_analyze_lcu = ctx->fn_mode_analyze_lcu, ctx->fn_mode_analyze_lcu = shim_route_mode_analyze_lcu, tree_engine = "oracle (CPU)";
        fprintf(stderr, "[xeve_hip_shim] CTU mode decision of I pictures served by the ORACLE through the route adapter (CPU test of the adapter)\n");
        atexit(report);
    }
    shim_install(ctx); /* the product's routes ($XEVE_HIP_LIB and the XEVE_HIP_SHIM_* switches) */
    if(getenv("XEVE_SHIM_TREE_CHECK") && hip_tree_host && hip_tree_host != oracle_tree_host && ctx->fn_mode_analyze_lcu == shim_route_mode_analyze_lcu && !gpu_tree_host) {
        void *oh = dlopen(getenv("XEVE_SHIM_TREE_CHECK"), RTLD_NOW | RTLD_LOCAL);
        if(!oh || !(xo_tree = dlsym(oh, "xo_mode_analyze_ctu_intra")) || !(xo_tree_any = dlsym(oh, "xo_mode_analyze_ctu"))) { fprintf(stderr, "[xeve_hip_shim] tree check: %s\n", dlerror()); abort(); }
        gpu_tree_host = hip_tree_host, hip_tree_host = checked_tree_host;
        if(hip_tree_any_host) gpu_tree_any_host = hip_tree_any_host, hip_tree_any_host = checked_walk;
        atexit(shadow_report);
    }
}
