/* oracle/ref_shim_affine.c -- TEST INFRASTRUCTURE ONLY: LD_PRELOAD interposer for the MAIN-profile reference library (oracle/_ref/libxevem_ref.so) that routes the affine
 * motion compensation of a CU to the GPU.  The Main encoder calls xeve_affine_mc BY NAME -- from the affine merge analysis (src_main/xevem_pinter.c:1947), from every round of
 * the affine gradient search (:4659) and from the affine bi-prediction (:4918) -- so the symbol is interposed and forwards to the HIP host form
 * (xeve_hip_affine_mc_host: the reference's arguments, host planes).  Nothing else of the encoder is touched: the affine SEARCH around it stays the reference's.
 * XEVE_HIP_LIB unset = plain reference run; with XEVE_HIP_SHIM_AFFINE_COUNT=1 the calls are counted and go to the reference's own function (which clips reach affine MC
 * at all).  Needs the reference's headers (XEVE_REFP / XEVE_PIC layouts): built by oracle/Makefile into oracle/_ref/.
 *   Round 6, second route: with XEVE_HIP_SHIM_AFFINE_ME=1 the affine gradient SEARCH itself goes to the GPU -- xevem_pinter_create (called by name, xevem_util.c:3979) is
 * interposed and, once the reference has filled its inter-prediction objects, every thread's pi->fn_affine_me (src_base/xeve_type.h:448; the reference binds the static
 * pinter_affine_me_gradient, xevem_pinter.c:6285) is bound to a forwarder to xeve_hip_affine_me_host.  XEVE_HIP_SHIM_AFFINE_VERIFY=1 then runs the reference's own function
 * behind every call and reports the first difference (vectors or value). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include "xevem_type.h"

typedef struct {
    const pel *y, *u, *v;
    int        poc, pad_;
} hip_refpic; /* xeve_hip_refpic (include/xeve_hip.h) */
static void *hip;
static int   hip_mc_off;
static int (*h_mc)(int, int, int, int, int, int, const s8 *, const s16 (*)[3][MV_D], const hip_refpic *, int, int, int, int, int, int, pel *, pel *, pel *, int, int);
static const char *(*h_err)(void);
static unsigned long long n_calls, n_eligible;
static void (*orig)(int, int, int, int, int, int, s8 *, s16 (*)[VER_NUM][MV_D], XEVE_REFP (*)[REFP_NUM], pel (*)[N_C][MAX_CU_DIM], int, pel *, int, int, int);

static int (*h_me)(int, int, int, int, int, int, int, int, const s16 (*)[MV_D], s16 (*)[MV_D], int, int, const pel *, int, int, const s16 *, int, int, u32, int, int, u32 *);
static u32 (*orig_me)(XEVE_PINTER *, int, int, int, int, s8 *, int, s16 (*)[MV_D], s16 (*)[MV_D], int, int, pel *, int, int, int);
static unsigned long long n_me, n_me_hip;
static void report(void)
{
    if(n_me) fprintf(stderr, "[xeve_hip_shim_affine] affine gradient searches %s: %llu (of %llu)\n", h_me ? "served by HIP" : "counted (reference's own function)", h_me ? n_me_hip : n_me, n_me);
    fprintf(stderr, "[xeve_hip_shim_affine] xeve_affine_mc calls %s: %llu (of %llu)\n", hip && !hip_mc_off ? "served by HIP" : "counted (reference's own function)", n_eligible, n_calls);
}
static void bind(void)
{
    static int done;
    if(done) return;
    done = 1;
    orig = dlsym(RTLD_NEXT, "xeve_affine_mc");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim_affine] reference xeve_affine_mc not found\n"); abort(); }
    const char *lib = getenv("XEVE_HIP_LIB");
    if(!lib) {
        if(getenv("XEVE_HIP_SHIM_AFFINE_COUNT")) atexit(report);
        return;
    }
    hip = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if(!hip) { fprintf(stderr, "[xeve_hip_shim_affine] %s\n", dlerror()); abort(); }
    int (*init)(int) = (int (*)(int))dlsym(hip, "xeve_hip_init");
    h_err = (const char *(*)(void))dlsym(hip, "xeve_hip_last_error");
    h_mc = dlsym(hip, "xeve_hip_affine_mc_host");
    if(getenv("XEVE_HIP_SHIM_AFFINE_ME")) {
        h_me = dlsym(hip, "xeve_hip_affine_me_host");
        if(!h_me) { fprintf(stderr, "[xeve_hip_shim_affine] xeve_hip_affine_me_host missing\n"); abort(); }
    }
    if(getenv("XEVE_HIP_SHIM_AFFINE_MC_OFF")) h_mc = NULL, hip_mc_off = 1; /* (the search's route alone: xeve_affine_mc stays the reference's) */
    if(!init || !h_err || (!h_mc && !hip_mc_off)) { fprintf(stderr, "[xeve_hip_shim_affine] entry points missing\n"); abort(); }
    const char *dev = getenv("XEVE_HIP_DEVICE");
    if(init(dev ? atoi(dev) : 0) != 0) { fprintf(stderr, "[xeve_hip_shim_affine] init: %s\n", h_err()); abort(); }
    fprintf(stderr, "[xeve_hip_shim_affine] HIP affine motion compensation bound\n");
    atexit(report);
}

void xeve_affine_mc(int x, int y, int pic_w, int pic_h, int w, int h, s8 refi[REFP_NUM], s16 mv[REFP_NUM][VER_NUM][MV_D], XEVE_REFP (*refp)[REFP_NUM], pel pred[2][N_C][MAX_CU_DIM],
                    int vertex_num, pel *tmp_buffer, int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc)
{
    bind();
    n_calls++;
    /* what the HIP entry covers: 4:2:0, one bit depth, CUs of 8 .. 128 (the reference's affine CUs are >= 8x8: xevem_pinter.c) */
    const int ok = chroma_format_idc == 1 && bit_depth_luma == bit_depth_chroma && w >= 8 && h >= 8 && (refi[0] >= 0 || refi[1] >= 0);
    if(!hip || !ok || hip_mc_off) {
        n_eligible += ok;
        orig(x, y, pic_w, pic_h, w, h, refi, mv, refp, pred, vertex_num, tmp_buffer, bit_depth_luma, bit_depth_chroma, chroma_format_idc);
        /* XEVE_HIP_SHIM_AFFINE_CHECK=<oracle/libxeve_oracle.so>: the C restatement (xo_affine_mc: same layouts as the library's) beside the reference on every call of the live
         * encoder; the first differing call is printed with its arguments (how a clip that differs on the GPU is traced to a gap of the restatement without a GPU) */
        static void *ora;
        static void (*xo)(const hip_refpic *, int, int, int, int, const void *, int, int, int, pel *, pel *, pel *, int *);
        static int  reported;
        const char *chk = getenv("XEVE_HIP_SHIM_AFFINE_CHECK");
        if(chk && ok && !reported) {
            if(!ora) {
                ora = dlopen(chk, RTLD_NOW | RTLD_LOCAL);
                if(ora) xo = dlsym(ora, "xo_affine_mc");
                if(!xo) { fprintf(stderr, "[xeve_hip_shim_affine] check: %s\n", dlerror()); abort(); }
            }
            struct { int x, y; s16 mv[2][3][2]; s8 refi[2]; s8 vertex_num, pad_; } J;
            hip_refpic tab[16];
            int s_l = 0, s_c = 0;
            for(int i = 0; i < 16; i++) tab[i].y = tab[i].u = tab[i].v = NULL, tab[i].poc = tab[i].pad_ = 0;
            J.x = x, J.y = y, J.vertex_num = (s8)vertex_num, J.pad_ = 0;
            for(int l = 0; l < 2; l++) {
                J.refi[l] = refi[l];
                for(int v = 0; v < 3; v++) J.mv[l][v][0] = mv[l][v][0], J.mv[l][v][1] = mv[l][v][1];
                if(refi[l] < 0) continue;
                const XEVE_PIC *p = refp[refi[l]][l].pic;
                tab[refi[l] * 2 + l].y = p->y, tab[refi[l] * 2 + l].u = p->u, tab[refi[l] * 2 + l].v = p->v, s_l = p->s_l, s_c = p->s_c;
            }
            static pel oy[128 * 128], ou[64 * 64], ov[64 * 64];
            int path[3] = {0, 0, 0};
            xo(tab, s_l, s_c, pic_w, pic_h, &J, w, h, bit_depth_luma, oy, ou, ov, path);
            for(int c = 0; c < 3 && !reported; c++) {
                const pel *a = c == 0 ? oy : c == 1 ? ou : ov, *r = pred[0][c];
                const int  n = c ? (w * h) >> 2 : w * h;
                for(int i = 0; i < n; i++)
                    if(a[i] != r[i]) {
                        fprintf(stderr, "[xeve_hip_shim_affine] CHECK: call %llu differs: x %d y %d pic %dx%d cu %dx%d refi %d %d vertex %d comp %d at %d: reference %d oracle %d; "
                                        "path sub %dx%d mem %d; mv0 (%d,%d) (%d,%d) (%d,%d) mv1 (%d,%d) (%d,%d) (%d,%d)\n", n_calls, x, y, pic_w, pic_h, w, h, refi[0], refi[1], vertex_num, c, i, r[i], a[i],
                                path[0], path[1], path[2], mv[0][0][0], mv[0][0][1], mv[0][1][0], mv[0][1][1], mv[0][2][0], mv[0][2][1], mv[1][0][0], mv[1][0][1], mv[1][1][0], mv[1][1][1],
                                mv[1][2][0], mv[1][2][1]);
                        reported = 1;
                        break;
                    }
            }
        }
        return;
    }
    n_eligible++;
    /* the table of the pictures the CU uses, [refi * 2 + list] as the library takes it; strides and padding are the same for every picture of the run */
    hip_refpic tab[2 * 8];
    int        s_l = 0, s_c = 0, pad_l = 0, pad_c = 0, nr[2] = {0, 0};
    for(int i = 0; i < 16; i++) tab[i].y = tab[i].u = tab[i].v = NULL, tab[i].poc = tab[i].pad_ = 0;
    for(int l = 0; l < 2; l++) {
        if(refi[l] < 0) continue;
        if(refi[l] >= 8) { fprintf(stderr, "[xeve_hip_shim_affine] reference index %d\n", refi[l]); abort(); }
        const XEVE_PIC *p = refp[refi[l]][l].pic;
        tab[refi[l] * 2 + l].y = p->y, tab[refi[l] * 2 + l].u = p->u, tab[refi[l] * 2 + l].v = p->v, tab[refi[l] * 2 + l].poc = (int)refp[refi[l]][l].poc;
        s_l = p->s_l, s_c = p->s_c, pad_l = p->pad_l, pad_c = p->pad_c, nr[l] = refi[l] + 1;
    }
    s16 cp[2][3][MV_D]; /* the library's record holds the three control points of a list (the reference's rows have VER_NUM = 4 entries) */
    for(int l = 0; l < 2; l++)
        for(int v = 0; v < 3; v++) cp[l][v][MV_X] = mv[l][v][MV_X], cp[l][v][MV_Y] = mv[l][v][MV_Y];
    if(h_mc(x, y, pic_w, pic_h, w, h, refi, (const s16(*)[3][MV_D])cp, tab, nr[0], nr[1], s_l, s_c, pad_l, pad_c, pred[0][Y_C], pred[0][U_C], pred[0][V_C], vertex_num, bit_depth_luma) != 0) {
        fprintf(stderr, "[xeve_hip_shim_affine] xeve_hip_affine_mc_host: %s\n", h_err());
        abort();
    }
    if(getenv("XEVE_HIP_SHIM_AFFINE_VERIFY")) { /* the reference's own function behind the GPU's on every call: the first differing call is printed, the reference's result is kept */
        static pel gy[128 * 128], gu[64 * 64], gv[64 * 64];
        static int reported;
        const int  n0 = w * h, n1 = n0 >> 2;
        for(int i = 0; i < n0; i++) gy[i] = pred[0][Y_C][i];
        for(int i = 0; i < n1; i++) gu[i] = pred[0][U_C][i], gv[i] = pred[0][V_C][i];
        orig(x, y, pic_w, pic_h, w, h, refi, mv, refp, pred, vertex_num, tmp_buffer, bit_depth_luma, bit_depth_chroma, chroma_format_idc);
        for(int c = 0; c < 3 && !reported; c++) {
            const pel *a = c == 0 ? gy : c == 1 ? gu : gv, *r = pred[0][c];
            for(int i = 0; i < (c ? n1 : n0); i++)
                if(a[i] != r[i]) {
                    fprintf(stderr, "[xeve_hip_shim_affine] VERIFY: call %llu differs: x %d y %d pic %dx%d cu %dx%d refi %d %d vertex %d comp %d at %d: reference %d gpu %d; s_l %d s_c %d pad %d %d; "
                                    "mv0 (%d,%d) (%d,%d) (%d,%d) mv1 (%d,%d) (%d,%d) (%d,%d)\n", n_calls, x, y, pic_w, pic_h, w, h, refi[0], refi[1], vertex_num, c, i, r[i], a[i], s_l, s_c, pad_l, pad_c,
                            mv[0][0][0], mv[0][0][1], mv[0][1][0], mv[0][1][1], mv[0][2][0], mv[0][2][1], mv[1][0][0], mv[1][0][1], mv[1][1][0], mv[1][1][1], mv[1][2][0], mv[1][2][1]);
                    reported = 1;
                    break;
                }
        }
    }
}

/* ---- the affine gradient search: pi->fn_affine_me bound to the HIP host form ---------------------------------------------------------------------------------------------- */
static u32 shim_affine_me(XEVE_PINTER *pi, int x, int y, int log2_cuw, int log2_cuh, s8 *refi, int lidx, s16 mvp[VER_NUM][MV_D], s16 mv[VER_NUM][MV_D], int bi, int vertex_num,
                          pel *tmp, int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc)
{
    n_me++;
    const int cuw = 1 << log2_cuw, cuh = 1 << log2_cuh, ri = *refi;
    /* what the HIP entry covers: CUs of 16 .. 128 (the encoder searches affine vectors for nothing smaller, xevem_pinter.c:5516), reference indices below 8 */
    if(!h_me || cuw < 16 || cuh < 16 || ri < 0 || ri >= 8) return orig_me(pi, x, y, log2_cuw, log2_cuh, refi, lidx, mvp, mv, bi, vertex_num, tmp, bit_depth_luma, bit_depth_chroma, chroma_format_idc);
    n_me_hip++;
    const XEVE_PIC *rp = pi->refp[ri][lidx].pic;
    const s16 *org = bi ? pi->org_bi : pi->o[Y_C] + x + y * pi->s_o[Y_C];
    const int  s_org = bi ? cuw : pi->s_o[Y_C];
    s16 start[VER_NUM][MV_D];
    memcpy(start, mv, sizeof(start));
    u32 cost = 0;
    /* (a row of the reference's arrays is one control point, VER_NUM = 4 of them: the first three are what the library's [3][2] reads) */
    if(h_me(x, y, rp->w_l, rp->h_l, cuw, cuh, ri, lidx, (const s16(*)[MV_D])mvp, mv, bi, vertex_num, rp->y, rp->s_l, rp->pad_l, org, s_org, bit_depth_luma, pi->lambda_mv, pi->num_refp,
            pi->mot_bits[1 - lidx], &cost) != 0) {
        fprintf(stderr, "[xeve_hip_shim_affine] xeve_hip_affine_me_host: %s\n", h_err());
        abort();
    }
    if(getenv("XEVE_HIP_SHIM_AFFINE_VERIFY")) { /* the reference's own search behind the GPU's: the first differing call is printed, the reference's result is kept */
        static int reported;
        s16 got[VER_NUM][MV_D];
        memcpy(got, mv, sizeof(got)), memcpy(mv, start, sizeof(start));
        const u32 want = orig_me(pi, x, y, log2_cuw, log2_cuh, refi, lidx, mvp, mv, bi, vertex_num, tmp, bit_depth_luma, bit_depth_chroma, chroma_format_idc);
        int same = want == cost;
        for(int v = 0; v < vertex_num; v++) same = same && got[v][MV_X] == mv[v][MV_X] && got[v][MV_Y] == mv[v][MV_Y];
        if(!same && !reported) {
            fprintf(stderr, "[xeve_hip_shim_affine] VERIFY: search %llu differs: x %d y %d cu %dx%d refi %d list %d bi %d vertex %d lambda %u num_refp %d: reference %u (%d,%d) (%d,%d) (%d,%d) "
                            "gpu %u (%d,%d) (%d,%d) (%d,%d); start (%d,%d) (%d,%d) (%d,%d)\n", n_me, x, y, cuw, cuh, ri, lidx, bi, vertex_num, pi->lambda_mv, pi->num_refp, want, mv[0][0], mv[0][1],
                    mv[1][0], mv[1][1], mv[2][0], mv[2][1], cost, got[0][0], got[0][1], got[1][0], got[1][1], got[2][0], got[2][1], start[0][0], start[0][1], start[1][0], start[1][1],
                    start[2][0], start[2][1]);
            reported = 1;
        }
        return want;
    }
    return cost;
}
int xevem_pinter_create(XEVE_CTX *ctx, int complexity)
{
    static int (*create)(XEVE_CTX *, int);
    bind();
    if(!create) create = dlsym(RTLD_NEXT, "xevem_pinter_create");
    if(!create) { fprintf(stderr, "[xeve_hip_shim_affine] reference xevem_pinter_create not found\n"); abort(); }
    const int ret = create(ctx, complexity);
    if(ret == XEVE_OK && (h_me || getenv("XEVE_HIP_SHIM_AFFINE_COUNT"))) {
        for(int i = 0; i < ctx->param.threads; i++) {
            if(!orig_me) orig_me = ctx->pinter[i].fn_affine_me;
            ctx->pinter[i].fn_affine_me = shim_affine_me;
        }
        if(h_me) fprintf(stderr, "[xeve_hip_shim_affine] HIP affine gradient search bound (%d inter-prediction objects)\n", ctx->param.threads);
    }
    return ret;
}
