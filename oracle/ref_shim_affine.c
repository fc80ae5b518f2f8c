/* oracle/ref_shim_affine.c -- TEST INFRASTRUCTURE ONLY: LD_PRELOAD interposer for the MAIN-profile reference library (oracle/_ref/libxevem_ref.so) that routes the affine
 * motion compensation of a CU to the GPU.  The Main encoder calls xeve_affine_mc BY NAME -- from the affine merge analysis (src_main/xevem_pinter.c:1947), from every round of
 * the affine gradient search (:4659) and from the affine bi-prediction (:4918) -- so the symbol is interposed and forwards to the HIP host form
 * (xeve_hip_affine_mc_host: the reference's arguments, host planes).  Nothing else of the encoder is touched: the affine SEARCH around it stays the reference's.
 * XEVE_HIP_LIB unset = plain reference run; with XEVE_HIP_SHIM_AFFINE_COUNT=1 the calls are counted and go to the reference's own function (which clips reach affine MC
 * at all).  Needs the reference's headers (XEVE_REFP / XEVE_PIC layouts): built by oracle/Makefile into oracle/_ref/. */
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
static int (*h_mc)(int, int, int, int, int, int, const s8 *, const s16 (*)[3][MV_D], const hip_refpic *, int, int, int, int, int, int, pel *, pel *, pel *, int, int);
static const char *(*h_err)(void);
static unsigned long long n_calls, n_eligible;
static void (*orig)(int, int, int, int, int, int, s8 *, s16 (*)[VER_NUM][MV_D], XEVE_REFP (*)[REFP_NUM], pel (*)[N_C][MAX_CU_DIM], int, pel *, int, int, int);

static void report(void)
{
    fprintf(stderr, "[xeve_hip_shim_affine] xeve_affine_mc calls %s: %llu (of %llu)\n", hip ? "served by HIP" : "counted (reference's own function)", n_eligible, n_calls);
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
    if(!init || !h_err || !h_mc) { fprintf(stderr, "[xeve_hip_shim_affine] entry points missing\n"); abort(); }
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
    if(!hip || !ok) {
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
