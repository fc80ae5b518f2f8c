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

#include "xeve_type.h"

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
static int (*hip_deblock_host)(pel *, pel *, pel *, int, int, int, int, const u32 *, const u32 *, const s8 *, const s16 *, const hip_df_params *);
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
    if(hip_deblock_host(pic->y, pic->u, pic->v, pic->s_l, pic->s_c, pic->pad_l, pic->pad_c, ctx->map_scu, ctx->map_cu_mode, (const s8 *)ctx->map_refi,
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

static void report(void)
{
    if(hip_table_calls) fprintf(stderr, "[xeve_hip_shim] dispatch-table calls served by HIP: %llu\n", hip_table_calls());
    if(df_calls || pad_calls) fprintf(stderr, "[xeve_hip_shim] pictures deblocked on the GPU: %llu, padded on the GPU: %llu\n", df_calls, pad_calls);
}

void xeve_platform_init_func(XEVE_CTX *ctx)
{
    void (*orig)(XEVE_CTX *) = (void (*)(XEVE_CTX *))dlsym(RTLD_NEXT, "xeve_platform_init_func");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim] reference xeve_platform_init_func not found\n"); abort(); }
    orig(ctx);
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
    int n = install(&ctx->fn_itxb);
    if(n != 9) { fprintf(stderr, "[xeve_hip_shim] install: %d (%s)\n", n, err()); abort(); }
    ctx->fn_recon = shim_recon;
    if(getenv("XEVE_HIP_SHIM_DF") && atoi(getenv("XEVE_HIP_SHIM_DF"))) {
        hip_deblock_host = dlsym(h, "xeve_hip_deblock_host"), hip_expand_host = dlsym(h, "xeve_hip_picbuf_expand_host"), hip_err = err;
        if(!hip_deblock_host || !hip_expand_host) { fprintf(stderr, "[xeve_hip_shim] deblock / expand entry points missing\n"); abort(); }
        ctx->fn_loop_filter = shim_loop_filter, ctx->fn_picbuf_expand = shim_pic_expand;
        fprintf(stderr, "[xeve_hip_shim] loop filter and picture padding routed to the GPU\n");
    }
    atexit(report);
    fprintf(stderr, "[xeve_hip_shim] HIP dispatch tables installed (%d pointers + fn_recon)\n", n);
}
