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

#include "xeve_type.h"

static void (*hip_recon_blk)(s16 *, pel *, int, int, int, int, pel *, int);
static unsigned long long (*hip_table_calls)(void);

static void shim_recon(XEVE_CTX *ctx, XEVE_CORE *core, s16 *coef, pel *pred, int is_coef, int cuw, int cuh, int s_rec, pel *rec, int bit_depth)
{
    hip_recon_blk(coef, pred, is_coef, cuw, cuh, s_rec, rec, bit_depth);
}

static void report(void)
{
    if(hip_table_calls) fprintf(stderr, "[xeve_hip_shim] dispatch-table calls served by HIP: %llu\n", hip_table_calls());
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
    atexit(report);
    fprintf(stderr, "[xeve_hip_shim] HIP dispatch tables installed (%d pointers + fn_recon)\n", n);
}
