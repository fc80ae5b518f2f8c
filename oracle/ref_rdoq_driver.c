/*
 * oracle/ref_rdoq_driver.c -- TEST INFRASTRUCTURE (build container only; output oracle/_ref/libref_rdoq.so).
 * Flat wrapper around the reference's xeve_rdoq_run_length_cc (src_base/xeve_tq.c:497-649), which wants an XEVE_CORE
 * with the CABAC-derived estimate tables and an XEVE_CTX with err_scale; both come from the reference's own headers.
 */
#include <stdlib.h>
#include <string.h>
#include "xeve_type.h"

typedef struct { int cbf[2]; int run[24][2], level[24][2], last[2][2]; } drv_est;

int refdrv_rdoq(s16 *coef, int log2w, int log2h, int qp, double lambda, int is_intra, int ch_type, int bit_depth, int tool_iqt, const drv_est *e)
{
    static __thread XEVE_CTX  *ctx;  /* per thread: oracle/cpu_bench.c calls this from its worker threads */
    static __thread XEVE_CORE *core;
    static __thread int        last_iqt = -1, last_bd = -1;
    if(!ctx) ctx = calloc(1, sizeof(*ctx)), core = calloc(1, sizeof(*core));
    if(last_iqt != tool_iqt || last_bd != bit_depth) {
        ctx->param.tool_iqt = tool_iqt, ctx->param.codec_bit_depth = bit_depth;
        ctx->fn_rdoq_set_ctx_cc = xeve_rdoq_set_ctx_cc;
        xeve_init_err_scale(ctx);
        last_iqt = tool_iqt, last_bd = bit_depth;
    }
    core->ctx = ctx;
    /* the caller passes the cbf pair the reference would pick; load it into all four slots */
    for(int b = 0; b < 2; b++) core->rdoq_est_cbf_all[b] = core->rdoq_est_cbf_luma[b] = core->rdoq_est_cbf_cb[b] = core->rdoq_est_cbf_cr[b] = e->cbf[b];
    memcpy(core->rdoq_est_run, e->run, sizeof(e->run));
    memcpy(core->rdoq_est_level, e->level, sizeof(e->level));
    memcpy(core->rdoq_est_last, e->last, sizeof(e->last));
    return xeve_rdoq_run_length_cc((u8)qp, lambda, (u8)is_intra, coef, coef, log2w, log2h, ch_type, core, bit_depth);
}

const u16 *refdrv_scan(int log2w, int log2h) { return xeve_tbl_scan[log2w - 1][log2h - 1]; }
long long refdrv_err_scale(int qp_rem, int log2_size, int bit_depth, int tool_iqt)
{
    static XEVE_CTX *ctx;
    if(!ctx) ctx = calloc(1, sizeof(*ctx));
    ctx->param.tool_iqt = tool_iqt, ctx->param.codec_bit_depth = bit_depth;
    xeve_init_err_scale(ctx);
    return ctx->err_scale[qp_rem][log2_size - 1];
}
