/* oracle/ref_param_pin.c -- TEST INFRASTRUCTURE ONLY (build container): an LD_PRELOAD interposer on the reference library's xeve_create that sets fields of XEVE_PARAM the
 * reference APPLICATION lists as options but fails to parse (app/xeve_app_args.h: --inter-slice-type, --qp-cb-offset, --qp-cr-offset have no variable bound).  With it the
 * unmodified encoder library can be run with those parameters, and the goldens of tests/golden/make_enc_golden.py can hold the product's frame loop to them.
 *   XEVE_PIN_INTER_SLICE_TYPE = 0 | 1 (B | P), XEVE_PIN_QP_CB_OFFSET, XEVE_PIN_QP_CR_OFFSET; XEVE_PIN_RDO_DBK_SWITCH, XEVE_PIN_ME_SUB, XEVE_PIN_ME_RANGE
 * Built by oracle/Makefile into oracle/_ref/libxeve_param_pin.so against the reference's own header (inc/xeve.h), where it lies. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdlib.h>
#include "xeve.h"

XEVE xeve_create(XEVE_CDSC *cdsc, int *err)
{
    static XEVE (*real)(XEVE_CDSC *, int *);
    if(!real) real = (XEVE(*)(XEVE_CDSC *, int *))dlsym(RTLD_NEXT, "xeve_create");
    const char *e;
    if(cdsc) {
        if((e = getenv("XEVE_PIN_INTER_SLICE_TYPE"))) cdsc->param.inter_slice_type = atoi(e);
        if((e = getenv("XEVE_PIN_QP_CB_OFFSET"))) cdsc->param.qp_cb_offset = atoi(e);
        if((e = getenv("XEVE_PIN_QP_CR_OFFSET"))) cdsc->param.qp_cr_offset = atoi(e);
        /* (taking preset slow apart when a restatement differs: the loop filter's share of the distortions, the search's sub-pel level, its range) */
        if((e = getenv("XEVE_PIN_RDO_DBK_SWITCH"))) cdsc->param.rdo_dbk_switch = atoi(e);
        if((e = getenv("XEVE_PIN_ME_SUB"))) cdsc->param.me_sub = atoi(e);
        if((e = getenv("XEVE_PIN_ME_RANGE"))) cdsc->param.me_range = atoi(e);
    }
    return real(cdsc, err);
}
