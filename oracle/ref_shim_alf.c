/* oracle/ref_shim_alf.c -- TEST INFRASTRUCTURE ONLY: LD_PRELOAD interposer for the MAIN-profile reference library (oracle/_ref/libxevem_ref.so) that binds the adaptive loop
 * filter's sample kernels as INTEGRATION.md describes: after the reference's own alf_init (src_main/xevem_alf.c:38-53) has set the ADAPTIVE_LOOP_FILTER object's three
 * function pointers, the shim stores the HIP host forms there (xeve_hip_alf_filter_blk_7_host / _5_host / _derive_classification_blk_host); alf_derive_classification
 * calls alf_derive_classification_blk BY NAME (:476) and xeve_alf_derive_stats_filtering calls xeve_alf_get_blk_stats by name (:3810), so those two symbols are interposed
 * as well and forward to the HIP entries.  Nothing else of the encoder is touched: the filter derivation and the CTU loops stay the reference's.  XEVE_HIP_LIB unset = plain reference run; with XEVE_HIP_SHIM_ALF_COUNT=1
 * instead, the calls are counted and go to the reference's own functions (which clips reach the filters at all).  Needs the reference's headers (the
 * object's layout): built by oracle/Makefile into oracle/_ref/. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include "xevem_type.h"
#include "xevem_alf.h"

static void *hip;
static void (*h_cls)(ALF_CLASSIFIER **, const pel *, const int, const AREA *, const int, int);
static void (*h_f7)(ALF_CLASSIFIER **, pel *, const int, const pel *, const int, const AREA *, const u8, short *, const CLIP_RANGE *);
static void (*h_f5)(ALF_CLASSIFIER **, pel *, const int, const pel *, const int, const AREA *, const u8, short *, const CLIP_RANGE *);
static void (*h_stats)(int, ALF_COVARIANCE *, ALF_CLASSIFIER **, const pel *, int, const pel *, int, int, int, int, int);
static unsigned long long n_cls, n_f7, n_f5, n_stats;
static int counting;
static void (*o_f7)(ALF_CLASSIFIER **, pel *, const int, const pel *, const int, const AREA *, const u8, short *, const CLIP_RANGE *);
static void (*o_f5)(ALF_CLASSIFIER **, pel *, const int, const pel *, const int, const AREA *, const u8, short *, const CLIP_RANGE *);

static void report(void)
{
    fprintf(stderr, "[xeve_hip_shim_alf] ALF calls %s: classification %llu, 7x7 filter %llu, 5x5 filter %llu, statistics %llu\n", counting ? "counted (reference's own functions)" : "served by HIP",
            n_cls, n_f7, n_f5, n_stats);
}
static void bind(void)
{
    const char *lib = getenv("XEVE_HIP_LIB");
    if(hip || !lib) return;
    hip = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if(!hip) { fprintf(stderr, "[xeve_hip_shim_alf] %s\n", dlerror()); abort(); }
    int (*init)(int)         = (int (*)(int))dlsym(hip, "xeve_hip_init");
    const char *(*err)(void) = (const char *(*)(void))dlsym(hip, "xeve_hip_last_error");
    h_cls = dlsym(hip, "xeve_hip_alf_derive_classification_blk_host"), h_f7 = dlsym(hip, "xeve_hip_alf_filter_blk_7_host"), h_f5 = dlsym(hip, "xeve_hip_alf_filter_blk_5_host");
    h_stats = dlsym(hip, "xeve_hip_alf_get_blk_stats_host");
    if(!init || !err || !h_cls || !h_f7 || !h_f5 || !h_stats) { fprintf(stderr, "[xeve_hip_shim_alf] entry points missing\n"); abort(); }
    const char *dev = getenv("XEVE_HIP_DEVICE");
    if(init(dev ? atoi(dev) : 0) != 0) { fprintf(stderr, "[xeve_hip_shim_alf] init: %s\n", err()); abort(); }
    fprintf(stderr, "[xeve_hip_shim_alf] HIP adaptive-loop-filter kernels bound\n");
    atexit(report);
}
static void f7(ALF_CLASSIFIER **c, pel *d, const int sd, const pel *s, const int ss, const AREA *b, const u8 comp, short *set, const CLIP_RANGE *cr)
{
    n_f7++;
    (counting ? o_f7 : h_f7)(c, d, sd, s, ss, b, comp, set, cr);
}
static void f5(ALF_CLASSIFIER **c, pel *d, const int sd, const pel *s, const int ss, const AREA *b, const u8 comp, short *set, const CLIP_RANGE *cr)
{
    n_f5++;
    (counting ? o_f5 : h_f5)(c, d, sd, s, ss, b, comp, set, cr);
}
void alf_derive_classification_blk(ALF_CLASSIFIER **classifier, const pel *src_luma, const int src_stride, const AREA *blk, const int shift, int bit_depth)
{
    bind();
    if(!hip) {
        void (*orig)(ALF_CLASSIFIER **, const pel *, const int, const AREA *, const int, int) = dlsym(RTLD_NEXT, "alf_derive_classification_blk");
        n_cls += counting;
        orig(classifier, src_luma, src_stride, blk, shift, bit_depth);
        return;
    }
    n_cls++;
    h_cls(classifier, src_luma, src_stride, blk, shift, bit_depth);
}
/* xeve_alf_derive_stats_filtering calls this one by name as well (:3810) */
void xeve_alf_get_blk_stats(int ch, ALF_COVARIANCE *alf_cov, const ALF_FILTER_SHAPE *shape, ALF_CLASSIFIER **classifier, pel *org, const int org_stride, pel *rec,
                            const int rec_stride, const int x, const int y, const int width, const int height)
{
    bind();
    if(!hip) {
        void (*orig)(int, ALF_COVARIANCE *, const ALF_FILTER_SHAPE *, ALF_CLASSIFIER **, pel *, const int, pel *, const int, const int, const int, const int, const int) =
            dlsym(RTLD_NEXT, "xeve_alf_get_blk_stats");
        n_stats += counting;
        orig(ch, alf_cov, shape, classifier, org, org_stride, rec, rec_stride, x, y, width, height);
        return;
    }
    n_stats++;
    (void)ch; /* (only scales the classifier's coordinates, and chroma has no classifier, :3857-3863) */
    h_stats(shape->filterLength, alf_cov, classifier, org, org_stride, rec, rec_stride, x, y, width, height);
}
void alf_init(ADAPTIVE_LOOP_FILTER *alf, int bit_depth)
{
    void (*orig)(ADAPTIVE_LOOP_FILTER *, int) = (void (*)(ADAPTIVE_LOOP_FILTER *, int))dlsym(RTLD_NEXT, "alf_init");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim_alf] reference alf_init not found\n"); abort(); }
    orig(alf, bit_depth);
    bind();
    if(!hip && getenv("XEVE_HIP_SHIM_ALF_COUNT")) {
        if(!counting) atexit(report);
        counting = 1, o_f7 = alf->filter_7x7_blk, o_f5 = alf->filter_5x5_blk, alf->filter_7x7_blk = f7, alf->filter_5x5_blk = f5;
        return;
    }
    if(!hip) return;
    alf->derive_classification_blk = alf_derive_classification_blk;
    alf->filter_7x7_blk            = f7;
    alf->filter_5x5_blk            = f5;
}
