/* oracle/ref_shim_main.c -- TEST INFRASTRUCTURE ONLY: LD_PRELOAD interposer for the MAIN-profile reference library (oracle/_ref/libxevem_ref.so).
 *
 * It shows (and tests/test_integration_ref.py runs) the zero-edit binding of the Main-profile slice: after the reference's own
 * xevem_platform_init_func() (src_main/xevem_util.c:3917-3966) has chosen its C / SSE / AVX tables, the shim calls
 * xeve_hip_install_tables_main(), which stores the *_hip tables into the same pointer globals.  Nothing else of the encoder is touched.
 * XEVE_HIP_LIB unset = plain reference run. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

static unsigned long long (*calls)(void), (*calls_main)(void);

static void report(void)
{
    if(calls) fprintf(stderr, "[xeve_hip_shim_main] dispatch-table calls served by HIP: %llu, of them by the Main-profile entries: %llu\n", calls(), calls_main());
}

void xevem_platform_init_func(void)
{
    void (*orig)(void) = (void (*)(void))dlsym(RTLD_NEXT, "xevem_platform_init_func");
    if(!orig) { fprintf(stderr, "[xeve_hip_shim_main] reference xevem_platform_init_func not found\n"); abort(); }
    orig();
    const char *lib = getenv("XEVE_HIP_LIB");
    if(!lib || calls) return;
    void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if(!h) { fprintf(stderr, "[xeve_hip_shim_main] %s\n", dlerror()); abort(); }
    int (*init)(int)         = (int (*)(int))dlsym(h, "xeve_hip_init");
    int (*install)(void *)   = (int (*)(void *))dlsym(h, "xeve_hip_install_tables_main");
    const char *(*err)(void) = (const char *(*)(void))dlsym(h, "xeve_hip_last_error");
    calls = dlsym(h, "xeve_hip_table_calls"), calls_main = dlsym(h, "xeve_hip_table_calls_main");
    if(!init || !install || !err || !calls || !calls_main) { fprintf(stderr, "[xeve_hip_shim_main] entry points missing\n"); abort(); }
    const char *dev = getenv("XEVE_HIP_DEVICE");
    if(init(dev ? atoi(dev) : 0) != 0) { fprintf(stderr, "[xeve_hip_shim_main] init: %s\n", err()); abort(); }
    /* the inverse pass-through slot ctx->fn_itxb is not reachable from here (this hook has no ctx): it stays with the reference, which only uses it with tool_iqt off */
    int n = install(NULL);
    if(n != 18) { fprintf(stderr, "[xeve_hip_shim_main] install: %d (%s)\n", n, err()); abort(); }
    fprintf(stderr, "[xeve_hip_shim_main] HIP dispatch tables installed, Main-profile entries included (%d pointers)\n", n);
    atexit(report);
}
