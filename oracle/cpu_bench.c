/*
 * oracle/cpu_bench.c -- TEST/BENCH INFRASTRUCTURE: the CPU baseline leg of bench.py.
 *
 * Times the SAME hot-path pass that xeve_amd/workload.py runs on the GPU (steps A..E there), on the
 * host cores, through either
 *   kind "reference": the reference's own best dispatch tables (AVX2 / SSE4.1, exactly the set
 *                     xeve_platform_init_func installs -- src_base/xeve_enc.c:745-753) taken from the
 *                     in-place build oracle/_ref/libxeveb_ref.so via dlopen; the two functions the
 *                     reference keeps static (plain quant, dequant: xeve_tq.c:704-727, xeve_itdq.c:442-452)
 *                     come from the oracle restatement;
 *   kind "port"     : the oracle restatement only (when oracle/_ref is absent).
 *
 * usage: cpu_bench <path/to/libxeveb_ref.so | port> <width> <height> <threads> [frac_percent [reps]]
 *   frac_percent: process only the first N % of the blocks of every level (bounded sample), default 100;
 *   reps: repeat the pass that many times (to get a stable >= 10 s sample on many-core hosts), default 1.
 * prints one JSON object on stdout.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "xeve_oracle.h"

typedef int16_t pel;
typedef int (*FN_SAD)(int, int, void *, void *, int, int, int);
typedef int64_t (*FN_SSD)(int, int, void *, void *, int, int, int);
typedef void (*FN_DIFF)(int, int, void *, void *, int, int, int, int16_t *, int);
typedef void (*FN_MC_L)(pel *, int, int, int, int, pel *, int, int, int, const int16_t (*)[8]);
typedef void (*FN_MC_C)(pel *, int, int, int, int, pel *, int, int, int, const int16_t (*)[4]);
typedef void (*FN_AVG)(int16_t *, int16_t *, int16_t *, int, int, int, int, int);
typedef void (*FN_TXB)(void *, void *, int, int, int);
typedef void (*FN_RECON)(int16_t *, pel *, int, int, int, int, pel *, int);

static struct {
    FN_SAD  (*sad)[8];
    FN_SSD  (*ssd)[8];
    FN_DIFF (*diff)[8];
    FN_SAD   *satd;
    FN_MC_L (*mc_l)[2];
    FN_MC_C (*mc_c)[2];
    FN_AVG    avg;
    FN_TXB   *txb, *itxb;
    FN_RECON  recon;
    int (*rdoq)(int16_t *, int, int, int, double, int, int, int, int, const xo_rdoq_est *); /* reference driver or NULL */
    int       is_ref;
} T;

/* port-mode adapters with the table signatures */
static int     p_sad(int w, int h, void *a, void *b, int s1, int s2, int bd) { return xo_sad(w, h, a, b, s1, s2, bd); }
static int64_t p_ssd(int w, int h, void *a, void *b, int s1, int s2, int bd) { return xo_ssd(w, h, a, b, s1, s2, bd); }
static int     p_satd(int w, int h, void *a, void *b, int s1, int s2, int bd) { return xo_satd(w, h, a, b, s1, s2, bd); }
static void    p_diff(int w, int h, void *a, void *b, int s1, int s2, int sd, int16_t *d, int bd) { (void)bd; xo_diff(w, h, a, b, s1, s2, sd, d); }
#define P_MC(name, fx, fy)                                                                                                   \
    static void p_mc_l_##name(pel *r, int gx, int gy, int sr, int sp, pel *p, int w, int h, int bd, const int16_t (*c)[8]) { xo_mc_l(fx, fy, r, gx, gy, sr, sp, p, w, h, bd, c); } \
    static void p_mc_c_##name(pel *r, int gx, int gy, int sr, int sp, pel *p, int w, int h, int bd, const int16_t (*c)[4]) { xo_mc_c(fx, fy, r, gx, gy, sr, sp, p, w, h, bd, c); }
P_MC(00, 0, 0) P_MC(0n, 0, 1) P_MC(n0, 1, 0) P_MC(nn, 1, 1)
static void p_avg(int16_t *a, int16_t *b, int16_t *d, int sa, int sb, int sd, int w, int h) { xo_avg(a, b, d, sa, sb, sd, w, h); }
#define P_TX(n) \
    static void p_tx##n(void *s, void *d, int sh, int l, int st) { xo_tx(n, s, d, sh, l, st); } \
    static void p_itx##n(void *s, void *d, int sh, int l, int st) { xo_itx(n, s, d, sh, l, st); }
P_TX(1) P_TX(2) P_TX(3) P_TX(4) P_TX(5) P_TX(6)
static void p_recon(int16_t *c, pel *p, int ic, int w, int h, int s, pel *r, int bd) { xo_recon(c, p, ic, w, h, s, r, bd); }

static FN_SAD  port_sad[8][8], port_satd[1];
static FN_SSD  port_ssd[8][8];
static FN_DIFF port_diff[8][8];
static FN_MC_L port_mc_l[2][2] = {{p_mc_l_00, p_mc_l_0n}, {p_mc_l_n0, p_mc_l_nn}};
static FN_MC_C port_mc_c[2][2] = {{p_mc_c_00, p_mc_c_0n}, {p_mc_c_n0, p_mc_c_nn}};
static FN_TXB  port_txb[6] = {p_tx1, p_tx2, p_tx3, p_tx4, p_tx5, p_tx6}, port_itxb[6] = {p_itx1, p_itx2, p_itx3, p_itx4, p_itx5, p_itx6};

static void *need(void *h, const char *n)
{
    void *p = dlsym(h, n);
    if(!p) { fprintf(stderr, "cpu_bench: symbol %s missing\n", n); exit(2); }
    return p;
}

static void bind(const char *path)
{
    if(strcmp(path, "port") == 0) {
        for(int i = 0; i < 8; i++) for(int j = 0; j < 8; j++) port_sad[i][j] = p_sad, port_ssd[i][j] = p_ssd, port_diff[i][j] = p_diff;
        port_satd[0] = p_satd;
        T.sad = port_sad, T.ssd = port_ssd, T.diff = port_diff, T.satd = port_satd, T.mc_l = port_mc_l, T.mc_c = port_mc_c;
        T.avg = p_avg, T.txb = port_txb, T.itxb = port_itxb, T.recon = p_recon, T.is_ref = 0;
        return;
    }
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if(!h) { fprintf(stderr, "cpu_bench: %s\n", dlerror()); exit(2); }
    int avx2 = __builtin_cpu_supports("avx2");
    T.sad  = need(h, avx2 ? "xeve_tbl_sad_16b_avx" : "xeve_tbl_sad_16b_sse");
    T.ssd  = need(h, "xeve_tbl_ssd_16b_sse");
    T.diff = need(h, "xeve_tbl_diff_16b_sse");
    T.satd = need(h, "xeve_tbl_satd_16b_sse");
    T.mc_l = need(h, avx2 ? "xeve_tbl_mc_l_avx" : "xeve_tbl_mc_l_sse");
    T.mc_c = need(h, avx2 ? "xeve_tbl_mc_c_avx" : "xeve_tbl_mc_c_sse");
    T.avg  = need(h, "xeve_average_16b_no_clip_sse");
    T.txb  = need(h, avx2 ? "xeve_tbl_txb_avx" : "xeve_tbl_txb");
    T.itxb = need(h, avx2 ? "xeve_tbl_itxb_avx" : "xeve_tbl_itxb_sse");
    T.recon = need(h, "xeve_recon_blk");
    T.is_ref = 1;
    { /* the reference's own xeve_rdoq_run_length_cc through oracle/_ref/libref_rdoq.so (next to libxeveb_ref.so) */
        char buf[4096];
        snprintf(buf, sizeof(buf), "%s", path);
        char *slash = strrchr(buf, '/');
        snprintf(slash ? slash + 1 : buf, sizeof(buf) - (slash ? (size_t)(slash + 1 - buf) : 0), "libref_rdoq.so");
        void *hr = dlopen(buf, RTLD_NOW | RTLD_LOCAL);
        T.rdoq = hr ? (int (*)(int16_t *, int, int, int, double, int, int, int, int, const xo_rdoq_est *))dlsym(hr, "refdrv_rdoq") : NULL;
    }
}

/* ---- workload (mirrors xeve_amd/workload.py) ------------------------------------------------------- */
#define PAD_L 144
#define PAD_C 72
#define N_LIST 2
#define N_PASS 3
#define N_MERGE 3
#define MV_RANGE 48
static int   W, H, s_l, s_c, BD = 10, QP = 32, FRAC = 100, REPS = 1;
static pel  *org[3], *ref[N_LIST][3], *rec[3];
static int   pat[128][2], npat;

static void make_pattern(void)
{
    npat = 0;
    for(int dy = -2; dy <= 2; dy++) for(int dx = -2; dx <= 2; dx++) pat[npat][0] = dx, pat[npat++][1] = dy;
    int steps[5] = {4, 8, 16, 32, 64};
    for(int k = 0; k < 5; k++) {
        int st = steps[k], n = st == 4 ? 4 : (st == 8 ? 8 : 16), q = n / 4;
        for(int i = 0; i < n; i++) {
            int a = i % q, b = i / q, dx, dy;
            if(b == 0) dx = a, dy = -(q - a); else if(b == 1) dx = q - a, dy = a; else if(b == 2) dx = -a, dy = q - a; else dx = -(q - a), dy = -a;
            pat[npat][0] = dx * st / q, pat[npat++][1] = dy * st / q;
        }
        pat[npat][0] = 0, pat[npat++][1] = 0;
    }
}

static uint64_t rng_state = 88172645463325252ULL;
static inline uint32_t rnd(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return (uint32_t)(*s >> 16); }
static inline int rmv(uint64_t *s, int r) { return (int)(rnd(s) % (2 * r + 1)) - r; }

static pel *plane(int h, int s)
{
    pel *p = malloc(sizeof(pel) * (size_t)h * s);
    for(size_t i = 0; i < (size_t)h * s; i++) p[i] = (pel)(rnd(&rng_state) & 1023);
    return p;
}

typedef struct { int tid, nthr; int64_t sad_calls; int64_t sink; } Arg;

static xo_rdoq_est EST;
static double      LAMBDA;

static void tq_chain(int16_t *coef, int lg, int is_luma, int64_t *sink)
{
    int32_t tb[64 * 64];
    const int qs = xo_quant_scale[0][QP % 6], dqs = xo_dq_scale[QP % 6] << (QP / 6);
    T.txb[lg - 1](coef, tb, 0, 1 << lg, 0);
    T.txb[lg - 1](tb, coef, (lg - 1 + BD - 8) + (lg + 6), 1 << lg, 1);
    /* xeve_quant_nnz with rdoq = 1 (preset medium): zero pre-test, then RDOQ */
    if(xo_rdoq_zero_test(coef, lg, lg, QP, qs, 0, BD))
        *sink += T.rdoq ? T.rdoq(coef, lg, lg, QP, LAMBDA, 0, is_luma ? 0 : 1, BD, 0, &EST) : xo_rdoq(coef, lg, lg, QP, LAMBDA, is_luma, BD, 0, &EST);
    else memset(coef, 0, sizeof(int16_t) << (2 * lg));
    xo_dquant(coef, lg, lg, dqs, BD);
    T.itxb[lg - 1](coef, tb, 0, 1 << lg, 0);
    T.itxb[lg - 1](tb, coef, 7 + 12 - (BD - 8), 1 << lg, 1);
}

static void *worker(void *vp)
{
    Arg *A = vp;
    uint64_t rs = 0x9E3779B97F4A7C15ULL * (A->tid + 1);
    static const int hp[4][2] = {{-8, 0}, {-8, 8}, {0, 8}, {8, 8}};
    pel *pl[2], *pc[4];
    int16_t *resi, *coef;
    pl[0] = malloc(2 * 64 * 64), pl[1] = malloc(2 * 64 * 64);
    for(int i = 0; i < 4; i++) pc[i] = malloc(2 * 32 * 32);
    resi = malloc(2 * 64 * 64), coef = malloc(2 * 64 * 64);
    for(int rep = 0; rep < REPS; rep++)
    for(int lg = 3; lg <= 6; lg++) {
        const int S = 1 << lg, Sc = S / 2, nx = W / S, ny = H / S, n = (int)((int64_t)nx * ny * FRAC / 100);
        for(int b = A->tid; b < n; b += A->nthr) {
            const int x = (b % nx) * S, y = (b / nx) * S;
            pel *o = org[0] + (PAD_L + y) * s_l + PAD_L + x;
            /* A. integer search */
            for(int i = 0; i < N_LIST * N_PASS; i++) {
                pel *c = ref[i % N_LIST][0] + (PAD_L + y + rmv(&rs, MV_RANGE)) * s_l + PAD_L + x + rmv(&rs, MV_RANGE);
                for(int k = 0; k < npat; k++) A->sink += T.sad[lg][lg](S, S, o, c + pat[k][1] * s_l + pat[k][0], s_l, s_l, BD);
                A->sad_calls += npat;
            }
            /* B. half-pel */
            for(int l = 0; l < N_LIST; l++) {
                int mvx = rmv(&rs, MV_RANGE), mvy = rmv(&rs, MV_RANGE);
                for(int k = 0; k < 4; k++) {
                    int gx = (PAD_L + x + mvx) * 16 + hp[k][0], gy = (PAD_L + y + mvy) * 16 + hp[k][1];
                    T.mc_l[(gx & 15) != 0][(gy & 15) != 0](ref[l][0], gx, gy, s_l, S, pl[0], S, S, BD, xo_mc_l_coeff);
                    A->sink += T.sad[lg][lg](S, S, o, pl[0], s_l, S, BD);
                    A->sad_calls++;
                }
            }
            /* C. merge candidates */
            for(int m = 0; m < N_MERGE; m++) {
                int mvx = rmv(&rs, MV_RANGE * 4) * 4, mvy = rmv(&rs, MV_RANGE * 4) * 4;
                int gx = (PAD_L + x) * 16 + mvx, gy = (PAD_L + y) * 16 + mvy;
                T.mc_l[(gx & 15) != 0][(gy & 15) != 0](ref[0][0], gx, gy, s_l, S, pl[0], S, S, BD, xo_mc_l_coeff);
                A->sink += T.ssd[lg][lg](S, S, o, pl[0], s_l, S, BD);
                int cx = (PAD_C + x / 2) * 32 + mvx, cy = (PAD_C + y / 2) * 32 + mvy;
                for(int c = 1; c <= 2; c++) {
                    T.mc_c[(cx & 31) != 0][(cy & 31) != 0](ref[0][c], cx, cy, s_c, Sc, pc[c - 1], Sc, Sc, BD, xo_mc_c_coeff);
                    A->sink += T.ssd[lg - 1][lg - 1](Sc, Sc, org[c] + (PAD_C + y / 2) * s_c + PAD_C + x / 2, pc[c - 1], s_c, Sc, BD);
                }
            }
            /* D. bi-predicted winner: MC x2, average, DIFF, TQ, ITDQ, recon, SSD */
            for(int l = 0; l < N_LIST; l++) {
                int mvx = rmv(&rs, MV_RANGE * 4) * 4, mvy = rmv(&rs, MV_RANGE * 4) * 4;
                int gx = (PAD_L + x) * 16 + mvx, gy = (PAD_L + y) * 16 + mvy;
                T.mc_l[(gx & 15) != 0][(gy & 15) != 0](ref[l][0], gx, gy, s_l, S, pl[l], S, S, BD, xo_mc_l_coeff);
                int cx = (PAD_C + x / 2) * 32 + mvx, cy = (PAD_C + y / 2) * 32 + mvy;
                for(int c = 1; c <= 2; c++)
                    T.mc_c[(cx & 31) != 0][(cy & 31) != 0](ref[l][c], cx, cy, s_c, Sc, pc[2 * l + c - 1], Sc, Sc, BD, xo_mc_c_coeff);
            }
            T.avg(pl[0], pl[1], pl[0], S, S, S, S, S);
            for(int c = 1; c <= 2; c++) T.avg(pc[c - 1], pc[2 + c - 1], pc[c - 1], Sc, Sc, Sc, Sc, Sc);
            for(int c = 0; c < 3; c++) {
                const int w = c ? Sc : S, l2 = c ? lg - 1 : lg, st = c ? s_c : s_l;
                pel *oc = c ? org[c] + (PAD_C + y / 2) * s_c + PAD_C + x / 2 : o;
                pel *rc = c ? rec[c] + (PAD_C + y / 2) * s_c + PAD_C + x / 2 : rec[0] + (PAD_L + y) * s_l + PAD_L + x;
                pel *pr = c ? pc[c - 1] : pl[0];
                T.diff[l2][l2](w, w, oc, pr, st, w, w, resi, BD);
                A->sink += T.ssd[l2][l2](w, w, oc, pr, st, w, BD);
                memcpy(coef, resi, sizeof(int16_t) * w * w);
                tq_chain(coef, l2, c == 0, &A->sink);
                T.recon(coef, pr, 1, w, w, st, rc, BD);
                A->sink += T.ssd[l2][l2](w, w, oc, rc, st, st, BD);
            }
            /* E. intra gate */
            A->sink += T.satd[0](S, S, o, pl[0], s_l, S, BD);
        }
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if(argc < 5) { fprintf(stderr, "usage: %s <libxeveb_ref.so|port> W H threads [frac%%]\n", argv[0]); return 2; }
    bind(argv[1]);
    W = atoi(argv[2]), H = atoi(argv[3]);
    int nthr = atoi(argv[4]);
    if(argc > 5) FRAC = atoi(argv[5]);
    if(argc > 6) REPS = atoi(argv[6]);
    if(REPS < 1) REPS = 1;
    if(nthr < 1) nthr = 1;
    s_l = W + 2 * PAD_L, s_c = W / 2 + 2 * PAD_C;
    make_pattern();
    /* same RDOQ inputs as xeve_amd/workload.py: lambda for qp 32, one-bit estimates */
    LAMBDA = 0.57 * pow(2.0, (QP - 12) / 3.0);
    EST.cbf[0] = EST.cbf[1] = 32768;
    for(int i = 0; i < 24; i++) EST.run[i][0] = EST.run[i][1] = EST.level[i][0] = EST.level[i][1] = 32768;
    for(int i = 0; i < 2; i++) EST.last[i][0] = EST.last[i][1] = 32768;
    for(int c = 0; c < 3; c++) {
        int h = c ? H / 2 + 2 * PAD_C : H + 2 * PAD_L, s = c ? s_c : s_l;
        org[c] = plane(h, s), rec[c] = plane(h, s);
        for(int l = 0; l < N_LIST; l++) ref[l][c] = plane(h, s);
    }
    pthread_t *th = malloc(sizeof(*th) * nthr);
    Arg *args = calloc(nthr, sizeof(*args));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for(int i = 0; i < nthr; i++) { args[i].tid = i, args[i].nthr = nthr; pthread_create(&th[i], NULL, worker, &args[i]); }
    int64_t calls = 0, sink = 0;
    for(int i = 0; i < nthr; i++) { pthread_join(th[i], NULL); calls += args[i].sad_calls, sink += args[i].sink; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    printf("{\"seconds\": %.6f, \"kind\": \"%s\", \"threads\": %d, \"width\": %d, \"height\": %d, \"frac_percent\": %d, \"reps\": %d, "
           "\"sad_calls\": %lld, \"checksum\": %lld}\n",
           sec, T.is_ref ? "reference" : "port", nthr, W, H, FRAC, REPS, (long long)calls, (long long)sink);
    return 0;
}
