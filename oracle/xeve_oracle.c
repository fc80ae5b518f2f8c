/*
 * oracle/xeve_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the XEVE hot-path arithmetic; see xeve_oracle.h for the
 * scope, the usage restriction and the parity status (PINNED against the
 * unmodified reference compiled in place, oracle/_ref).
 *
 * The code below is written from the arithmetic definitions, not transcribed:
 * distortions are straight double loops, the Hadamard is a generic separable
 * Walsh-Hadamard butterfly, the DCT matrices are generated from the closed form
 * of the EVC integer DCT-II, and the 1-D transforms are plain 64-bit matrix
 * products (the reference's partial butterflies are an exact integer
 * factorisation of the same product, so results are identical).
 */
#include "xeve_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* helpers                                                                   */
/* ------------------------------------------------------------------------- */
static inline int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------------- */
/* a1  SAD   (reference: xeve_sad.c:40-61; abs macro xeve_util.h:55)          */
/* ------------------------------------------------------------------------- */
int xo_sad(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int bit_depth)
{
    int acc = 0;
    for(int y = 0; y < h; y++) {
        for(int x = 0; x < w; x++) {
            /* The reference uses the 16-bit sign-mask abs on an int difference
             * ((d ^ (d>>15)) - (d>>15)); it equals |d| whenever |d| < 32768,
             * i.e. on the whole codec domain.  Restated literally so that the
             * oracle agrees with the reference C path on ANY int16 input. */
            int d = (int)s1[y * st1 + x] - (int)s2[y * st2 + x];
            int m = d >> 15;
            acc += (d ^ m) - m;
        }
    }
    return acc >> (bit_depth - 8);
}

/* ------------------------------------------------------------------------- */
/* a2  SSD   (reference: xeve_sad.c:275-297) -- shift applied PER PIXEL       */
/* ------------------------------------------------------------------------- */
int64_t xo_ssd(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int bit_depth)
{
    const int sh  = (bit_depth - 8) * 2;
    int64_t   acc = 0;
    for(int y = 0; y < h; y++) {
        for(int x = 0; x < w; x++) {
            int d = (int)s1[y * st1 + x] - (int)s2[y * st2 + x];
            acc += (d * d) >> sh;
        }
    }
    return acc;
}

/* ------------------------------------------------------------------------- */
/* a3  DIFF  (reference: xeve_sad.c:160-178)                                  */
/* ------------------------------------------------------------------------- */
void xo_diff(int w, int h, const xo_pel *s1, const xo_pel *s2, int st1, int st2, int st_diff, int16_t *diff)
{
    for(int y = 0; y < h; y++)
        for(int x = 0; x < w; x++)
            diff[y * st_diff + x] = (int16_t)((int)s1[y * st1 + x] - (int)s2[y * st2 + x]);
}

/* ------------------------------------------------------------------------- */
/* a4  SATD  (reference: xeve_sad.c:394-1140)                                 */
/* ------------------------------------------------------------------------- */
/* In-place unnormalised Walsh-Hadamard transform of n values with stride st. */
static void wht(int *v, int n, int st)
{
    for(int len = 1; len < n; len <<= 1) {
        for(int base = 0; base < n; base += 2 * len) {
            for(int i = base; i < base + len; i++) {
                int a = v[i * st], b = v[(i + len) * st];
                v[i * st]         = a + b;
                v[(i + len) * st] = a - b;
            }
        }
    }
}

/* Sum of |2-D Hadamard coefficients| of one tw x th tile of (org - cur), with
 * the DC term taken >> 2 (reference: xeve_sad.c:411,501,592,740,...).        */
static int had_tile_sum(const xo_pel *org, const xo_pel *cur, int s_org, int s_cur, int tw, int th)
{
    int t[16 * 16];
    for(int y = 0; y < th; y++)
        for(int x = 0; x < tw; x++)
            t[y * tw + x] = (int)org[y * s_org + x] - (int)cur[y * s_cur + x];
    for(int y = 0; y < th; y++) wht(t + y * tw, tw, 1);
    for(int x = 0; x < tw; x++) wht(t + x, th, tw);
    int sum = iabs(t[0]) >> 2;
    for(int i = 1; i < tw * th; i++) sum += iabs(t[i]);
    return sum;
}

static int had_tile(const xo_pel *org, const xo_pel *cur, int s_org, int s_cur, int tw, int th)
{
    int s = had_tile_sum(org, cur, s_org, s_cur, tw, th);
    if(tw == 2 && th == 2) return s;                                   /* xeve_sad.c:394-416  */
    if(tw == 4 && th == 4) return (s + 1) >> 1;                        /* xeve_sad.c:507      */
    if(tw == 8 && th == 8) return (s + 2) >> 2;                        /* xeve_sad.c:602      */
    if((tw == 16 && th == 8) || (tw == 8 && th == 16))                 /* xeve_sad.c:748,885  */
        return (int)(s / (2.0 * sqrt(8.0)));
    /* 8x4 and 4x8: xeve_sad.c:964,1038 */
    return (int)(s / sqrt(8.0));
}

int xo_satd(int w, int h, const xo_pel *org, const xo_pel *cur, int s_org, int s_cur, int bit_depth)
{
    int tw, th;
    /* tile selection, same precedence as xeve_had (xeve_sad.c:1051-1135) */
    if(w > h && (h & 7) == 0 && (w & 15) == 0)      { tw = 16; th = 8; }
    else if(w < h && (w & 7) == 0 && (h & 15) == 0) { tw = 8;  th = 16; }
    else if(w > h && (h & 3) == 0 && (w & 7) == 0)  { tw = 8;  th = 4; }
    else if(w < h && (w & 3) == 0 && (h & 7) == 0)  { tw = 4;  th = 8; }
    else if((w % 8 == 0) && (h % 8 == 0))           { tw = 8;  th = 8; }
    else if((w % 4 == 0) && (h % 4 == 0))           { tw = 4;  th = 4; }
    else if((w % 2 == 0) && (h % 2 == 0))           { tw = 2;  th = 2; }
    else abort();
    int sum = 0;
    for(int y = 0; y < h; y += th)
        for(int x = 0; x < w; x += tw)
            sum += had_tile(org + y * s_org + x, cur + y * s_cur + x, s_org, s_cur, tw, th);
    return sum >> (bit_depth - 8);
}

/* ------------------------------------------------------------------------- */
/* a5/a6  motion compensation  (reference: xeve_mc.c:39-381, xeve_mc.h:36-62)  */
/* ------------------------------------------------------------------------- */
/* Baseline luma filter: only the quarter-pel rows 0,4,8,12 of the 1/16-pel
 * table are populated (xeve_mc.c:39-57). */
const int16_t xo_mc_l_coeff[16][8] = {
    [0]  = {0, 0, 0, 64, 0, 0, 0, 0},
    [4]  = {0, 1, -5, 52, 20, -5, 1, 0},
    [8]  = {0, 2, -10, 40, 40, -10, 2, 0},
    [12] = {0, 1, -5, 20, 52, -5, 1, 0},
};
/* Baseline chroma filter: 1/8-pel rows 0,4,...,28 of the 1/32-pel table
 * (xeve_mc.c:59-93). */
const int16_t xo_mc_c_coeff[32][4] = {
    [0]  = {0, 64, 0, 0},   [4]  = {-2, 58, 10, -2}, [8]  = {-4, 52, 20, -4}, [12] = {-6, 46, 30, -6},
    [16] = {-8, 40, 40, -8}, [20] = {-6, 30, 46, -6}, [24] = {-4, 20, 52, -4}, [28] = {-2, 10, 58, -2},
};

/* Generic separable interpolation with `taps` taps (8 luma / 4 chroma) and
 * fractional-position mask fmask (15 luma / 31 chroma), fshift (4 / 5). */
static void mc_generic(int taps, int fshift, int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y,
                       int s_ref, int s_pred, xo_pel *pred, int w, int h, int bit_depth, const int16_t *coef)
{
    const int fmask = (1 << fshift) - 1;
    const int back  = taps / 2 - 1; /* 3 for luma, 1 for chroma */
    const int maxv  = (1 << bit_depth) - 1;
    const int ix = gmv_x >> fshift, iy = gmv_y >> fshift;
    const int16_t *cx = coef + (gmv_x & fmask) * taps;
    const int16_t *cy = coef + (gmv_y & fmask) * taps;

    if(!frac_x && !frac_y) { /* _00: row copies (xeve_mc.c:99-121, 259-280) */
        for(int y = 0; y < h; y++)
            memcpy(pred + y * s_pred, ref + (iy + y) * s_ref + ix, sizeof(xo_pel) * w);
        return;
    }
    if(frac_x && !frac_y) { /* _n0: (sum + 0) >> 6, clip (xeve_mc.c:123-156; xeve_mc.h:36-46) */
        for(int y = 0; y < h; y++)
            for(int x = 0; x < w; x++) {
                const xo_pel *r = ref + (iy + y) * s_ref + ix + x - back;
                int acc = 0;
                for(int t = 0; t < taps; t++) acc += cx[t] * r[t];
                pred[y * s_pred + x] = (xo_pel)clip3i(0, maxv, acc >> 6);
            }
        return;
    }
    if(!frac_x && frac_y) { /* _0n (xeve_mc.c:158-192) */
        for(int y = 0; y < h; y++)
            for(int x = 0; x < w; x++) {
                const xo_pel *r = ref + (iy + y - back) * s_ref + ix + x;
                int acc = 0;
                for(int t = 0; t < taps; t++) acc += cy[t] * r[t * s_ref];
                pred[y * s_pred + x] = (xo_pel)clip3i(0, maxv, acc >> 6);
            }
        return;
    }
    /* _nn: horizontal pass to an int16 buffer with shift1 and NO rounding
     * offset, then vertical pass with rounding (xeve_mc.c:194-254, 333-381) */
    const int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4;
    const int shift2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8;
    const int round2 = 1 << (shift2 - 1);
    const int rows   = h + taps - 1;
    int16_t  *buf    = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * rows);
    for(int y = 0; y < rows; y++)
        for(int x = 0; x < w; x++) {
            const xo_pel *r = ref + (iy + y - back) * s_ref + ix + x - back;
            int acc = 0;
            for(int t = 0; t < taps; t++) acc += cx[t] * r[t];
            buf[y * w + x] = (int16_t)(acc >> shift1);
        }
    for(int y = 0; y < h; y++)
        for(int x = 0; x < w; x++) {
            int acc = 0;
            for(int t = 0; t < taps; t++) acc += cy[t] * buf[(y + t) * w + x];
            pred[y * s_pred + x] = (xo_pel)clip3i(0, maxv, (acc + round2) >> shift2);
        }
    free(buf);
}

void xo_mc_l(int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred,
             xo_pel *pred, int w, int h, int bit_depth, const int16_t (*coef)[8])
{
    mc_generic(8, 4, frac_x, frac_y, ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, &coef[0][0]);
}

void xo_mc_c(int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred,
             xo_pel *pred, int w, int h, int bit_depth, const int16_t (*coef)[4])
{
    mc_generic(4, 5, frac_x, frac_y, ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, &coef[0][0]);
}

/* ------------------------------------------------------------------------- */
/* Main profile, first slice: the interpolation variants the   */
/* Main tools add to the dispatch tables.  Reference: src_main/xevem_mc.c.       */
/* ------------------------------------------------------------------------- */
/* Main luma 1/16-pel and chroma 1/32-pel filters (all phases populated; ISO/IEC 23094-1 8.5.4.3.2/3; the reference's copy is
 * xevem_mc.c:48-104).  tests/test_main_oracle_vs_ref.py compares both with the library's xevem_tbl_mc_l_coeff / _c_coeff. */
const int16_t xom_mc_l_coeff[16][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0},      {0, 1, -3, 63, 4, -2, 1, 0},    {-1, 2, -5, 62, 8, -3, 1, 0},    {-1, 3, -8, 60, 13, -4, 1, 0},
    {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 52, 26, -8, 3, -1}, {-1, 3, -9, 47, 31, -10, 4, -1}, {-1, 4, -11, 45, 34, -10, 4, -1},
    {-1, 4, -11, 40, 40, -11, 4, -1}, {-1, 4, -10, 34, 45, -11, 4, -1}, {-1, 4, -10, 31, 47, -9, 3, -1}, {-1, 3, -8, 26, 52, -11, 4, -1},
    {0, 1, -5, 17, 58, -10, 4, -1}, {0, 1, -4, 13, 60, -8, 3, -1},  {0, 1, -3, 8, 62, -5, 2, -1},    {0, 1, -2, 4, 63, -3, 1, 0},
};
const int16_t xom_mc_c_coeff[32][4] = {
    {0, 64, 0, 0},   {-1, 63, 2, 0},  {-2, 62, 4, 0},  {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2}, {-4, 56, 14, -2}, {-4, 55, 15, -2},
    {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3}, {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4},
    {-4, 36, 36, -4}, {-4, 33, 39, -4}, {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},  {0, 4, 62, -2},   {0, 2, 63, -1},
};

/* xevem_tbl_dmvr_mc_l / _c / xevem_tbl_bl_mc_l [frac_x != 0][frac_y != 0] (xevem_mc.c:167-463).
 *   kind 0, DMVR luma   (xevem_mc.c:167-291): 8 taps of xom_mc_l_coeff.  `ref` already points AT the block: the variants use only the
 *           FRACTION of gmv (gmv & 15) -- the _00 variant shifts gmv and then ignores it -- so the integer part is dropped here;
 *   kind 1, DMVR chroma (xevem_mc.c:383-463): the same with 4 taps of xom_mc_c_coeff and gmv & 31;
 *   kind 2, bilinear luma (xevem_mc.c:293-378): 2 taps {64 - 4f, 4f} (xevem_mc.c:108-126), the integer part of gmv DOES move ref, the taps
 *           start AT the sample (no step back), and the _nn variant filters h + 1 rows with the same two-stage shifts as the 8-tap one.
 * All share MAC_*_N0 / _0N / _NN_S1 / _NN_S2 with the Baseline functions (xeve_mc.h:36-73), which is what mc_generic restates. */
void xo_mc_main(int kind, int frac_x, int frac_y, const xo_pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, xo_pel *pred, int w, int h,
                int bit_depth)
{
    if(kind == 0) mc_generic(8, 4, frac_x, frac_y, ref, gmv_x & 15, gmv_y & 15, s_ref, s_pred, pred, w, h, bit_depth, &xom_mc_l_coeff[0][0]);
    else if(kind == 1) mc_generic(4, 5, frac_x, frac_y, ref, gmv_x & 31, gmv_y & 31, s_ref, s_pred, pred, w, h, bit_depth, &xom_mc_c_coeff[0][0]);
    else {
        int16_t bl[16][2];
        for(int f = 0; f < 16; f++) { bl[f][0] = (int16_t)(64 - 4 * f); bl[f][1] = (int16_t)(4 * f); }
        mc_generic(2, 4, frac_x, frac_y, ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, &bl[0][0]);
    }
}

/* ---- Main profile: the other entries of xevem_platform_init_func's dispatch list (src_main/xevem_util.c:3917-3966) ---- */
/* The ATS matrices xevem_tbl_tr[DCT8 | DST7][log2 N - 2] (xevem_tbl.c:421-565) in closed form -- every entry of all eight equals
 *   round(64 * sqrt(N) * sqrt(4 / (2N + 1)) * { cos(pi (2k + 1)(2j + 1) / (4N + 2)) | sin(pi (2k + 1)(j + 1) / (2N + 1)) })      [k][j]
 * (tests/test_main_oracle_vs_ref.py compares them with the library's table).  type 0 = DCT-VIII, 1 = DST-VII (xeve_def.h:557). */
void xo_ats_matrix(int type, int log2n, int8_t *m)
{
    const int    n = 1 << log2n;
    const double pi = 3.14159265358979323846, sc = 64.0 * sqrt((double)n) * sqrt(4.0 / (2 * n + 1));
    for(int k = 0; k < n; k++)
        for(int j = 0; j < n; j++) {
            const double v = sc * (type ? sin(pi * (2 * k + 1) * (j + 1) / (2 * n + 1)) : cos(pi * (2 * k + 1) * (2 * j + 1) / (4 * n + 2)));
            m[k * n + j] = (int8_t)(v >= 0 ? (int)floor(v + 0.5) : -(int)floor(-v + 0.5));
        }
}
/* xeve_itrans_map_tbl[type][log2 N - 1] (xevem_itdq.c:42-47, the functions :63-276): 1-D inverse ATS of `line` columns,
 *   block[i * N + j] = clip16((sum_{k < cut} coef[k * line + i] * M[k][j] + (1 << (shift - 1))) >> shift)   for i < line - skip_line, 0 for the other rows;
 * cut = N - skip_line_2, except that the 4-point forms are written out over all four inputs (:63-90, :170-198: the factorised DST-VII / DCT-VIII, equal to the
 * matrix product because the 4-point matrices satisfy a + b = d). */
void xo_itrans_ats(int type, int log2n, const int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2)
{
    const int n = 1 << log2n, cut = log2n == 2 ? 4 : n - skip_line_2, rnd = 1 << (shift - 1);
    int8_t    m[32 * 32];
    xo_ats_matrix(type, log2n, m);
    for(int i = 0; i < line; i++)
        for(int j = 0; j < n; j++) {
            int sum = 0;
            if(i < line - skip_line) {
                for(int k = 0; k < cut; k++) sum += coef[k * line + i] * m[k * n + j];
                sum = clip3i(-32768, 32767, (sum + rnd) >> shift);
            }
            block[i * n + j] = (int16_t)sum;
        }
}
void xo_trans_ats(int type, int log2n, const int16_t *block, int16_t *coef, int shift, int line, int skip_line, int skip_line_2)
{ /* xeve_trans_DST7_B4 .. _DCT8_B32 (xevem_tq.c:336-680): the 4-point forms are the same products factorised (29 + 55 = 84) */
    const int n = 1 << log2n, cut = log2n == 2 ? 4 : n - skip_line_2, rnd = 1 << (shift - 1);
    int8_t    m[32 * 32];
    xo_ats_matrix(type, log2n, m);
    for(int j = 0; j < n; j++)
        for(int i = 0; i < line; i++) {
            int sum = 0;
            if(i < line - skip_line && j < cut) {
                for(int k = 0; k < n; k++) sum += m[j * n + k] * block[i * n + k];
                sum = (sum + rnd) >> shift;
            }
            coef[j * line + i] = (int16_t)sum;
        }
}
/* xevem_scaled_horizontal / _vertical_sobel_filter (xevem_mc.c:2341-2395): 3x3 Sobel gradient (weights 1 2 1, unnormalised) of the interior samples; the border
 * row / column / corner take the value of the nearest interior sample -- both functions write the border that way, in different orders.  w, h >= 3. */
void xo_sobel(int vertical, const xo_pel *pred, int s_pred, int32_t *der, int s_der, int w, int h)
{
    for(int y = 0; y < h; y++)
        for(int x = 0; x < w; x++) {
            const int cy = clip3i(1, h - 2, y), cx = clip3i(1, w - 2, x);
            const xo_pel *c = pred + cy * s_pred + cx;
            der[y * s_der + x] = vertical ? c[s_pred - 1] - c[-s_pred - 1] + 2 * c[s_pred] - 2 * c[-s_pred] + c[s_pred + 1] - c[-s_pred + 1]
                                          : c[1 - s_pred] - c[-1 - s_pred] + 2 * c[1] - 2 * c[-1] + c[1 + s_pred] - c[-1 + s_pred];
        }
}
/* xevem_equal_coeff_computer (xevem_mc.c:2397-2447): accumulates the normal equations of the affine gradient search into eq[1 .. 2v][0 .. 2v] (64-bit sums of
 * products of the 32-bit terms iC[]; the right-hand side x 8).  The residual is read with the DERIVATIVE buffers' pitch (the reference indexes both with
 * j * derivate_buf_stride + k; residue_stride is unused there). */
void xo_equal_coeff(const xo_pel *residue, const int32_t *d0, const int32_t *d1, int s_der, int64_t (*eq)[7], int w, int h, int vertex_num)
{
    const int np = vertex_num << 1;
    for(int j = 0; j < h; j++)
        for(int k = 0; k < w; k++) {
            const int i = j * s_der + k;
            int32_t   c[6];
            if(vertex_num == 2) {
                c[0] = d0[i], c[1] = (int32_t)((uint32_t)k * (uint32_t)d0[i] + (uint32_t)j * (uint32_t)d1[i]);
                c[2] = d1[i], c[3] = (int32_t)((uint32_t)j * (uint32_t)d0[i] - (uint32_t)k * (uint32_t)d1[i]);
            }
            else {
                c[0] = d0[i], c[1] = (int32_t)((uint32_t)k * (uint32_t)d0[i]), c[2] = d1[i], c[3] = (int32_t)((uint32_t)k * (uint32_t)d1[i]);
                c[4] = (int32_t)((uint32_t)j * (uint32_t)d0[i]), c[5] = (int32_t)((uint32_t)j * (uint32_t)d1[i]);
            }
            for(int col = 0; col < np; col++) {
                for(int row = 0; row < np; row++) eq[col + 1][row] += (int64_t)c[col] * c[row];
                eq[col + 1][np] += (int64_t)c[col] * residue[i] * 8;
            }
        }
}

/* xeve_tbl_intra_pred_ang[group][right] (src_main/xevem_ipred.c:456-815): the angular predictors of the Main profile.  Every sample is a 4-tap interpolation
 * {32 - f, 64 - f, 32 + f, f} / 128 (xevem_tbl_ipred_adi, xevem_tbl.c:54-88: f = the 1/32 position) of one of the three neighbour lines, at a position that the
 * mode's slopes (xevem_tbl_ipred_dxdy, xevem_tbl.c:90-100: {dx/dy, dy/dx} << 10) project the sample onto; positions are clipped to [-1, w + h - 1].
 *   group 0 (ipm < IPD_VER): from the line above, to the right;  with `right`: the part that projects beyond the block's width comes from the right line
 *   group 1 (ipm > IPD_HOR): from the left line, downwards;      with `right`: from above (projected from the right edge) or from the right line
 *   group 2 (between):       from above (leftwards) or from the left line; with `right`: the second source is the right line, projected from the right edge */
static const int xo_ipred_dxdy[33][2] = {
    {0, 0}, {0, 0}, {0, 0}, {2816, 372}, {2048, 512}, {1408, 744}, {1024, 1024}, {744, 1408}, {512, 2048}, {372, 2816}, {256, 4096},
    {128, 8192}, {0, 0}, {128, 8192}, {256, 4096}, {372, 2816}, {512, 2048}, {744, 1408}, {1024, 1024}, {1408, 744}, {2048, 512},
    {2816, 372}, {4096, 256}, {8192, 128}, {0, 0}, {8192, 128}, {4096, 256}, {2816, 372}, {2048, 512}, {1408, 744}, {1024, 1024}, {744, 1408}, {512, 2048}};
void xo_ipred_ang(int group, int right, const xo_pel *le, const xo_pel *up, const xo_pel *ri, xo_pel *dst, int w, int h, int ipm, int bit_depth)
{
    const int *mt = xo_ipred_dxdy[ipm], pmax = w + h - 1, maxv = (1 << bit_depth) - 1;
    for(int j = 0; j < h; j++)
        for(int i = 0; i < w; i++) {
            const xo_pel *src;
            int pos, dir, d; /* taps at pos - dir, pos, pos + dir, pos + 2 dir; d = the projected distance << 10 whose 1/32 fraction picks the filter */
#define PROJ(D_IN, M) (d = (D_IN) * (M))
            if(group == 0) {
                PROJ(j + 1, mt[0]);
                if(!right || i < w - (d >> 10)) src = up, pos = i + (d >> 10), dir = 1;
                else PROJ(w - i, mt[1]), src = ri, pos = j - (d >> 10), dir = -1;
            }
            else if(group == 1) {
                if(!right) PROJ(i + 1, mt[1]), src = le, pos = j + (d >> 10), dir = 1;
                else {
                    PROJ(w - i, mt[1]);
                    if(j < (d >> 10)) PROJ(w - i, mt[0]), src = up, pos = i + (d >> 10), dir = 1;
                    else src = ri, pos = j - (d >> 10), dir = -1;
                }
            }
            else {
                PROJ(i + 1, mt[1]);
                if(j < (d >> 10)) PROJ(j + 1, mt[0]), src = up, pos = i - (d >> 10), dir = -1;
                else if(!right) src = le, pos = j - (d >> 10), dir = -1;
                else PROJ(w - i, mt[1]), src = ri, pos = j + (d >> 10), dir = 1;
            }
#undef PROJ
            const int f = (d >> 5) - ((d >> 10) << 5);
            const int a = src[clip3i(-1, pmax, pos - dir)], b = src[clip3i(-1, pmax, pos)], c = src[clip3i(-1, pmax, pos + dir)], e = src[clip3i(-1, pmax, pos + 2 * dir)];
            const int16_t t = (int16_t)((a * (32 - f) + b * (64 - f) + c * (32 + f) + e * f + 64) >> 7);
            dst[j * w + i] = (xo_pel)clip3i(0, maxv, t);
        }
}

/* a7 (reference: xeve_mc.c:449-463) */
void xo_avg(const int16_t *src, const int16_t *ref, int16_t *dst, int s_src, int s_ref, int s_dst, int w, int h)
{
    for(int y = 0; y < h; y++)
        for(int x = 0; x < w; x++)
            dst[y * s_dst + x] = (int16_t)(((int)src[y * s_src + x] + (int)ref[y * s_ref + x] + 1) >> 1);
}

/* ------------------------------------------------------------------------- */
/* transforms                                                                */
/* ------------------------------------------------------------------------- */
/* The EVC integer DCT-II family (xeve_tbl.c:83-236): every entry of every
 * size derives from g[j] = round(64*sqrt(2)*cos(j*pi/128)), j = 1..63 (g[0]=64):
 *   M_N[k][x] = +-g[fold((2x+1) * k * 64/N mod 256)].
 * tests/test_oracle_vs_ref.py checks all six generated matrices against the
 * reference's xeve_tbl_tm{2..64} symbols. */
void xo_dct_matrix(int n, int8_t *m)
{
    int g[65];
    g[0] = 64;
    for(int j = 1; j < 64; j++) g[j] = (int)floor(64.0 * sqrt(2.0) * cos(j * 3.14159265358979323846 / 128.0) + 0.5);
    g[64] = 0;
    for(int k = 0; k < n; k++)
        for(int x = 0; x < n; x++) {
            int th = ((2 * x + 1) * k * (64 / n)) % 256, sg = 1;
            if(th > 128) th = 256 - th;
            if(th > 64) { sg = -1; th = 128 - th; }
            m[k * n + x] = (int8_t)(sg * g[th]);
        }
}

static const int8_t *dct(int log2n)
{
    static int8_t tm[7][64 * 64];
    static int    ready[7];
    if(!ready[log2n]) { xo_dct_matrix(1 << log2n, tm[log2n]); ready[log2n] = 1; }
    return tm[log2n];
}

/* a9 forward 1-D (reference: xeve_tq.c:40-392).
 * dst[k*line + j] = (T)((sum_x M[k][x]*src[j*N + x] + add) >> shift); the cast
 * to the destination type truncates (no clip); the 64-point transform forces
 * outputs k >= 32 to zero (xeve_tq.c:321-381). step 0: s16 -> s32, step 1:
 * s32 -> s16.  step 2: s16 -> s16 -- the Main profile's tx_pb{2..64}
 * (src_main/xevem_tq.c:58-330; the 16-bit intermediate the iqt tool uses):
 * the same sums (a butterfly there, no intermediate overflow: |sum| <=
 * 64 * 90 * 32768 < 2^31), the same truncating store. */
void xo_tx(int log2n, const void *src, void *dst, int shift, int line, int step)
{
    const int     n   = 1 << log2n;
    const int8_t *m   = dct(log2n);
    const int64_t add = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    for(int j = 0; j < line; j++)
        for(int k = 0; k < n; k++) {
            int64_t acc = 0;
            if(!(n == 64 && k >= 32)) {
                for(int x = 0; x < n; x++) {
                    int64_t v = step != 1 ? ((const int16_t *)src)[j * n + x] : ((const int32_t *)src)[j * n + x];
                    acc += m[k * n + x] * v;
                }
                acc = (acc + add) >> shift;
            }
            if(step == 0) ((int32_t *)dst)[k * line + j] = (int32_t)acc;
            else          ((int16_t *)dst)[k * line + j] = (int16_t)acc;
        }
}

/* a12 inverse 1-D (reference: xeve_itdq.c:34-430, clips xeve_itdq.h:41-48).
 * dst[j*N + x] = clip((sum_k M[k][x]*src[k*line + j] + add) >> shift).
 * step 2: s16 -> s16 with ITX_CLIP -- the Main profile's itx_pb{2..64}
 * (src_main/xevem_itdq.c:302-547). */
void xo_itx(int log2n, const void *src, void *dst, int shift, int line, int step)
{
    const int     n   = 1 << log2n;
    const int8_t *m   = dct(log2n);
    const int64_t add = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    for(int j = 0; j < line; j++)
        for(int x = 0; x < n; x++) {
            int64_t acc = 0;
            for(int k = 0; k < n; k++) {
                int64_t v = step != 1 ? ((const int16_t *)src)[k * line + j] : ((const int32_t *)src)[k * line + j];
                acc += m[k * n + x] * v;
            }
            acc = (acc + add) >> shift;
            if(step == 0) {
                if(acc < INT32_MIN) acc = INT32_MIN;
                if(acc > INT32_MAX) acc = INT32_MAX;
                ((int32_t *)dst)[j * n + x] = (int32_t)acc;
            }
            else {
                if(acc < -32768) acc = -32768;
                if(acc > 32767) acc = 32767;
                ((int16_t *)dst)[j * n + x] = (int16_t)acc;
            }
        }
}

/* reference: xeve_tq.c:396-404, shifts xeve_util.c:34-35 */
void xo_trans(int16_t *coef, int log2w, int log2h, int bit_depth)
{
    int32_t   tb[64 * 64];
    const int shift1 = log2w - 1 + bit_depth - 8;
    const int shift2 = log2h + 6;
    xo_tx(log2w, coef, tb, 0, 1 << log2h, 0);
    xo_tx(log2h, tb, coef, shift1 + shift2, 1 << log2w, 1);
}

/* reference: xeve_itdq.c:435-440, shifts xeve_itdq.h:38-39 */
void xo_itrans(int16_t *coef, int log2w, int log2h, int bit_depth)
{
    int32_t tb[64 * 64];
    xo_itx(log2h, coef, tb, 0, 1 << log2w, 0);
    xo_itx(log2w, tb, coef, 7 + (12 - (bit_depth - 8)), 1 << log2h, 1);
}

/* quantisation scale tables (reference: xeve_tq.c:37-38, xeve_tbl.c:237) */
const int xo_quant_scale[2][6] = {{26214, 23302, 20560, 18396, 16384, 14764}, {26214, 23302, 20560, 18396, 16384, 14564}};
const int xo_dq_scale[6]       = {40, 45, 51, 57, 64, 71};

/* a10 (reference: xeve_tq.c:704-727; constants xeve_def.h:793-798) */
int xo_quant(int16_t *coef, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth)
{
    const int     log2_size = (log2w + log2h) >> 1;
    const int     tr_shift  = 15 - bit_depth - log2_size;
    const int     shift     = 14 + tr_shift + qp / 6;
    const int32_t offset    = (int32_t)(is_intra_slice ? 171 : 85) << (shift - 9);
    int           nnz       = 0;
    for(int i = 0; i < (1 << (log2w + log2h)); i++) {
        int     neg = coef[i] < 0;
        int32_t lev = (int32_t)iabs(coef[i]) * (int32_t)scale;
        lev         = (int16_t)((lev + offset) >> shift);
        coef[i]     = (int16_t)(neg ? -lev : lev);
        nnz += coef[i] != 0;
    }
    return nnz;
}

/* RDOQ zero-block pre-test (reference: xeve_tq.c:666-699) */
int xo_rdoq_zero_test(const int16_t *coef, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth)
{
    const int     odd       = (log2w + log2h) & 1;
    const int     log2_size = (log2w + log2h) >> 1;
    const int     tr_shift  = 15 - bit_depth - log2_size + (odd ? 7 : 0);
    const int     shift     = 14 + tr_shift + qp / 6;
    const int64_t offset    = (int64_t)(is_intra_slice ? 201 : 153) << (shift - 9);
    const int64_t thr       = ((int64_t)1 << shift) - offset;
    for(int i = 0; i < (1 << (log2w + log2h)); i++) {
        int64_t lev = (int64_t)iabs(coef[i]) * (int64_t)scale * (odd ? 181 : 1);
        if(lev >= thr) return 1;
    }
    return 0;
}

/* a13 (reference: xeve_itdq.c:442-475) */
void xo_dquant(int16_t *coef, int log2w, int log2h, int scale, int bit_depth)
{
    const int     odd       = (log2w + log2h) & 1;
    const int     log2_size = (log2w + log2h) >> 1;
    const int     tr_shift  = 15 - bit_depth - log2_size;
    const int     shift     = (uint8_t)(20 - 14 - tr_shift + (odd ? 8 : 0));
    const int32_t offset    = shift == 0 ? 0 : 1 << (shift - 1);
    for(int i = 0; i < (1 << (log2w + log2h)); i++) {
        int64_t lev = ((int64_t)coef[i] * ((int64_t)scale * (odd ? 181 : 1)) + offset) >> shift;
        if(lev < -32768) lev = -32768;
        if(lev > 32767) lev = 32767;
        coef[i] = (int16_t)lev;
    }
}

/* a15 (reference: xeve_recon.c:34-57) -- the sum wraps to int16 before the clip */
void xo_recon(const int16_t *coef, const xo_pel *pred, int is_coef, int cuw, int cuh, int s_rec, xo_pel *rec, int bit_depth)
{
    const int maxv = (1 << bit_depth) - 1;
    for(int y = 0; y < cuh; y++)
        for(int x = 0; x < cuw; x++) {
            int16_t t = is_coef ? (int16_t)(coef[y * cuw + x] + pred[y * cuw + x]) : pred[y * cuw + x];
            rec[y * s_rec + x] = (xo_pel)clip3i(0, maxv, t);
        }
}


/* ------------------------------------------------------------------------- */
/* integer-pel diamond search (reference: xeve_pinter.c:74-140, 363-551)      */
/* ------------------------------------------------------------------------- */
/* One component of get_mv_bits.  The reference reads xeve_tbl_mv_bits[mvd] for -2048 < mvd <= 2048
 * (xeve_tbl.c:286-496: 1 bit for 0, else 2*floor(log2(|mvd|+1)) + 2; the table's first entry, mvd = -2047,
 * holds 22 where the formula gives 24 -- restated as is) and an exp-Golomb length beyond (xeve_pinter.c:74-93). */
static int mvd_bits(int mvd)
{
    if(mvd > 2048 || mvd <= -2048) {
        unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd), nn = (a + 1) >> 12;
        int len_i;
        for(len_i = 11; len_i < 16 && nn != 0; len_i++) nn >>= 1;
        return (len_i << 1) + 1 + 1;
    }
    if(mvd == 0) return 1;
    if(mvd == -2047) return 22;
    unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd) + 1;
    int l = 0;
    while((a >> (l + 1)) != 0) l++;
    return 2 * l + 2;
}
int xo_mv_bits(int mvd_x, int mvd_y) { return mvd_bits(mvd_x) + mvd_bits(mvd_y); }

/* 16-point diamond, unit L1 radius 4 (xeve_pinter.c:57-65); the 8-point form is every other point halved */
static const int8_t dia16[16][2] = {{-4, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {0, 4}, {1, 3}, {2, 2}, {3, 1},
                                    {4, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -4}, {-1, -3}, {-2, -2}, {-3, -1}};

static void me_ipel_diamond_ex(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_me_job *job, int log2w, int log2h,
                               int bit_depth, const xo_me_params *p, xo_me_result *res, int16_t range_out[4]);
void xo_me_ipel_diamond(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_me_job *job,
                        int log2w, int log2h, int bit_depth, const xo_me_params *p, xo_me_result *res)
{
    me_ipel_diamond_ex(org0, s_org, org_bi, ref0, s_ref, job, log2w, log2h, bit_depth, p, res, NULL);
}
/* range_out: the caller's `range` array as the function leaves it -- it is re-centred in place after the dense round (xeve_pinter.c:463-468), and
 * pinter_me_epzs hands that array on to me_raster */
static void me_ipel_diamond_ex(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_me_job *job, int log2w, int log2h,
                               int bit_depth, const xo_me_params *p, xo_me_result *res, int16_t range_out[4])
{
    const int w = 1 << log2w, h = 1 << log2h;
    const xo_pel *org = p->bi ? org_bi + job->org_off : org0 + job->y * s_org + job->x;
    const int so = p->bi ? w : s_org;
    int range[4] = {job->range[0], job->range[1], job->range[2], job->range[3]};
    uint32_t cost_best = 0xFFFFFFFFu;
    int best_bits = 0, beststep = job->beststep_in, step = 0, not_found = 0;
    int bx = clip3i(p->min_clip[0], p->max_clip[0], job->mvi[0] >> 2);
    int by = clip3i(p->min_clip[1], p->max_clip[1], job->mvi[1] >> 2);
    const int ix = bx, iy = by;

    for(;;) {
        not_found++;
        int cand[128][2], nc = 0, round_step;
        if(step <= 2) {
            const int d = p->bi == 1 ? 5 : 2; /* BI_STEP : 2 */
            const int x0 = bx <= range[0] ? bx : bx - d, y0 = by <= range[1] ? by : by - d;
            const int x1 = bx >= range[2] ? bx : bx + d, y1 = by >= range[3] ? by : by + d;
            for(int yy = y0; yy <= y1; yy++)
                for(int xx = x0; xx <= x1; xx++) cand[nc][0] = xx, cand[nc][1] = yy, nc++;
            round_step = 2;
        }
        else {
            const int coarse = step > 8;
            for(int i = 0; i < 16; i++) {
                if(!coarse && i > 8) continue;                                  /* 8-point ring + centre */
                if(step == 4 && (i == 1 || i == 3 || i == 5 || i == 7)) continue; /* 4-point ring + centre */
                int dx, dy;
                if(coarse) dx = dia16[i][0], dy = dia16[i][1];
                else if(i < 8) dx = dia16[2 * i][0] / 2, dy = dia16[2 * i][1] / 2;
                else dx = dy = 0;
                cand[nc][0] = ix + (step >> (coarse ? 2 : 1)) * dx;
                cand[nc][1] = iy + (step >> (coarse ? 2 : 1)) * dy;
                nc++;
            }
            round_step = step;
        }
        for(int k = 0; k < nc; k++) {
            const int mx = cand[k][0], my = cand[k][1];
            if(mx > range[2] || mx < range[0] || my > range[3] || my < range[1]) continue;
            int bits = xo_mv_bits((mx << 2) - job->gmvp[0], (my << 2) - job->gmvp[1]) + p->refi_bits;
            if(p->bi) bits += p->extra_bits;
            uint32_t cost = (uint32_t)(p->lambda_mv * (uint32_t)bits + (1u << 15)) >> 16; /* u32 arithmetic as MV_COST, xeve_pinter.c:47 */
            int sad = xo_sad(w, h, org, ref0 + my * s_ref + mx, so, s_ref, bit_depth);
            cost += (uint32_t)(p->bi ? sad >> 1 : sad);
            if(cost < cost_best) bx = mx, by = my, beststep = round_step, not_found = 0, cost_best = cost, best_bits = bits;
        }
        if(step <= 2) {
            const int sr = p->bi == 1 ? 5 : p->range_recentre; /* get_range_ipel (xeve_pinter.c:122-140) */
            range[0] = clip3i(p->min_clip[0], p->max_clip[0], bx - sr);
            range[2] = clip3i(p->min_clip[0], p->max_clip[0], bx + sr);
            range[1] = clip3i(p->min_clip[1], p->max_clip[1], by - sr);
            range[3] = clip3i(p->min_clip[1], p->max_clip[1], by + sr);
            step += 2;
        }
        if(not_found == p->faststep) break;
        if(p->bi == 1) break;
        step <<= 1;
        if(step > p->max_search_range) break;
    }
    res->mv[0] = (int16_t)((bx - job->x) << 2);
    res->mv[1] = (int16_t)((by - job->y) << 2);
    res->cost = cost_best;
    res->beststep = beststep;
    res->best_mv_bits = best_bits;
    if(range_out)
        for(int i = 0; i < 4; i++) range_out[i] = (int16_t)range[i];
}


/* ------------------------------------------------------------------------- */
/* sub-pel pattern search (reference: xeve_pinter.c:50-70, 553-697)           */
/* ------------------------------------------------------------------------- */
static const int8_t pat_hpel[8][2] = {{-2, 0}, {-2, 2}, {0, 2}, {2, 2}, {2, 0}, {2, -2}, {0, -2}, {-2, -2}};
static const int8_t pat_qpel[8][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, 1}, {1, 1}, {-1, -1}, {1, -1}};

void xo_me_spel_pattern(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, const xo_spel_job *job,
                        int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_spel_params *p, xo_me_result *res)
{
    const int w = 1 << log2w, h = 1 << log2h;
    const xo_pel *org = p->bi ? org_bi + job->org_off : org0 + job->y * s_org + job->x;
    const int so = p->bi ? w : s_org;
    xo_pel *pred = (xo_pel *)malloc(sizeof(xo_pel) * (size_t)w * h);
    uint32_t cost_best = 0xFFFFFFFFu;
    int best_bits = 0, mvx = job->mvi[0], mvy = job->mvi[1];
    for(int stage = 0; stage < 2; stage++) {
        const int cnt = stage ? p->qpel_cnt : p->hpel_cnt;
        const int8_t(*pat)[2] = stage ? pat_qpel : pat_hpel;
        const int cx = mvx + (job->x << 2), cy = mvy + (job->y << 2); /* centre of this stage: picture coordinates, quarter pel */
        for(int i = 0; i < cnt; i++) {
            const int mx = cx + pat[i][0], my = cy + pat[i][1];
            int bits = xo_mv_bits(mx - job->gmvp[0], my - job->gmvp[1]) + p->refi_bits;
            if(p->bi) bits += p->extra_bits;
            uint32_t cost = (uint32_t)(p->lambda_mv * (uint32_t)bits + (1u << 15)) >> 16;
            /* xeve_mc_l picks the variant from the low 4 bits of (mv << 2) and positions with the same value */
            xo_mc_l((mx << 2) & 15, (my << 2) & 15, ref0, mx << 2, my << 2, s_ref, w, pred, w, h, bit_depth, coef);
            const int sad = xo_sad(w, h, org, pred, so, w, bit_depth);
            cost += (uint32_t)(p->bi ? sad >> 1 : sad);
            if(cost < cost_best) {
                mvx = mx - (job->x << 2), mvy = my - (job->y << 2), cost_best = cost;
                if(stage) best_bits = bits; /* only the quarter-pel loop records the bits, xeve_pinter.c:683 */
            }
        }
    }
    free(pred);
    res->mv[0] = (int16_t)mvx, res->mv[1] = (int16_t)mvy;
    res->cost = cost_best, res->beststep = 0, res->best_mv_bits = best_bits;
}


/* ------------------------------------------------------------------------- */
/* EPZS driver (reference: pinter_me_epzs, xeve_pinter.c:699-869)             */
/* ------------------------------------------------------------------------- */
static void epzs_range(const xo_me_params *p, int cx, int cy, int16_t range[4])
{
    const int sr = p->bi == 1 ? 5 : p->range_recentre; /* get_range_ipel, xeve_pinter.c:122-140 */
    range[0] = (int16_t)clip3i(p->min_clip[0], p->max_clip[0], cx - sr);
    range[1] = (int16_t)clip3i(p->min_clip[1], p->max_clip[1], cy - sr);
    range[2] = (int16_t)clip3i(p->min_clip[0], p->max_clip[0], cx + sr);
    range[3] = (int16_t)clip3i(p->min_clip[1], p->max_clip[1], cy + sr);
}

static uint32_t me_cand_cost(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, int mx, int my, const int16_t gmvp[2],
                             int log2w, int log2h, int bit_depth, const xo_me_params *p, int *bits)
{   /* get_mv_bits + MV_COST + SAD at one integer position (xeve_pinter.c:200-209, 323-341) */
    const int w = 1 << log2w, h = 1 << log2h;
    int mv_bits = xo_mv_bits((mx << 2) - gmvp[0], (my << 2) - gmvp[1]) + p->refi_bits;
    if(p->bi) mv_bits += p->extra_bits;
    uint32_t cost = (uint32_t)(((uint64_t)p->lambda_mv * (uint32_t)mv_bits + (1u << 15)) >> 16);
    const xo_pel *ref = ref0 + (long)my * s_ref + mx;
    if(p->bi) cost += (uint32_t)(xo_sad(w, h, org_bi, ref, w, s_ref, bit_depth) >> 1);
    else cost += (uint32_t)xo_sad(w, h, org0 + (long)y * s_org + x, ref, s_org, s_ref, bit_depth);
    *bits = mv_bits;
    return cost;
}

void xo_me_raster(const xo_pel *org0, int s_org, const xo_pel *ref0, int s_ref, int x, int y, const int16_t range[4], const int16_t gmvp[2], int log2w, int log2h,
                  int bit_depth, const xo_me_params *p, int refi, xo_me_result *res)
{
    const int lmin = log2w < log2h ? log2w : log2h, st = (1 << (lmin - 1)) > 5 ? (1 << (lmin - 1)) : 5; /* max(RASTER_SEARCH_STEP, half the CU) */
    uint32_t cost_best = 0xFFFFFFFFu;
    int best_bits = 0, bits;
    int mvx = res->mv[0], mvy = res->mv[1]; /* (the reference leaves `mv` as handed in when nothing is evaluated) */
    for(int i = range[1]; i <= range[3]; i += st * (refi + 1))
        for(int j = range[0]; j <= range[2]; j += st * (refi + 1)) {
            const uint32_t c = me_cand_cost(org0, s_org, NULL, ref0, s_ref, x, y, j, i, gmvp, log2w, log2h, bit_depth, p, &bits);
            if(c < cost_best) mvx = (j - x) << 2, mvy = (i - y) << 2, cost_best = c, best_bits = bits;
        }
    for(int ss = ((refi + 1) * st) >> 1; ss > 0; ss >>= 1) {
        const int cx = mvx, cy = mvy;
        for(int i = -ss; i <= ss; i += ss)
            for(int j = -ss; j <= ss; j += ss) {
                const int mx = (cx >> 2) + x + j, my = (cy >> 2) + y + i;
                if(mx < range[0] || mx > range[2] || my < range[1] || my > range[3]) continue;
                const uint32_t c = me_cand_cost(org0, s_org, NULL, ref0, s_ref, x, y, mx, my, gmvp, log2w, log2h, bit_depth, p, &bits);
                if(c < cost_best) mvx = (mx - x) << 2, mvy = (my - y) << 2, cost_best = c, best_bits = bits;
            }
    }
    res->mv[0] = (int16_t)mvx, res->mv[1] = (int16_t)mvy, res->cost = cost_best, res->beststep = 0, res->best_mv_bits = best_bits;
}

void xo_me_ipel_refinement(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t range[4],
                           const int16_t gmvp[2], const int16_t mvi[2], int log2w, int log2h, int bit_depth, const xo_me_params *p, xo_me_result *res)
{
    static const int pos[9][2] = {{0, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
    const int ix = clip3i(p->min_clip[0], p->max_clip[0], mvi[0] >> 2), iy = clip3i(p->min_clip[1], p->max_clip[1], mvi[1] >> 2);
    int bx = ix, by = iy, best_bits = 0, bits;
    uint32_t cost_best = 0xFFFFFFFFu;
    for(int i = 0; i < 9; i++) {
        const int mx = ix + pos[i][0], my = iy + pos[i][1];
        if(mx > range[2] || mx < range[0] || my > range[3] || my < range[1]) continue;
        const uint32_t c = me_cand_cost(org0, s_org, org_bi, ref0, s_ref, x, y, mx, my, gmvp, log2w, log2h, bit_depth, p, &bits);
        if(c < cost_best) bx = mx, by = my, cost_best = c, best_bits = bits;
    }
    res->mv[0] = (int16_t)((bx - x) << 2), res->mv[1] = (int16_t)((by - y) << 2), res->cost = cost_best, res->beststep = 0, res->best_mv_bits = best_bits;
}

uint32_t xo_me_epzs(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t mvp[2],
                    int16_t mv[2], int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_epzs_params *p)
{
    int mot = 0;
    return xo_me_epzs_mot(org0, s_org, org_bi, ref0, s_ref, x, y, mvp, mv, log2w, log2h, bit_depth, coef, p, &mot);
}

uint32_t xo_me_epzs_mot(const xo_pel *org0, int s_org, const xo_pel *org_bi, const xo_pel *ref0, int s_ref, int x, int y, const int16_t mvp[2],
                        int16_t mv[2], int log2w, int log2h, int bit_depth, const int16_t (*coef)[8], const xo_epzs_params *p, int *mot_bits)
{
    xo_me_params me = p->me;
    xo_me_job    job;
    xo_me_result r;
    uint32_t     cost_best = 0xFFFFFFFFu;
    int          beststep = 0, tmpstep = 0;
    job.x = x, job.y = y, job.org_off = 0;
    job.gmvp[0] = (int16_t)(mvp[0] + (x << 2)), job.gmvp[1] = (int16_t)(mvp[1] + (y << 2));
    const int16_t *start = me.bi == 1 ? mv : mvp;
    job.mvi[0] = (int16_t)(start[0] + (x << 2)), job.mvi[1] = (int16_t)(start[1] + (y << 2));
    epzs_range(&me, clip3i(me.min_clip[0], me.max_clip[0], x + (start[0] >> 2)), clip3i(me.min_clip[1], me.max_clip[1], y + (start[1] >> 2)),
               job.range);
    me.faststep = 3; /* MAX_FIRST_SEARCH_STEP */
    job.beststep_in = tmpstep;
    int16_t range_after[4];
    me_ipel_diamond_ex(org0, s_org, org_bi, ref0, s_ref, &job, log2w, log2h, bit_depth, &me, &r, range_after);
    tmpstep = r.beststep;
    if(me.bi != 1 && r.best_mv_bits > 0) *mot_bits = r.best_mv_bits; /* pi->mot_bits[lidx] (xeve_pinter.c:546-548) */
    if(r.cost < cost_best) {
        cost_best = r.cost, mv[0] = r.mv[0], mv[1] = r.mv[1];
        beststep = (abs(mvp[0] - mv[0]) < 2 && abs(mvp[1] - mv[1]) < 2) ? 0 : tmpstep;
    }
    if(me.bi == 0 && beststep > 5 && (p->me.reserved & 1)) { /* me_raster: bi == BI_NON && beststep > RASTER_SEARCH_THD && me_complexity > 1 (:757-767) */
        xo_me_result rr;
        rr.mv[0] = r.mv[0], rr.mv[1] = r.mv[1]; /* mvt as the diamond search left it */
        xo_me_raster(org0, s_org, ref0, s_ref, x, y, range_after, job.gmvp, log2w, log2h, bit_depth, &me, (p->me.reserved >> 8) & 0xFF, &rr);
        if(rr.best_mv_bits > 0) *mot_bits = rr.best_mv_bits;
        if(rr.cost < cost_best) beststep = 5, cost_best = rr.cost, mv[0] = rr.mv[0], mv[1] = rr.mv[1];
    }
    while(me.bi != 1 && beststep > 0) { /* REFINE_SEARCH_THD 0 */
        /* note: get_range_ipel is given the UNCLIPPED centre here (xeve_pinter.c:785-788) */
        epzs_range(&me, x + (mv[0] >> 2), y + (mv[1] >> 2), job.range);
        job.mvi[0] = (int16_t)(mv[0] + (x << 2)), job.mvi[1] = (int16_t)(mv[1] + (y << 2));
        beststep = 0;
        me.faststep = 2; /* MAX_REFINE_SEARCH_STEP */
        job.beststep_in = tmpstep;
        xo_me_ipel_diamond(org0, s_org, org_bi, ref0, s_ref, &job, log2w, log2h, bit_depth, &me, &r);
        tmpstep = r.beststep;
        if(r.best_mv_bits > 0) *mot_bits = r.best_mv_bits;
        if(r.cost < cost_best) {
            cost_best = r.cost, mv[0] = r.mv[0], mv[1] = r.mv[1];
            beststep = (abs(mvp[0] - mv[0]) < 2 && abs(mvp[1] - mv[1]) < 2) ? 0 : tmpstep;
        }
    }
    if(p->spel.hpel_cnt == 0) { /* me_level <= ME_LEV_IPEL: integer refinement instead of the sub-pel pattern (:835-866) */
        int16_t mvi[2] = {(int16_t)(mv[0] + (x << 2)), (int16_t)(mv[1] + (y << 2))};
        epzs_range(&me, x + (mv[0] >> 2), y + (mv[1] >> 2), job.range);
        xo_me_ipel_refinement(org0, s_org, org_bi, ref0, s_ref, x, y, job.range, job.gmvp, mvi, log2w, log2h, bit_depth, &me, &r);
        if(me.bi != 1 && r.best_mv_bits > 0) *mot_bits = r.best_mv_bits;
        if(r.cost < cost_best) cost_best = r.cost, mv[0] = r.mv[0], mv[1] = r.mv[1];
        return cost_best;
    }
    xo_spel_params sp = p->spel;
    sp.lambda_mv = me.lambda_mv, sp.refi_bits = me.refi_bits, sp.extra_bits = me.extra_bits, sp.bi = me.bi;
    xo_spel_job sj;
    sj.x = x, sj.y = y, sj.org_off = 0, sj.gmvp[0] = job.gmvp[0], sj.gmvp[1] = job.gmvp[1], sj.mvi[0] = mv[0], sj.mvi[1] = mv[1];
    xo_me_spel_pattern(org0, s_org, org_bi, ref0, s_ref, &sj, log2w, log2h, bit_depth, coef, &sp, &r);
    if(!me.bi && r.best_mv_bits > 0) *mot_bits = r.best_mv_bits; /* xeve_pinter.c:690-692 */
    if(r.cost < cost_best) cost_best = r.cost, mv[0] = r.mv[0], mv[1] = r.mv[1];
    return cost_best;
}


/* ------------------------------------------------------------------------- */
/* RDOQ (reference: xeve_tq.c:406-649)                                        */
/* ------------------------------------------------------------------------- */
void xo_zigzag(int log2w, int log2h, uint16_t *scan)
{
    const int w = 1 << log2w, h = 1 << log2h;
    int pos = 0;
    /* anti-diagonals x + y = l; odd l run down-left (x decreasing), even l run up-right (xeve_util.c:1301-1325) */
    for(int l = 0; l < w + h - 1; l++) {
        if(l & 1) {
            for(int x = l < w - 1 ? l : w - 1, y = l - x; x >= 0 && y < h; x--, y++) scan[pos++] = (uint16_t)(y * w + x);
        }
        else {
            for(int y = l < h - 1 ? l : h - 1, x = l - y; y >= 0 && x < w; x++, y--) scan[pos++] = (uint16_t)(y * w + x);
        }
    }
}

int64_t xo_err_scale(int qp_rem, int log2_size, int bit_depth, int tool_iqt)
{
    const int tr_shift = 15 - bit_depth - log2_size; /* MAX_TX_DYNAMIC_RANGE - bd - (i + 1), i = log2_size - 1 */
    double e = (double)(1 << 15) * pow(2.0, -tr_shift);
    e = e / xo_quant_scale[tool_iqt][qp_rem] / (1 << (bit_depth - 8));
    return (int64_t)(e * (double)(1 << 20));
}

/* rate of coding |level| after `run_nonzero ? run > 0 : run == 0` zeros (get_ic_rate_cost_rl, xeve_tq.c:425-456);
 * s32 arithmetic as the reference, context index c = 0 (luma) / 2 (chroma) (xeve_rdoq_set_ctx_cc, xeve_tq.c:492-495) */
static int64_t rl_cost(uint32_t abs_level, int run_nonzero, int c, int64_t lambda, const xo_rdoq_est *e)
{
    uint32_t rate;
    if(abs_level == 0) rate = (uint32_t)e->run[c + run_nonzero][1];
    else {
        rate = 32768u + (uint32_t)e->run[c + run_nonzero][0];
        if(abs_level == 1) rate += (uint32_t)e->level[c][0];
        else rate += (uint32_t)e->level[c][1] + (uint32_t)e->level[c + 1][1] * (abs_level - 2) + (uint32_t)e->level[c + 1][0];
    }
    return (int64_t)(int32_t)rate * lambda;
}

int xo_rdoq(int16_t *coef, int log2w, int log2h, int qp, double d_lambda, int is_luma, int bit_depth, int tool_iqt, const xo_rdoq_est *est)
{
    const int odd = (log2w + log2h) & 1, ns_shift = odd ? 7 : 0, ns_scale = odd ? 181 : 1, ns_offset = odd ? 1 << (ns_shift - 1) : 0;
    const int q_value   = (xo_quant_scale[tool_iqt][qp % 6] * ns_scale + ns_offset) >> ns_shift;
    const int log2_size = (log2w + log2h) >> 1;
    const int q_bits    = 14 + (15 - bit_depth - log2_size) + qp / 6;
    const int n = 1 << (log2w + log2h), c = is_luma ? 0 : 2, ctx_last = is_luma ? 0 : 1;
    const int64_t lambda = (int64_t)(d_lambda * (double)(1 << 15) + 0.5);
    const int64_t es     = xo_err_scale(qp % 6, log2_size, bit_depth, tool_iqt);
    uint16_t *scan = (uint16_t *)malloc(sizeof(uint16_t) * n);
    int64_t  *ld   = (int64_t *)malloc(sizeof(int64_t) * n);
    int      *mx   = (int *)malloc(sizeof(int) * n);
    int16_t  *out  = (int16_t *)calloc(n, sizeof(int16_t));
    xo_zigzag(log2w, log2h, scan);
    int64_t block_uncoded = 0;
    int     sum_all = 0, nnz = 0;
    for(int p = 0; p < n; p++) {
        const int     v = coef[scan[p]];
        const int64_t t = (int64_t)iabs(v) * q_value, cap = (int64_t)INT32_MAX - ((int64_t)1 << (q_bits - 1));
        const int64_t level_double = (int)(t < cap ? t : cap);
        uint32_t m = (uint32_t)(level_double >> q_bits);
        if(!((level_double - ((int64_t)m << q_bits)) < ((int64_t)1 << (q_bits - 1)))) m++;
        const int64_t err = (level_double * es) >> 20;
        block_uncoded += err * err;
        ld[p] = level_double, mx[p] = v > 0 ? (int16_t)m : -(int16_t)m;
        sum_all += (int)m;
    }
    if(sum_all != 0) {
        int64_t best_cost = block_uncoded + (int64_t)est->cbf[0] * lambda, base_cost = block_uncoded + (int64_t)est->cbf[1] * lambda;
        uint32_t run = 0, best_last = 0;
        for(int p = 0; p < n; p++) {
            const uint32_t max_abs = (uint32_t)iabs(mx[p]);
            const int64_t  e1 = (ld[p] * es) >> 20, uncoded = e1 * e1;
            int64_t  coded = uncoded + rl_cost(0, run != 0, c, lambda, est);
            uint32_t best = 0;
            const uint32_t lo = max_abs > 1 ? max_abs - 1 : 1;
            for(uint32_t a = max_abs; a >= lo; a--) { /* get_coded_level_rl, xeve_tq.c:458-490 */
                const int64_t d = ld[p] - ((int64_t)a << q_bits), e2 = (d * es) >> 20, cost = e2 * e2 + rl_cost(a, run != 0, c, lambda, est);
                if(cost < coded) best = a, coded = cost;
            }
            out[scan[p]] = (int16_t)(mx[p] < 0 ? -(int32_t)best : (int32_t)best);
            base_cost += coded - uncoded;
            if(best) {
                const int64_t cur_is_last = base_cost + (int64_t)est->last[ctx_last][1] * lambda;
                base_cost += (int64_t)est->last[ctx_last][0] * lambda;
                if(cur_is_last < best_cost) best_cost = cur_is_last, best_last = (uint32_t)p + 1;
                run = 0;
            }
            else run++;
        }
        for(int p = 0; p < n; p++) {
            if((uint32_t)p < best_last) nnz += out[scan[p]] != 0;
            else out[scan[p]] = 0;
        }
    }
    memcpy(coef, out, sizeof(int16_t) * n);
    free(scan), free(ld), free(mx), free(out);
    return nnz;
}

/* ===================================================================================================================
 * CABAC (SBAC) bit counting -- reference: src_base/xeve_eco.c, src_base/xeve_mode.c
 * =================================================================================================================== */
void xo_sbac_reset(xo_sbac *s)
{   /* xeve_eco.c:597-620 */
    memset(s, 0, sizeof(*s));
    s->range = 16384, s->code_bits = 11;
    for(int i = 0; i < XO_SBAC_NCTX; i++) s->ctx[i] = 512;
}

void xo_sbac_bit_reset(xo_sbac *s)
{   /* xeve_mode.c:39-49 */
    s->code &= 0x7FFFF;
    s->code_bits = 11;
    s->pending_byte = s->is_pending_byte = s->stacked_ff = s->stacked_zero = s->bitcounter = s->bin_counter = 0;
}

uint32_t xo_sbac_bits(const xo_sbac *s)
{   /* xeve_mode.c:51-55 */
    return s->bitcounter + 8 * (s->stacked_zero + s->stacked_ff) + 8 * (s->is_pending_byte ? 1 : 0) + 8 - s->code_bits + 3;
}

/* sbac_put_byte with is_bitcount set (xeve_eco.c:397-427): written bytes only advance bitcounter (xeve_bsw_write_est, :392) */
/* write mode (is_bitcount clear: the bitstream writer's coder): the bytes go to the sink xo_eco_ctu sets instead of advancing the counter */
static __thread uint8_t *g_sink;
static __thread int      g_sink_n, g_sink_cap;
static void sink_put(uint8_t b)
{
    if(g_sink_n < g_sink_cap) g_sink[g_sink_n] = b;
    g_sink_n++;
}
static void sbac_byte(xo_sbac *s, uint8_t b)
{
    if(s->is_pending_byte) {
        if(s->pending_byte == 0) s->stacked_zero++;
        else if(g_sink) {
            for(; s->stacked_zero; s->stacked_zero--) sink_put(0x00);
            sink_put((uint8_t)s->pending_byte);
        }
        else {
            s->bitcounter += 8 * s->stacked_zero + 8;
            s->stacked_zero = 0;
        }
    }
    s->pending_byte = b, s->is_pending_byte = 1;
}

/* sbac_carry_propagate (xeve_eco.c:429-453) */
static void sbac_carry(xo_sbac *s)
{
    uint32_t out = s->code >> 17;
    s->code &= (1u << 17) - 1;
    if(out < 0xFF) {
        for(; s->stacked_ff; s->stacked_ff--) sbac_byte(s, 0xFF);
        sbac_byte(s, (uint8_t)out);
    }
    else if(out > 0xFF) {
        s->pending_byte++;
        for(; s->stacked_ff; s->stacked_ff--) sbac_byte(s, 0x00);
        sbac_byte(s, (uint8_t)(out & 0xFF));
    }
    else s->stacked_ff++;
}

static void sbac_shift(xo_sbac *s)
{
    s->code <<= 1;
    if(--s->code_bits == 0) {
        sbac_carry(s);
        s->code_bits = 8;
    }
}

void xo_sbac_bin_ep(xo_sbac *s, uint32_t bin)
{   /* xeve_eco.c:455-472 -- note the range loses its LSB */
    s->bin_counter++;
    s->range >>= 1;
    if(bin) s->code += s->range;
    s->range <<= 1;
    sbac_shift(s);
}

void xo_sbac_bin(xo_sbac *s, int ci, uint32_t bin)
{   /* xeve_eco.c:521-575 */
    uint16_t state = s->ctx[ci] >> 1, mps = s->ctx[ci] & 1;
    uint32_t lps = ((uint32_t)state * s->range) >> 9;
    if(lps < 437) lps = 437;
    s->bin_counter++;
    s->range -= lps;
    if(bin != mps) {
        if(s->range >= lps) {
            s->code += s->range;
            s->range = lps;
        }
        state = state + ((512 - state + 16) >> 5);
        if(state > 256) mps = 1 - mps, state = 512 - state;
    }
    else state = state - ((state + 16) >> 5);
    s->ctx[ci] = (uint16_t)((state << 1) + mps);
    while(s->range < 8192) {
        s->range <<= 1;
        sbac_shift(s);
    }
}

/* sbac_write_unary_sym (xeve_eco.c:474-490) with num_ctx = 2 */
static void sbac_unary2(xo_sbac *s, uint32_t sym, int ci)
{
    xo_sbac_bin(s, ci, sym ? 1 : 0);
    while(sym) {
        sym--;
        xo_sbac_bin(s, ci + 1, sym ? 1 : 0);
    }
}

void xo_eco_run_length_cc(xo_sbac *s, const int16_t *coef, int log2w, int log2h, int num_sig, int ch, int cm_init)
{   /* xeve_eco.c:707-771 */
    uint16_t scan[64 * 64];
    uint32_t n = 1u << (log2w + log2h), run = 0, prev_level = 6;
    xo_zigzag(log2w, log2h, scan);
    for(uint32_t pos = 0; pos < n; pos++) {
        int c = coef[scan[pos]];
        if(!c) {
            run++;
            continue;
        }
        uint32_t level = (uint32_t)(c < 0 ? -c : c) & 0xFFFF; /* XEVE_ABS16 */
        uint32_t pl = prev_level - 1 < 5 ? prev_level - 1 : 5;
        int t0 = cm_init == 1 ? (int)(pl << 1) + (ch ? 12 : 0) : (ch ? 2 : 0);
        sbac_unary2(s, run, XO_CTX_RUN + t0);
        sbac_unary2(s, level - 1, XO_CTX_LEVEL + t0);
        xo_sbac_bin_ep(s, c < 0);
        if(pos == n - 1) break;
        run = 0, prev_level = level, num_sig--;
        xo_sbac_bin(s, XO_CTX_LAST + (ch ? 1 : 0), num_sig == 0);
        if(num_sig == 0) break;
    }
}

/* xeve_eco_abs_mvd + sign (xeve_eco.c:1205-1270) */
static void sbac_mvd1(xo_sbac *s, int v)
{
    uint32_t a = (uint32_t)(v < 0 ? -v : v), nn = (a + 1) >> 1;
    int len = 0;
    for(; len < 16 && nn; len++) nn >>= 1;
    uint32_t code = (1u << len) | ((a + 1 - (1u << len)) & ((1u << len) - 1));
    int nbin = 2 * len + 1;
    for(int i = 0; i < nbin; i++) {
        uint32_t b = (code >> (nbin - 1 - i)) & 1;
        if(i <= 1) xo_sbac_bin(s, XO_CTX_MVD, b);
        else xo_sbac_bin_ep(s, b);
    }
    if(a) xo_sbac_bin_ep(s, v < 0);
}

/* xeve_eco_mvp_idx = sbac_write_truncate_unary_sym(idx, 3, 4) (xeve_eco.c:492-511, 1190-1203) */
static void sbac_mvp_idx(xo_sbac *s, int idx)
{
    for(int i = 0; i < 3; i++) {
        int sym = i == idx ? 0 : 1;
        xo_sbac_bin(s, XO_CTX_MVP_IDX + i, sym);
        if(!sym) break;
    }
}

/* xeve_eco_refi (xeve_eco.c:1158-1188) */
static void sbac_refi(xo_sbac *s, int num_refp, int refi)
{
    if(num_refp <= 1) return;
    if(refi == 0) {
        xo_sbac_bin(s, XO_CTX_REFI, 0);
        return;
    }
    xo_sbac_bin(s, XO_CTX_REFI, 1);
    for(int i = 2; i < num_refp; i++) {
        int bin = i == refi + 1 ? 0 : 1;
        if(i == 2) xo_sbac_bin(s, XO_CTX_REFI + 1, bin);
        else xo_sbac_bin_ep(s, bin);
        if(!bin) break;
    }
}

/* xeve_eco_coef -> xeve_eco_coefficient (xeve_eco.c:925-1089) for MODE_INTER, one transform block per component,
 * no delta QP, b_no_cbf 0; run[] = the components this call covers */
static void sbac_coef(xo_sbac *s, const xo_cu_bits_params *p, const xo_cu_bits_job *j, const int16_t *coef, const int run[3], int is_intra, int b_no_cbf)
{
    int ws = p->chroma_format_idc <= 2, hs = p->chroma_format_idc <= 1;
    int cbf[3] = {!!j->nnz[0], !!j->nnz[1], !!j->nnz[2]}, cbf_all = 0;
    for(int c = 0; c < 3; c++) cbf_all += run[c] && cbf[c];
    /* xeve_eco_cbf (xeve_eco.c:793-894), sub_pos 0, is_sub 0 */
    if(!is_intra) {
        if(b_no_cbf != 1 && run[0] + run[1] + run[2] == 3) { /* (b_no_cbf: the flag is implied, xeve_eco.c:817-819) */
            xo_sbac_bin(s, XO_CTX_CBF_ALL, cbf_all != 0);
            if(!cbf_all) return;
        }
        if(run[1] && p->chroma_format_idc) xo_sbac_bin(s, XO_CTX_CBF_CB, cbf[1]);
        if(run[2] && p->chroma_format_idc) xo_sbac_bin(s, XO_CTX_CBF_CR, cbf[2]);
        if(run[0] && cbf[1] + cbf[2] != 0) xo_sbac_bin(s, XO_CTX_CBF_LUMA, cbf[0]);
    }
    else { /* intra: one flag per component that is coded (xeve_eco.c:864-890) */
        if(run[1] && p->chroma_format_idc) xo_sbac_bin(s, XO_CTX_CBF_CB, cbf[1]);
        if(run[2] && p->chroma_format_idc) xo_sbac_bin(s, XO_CTX_CBF_CR, cbf[2]);
        if(run[0]) xo_sbac_bin(s, XO_CTX_CBF_LUMA, cbf[0]);
    }
    for(int c = 0; c < 3; c++)
        if(j->nnz[c] && run[c])
            xo_eco_run_length_cc(s, coef + j->coef_off[c], p->log2_cuw - (c ? ws : 0), p->log2_cuh - (c ? hs : 0), j->nnz[c], c != 0, p->cm_init);
}

uint32_t xo_cu_bits(const xo_sbac *in, xo_sbac *out, const xo_cu_bits_params *p, const xo_cu_bits_job *j, const int16_t *coef)
{
    xo_sbac s = in[j->sbac];
    if(j->mode == XO_BITS_ECO_COEF) { /* ctx->fn_eco_coef = xeve_eco_coef (xeve_eco.c:1067-1089) alone, wherever the coder stands */
        const int f = j->dir_flag, run[3] = {(f >> 2) & 1, (f >> 3) & 1, (f >> 4) & 1};
        if(!(f & XO_ECO_NO_RESET)) xo_sbac_bit_reset(&s);
        sbac_coef(&s, p, j, coef, run, f & XO_ECO_INTRA, (f & XO_ECO_NO_CBF) ? 1 : 0);
        if(out) *out = s;
        return xo_sbac_bits(&s);
    }
    xo_sbac_bit_reset(&s);
    if(j->mode == XO_BITS_CU_SKIP) { /* xeve_mode.c:276-295 */
        if(p->slice_type != 2) {
            xo_sbac_bin(&s, XO_CTX_SKIP_FLAG + j->ctx_skip, 1);
            sbac_mvp_idx(&s, j->mvp_idx[0]);
            if(p->slice_type == 0) sbac_mvp_idx(&s, j->mvp_idx[1]);
        }
    }
    else if(j->mode == XO_BITS_CU_INTER) { /* xeve_mode.c:201-274 */
        static const int run_all[3] = {1, 1, 1};
        if(p->slice_type != 2) {
            xo_sbac_bin(&s, XO_CTX_SKIP_FLAG + j->ctx_skip, 0);
            xo_sbac_bin(&s, XO_CTX_PRED_MODE + j->ctx_pred_mode, 0); /* xeve_eco_pred_mode(MODE_INTER) */
            xo_sbac_bin(&s, XO_CTX_DIRECT, j->dir_flag);
            if(!j->dir_flag) {
                /* xeve_eco_inter_pred_idc (xeve_eco.c:1123-1156); check_bi_applicability == slice B without admvp */
                int v0 = j->refi[0] >= 0, v1 = j->refi[1] >= 0;
                if(v0 && v1) xo_sbac_bin(&s, XO_CTX_INTER_DIR, 0);
                else {
                    if(p->slice_type == 0) xo_sbac_bin(&s, XO_CTX_INTER_DIR, 1);
                    xo_sbac_bin(&s, XO_CTX_INTER_DIR + 1, v0 ? 0 : 1);
                }
                if(v0) {
                    sbac_refi(&s, p->num_refp[0], j->refi[0]);
                    sbac_mvp_idx(&s, j->mvp_idx[0]);
                    sbac_mvd1(&s, j->mvd[0][0]), sbac_mvd1(&s, j->mvd[0][1]);
                }
                if(p->slice_type == 0 && v1) {
                    sbac_refi(&s, p->num_refp[1], j->refi[1]);
                    sbac_mvp_idx(&s, j->mvp_idx[1]);
                    sbac_mvd1(&s, j->mvd[1][0]), sbac_mvd1(&s, j->mvd[1][1]);
                }
            }
        }
        sbac_coef(&s, p, j, coef, run_all, 0, 0);
    }
    else if(j->mode == XO_BITS_CU_INTRA || j->mode == XO_BITS_INTRA_LUMA) {
        /* xeve_rdo_bit_cnt_cu_intra (xeve_mode.c:141-175) / xeve_rdo_bit_cnt_cu_intra_luma (:81-117): Baseline (tool_admvp 0, xeve_check_all_preds true,
         * no fn_rdo_intra_ext, no delta QP): outside I slices skip_flag = 0 and pred_mode = MODE_INTRA, the prediction mode as the unary index
         * mpm[ipm] over the two intra_dir models (xeve_eco_intra_dir, xeve_eco.c:1104-1121), the cbf flags and coefficients of an intra CU */
        static const int run_all[3] = {1, 1, 1}, run_y[3] = {1, 0, 0};
        if(p->slice_type != 2) {
            xo_sbac_bin(&s, XO_CTX_SKIP_FLAG + j->ctx_skip, 0);
            xo_sbac_bin(&s, XO_CTX_PRED_MODE + j->ctx_pred_mode, 1);
        }
        sbac_unary2(&s, j->mvp_idx[0], XO_CTX_INTRA_DIR);
        sbac_coef(&s, p, j, coef, j->mode == XO_BITS_CU_INTRA ? run_all : run_y, 1, 0);
    }
    else if(j->mode == XO_BITS_INTRA_DIR) sbac_unary2(&s, j->mvp_idx[0], XO_CTX_INTRA_DIR); /* xeve_rdo_bit_cnt_intra_dir (xeve_mode.c:136-139) */
    else if(j->mode == XO_BITS_MVP) { /* xeve_rdo_bit_cnt_mvp (xeve_mode.c:57-79), pidx != PRED_DIR */
        if(p->slice_type != 2 && j->refi[0] >= 0) sbac_mvp_idx(&s, j->mvp_idx[0]), sbac_mvd1(&s, j->mvd[0][0]), sbac_mvd1(&s, j->mvd[0][1]);
        if(p->slice_type == 0 && j->refi[1] >= 0) sbac_mvp_idx(&s, j->mvp_idx[1]), sbac_mvd1(&s, j->mvd[1][0]), sbac_mvd1(&s, j->mvd[1][1]);
    }
    else { /* xeve_mode.c:177-199: one component, RUN_L / RUN_CB / RUN_CR */
        int run[3] = {j->mode == XO_BITS_COMP_Y, j->mode == XO_BITS_COMP_U, j->mode == XO_BITS_COMP_V};
        sbac_coef(&s, p, j, coef, run, 0, 0);
    }
    if(out) *out = s;
    return xo_sbac_bits(&s);
}

/* xeve_init_bits_est (xeve_mode.c:304-313) */
int32_t xo_entropy_bits(int i)
{
    double p = (512 * (i + 0.5)) / 1024;
    return (int32_t)(-32768 * (log(p) / log(2.0) - 9));
}
/* biari_no_bits (xeve_mode.c:315-324) */
static int32_t no_bits(int symbol, uint16_t cm)
{
    uint16_t mps = cm & 1, state = cm >> 1;
    state = ((uint16_t)(symbol != 0) != mps) ? state : (uint16_t)(512 - state);
    return xo_entropy_bits(state << 1);
}
void xo_rdoq_bit_est(const xo_sbac *s, xo_rdoq_est_full *e)
{   /* xeve_mode.c:326-372 (the tables the Baseline run-level syntax uses) */
    for(int b = 0; b < 2; b++) {
        e->cbf_luma[b] = no_bits(b, s->ctx[XO_CTX_CBF_LUMA]), e->cbf_cb[b] = no_bits(b, s->ctx[XO_CTX_CBF_CB]);
        e->cbf_cr[b] = no_bits(b, s->ctx[XO_CTX_CBF_CR]), e->cbf_all[b] = no_bits(b, s->ctx[XO_CTX_CBF_ALL]);
        for(int c = 0; c < 24; c++) e->run[c][b] = no_bits(b, s->ctx[XO_CTX_RUN + c]), e->level[c][b] = no_bits(b, s->ctx[XO_CTX_LEVEL + c]);
        for(int c = 0; c < 2; c++) e->last[c][b] = no_bits(b, s->ctx[XO_CTX_LAST + c]);
    }
}
void xo_rdoq_est_select(const xo_rdoq_est_full *f, int ch_type, int is_intra, xo_rdoq_est *e)
{
    const int32_t *cbf = (!is_intra && ch_type == 0) ? f->cbf_all : ch_type == 0 ? f->cbf_luma : ch_type == 1 ? f->cbf_cb : f->cbf_cr;
    e->cbf[0] = cbf[0], e->cbf[1] = cbf[1];
    memcpy(e->run, f->run, sizeof(e->run));
    memcpy(e->level, f->level, sizeof(e->level));
    memcpy(e->last, f->last, sizeof(e->last));
}

/* ===================================================================================================================
 * Deblocking filter and picture padding -- reference: src_base/xeve_df.c, src_base/xeve_util.c:190-248
 * =================================================================================================================== */
const uint8_t xo_df_st[4][52] = { /* xeve_tbl.c:239-257: intra; luma cbf; mv difference >= 4 / other reference; no filtering */
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 12, 12, 12, 12},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 11, 11, 11, 11, 11},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10, 10},
    {0}};

#define SCU_IF(m)   (((m) >> 15) & 1)
#define SCU_QP(m)   (((m) >> 16) & 0x7F)
#define SCU_CBFL(m) (((m) >> 24) & 1)
#define SCU_IBC(m)  (((m) >> 26) & 1)
#define SCU_COD(m)  (((m) >> 31) & 1)

/* get_tbl_qp_to_st (xeve_df.c:34-87): filter strength class of the edge between two 4x4 units */
static int df_class(uint32_t m0, uint32_t m1, const int8_t *r0, const int8_t *r1, const int16_t *mv0, const int16_t *mv1)
{
    if(SCU_IF(m0) || SCU_IF(m1)) return 0;
    if(SCU_CBFL(m0) == 1 || SCU_CBFL(m1) == 1) return 1;
    if(SCU_IBC(m0) || SCU_IBC(m1)) return 2;
    int a0[2] = {mv0[0], mv0[1]}, a1[2] = {mv0[2], mv0[3]}, b0[2] = {mv1[0], mv1[1]}, b1[2] = {mv1[2], mv1[3]};
    if(r0[0] < 0) a0[0] = a0[1] = 0;
    if(r0[1] < 0) a1[0] = a1[1] = 0;
    if(r1[0] < 0) b0[0] = b0[1] = 0;
    if(r1[1] < 0) b1[0] = b1[1] = 0;
    if(r0[0] == r1[0] && r0[1] == r1[1])
        return (iabs(a0[0] - b0[0]) >= 4 || iabs(a0[1] - b0[1]) >= 4 || iabs(a1[0] - b1[0]) >= 4 || iabs(a1[1] - b1[1]) >= 4) ? 2 : 3;
    if(r0[0] == r1[1] && r0[1] == r1[0])
        return (iabs(a0[0] - b1[0]) >= 4 || iabs(a0[1] - b1[1]) >= 4 || iabs(a1[0] - b0[0]) >= 4 || iabs(a1[1] - b0[1]) >= 4) ? 2 : 3;
    return 2;
}

/* deblock_scu_hor / _ver and their chroma forms (xeve_df.c:89-251): `n` samples along the edge (step `along`), the four
 * samples A B | C D across it at -2, -1, 0, +1 times `across`; s16 arithmetic as the reference */
static void df_edge(xo_pel *buf, int n, int along, int across, int st, int maxv, int chroma)
{
    if(!st) return;
    for(int i = 0; i < n; i++, buf += along) {
        int16_t A = buf[-2 * across], B = buf[-across], C = buf[0], D = buf[across];
        int16_t d = (int16_t)((A - (B << 2) + (C << 2) - D) / 8);
        int16_t abs = (int16_t)((d ^ (d >> 15)) - (d >> 15)), sign = d < 0;
        int16_t t16 = (int16_t)(((abs - st) << 1) > 0 ? ((abs - st) << 1) : 0);
        int16_t clip = (int16_t)((abs - t16) > 0 ? (abs - t16) : 0);
        int16_t d1 = (int16_t)(sign ? -clip : clip);
        if(!chroma) {
            clip >>= 1;
            int16_t d2 = (int16_t)clip3i(-clip, clip, (A - D) / 4);
            A = (int16_t)(A - d2), D = (int16_t)(D + d2);
            buf[-2 * across] = (xo_pel)clip3i(0, maxv, A), buf[across] = (xo_pel)clip3i(0, maxv, D);
        }
        B = (int16_t)(B + d1), C = (int16_t)(C - d1);
        buf[-across] = (xo_pel)clip3i(0, maxv, B), buf[0] = (xo_pel)clip3i(0, maxv, C);
    }
}

typedef struct df_ctx {
    xo_pel *y, *u, *v;
    int s_l, s_c, ws, hs;
    uint32_t *map_scu;
    const uint32_t *map_cu_mode;
    const int8_t *refi;
    const int16_t *mv;
    const xo_deblock_params *p;
    const uint8_t *tidx; /* ctx->map_tidx or NULL (one tile) */
} df_ctx;

/* no_boundary of xeve_deblock_cu_hor / _ver (xeve_df.c:296-302,386-391,419-427) with boundary_filtering = 0 (xeve_df.c:528): the two units share a tile */
static int df_same_tile(const df_ctx *c, int a, int b) { return !c->tidx || c->tidx[a] == c->tidx[b]; }

/* one 4-sample edge segment: luma + both chroma planes; `cur` = the unit whose QP is used, `nb` = the unit across the edge */
static void df_segment(const df_ctx *c, int cur, int nb, int x, int y, int hor)
{
    const xo_deblock_params *p = c->p;
    int cls = df_class(c->map_scu[cur], c->map_scu[nb], c->refi + 2 * cur, c->refi + 2 * nb, c->mv + 4 * cur, c->mv + 4 * nb);
    int qp = SCU_QP(c->map_scu[cur]), bl = p->bit_depth_luma - 8, bc = p->bit_depth_chroma - 8;
    xo_pel *py = c->y + y * c->s_l + x;
    if(hor) df_edge(py, 4, 1, c->s_l, xo_df_st[cls][qp] << bl, (1 << p->bit_depth_luma) - 1, 0);
    else df_edge(py, 4, c->s_l, 1, xo_df_st[cls][qp] << bl, (1 << p->bit_depth_luma) - 1, 0);
    if(p->chroma_format_idc) {
        int qu = clip3i(-6 * bc, 57, qp + p->qp_u_offset), qv = clip3i(-6 * bc, 57, qp + p->qp_v_offset);
        int off = (y >> c->hs) * c->s_c + (x >> c->ws);
        int st_u = xo_df_st[cls][p->qp_chroma[0][qu + 6 * bc]] << bc, st_v = xo_df_st[cls][p->qp_chroma[1][qv + 6 * bc]] << bc;
        /* the reference sizes the chroma segment with the W shift for horizontal and the H shift for vertical edges (xeve_df.c:143,225) */
        if(hor) df_edge(c->u + off, 4 >> c->ws, 1, c->s_c, st_u, (1 << p->bit_depth_chroma) - 1, 1), df_edge(c->v + off, 4 >> c->ws, 1, c->s_c, st_v, (1 << p->bit_depth_chroma) - 1, 1);
        else df_edge(c->u + off, 4 >> c->hs, c->s_c, 1, st_u, (1 << p->bit_depth_chroma) - 1, 1), df_edge(c->v + off, 4 >> c->hs, c->s_c, 1, st_v, (1 << p->bit_depth_chroma) - 1, 1);
    }
}

/* xeve_deblock_cu_hor (xeve_df.c:253-333) / xeve_deblock_cu_ver (:335-471) */
static void df_cu(const df_ctx *c, int x, int y, int cuw, int cuh, int hor)
{
    int w_scu = c->p->w_scu, t = (x >> 2) + (y >> 2) * w_scu, w = cuw >> 2, h = cuh >> 2;
    if(hor) {
        if(y > 0 && df_same_tile(c, t, t - w_scu))
            for(int i = 0; i < w; i++) df_segment(c, t + i, t + i - w_scu, x + 4 * i, y, 1);
    }
    else {
        if(x > 0 && SCU_COD(c->map_scu[t - 1]) && df_same_tile(c, t, t - 1))
            for(int i = 0; i < h; i++) df_segment(c, t + i * w_scu, t + i * w_scu - 1, x, y + 4 * i, 0);
        if(x + cuw < c->p->w && SCU_COD(c->map_scu[t + w]) && df_same_tile(c, t, t + w)) /* right neighbour already filtered in this pass */
            for(int i = 0; i < h; i++) df_segment(c, t + i * w_scu + w, t + i * w_scu + w - 1, x + cuw, y + 4 * i, 0);
    }
    for(int j = 0; j < h; j++)
        for(int i = 0; i < w; i++) c->map_scu[t + j * w_scu + i] |= 1u << 31;
}

/* xeve_deblock_tree (xeve_df.c:575-639): quad-tree walk; a node is a leaf when the CU recorded at its origin has its size */
static void df_tree(const df_ctx *c, int x, int y, int size, int hor)
{
    int t = (x >> 2) + (y >> 2) * c->p->w_scu;
    int logw = (c->map_cu_mode[t] >> 24) & 0xF;
    if((1 << logw) < size) {
        int hs = size >> 1;
        for(int k = 0; k < 4; k++) {
            int xs = x + (k & 1) * hs, ys = y + (k >> 1) * hs;
            if(xs < c->p->w && ys < c->p->h) df_tree(c, xs, ys, hs, hor);
        }
    }
    else df_cu(c, x, y, size, size, hor);
}

void xo_deblock_picture(xo_pel *y, xo_pel *u, xo_pel *v, int s_l, int s_c, uint32_t *map_scu, const uint32_t *map_cu_mode,
                        const int8_t *map_refi, const int16_t *map_mv, const xo_deblock_params *p)
{
    xo_deblock_picture_tiles(y, u, v, s_l, s_c, map_scu, map_cu_mode, NULL, map_refi, map_mv, p);
}

/* ... with the tile map (ctx->map_tidx): the reference filters tile after tile (xeve_loop_filter -> xeve_deblock per tile, xeve_enc.c:2355-2415), each in
 * raster CTU order; since no edge between two tiles is ever filtered the tiles are independent and one raster walk over the picture gives the same planes */
void xo_deblock_picture_tiles(xo_pel *y, xo_pel *u, xo_pel *v, int s_l, int s_c, uint32_t *map_scu, const uint32_t *map_cu_mode, const uint8_t *map_tidx,
                              const int8_t *map_refi, const int16_t *map_mv, const xo_deblock_params *p)
{
    df_ctx c = {y, u, v, s_l, s_c, p->chroma_format_idc <= 2, p->chroma_format_idc <= 1, map_scu, map_cu_mode, map_refi, map_mv, p, map_tidx};
    int ctu = 1 << p->log2_max_cuwh;
    for(int hor = 0; hor <= 1; hor++) { /* xeve_loop_filter: vertical edges of the whole picture first */
        for(int i = 0; i < p->w_scu * p->h_scu; i++) map_scu[i] &= 0x7FFFFFFFu;
        for(int cy = 0; cy < p->h; cy += ctu)
            for(int cx = 0; cx < p->w; cx += ctu) df_tree(&c, cx, cy, ctu, hor);
    }
}

/* ===================================================================================================================
 * calc_delta_dist_filter_boundary (src_base/xeve_mode.c:1534-2005): rdo_dbk_switch = 1 (preset slow).  What the loop filter will do to a candidate's
 * reconstruction enters its distortion: the candidate block `src` is laid into a scratch picture with the 4 rows above and the 4 columns to its left of the
 * reconstruction so far (PIC_MODE, not yet filtered), its TOP edge is filtered, then its LEFT edge (xeve_deblock_unit(.., is_hor_edge 1) first, :1852, then 0,
 * :1864 -- the reverse of the picture's own order), and delta = SSD after - SSD before over the block, 2 rows above it and 2 columns to its left (chroma: 1).
 * The current side of an edge takes the candidate's flags (intra, luma cbf, motion), the far side what the unit maps hold: flags and motion of decided CUs,
 * the luma cbf flag only of CTUs the WRITER has been through (xeve_eco_unit sets it, xeve_eco.c:1591; copy_to_cu_data never does).  The scratch picture's own
 * chroma qp offsets are zero (a zeroed XEVE_PIC nobody sets them on, xeve_enc.c:1201-1203).  One tile; the right neighbour is never coded before the CU
 * (avail_lr is LR_00 or LR_10).  The reference leaves its maps altered (the CU's motion and QP fields; flags restored from the top-left unit): nothing reads
 * them before the CU's decision rewrites them, so this restatement has no side effect.
 * =================================================================================================================== */
static __thread xo_dbk_ctx g_dbk; /* set around a CTU's analysis (xo_rdo_dbk_begin / _end); on == 0: rdo_dbk_switch 0 */
void xo_rdo_dbk_begin(const xo_dbk_ctx *c) { g_dbk = *c, g_dbk.on = 1; }
void xo_rdo_dbk_end(void) { g_dbk.on = 0; }
int  xo_rdo_dbk_on(void) { return g_dbk.on; }

void xo_delta_dist(const xo_dbk_ctx *D, const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_pel *const src[3], int x, int y, int cuw, int cuh,
                   int intra_flag, int cbf_l, const int8_t refi[2], const int16_t mv[2][2], int64_t delta[3])
{
    const xo_deblock_params *p = D->dp;
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, w_scu = p->w_scu, t = (x >> 2) + (y >> 2) * w_scu;
    const int top = y > 0 && (!D->map_tidx || D->map_tidx[t] == D->map_tidx[t - w_scu]);
    const int left = x > 0 && SCU_COD(D->map_scu[t - 1]) && (!D->map_tidx || D->map_tidx[t] == D->map_tidx[t - 1]);
    const int bl = p->bit_depth_luma - 8, bc = p->bit_depth_chroma - 8, qp = D->qp;
    /* the current side of every edge: the candidate's flags and motion (xeve_mode.c:1800-1843) */
    const uint32_t m_cur = ((uint32_t)(intra_flag != 0) << 15) | ((uint32_t)qp << 16) | ((uint32_t)(cbf_l != 0) << 24);
    int8_t  r_cur[2] = {-1, -1};
    int16_t v_cur[4] = {0, 0, 0, 0};
    if(refi) r_cur[0] = refi[0], r_cur[1] = refi[1], v_cur[0] = mv[0][0], v_cur[1] = mv[0][1], v_cur[2] = mv[1][0], v_cur[3] = mv[1][1];
    /* The filter compares ctx->map_unrefined_mv, not map_mv (xeve_deblock_unit, xeve_df.c:491,509).  The picture's own filter pass copies map_mv into it first
     * (xeve_deblock, :545-549); during the mode decision it holds what update_map_scu (xeve_mode.c:1102) copies out of cu_data->unrefined_mv -- after every decided
     * CU and once more at the end of update_to_ctx_map (:2518), over what that function had just stored -- and nothing in the Baseline encoder ever writes that array:
     * it is zero.  So every neighbour compares as MOTIONLESS (its reference indices are real); only the candidate's own side carries vectors (:1827-1830). */
    static const int16_t zero4[4] = {0, 0, 0, 0};
    const int top_in_ctu = 1, left_in_ctu = 1;
    delta[0] = delta[1] = delta[2] = 0;
    for(int c = 0; c < (idc ? 3 : 1); c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, w = cuw >> sx, h = cuh >> sy, xo_ = 4 >> sx, yo = 4 >> sy, xt = 2 >> sx, yt = 2 >> sy, bd = c ? p->bit_depth_chroma : p->bit_depth_luma;
        const int S = w + xo_, so = c ? s_org_c : s_org_l, sm = c ? D->s_mod_c : D->s_mod_l, xc = x >> sx, yc = y >> sy, maxv = (1 << bd) - 1;
        if(!src[c]) continue; /* (the luma-only call of the intra analysis: the reference copies stale chroma there and never reads the result) */
        xo_pel *buf = calloc((size_t)S * (h + yo), sizeof(xo_pel)), *dst = buf + (size_t)yo * S + xo_;
        const xo_pel *o = org[c] + (size_t)yc * so + xc, *rec = D->mod[c] + (size_t)yc * sm + xc;
        for(int i = 0; i < h; i++) memcpy(dst + (size_t)i * S, src[c] + (size_t)i * w, sizeof(xo_pel) * (size_t)w);
        if(top)
            for(int i = 0; i < yo; i++) memcpy(dst + (ptrdiff_t)(i - yo) * S, rec + (ptrdiff_t)(i - yo) * sm, sizeof(xo_pel) * (size_t)w);
        if(left)
            for(int i = 0; i < h; i++) memcpy(dst + (size_t)i * S - xo_, rec + (size_t)i * sm - xo_, sizeof(xo_pel) * (size_t)xo_);
        int64_t before = xo_ssd(w, h, dst, o, S, so, bd);
        if(top) before += xo_ssd(w, yt, dst - (ptrdiff_t)yt * S, o - (ptrdiff_t)yt * so, S, so, bd);
        if(left) before += xo_ssd(xt, h, dst - xt, o - xt, S, so, bd);
        /* the top edge, then the left edge: 4-sample segments (chroma: 4 >> shift), the strength from the two units across the segment */
        if(top)
            for(int i = 0; i < cuw >> 2; i++) {
                const int nb = t + i - w_scu;
                const int cls = df_class(m_cur, D->map_scu[nb], r_cur, D->map_refi + 2 * nb, v_cur, top_in_ctu ? zero4 : D->map_mv + 4 * nb);
                if(c == 0) df_edge(dst + 4 * i, 4, 1, S, xo_df_st[cls][qp] << bl, maxv, 0);
                else {
                    const int q = clip3i(-6 * bc, 57, qp);
                    df_edge(dst + ((4 * i) >> ws), 4 >> ws, 1, S, xo_df_st[cls][p->qp_chroma[c - 1][q + 6 * bc]] << bc, maxv, 1);
                }
            }
        if(left)
            for(int i = 0; i < cuh >> 2; i++) {
                const int nb = t + i * w_scu - 1;
                const int cls = df_class(m_cur, D->map_scu[nb], r_cur, D->map_refi + 2 * nb, v_cur, left_in_ctu ? zero4 : D->map_mv + 4 * nb);
                if(c == 0) df_edge(dst + (size_t)(4 * i) * S, 4, S, 1, xo_df_st[cls][qp] << bl, maxv, 0);
                else {
                    const int q = clip3i(-6 * bc, 57, qp);
                    df_edge(dst + (size_t)((4 * i) >> hs) * S, 4 >> hs, S, 1, xo_df_st[cls][p->qp_chroma[c - 1][q + 6 * bc]] << bc, maxv, 1);
                }
            }
        int64_t after = xo_ssd(w, h, dst, o, S, so, bd);
        if(top) after += xo_ssd(w, yt, dst - (ptrdiff_t)yt * S, o - (ptrdiff_t)yt * so, S, so, bd);
        if(left) after += xo_ssd(xt, h, dst - xt, o - xt, S, so, bd);
        delta[c] = after - before;
        free(buf);
    }
    if(getenv("XO_DBK_LOG")) { /* (debugging aid: the same line oracle/ref_shadow.c logs for the reference's own function) */
        FILE *f = fopen(getenv("XO_DBK_LOG"), "a");
        fprintf(f, "x %d y %d cu %d intra %d cbf %d refi %d %d mv %d %d %d %d lr %d -> %lld %lld %lld\n", x, y, cuw, intra_flag, cbf_l, refi ? refi[0] : -9, refi ? refi[1] : -9,
                mv ? mv[0][0] : 0, mv ? mv[0][1] : 0, mv ? mv[1][0] : 0, mv ? mv[1][1] : 0, left, src[0] ? (long long)delta[0] : -999999LL, src[1 % 3] && idc ? (long long)delta[1] : -999999LL, src[2] && idc ? (long long)delta[2] : -999999LL);
        fclose(f);
    }
}

void xo_picbuf_expand(xo_pel *a, int s, int w, int h, int exp)
{   /* xeve_util.c:190-238 */
    for(int i = 0; i < h; i++)
        for(int j = 0; j < exp; j++) a[i * s - exp + j] = a[i * s], a[i * s + w + j] = a[i * s + w - 1];
    for(int i = 0; i < exp; i++) {
        memcpy(a - exp - (i + 1) * s, a - exp, s * sizeof(xo_pel));
        memcpy(a + (h - 1) * s - exp + (i + 1) * s, a + (h - 1) * s - exp, s * sizeof(xo_pel));
    }
}

/* ===================================================================================================================
 * a8: xeve_mc (src_base/xeve_mc.c:465-610) -- clip, per-list interpolation of Y / U / V, bi-prediction average
 * =================================================================================================================== */
void xo_mc_cu(const xo_refpic *refp, int s_l, int s_c, int pic_w, int pic_h, const xo_cu_mc_job *job, int w, int h, int bit_depth_luma,
              int bit_depth_chroma, int chroma_format_idc, xo_pel *pred_y, xo_pel *pred_u, xo_pel *pred_v)
{
    const int ws = chroma_format_idc <= 2, hs = chroma_format_idc <= 1, wfac = 2 / (ws + 1), hfac = 2 / (hs + 1);
    const int cw = w >> ws, chh = h >> hs;
    int mvt[2][2], valid[2] = {job->refi[0] >= 0, job->refi[1] >= 0}, bidx = 0;
    xo_pel *p1[3] = {0, 0, 0};
    /* xeve_mv_clip (xeve_mc.c:401-447): the block may leave the picture by at most MAX_CU_SIZE (128) samples */
    for(int l = 0; l < 2; l++) {
        const int x4 = job->x << 2, y4 = job->y << 2, w4 = w << 2, h4 = h << 2;
        const int min_c = -(128 << 2), max_x = (pic_w - 1 + 128) << 2, max_y = (pic_h - 1 + 128) << 2;
        mvt[l][0] = job->mv[l][0], mvt[l][1] = job->mv[l][1];
        if(!valid[l]) continue;
        if(x4 + job->mv[l][0] < min_c) mvt[l][0] = (int16_t)(min_c - x4);
        if(y4 + job->mv[l][1] < min_c) mvt[l][1] = (int16_t)(min_c - y4);
        if(x4 + job->mv[l][0] + w4 - 4 > max_x) mvt[l][0] = (int16_t)(max_x - x4 - w4 + 4);
        if(y4 + job->mv[l][1] + h4 - 4 > max_y) mvt[l][1] = (int16_t)(max_y - y4 - h4 + 4);
    }
    for(int l = 0; l < 2; l++) {
        if(!valid[l]) continue;
        if(l == 1 && valid[0]) { /* identical motion: the second list adds nothing (xeve_mc.c:546-551) */
            const xo_refpic *r0 = &refp[job->refi[0] * 2], *r1 = &refp[job->refi[1] * 2 + 1];
            if(r0->poc == r1->poc && mvt[0][0] == mvt[1][0] && mvt[0][1] == mvt[1][1]) return;
        }
        const xo_refpic *r = &refp[job->refi[l] * 2 + l];
        const int gx = ((job->x << 2) + mvt[l][0]) << 2, gy = ((job->y << 2) + mvt[l][1]) << 2;
        /* the filter variant follows the UNCLIPPED vector's fraction, the position the clipped one (xeve_mc.c:490-507, xeve_mc.h:95-104) */
        const int ox = job->mv[l][0] << 2, oy = job->mv[l][1] << 2;
        xo_pel *py = pred_y, *pu = pred_u, *pv = pred_v;
        if(bidx == 1) {
            for(int c = 0; c < 3; c++) p1[c] = malloc(sizeof(xo_pel) * (size_t)(c ? cw * chh : w * h));
            py = p1[0], pu = p1[1], pv = p1[2];
        }
        xo_mc_l(ox & 0xF, oy & 0xF, r->y, gx, gy, s_l, w, py, w, h, bit_depth_luma, xo_mc_l_coeff);
        if(chroma_format_idc) {
            xo_mc_c(ox & 0x1F, oy & 0x1F, r->u, gx * wfac, gy * hfac, s_c, cw, pu, cw, chh, bit_depth_chroma, xo_mc_c_coeff);
            xo_mc_c(ox & 0x1F, oy & 0x1F, r->v, gx * wfac, gy * hfac, s_c, cw, pv, cw, chh, bit_depth_chroma, xo_mc_c_coeff);
        }
        bidx++;
    }
    if(bidx == 2) {
        xo_avg(pred_y, p1[0], pred_y, w, w, w, w, h);
        if(chroma_format_idc) xo_avg(pred_u, p1[1], pred_u, cw, cw, cw, cw, chh), xo_avg(pred_v, p1[2], pred_v, cw, cw, cw, cw, chh);
    }
    for(int c = 0; c < 3; c++) free(p1[c]);
}

/* ===================================================================================================================
 * pinter_residue_rdo (src_base/xeve_pinter.c:906-1336)
 * =================================================================================================================== */
void xo_residue_rdo(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                    const xo_rdo_params *p, const xo_rdo_job *job, xo_rdo_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                    xo_sbac *best)
{
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth;
    const int lw[3] = {p->log2_cuw, p->log2_cuw - ws, p->log2_cuw - ws}, lh[3] = {p->log2_cuh, p->log2_cuh - hs, p->log2_cuh - hs};
    const int ncomp = idc ? 3 : 1;
    int16_t  *coef[3] = {coef_y, coef_u, coef_v};
    xo_pel   *pred[3], *rec;
    int16_t  *tmp;
    int64_t   dist[2][3] = {{0, 0, 0}, {0, 0, 0}};
    int       nnz_store[3] = {0, 0, 0}, tnnz = 0;
    xo_rdoq_est_full full;

    for(int c = 0; c < 3; c++) pred[c] = malloc(sizeof(xo_pel) << (lw[0] + lh[0]));
    rec = malloc(sizeof(xo_pel) << (lw[0] + lh[0])), tmp = malloc(sizeof(int16_t) << (lw[0] + lh[0]));
    /* prediction (pi->fn_mc = pinter_mc -> xeve_mc, :962) */
    xo_cu_mc_job mj;
    memset(&mj, 0, sizeof(mj));
    mj.x = job->x, mj.y = job->y, memcpy(mj.mv, job->mv, sizeof(mj.mv)), mj.refi[0] = job->refi[0], mj.refi[1] = job->refi[1];
    xo_mc_cu(refp, s_l, s_c, p->pic_w, p->pic_h, &mj, 1 << lw[0], 1 << lh[0], bd, bd, idc, pred[0], pred[1], pred[2]);
    /* the estimates RDOQ reads: xeve_rdoq_bit_est(&core->s_curr_best[..]) (xeve_mode.c:792) */
    xo_rdoq_bit_est(&states[job->sbac], &full);
    /* residual, SSD of the prediction, transform + quantisation (:969-998) */
    for(int c = 0; c < ncomp; c++) {
        const int w = 1 << lw[c], h = 1 << lh[c], so = c ? s_org_c : s_org_l;
        const xo_pel *o = org[c] + (c ? (job->y >> hs) * so + (job->x >> ws) : job->y * so + job->x);
        xo_diff(w, h, o, pred[c], so, w, w, coef[c]);
        dist[0][c] = xo_ssd(w, h, pred[c], o, w, so, bd);
        xo_trans(coef[c], lw[c], lh[c], bd);
        if(xo_rdoq_zero_test(coef[c], lw[c], lh[c], p->qp[c], xo_quant_scale[p->tool_iqt][p->qp[c] % 6], p->slice_type == 2, bd)) {
            xo_rdoq_est e;
            xo_rdoq_est_select(&full, c, 0, &e);
            nnz_store[c] = xo_rdoq(coef[c], lw[c], lh[c], p->qp[c], p->lambda[c], c == 0, bd, p->tool_iqt, &e);
        }
        else memset(coef[c], 0, sizeof(int16_t) << (lw[c] + lh[c])), nnz_store[c] = 0;
        tnnz += nnz_store[c];
    }
    for(int c = ncomp; c < 3; c++) dist[0][c] = 0;

    xo_cu_bits_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.log2_cuw = p->log2_cuw, bp.log2_cuh = p->log2_cuh, bp.slice_type = p->slice_type, bp.num_refp[0] = p->num_refp[0], bp.num_refp[1] = p->num_refp[1];
    bp.cm_init = 0, bp.chroma_format_idc = idc;
    xo_cu_bits_job bj;
    memset(&bj, 0, sizeof(bj));
    bj.coef_off[0] = bj.coef_off[1] = bj.coef_off[2] = 0; /* blocks are passed as separate buffers below */
    memcpy(bj.mvd, job->mvd, sizeof(bj.mvd)), bj.refi[0] = job->refi[0], bj.refi[1] = job->refi[1];
    bj.mvp_idx[0] = job->mvp_idx[0], bj.mvp_idx[1] = job->mvp_idx[1], bj.dir_flag = job->dir_flag, bj.ctx_skip = job->ctx_skip, bj.ctx_pred_mode = job->ctx_pred_mode;
    /* xo_cu_bits addresses Y / U / V through offsets into one buffer: lay the three blocks out back to back */
    const int n0 = 1 << (lw[0] + lh[0]), n1 = idc ? 1 << (lw[1] + lh[1]) : 0;
    int16_t *all = malloc(sizeof(int16_t) * (n0 + 2 * n1 + 1));
    bj.coef_off[1] = n0, bj.coef_off[2] = n0 + n1;
#define PACK() (memcpy(all, coef[0], sizeof(int16_t) * n0), (void)(idc ? (memcpy(all + n0, coef[1], sizeof(int16_t) * n1), memcpy(all + n0 + n1, coef[2], sizeof(int16_t) * n1)) : 0))
    PACK();
    const xo_sbac *entry = &states[job->sbac];
    xo_sbac run;
    double cost, cost_best = 1.7e+308; /* MAX_COST */
    int    cbf_idx[3] = {0, 0, 0}, nnz[3];
#define FULL_BITS(N0, N1, N2) (bj.mode = XO_BITS_CU_INTER, bj.sbac = 0, bj.nnz[0] = (N0), bj.nnz[1] = (N1), bj.nnz[2] = (N2), xo_cu_bits(entry, &run, &bp, &bj, all))
#define SUM_COST(IY, IU, IV) ((double)dist[IY][0] + (((double)dist[IU][1] * p->dist_chroma_weight[0]) + ((double)dist[IV][2] * p->dist_chroma_weight[1])))
    /* rdo_dbk_switch (:1016-1095, 1290-1312): what the loop filter will do to the CU's left / top boundary, for the prediction alone (no luma cbf) and for the
     * reconstruction (luma cbf as quantised) */
    const int64_t dist_no_resi[3] = {dist[0][0], dist[0][1], dist[0][2]};
    const int     dbk = xo_rdo_dbk_on();
    if(dbk) {
        int64_t d[3];
        const xo_pel *const src[3] = {pred[0], idc ? pred[1] : NULL, idc ? pred[2] : NULL};
        xo_delta_dist(&g_dbk, org, s_org_l, s_org_c, src, job->x, job->y, 1 << lw[0], 1 << lh[0], 0, 0, job->refi, job->mv, d);
        for(int c = 0; c < ncomp; c++) dist[0][c] += d[c];
    }
    if(tnnz) {
        /* reconstruct what was quantised (:1000-1051): dist[1] */
        xo_pel *recs[3] = {NULL, NULL, NULL};
        for(int c = 0; c < ncomp && dbk; c++) recs[c] = malloc(sizeof(xo_pel) << (lw[c] + lh[c]));
        for(int c = 0; c < ncomp; c++) {
            if(!nnz_store[c]) {
                dist[1][c] = dist_no_resi[c];
                if(dbk) memcpy(recs[c], pred[c], sizeof(xo_pel) << (lw[c] + lh[c])); /* "complete rec" (:1061-1076) */
                continue;
            }
            const int w = 1 << lw[c], h = 1 << lh[c], so = c ? s_org_c : s_org_l;
            const xo_pel *o = org[c] + (c ? (job->y >> hs) * so + (job->x >> ws) : job->y * so + job->x);
            memcpy(tmp, coef[c], sizeof(int16_t) << (lw[c] + lh[c]));
            xo_dquant(tmp, lw[c], lh[c], xo_dq_scale[p->qp[c] % 6] << (p->qp[c] / 6), bd);
            xo_itrans(tmp, lw[c], lh[c], bd);
            xo_recon(tmp, pred[c], nnz_store[c], w, h, w, rec, bd);
            dist[1][c] = xo_ssd(w, h, rec, o, w, so, bd);
            if(dbk) memcpy(recs[c], rec, sizeof(xo_pel) << (lw[c] + lh[c]));
        }
        for(int c = ncomp; c < 3; c++) dist[1][c] = 0;
        if(dbk) {
            int64_t d[3];
            const xo_pel *const src[3] = {recs[0], recs[1], recs[2]};
            xo_delta_dist(&g_dbk, org, s_org_l, s_org_c, src, job->x, job->y, 1 << lw[0], 1 << lh[0], 0, nnz_store[0] != 0, job->refi, job->mv, d);
            for(int c = 0; c < ncomp; c++) dist[1][c] += d[c], free(recs[c]);
        }
        if(!job->dir_flag) { /* all-zero alternative (:1103-1142) */
            cost = SUM_COST(0, 0, 0);
            cost += (double)(int)FULL_BITS(0, 0, 0) * p->lambda[0];
            if(cost < cost_best) cost_best = cost, cbf_idx[0] = cbf_idx[1] = cbf_idx[2] = 0, *best = run;
        }
        /* as quantised (:1144-1178) */
        int iy = nnz_store[0] > 0, iu = nnz_store[1] > 0, iv = nnz_store[2] > 0;
        cost = SUM_COST(iy, iu, iv);
        cost += (double)(int)FULL_BITS(nnz_store[0], nnz_store[1], nnz_store[2]) * p->lambda[0];
        if(cost < cost_best) cost_best = cost, cbf_idx[0] = iy, cbf_idx[1] = iu, cbf_idx[2] = iv, *best = run;
        /* each component with / without its coefficients, the coder state handed on (:1180-1218) */
        xo_sbac prev_best = *entry, prev_run;
        int     idx_best[3] = {0, 0, 0};
        nnz[0] = nnz_store[0], nnz[1] = nnz_store[1], nnz[2] = nnz_store[2];
        for(int i = 0; i < 3; i++) {
            if(nnz_store[i] <= 0) continue;
            double comp_best = 1.7e+308;
            prev_run = prev_best;
            for(int j = 0; j < 2; j++) {
                cost = (double)dist[j][i] * (i == 0 ? 1 : p->dist_chroma_weight[i - 1]);
                nnz[i] = j ? nnz_store[i] : 0;
                bj.mode = (uint8_t)(XO_BITS_COMP_Y + i), bj.sbac = 0, bj.nnz[0] = nnz[0], bj.nnz[1] = nnz[1], bj.nnz[2] = nnz[2];
                cost += (double)(int)xo_cu_bits(&prev_run, &run, &bp, &bj, all) * p->lambda[i];
                if(cost < comp_best) comp_best = cost, idx_best[i] = j, prev_best = run;
            }
        }
        if(idx_best[0] || idx_best[1] || idx_best[2]) {
            iy = idx_best[0], iu = idx_best[1], iv = idx_best[2];
            nnz[0] = iy ? nnz_store[0] : 0, nnz[1] = iu ? nnz_store[1] : 0, nnz[2] = iv ? nnz_store[2] : 0;
        }
        if(nnz[0] != nnz_store[0] || nnz[1] != nnz_store[1] || nnz[2] != nnz_store[2]) { /* the combination the component tests chose (:1220-1262) */
            cost = SUM_COST(iy, iu, iv);
            cost += (double)(int)FULL_BITS(nnz[0], nnz[1], nnz[2]) * p->lambda[0];
            if(cost < cost_best) cost_best = cost, cbf_idx[0] = iy, cbf_idx[1] = iu, cbf_idx[2] = iv, *best = run;
        }
        for(int c = 0; c < 3; c++) {
            res->nnz[c] = cbf_idx[c] ? nnz_store[c] : 0;
            if(res->nnz[c] == 0 && nnz_store[c] != 0) memset(coef[c], 0, sizeof(int16_t) << (lw[c] + lh[c]));
        }
    }
    else { /* nothing survived quantisation (:1276-1331) */
        dist[1][0] = dist[1][1] = dist[1][2] = 0;
        cost_best = (double)dist[0][0] + (p->dist_chroma_weight[0] * (double)dist[0][1]) + (p->dist_chroma_weight[1] * (double)dist[0][2]);
        cost_best += (double)(int)FULL_BITS(0, 0, 0) * p->lambda[0];
        *best = run;
        res->nnz[0] = res->nnz[1] = res->nnz[2] = 0;
    }
    res->cost = cost_best, res->pad_ = 0;
    memcpy(res->dist, dist, sizeof(dist));
    free(all), free(tmp), free(rec);
    for(int c = 0; c < 3; c++) free(pred[c]);
#undef PACK
#undef FULL_BITS
#undef SUM_COST
}

/* ===================================================================================================================
 * xeve_analyze_skip (src_base/xeve_pinter.c:1337-1530), rdo_dbk_switch 0
 * =================================================================================================================== */
void xo_analyze_skip(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                     const xo_rdo_params *p, const xo_skip_job *job, xo_skip_result *res, xo_pel *pred_y, xo_pel *pred_u, xo_pel *pred_v, xo_sbac *best)
{
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth;
    const int w = 1 << p->log2_cuw, h = 1 << p->log2_cuh, cw = w >> ws, ch = h >> hs, isb = p->slice_type == 0;
    xo_pel *t[3];
    double  cost_best = 1.7e+308;
    for(int c = 0; c < 3; c++) t[c] = malloc(sizeof(xo_pel) * (size_t)w * h);
    memset(res, 0, sizeof(*res));
    res->best_ssd = (int64_t)1 << (p->log2_cuw + p->log2_cuh + 16);
    xo_cu_bits_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.log2_cuw = p->log2_cuw, bp.log2_cuh = p->log2_cuh, bp.slice_type = p->slice_type, bp.num_refp[0] = p->num_refp[0], bp.num_refp[1] = p->num_refp[1];
    bp.chroma_format_idc = idc;
    for(int i0 = 0; i0 < job->ncand; i0++) {
        int dup = 0;
        for(int k = i0 - 1; k >= 0 && !dup; k--) dup = job->mvp[0][k][0] == job->mvp[0][i0][0] && job->mvp[0][k][1] == job->mvp[0][i0][1];
        if(dup) continue; /* encoder side pruning (:1396-1409) */
        const int cnt = isb ? job->ncand : 1;
        for(int i1 = 0; i1 < cnt; i1++) {
            dup = 0;
            for(int k = i1 - 1; k >= 0 && !dup; k--) dup = job->mvp[1][k][0] == job->mvp[1][i1][0] && job->mvp[1][k][1] == job->mvp[1][i1][1];
            if(dup) continue;
            xo_cu_mc_job mj;
            memset(&mj, 0, sizeof(mj));
            mj.x = job->x, mj.y = job->y;
            mj.mv[0][0] = job->mvp[0][i0][0], mj.mv[0][1] = job->mvp[0][i0][1], mj.mv[1][0] = job->mvp[1][i1][0], mj.mv[1][1] = job->mvp[1][i1][1];
            mj.refi[0] = job->refi_pred[0][i0], mj.refi[1] = isb ? job->refi_pred[1][i1] : -1;
            if(mj.refi[0] < 0 && mj.refi[1] < 0) continue;
            xo_mc_cu(refp, s_l, s_c, p->pic_w, p->pic_h, &mj, w, h, bd, bd, idc, t[0], t[1], t[2]);
            int64_t cy = xo_ssd(w, h, t[0], org[0] + job->y * s_org_l + job->x, w, s_org_l, bd), cu = 0, cv = 0;
            if(idc) {
                const int off = (job->y >> hs) * s_org_c + (job->x >> ws);
                cu = xo_ssd(cw, ch, t[1], org[1] + off, cw, s_org_c, bd), cv = xo_ssd(cw, ch, t[2], org[2] + off, cw, s_org_c, bd);
            }
            const int64_t temp_ssd = cy + cu + cv; /* (pi->best_ssd: without the loop filter's share, :1461) */
            if(xo_rdo_dbk_on()) { /* (:1463-1485) */
                int64_t d[3];
                const xo_pel *const src[3] = {t[0], idc ? t[1] : NULL, idc ? t[2] : NULL};
                xo_delta_dist(&g_dbk, org, s_org_l, s_org_c, src, job->x, job->y, w, h, 0, 0, mj.refi, mj.mv, d);
                cy += d[0];
                if(idc) cu += d[1], cv += d[2];
            }
            double cost = (double)cy + (p->dist_chroma_weight[0] * (double)cu) + (p->dist_chroma_weight[1] * (double)cv);
            xo_cu_bits_job bj;
            memset(&bj, 0, sizeof(bj));
            bj.mode = XO_BITS_CU_SKIP, bj.mvp_idx[0] = (uint8_t)i0, bj.mvp_idx[1] = (uint8_t)i1, bj.ctx_skip = job->ctx_skip, bj.sbac = job->sbac;
            xo_sbac run;
            cost += (double)(int)xo_cu_bits(states, &run, &bp, &bj, (const int16_t *)t[0]) * p->lambda[0];
            if(cost < cost_best) {
                cost_best = cost;
                res->idx0 = i0, res->idx1 = i1, res->best_ssd = temp_ssd;
                memcpy(res->mv, mj.mv, sizeof(res->mv)), res->refi[0] = mj.refi[0], res->refi[1] = mj.refi[1];
                memcpy(pred_y, t[0], sizeof(xo_pel) * (size_t)w * h);
                if(idc) memcpy(pred_u, t[1], sizeof(xo_pel) * (size_t)cw * ch), memcpy(pred_v, t[2], sizeof(xo_pel) * (size_t)cw * ch);
                *best = run;
            }
        }
    }
    res->cost = cost_best;
    for(int c = 0; c < 3; c++) free(t[c]);
}

/* ===================================================================================================================
 * xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047), Baseline
 * =================================================================================================================== */
int xo_check_best_mvp(const xo_sbac *entry, int slice_type, const int8_t refi[2], int lidx, const int16_t mvp[4][2], const int16_t mv[2], int mvp_idx,
                      double lambda0, int16_t mvd[2])
{
    xo_cu_bits_params bp;
    xo_cu_bits_job    bj;
    memset(&bp, 0, sizeof(bp)), memset(&bj, 0, sizeof(bj));
    bp.log2_cuw = bp.log2_cuh = 3, bp.slice_type = slice_type; /* (the CU size does not enter this syntax) */
    bj.mode = XO_BITS_MVP, bj.refi[0] = refi[0], bj.refi[1] = refi[1];
#define MVP_COST(IDX) (bj.mvp_idx[0] = bj.mvp_idx[1] = (uint8_t)(IDX), bj.mvd[lidx][0] = (int16_t)(mv[0] - mvp[IDX][0]), bj.mvd[lidx][1] = (int16_t)(mv[1] - mvp[IDX][1]), \
                       (double)(int)xo_cu_bits(entry, NULL, &bp, &bj, NULL) * lambda0)
    const double best_cost = MVP_COST(mvp_idx); /* never updated below: the LAST index cheaper than the entry index wins (:1829-1831) */
    int best_idx = mvp_idx;
    for(int idx = 0; idx < 4; idx++) { /* ORG_MAX_NUM_MVP */
        int same = 0;
        for(int t = idx - 1; t >= 0 && !same; t--) same = mvp[idx][0] == mvp[t][0] && mvp[idx][1] == mvp[t][1];
        if(same) continue; /* encoder side pruning */
        if(MVP_COST(idx) < best_cost) best_idx = idx;
    }
#undef MVP_COST
    mvd[0] = (int16_t)(mv[0] - mvp[best_idx][0]), mvd[1] = (int16_t)(mv[1] - mvp[best_idx][1]);
    return best_idx;
}

static void pinter_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                              const xo_inter_params *P, const xo_inter_job *job, xo_inter_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                              xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *next_best, xo_pel *pred_y_best);
void xo_pinter_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                          const xo_inter_params *P, const xo_inter_job *job, xo_inter_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                          xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *next_best)
{
    pinter_analyze_cu(org, s_org_l, s_org_c, refp, s_l, s_c, states, P, job, res, coef_y, coef_u, coef_v, rec_y, rec_u, rec_v, next_best, NULL);
}
/* pred_y_best (may be NULL): mi->pred_y_best = the winner's luma prediction (:2034), which mode_check_intra prices the intra candidates against */
static void pinter_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_refpic *refp, int s_l, int s_c, const xo_sbac *states,
                              const xo_inter_params *P, const xo_inter_job *job, xo_inter_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                              xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *next_best, xo_pel *pred_y_best)
{
    enum { L0 = 0, L1 = 1, BI = 2, SKIP = 3, DIR = 4, NP = 5 };
    const xo_rdo_params *p = &P->rdo;
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth, isb = p->slice_type == 0;
    const int lw = p->log2_cuw, lh = p->log2_cuh, w = 1 << lw, h = 1 << lh, n0 = w * h, n1 = idc ? n0 >> (ws + hs) : 0, ncomp = idc ? 3 : 1;
    const int x = job->x, y = job->y;
    double  cost_inter[NP], cost_best = 1.7e+308, cost;
    int     best_idx = SKIP, cu_mode = -1;
    /* per mode: motion data, coefficients, nnz */
    int16_t mv[NP][2][2], mvd[NP][2][2];
    int8_t  refi[NP][2];
    uint8_t mvpi[NP][2];
    int     nnz[NP][3];
    int16_t *coef[NP][3];
    xo_pel  *pred_skip[3];
    memset(mv, 0, sizeof(mv)), memset(mvd, 0, sizeof(mvd)), memset(refi, 0, sizeof(refi)), memset(mvpi, 0, sizeof(mvpi)), memset(nnz, 0, sizeof(nnz));
    for(int m = 0; m < NP; m++) {
        cost_inter[m] = 1.7e+308;
        for(int c = 0; c < 3; c++) coef[m][c] = calloc((size_t)n0, sizeof(int16_t));
    }
    for(int c = 0; c < 3; c++) pred_skip[c] = calloc((size_t)n0, sizeof(xo_pel));
    xo_rdo_result rr;
    xo_sbac       st;

    /* skip mode (:1872-1883) */
    xo_skip_job    sj;
    xo_skip_result sr;
    memset(&sj, 0, sizeof(sj));
    sj.x = x, sj.y = y, memcpy(sj.mvp, job->mvp, sizeof(sj.mvp)), sj.ncand = P->max_cand, sj.sbac = job->sbac, sj.ctx_skip = job->ctx_skip;
    xo_analyze_skip(org, s_org_l, s_org_c, refp, s_l, s_c, states, p, &sj, &sr, pred_skip[0], pred_skip[1], pred_skip[2], &st);
    cost = cost_inter[SKIP] = sr.cost;
    memcpy(mv[SKIP], sr.mv, sizeof(sr.mv)), refi[SKIP][0] = sr.refi[0], refi[SKIP][1] = sr.refi[1], mvpi[SKIP][0] = (uint8_t)sr.idx0, mvpi[SKIP][1] = (uint8_t)sr.idx1;
    if(!isb) mv[SKIP][1][0] = mv[SKIP][1][1] = 0; /* (stale in the reference) */
    if(cost < cost_best) cu_mode = 2, best_idx = SKIP, cost_best = cost, *next_best = st;

    if(cu_mode == 2 && (double)sr.best_ssd > (double)((int64_t)1 << (lw + lh + (bd - 8) + (bd - 8))) * P->skip_th) { /* (:1885-1887) */
        xo_rdo_job rj;
        if(isb) { /* analyze_t_direct (:1534-1565) + xeve_get_mv_dir (xeve_util.c:619-650) */
            const int dpoc_co = refp[0 * 2 + 1].poc - P->col_list_poc0, dpoc_l0 = P->poc - refp[0 * 2 + 0].poc, dpoc_l1 = refp[0 * 2 + 1].poc - P->poc;
            if(dpoc_co != 0) {
                mv[DIR][0][0] = (int16_t)(dpoc_l0 * job->mv_col[0] / dpoc_co), mv[DIR][0][1] = (int16_t)(dpoc_l0 * job->mv_col[1] / dpoc_co);
                mv[DIR][1][0] = (int16_t)(-dpoc_l1 * job->mv_col[0] / dpoc_co), mv[DIR][1][1] = (int16_t)(-dpoc_l1 * job->mv_col[1] / dpoc_co);
            }
            memset(&rj, 0, sizeof(rj));
            rj.x = x, rj.y = y, memcpy(rj.mv, mv[DIR], sizeof(rj.mv)), rj.dir_flag = 1, rj.ctx_skip = job->ctx_skip, rj.ctx_pred_mode = job->ctx_pred_mode, rj.sbac = job->sbac;
            xo_residue_rdo(org, s_org_l, s_org_c, refp, s_l, s_c, states, p, &rj, &rr, coef[DIR][0], coef[DIR][1], coef[DIR][2], &st);
            memcpy(nnz[DIR], rr.nnz, sizeof(rr.nnz));
            cost = cost_inter[DIR] = rr.cost;
            if(cost < cost_best) cu_mode = 3, best_idx = DIR, cost_best = cost, *next_best = st;
        }
        /* motion search per list (:1906-1989) */
        int16_t  mv_scale[2][XO_MAX_REFP][2];
        int      mot_bits[2] = {0, 0};
        uint8_t  mvp_idx[2] = {0, 0};
        xo_epzs_params ep = P->me;
        for(int l = 0; l <= (isb ? 1 : 0); l++) {
            uint32_t best_mecost = 0xFFFFFFFFu;
            int      refi_temp = 0;
            mvp_idx[l] = mvpi[SKIP][l];
            for(int r = 0; r < p->num_refp[l]; r++) {
                ep.me.bi = 0, ep.me.extra_bits = 0, ep.me.refi_bits = P->refi_bits[l][r], ep.me.range_recentre = P->range_recentre[l][r];
                ep.me.reserved = (P->me.me.reserved & 1) | (r << 8); /* me_raster's step scales with refi + 1 */
                int16_t m[2] = {0, 0};
                const uint32_t mecost = xo_me_epzs_mot(org[0], s_org_l, NULL, refp[r * 2 + l].y, s_l, x, y, job->mvp[l][mvp_idx[l]], m, lw, lh, bd, xo_mc_l_coeff, &ep,
                                                       &mot_bits[l]);
                mv_scale[l][r][0] = m[0], mv_scale[l][r][1] = m[1];
                if(mecost < best_mecost) best_mecost = mecost, refi_temp = r;
            }
            mv[l][l][0] = mv_scale[l][refi_temp][0], mv[l][l][1] = mv_scale[l][refi_temp][1];
            refi[l][0] = (int8_t)(l == 0 ? refi_temp : -1), refi[l][1] = (int8_t)(l == 1 ? refi_temp : -1);
            mvp_idx[l] = (uint8_t)xo_check_best_mvp(&states[job->sbac], p->slice_type, refi[l], l, job->mvp[l], mv[l][l], mvp_idx[l], p->lambda[0], mvd[l][l]);
            mvpi[l][0] = mvp_idx[0], mvpi[l][1] = mvp_idx[1]; /* (the local pair is what pinter_residue_rdo is given) */
            memset(&rj, 0, sizeof(rj));
            rj.x = x, rj.y = y, memcpy(rj.mv, mv[l], sizeof(rj.mv)), memcpy(rj.mvd, mvd[l], sizeof(rj.mvd)), rj.refi[0] = refi[l][0], rj.refi[1] = refi[l][1];
            rj.mvp_idx[0] = mvp_idx[0], rj.mvp_idx[1] = mvp_idx[1], rj.ctx_skip = job->ctx_skip, rj.ctx_pred_mode = job->ctx_pred_mode, rj.sbac = job->sbac;
            xo_residue_rdo(org, s_org_l, s_org_c, refp, s_l, s_c, states, p, &rj, &rr, coef[l][0], coef[l][1], coef[l][2], &st);
            memcpy(nnz[l], rr.nnz, sizeof(rr.nnz));
            cost = cost_inter[l] = rr.cost;
            if(cost < cost_best) cu_mode = 1, best_idx = l, cost_best = cost, *next_best = st;
        }
        if(isb) { /* analyze_bi (:1567-1714) */
            int      lidx_ref = cost_inter[L0] <= cost_inter[L1] ? 0 : 1, lidx_cnd = 1 - lidx_ref, t;
            int8_t   rf[2];
            uint32_t best_mecost = 0xFFFFFFFFu;
            int      refi_best = 0, changed;
            const int nb = p->num_refp[1]; /* pi->num_refp as the list-1 search left it */
            xo_pel  *pr[3];
            int16_t *org_bi = malloc(sizeof(int16_t) * (size_t)n0);
            for(int c = 0; c < 3; c++) pr[c] = malloc(sizeof(xo_pel) * (size_t)n0);
            mvpi[BI][0] = mvpi[L0][0], mvpi[BI][1] = mvpi[L1][1], refi[BI][0] = refi[L0][0], refi[BI][1] = refi[L1][1];
            memcpy(mv[BI][0], mv[L0][0], 4), memcpy(mv[BI][1], mv[L1][1], 4);
            rf[lidx_ref] = refi[BI][lidx_ref], rf[lidx_cnd] = -1;
            for(int i = 0; i < 4; i++) { /* BI_ITER */
                xo_cu_mc_job mj;
                memset(&mj, 0, sizeof(mj));
                mj.x = x, mj.y = y, memcpy(mj.mv, mv[BI], sizeof(mj.mv)), mj.refi[0] = rf[0], mj.refi[1] = rf[1];
                xo_mc_cu(refp, s_l, s_c, p->pic_w, p->pic_h, &mj, w, h, bd, bd, idc, pr[0], pr[1], pr[2]);
                for(int yy = 0; yy < h; yy++) /* get_org_bi (:143-156) */
                    for(int xx = 0; xx < w; xx++) org_bi[yy * w + xx] = (int16_t)((org[0][(y + yy) * s_org_l + x + xx] << 1) - pr[0][yy * w + xx]);
                t = rf[lidx_ref], rf[lidx_ref] = rf[lidx_cnd], rf[lidx_cnd] = (int8_t)t;
                t = lidx_ref, lidx_ref = lidx_cnd, lidx_cnd = t;
                const int idx = mvpi[BI][lidx_ref];
                changed = 0;
                for(int r = 0; r < nb; r++) {
                    rf[lidx_ref] = (int8_t)r;
                    ep.me.bi = 1, ep.me.extra_bits = mot_bits[lidx_cnd], ep.me.refi_bits = P->refi_bits[1][r], ep.me.range_recentre = P->range_recentre[lidx_ref][r];
                    int dummy = 0;
                    const uint32_t mecost = xo_me_epzs_mot(org[0], s_org_l, org_bi, refp[r * 2 + lidx_ref].y, s_l, x, y, job->mvp[lidx_ref][idx], mv_scale[lidx_ref][r], lw,
                                                           lh, bd, xo_mc_l_coeff, &ep, &dummy);
                    if(mecost < best_mecost) {
                        refi_best = r, best_mecost = mecost, changed = 1;
                        refi[BI][lidx_ref] = (int8_t)refi_best; /* (the other list keeps pi->refi[pidx][lidx_cnd]) */
                        mv[BI][lidx_ref][0] = mv_scale[lidx_ref][r][0], mv[BI][lidx_ref][1] = mv_scale[lidx_ref][r][1];
                    }
                }
                rf[lidx_ref] = (int8_t)refi_best, rf[lidx_cnd] = -1;
                if(!changed) break;
            }
            for(int l = 0; l < 2; l++)
                for(int d = 0; d < 2; d++) mvd[BI][l][d] = (int16_t)(mv[BI][l][d] - job->mvp[l][mvpi[BI][l]][d]);
            memset(&rj, 0, sizeof(rj));
            rj.x = x, rj.y = y, memcpy(rj.mv, mv[BI], sizeof(rj.mv)), memcpy(rj.mvd, mvd[BI], sizeof(rj.mvd)), rj.refi[0] = refi[BI][0], rj.refi[1] = refi[BI][1];
            rj.mvp_idx[0] = mvpi[BI][0], rj.mvp_idx[1] = mvpi[BI][1], rj.ctx_skip = job->ctx_skip, rj.ctx_pred_mode = job->ctx_pred_mode, rj.sbac = job->sbac;
            xo_residue_rdo(org, s_org_l, s_org_c, refp, s_l, s_c, states, p, &rj, &rr, coef[BI][0], coef[BI][1], coef[BI][2], &st);
            memcpy(nnz[BI], rr.nnz, sizeof(rr.nnz));
            cost = cost_inter[BI] = rr.cost;
            if(cost < cost_best) cu_mode = 1, best_idx = BI, cost_best = cost, *next_best = st;
            free(org_bi);
            for(int c = 0; c < 3; c++) free(pr[c]);
        }
    }

    /* the winner: coefficients, reconstruction (:2004-2032), motion data (:2036-2046) */
    int16_t *co[3] = {coef_y, coef_u, coef_v};
    xo_pel  *rec[3] = {rec_y, rec_u, rec_v}, *pr[3];
    int16_t *tmp = malloc(sizeof(int16_t) * (size_t)n0);
    for(int c = 0; c < 3; c++) pr[c] = malloc(sizeof(xo_pel) * (size_t)n0);
    if(best_idx == SKIP) for(int c = 0; c < ncomp; c++) memcpy(pr[c], pred_skip[c], sizeof(xo_pel) * (size_t)(c ? n1 : n0));
    else {
        xo_cu_mc_job mj;
        memset(&mj, 0, sizeof(mj));
        mj.x = x, mj.y = y, memcpy(mj.mv, mv[best_idx], sizeof(mj.mv)), mj.refi[0] = refi[best_idx][0], mj.refi[1] = refi[best_idx][1];
        xo_mc_cu(refp, s_l, s_c, p->pic_w, p->pic_h, &mj, w, h, bd, bd, idc, pr[0], pr[1], pr[2]);
    }
    if(pred_y_best) memcpy(pred_y_best, pr[0], sizeof(xo_pel) * (size_t)n0);
    for(int c = 0; c < ncomp; c++) {
        const int n = c ? n1 : n0, l2w = c ? lw - ws : lw, l2h = c ? lh - hs : lh;
        memcpy(co[c], coef[best_idx][c], sizeof(int16_t) * (size_t)n);
        memcpy(tmp, co[c], sizeof(int16_t) * (size_t)n);
        if(nnz[best_idx][c]) xo_dquant(tmp, l2w, l2h, xo_dq_scale[p->qp[c] % 6] << (p->qp[c] / 6), bd), xo_itrans(tmp, l2w, l2h, bd);
        xo_recon(tmp, pr[c], nnz[best_idx][c], 1 << l2w, 1 << l2h, 1 << l2w, rec[c], bd);
    }
    memset(res, 0, sizeof(*res));
    res->cost = cost_inter[best_idx], memcpy(res->cost_inter, cost_inter, sizeof(cost_inter)), res->cu_mode = cu_mode, res->best_idx = best_idx;
    for(int l = 0; l < 2; l++) {
        const int used = refi[best_idx][l] >= 0 && (isb || l == 0);
        res->refi[l] = (isb || l == 0) ? refi[best_idx][l] : -1;
        if(used) memcpy(res->mv[l], mv[best_idx][l], 4), memcpy(res->mvd[l], mvd[best_idx][l], 4), res->mvp_idx[l] = mvpi[best_idx][l];
    }
    if(best_idx == DIR) res->mvp_idx[0] = res->mvp_idx[1] = 0;
    memcpy(res->nnz, nnz[best_idx], sizeof(res->nnz));
    free(tmp);
    for(int c = 0; c < 3; c++) free(pr[c]), free(pred_skip[c]);
    for(int m = 0; m < NP; m++)
        for(int c = 0; c < 3; c++) free(coef[m][c]);
}

/* xeve_get_avail_inter (left, up, up-right) + xeve_get_motion + the collocated vector of the temporal direct mode */
void xo_inter_candidates(const uint32_t *map_scu, const uint8_t *map_tidx, const int16_t (*map_mv)[2][2], const int16_t (*col0)[2][2],
                         const int16_t (*col1)[2][2], int w_scu, int h_scu, int log2_cuw, int log2_cuh, int slice_type, xo_inter_job *job)
{
#define M_IF(m) (((m) >> 15) & 1)
#define M_IBC(m) (((m) >> 26) & 1)
#define M_COD(m) (((m) >> 31) & 1)
    const int x_scu = job->x >> 2, y_scu = job->y >> 2, scup = y_scu * w_scu + x_scu, scuw = 1 << (log2_cuw - 2), scuh = 1 << (log2_cuh - 2);
    (void)h_scu;
    const int t = map_tidx[scup];
    int le = 0, up = 0, ur = 0;
    if(x_scu > 0) {
        const uint32_t m = map_scu[scup - 1];
        le = !M_IF(m) && M_COD(m) && map_tidx[scup - 1] == t && !M_IBC(m);
    }
    if(y_scu > 0) {
        const uint32_t m = map_scu[scup - w_scu];
        up = !M_IF(m) && map_tidx[scup - w_scu] == t && !M_IBC(m); /* (no COD test for this one, :681-684) */
        if(x_scu + scuw < w_scu) {
            const uint32_t r = map_scu[scup - w_scu + scuw];
            ur = (((r >> 15) & 0x10001) == 0x10000) && M_COD(r) && map_tidx[scup - w_scu + scuw] == t; /* MCU_IS_COD_NIF */
        }
    }
    memset(job->mvp, 0, sizeof(job->mvp)), job->mv_col[0] = job->mv_col[1] = 0;
    for(int l = 0; l <= (slice_type == 0 ? 1 : 0); l++) {
        const int16_t(*col)[2][2] = l ? col1 : col0;
        const int at[3] = {scup - 1, scup - w_scu, scup - w_scu + scuw}, ok[3] = {le, up, ur};
        for(int k = 0; k < 3; k++) job->mvp[l][k][0] = ok[k] ? map_mv[at[k]][l][0] : 1, job->mvp[l][k][1] = ok[k] ? map_mv[at[k]][l][1] : 1;
        job->mvp[l][3][0] = col[scup][0][0], job->mvp[l][3][1] = col[scup][0][1];
    }
    if(slice_type == 0) {
        const int corner = scup + (scuw - 1) + (scuh - 1) * w_scu;
        job->mv_col[0] = col1[corner][0][0], job->mv_col[1] = col1[corner][0][1];
    }
#undef M_IF
#undef M_IBC
#undef M_COD
}


/* ===================================================================================================================
 * Intra analysis of one CU: pintra_analyze_cu (src_base/xeve_pintra.c:544-698) = ctx->fn_pintra_analyze_cu, Baseline profile,
 * rdo_dbk_switch 0, no delta QP.  Neighbour samples (xeve_get_nbr, xeve_ipred.c:32-105), the five Baseline predictors
 * (xeve_ipred.c:107-202), the most-probable-mode list (xeve_get_mpm, :229-252), the SATD + mode-bits candidate list
 * (make_ipred_list, xeve_pintra.c:308-374), the luma RDO of the list and the chroma RDO of its winner (pintra_residue_rdo, :69-272).
 * =================================================================================================================== */
#define SCU_COD_(m) (((m) >> 31) & 1)
#define SCU_IF_(m)  (((m) >> 15) & 1)
const uint8_t xo_tbl_mpm[6][6][5] = { /* xeve_tbl_mpm (xeve_tbl.c:40-48): [left mode + 1 | 0][up mode + 1 | 0] -> rank of every mode */
    {{0, 2, 3, 1, 4}, {0, 2, 1, 3, 4}, {0, 2, 1, 3, 4}, {1, 2, 0, 3, 4}, {0, 2, 1, 3, 4}, {0, 1, 2, 3, 4}},
    {{1, 0, 2, 3, 4}, {0, 1, 2, 3, 4}, {0, 1, 2, 3, 4}, {1, 2, 0, 3, 4}, {0, 1, 3, 2, 4}, {0, 2, 1, 4, 3}},
    {{1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {2, 0, 1, 3, 4}, {1, 0, 3, 2, 4}, {0, 1, 2, 4, 3}},
    {{1, 0, 2, 3, 4}, {0, 2, 1, 3, 4}, {1, 0, 2, 3, 4}, {1, 2, 0, 3, 4}, {0, 1, 2, 3, 4}, {0, 2, 1, 4, 3}},
    {{0, 1, 2, 3, 4}, {0, 3, 2, 1, 4}, {1, 0, 2, 3, 4}, {1, 2, 0, 3, 4}, {1, 2, 3, 0, 4}, {0, 2, 1, 4, 3}},
    {{0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}, {0, 1, 2, 4, 3}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}}};

/* xeve_get_nbr (xeve_ipred.c:32-105) for one component.  x, y, cuw, cuh in samples OF THAT COMPONENT; src = the component's plane of the picture
 * being reconstructed (PIC_MODE) at (x, y).  up[-1 .. cuw + cuh - 1], left[-1 .. cuw + cuh - 1]: the caller passes pointers to element 0.
 * Unavailable 4x4 units (not yet coded, outside the picture, another tile, or -- with constrained intra prediction -- not intra) read as mid-grey.
 * Only the up-left bit of avail_cu is used (xeve_get_avail_intra, xeve_util.c:753-755). */
void xo_get_nbr(int x, int y, int cuw, int cuh, const xo_pel *src, int s_src, const uint32_t *map_scu, const uint8_t *map_tidx, int w_scu, int h_scu, int ch,
                int constrained_intra_pred, int bit_depth, int chroma_format_idc, xo_pel *left, xo_pel *up)
{
    const int ws = chroma_format_idc <= 2, hs = chroma_format_idc <= 1; /* XEVE_GET_CHROMA_W/H_SHIFT for idc 1, 2, 3 (0: no chroma call) */
    int scuw = ch == 0 ? cuw >> 2 : cuw >> (2 - ws), scuh = ch == 0 ? cuh >> 2 : cuh >> (2 - hs);
    int unit = ch == 0 ? 4 : 2;
    const int x_scu = (ch == 0 ? x : x << ws) >> 2, y_scu = (ch == 0 ? y : y << hs) >> 2, scup = y_scu * w_scu + x_scu;
    const xo_pel grey = (xo_pel)(1 << (bit_depth - 1));
    if(ch != 0 && chroma_format_idc == 2) scuh *= 2;
    if(ch != 0 && chroma_format_idc == 3) unit *= 2;
#define USABLE(u) (SCU_COD_(map_scu[u]) && (!constrained_intra_pred || SCU_IF_(map_scu[u])) && map_tidx[scup] == map_tidx[u])
    /* the up-left sample: avail_cu & AVAIL_UP_LE = (x_scu > 0 && y_scu > 0 && coded && same tile), then the constrained-intra test */
    if(x_scu > 0 && y_scu > 0 && SCU_COD_(map_scu[scup - w_scu - 1]) && map_tidx[scup] == map_tidx[scup - w_scu - 1] &&
       (!constrained_intra_pred || SCU_IF_(map_scu[scup - w_scu - 1])))
        up[-1] = src[-s_src - 1];
    else up[-1] = grey;
    for(int i = 0; i < scuw + scuh; i++) {
        const int ok = y_scu > 0 && x_scu + i < w_scu && USABLE(scup - w_scu + i);
        for(int k = 0; k < unit; k++) up[i * unit + k] = ok ? src[-s_src + i * unit + k] : grey;
    }
    for(int i = 0; i < scuh + scuw; i++) {
        const int ok = x_scu > 0 && y_scu + i < h_scu && USABLE(scup - 1 + i * w_scu);
        for(int k = 0; k < unit; k++) left[i * unit + k] = ok ? src[(i * unit + k) * s_src - 1] : grey;
    }
    left[-1] = up[-1];
#undef USABLE
}

/* xeve_ipred / xeve_ipred_uv (xeve_ipred.c:107-227): DC 0, horizontal 1, vertical 2, up-left diagonal 3, up-right average 4; dst dense w x h */
void xo_ipred(const xo_pel *left, const xo_pel *up, xo_pel *dst, int ipm, int w, int h)
{
    if(ipm == 0) {
        int dc = 0, sh = 0;
        for(int i = 0; i < h; i++) dc += left[i];
        for(int j = 0; j < w; j++) dc += up[j];
        while((1 << sh) < w) sh++;     /* xeve_tbl_log2[w] */
        dc = (dc + w) >> (sh + 1);     /* (the divisor is 2w whatever h is: the reference's expression, exact for square blocks) */
        for(int i = 0; i < w * h; i++) dst[i] = (xo_pel)dc;
        return;
    }
    for(int i = 0; i < h; i++)
        for(int j = 0; j < w; j++) {
            int v;
            if(ipm == 1) v = left[i];
            else if(ipm == 2) v = up[j];
            else if(ipm == 3) v = i > j ? left[i - j - 1] : (i == j ? up[-1] : up[j - i - 1]);
            else v = (up[i + j + 1] + left[i + j + 1]) >> 1;
            dst[i * w + j] = (xo_pel)v;
        }
}

/* xeve_get_mpm (xeve_ipred.c:229-252): the row of xeve_tbl_mpm the left and upper neighbours' luma modes select */
const uint8_t *xo_get_mpm(int x_scu, int y_scu, const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, int w_scu)
{
    const int scup = y_scu * w_scu + x_scu;
    int l = 0, u = 0;
    if(x_scu > 0 && SCU_IF_(map_scu[scup - 1]) && SCU_COD_(map_scu[scup - 1]) && map_tidx[scup] == map_tidx[scup - 1]) l = map_ipm[scup - 1] + 1;
    if(y_scu > 0 && SCU_IF_(map_scu[scup - w_scu]) && SCU_COD_(map_scu[scup - w_scu]) && map_tidx[scup] == map_tidx[scup - w_scu]) u = map_ipm[scup - w_scu] + 1;
    return xo_tbl_mpm[l][u];
}

void xo_pintra_analyze_cu(const xo_pel *const org[3], int s_org_l, int s_org_c, const xo_pel *const mod[3], int s_mod_l, int s_mod_c, const uint32_t *map_scu,
                          const int8_t *map_ipm, const uint8_t *map_tidx, const xo_sbac *states, const xo_intra_params *p, const xo_intra_job *job,
                          xo_intra_result *res, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, xo_pel *rec_y, xo_pel *rec_u, xo_pel *rec_v, xo_sbac *best)
{
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth;
    const int lw[3] = {p->log2_cuw, p->log2_cuw - ws, p->log2_cuw - ws}, lh[3] = {p->log2_cuh, p->log2_cuh - hs, p->log2_cuh - hs};
    const int cuw = 1 << lw[0], cuh = 1 << lh[0], n0 = cuw * cuh, n1 = idc ? 1 << (lw[1] + lh[1]) : 0;
    const int x = job->x, y = job->y;
    const xo_sbac *entry = &states[job->sbac];
    int16_t *coef[3] = {coef_y, coef_u, coef_v};
    xo_pel  *rec[3]  = {rec_y, rec_u, rec_v};
    xo_pel   nb[3][2][2 * 64 + 2 * 64 + 8];
#define LEFT(c) (nb[c][0] + 2)
#define UP(c)   (nb[c][1] + 2)
    /* neighbours (pintra_get_nbr, :390-461) and the mode ranks (pintra_get_mpm) */
    xo_get_nbr(x, y, cuw, cuh, mod[0] + (size_t)y * s_mod_l + x, s_mod_l, map_scu, map_tidx, p->w_scu, p->h_scu, 0, p->constrained_intra_pred, bd, idc, LEFT(0), UP(0));
    for(int c = 1; c < 3 && idc; c++)
        xo_get_nbr(x >> ws, y >> hs, cuw >> ws, cuh >> hs, mod[c] + (size_t)(y >> hs) * s_mod_c + (x >> ws), s_mod_c, map_scu, map_tidx, p->w_scu, p->h_scu, c,
                   p->constrained_intra_pred, bd, idc, LEFT(c), UP(c));
    const uint8_t *mpm = xo_get_mpm(x >> 2, y >> 2, map_scu, map_ipm, map_tidx, p->w_scu);

    xo_cu_bits_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.log2_cuw = lw[0], bp.log2_cuh = lh[0], bp.slice_type = p->slice_type, bp.cm_init = 0, bp.chroma_format_idc = idc;
    xo_cu_bits_job bj;
    memset(&bj, 0, sizeof(bj));
    bj.ctx_skip = job->ctx_skip, bj.ctx_pred_mode = job->ctx_pred_mode, bj.coef_off[1] = n0, bj.coef_off[2] = n0 + n1;
    int16_t *all = malloc(sizeof(int16_t) * (n0 + 2 * n1 + 1));
    xo_pel  *pred_cache = malloc(sizeof(xo_pel) * 5 * n0), *pred_c = malloc(sizeof(xo_pel) * (n1 + 1)), *rec_t = malloc(sizeof(xo_pel) * n0);
    int16_t *tmp = malloc(sizeof(int16_t) * n0), *coef_t = malloc(sizeof(int16_t) * n0);
    const xo_pel *o[3] = {org[0] + (size_t)y * s_org_l + x, idc ? org[1] + (size_t)(y >> hs) * s_org_c + (x >> ws) : NULL,
                          idc ? org[2] + (size_t)(y >> hs) * s_org_c + (x >> ws) : NULL};

    /* make_ipred_list (:308-374): SATD + sqrt(lambda) * bits of the mode, an insertion-sorted list of the IPD_RDO_CNT (5; 4 for 1:4 shapes) cheapest, cut
     * from the tail while the SATD alone exceeds 1.2 x the SATD of the best inter prediction */
    const int rdo_cnt = iabs(lw[0] - lh[0]) >= 2 ? 4 : 5;
    int       list[5];
    double    cand_cost[5];
    uint32_t  cand_satd[5];
    for(int i = 0; i < rdo_cnt; i++) list[i] = 0, cand_cost[i] = 1.7e+308, cand_satd[i] = 0xFFFFFFFFu;
    for(int m = 0; m < 5; m++) {
        xo_sbac run;
        xo_ipred(LEFT(0), UP(0), pred_cache + m * n0, m, cuw, cuh);
        const uint32_t satd = (uint32_t)xo_satd(cuw, cuh, o[0], pred_cache + m * n0, s_org_l, cuw, bd);
        bj.mode = XO_BITS_INTRA_DIR, bj.sbac = 0, bj.mvp_idx[0] = mpm[m];
        const int bits = (int)xo_cu_bits(entry, &run, &bp, &bj, all);
        const double cost = (double)satd + (double)bits * p->sqrt_lambda0;
        int shift = 0;
        while(shift < rdo_cnt && cost < cand_cost[rdo_cnt - 1 - shift]) shift++;
        if(shift) {
            for(int j = 1; j < shift; j++)
                list[rdo_cnt - j] = list[rdo_cnt - 1 - j], cand_cost[rdo_cnt - j] = cand_cost[rdo_cnt - 1 - j], cand_satd[rdo_cnt - j] = cand_satd[rdo_cnt - 1 - j];
            list[rdo_cnt - shift] = m, cand_cost[rdo_cnt - shift] = cost, cand_satd[rdo_cnt - shift] = satd;
        }
    }
    int pred_cnt = rdo_cnt;
    for(int i = rdo_cnt - 1; i >= 1; i--) {
        if((double)cand_satd[i] > (double)job->inter_satd * (1.2)) pred_cnt--;
        else break;
    }

    /* luma RDO of the list (:604-637; pintra_residue_rdo mode 0, :102-148): every candidate starts from the entry coder state */
    xo_rdoq_est_full full;
    xo_rdoq_est      e;
    xo_rdoq_bit_est(entry, &full); /* core->rdoq_est_* of mode_coding_unit (xeve_mode.c:792) */
    double  cost_best = 1.7e+308;
    int     best_ipd = -1, nnz_best[3] = {0, 0, 0};
    int32_t best_dist_y = 0, best_dist_c = 0;
    xo_sbac run;
    for(int j = 0; j < pred_cnt; j++) {
        const int     m = list[j];
        const xo_pel *pr = pred_cache + m * n0;
        int           nnz = 0;
        xo_diff(cuw, cuh, o[0], pr, s_org_l, cuw, cuw, coef_t);
        xo_trans(coef_t, lw[0], lh[0], bd);
        if(xo_rdoq_zero_test(coef_t, lw[0], lh[0], p->qp[0], xo_quant_scale[p->tool_iqt][p->qp[0] % 6], p->slice_type == 2, bd)) {
            xo_rdoq_est_select(&full, 0, 1, &e);
            nnz = xo_rdoq(coef_t, lw[0], lh[0], p->qp[0], p->lambda[0], 1, bd, p->tool_iqt, &e);
        }
        else memset(coef_t, 0, sizeof(int16_t) * n0);
        memcpy(all, coef_t, sizeof(int16_t) * n0);
        bj.mode = XO_BITS_INTRA_LUMA, bj.sbac = 0, bj.mvp_idx[0] = mpm[m], bj.nnz[0] = nnz, bj.nnz[1] = bj.nnz[2] = 0;
        const int bits = (int)xo_cu_bits(entry, &run, &bp, &bj, all); /* `run` = core->s_temp_run: the chroma count below continues from the LAST candidate's */
        memcpy(tmp, coef_t, sizeof(int16_t) * n0);
        if(nnz) {
            xo_dquant(tmp, lw[0], lh[0], xo_dq_scale[p->qp[0] % 6] << (p->qp[0] / 6), bd);
            xo_itrans(tmp, lw[0], lh[0], bd);
        }
        xo_recon(tmp, pr, nnz, cuw, cuh, cuw, rec_t, bd);
        double cost = 0;
        cost += (double)xo_ssd(cuw, cuh, rec_t, o[0], cuw, s_org_l, bd);
        if(xo_rdo_dbk_on()) { /* (xeve_pintra.c:131-149) */
            int64_t d[3];
            const xo_pel *const src[3] = {rec_t, NULL, NULL};
            xo_delta_dist(&g_dbk, org, s_org_l, s_org_c, src, x, y, cuw, cuh, 1, nnz != 0, NULL, NULL, d);
            cost += d[0];
        }
        const int32_t dist = (int32_t)cost;
        cost += (double)bits * p->lambda[0];
        if(cost < cost_best) {
            cost_best = cost, best_dist_y = dist, best_ipd = m, nnz_best[0] = nnz;
            memcpy(coef[0], coef_t, sizeof(int16_t) * n0), memcpy(rec[0], rec_t, sizeof(xo_pel) * n0);
        }
    }

    /* chroma with the luma winner's mode (:639-658; pintra_residue_rdo mode 1, :150-269): bits counted after xeve_sbac_bit_reset on the coder state the
     * last luma candidate left (no SBAC_LOAD there) */
    if(idc) {
        double cost = 0;
        for(int c = 1; c < 3; c++) {
            const int w = 1 << lw[c], h = 1 << lh[c];
            int nnz = 0;
            xo_ipred(LEFT(c), UP(c), pred_c, best_ipd, w, h);
            xo_diff(w, h, o[c], pred_c, s_org_c, w, w, coef[c]);
            xo_trans(coef[c], lw[c], lh[c], bd);
            if(xo_rdoq_zero_test(coef[c], lw[c], lh[c], p->qp[c], xo_quant_scale[p->tool_iqt][p->qp[c] % 6], p->slice_type == 2, bd)) {
                xo_rdoq_est_select(&full, c, 1, &e);
                nnz = xo_rdoq(coef[c], lw[c], lh[c], p->qp[c], p->lambda[c], 0, bd, p->tool_iqt, &e);
            }
            else memset(coef[c], 0, sizeof(int16_t) * n1);
            nnz_best[c] = nnz;
            memcpy(tmp, coef[c], sizeof(int16_t) * n1);
            if(nnz) {
                xo_dquant(tmp, lw[c], lh[c], xo_dq_scale[p->qp[c] % 6] << (p->qp[c] / 6), bd);
                xo_itrans(tmp, lw[c], lh[c], bd);
            }
            xo_recon(tmp, pred_c, nnz, w, h, w, rec[c], bd);
        }
        memcpy(all + n0, coef[1], sizeof(int16_t) * n1), memcpy(all + n0 + n1, coef[2], sizeof(int16_t) * n1);
        bj.mode = XO_BITS_ECO_COEF, bj.dir_flag = XO_ECO_INTRA | XO_ECO_RUN_U | XO_ECO_RUN_V, bj.sbac = 0, bj.nnz[0] = 0, bj.nnz[1] = nnz_best[1], bj.nnz[2] = nnz_best[2];
        xo_sbac after_luma = run;
        (void)xo_cu_bits(&after_luma, &run, &bp, &bj, all); /* (its bits only enter the local cost the reference discards: cost_t is not used after, :643-646) */
        cost += p->dist_chroma_weight[0] * (double)xo_ssd(1 << lw[1], 1 << lh[1], rec[1], o[1], 1 << lw[1], s_org_c, bd);
        cost += p->dist_chroma_weight[1] * (double)xo_ssd(1 << lw[2], 1 << lh[2], rec[2], o[2], 1 << lw[2], s_org_c, bd);
        if(xo_rdo_dbk_on()) { /* (xeve_pintra.c:244-263; the luma plane it copies does not reach the chroma figures) */
            int64_t d[3];
            const xo_pel *const src[3] = {NULL, rec[1], rec[2]};
            xo_delta_dist(&g_dbk, org, s_org_l, s_org_c, src, x, y, cuw, cuh, 1, 0, NULL, NULL, d);
            cost += ((double)d[1] * p->dist_chroma_weight[0]) + ((double)d[2] * p->dist_chroma_weight[1]);
        }
        best_dist_c = (int32_t)cost;
    }

    /* the CU's cost (:679-695): the whole syntax from the entry state */
    memcpy(all, coef[0], sizeof(int16_t) * n0);
    bj.mode = XO_BITS_CU_INTRA, bj.dir_flag = 0, bj.sbac = 0, bj.mvp_idx[0] = mpm[best_ipd], bj.nnz[0] = nnz_best[0], bj.nnz[1] = nnz_best[1], bj.nnz[2] = nnz_best[2];
    const int bits = (int)xo_cu_bits(entry, best, &bp, &bj, all);
    double cost = (double)bits * p->lambda[0];
    cost += best_dist_y;
    if(idc) cost += best_dist_c;
    memset(res, 0, sizeof(*res));
    res->cost = cost, res->dist_cu = best_dist_y + (idc ? best_dist_c : 0), res->ipm[0] = (int8_t)best_ipd, res->ipm[1] = (int8_t)(idc ? best_ipd : 0);
    res->nnz[0] = nnz_best[0], res->nnz[1] = nnz_best[1], res->nnz[2] = nnz_best[2], res->pred_cnt = pred_cnt;
    free(all), free(pred_cache), free(pred_c), free(rec_t), free(tmp), free(coef_t);
#undef LEFT
#undef UP
}

/* ===================================================================================================================
 * The mode decision of one CTU of an I picture: mode_analyze_lcu -> mode_coding_tree (src_base/xeve_mode.c:2007-2375, 2518-2610) with
 * mode_coding_unit -> mode_check_intra -> pintra_analyze_cu at every node; Baseline quad-tree, no delta QP, rdo_dbk_switch 0.
 * Every node: [the CU as a whole: split_cu_flag = 0 priced from the node's entry coder state, then the intra analysis] -- unless the CU is larger
 * than max_cu_intra or crosses the picture edge -- then, unless the early-termination rule for I pictures fires, [split_cu_flag = 1 priced from the
 * same entry state, the four quadrants in z order, each starting from the coder state its predecessor left]; the cheaper alternative's data, reconstruction
 * and coder state are kept (a split must be cheaper by more than 0.0001).  The 4x4-unit maps and the picture being reconstructed are updated as the walk
 * goes, because the next CU's neighbours are read from them: clear_map_scu (:1129), update_map_scu (:1036), mode_cpy_rec_to_ref (:797).
 * =================================================================================================================== */
typedef struct tree_ctx {
    const xo_pel *const *org;
    xo_pel *const       *mod;
    int                  s_org_l, s_org_c, s_mod_l, s_mod_c;
    uint32_t            *map_scu, *map_cu_mode;
    int8_t              *map_ipm;
    const uint8_t       *map_tidx;
    const xo_tree_params *P;
    const xo_tree_inter *I;                          /* NULL in I slices */
    int                  cu_mode;                    /* core->cu_mode of the last mode_coding_unit */
    xo_sbac              curr_best[5], next_best[5]; /* core->s_curr_best / s_next_best [log2 - 2][log2 - 2] */
    xo_ctu_data         *best[5], *temp[5];          /* core->cu_data_best / cu_data_temp, node-local indexing (pitch = the node's size) */
    int32_t              dist_cu_best;               /* core->dist_cu_best */
} tree_ctx;

static void cud_init(xo_ctu_data *d, int log2)
{   /* init_cu_data (:374-428): what the I-slice walk reads back -- split modes and luma / chroma modes cleared */
    const int n = 1 << (2 * (log2 - 2));
    for(int k = 0; k < XO_CU_DEPTHS; k++) memset(d->split_mode[k], 0, n);
    memset(d->ipm[0], 0, n), memset(d->ipm[1], 0, n);
}
/* copy_cu_data (:430-620): the sub-block (x, y; log2 size) of dst (pitch 1 << log2_cus) <- all of src, split modes from depth cud on */
static void cud_copy(xo_ctu_data *dst, const xo_ctu_data *src, int x, int y, int log2, int log2_cus, int cud, int idc)
{
    const int ws = idc <= 2, hs = idc <= 1, n = 1 << (log2 - 2), cus = 1 << (log2_cus - 2), cw = 1 << log2, cs = 1 << log2_cus;
    for(int j = 0; j < n; j++) {
        const int di = ((y >> 2) + j) * cus + (x >> 2), si = j * n;
        for(int k = cud; k < XO_CU_DEPTHS; k++) memcpy(dst->split_mode[k] + di, src->split_mode[k] + si, n);
        memcpy(dst->pred_mode + di, src->pred_mode + si, n), memcpy(dst->ipm[0] + di, src->ipm[0] + si, n), memcpy(dst->ipm[1] + di, src->ipm[1] + si, n);
        memcpy(dst->depth + di, src->depth + si, n);
        memcpy(dst->map_scu + di, src->map_scu + si, 4 * n), memcpy(dst->map_cu_mode + di, src->map_cu_mode + si, 4 * n);
        for(int c = 0; c < 3; c++) memcpy(dst->nnz[c] + di, src->nnz[c] + si, 4 * n);
        memcpy(dst->mv + di, src->mv + si, sizeof(src->mv[0]) * n), memcpy(dst->mvd + di, src->mvd + si, sizeof(src->mvd[0]) * n);
        memcpy(dst->refi + di, src->refi + si, 2 * n), memcpy(dst->mvp_idx + di, src->mvp_idx + si, 2 * n);
    }
    for(int j = 0; j < cw; j++) {
        memcpy(dst->coef[0] + (y + j) * cs + x, src->coef[0] + j * cw, 2 * cw);
        memcpy(dst->reco[0] + (y + j) * cs + x, src->reco[0] + j * cw, 2 * cw);
    }
    if(idc)
        for(int c = 1; c < 3; c++)
            for(int j = 0; j < cw >> hs; j++) {
                memcpy(dst->coef[c] + ((y >> hs) + j) * (cs >> ws) + (x >> ws), src->coef[c] + j * (cw >> ws), 2 * (cw >> ws));
                memcpy(dst->reco[c] + ((y >> hs) + j) * (cs >> ws) + (x >> ws), src->reco[c] + j * (cw >> ws), 2 * (cw >> ws));
            }
}
static void tree_clear_map(tree_ctx *T, int x, int y, int cu)
{   /* clear_map_scu (:1129-1155) */
    const xo_tree_params *P = T->P;
    const int w = ((x + cu > P->pic_w ? P->pic_w - x : cu) >> 2), h = ((y + cu > P->pic_h ? P->pic_h - y : cu) >> 2);
    for(int i = 0; i < h; i++) {
        memset(T->map_scu + ((y >> 2) + i) * P->ip.w_scu + (x >> 2), 0, 4 * w);
        memset(T->map_cu_mode + ((y >> 2) + i) * P->ip.w_scu + (x >> 2), 0, 4 * w);
    }
}
static void tree_update_map(tree_ctx *T, const xo_ctu_data *d, int x, int y, int cu)
{   /* update_map_scu (:1036-1127): the maps the intra analysis of later CUs reads */
    const xo_tree_params *P = T->P;
    const int w = ((x + cu > P->pic_w ? P->pic_w - x : cu) >> 2), h = ((y + cu > P->pic_h ? P->pic_h - y : cu) >> 2), n = cu >> 2;
    for(int i = 0; i < h; i++) {
        const int g = ((y >> 2) + i) * P->ip.w_scu + (x >> 2);
        memcpy(T->map_scu + g, d->map_scu + i * n, 4 * w), memcpy(T->map_cu_mode + g, d->map_cu_mode + i * n, 4 * w);
        memcpy(T->map_ipm + g, d->ipm[0] + i * n, w);
        if(T->I) memcpy(T->I->map_mv + g, d->mv + i * n, sizeof(d->mv[0]) * w), memcpy(T->I->map_refi + g, d->refi + i * n, 2 * w);
    }
}
static void tree_rec_to_pic(tree_ctx *T, const xo_ctu_data *d, int x, int y, int cu)
{   /* mode_cpy_rec_to_ref (:797-866) */
    const xo_tree_params *P = T->P;
    const int idc = P->ip.chroma_format_idc, ws = idc <= 2, hs = idc <= 1;
    const int w = x + cu > P->pic_w ? P->pic_w - x : cu, h = y + cu > P->pic_h ? P->pic_h - y : cu;
    for(int j = 0; j < h; j++) memcpy(T->mod[0] + (size_t)(y + j) * T->s_mod_l + x, d->reco[0] + j * cu, 2 * w);
    if(idc)
        for(int c = 1; c < 3; c++)
            for(int j = 0; j < h >> hs; j++) memcpy(T->mod[c] + (size_t)((y >> hs) + j) * T->s_mod_c + (x >> ws), d->reco[c] + j * (cu >> ws), 2 * (w >> ws));
}
static uint32_t tree_split_bits(const xo_sbac *from, xo_sbac *to, int split)
{   /* SBAC_LOAD + xeve_sbac_bit_reset + xeve_eco_split_mode (xeve_eco.c:1377-1429; Baseline: one bin on split_cu_flag) + xeve_get_bit_number */
    *to = *from;
    xo_sbac_bit_reset(to);
    xo_sbac_bin(to, XO_CTX_SPLIT_CU, split != 0);
    return xo_sbac_bits(to);
}

/* mode_coding_unit (:1310-1350): mode_cu_init, then mode_check_inter (:1169-1222) and mode_check_intra (:1226-1308); the cheaper mode's data into the node's
 * cu_data_temp (copy_to_cu_data, :868-1034), its coder state into s_next_best.  I slices: the intra analysis alone.  P / B slices (Baseline, tool_admvp 0):
 * the whole inter analysis first; the intra analysis only when the inter winner has a residual, its candidate list cut against the SATD of the inter
 * winner's luma prediction. */
static double tree_unit(tree_ctx *T, int x0, int y0, int log2, int cud, xo_ctu_data *t)
{
    const xo_tree_params *P = T->P;
    const int L = log2 - 2, cu = 1 << log2, idc = P->ip.chroma_format_idc, n = 1 << (2 * L), n0 = cu * cu;
    const uint32_t scu_base = ((uint32_t)P->slice_num & 0x7F) | ((uint32_t)P->slice_qp << 16) | (1u << 31); /* MCU_SET_IF_COD_SN_QP without the intra flag */
    const uint32_t cum      = ((uint32_t)log2 << 24) | ((uint32_t)log2 << 28);                                /* MCU_SET_LOGW / LOGH */
    double   cost_best = 1.7e+308;
    uint32_t inter_satd = 0xFFFFFFFFu;
    int      try_intra = 1;
    T->cu_mode = 0, T->dist_cu_best = 0x7FFFFFFF;
    if(T->I) { /* mode_check_inter */
        const xo_tree_inter *I = T->I;
        xo_inter_params ipar = I->ipar;
        ipar.rdo.log2_cuw = ipar.rdo.log2_cuh = log2;
        xo_inter_job ij;
        memset(&ij, 0, sizeof(ij));
        ij.x = x0, ij.y = y0, ij.sbac = 0; /* ctx_skip / ctx_pred_mode: 0 without sps_cm_init_flag (xeve_get_ctx_some_flags, xeve_util.c:1181-1288) */
        xo_inter_candidates(T->map_scu, T->map_tidx, (const int16_t(*)[2][2])I->map_mv, I->col0, I->col1, P->ip.w_scu, P->ip.h_scu, log2, log2, P->ip.slice_type, &ij);
        xo_inter_result r;
        xo_pel *pred_y = malloc(sizeof(xo_pel) * (size_t)n0);
        pinter_analyze_cu(T->org, T->s_org_l, T->s_org_c, I->refp, I->s_ref_l, I->s_ref_c, &T->curr_best[L], &ipar, &ij, &r, t->coef[0], t->coef[1], t->coef[2], t->reco[0],
                          t->reco[1], t->reco[2], &T->next_best[L], pred_y);
        cost_best = r.cost, T->cu_mode = r.cu_mode;
        for(int i = 0; i < n; i++) { /* copy_to_cu_data of an inter CU */
            t->pred_mode[i] = (uint8_t)r.cu_mode, t->depth[i] = (int8_t)cud;
            t->nnz[0][i] = r.nnz[0], t->nnz[1][i] = idc ? r.nnz[1] : 0, t->nnz[2][i] = idc ? r.nnz[2] : 0;
            t->map_scu[i] = scu_base | (r.cu_mode == 2 /* MODE_SKIP */ ? 1u << 23 : 0), t->map_cu_mode[i] = cum;
            memcpy(t->mv[i], r.mv, sizeof(r.mv)), memcpy(t->mvd[i], r.mvd, sizeof(r.mvd));
            t->refi[i][0] = r.refi[0], t->refi[i][1] = r.refi[1], t->mvp_idx[i][0] = r.mvp_idx[0], t->mvp_idx[i][1] = r.mvp_idx[1];
        }
        try_intra = r.nnz[0] != 0 || r.nnz[1] != 0 || r.nnz[2] != 0; /* (:1245-1247; cost_best cannot be MAX_COST here) */
        if(try_intra) inter_satd = (uint32_t)xo_satd(cu, cu, T->org[0] + (size_t)y0 * T->s_org_l + x0, pred_y, T->s_org_l, cu, P->ip.bit_depth);
        free(pred_y);
    }
    if(try_intra) { /* mode_check_intra */
        xo_intra_params ip = P->ip;
        ip.log2_cuw = ip.log2_cuh = log2;
        xo_intra_job ij;
        memset(&ij, 0, sizeof(ij));
        ij.x = x0, ij.y = y0, ij.inter_satd = inter_satd, ij.sbac = 0;
        xo_intra_result ir;
        xo_sbac         best;
        const xo_pel *const mod_c[3] = {T->mod[0], T->mod[1], T->mod[2]};
        int16_t *cf[3];
        xo_pel  *rc[3];
        for(int c = 0; c < 3; c++) cf[c] = malloc(sizeof(int16_t) * (size_t)n0), rc[c] = malloc(sizeof(xo_pel) * (size_t)n0);
        xo_pintra_analyze_cu(T->org, T->s_org_l, T->s_org_c, mod_c, T->s_mod_l, T->s_mod_c, T->map_scu, T->map_ipm, T->map_tidx, &T->curr_best[L], &ip, &ij, &ir, cf[0],
                             cf[1], cf[2], rc[0], rc[1], rc[2], &best);
        if(ir.cost < cost_best) {
            const int n1 = idc ? n0 >> ((idc <= 2) + (idc <= 1)) : 0;
            cost_best = ir.cost, T->cu_mode = 0, T->next_best[L] = best, T->dist_cu_best = ir.dist_cu;
            for(int c = 0; c < (idc ? 3 : 1); c++) memcpy(t->coef[c], cf[c], sizeof(int16_t) * (size_t)(c ? n1 : n0)), memcpy(t->reco[c], rc[c], sizeof(xo_pel) * (size_t)(c ? n1 : n0));
            for(int i = 0; i < n; i++) { /* copy_to_cu_data of an intra CU */
                t->pred_mode[i] = 0 /* MODE_INTRA */, t->ipm[0][i] = ir.ipm[0], t->ipm[1][i] = idc ? ir.ipm[1] : 0, t->depth[i] = (int8_t)cud;
                t->nnz[0][i] = ir.nnz[0], t->nnz[1][i] = idc ? ir.nnz[1] : 0, t->nnz[2][i] = idc ? ir.nnz[2] : 0;
                t->map_scu[i] = scu_base | (1u << 15), t->map_cu_mode[i] = cum;
                memset(t->mv[i], 0, sizeof(t->mv[i])), memset(t->mvd[i], 0, sizeof(t->mvd[i]));
                t->refi[i][0] = t->refi[i][1] = -1, t->mvp_idx[i][0] = t->mvp_idx[i][1] = 0;
            }
        }
        for(int c = 0; c < 3; c++) free(cf[c]), free(rc[c]);
    }
    return cost_best;
}

static double tree_node(tree_ctx *T, int x0, int y0, int log2, int cud, int next_split)
{
    const xo_tree_params *P = T->P;
    const int L = log2 - 2, cu = 1 << log2, idc = P->ip.chroma_format_idc;
    const int boundary = !(x0 + cu <= P->pic_w && y0 + cu <= P->pic_h);
    const xo_sbac before_split = T->curr_best[L];
    xo_sbac       temp_depth;
    double        cost_best = 1.7e+308, cost_temp;
    int           best_split = 0;
    memset(&temp_depth, 0, sizeof(temp_depth));
    if(!boundary) {
        cost_temp = 0.0;
        cud_init(T->temp[L], log2);
        if(cu <= P->max_cu) {
            if(cu > P->min_cuwh) { /* split_cu_flag = 0 (:2079-2091) */
                xo_sbac run;
                cost_temp += (double)(int)tree_split_bits(&T->curr_best[L], &run, 0) * P->ip.lambda[0];
                T->curr_best[L] = run;
            }
            cud_init(T->temp[L], log2);
            tree_clear_map(T, x0, y0, cu);
            xo_ctu_data *t = T->temp[L];
            cost_temp += tree_unit(T, x0, y0, log2, cud, t);
            if(cost_best > cost_temp) { /* (:2116-2135) */
                cud_copy(T->best[L], t, 0, 0, log2, log2, cud, idc);
                cost_best = cost_temp, best_split = 0, temp_depth = T->next_best[L];
                tree_rec_to_pic(T, T->best[L], x0, y0, cu);
            }
            cost_temp = cost_best;
        }
        else cost_temp = 1.7e+308;
    }
    /* early CU termination outside I pictures (:2162-2172): a skipped CU at or below a depth that depends on the POC's parity is not split */
    if(cost_best != 1.7e+308 && T->I && cud >= T->I->ecu_depth && T->cu_mode == 2 /* MODE_SKIP */) next_split = 0;
    /* early termination in I pictures (:2174-2187) */
    if(cost_best != 1.7e+308 && P->ip.slice_type == 2) {
        const int dist_cu = T->dist_cu_best, th = 1 << (2 * log2 + 7);
        if(dist_cu < th) {
            const int bits_inc = (2 * log2 >= 6 ? 2 : 0) + 8;
            if(dist_cu < P->ip.lambda[0] * bits_inc) next_split = 0;
        }
    }
    if(cu > 4 && next_split && cu > P->min_cu && cu > P->min_cuwh) { /* SPLIT_QUAD (:2189-2329) */
        xo_sbac run;
        cud_init(T->temp[L], log2);
        tree_clear_map(T, x0, y0, cu);
        cost_temp = (double)(int)tree_split_bits(&before_split, &run, 1) * P->ip.lambda[0];
        T->curr_best[L] = run;
        cud_init(T->temp[L], log2);
        tree_clear_map(T, x0, y0, cu);
        const int half = cu >> 1;
        int first = 1;
        for(int part = 0; part < 4; part++) {
            const int xp = x0 + (part & 1) * half, yp = y0 + (part >> 1) * half;
            if(xp < P->pic_w && yp < P->pic_h) {
                T->curr_best[L - 1] = part == 0 ? T->curr_best[L] : T->next_best[L - 1];
                (void)first;
                cost_temp += tree_node(T, xp, yp, log2 - 1, cud + 2, 1); /* a quad split counts as two binary levels (xeve_split_get_part_structure, xeve_util.c:1407-1410) */
                cud_copy(T->temp[L], T->best[L - 1], xp - x0, yp - y0, log2 - 1, log2, cud, idc);
                tree_update_map(T, T->best[L - 1], xp, yp, half);
            }
        }
        if(cost_best - 0.0001 > cost_temp) {
            cud_copy(T->best[L], T->temp[L], 0, 0, log2, log2, cud, idc);
            cost_best = cost_temp, temp_depth = T->next_best[L - 1], best_split = 5 /* SPLIT_QUAD */;
        }
    }
    tree_rec_to_pic(T, T->best[L], x0, y0, cu);
    if(cu >= 8) T->best[L]->split_mode[cud][((cu >> 1) >> 2) * (cu >> 2) + ((cu >> 1) >> 2)] = (int8_t)best_split; /* xeve_set_split_mode (xeve_util.c:1148-1161), cup 0, pitch = the node */
    T->next_best[L] = temp_depth;
    return cost_best;
}

static double analyze_ctu(const xo_pel *const org[3], int s_org_l, int s_org_c, xo_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                          const uint8_t *map_tidx, uint32_t *map_cu_mode, const xo_sbac *entry, const xo_tree_params *P, const xo_tree_inter *I, int x0, int y0,
                          xo_ctu_data *out, xo_sbac *next_best)
{
    tree_ctx T;
    memset(&T, 0, sizeof(T));
    T.org = org, T.mod = mod, T.s_org_l = s_org_l, T.s_org_c = s_org_c, T.s_mod_l = s_mod_l, T.s_mod_c = s_mod_c;
    T.map_scu = map_scu, T.map_ipm = map_ipm, T.map_tidx = map_tidx, T.map_cu_mode = map_cu_mode, T.P = P, T.I = I;
    for(int l = 0; l < 5; l++) T.best[l] = calloc(1, sizeof(xo_ctu_data)), T.temp[l] = calloc(1, sizeof(xo_ctu_data));
    const int L = P->log2_ctu - 2;
    T.curr_best[L] = *entry;
    const double cost = tree_node(&T, x0, y0, P->log2_ctu, 0, 1);
    tree_update_map(&T, T.best[L], x0, y0, 1 << P->log2_ctu); /* update_to_ctx_map + update_map_scu (:2455-2516) */
    *out = *T.best[L], *next_best = T.next_best[L];
    for(int l = 0; l < 5; l++) free(T.best[l]), free(T.temp[l]);
    return cost;
}

double xo_mode_analyze_ctu_intra(const xo_pel *const org[3], int s_org_l, int s_org_c, xo_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu,
                                 int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const xo_sbac *entry, const xo_tree_params *P, int x0, int y0,
                                 xo_ctu_data *out, xo_sbac *next_best)
{
    return analyze_ctu(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, entry, P, NULL, x0, y0, out, next_best);
}

double xo_mode_analyze_ctu(const xo_pel *const org[3], int s_org_l, int s_org_c, xo_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                           const uint8_t *map_tidx, uint32_t *map_cu_mode, const xo_sbac *entry, const xo_tree_params *P, const xo_tree_inter *I, int x0, int y0,
                           xo_ctu_data *out, xo_sbac *next_best)
{
    return analyze_ctu(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, entry, P, P->ip.slice_type == 2 ? NULL : I, x0, y0, out,
                       next_best);
}


/* ===================================================================================================================
 * The bitstream writer's side of one CTU: xeve_eco_tree (src_base/xeve_enc.c:35-100) -> xeve_eco_split_mode (xeve_eco.c:1377-1429) and xeve_eco_unit
 * (:1431-1640) for every CU of the decided tree, on the WRITER's coder (no reset between CUs or CTUs: CTU n + 1 starts its mode decision from the state this leaves,
 * xeve_enc.c:139).  Baseline, no delta QP.  The syntax is not the rate estimate's: in P slices the writer codes neither direct_mode_flag nor inter_pred_idc, the
 * estimate (xeve_rdo_bit_cnt_cu_inter, xeve_mode.c:201-274) codes both.  As the CUs are written their units get what xeve_eco_unit stores: the coded flag, the skip
 * flag, the luma cbf flag, the CU's size.  Bytes the coder emits are collected (the pending byte and the code register stay in the state).
 * =================================================================================================================== */
typedef struct eco_ctx {
    xo_sbac              *s;
    const xo_ctu_data    *d;
    const xo_tree_params *P;
    int                   num_refp[2], x0, y0, pitch;
    uint32_t             *map_scu, *map_cu_mode;
    const int8_t         *map_ipm;
    const uint8_t        *map_tidx;
} eco_ctx;

static void eco_unit(eco_ctx *E, int x, int y, int log2, int cup)
{
    const xo_tree_params *P = E->P;
    const xo_ctu_data    *d = E->d;
    xo_sbac *s = E->s;
    const int idc = P->ip.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, cu = 1 << log2, st = P->ip.slice_type, mode = d->pred_mode[cup], skip = mode == 2 /* MODE_SKIP */;
    const int ctu = 1 << P->log2_ctu, lx = x - E->x0, ly = y - E->y0;
    xo_cu_bits_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.log2_cuw = bp.log2_cuh = log2, bp.slice_type = st, bp.num_refp[0] = E->num_refp[0], bp.num_refp[1] = E->num_refp[1], bp.chroma_format_idc = idc;
    if(st != 2) {
        xo_sbac_bin(s, XO_CTX_SKIP_FLAG, skip); /* (ctx_flags: 0 without sps_cm_init_flag) */
        if(skip) {
            sbac_mvp_idx(s, d->mvp_idx[cup][0]);
            if(st == 0) sbac_mvp_idx(s, d->mvp_idx[cup][1]);
        }
        else {
            xo_sbac_bin(s, XO_CTX_PRED_MODE, mode == 0);
            if(mode != 0) {
                if(st == 0) xo_sbac_bin(s, XO_CTX_DIRECT, mode == 3 /* MODE_DIR */);
                if(mode != 3) {
                    const int r0 = d->refi[cup][0], r1 = d->refi[cup][1];
                    if(st == 0) { /* xeve_eco_inter_pred_idc (xeve_eco.c:1123-1156) */
                        if(r0 >= 0 && r1 >= 0) xo_sbac_bin(s, XO_CTX_INTER_DIR, 0);
                        else xo_sbac_bin(s, XO_CTX_INTER_DIR, 1), xo_sbac_bin(s, XO_CTX_INTER_DIR + 1, r0 >= 0 ? 0 : 1);
                    }
                    if(r0 >= 0) sbac_refi(s, E->num_refp[0], r0), sbac_mvp_idx(s, d->mvp_idx[cup][0]), sbac_mvd1(s, d->mvd[cup][0][0]), sbac_mvd1(s, d->mvd[cup][0][1]);
                    if(st == 0 && r1 >= 0) sbac_refi(s, E->num_refp[1], r1), sbac_mvp_idx(s, d->mvp_idx[cup][1]), sbac_mvd1(s, d->mvd[cup][1][0]), sbac_mvd1(s, d->mvd[cup][1][1]);
                }
            }
        }
    }
    if(mode == 0) { /* xeve_get_mpm from the live maps: the units written so far are coded */
        const uint8_t *mpm = xo_get_mpm(x >> 2, y >> 2, E->map_scu, E->map_ipm, E->map_tidx, P->ip.w_scu);
        sbac_unary2(s, mpm[d->ipm[0][cup]], XO_CTX_INTRA_DIR);
    }
    int nnz[3] = {0, 0, 0};
    if(!skip) { /* coef_rect_to_series + xeve_eco_coef(RUN_L | RUN_CB | RUN_CR) */
        static const int run_all[3] = {1, 1, 1};
        const int n0 = cu * cu, cw = cu >> ws, ch = cu >> hs, n1 = idc ? cw * ch : 0;
        int16_t *all = malloc(sizeof(int16_t) * (size_t)(n0 + 2 * n1 + 1));
        for(int j = 0; j < cu; j++) memcpy(all + j * cu, d->coef[0] + (ly + j) * ctu + lx, sizeof(int16_t) * (size_t)cu);
        for(int c = 1; c < 3 && idc; c++)
            for(int j = 0; j < ch; j++) memcpy(all + n0 + (c - 1) * n1 + j * cw, d->coef[c] + ((ly >> hs) + j) * (ctu >> ws) + (lx >> ws), sizeof(int16_t) * (size_t)cw);
        xo_cu_bits_job bj;
        memset(&bj, 0, sizeof(bj));
        bj.coef_off[1] = n0, bj.coef_off[2] = n0 + n1;
        for(int c = 0; c < 3; c++) nnz[c] = bj.nnz[c] = d->nnz[c][cup];
        sbac_coef(s, &bp, &bj, all, run_all, mode == 0, 0);
        free(all);
    }
    for(int j = 0; j < cu >> 2; j++)
        for(int i = 0; i < cu >> 2; i++) {
            const int g = ((y >> 2) + j) * P->ip.w_scu + (x >> 2) + i;
            uint32_t  m = E->map_scu[g];
            m = skip ? m | (1u << 23) : m & ~(1u << 23);       /* MCU_SET_SF / CLR_SF */
            m = nnz[0] > 0 ? m | (1u << 24) : m & ~(1u << 24); /* MCU_SET_CBFL / CLR_CBFL: core->nnz_sub[Y_C][0] */
            E->map_scu[g] = m | (1u << 31);                    /* MCU_SET_COD */
            E->map_cu_mode[g] = (E->map_cu_mode[g] & 0x00FFFFFFu) | ((uint32_t)log2 << 24) | ((uint32_t)log2 << 28);
        }
}

static void eco_node(eco_ctx *E, int x, int y, int log2, int cud, int cup)
{
    const xo_tree_params *P = E->P;
    const int cu = 1 << log2, half = cu >> 1;
    const int split = cu >= 8 ? E->d->split_mode[cud][cup + (half >> 2) * E->pitch + (half >> 2)] : 0; /* xeve_get_split_mode (xeve_util.c:1125-1144) */
    if(split) {
        xo_sbac_bin(E->s, XO_CTX_SPLIT_CU, 1); /* (always coded without sps_btt_flag, also where the picture edge implies it) */
        for(int part = 0; part < 4; part++) {
            const int xp = x + (part & 1) * half, yp = y + (part >> 1) * half;
            if(xp < P->pic_w && yp < P->pic_h) eco_node(E, xp, yp, log2 - 1, cud + 2, cup + (part & 1) * (half >> 2) + (part >> 1) * (half >> 2) * E->pitch);
        }
    }
    else {
        if(cu > 4) xo_sbac_bin(E->s, XO_CTX_SPLIT_CU, 0);
        eco_unit(E, x, y, log2, cup);
    }
}

int xo_eco_ctu(xo_sbac *s, const xo_ctu_data *d, const xo_tree_params *P, const int num_refp[2], uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx,
               uint32_t *map_cu_mode, int x0, int y0, uint8_t *bytes, int bytes_cap)
{
    eco_ctx E;
    memset(&E, 0, sizeof(E));
    E.s = s, E.d = d, E.P = P, E.num_refp[0] = num_refp[0], E.num_refp[1] = num_refp[1], E.x0 = x0, E.y0 = y0, E.pitch = 1 << (P->log2_ctu - 2);
    E.map_scu = map_scu, E.map_cu_mode = map_cu_mode, E.map_ipm = map_ipm, E.map_tidx = map_tidx;
    static uint8_t none[1];
    g_sink = bytes ? bytes : none, g_sink_n = 0, g_sink_cap = bytes ? bytes_cap : 0;
    eco_node(&E, x0, y0, P->log2_ctu, 0, 0);
    const int n = g_sink_n;
    g_sink = NULL, g_sink_n = g_sink_cap = 0;
    return n;
}


/* The end of a tile on the writer's coder: xeve_eco_tile_end_flag(bs, 1) = xeve_sbac_encode_bin_trm (xeve_eco.c:577-595) and xeve_sbac_finish (:622-672).  Returns the
 * bytes that come out (the first cap of them stored): what the coder still held, then -- where no pending byte is left and fewer than four code bits remain -- the
 * zero bits up to the byte boundary. */
int xo_eco_tile_end(xo_sbac *s, uint8_t *bytes, int cap)
{
    static uint8_t none[1];
    g_sink = bytes ? bytes : none, g_sink_n = 0, g_sink_cap = bytes ? cap : 0;
    s->bin_counter++;
    s->range--;
    s->code += s->range, s->range = 1; /* bin = 1 */
    while(s->range < 8192) s->range <<= 1, sbac_shift(s);
    uint32_t tmp = (s->code + s->range - 1) & (0xFFFFFFFFu << 14);
    if(tmp < s->code) tmp += 8192;
    s->code = tmp << s->code_bits;
    sbac_carry(s);
    s->code <<= 8;
    sbac_carry(s);
    for(; s->stacked_zero; s->stacked_zero--) sink_put(0x00);
    if(s->pending_byte != 0) sink_put((uint8_t)s->pending_byte);
    else if(s->code_bits < 4) sink_put(0x00); /* 4 - code_bits zero bits, then zero bits to the byte boundary: one zero byte */
    const int n = g_sink_n;
    g_sink = NULL, g_sink_n = g_sink_cap = 0;
    return n;
}

/* ==================================================================================================================================================================
 * Main profile: the adaptive loop filter's sample kernels (src_main/xevem_alf.c).  TEST INFRASTRUCTURE like the rest of this file.
 * ================================================================================================================================================================== */
void xo_alf_copy_and_extend(xo_pel *tmp, int s_tmp, const xo_pel *rec, int s_rec, int w, int h, int m)
{ /* xevem_alf.c:91-168: rows copied, left / right margins from the row's end samples, then whole bottom / top rows (margins included) repeated */
    for(int r = 0; r < h; r++) {
        xo_pel *d = tmp + (ptrdiff_t)r * s_tmp;
        memcpy(d, rec + (ptrdiff_t)r * s_rec, sizeof(xo_pel) * (size_t)w);
        for(int k = 1; k <= m; k++) d[-k] = d[0], d[w - 1 + k] = d[w - 1];
    }
    for(int k = 1; k <= m; k++) {
        memcpy(tmp + (ptrdiff_t)(h - 1 + k) * s_tmp - m, tmp + (ptrdiff_t)(h - 1) * s_tmp - m, sizeof(xo_pel) * (size_t)(w + 2 * m));
        memcpy(tmp - (ptrdiff_t)k * s_tmp - m, tmp - m, sizeof(xo_pel) * (size_t)(w + 2 * m));
    }
}

/* alf_derive_classification_blk (:488-654) of one piece of at most 32 x 32: 2x2-subsampled sums of four 1-D Laplacians, summed over the 8x8 window around every 4x4 block */
static void xo_alf_classify_piece(uint8_t *classifier, int s_cls, const xo_pel *src, int s_src, int px, int py, int pw, int ph, int bit_depth)
{
    static const int th[16] = {0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4};
    static const int trans_tbl[8] = {0, 1, 0, 2, 2, 3, 1, 3};
    enum { XV = 0, XH = 1, XD0 = 2, XD1 = 3 };
    int lap[4][32 + 5][32 + 5];
    const int rows = ph + 4, cols = pw + 4;
    for(int i = 0; i < rows; i += 2) {
        /* the 2x2 group of samples (r, c) .. (r + 1, c + 1), r = py - 2 + i, c = px - 2 + j */
        const xo_pel *rm = src + (ptrdiff_t)(py - 3 + i) * s_src, *r0 = rm + s_src, *r1 = r0 + s_src, *r2 = r1 + s_src;
        for(int j = 0; j < cols; j += 2) {
            const int c = px - 2 + j;
            /* (xo_pel)(v << 1): the doubled centre is held in a pel in the reference (:534-537) */
            const int a0 = (xo_pel)(r0[c] << 1), a1 = (xo_pel)(r0[c + 1] << 1), b0 = (xo_pel)(r1[c] << 1), b1 = (xo_pel)(r1[c + 1] << 1);
            lap[XV][i][j]  = abs(a0 - rm[c] - r1[c]) + abs(a1 - rm[c + 1] - r1[c + 1]) + abs(b0 - r0[c] - r2[c]) + abs(b1 - r0[c + 1] - r2[c + 1]);
            lap[XH][i][j]  = abs(a0 - r0[c + 1] - r0[c - 1]) + abs(a1 - r0[c + 2] - r0[c]) + abs(b0 - r1[c + 1] - r1[c - 1]) + abs(b1 - r1[c + 2] - r1[c]);
            lap[XD0][i][j] = abs(a0 - rm[c - 1] - r1[c + 1]) + abs(a1 - rm[c] - r1[c + 2]) + abs(b0 - r0[c - 1] - r2[c + 1]) + abs(b1 - r0[c] - r2[c + 2]);
            lap[XD1][i][j] = abs(a0 - r1[c - 1] - rm[c + 1]) + abs(a1 - r1[c] - rm[c + 2]) + abs(b0 - r2[c - 1] - r0[c + 1]) + abs(b1 - r2[c] - r0[c + 2]);
            if(j > 4 && (j - 6) % 4 == 0) /* every fourth group closes a run of four: their sum lands in the run's first entry (:551-559) */
                for(int d = 0; d < 4; d++) lap[d][i][j - 6] += lap[d][i][j - 4] + lap[d][i][j - 2] + lap[d][i][j];
        }
    }
    for(int i = 0; i < ph; i += 4)
        for(int j = 0; j < pw; j += 4) {
            int sum[4];
            for(int d = 0; d < 4; d++) sum[d] = lap[d][i][j] + lap[d][i + 2][j] + lap[d][i + 4][j] + lap[d][i + 6][j];
            const int sv = sum[XV], sh = sum[XH], sd0 = sum[XD0], sd1 = sum[XD1];
            int act = (sv + sh) >> (bit_depth - 2);
            act = (xo_pel)(act < 0 ? 0 : act > 15 ? 15 : act);
            int cls = th[act];
            const int hv1 = sv > sh ? sv : sh, hv0 = sv > sh ? sh : sv, dir_hv = sv > sh ? 1 : 3;
            const int d1 = sd0 > sd1 ? sd0 : sd1, d0 = sd0 > sd1 ? sd1 : sd0, dir_d = sd0 > sd1 ? 0 : 2;
            /* `d1 * hv0 > hv1 * d0` in int arithmetic (:607): on noise the products pass 2^31 and wrap, and the compiled reference compares the wrapped values */
            const int32_t pa = (int32_t)((uint32_t)d1 * (uint32_t)hv0), pb = (int32_t)((uint32_t)hv1 * (uint32_t)d0);
            const int diag = pa > pb;
            const int hvd1 = diag ? d1 : hv1, hvd0 = diag ? d0 : hv0, main_dir = diag ? dir_d : dir_hv, sec_dir = diag ? dir_hv : dir_d;
            int strength = 0;
            if(hvd1 > 2 * hvd0) strength = 1;
            if(hvd1 * 2 > 9 * hvd0) strength = 2;
            if(strength) cls += (((main_dir & 1) << 1) + strength) * 5;
            const uint8_t v = (uint8_t)(((cls << 2) + trans_tbl[main_dir * 2 + (sec_dir >> 1)]) & 0xFF);
            for(int a = 0; a < 4; a++)
                for(int b = 0; b < 4; b++) classifier[(ptrdiff_t)(py + i + a) * s_cls + px + j + b] = v;
        }
}
void xo_alf_classify(uint8_t *classifier, int s_cls, const xo_pel *src, int s_src, const xo_alf_area *blk, int bit_depth)
{ /* alf_derive_classification (:463-486) */
    for(int i = blk->y; i < blk->y + blk->h; i += 32)
        for(int j = blk->x; j < blk->x + blk->w; j += 32) {
            const int h = (i + 32 < blk->y + blk->h ? i + 32 : blk->y + blk->h) - i, w = (j + 32 < blk->x + blk->w ? j + 32 : blk->x + blk->w) - j;
            xo_alf_classify_piece(classifier, s_cls, src, s_src, j, i, w, h, bit_depth);
        }
}
void xo_alf_filter7(const uint8_t *classifier, int s_cls, xo_pel *dst, int s_dst, const xo_pel *src, int s_src, const xo_alf_area *blk, const int16_t *filter_set,
                    int clip_min, int clip_max)
{ /* alf_filter_blk_7 (:656-787): per 4x4 block the class's 13 coefficients in the block's transposition, a 7x7 diamond with point symmetry, rounding shift 9 */
    static const int order[4][13] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, {9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12}, {0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12},
                                     {9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12}};
    for(int i = 0; i < blk->h; i += 4)
        for(int j = 0; j < blk->w; j += 4) {
            const uint8_t cl = classifier[(ptrdiff_t)(blk->y + i) * s_cls + blk->x + j];
            const int16_t *set = filter_set + ((cl >> 2) & 0x1F) * 13;
            xo_pel c[13];
            for(int k = 0; k < 13; k++) c[k] = set[order[cl & 3][k]];
            for(int a = 0; a < 4; a++)
                for(int b = 0; b < 4; b++) {
                    const xo_pel *p = src + (ptrdiff_t)(i + a) * s_src + j + b;
                    const ptrdiff_t s = s_src;
                    int sum = c[0] * (p[3 * s] + p[-3 * s]);
                    sum += c[1] * (p[2 * s + 1] + p[-2 * s - 1]) + c[2] * (p[2 * s] + p[-2 * s]) + c[3] * (p[2 * s - 1] + p[-2 * s + 1]);
                    sum += c[4] * (p[s + 2] + p[-s - 2]) + c[5] * (p[s + 1] + p[-s - 1]) + c[6] * (p[s] + p[-s]) + c[7] * (p[s - 1] + p[-s + 1]) + c[8] * (p[s - 2] + p[-s + 2]);
                    sum += c[9] * (p[3] + p[-3]) + c[10] * (p[2] + p[-2]) + c[11] * (p[1] + p[-1]) + c[12] * p[0];
                    sum = (sum + 256) >> 9;
                    dst[(ptrdiff_t)(i + a) * s_dst + j + b] = (xo_pel)(sum < clip_min ? clip_min : sum > clip_max ? clip_max : sum);
                }
        }
}
void xo_alf_filter5(xo_pel *dst, int s_dst, const xo_pel *src, int s_src, const xo_alf_area *blk, const int16_t *filter_set, int clip_min, int clip_max)
{ /* alf_filter_blk_5 (:789-882): one 7-coefficient filter for the whole area, a 5x5 diamond */
    xo_pel c[7];
    for(int k = 0; k < 7; k++) c[k] = filter_set[k];
    for(int i = 0; i < blk->h; i++)
        for(int j = 0; j < blk->w; j++) {
            const xo_pel *p = src + (ptrdiff_t)i * s_src + j;
            const ptrdiff_t s = s_src;
            int sum = c[0] * (p[2 * s] + p[-2 * s]) + c[1] * (p[s + 1] + p[-s - 1]) + c[2] * (p[s] + p[-s]) + c[3] * (p[s - 1] + p[-s + 1]);
            sum += c[4] * (p[2] + p[-2]) + c[5] * (p[1] + p[-1]) + c[6] * p[0];
            sum = (sum + 256) >> 9;
            dst[(ptrdiff_t)i * s_dst + j] = (xo_pel)(sum < clip_min ? clip_min : sum > clip_max ? clip_max : sum);
        }
}
/* xeve_alf_clac_covariance (:3890-3952): the sums of the sample pairs every coefficient multiplies, in the block's transposition */
static void xo_alf_local(int *e, const xo_pel *rec, int stride, const int *pattern, int half, int trans_idx)
{
    int k = 0;
    const ptrdiff_t s = stride;
    if(trans_idx == 0 || trans_idx == 2) { /* rows above the centre with their mirror rows; transposition 2 walks every row right to left */
        for(int i = -half; i < 0; i++)
            for(int t = -half - i; t <= half + i; t++) {
                const int j = trans_idx == 0 ? t : -t;
                e[pattern[k++]] += rec[i * s + j] + rec[-i * s - j];
            }
        for(int j = -half; j < 0; j++) e[pattern[k++]] += rec[j] + rec[-j];
    }
    else { /* 1, 3: the same walk with rows and columns exchanged */
        for(int j = -half; j < 0; j++)
            for(int t = -half - j; t <= half + j; t++) {
                const int i = trans_idx == 1 ? t : -t;
                e[pattern[k++]] += rec[i * s + j] + rec[-i * s - j];
            }
        for(int i = -half; i < 0; i++) e[pattern[k++]] += rec[i * s] + rec[-i * s];
    }
    e[pattern[k++]] += rec[0];
}
void xo_alf_blk_stats(int taps, const uint8_t *classifier, int s_cls, const xo_pel *org, int s_org, const xo_pel *rec, int s_rec, int x, int y, int w, int h,
                      double *E, double *yv, double *pix)
{ /* xeve_alf_get_blk_stats (:3836-3888) */
    static const int pattern5[13] = {0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0}, pattern7[25] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0}; /* xevem_alf.h:118-136 */
    const int ncoef = taps * taps / 4 + 1, *pattern = taps == 5 ? pattern5 : pattern7, nclasses = classifier ? 25 : 1;
    for(int i = 0; i < h; i++)
        for(int j = 0; j < w; j++) {
            int e[13] = {0}, cls = 0, tr = 0;
            if(classifier) {
                const uint8_t cl = classifier[(ptrdiff_t)(y + i) * s_cls + x + j];
                tr = cl & 3, cls = (cl >> 2) & 0x1F;
            }
            const xo_pel *r = rec + (ptrdiff_t)(y + i) * s_rec + x + j;
            const int d = org[(ptrdiff_t)(y + i) * s_org + x + j] - r[0];
            xo_alf_local(e, r, s_rec, pattern, taps >> 1, tr);
            for(int k = 0; k < ncoef; k++) {
                for(int l = k; l < ncoef; l++) E[(cls * 13 + k) * 13 + l] += e[k] * e[l];
                yv[cls * 13 + k] += e[k] * d;
            }
            pix[cls] += d * d;
        }
    for(int c = 0; c < nclasses; c++) /* (:3880-3887) the lower triangle from the upper one */
        for(int k = 1; k < ncoef; k++)
            for(int l = 0; l < k; l++) E[(c * 13 + k) * 13 + l] = E[(c * 13 + l) * 13 + k];
}

/* ==================================================================================================================================================================
 * Main profile: affine motion compensation (src_main/xevem_mc.c:1457-2339, xevem_util.c:1190-1480).  TEST INFRASTRUCTURE like the rest of this file.
 * ================================================================================================================================================================== */
static int xa_log2(int v) { int l = 0; while((1 << l) < v) l++; return l; }
static int xa_round(int v, int rs) { return (v + (rs > 0 ? 1 << (rs - 1) : 0) - (v >= 0)) >> rs; } /* xeve_rounding_s32 with left_shift 0 (xevem_util.c:1197-1201) */
/* the model's per-sample increments at `bit` extra bits: calculate_affine_motion_model_parameters (xevem_util.c:1331-1356) = the head of xeve_affine_mc_lc (:1717-1728) */
static void xa_model(const int16_t mv[3][2], int cuw, int cuh, int vertex_num, int bit, int d_hor[2], int d_ver[2])
{
    for(int c = 0; c < 2; c++) d_hor[c] = ((mv[1][c] - mv[0][c]) << bit) >> xa_log2(cuw);
    if(vertex_num == 3)
        for(int c = 0; c < 2; c++) d_ver[c] = ((mv[2][c] - mv[0][c]) << bit) >> xa_log2(cuh);
    else d_ver[0] = -d_hor[1], d_ver[1] = d_hor[0];
}
/* check_eif_applicability_uni (xevem_util.c:1421-1449): the 4x4 sub-block's bounding box in the reference picture (memory bandwidth) and the lines its first row fetches */
static int xa_eif_ok_uni(const int16_t mv[3][2], int cuw, int cuh, int vertex_num, int *mem_ok)
{
    const int prec = 2 + 7;
    int d_hor[2], d_ver[2], mx[2], mn[2], diff[2];
    xa_model(mv, cuw, cuh, vertex_num, 7, d_hor, d_ver);
    for(int c = 0; c < 2; c++) { /* calculate_bounding_box_size (:1358-1404) with w = h = EIF_SUBBLOCK_SIZE 4 */
        const int c1 = 5 * (d_hor[c] + (c == 0 ? 1 << prec : 0)), c2 = 5 * (d_ver[c] + (c == 1 ? 1 << prec : 0)), c3 = c1 + c2;
        mx[c] = 0, mn[c] = 0;
        if(c1 > mx[c]) mx[c] = c1;
        if(c2 > mx[c]) mx[c] = c2;
        if(c3 > mx[c]) mx[c] = c3;
        if(c1 < mn[c]) mn[c] = c1;
        if(c2 < mn[c]) mn[c] = c2;
        if(c3 < mn[c]) mn[c] = c3;
        diff[c] = (mx[c] - mn[c] + (1 << prec) - 1) >> prec;
    }
    *mem_ok = (diff[0] + 2) * (diff[1] + 2) <= 72; /* MAX_MEMORY_ACCESS_BI */
    if(d_ver[1] < -(1 << prec)) return 0; /* check_eif_num_fetched_lines_restrictions (:1406-1419) */
    if(((d_ver[1] > 0 ? d_ver[1] : 0) + abs(d_hor[1])) * (1 + 4) > (3 - 2) << prec) return 0;
    return 1;
}
static void xa_subblock_size(const xo_affine_job *j, int cuw, int cuh, int *sub_w, int *sub_h, int *mem_ok)
{ /* derive_affine_subblock_size_bi (xevem_util.c:1203-1272) */
    static const int lut[4] = {32, 16, 8, 8};
    int eif = 1;
    *sub_w = cuw, *sub_h = cuh, *mem_ok = 1;
    for(int l = 0; l < 2; l++) {
        if(j->refi[l] < 0) continue;
        int d_hor[2], d_ver[2];
        xa_model(j->mv[l], cuw, cuh, j->vertex_num, 7, d_hor, d_ver);
        const int wx = abs(d_hor[0]) > abs(d_hor[1]) ? abs(d_hor[0]) : abs(d_hor[1]), wy = abs(d_ver[0]) > abs(d_ver[1]) ? abs(d_ver[0]) : abs(d_ver[1]);
        const int w = wx > 4 ? 4 : wx == 0 ? cuw : lut[wx - 1], h = wy > 4 ? 4 : wy == 0 ? cuh : lut[wy - 1];
        if(w < *sub_w) *sub_w = w;
        if(h < *sub_h) *sub_h = h;
    }
    for(int l = 0; l < 2 && eif; l++) { /* check_eif_applicability_bi (:1451-1480): stops at the first list that fails (the later list's bandwidth answer is then not taken) */
        if(j->refi[l] < 0) continue;
        int m = 0;
        const int ok = xa_eif_ok_uni(j->mv[l], cuw, cuh, j->vertex_num, &m);
        *mem_ok &= m;
        if(!ok) eif = 0;
    }
    if(!eif) {
        if(*sub_w < 8) *sub_w = 8;
        if(*sub_h < 8) *sub_h = 8;
    }
}
/* xeve_eif_mc (xevem_mc.c:2123-2234) of one component: bw x bh samples at (x, y) of `ref` */
static void xa_eif(int bw, int bh, int x, int y, const int mv0_in[2], const int dx_in[2], const int dy_in[2], const int mv_max_in[2], const int mv_min_in[2], const xo_pel *ref,
                   int s_ref, xo_pel *dst, int s_dst, int chroma, int bit_depth)
{
    int mv0[2] = {mv0_in[0], mv0_in[1]}, mx[2] = {mv_max_in[0], mv_max_in[1]}, mn[2] = {mv_min_in[0], mv_min_in[1]};
    if(chroma) { /* (:2163-2175; the per-sample increments are NOT halved: a chroma sample is two luma samples apart) */
        for(int c = 0; c < 2; c++) mv0[c] >>= 1, mx[c] >>= 1, mn[c] >>= 1;
        bw >>= 1, bh >>= 1, x >>= 1, y >>= 1;
    }
    ref += (ptrdiff_t)s_ref * y + x;
    const int ts = 128 + 2, sh2 = bit_depth + 5 - 16 > 0 ? bit_depth + 5 - 16 : 0, sh3 = 6 - sh2, of2 = sh2 > 0 ? 1 << (sh2 - 1) : 0, of3 = 1 << (sh3 - 1);
    const int s1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, s2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8, o2 = 1 << (s2 - 1);
    xo_pel *buf = (xo_pel *)malloc(sizeof(xo_pel) * (size_t)ts * ts);
    /* can_mv_clipping_occurs (:1917-1957): the vectors of the four corners of the (bw + 2) x (bh + 2) positions against the range */
    int clip = 0;
    for(int c = 0; c < 2; c++) {
        const int m = mv0[c] - dx_in[c] - dy_in[c], k[4] = {m >> 4, (m + (bw + 1) * dx_in[c]) >> 4, (m + (bh + 1) * dy_in[c]) >> 4, (m + (bw + 1) * dx_in[c] + (bh + 1) * dy_in[c]) >> 4};
        for(int q = 0; q < 4; q++)
            if(k[q] > mx[c] || k[q] < mn[c]) clip = 1;
    }
    /* xeve_eif_bilinear_clip / _no_clip (:1991-2121): position (px, py) in -1 .. bw / bh takes the sample its own vector (1/32 pel after >> 4) points at */
    for(int py = -1; py <= bh; py++)
        for(int px = -1; px <= bw; px++) {
            int v[2];
            for(int c = 0; c < 2; c++) {
                v[c] = (mv0[c] + px * dx_in[c] + py * dy_in[c]) >> 4;
                if(clip) v[c] = v[c] < mn[c] ? mn[c] : v[c] > mx[c] ? mx[c] : v[c];
            }
            const xo_pel *r = ref + (ptrdiff_t)(py + (v[1] >> 5)) * s_ref + px + (v[0] >> 5);
            const int fx = v[0] & 31, fy = v[1] & 31;
            const xo_pel a = (xo_pel)(((64 - 2 * fx) * r[0] + 2 * fx * r[1]) >> s1), b = (xo_pel)(((64 - 2 * fx) * r[s_ref] + 2 * fx * r[s_ref + 1]) >> s1);
            buf[(py + 1) * ts + px + 1] = (xo_pel)(((64 - 2 * fy) * a + 2 * fy * b + o2) >> s2);
        }
    /* xeve_eif_filter (:1959-1989): {-1, 10, -1} along the rows in place (the result one column to the left), then down the columns */
    for(int r = 0; r <= bh + 1; r++) {
        xo_pel *t = buf + r * ts;
        for(int c = 0; c < bw; c++) t[c] = (xo_pel)((-t[c] + t[c + 1] * 10 - t[c + 2] + of2) >> sh2);
    }
    for(int r = 0; r < bh; r++)
        for(int c = 0; c < bw; c++) {
            const xo_pel *t = buf + (r + 1) * ts + c;
            const xo_pel res = (xo_pel)((-t[-ts] + t[0] * 10 - t[ts] + of3) >> sh3);
            dst[r * s_dst + c] = (xo_pel)clip3i(0, (1 << bit_depth) - 1, res);
        }
    free(buf);
}
/* xeve_affine_mc_lc (xevem_mc.c:1671-1915) */
static void xa_mc_lc(int x, int y, int pic_w, int pic_h, int cuw, int cuh, const int16_t mv[3][2], const xo_refpic *rp, int s_l, int s_c, xo_pel *pred[3], int vertex_num,
                     int sub_w, int sub_h, int mem_ok, int bit_depth)
{
    const int bit = 7, mc_prec = 4, shift = bit - 2;
    const int msh = mv[0][0] << bit, msv = mv[0][1] << bit;
    int d_hor[2], d_ver[2];
    xa_model(mv, cuw, cuh, vertex_num, bit, d_hor, d_ver);
    if(sub_w < 8 || sub_h < 8) { /* AFFINE_ADAPT_EIF_SIZE */
        /* eif_derive_mv_clip_range (:1481-1530) */
        const int max_pic[2] = {(pic_w + 128 - x - cuw - 1) << 5, (pic_h + 128 - y - cuh - 1) << 5}, min_pic[2] = {(-x - 128) << 5, (-y - 128) << 5};
        static const int dev[5] = {128, 256, 544, 1120, 2272};
        const int scale[2] = {msh, msv}, centre[2] = {cuw >> 1, cuh >> 1};
        int mx[2], mn[2];
        for(int c = 0; c < 2; c++) {
            if(mem_ok) mx[c] = max_pic[c], mn[c] = min_pic[c];
            else {
                const int mid = xa_round(scale[c] + d_hor[c] * centre[0] + d_ver[c] * centre[1], 4), spread = dev[xa_log2(c == 0 ? cuw : cuh) - 3];
                mn[c] = mid - spread, mx[c] = mid + spread;
                if(mn[c] < min_pic[c]) mn[c] = min_pic[c], mx[c] = max_pic[c] < min_pic[c] + 2 * spread ? max_pic[c] : min_pic[c] + 2 * spread;
                else if(mx[c] > max_pic[c]) mx[c] = max_pic[c], mn[c] = min_pic[c] > max_pic[c] - 2 * spread ? min_pic[c] : max_pic[c] - 2 * spread;
            }
            mx[c] = clip3i(-(1 << 17), (1 << 17) - 1, mx[c]), mn[c] = clip3i(-(1 << 17), (1 << 17) - 1, mn[c]);
        }
        /* (affine_mv_prec = bit + 2 = EIF_MV_PRECISION_INTERNAL: no further shift of the model, :2150-2156) */
        xa_eif(cuw, cuh, x, y, scale, d_hor, d_ver, mx, mn, rp->y, s_l, pred[0], cuw, 0, bit_depth);
        if(!pred[1]) return; /* (xeve_affine_mc_l, :1592-1634: the luma plane alone) */
        xa_eif(cuw, cuh, x, y, scale, d_hor, d_ver, mx, mn, rp->u, s_c, pred[1], cuw >> 1, 1, bit_depth);
        xa_eif(cuw, cuh, x, y, scale, d_hor, d_ver, mx, mn, rp->v, s_c, pred[2], cuw >> 1, 1, bit_depth);
        return;
    }
    const int hor_max = (pic_w + 128 - x - cuw) << mc_prec, ver_max = (pic_h + 128 - y - cuh) << mc_prec, hor_min = (-128 - x) << mc_prec, ver_min = (-128 - y) << mc_prec;
    const int half_w = sub_w >> 1, half_h = sub_h >> 1;
    for(int h = 0; h < cuh; h += sub_h)
        for(int w = 0; w < cuw; w += sub_w) {
            /* (the reference adds half_w / half_h to a running position that never moves, :1829-1830: every sub-block takes the FIRST sub-block's centre -- see below) */
            int th = xa_round(msh + d_hor[0] * half_w + d_ver[0] * half_h, shift), tv = xa_round(msv + d_hor[1] * half_w + d_ver[1] * half_h, shift);
            th = clip3i(-(1 << 17), (1 << 17) - 1, th), tv = clip3i(-(1 << 17), (1 << 17) - 1, tv);
            const int oh = th, ov = tv;
            th = th < hor_min ? hor_min : th > hor_max ? hor_max : th, tv = tv < ver_min ? ver_min : tv > ver_max ? ver_max : tv;
            const int gx = ((x + w) << mc_prec) + th, gy = ((y + h) << mc_prec) + tv;
            xo_mc_l((oh & 15) != 0, (ov & 15) != 0, rp->y, gx, gy, s_l, cuw, pred[0] + h * cuw + w, sub_w, sub_h, bit_depth, xom_mc_l_coeff);
            if(!pred[1]) continue; /* (xeve_affine_mc_l, :1636-1668) */
            xo_mc_c((oh & 31) != 0, (ov & 31) != 0, rp->u, gx, gy, s_c, cuw >> 1, pred[1] + (h >> 1) * (cuw >> 1) + (w >> 1), sub_w >> 1, sub_h >> 1, bit_depth, xom_mc_c_coeff);
            xo_mc_c((oh & 31) != 0, (ov & 31) != 0, rp->v, gx, gy, s_c, cuw >> 1, pred[2] + (h >> 1) * (cuw >> 1) + (w >> 1), sub_w >> 1, sub_h >> 1, bit_depth, xom_mc_c_coeff);
        }
}
void xo_affine_mc(const xo_refpic *refp, int s_l, int s_c, int pic_w, int pic_h, const xo_affine_job *job, int w, int h, int bit_depth, xo_pel *pred_y, xo_pel *pred_u,
                  xo_pel *pred_v, int *path)
{
    int sub_w, sub_h, mem_ok, n = 0;
    xa_subblock_size(job, w, h, &sub_w, &sub_h, &mem_ok);
    if(path) path[0] = sub_w, path[1] = sub_h, path[2] = mem_ok;
    const size_t nl = (size_t)w * h, nc = nl / 4;
    xo_pel *second = (xo_pel *)malloc(sizeof(xo_pel) * (nl + 2 * nc));
    for(int l = 0; l < 2; l++) {
        if(job->refi[l] < 0) continue;
        xo_pel *p[3] = {n ? second : pred_y, n ? second + nl : pred_u, n ? second + nl + nc : pred_v};
        xa_mc_lc(job->x, job->y, pic_w, pic_h, w, h, job->mv[l], &refp[job->refi[l] * 2 + l], s_l, s_c, p, job->vertex_num, sub_w, sub_h, mem_ok, bit_depth);
        n++;
    }
    if(n == 2) { /* (:2305-2338) */
        for(size_t i = 0; i < nl; i++) pred_y[i] = (xo_pel)((pred_y[i] + second[i] + 1) >> 1);
        for(size_t i = 0; i < nc; i++) pred_u[i] = (xo_pel)((pred_u[i] + second[nl + i] + 1) >> 1), pred_v[i] = (xo_pel)((pred_v[i] + second[nl + nc + i] + 1) >> 1);
    }
    free(second);
}

/* ==================================================================================================================================================================
 * Main profile: the affine gradient search (src_main/xevem_pinter.c:4213-4501).  TEST INFRASTRUCTURE like the rest of this file.
 * ================================================================================================================================================================== */
/* xeve_affine_mc_l (xevem_mc.c:1532-1669) = derive_affine_subblock_size (xevem_util.c:1273-1330: one list's model) + the luma part of the per-list compensation */
void xo_affine_mc_l(const xo_pel *ref_y, int s_l, int pic_w, int pic_h, int x, int y, const int16_t mv[3][2], int vertex_num, int w, int h, int bit_depth, xo_pel *pred)
{
    xo_affine_job j;
    memset(&j, 0, sizeof(j));
    j.x = x, j.y = y, j.refi[0] = 0, j.refi[1] = -1, j.vertex_num = (int8_t)vertex_num;
    memcpy(j.mv[0], mv, sizeof(j.mv[0]));
    int sub_w, sub_h, mem_ok;
    xa_subblock_size(&j, w, h, &sub_w, &sub_h, &mem_ok);
    xo_refpic rp;
    memset(&rp, 0, sizeof(rp));
    rp.y = ref_y;
    xo_pel *p[3] = {pred, NULL, NULL};
    xa_mc_lc(x, y, pic_w, pic_h, w, h, mv, &rp, s_l, 0, p, vertex_num, sub_w, sub_h, mem_ok, bit_depth);
}
void xo_affine_solve(double (*eq)[7], int order, double *para)
{ /* solve_equal (:4213-4255) */
    for(int i = 1; i < order; i++) {
        double best = fabs(eq[i][i - 1]);
        int    at = i;
        for(int j = i + 1; j < order + 1; j++)
            if(fabs(eq[j][i - 1]) > best) best = fabs(eq[j][i - 1]), at = j;
        if(at != i)
            for(int j = 0; j < order + 1; j++) eq[0][j] = eq[i][j], eq[i][j] = eq[at][j], eq[at][j] = eq[0][j];
        for(int j = i + 1; j < order + 1; j++)
            for(int k = i; k < order + 1; k++) eq[j][k] = eq[j][k] - eq[i][k] * eq[j][i - 1] / eq[i][i - 1];
    }
    para[order - 1] = eq[order][order] / eq[order][order - 1];
    for(int i = order - 2; i >= 0; i--) {
        double t = 0;
        for(int j = i + 1; j < order; j++) t += eq[i + 1][j] * para[j];
        para[i] = (eq[i + 1][order] - t) / eq[i + 1][i];
    }
}
/* one component of get_affine_mv_bits: xeve_tbl_mv_bits inside (-2048, 2048], the MAIN profile's exp-Golomb length beyond (xevem_pinter.c:217-237: other than the
 * Baseline one -- every bit of the prefix counted from 0, the sign only with a non-zero value) */
static int xa_mvd_bits(int mvd)
{
    if(mvd > 2048 || mvd <= -2048) {
        unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd), nn = (a + 1) >> 1;
        int len_i;
        for(len_i = 0; len_i < 16 && nn != 0; len_i++) nn >>= 1;
        return (len_i << 1) + 1 + (a ? 1 : 0);
    }
    return mvd_bits(mvd);
}
static int xa_refi_bits(int num_refp, int refi) { return num_refp <= 1 ? 0 : refi == num_refp - 1 ? refi : refi + 1; } /* xeve_tbl_refi_bits (xeve_tbl.c:498-516) */
static int xa_mv_bits(const int16_t mv[3][2], const int16_t mvp[3][2], int num_refp, int refi, int vertex_num)
{ /* get_affine_mv_bits (:4257-4288) */
    int zero = 1, bits = 1;
    for(int v = 0; v < vertex_num; v++)
        if(mv[v][0] != mvp[v][0] || mv[v][1] != mvp[v][1]) { zero = 0; break; }
    if(zero) return bits;
    for(int v = 0; v < vertex_num; v++) {
        int dx = mv[v][0] - mvp[v][0], dy = mv[v][1] - mvp[v][1];
        if(v) dx -= mv[0][0] - mvp[0][0], dy -= mv[0][1] - mvp[0][1];
        bits += xa_mvd_bits(dx) + xa_mvd_bits(dy);
    }
    return bits + xa_refi_bits(num_refp, refi);
}
/* (s16)(double): what the reference's build does with the conversion -- cvttsd2si to 32 bits (0x80000000 for NaN and values beyond the range), the low 16 bits kept */
static int16_t xa_to_s16(double v)
{
    const int32_t t = (v >= -2147483648.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
    return (int16_t)(uint16_t)(uint32_t)t;
}
void xo_affine_me_gradient(const xo_refpic *refp, int s_l, int pic_w, int pic_h, const int16_t *org_in, int s_org_in, xo_affine_me_job *job, int w, int h, int bit_depth,
                           uint32_t lambda_mv, int num_refp)
{
    const int bi = job->bi, vn = job->vertex_num, ri = job->refi, np = vn << 1;
    const xo_pel *ref = refp[ri * 2 + job->list].y;
    const int16_t *org = bi ? org_in : org_in + (ptrdiff_t)job->y * s_org_in + job->x;
    const int s_org = bi ? w : s_org_in;
    xo_pel  *pred = (xo_pel *)malloc(sizeof(xo_pel) * w * h), *err = (xo_pel *)malloc(sizeof(xo_pel) * w * h);
    int32_t *der[2] = {(int32_t *)malloc(sizeof(int32_t) * w * h), (int32_t *)malloc(sizeof(int32_t) * w * h)};
    int16_t mvt[3][2], mvd[3][2];
    memcpy(mvt, job->mv, sizeof(mvt)), memset(mvd, 0, sizeof(mvd));
#define XA_MV_COST(bits) ((uint32_t)((lambda_mv * (uint32_t)(bits) + (1 << 15)) >> 16)) /* MV_COST (:53): 32-bit unsigned arithmetic */
    xo_affine_mc_l(ref, s_l, pic_w, pic_h, job->x, job->y, mvt, vn, w, h, bit_depth, pred);
    int best_bits = xa_mv_bits(mvt, job->mvp, num_refp, ri, vn) + (bi ? job->mot_bits_other : 0);
    uint32_t cost_best = XA_MV_COST(best_bits) + (uint32_t)(xo_satd(w, h, org, pred, s_org, w, bit_depth) >> bi);
    int rounds = bi ? 5 : 7; /* AF_ITER_BI / AF_ITER_UNI (xeve_def.h:168-169) */
    if(vn == 3) rounds -= 2;
    for(int it = 0; it < rounds; it++) {
        for(int r = 0; r < h; r++)
            for(int c = 0; c < w; c++) err[r * w + c] = (xo_pel)(org[r * s_org + c] - pred[r * w + c]); /* xeve_diff_16b */
        xo_sobel(0, pred, w, der[0], w, w, h), xo_sobel(1, pred, w, der[1], w, w, h);
        int64_t eqi[7][7];
        double  eq[7][7], para[6], dmv[6];
        memset(eqi, 0, sizeof(eqi));
        xo_equal_coeff(err, der[0], der[1], w, eqi, w, h, vn);
        for(int r = 0; r < np + 1; r++)
            for(int c = 0; c < np + 1; c++) eq[r][c] = (double)eqi[r][c];
        xo_affine_solve(eq, np, para);
        dmv[0] = para[0], dmv[2] = para[2], dmv[1] = para[1] * w + para[0];
        if(vn == 3) dmv[3] = para[3] * w + para[2], dmv[4] = para[4] * h + para[0], dmv[5] = para[5] * h + para[2];
        else dmv[3] = -para[3] * w + para[2];
#define XA_Q(d) xa_to_s16((d) * 4 + ((d) >= 0 ? 0.5 : -0.5))
        mvd[0][0] = XA_Q(dmv[0]), mvd[0][1] = XA_Q(dmv[2]), mvd[1][0] = XA_Q(dmv[1]), mvd[1][1] = XA_Q(dmv[3]);
        if(vn == 3) mvd[2][0] = XA_Q(dmv[4]), mvd[2][1] = XA_Q(dmv[5]);
        int zero = 1;
        for(int v = 0; v < vn; v++)
            if(mvd[v][0] || mvd[v][1]) zero = 0;
        if(zero) break;
        for(int v = 0; v < vn; v++) mvt[v][0] = (int16_t)(mvt[v][0] + mvd[v][0]), mvt[v][1] = (int16_t)(mvt[v][1] + mvd[v][1]);
        xo_affine_mc_l(ref, s_l, pic_w, pic_h, job->x, job->y, mvt, vn, w, h, bit_depth, pred);
        const int bits = xa_mv_bits(mvt, job->mvp, num_refp, ri, vn) + (bi ? job->mot_bits_other : 0);
        const uint32_t cost = XA_MV_COST(bits) + (uint32_t)(xo_satd(w, h, org, pred, s_org, w, bit_depth) >> bi);
        if(cost < cost_best) {
            cost_best = cost, best_bits = bits;
            for(int v = 0; v < vn; v++) job->mv[v][0] = mvt[v][0], job->mv[v][1] = mvt[v][1];
        }
    }
    job->cost = cost_best - XA_MV_COST(best_bits);
    free(pred), free(err), free(der[0]), free(der[1]);
}
