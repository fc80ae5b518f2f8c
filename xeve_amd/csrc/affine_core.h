// xeve_amd/csrc/affine_core.h -- Main profile, affine motion compensation of a CU (xeve_affine_mc, src_main/xevem_mc.c:2236-2339) as per-CU set-up and per-sample
// functions: what the lanes of affine.hip compute.  __host__ __device__: tests/native/affine_host.cpp compiles the same source for the host and holds it to goldens of the
// reference's own function without a GPU.
//   derive_affine_subblock_size_bi + check_eif_applicability_bi (xevem_util.c:1203-1272, 1421-1480)  -> subblock_size
//   xeve_affine_mc_lc's sub-block branch (:1826-1915)                                                 -> block_vector + mc_sample<8 | 4>
//   eif_derive_mv_clip_range + xeve_eif_mc (:1481-1530, 2123-2234)                                     -> eif_range, eif_vector, eif_bilinear, eif_out
// Two facts of the reference this form rests on (both held by the goldens): every sub-block of a CU takes the vector of the FIRST sub-block's centre (the loop adds
// half_w / half_h to a position that never moves, :1832-1833), so the sub-block branch is one translation of the whole CU and the partition cannot show; and the vector of
// a position of the enhanced filter is monotone in the position, so clipping it always = the reference's choice between a clipping and a non-clipping loop by the four
// corner vectors (can_mv_clipping_occurs, :1917-1957).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define XF __host__ __device__ inline
#else
#define XF inline
#endif

namespace xaff {

XF int iabs(int v) { return v < 0 ? -v : v; }
XF int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
XF int ilog2(int v) { int l = 0; while((1 << l) < v) l++; return l; }
XF int round_s32(int v, int rs) { return (v + (rs > 0 ? 1 << (rs - 1) : 0) - (v >= 0)) >> rs; } // xeve_rounding_s32 (xevem_util.c:1197-1201)

struct Model { // one list's motion model at 2 + 7 fractional bits: the CU's top-left vector and the change per luma sample to the right / down
    int scale[2], d_hor[2], d_ver[2];
};
XF Model model(const int16_t mv[3][2], int cuw, int cuh, int vertex_num)
{ // (:1697-1728; calculate_affine_motion_model_parameters, xevem_util.c:1331-1356)
    Model m;
    for(int c = 0; c < 2; c++) m.scale[c] = mv[0][c] << 7, m.d_hor[c] = ((mv[1][c] - mv[0][c]) << 7) >> ilog2(cuw);
    if(vertex_num == 3)
        for(int c = 0; c < 2; c++) m.d_ver[c] = ((mv[2][c] - mv[0][c]) << 7) >> ilog2(cuh);
    else m.d_ver[0] = -m.d_hor[1], m.d_ver[1] = m.d_hor[0];
    return m;
}

// sub_w x sub_h and whether the 4x4 block's footprint stays within the memory-bandwidth budget, over the lists in use
XF void subblock_size(const int8_t refi[2], const int16_t mv[2][3][2], int vertex_num, int cuw, int cuh, int &sub_w, int &sub_h, bool &mem_ok)
{
    bool eif = true, checking = true;
    sub_w = cuw, sub_h = cuh, mem_ok = true;
    for(int l = 0; l < 2; l++) {
        if(refi[l] < 0) continue;
        const Model m = model(mv[l], cuw, cuh, vertex_num);
        const int wx = iabs(m.d_hor[0]) > iabs(m.d_hor[1]) ? iabs(m.d_hor[0]) : iabs(m.d_hor[1]), wy = iabs(m.d_ver[0]) > iabs(m.d_ver[1]) ? iabs(m.d_ver[0]) : iabs(m.d_ver[1]);
        const int w = wx > 4 ? 4 : wx == 0 ? cuw : wx == 1 ? 32 : wx == 2 ? 16 : 8, h = wy > 4 ? 4 : wy == 0 ? cuh : wy == 1 ? 32 : wy == 2 ? 16 : 8;
        sub_w = w < sub_w ? w : sub_w, sub_h = h < sub_h ? h : sub_h;
        if(!checking) continue; // (check_eif_applicability_bi returns at the first list that fails: a later list's bandwidth answer is not taken, xevem_util.c:1465-1476)
        // check_eif_applicability_uni: the bounding box of a 4x4 block (+ 1 sample each way) in the reference picture ...
        int box[2];
        for(int c = 0; c < 2; c++) {
            const int c1 = 5 * (m.d_hor[c] + (c == 0 ? 512 : 0)), c2 = 5 * (m.d_ver[c] + (c == 1 ? 512 : 0)), c3 = c1 + c2;
            int mx = 0, mn = 0;
            mx = c1 > mx ? c1 : mx, mx = c2 > mx ? c2 : mx, mx = c3 > mx ? c3 : mx, mn = c1 < mn ? c1 : mn, mn = c2 < mn ? c2 : mn, mn = c3 < mn ? c3 : mn;
            box[c] = ((mx - mn + 511) >> 9) + 2;
        }
        mem_ok = mem_ok && box[0] * box[1] <= 72;
        // ... and the lines its first row fetches
        if(m.d_ver[1] < -512 || ((m.d_ver[1] > 0 ? m.d_ver[1] : 0) + iabs(m.d_hor[1])) * 5 > 512) eif = false, checking = false;
    }
    if(!eif) sub_w = sub_w < 8 ? 8 : sub_w, sub_h = sub_h < 8 ? 8 : sub_h;
}

// the sub-block branch's vector in 1/16 sample: (oh, ov) as rounded -- its fractions choose the filter variant -- and (th, tv) clipped to 128 samples around the picture
XF void block_vector(const Model &m, int sub_w, int sub_h, int x, int y, int cuw, int cuh, int pic_w, int pic_h, int &th, int &tv, int &oh, int &ov)
{
    oh = clip3(-(1 << 17), (1 << 17) - 1, round_s32(m.scale[0] + m.d_hor[0] * (sub_w >> 1) + m.d_ver[0] * (sub_h >> 1), 5));
    ov = clip3(-(1 << 17), (1 << 17) - 1, round_s32(m.scale[1] + m.d_hor[1] * (sub_w >> 1) + m.d_ver[1] * (sub_h >> 1), 5));
    th = clip3((-128 - x) << 4, (pic_w + 128 - x - cuw) << 4, oh), tv = clip3((-128 - y) << 4, (pic_h + 128 - y - cuh) << 4, ov);
}

// one sample of xeve_mc_l / xeve_mc_c's four variants (xeve_mc.c:99-381 with the Main coefficient tables): at(dy, dx) = the reference sample dy / dx from the sample the
// vector's integer part points at; cx / cy: the TAPS coefficients of the vector's fractions; fx / fy: does the UNCLIPPED vector have a fraction (the variant)
template <int TAPS, class At> XF int mc_sample(At at, bool fx, bool fy, const int16_t *cx, const int16_t *cy, int bit_depth)
{
    constexpr int back = TAPS / 2 - 1;
    const int maxv = (1 << bit_depth) - 1;
    if(!fx && !fy) return at(0, 0);
    if(!fy) {
        int acc = 0;
        for(int t = 0; t < TAPS; t++) acc += cx[t] * at(0, t - back);
        return clip3(0, maxv, acc >> 6);
    }
    if(!fx) {
        int acc = 0;
        for(int t = 0; t < TAPS; t++) acc += cy[t] * at(t - back, 0);
        return clip3(0, maxv, acc >> 6);
    }
    const int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8;
    int acc2 = 0;
    for(int r = 0; r < TAPS; r++) {
        int acc = 0;
        for(int t = 0; t < TAPS; t++) acc += cx[t] * at(r - back, t - back);
        acc2 += cy[r] * (int16_t)(acc >> shift1);
    }
    return clip3(0, maxv, (acc2 + (1 << (shift2 - 1))) >> shift2);
}

// ---- the enhanced interpolation filter --------------------------------------------------------------------------------------------------------------------------------
// the range the positions' vectors are clipped to, in 1/32 luma sample (eif_derive_mv_clip_range): the picture plus 128 samples, or -- bandwidth condition failed -- a
// window around the CU centre's vector
XF void eif_range(const Model &m, bool mem_ok, int x, int y, int cuw, int cuh, int pic_w, int pic_h, int mx[2], int mn[2])
{
    const int max_pic[2] = {(pic_w + 128 - x - cuw - 1) << 5, (pic_h + 128 - y - cuh - 1) << 5}, min_pic[2] = {(-x - 128) << 5, (-y - 128) << 5};
    for(int c = 0; c < 2; c++) {
        if(mem_ok) mx[c] = max_pic[c], mn[c] = min_pic[c];
        else {
            const int lg = ilog2(c == 0 ? cuw : cuh), spread = lg == 3 ? 128 : lg == 4 ? 256 : lg == 5 ? 544 : lg == 6 ? 1120 : 2272; // aff_mv_dev_bb2_125 (:104)
            const int mid = round_s32(m.scale[c] + m.d_hor[c] * (cuw >> 1) + m.d_ver[c] * (cuh >> 1), 4);
            mn[c] = mid - spread, mx[c] = mid + spread;
            if(mn[c] < min_pic[c]) mn[c] = min_pic[c], mx[c] = max_pic[c] < min_pic[c] + 2 * spread ? max_pic[c] : min_pic[c] + 2 * spread;
            else if(mx[c] > max_pic[c]) mx[c] = max_pic[c], mn[c] = min_pic[c] > max_pic[c] - 2 * spread ? min_pic[c] : max_pic[c] - 2 * spread;
        }
        mx[c] = clip3(-(1 << 17), (1 << 17) - 1, mx[c]), mn[c] = clip3(-(1 << 17), (1 << 17) - 1, mn[c]);
    }
}
// a component's view of model and range: chroma halves the top-left vector and the range, not the change per (chroma) sample (xeve_eif_mc, :2163-2175)
struct Eif {
    int mv0[2], dx[2], dy[2], mx[2], mn[2];
};
XF Eif eif_component(const Model &m, const int mx[2], const int mn[2], bool chroma)
{
    Eif e;
    for(int c = 0; c < 2; c++) e.mv0[c] = m.scale[c] >> (chroma ? 1 : 0), e.dx[c] = m.d_hor[c], e.dy[c] = m.d_ver[c], e.mx[c] = mx[c] >> (chroma ? 1 : 0), e.mn[c] = mn[c] >> (chroma ? 1 : 0);
    return e;
}
// the bilinear sample of position (px, py), -1 .. bw / bh: at(dy, dx) = the component's reference sample dy / dx from the CU's first sample (xeve_eif_bilinear_clip, :1991-2058)
template <class At> XF int eif_bilinear(At at, const Eif &e, int px, int py, int bit_depth)
{
    int v[2];
    for(int c = 0; c < 2; c++) v[c] = clip3(e.mn[c], e.mx[c], (e.mv0[c] + px * e.dx[c] + py * e.dy[c]) >> 4);
    const int ix = px + (v[0] >> 5), iy = py + (v[1] >> 5), fx = v[0] & 31, fy = v[1] & 31;
    const int s1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, s2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8;
    const int16_t a = (int16_t)(((64 - 2 * fx) * at(iy, ix) + 2 * fx * at(iy, ix + 1)) >> s1), b = (int16_t)(((64 - 2 * fx) * at(iy + 1, ix) + 2 * fx * at(iy + 1, ix + 1)) >> s1);
    return (int16_t)(((64 - 2 * fy) * a + 2 * fy * b + (1 << (s2 - 1))) >> s2);
}
// the output sample (px, py) from the bilinear samples: bl(r, c) = the one of position (c - 1, r - 1); {-1, 10, -1} along the rows, then down the columns, 16-bit
// intermediates (xeve_eif_filter, :1959-1989)
template <class Bl> XF int eif_out(Bl bl, int px, int py, int bit_depth)
{
    const int sh2 = bit_depth + 5 - 16 > 0 ? bit_depth + 5 - 16 : 0, sh3 = 6 - sh2, of2 = sh2 > 0 ? 1 << (sh2 - 1) : 0, of3 = 1 << (sh3 - 1);
    int hrow[3];
    for(int r = 0; r < 3; r++) hrow[r] = (int16_t)((-bl(py + r, px) + bl(py + r, px + 1) * 10 - bl(py + r, px + 2) + of2) >> sh2);
    const int16_t res = (int16_t)((-hrow[0] + hrow[1] * 10 - hrow[2] + of3) >> sh3);
    return clip3(0, (1 << bit_depth) - 1, res);
}

// ---- the affine gradient search's scalar steps (pinter_affine_me_gradient, src_main/xevem_pinter.c:4290-4501): what ONE lane of k_affine_me does between the block-wide
// passes (compensation, SATD, the normal equations' sums).  Double arithmetic exactly as the reference's build does it: no contraction (the library is compiled with
// -ffp-contract=off), IEEE division.
// one component of get_affine_mv_bits (:4257-4288): xeve_tbl_mv_bits in closed form inside (-2048, 2048] (xeve_tbl.c:286-496, its -2047 entry holds 22), the MAIN profile's
// exp-Golomb length beyond (xevem_pinter.c:217-237)
XF int me_mvd_bits(int mvd)
{
    const unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd);
    if(mvd > 2048 || mvd <= -2048) {
        unsigned nn = (a + 1) >> 1;
        int len = 0;
        for(; len < 16 && nn != 0; len++) nn >>= 1;
        return (len << 1) + 2; // (a != 0 here: the sign bit)
    }
    if(mvd == 0) return 1;
    if(mvd == -2047) return 22;
    int l = 0;
    while(((a + 1) >> (l + 1)) != 0) l++;
    return 2 * l + 2;
}
XF int me_refi_bits(int num_refp, int refi) { return num_refp <= 1 ? 0 : refi == num_refp - 1 ? refi : refi + 1; } // xeve_tbl_refi_bits (xeve_tbl.c:498-516)
XF int me_mv_bits(const int16_t mv[3][2], const int16_t mvp[3][2], int num_refp, int refi, int vertex_num)
{
    bool zero = true;
    for(int v = 0; v < vertex_num; v++) zero = zero && mv[v][0] == mvp[v][0] && mv[v][1] == mvp[v][1];
    if(zero) return 1;
    int bits = 1;
    for(int v = 0; v < vertex_num; v++) {
        int dx = mv[v][0] - mvp[v][0], dy = mv[v][1] - mvp[v][1];
        if(v) dx -= mv[0][0] - mvp[0][0], dy -= mv[0][1] - mvp[0][1];
        bits += me_mvd_bits(dx) + me_mvd_bits(dy);
    }
    return bits + me_refi_bits(num_refp, refi);
}
XF uint32_t me_mv_cost(uint32_t lambda_mv, int bits) { return (lambda_mv * (uint32_t)bits + (1u << 15)) >> 16; } // MV_COST (:53)
// (s16)(double) as the reference's x86-64 build converts: cvttsd2si to 32 bits (0x80000000 for NaN and anything outside the range), the low 16 bits kept
XF int16_t me_to_s16(double v)
{
    const int32_t t = (v >= -2147483648.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
    return (int16_t)(uint16_t)(uint32_t)t;
}
XF double me_abs(double v) { return v < 0 ? -v : v; } // (fabs of a NaN stays a NaN, of -0.0 becomes -0.0 here: neither changes a `>` comparison below)
// solve_equal (:4213-4255): Gaussian elimination with row pivoting on eq[1 .. order][0 .. order], row 0 as scratch
XF void me_solve(double (*eq)[7], int order, double *para)
{
    for(int i = 1; i < order; i++) {
        double best = me_abs(eq[i][i - 1]);
        int at = i;
        for(int j = i + 1; j < order + 1; j++)
            if(me_abs(eq[j][i - 1]) > best) best = me_abs(eq[j][i - 1]), at = j;
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
// the solution as the control points' update (:4407-4447): the change per sample at the CU's corners, rounded to quarter samples; returns whether it is all zero
XF bool me_step(const double para[6], int vertex_num, int cuw, int cuh, int16_t mvd[3][2])
{
    double d[6];
    d[0] = para[0], d[2] = para[2], d[1] = para[1] * cuw + para[0];
    if(vertex_num == 3) d[3] = para[3] * cuw + para[2], d[4] = para[4] * cuh + para[0], d[5] = para[5] * cuh + para[2];
    else d[3] = -para[3] * cuw + para[2], d[4] = d[5] = 0;
    const int order[6] = {0, 2, 1, 3, 4, 5}; // mvd[0] = (d0, d2), mvd[1] = (d1, d3), mvd[2] = (d4, d5)
    bool zero = true;
    for(int v = 0; v < 3; v++)
        for(int c = 0; c < 2; c++) {
            const double t = d[order[v * 2 + c]];
            mvd[v][c] = v < vertex_num ? me_to_s16(t * 4 + (t >= 0 ? 0.5 : -0.5)) : (int16_t)0;
            zero = zero && mvd[v][c] == 0;
        }
    return zero;
}
// the round's update of the control points from the 64-bit sums (:4404-4447): sums[r][c] = equal_coeff_t[r][c], rows 1 .. 2 * vertex_num; returns whether it is all zero
XF bool me_update(const int64_t (*sums)[7], int vertex_num, int cuw, int cuh, int16_t mvd[3][2])
{
    const int np = vertex_num << 1;
    double eq[7][7], para[6];
    for(int r = 0; r < np + 1; r++)
        for(int c = 0; c < np + 1; c++) eq[r][c] = (double)sums[r][c];
    me_solve(eq, np, para);
    return me_step(para, vertex_num, cuw, cuh, mvd);
}
// the Sobel derivatives of the prediction at (row j, column k) (xevem_scaled_horizontal / _vertical_sobel_filter, xevem_mc.c:2341-2395: border samples take the inner
// neighbour's value) and the sample's terms of the normal equations (xevem_equal_coeff_computer, :2397-2447: 32-bit products, wrapping)
template <class P> XF void me_terms(P pred /* (row, col) */, int w, int h, int j, int k, int vertex_num, int32_t c[6])
{
    const int cy = clip3(1, h - 2, j), cx = clip3(1, w - 2, k);
    const int a = pred(cy - 1, cx - 1), b = pred(cy - 1, cx), e = pred(cy - 1, cx + 1), l = pred(cy, cx - 1), r = pred(cy, cx + 1), f = pred(cy + 1, cx - 1), g = pred(cy + 1, cx),
              q = pred(cy + 1, cx + 1);
    const int32_t d0 = e - a + 2 * r - 2 * l + q - f, d1 = f - a + 2 * g - 2 * b + q - e;
    const uint32_t u0 = (uint32_t)d0, u1 = (uint32_t)d1, uj = (uint32_t)j, uk = (uint32_t)k;
    if(vertex_num == 2) c[0] = d0, c[1] = (int32_t)(uk * u0 + uj * u1), c[2] = d1, c[3] = (int32_t)(uj * u0 - uk * u1), c[4] = c[5] = 0;
    else c[0] = d0, c[1] = (int32_t)(uk * u0), c[2] = d1, c[3] = (int32_t)(uk * u1), c[4] = (int32_t)(uj * u0), c[5] = (int32_t)(uj * u1);
}

} // namespace xaff
