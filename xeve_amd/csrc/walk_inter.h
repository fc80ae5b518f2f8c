// xeve_amd/csrc/walk_inter.h -- xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047) of the node every chain of the team stands at, as team stages (walk.h).
//   candidates from the unit maps (xeve_get_avail_inter + xeve_get_motion)
//   skip / merge (xeve_analyze_skip): the uni-directional predictions of every candidate once, SSD of every (idx0, idx1) pair on the fly, the pair's bits
//   temporal direct, per list the motion search over every reference picture (pinter_me_epzs as a state machine: rounds of candidates evaluated by all lanes),
//   check_best_mvp, pinter_residue_rdo of direct + L0 + L1 side by side; analyze_bi (up to 4 rounds) + its pinter_residue_rdo
//   the cheapest mode, its reconstruction, core->s_next_best, the SATD of its luma prediction (core->inter_satd).
// pinter_residue_rdo's bit counts per candidate: lane A the all-zero CU, lane B the CU as quantised, lane C the per-component coded-block-flag tests ONE AFTER THE
// OTHER in one lane (Y without / with, U, V: each from the state the previous component's winner left -- a true dependency, so a lane of its own is all it needs),
// then, only where the tests chose another combination, that combination.
#pragma once
namespace xw {

XW int cand_slot(int m) { return m == M_DIR ? 0 : m == M_L0 ? 1 : m == M_L1 ? 2 : 3; }

// one component of get_mv_bits: xeve_tbl_mv_bits in closed form (incl. its -2047 entry) / exp-Golomb beyond +-2048 (xeve_pinter.c:74-93, xeve_tbl.c:286-496)
XW int mvd_bits(int mvd)
{
    const unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd);
    if(mvd > 2048 || mvd <= -2048) {
        unsigned nn = (a + 1) >> 12;
        int len_i = 11;
        while(len_i < 16 && nn != 0) nn >>= 1, len_i++;
        return (len_i << 1) + 2;
    }
    if(mvd == 0) return 1;
    if(mvd == -2047) return 22;
    return 2 * ilog2(a + 1) + 2;
}

// ---- motion compensation (xeve_mc.c:99-381, 401-610) ------------------------------------------------------------------------------------------------------------
// the samples under the taps of one filter position, fetched with ONE (unaligned) vector load instead of a load per tap; sum_t c[t] * r[t] over them
template <int TAPS> struct TapVec {
#if XW_DEVICE
    uint32_t w[TAPS / 2];
#else
    int v[TAPS];
#endif
};
template <int TAPS> XW TapVec<TAPS> ld_taps(const pel *r)
{
    TapVec<TAPS> t;
#if XW_DEVICE
    if(TAPS == 8) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4    u32x4_a2 __attribute__((aligned(2)));
        const u32x4 v = *reinterpret_cast<const u32x4_a2 *>(r);
        t.w[0] = v.x, t.w[1] = v.y, t.w[TAPS / 2 - 2] = v.z, t.w[TAPS / 2 - 1] = v.w;
    }
    else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef u32x2    u32x2_a2 __attribute__((aligned(2)));
        const u32x2 v = *reinterpret_cast<const u32x2_a2 *>(r);
        t.w[0] = v.x, t.w[1] = v.y;
    }
#else
    for(int i = 0; i < TAPS; i++) t.v[i] = r[i];
#endif
    return t;
}
template <int TAPS> XW int dot_taps(const int16_t *c, const TapVec<TAPS> &t)
{
    int acc = 0;
#if XW_DEVICE
#pragma unroll
    for(int i = 0; i < TAPS / 2; i++) acc += c[2 * i] * (int)(int16_t)(t.w[i] & 0xFFFF) + c[2 * i + 1] * ((int)t.w[i] >> 16);
#else
    for(int i = 0; i < TAPS; i++) acc += c[i] * t.v[i];
#endif
    return acc;
}
// one output column of one block: TAPS-tap separable interpolation at the fractions the position carries; variant from the UNCLIPPED vector's fractions (fx, fy).
// Rows are fetched four at a time (their loads in flight together).
template <int TAPS, class F> XW void mc_column(const pel *ref, int s, int gx, int gy, int fshift, const int16_t *coef, int fx, int fy, int x, int h, int bd, F out)
{
    const int fmask = (1 << fshift) - 1, back = TAPS / 2 - 1, maxv = (1 << bd) - 1, ix = gx >> fshift, iy = gy >> fshift;
    const int16_t *cx = coef + (gx & fmask) * TAPS, *cy = coef + (gy & fmask) * TAPS;
    if(!fx && !fy) {
        for(int y = 0; y < h; y++) out(y, (int)ref[(long)(iy + y) * s + ix + x]);
        return;
    }
    if(fx && !fy) {
        for(int y0 = 0; y0 < h; y0 += 4) {
            TapVec<TAPS> v[4];
#pragma unroll
            for(int t = 0; t < 4; t++)
                if(y0 + t < h) v[t] = ld_taps<TAPS>(ref + (long)(iy + y0 + t) * s + ix + x - back);
#pragma unroll
            for(int t = 0; t < 4; t++)
                if(y0 + t < h) out(y0 + t, clip3(0, maxv, dot_taps<TAPS>(cx, v[t]) >> 6));
        }
        return;
    }
    int win[TAPS];
    const int rows = h + TAPS - 1;
    if(!fx) {
        for(int r0 = 0; r0 < rows; r0 += 4) {
            int v[4];
#pragma unroll
            for(int t = 0; t < 4; t++)
                if(r0 + t < rows) v[t] = ref[(long)(iy + r0 + t - back) * s + ix + x];
#pragma unroll
            for(int t = 0; t < 4; t++) {
                const int rr = r0 + t;
                if(rr >= rows) break;
                for(int q = 0; q < TAPS - 1; q++) win[q] = win[q + 1];
                win[TAPS - 1] = v[t];
                if(rr >= TAPS - 1) {
                    int acc = 0;
                    for(int q = 0; q < TAPS; q++) acc += cy[q] * win[q];
                    out(rr - (TAPS - 1), clip3(0, maxv, acc >> 6));
                }
            }
        }
        return;
    }
    const int shift1 = bd - 8 < 4 ? bd - 8 : 4, shift2 = 20 - bd > 8 ? 20 - bd : 8, round2 = 1 << (shift2 - 1);
    for(int r0 = 0; r0 < rows; r0 += 4) {
        TapVec<TAPS> v[4];
#pragma unroll
        for(int t = 0; t < 4; t++)
            if(r0 + t < rows) v[t] = ld_taps<TAPS>(ref + (long)(iy + r0 + t - back) * s + ix + x - back);
#pragma unroll
        for(int t = 0; t < 4; t++) {
            const int rr = r0 + t;
            if(rr >= rows) break;
            const int acc = dot_taps<TAPS>(cx, v[t]);
            for(int q = 0; q < TAPS - 1; q++) win[q] = win[q + 1];
            win[TAPS - 1] = (int16_t)(acc >> shift1);
            if(rr >= TAPS - 1) {
                int a2 = 0;
                for(int q = 0; q < TAPS; q++) a2 += cy[q] * win[q];
                out(rr - (TAPS - 1), clip3(0, maxv, (a2 + round2) >> shift2));
            }
        }
    }
}
// xeve_mv_clip (xeve_mc.c:401-447): the block may leave the picture by at most MAX_CU_SIZE (128) samples
XW void mv_clip(const P &p, int x, int y, int w, int h, const int16_t mv[2], int out[2])
{
    const int x4 = x << 2, y4 = y << 2, w4 = w << 2, h4 = h << 2, min_c = -(128 << 2), max_x = (p.pic_w - 1 + 128) << 2, max_y = (p.pic_h - 1 + 128) << 2;
    out[0] = mv[0], out[1] = mv[1];
    if(x4 + mv[0] < min_c) out[0] = (int16_t)(min_c - x4);
    if(y4 + mv[1] < min_c) out[1] = (int16_t)(min_c - y4);
    if(x4 + mv[0] + w4 - 4 > max_x) out[0] = (int16_t)(max_x - x4 - w4 + 4);
    if(y4 + mv[1] + h4 - 4 > max_y) out[1] = (int16_t)(max_y - y4 - h4 + 4);
}
XW const pel *ref_plane(const P &p, int refi, int l, int c, int pic)
{
    const RefPic &r = p.refp[refi * 2 + l];
    return c == 0 ? r.y + (long)pic * p.vh * p.s_ref_l : (c == 1 ? r.u : r.v) + (long)pic * (p.vh >> p.hs) * p.s_ref_c;
}
// one column of the uni-directional prediction of component c from (list l, picture refi) with vector mv (xeve_mc.c:465-610: one list); dst dense
XW void mc_uni_column(const P &p, int x, int y, int N, int pic, int l, int refi, const int16_t mv[2], int c, int col, pel *dst)
{
    int mvt[2];
    mv_clip(p, x, y, N, N, mv, mvt);
    const int gx = ((x << 2) + mvt[0]) << 2, gy = ((y << 2) + mvt[1]) << 2, ox = mv[0] << 2, oy = mv[1] << 2;
    const pel *ref = ref_plane(p, refi, l, c, pic);
    if(c == 0) {
        mc_column<8>(ref, p.s_ref_l, gx, gy, 4, p.mc_l, ox & 0xF, oy & 0xF, col, N, p.bd, [&](int yy, int v) { dst[yy * N + col] = (pel)v; });
    }
    else {
        const int wfac = 2 / (p.ws + 1), hfac = 2 / (p.hs + 1), cw = N >> p.ws, ch = N >> p.hs;
        mc_column<4>(ref, p.s_ref_c, gx * wfac, gy * hfac, 5, p.mc_c, ox & 0x1F, oy & 0x1F, col, ch, p.bd, [&](int yy, int v) { dst[yy * cw + col] = (pel)v; });
    }
}
// does xeve_mc predict from one list only although both are valid?  identical motion: same picture, same clipped vectors (xeve_mc.c:546-551)
XW bool mc_identical(const P &p, int x, int y, int N, const int8_t refi[2], const int16_t mv[2][2])
{
    if(refi[0] < 0 || refi[1] < 0) return false;
    int a[2], b[2];
    mv_clip(p, x, y, N, N, mv[0], a), mv_clip(p, x, y, N, N, mv[1], b);
    return p.refp[refi[0] * 2].poc == p.refp[refi[1] * 2 + 1].poc && a[0] == b[0] && a[1] == b[1];
}
// the prediction of a whole CU (Y, U, V; one or two lists, averaged) into dst[3] for every chain that is `on`: xeve_mc.  Two stages (uni-directional columns of
// the first list -- or the only one --, then the second list averaged in).  job(k, on, refi, mv, dst) describes chain k.
template <class J> XW void mc_cus(const Tm &tm, const P &p, Lds &S, int nC, int log2n, int ncomp, J job)
{
    const int N = 1 << log2n;
    for(int pass = 0; pass < 2; pass++) {
        const int per = N + (ncomp > 1 ? 2 * (N >> p.ws) : 0); // columns of Y, U, V
        for(int i = tm.tid; i < nC * per; i += tm.n) {
            const int k = i / per, e = i - k * per;
            int     on = 0;
            int8_t  refi[2];
            int16_t mv[2][2];
            pel    *dst[3];
            job(k, on, refi, mv, dst);
            if(!on) continue;
            const ISt &I = S.ist[k];
            const int cwc = N >> p.ws, c = e < N ? 0 : e < N + cwc ? 1 : 2, col = c == 0 ? e : c == 1 ? e - N : e - N - cwc, h = c ? N >> p.hs : N, w = c ? cwc : N;
            const int v0 = refi[0] >= 0, v1 = refi[1] >= 0;
            if(pass == 0) {
                const int l = v0 ? 0 : 1;
                if(!v0 && !v1) continue;
                mc_uni_column(p, I.x, I.y, N, I.pic, l, refi[l], mv[l], c, col, dst[c]);
            }
            else if(v0 && v1 && !mc_identical(p, I.x, I.y, N, refi, mv)) { // the second list, averaged in (xeve_average_16b_no_clip)
                int mvt[2];
                mv_clip(p, I.x, I.y, N, N, mv[1], mvt);
                const int gx = ((I.x << 2) + mvt[0]) << 2, gy = ((I.y << 2) + mvt[1]) << 2, ox = mv[1][0] << 2, oy = mv[1][1] << 2;
                const pel *ref = ref_plane(p, refi[1], 1, c, I.pic);
                pel *d = dst[c];
                if(c == 0) mc_column<8>(ref, p.s_ref_l, gx, gy, 4, p.mc_l, ox & 0xF, oy & 0xF, col, h, p.bd, [&](int yy, int v) { d[yy * w + col] = (pel)((d[yy * w + col] + v + 1) >> 1); });
                else {
                    const int wfac = 2 / (p.ws + 1), hfac = 2 / (p.hs + 1);
                    mc_column<4>(ref, p.s_ref_c, gx * wfac, gy * hfac, 5, p.mc_c, ox & 0x1F, oy & 0x1F, col, h, p.bd, [&](int yy, int v) { d[yy * w + col] = (pel)((d[yy * w + col] + v + 1) >> 1); });
                }
            }
        }
        sync(tm);
    }
}

// ---- the motion search (pinter_me_epzs, xeve_pinter.c:699-869; me_ipel_diamond :363-551; me_raster :158-268; me_ipel_refinement :270-361; me_spel_pattern :553-697) ----
XW void epzs_range(const P &p, const MeJob &J, int cx, int cy, int range[4])
{ // get_range_ipel (xeve_pinter.c:122-140)
    const xeve_hip_me_params &m = p.me.me;
    const int sr = J.bi == 1 ? 5 : J.range_rc;
    range[0] = clip3(m.min_clip[0], m.max_clip[0], cx - sr), range[1] = clip3(m.min_clip[1], m.max_clip[1], cy - sr);
    range[2] = clip3(m.min_clip[0], m.max_clip[0], cx + sr), range[3] = clip3(m.min_clip[1], m.max_clip[1], cy + sr);
}
XW void dia_begin(const P &p, MeJob &J, int faststep, int beststep_in, int mvix, int mviy)
{
    const xeve_hip_me_params &m = p.me.me;
    J.d_cost = 0xFFFFFFFFu, J.d_bits = 0, J.d_beststep = beststep_in, J.step = 0, J.not_found = 0, J.faststep = faststep;
    J.bx = clip3(m.min_clip[0], m.max_clip[0], mvix >> 2), J.by = clip3(m.min_clip[1], m.max_clip[1], mviy >> 2), J.ix = J.bx, J.iy = J.by, J.d_run = 1;
}
// the candidates of the running diamond's next round
XW void dia_round(MeJob &J, int max_range)
{
    constexpr signed char dia16[16][2] = {{-4, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {0, 4}, {1, 3}, {2, 2}, {3, 1}, {4, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -4}, {-1, -3}, {-2, -2}, {-3, -1}};
    J.not_found++;
    if(J.step <= 2) {
        const int d = J.bi == 1 ? 5 : 2; // BI_STEP : 2
        const int x0 = J.bx <= J.range[0] ? J.bx : J.bx - d, y0 = J.by <= J.range[1] ? J.by : J.by - d;
        const int x1 = J.bx >= J.range[2] ? J.bx : J.bx + d, y1 = J.by >= J.range[3] ? J.by : J.by + d;
        J.ctype = CT_DENSE, J.c0 = x0, J.c1 = y0, J.c2 = x1 - x0 + 1, J.nc = (x1 - x0 + 1) * (y1 - y0 + 1);
    }
    else { // every ring still to come, in one round: their candidates sit around the search's FIRST centre and the range no longer moves, so nothing a ring decides
           // changes what the next one evaluates -- only whether it is evaluated at all (dia_rings applies that afterwards, in order)
        int nc = 0, nr = 0;
        for(int st = J.step; st <= max_range && nr < 8; st <<= 1) {
            const int coarse = st > 8;
            for(int i = 0; i < 16; i++) {
                if(!coarse && i > 8) continue;
                if(st == 4 && (i == 1 || i == 3 || i == 5 || i == 7)) continue;
                int dx, dy;
                if(coarse) dx = dia16[i][0], dy = dia16[i][1];
                else if(i < 8) dx = dia16[2 * i][0] / 2, dy = dia16[2 * i][1] / 2;
                else dx = dy = 0;
                J.cx[nc] = (short)(J.ix + (st >> (coarse ? 2 : 1)) * dx), J.cy[nc] = (short)(J.iy + (st >> (coarse ? 2 : 1)) * dy), nc++;
            }
            J.ring_end[nr] = (short)nc, J.ring_step[nr] = (short)st, nr++;
            if(J.bi == 1) break;
        }
        J.ring_n = (short)nr, J.ctype = CT_RINGS, J.nc = nc;
    }
}
// the rings of a CT_RINGS round, one after the other as me_ipel_diamond walks them: the first strictly cheaper candidate of a ring moves the best, a ring without
// one counts towards faststep; the diamond is over when that count is reached or the step outgrows the search range.  cost / bits: per candidate (bits < 0: outside)
XW void dia_rings(const P &p, MeJob &J, const unsigned *cost, const short *bits)
{
    int from = 0;
    for(int r = 0; r < J.ring_n; r++) {
        if(r > 0) J.not_found++; // (the first ring's count was taken when the round was set up)
        const int st = J.ring_step[r];
        for(int k = from; k < J.ring_end[r]; k++) {
            if(bits[k] < 0) continue;
            if(cost[k] < J.d_cost) J.bx = J.cx[k], J.by = J.cy[k], J.d_beststep = st, J.not_found = 0, J.d_cost = cost[k], J.d_bits = bits[k];
        }
        from = J.ring_end[r];
        if(J.not_found == J.faststep || J.bi == 1) return;
    }
}
XW void cand_xy(const MeJob &J, int i, int &mx, int &my)
{
    if(J.ctype == CT_DENSE) my = J.c1 + i / J.c2, mx = J.c0 + i - (i / J.c2) * J.c2;
    else if(J.ctype == CT_GRID) {
        const int g = J.r_pos + i;
        my = J.range[1] + (g / J.r_nx) * J.r_stp, mx = J.range[0] + (g % J.r_nx) * J.r_stp;
    }
    else mx = J.cx[i], my = J.cy[i]; // CT_LIST, CT_SPEL, CT_RINGS
}
// after a round of the diamond: the best candidate (lowest index among the cheapest) against the running best; true when the search is over
XW bool dia_finish(const P &p, MeJob &J, unsigned cost, int idx, int bits)
{
    const xeve_hip_me_params &m = p.me.me;
    if(idx >= 0 && cost < J.d_cost) {
        int mx, my;
        cand_xy(J, idx, mx, my);
        J.bx = mx, J.by = my, J.d_beststep = J.step <= 2 ? 2 : J.step, J.not_found = 0, J.d_cost = cost, J.d_bits = bits;
    }
    if(J.step <= 2) {
        const int sr = J.bi == 1 ? 5 : J.range_rc; // get_range_ipel: re-centred on the best so far (xeve_pinter.c:463-468)
        J.range[0] = clip3(m.min_clip[0], m.max_clip[0], J.bx - sr), J.range[2] = clip3(m.min_clip[0], m.max_clip[0], J.bx + sr);
        J.range[1] = clip3(m.min_clip[1], m.max_clip[1], J.by - sr), J.range[3] = clip3(m.min_clip[1], m.max_clip[1], J.by + sr);
        J.step += 2;
    }
    if(J.not_found == J.faststep) return true;
    if(J.bi == 1) return true;
    J.step <<= 1;
    return J.step > m.max_search_range;
}
// sub-pel patterns (xeve_pinter.c:50-70)
XW void spel_round(const P &p, MeJob &J, int stage)
{
    constexpr signed char hp[8][2] = {{-2, 0}, {-2, 2}, {0, 2}, {2, 2}, {2, 0}, {2, -2}, {0, -2}, {-2, -2}}, qp[8][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, 1}, {1, 1}, {-1, -1}, {1, -1}};
    const int cnt = stage ? p.me.qpel_cnt : p.me.hpel_cnt, cx = J.s_mv[0] + (J.x << 2), cy = J.s_mv[1] + (J.y << 2);
    for(int i = 0; i < cnt; i++) J.cx[i] = (short)(cx + (stage ? qp[i][0] : hp[i][0])), J.cy[i] = (short)(cy + (stage ? qp[i][1] : hp[i][1]));
    J.ctype = CT_SPEL, J.nc = cnt;
}
// the nine points of a raster refinement round around (r_cx, r_cy) (xeve_pinter.c:233-262)
XW void raster_ref_round(MeJob &J)
{
    J.r_cx = J.r_mv[0], J.r_cy = J.r_mv[1];
    int nc = 0;
    for(int i = -J.r_ss; i <= J.r_ss; i += J.r_ss)
        for(int j = -J.r_ss; j <= J.r_ss; j += J.r_ss) J.cx[nc] = (short)((J.r_cx >> 2) + J.x + j), J.cy[nc] = (short)((J.r_cy >> 2) + J.y + i), nc++;
    J.ctype = CT_LIST, J.nc = nc;
}
// the sub-pel stage or, at ME_LEV_IPEL, the integer refinement (xeve_pinter.c:808-866): sets up its first round
XW void me_last_stage(const P &p, MeJob &J)
{
    const xeve_hip_me_params &m = p.me.me;
    if(p.me.hpel_cnt == 0) {
        constexpr signed char pos[9][2] = {{0, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
        epzs_range(p, J, J.x + (J.mv[0] >> 2), J.y + (J.mv[1] >> 2), J.range);
        const int ix = clip3(m.min_clip[0], m.max_clip[0], (int16_t)(J.mv[0] + (J.x << 2)) >> 2), iy = clip3(m.min_clip[1], m.max_clip[1], (int16_t)(J.mv[1] + (J.y << 2)) >> 2);
        for(int i = 0; i < 9; i++) J.cx[i] = (short)(ix + pos[i][0]), J.cy[i] = (short)(iy + pos[i][1]);
        J.phase = PH_IREF, J.ctype = CT_LIST, J.nc = 9, J.s_mv[0] = (ix - J.x) << 2, J.s_mv[1] = (iy - J.y) << 2, J.s_cost = 0xFFFFFFFFu, J.s_bits = 0;
        return;
    }
    J.phase = PH_HPEL, J.s_mv[0] = J.mv[0], J.s_mv[1] = J.mv[1], J.s_cost = 0xFFFFFFFFu, J.s_bits = 0;
    spel_round(p, J, 0);
}
// after the first diamond (and the raster search): the refinement loop while beststep > 0 (xeve_pinter.c:780-806), then the last stage
XW void me_refine_or_last(const P &p, MeJob &J)
{
    if(J.bi != 1 && J.beststep > 0) { // get_range_ipel on the UNCLIPPED centre
        epzs_range(p, J, J.x + (J.mv[0] >> 2), J.y + (J.mv[1] >> 2), J.range);
        J.beststep = 0, J.phase = PH_DREF;
        dia_begin(p, J, 2, J.tmpstep, (int16_t)(J.mv[0] + (J.x << 2)), (int16_t)(J.mv[1] + (J.y << 2)));
        dia_round(J, p.me.me.max_search_range);
        return;
    }
    me_last_stage(p, J);
}
XW void me_raster_done(const P &p, MeJob &J)
{ // (:757-767)
    if(J.r_bits > 0) J.mot_bits = J.r_bits;
    if(J.r_cost < J.cost_best) J.beststep = 5, J.cost_best = J.r_cost, J.mv[0] = (int16_t)J.r_mv[0], J.mv[1] = (int16_t)J.r_mv[1];
    me_refine_or_last(p, J);
}
// One step of the search's control: takes the result of the round just evaluated (its cheapest candidate: cost, index, bits; index < 0: none was inside the range)
// and sets up the next round, or ends the search (phase PH_DONE, nc 0).  first: nothing has been evaluated yet (the first diamond's first round is set up).
XW void me_advance(const P &p, MeJob &J, bool first, unsigned cost, int idx, int bits, const unsigned *costs = nullptr, const short *bitv = nullptr)
{
    const xeve_hip_me_params &m = p.me.me;
    if(first) {
        dia_round(J, p.me.me.max_search_range);
        return;
    }
    switch(J.phase) {
    case PH_D1:
    case PH_DREF: {
        if(J.ctype == CT_RINGS) dia_rings(p, J, costs, bitv); // (every remaining ring was in the round: the diamond is over)
        else if(!dia_finish(p, J, cost, idx, bits)) {
            dia_round(J, p.me.me.max_search_range);
            return;
        }
        // the diamond is over: its result against the running best (xeve_pinter.c:742-756, 790-806)
        const int rmvx = (J.bx - J.x) << 2, rmvy = (J.by - J.y) << 2;
        J.tmpstep = J.d_beststep;
        if(J.bi != 1 && J.d_bits > 0) J.mot_bits = J.d_bits;
        if(J.d_cost < J.cost_best) {
            J.cost_best = J.d_cost, J.mv[0] = (int16_t)rmvx, J.mv[1] = (int16_t)rmvy;
            J.beststep = (iabs(J.mvp[0] - J.mv[0]) < 2 && iabs(J.mvp[1] - J.mv[1]) < 2) ? 0 : J.tmpstep;
        }
        if(J.phase == PH_D1 && J.bi == 0 && J.beststep > 5 && (m.reserved & 1)) { // me_raster (:757-767): over the range the diamond left, from the vector it left
            J.phase = PH_RASTER, J.r_mv[0] = (int16_t)rmvx, J.r_mv[1] = (int16_t)rmvy, J.r_cost = 0xFFFFFFFFu, J.r_bits = 0, J.r_pos = 0;
            J.r_nx = (J.range[2] - J.range[0]) / J.r_stp + 1;
            J.r_total = J.range[2] >= J.range[0] && J.range[3] >= J.range[1] ? J.r_nx * ((J.range[3] - J.range[1]) / J.r_stp + 1) : 0;
            J.r_ss = J.r_stp >> 1;
            if(J.r_total > 0) J.ctype = CT_GRID, J.nc = imin(XW_MEC, J.r_total);
            else if(J.r_ss > 0) J.phase = PH_RASTER_REF, raster_ref_round(J);
            else me_raster_done(p, J);
            return;
        }
        if(J.phase == PH_D1) me_refine_or_last(p, J);
        else if(J.bi != 1 && J.beststep > 0) me_refine_or_last(p, J);
        else me_last_stage(p, J);
        return;
    }
    case PH_RASTER: {
        if(idx >= 0 && cost < J.r_cost) {
            int mx, my;
            cand_xy(J, idx, mx, my);
            J.r_mv[0] = (mx - J.x) << 2, J.r_mv[1] = (my - J.y) << 2, J.r_cost = cost, J.r_bits = bits;
        }
        J.r_pos += J.nc;
        if(J.r_pos < J.r_total) {
            J.ctype = CT_GRID, J.nc = imin(XW_MEC, J.r_total - J.r_pos);
            return;
        }
        if(J.r_ss > 0) J.phase = PH_RASTER_REF, raster_ref_round(J);
        else me_raster_done(p, J);
        return;
    }
    case PH_RASTER_REF: {
        if(idx >= 0 && cost < J.r_cost) J.r_mv[0] = (J.cx[idx] - J.x) << 2, J.r_mv[1] = (J.cy[idx] - J.y) << 2, J.r_cost = cost, J.r_bits = bits;
        J.r_ss >>= 1;
        if(J.r_ss > 0) raster_ref_round(J);
        else me_raster_done(p, J);
        return;
    }
    case PH_IREF: { // me_ipel_refinement's result (:835-866)
        if(idx >= 0 && cost < J.s_cost) J.s_mv[0] = (J.cx[idx] - J.x) << 2, J.s_mv[1] = (J.cy[idx] - J.y) << 2, J.s_cost = cost, J.s_bits = bits;
        if(J.bi != 1 && J.s_bits > 0) J.mot_bits = J.s_bits;
        if(J.s_cost < J.cost_best) J.cost_best = J.s_cost, J.mv[0] = (int16_t)J.s_mv[0], J.mv[1] = (int16_t)J.s_mv[1];
        J.phase = PH_DONE, J.nc = 0;
        return;
    }
    case PH_HPEL: {
        if(idx >= 0 && cost < J.s_cost) J.s_mv[0] = J.cx[idx] - (J.x << 2), J.s_mv[1] = J.cy[idx] - (J.y << 2), J.s_cost = cost; // (only the quarter-pel loop records the bits, :683)
        if(p.me.qpel_cnt > 0) {
            J.phase = PH_QPEL;
            spel_round(p, J, 1);
            return;
        }
        idx = -1;
    } // fall through: no quarter-pel stage
    case PH_QPEL: {
        if(idx >= 0 && cost < J.s_cost) J.s_mv[0] = J.cx[idx] - (J.x << 2), J.s_mv[1] = J.cy[idx] - (J.y << 2), J.s_cost = cost, J.s_bits = bits;
        if(!J.bi && J.s_bits > 0) J.mot_bits = J.s_bits; // (:690-692)
        if(J.s_cost < J.cost_best) J.cost_best = J.s_cost, J.mv[0] = (int16_t)J.s_mv[0], J.mv[1] = (int16_t)J.s_mv[1];
        J.phase = PH_DONE, J.nc = 0;
        return;
    }
    default: J.nc = 0; return;
    }
}
// SAD of 8 samples of a row: packed 16-bit absolute differences (v_sad_u16 is unsigned: a bi search's org_bi, which may be negative, is biased together with the
// reference)
XW int sad8(const pel *o, const pel *r, int bi)
{
#if XW_DEVICE
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4    u32x4_a2 __attribute__((aligned(2)));
    u32x4 a = *reinterpret_cast<const u32x4_a2 *>(o), b = *reinterpret_cast<const u32x4_a2 *>(r);
    if(bi) a ^= 0x80008000u, b ^= 0x80008000u;
    unsigned acc = 0;
    acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc), acc = __builtin_amdgcn_sad_u16(a.y, b.y, acc);
    acc = __builtin_amdgcn_sad_u16(a.z, b.z, acc), acc = __builtin_amdgcn_sad_u16(a.w, b.w, acc);
    return (int)acc;
#else
    (void)bi;
    int acc = 0;
    for(int t = 0; t < 8; t++) acc += iabs((int)o[t] - (int)r[t]);
    return acc;
#endif
}
// one candidate's validity and bits (get_mv_bits, xeve_pinter.c:74-120): < 0 = outside the search range
XW int me_cand_bits(const MeJob &J, int i)
{
    int mx, my;
    cand_xy(J, i, mx, my);
    if(J.ctype == CT_SPEL) return mvd_bits(mx - J.gmvp[0]) + mvd_bits(my - J.gmvp[1]) + J.refi_bits + (J.bi ? J.extra_bits : 0);
    if(mx > J.range[2] || mx < J.range[0] || my > J.range[3] || my < J.range[1]) return -1;
    return mvd_bits((mx << 2) - J.gmvp[0]) + mvd_bits((my << 2) - J.gmvp[1]) + J.refi_bits + (J.bi ? J.extra_bits : 0);
}
// part `part` of `parts` of one candidate's SAD: rows [part * N / parts ..) of an integer position, columns [part * N / parts ..) of a sub-pel position (each column
// interpolated on the fly: xeve_mc_l picks the variant from the low 4 bits of (mv << 2) and positions with the same value)
XW int me_cand_sad_part(const P &p, const MeJob &J, int i, int N, int part, int parts)
{
    int mx, my;
    cand_xy(J, i, mx, my);
    int sad = 0;
    const int groups = N >= 8 ? N / 8 : 1; // (rows of an integer position go in groups of 8; a 4x4 CU is one group)
    if(J.ctype != CT_SPEL && parts > groups) { // (a round that mixes sub-pel and integer searches is cut for the sub-pel ones: whole groups of rows here)
        const int g = parts / groups;
        if(part % g) return 0;
        part /= g, parts = groups;
    }
    const int n = N / parts, from = part * n;
    if(J.ctype == CT_SPEL) {
        const int gx = mx << 2, gy = my << 2;
        for(int col = from; col < from + n; col++)
            mc_column<8>(J.ref, p.s_ref_l, gx, gy, 4, p.mc_l, gx & 15, gy & 15, col, N, p.bd, [&](int yy, int v) { sad += iabs((int)J.org[(long)yy * J.so + col] - v); });
        return sad;
    }
    const pel *r = J.ref + (long)my * p.s_ref_l + mx;
    if(N < 8) { // a 4x4 CU (preset placebo's min_cu_inter): rows of four samples
        for(int yy = from; yy < from + n; yy++)
            for(int xx = 0; xx < N; xx++) sad += iabs((int)J.org[(long)yy * J.so + xx] - (int)r[(long)yy * p.s_ref_l + xx]);
        return sad;
    }
#if XW_DEVICE
    { // four rows at a time, their loads issued together (v_sad_u16 is unsigned: a bi search's org_bi is biased together with the reference)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4    u32x4_a2 __attribute__((aligned(2)));
        unsigned acc = 0;
        for(int yy = from; yy < from + n; yy += 4)
            for(int xx = 0; xx < N; xx += 8) {
                u32x4 a[4], b[4];
#pragma unroll
                for(int t = 0; t < 4; t++)
                    a[t] = *reinterpret_cast<const u32x4_a2 *>(J.org + (long)(yy + t) * J.so + xx), b[t] = *reinterpret_cast<const u32x4_a2 *>(r + (long)(yy + t) * p.s_ref_l + xx);
#pragma unroll
                for(int t = 0; t < 4; t++) {
                    if(J.bi) a[t] ^= 0x80008000u, b[t] ^= 0x80008000u;
                    acc = __builtin_amdgcn_sad_u16(a[t].x, b[t].x, acc), acc = __builtin_amdgcn_sad_u16(a[t].y, b[t].y, acc);
                    acc = __builtin_amdgcn_sad_u16(a[t].z, b[t].z, acc), acc = __builtin_amdgcn_sad_u16(a[t].w, b[t].w, acc);
                }
            }
        return (int)acc;
    }
#else
    for(int yy = from; yy < from + n; yy++) {
        const pel *o = J.org + (long)yy * J.so, *q = r + (long)yy * p.s_ref_l;
        for(int xx = 0; xx < N; xx += 8) sad += sad8(o + xx, q + xx, J.bi);
    }
    return sad;
#endif
}
// MV_COST + the SAD term (xeve_pinter.c:47, 200-209, 323-341, 443-451, 600-627)
XW unsigned me_cand_total(const P &p, const MeJob &J, int bits, int sad)
{
    const xeve_hip_me_params &m = p.me.me;
    unsigned cost;
    if(J.ctype == CT_SPEL || J.phase == PH_D1 || J.phase == PH_DREF) cost = (uint32_t)(m.lambda_mv * (uint32_t)bits + (1u << 15)) >> 16; // u32 arithmetic as MV_COST
    else cost = (uint32_t)(((uint64_t)m.lambda_mv * (uint32_t)bits + (1u << 15)) >> 16);
    sad >>= (p.bd - 8);
    return cost + (uint32_t)(J.bi ? sad >> 1 : sad);
}
// all searches of S.mej[0 .. nj) to their end: rounds of {every candidate's bits, the candidates' SADs spread over all lanes in parts, the search's control on its own thread}
XW_ST void me_run(const Tm &tm, const P &p, Lds &S, int nj, int log2n)
{
    const int N = 1 << log2n;
    int *msad = (int *)S.mcost;
    for(int j = tm.tid; j < nj; j += tm.n)
        if(S.mej[j].on) me_advance(p, S.mej[j], true, 0, -1, 0);
        else S.mej[j].nc = 0, S.mej[j].phase = PH_DONE;
    sync(tm);
    for(;;) {
        if(tm.tid == 0) S.flag[0] = 0;
        int total = 0;
        for(int j = 0; j < nj; j++) total += S.mej[j].nc;
        for(int it = tm.tid; it < total; it += tm.n) { // (the items are the candidates that exist, search after search)
            int c = it, j = 0;
            while(c >= S.mej[j].nc) c -= S.mej[j].nc, j++;
            S.mbits[j * XW_MEC + c] = (short)me_cand_bits(S.mej[j], c), msad[j * XW_MEC + c] = 0;
        }
        sync(tm), mark(tm, p, S, PR_M_BITS);
        // the SADs: every candidate in `parts` parts (whole rows of an integer position, columns of a sub-pel position) so that a round fills the lanes
        int parts = 1, spel = 0;
        for(int j = 0; j < nj; j++) spel |= S.mej[j].nc > 0 && S.mej[j].ctype == CT_SPEL;
        while(parts < (spel ? N : N / 8) && total * parts < 4 * tm.n) parts <<= 1; // (rows of an integer position in groups of >= 8; a sub-pel position down to single columns)
        if(tm.n == 1) parts = 1;
        for(int it = tm.tid; it < total * parts; it += tm.n) { // (the items are the candidates that exist, search after search)
            const int part = it % parts;
            int c = it / parts, j = 0;
            while(c >= S.mej[j].nc) c -= S.mej[j].nc, j++;
            const int e = j * XW_MEC + c;
            const MeJob &J = S.mej[j];
            if(S.mbits[e] < 0) continue;
            aadd(&msad[e], me_cand_sad_part(p, J, c, N, part, parts));
#if XW_DEVICE
            if(p.sad_units) atomicAdd(p.sad_units + (blockIdx.x & 255), (u64)(N * N / (J.ctype != CT_SPEL && parts > (N >= 8 ? N / 8 : 1) ? (N >= 8 ? N / 8 : 1) : parts))); // (roofline: algorithmic work)
#endif
        }
        sync(tm);
        for(int it = tm.tid; it < total; it += tm.n) { // MV_COST + the SAD term of every candidate
            int c = it, j = 0;
            while(c >= S.mej[j].nc) c -= S.mej[j].nc, j++;
            const int e = j * XW_MEC + c;
            if(S.mbits[e] >= 0) S.mcost[e] = me_cand_total(p, S.mej[j], S.mbits[e], msad[e]);
        }
        sync(tm), mark(tm, p, S, PR_M_SAD);
        for(int j = tm.tid; j < nj; j += tm.n) {
            MeJob &J = S.mej[j];
            if(J.nc == 0) continue;
            unsigned best = 0xFFFFFFFFu;
            int idx = -1, bits = 0;
            for(int c = 0; c < J.nc; c++) { // the first strictly cheaper candidate in evaluation order wins
                const int b = S.mbits[j * XW_MEC + c];
                if(b < 0) continue;
                const unsigned v = S.mcost[j * XW_MEC + c];
                if(idx < 0 || v < best) best = v, idx = c, bits = b;
            }
            me_advance(p, J, false, best, idx, bits, S.mcost + j * XW_MEC, S.mbits + j * XW_MEC);
            if(J.nc) aor(&S.flag[0], 1); // (several jobs may say so at once)
        }
        sync(tm), mark(tm, p, S, PR_M_SEL);
        if(!S.flag[0]) break;
        sync(tm);
    }
}
// a search of chain k, list l, reference picture r, from the chain's state
XW void me_job_init(const P &p, MeJob &J, const ISt &I, int k, int l, int r, int bi, int N, const pel *org, int so, const int16_t mvp[2], const int16_t start[2],
                    const int16_t mv0[2], int refi_bits, int extra_bits)
{
    const xeve_hip_me_params &m = p.me.me;
    J.on = 1, J.k = k, J.l = l, J.r = r, J.bi = bi, J.x = I.x, J.y = I.y, J.so = so, J.org = org;
    J.ref = ref_plane(p, r, l, 0, I.pic);
    J.mvp[0] = mvp[0], J.mvp[1] = mvp[1], J.gmvp[0] = (int16_t)(mvp[0] + (I.x << 2)), J.gmvp[1] = (int16_t)(mvp[1] + (I.y << 2));
    J.refi_bits = refi_bits, J.extra_bits = extra_bits, J.range_rc = p.range_recentre[l][r];
    J.phase = PH_D1, J.tmpstep = 0, J.beststep = 0, J.mot_bits = 0, J.cost_best = 0xFFFFFFFFu, J.mv[0] = mv0[0], J.mv[1] = mv0[1];
    const int lmin = ilog2((unsigned)N);
    const int st = (1 << (lmin - 1)) > 5 ? (1 << (lmin - 1)) : 5; // max(RASTER_SEARCH_STEP, half the CU)
    J.r_stp = st * (r + 1);
    epzs_range(p, J, clip3(m.min_clip[0], m.max_clip[0], I.x + (start[0] >> 2)), clip3(m.min_clip[1], m.max_clip[1], I.y + (start[1] >> 2)), J.range);
    dia_begin(p, J, 3, 0, (int16_t)(start[0] + (I.x << 2)), (int16_t)(start[1] + (I.y << 2)));
}

} // namespace xw
#include "walk_inter2.h"
