// xeve_amd/csrc/walk_dbk.h -- rdo_dbk_switch = 1 (preset slow): calc_delta_dist_filter_boundary (src_base/xeve_mode.c:1534-2005) as a team stage of the fused walk.
// A candidate's distortion includes what the loop filter will do to the CU's TOP edge, then its LEFT edge (xeve_deblock_unit with is_hor_edge 1, then 0, :1852-1864):
//   delta[c] = SSD after - SSD before over the block, the 2 rows above it and the 2 columns to its left (chroma: 1), against the original.
// The reference lays the block and its borders into a scratch picture, filters, and takes two SSDs.  Only the samples within 2 (chroma: 1) of the two edges change, so
// here ONE LANE PER EDGE POSITION (a column of the top edge, a row of the left edge) filters its four samples across the edge and adds (after - before) of exactly
// those to the block's accumulator: no scratch picture, no barrier inside the stage.  The left filter runs after the top filter, so the lanes of the first two rows of
// the left edge redo the top filter of the first two columns to get the samples it left (and own their contribution; the top lanes leave those to them).
// The current side of an edge takes the CANDIDATE's flags (intra, luma cbf) and motion, the far side what the unit maps hold -- flags and reference indices of decided
// CUs, the luma cbf flag only where the WRITER has been (xeve_eco_unit; the mode decision never sets it), and NO motion: the filter compares ctx->map_unrefined_mv,
// which during the mode decision is what update_map_scu copies out of cu_data->unrefined_mv -- an array the Baseline encoder never writes (oracle/xeve_oracle.c
// xo_delta_dist holds the same against six reference bitstreams).  Chroma strengths: the scratch picture's qp offsets are zero, so the chroma QP is the CU's
// (P::qp[c] without the bit-depth offset) -- preset slow is refused with chroma qp offsets (enc_plan.h).
#pragma once
namespace xw {

XW int df_st(int cls, int qp)
{ // xeve_tbl_df_st (xeve_tbl.c:239-257): intra; luma cbf; other motion / reference; none
    constexpr uint8_t t[3][52] = {
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 12, 12, 12, 12},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 11, 11, 11, 11, 11},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10, 10}};
    return cls > 2 || qp < 0 || qp > 51 ? 0 : t[cls][qp];
}
// get_tbl_qp_to_st (xeve_df.c:34-87) with a motionless far side: m0 / r0 / v0 the candidate's, m1 / r1 the neighbour unit's
XW int df_class(uint32_t m0, uint32_t m1, const int8_t r0[2], const int8_t r1[2], const int16_t v0[4])
{
    if(((m0 | m1) >> 15) & 1) return 0;
    if(((m0 | m1) >> 24) & 1) return 1;
    if(((m0 | m1) >> 26) & 1) return 2;
    const int a0x = r0[0] < 0 ? 0 : v0[0], a0y = r0[0] < 0 ? 0 : v0[1], a1x = r0[1] < 0 ? 0 : v0[2], a1y = r0[1] < 0 ? 0 : v0[3];
    if(r0[0] == r1[0] && r0[1] == r1[1]) return (iabs(a0x) >= 4 || iabs(a0y) >= 4 || iabs(a1x) >= 4 || iabs(a1y) >= 4) ? 2 : 3;
    if(r0[0] == r1[1] && r0[1] == r1[0]) return (iabs(a0x) >= 4 || iabs(a0y) >= 4 || iabs(a1x) >= 4 || iabs(a1y) >= 4) ? 2 : 3;
    return 2;
}
// deblock_scu_hor / _ver and their chroma forms on ONE sample quadruple A B | C D across the edge (xeve_df.c:89-251; s16 arithmetic as the reference)
XW void df_quad(int &A, int &B, int &C, int &D, int st, int maxv, int chroma)
{
    if(!st) return;
    const int16_t d = (int16_t)((A - (B << 2) + (C << 2) - D) / 8);
    const int16_t ab = (int16_t)((d ^ (d >> 15)) - (d >> 15)), sign = d < 0;
    const int16_t t16 = (int16_t)(((ab - st) << 1) > 0 ? ((ab - st) << 1) : 0);
    int16_t       clip = (int16_t)((ab - t16) > 0 ? (ab - t16) : 0);
    const int16_t d1 = (int16_t)(sign ? -clip : clip);
    if(!chroma) {
        clip >>= 1;
        const int16_t d2 = (int16_t)clip3(-clip, clip, (A - D) / 4);
        A = clip3(0, maxv, (int16_t)(A - d2)), D = clip3(0, maxv, (int16_t)(D + d2));
    }
    B = clip3(0, maxv, (int16_t)(B + d1)), C = clip3(0, maxv, (int16_t)(C - d1));
}
XW long long ssd1(int v, int o, int sh) { return (long long)(((v - o) * (v - o)) >> sh); }

struct DbkJob { // one candidate CU
    int        on, pic, x, y, intra, cbf, two; // two: the candidate is the rounded mean of a[] and b[] (a merge pair's prediction is never laid down)
    int8_t     refi[2];
    int16_t    mv[4];
    const pel *a[3], *b[3];                    // dense [h][w] per component (a null component is skipped)
    u64       *acc[3];                         // where delta[c] is added (two's complement)
};
XW int dbk_src(const DbkJob &J, int c, int q) { return J.two ? (J.a[c][q] + J.b[c][q] + 1) >> 1 : J.a[c][q]; }

// the top-edge filter of column q of component c: the samples at rows -2, -1, 0, 1 before (u) and after (f)
XW void dbk_top(const P &p, const DbkJob &J, int c, int w, int q, const pel *rec, int sm, const uint32_t *ms, const int8_t (*mr)[2], int t, int u[4], int f[4])
{
    const int sx = c ? p.ws : 0, maxv = (1 << p.bd) - 1, bdo = p.bd - 8, nb = t + ((q << sx) >> 2) - p.w_scu;
    const uint32_t m_cur = ((uint32_t)(J.intra != 0) << 15) | ((uint32_t)(J.cbf != 0) << 24);
    const int8_t none[2] = {-1, -1};
    const int cls = df_class(m_cur, ms[nb], J.refi, mr ? mr[nb] : none, J.mv);
    const int st = df_st(cls, c ? p.qp[c] - 6 * bdo : p.slice_qp) << bdo;
    u[0] = rec[-2 * sm + q], u[1] = rec[-sm + q], u[2] = dbk_src(J, c, q), u[3] = dbk_src(J, c, w + q);
    f[0] = u[0], f[1] = u[1], f[2] = u[2], f[3] = u[3];
    df_quad(f[0], f[1], f[2], f[3], st, maxv, c != 0);
}

// jobs 0 .. njobs - 1 (get(j, J) describes job j; all of CU size 1 << log2n): every delta added to its accumulator.  No barrier inside; the caller syncs before
// (accumulators ready) and after.
template <class Get> XW void dbk_stage(const Tm &tm, const P &p, int njobs, int log2n, Get get)
{
    const int N = 1 << log2n, ncomp = p.idc ? 3 : 1, Nc = p.idc ? N >> p.ws : 0, Hc = p.idc ? N >> p.hs : 0, per = 2 * N + (ncomp > 1 ? 2 * (Nc + Hc) : 0);
    const int sh = (p.bd - 8) * 2, maxv = (1 << p.bd) - 1, bdo = p.bd - 8;
    for(int it = tm.tid; it < njobs * per; it += tm.n) {
        const int j = it / per;
        int r = it - j * per, c = 0;
        if(r >= 2 * N) r -= 2 * N, c = 1 + r / (Nc + Hc), r -= (c - 1) * (Nc + Hc);
        const int w = c ? Nc : N, left_lane = r >= w, pos = left_lane ? r - w : r;
        DbkJob J;
        get(j, J);
        if(!J.on || !J.a[c]) continue;
        const int sx = c ? p.ws : 0, sy = c ? p.hs : 0, chroma = c != 0, nt = chroma ? 1 : 2; // nt: samples each side the filter may change = the template's width
        const long mo = (long)J.pic * p.map_pic;
        const uint32_t *ms = p.map_scu + mo;
        const uint8_t  *mt = p.map_tidx + mo;
        const int8_t (*mr)[2] = p.map_refi ? p.map_refi + mo : nullptr;
        const int t = (J.x >> 2) + (J.y >> 2) * p.w_scu;
        const bool top = J.y > 0 && mt[t] == mt[t - p.w_scu], left = J.x > 0 && (ms[t - 1] >> 31) && mt[t] == mt[t - 1];
        const int sm = c ? p.s_mod_c : p.s_mod_l, so = c ? p.s_org_c : p.s_org_l, xc = J.x >> sx, yc = J.y >> sy;
        const pel *rec = (c ? p.mod[c] + (long)J.pic * p.mod_pic_c : p.mod[0] + (long)J.pic * p.mod_pic_l) + (long)yc * sm + xc;
        const pel *org = (c ? p.org[c] + (long)J.pic * p.org_pic_c : p.org[0] + (long)J.pic * p.org_pic_l) + (long)yc * so + xc;
        const int8_t none[2] = {-1, -1};
        long long d = 0;
        if(!left_lane) { // a column of the top edge
            if(!top) continue;
            int u[4], f[4];
            dbk_top(p, J, c, w, pos, rec, sm, ms, mr, t, u, f);
            const bool corner = left && pos < 2; // (rows 0, 1 of the first two columns go on into the left filter: its lanes account for them)
            for(int k = 0; k < 4; k++) {
                const int row = k - 2; // -2, -1 | 0, 1
                if(row < -nt || row >= nt) continue; // outside the template / untouched by the chroma filter
                if(corner && row >= 0) continue;
                const int o = org[(long)row * so + pos];
                d += ssd1(f[k], o, sh) - ssd1(u[k], o, sh);
            }
        }
        else { // a row of the left edge
            if(!left) continue;
            const int nb = t + ((pos << sy) >> 2) * p.w_scu - 1;
            const uint32_t m_cur = ((uint32_t)(J.intra != 0) << 15) | ((uint32_t)(J.cbf != 0) << 24);
            const int cls = df_class(m_cur, ms[nb], J.refi, mr ? mr[nb] : none, J.mv);
            const int st = df_st(cls, c ? p.qp[c] - 6 * bdo : p.slice_qp) << bdo;
            int base[4] = {rec[(long)pos * sm - 2], rec[(long)pos * sm - 1], dbk_src(J, c, pos * w), dbk_src(J, c, pos * w + 1)}; // unfiltered: what "before" holds
            int v[4] = {base[0], base[1], base[2], base[3]};
            if(top && pos < 2) { // the top filter has been over this row's first two samples
                for(int q = 0; q < 2; q++) {
                    int u[4], f[4];
                    dbk_top(p, J, c, w, q, rec, sm, ms, mr, t, u, f);
                    v[2 + q] = f[2 + pos];
                }
            }
            df_quad(v[0], v[1], v[2], v[3], st, maxv, chroma);
            for(int k = 0; k < 4; k++) {
                const int col = k - 2;
                if(col < -nt) continue; // (luma: columns -2, -1; chroma: -1)
                if(col >= nt && !(top && pos < 2)) continue; // (an inner sample the left filter leaves alone -- unless the top filter changed it and it is ours to account for)
                if(col >= nt && pos >= nt) continue;
                const int o = org[(long)pos * so + col];
                d += ssd1(v[k], o, sh) - ssd1(base[k], o, sh);
            }
        }
        if(d) aadd64(J.acc[c], (u64)d);
    }
}

} // namespace xw
