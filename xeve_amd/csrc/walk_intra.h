// xeve_amd/csrc/walk_intra.h -- pintra_analyze_cu (src_base/xeve_pintra.c:544-698) of the node every chain of the team stands at, as team stages (walk.h).
//   neighbours + mode ranks (xeve_get_nbr, xeve_get_mpm) -> the five Baseline predictors -> SATD per mode (tile per lane) and the mode's index bits (coder lane per
//   (chain, mode)) -> make_ipred_list per chain -> the luma RDO of the list (blocks_chain over chains x slots; the luma syntax counted, coder lane per (chain, slot))
//   -> chroma with the winner's mode -> the CU's cost from the whole intra syntax + core->s_temp_best.
// The bit count of the chroma RDO never reaches an output (cost_t is discarded, :643-646) and is not computed.
#pragma once
namespace xw {

enum { SH_ON = 0, SH_X, SH_Y, SH_PIC, SH_MPM, SH_CNT, SH_LIST, SH_BEST = SH_LIST + 5, SH_IPD, SH_DC, SH_ISATD = SH_DC + 3, SH_END };
#define XW_ACC 40 // accumulators per chain: [0..4] SATD per mode, [8..12] bits

#define XW_COD_IF(m) (((m) >> 15) & 1u)
#define XW_COD_COD(m) (((m) >> 31) & 1u)


// the head of an intra CU (xeve_rdo_bit_cnt_cu_intra*, xeve_mode.c:81-175): skip flag and pred_mode outside I slices, the mode as its rank among the most probable
template <bool FULL> XW void cod_intra_head(Cod &c, const P &p, int rank)
{
    if(p.slice_type != 2) {
        cod_bin<FULL>(c, XEVE_HIP_CTX_SKIP_FLAG, 0); // ctx_skip / ctx_pred_mode: 0 without sps_cm_init_flag (xeve_get_ctx_some_flags, xeve_util.c:1181-1288)
        cod_bin<FULL>(c, XEVE_HIP_CTX_PRED_MODE, 1);
    }
    cod_unary2<FULL>(c, (unsigned)rank, XEVE_HIP_CTX_INTRA_DIR);
}

template <bool FULL> XW void intra_node(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L)
{
    const int log2n = L + 2, N = 1 << log2n, n0 = N * N, lc = log2n - p.ws, Nc = p.idc ? N >> p.ws : 0, n1 = Nc * Nc, ncomp = p.idc ? 3 : 1;
    // A: which chains analyse, where
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        const Node &nd = W.node[L];
        const int on = nd.leaf && (!p.inter || nd.try_intra);
        int *sh = S.sh[k];
        sh[SH_ON] = on, sh[SH_X] = nd.x0, sh[SH_Y] = nd.y0, sh[SH_PIC] = p.jobs[c0 + k].pic;
        sh[SH_ISATD] = (p.inter && nd.try_intra) ? (int)W.eres.satd : (int)0xFFFFFFFFu;
        for(int i = 0; i < XW_ACC; i++) S.acc[k * XW_ACC + i] = 0;
        W.ires.on = on;
    }
    sync(tm), mark(tm, p, S, PR_I_SETUP);
    // B: neighbours of every component (xeve_get_nbr, xeve_ipred.c:32-105), the rank row (xeve_get_mpm, :229-252)
    for(int c = 0; c < ncomp; c++) {
        const int cw = c ? Nc : N, nline = 2 * cw, per = 2 * nline + 1;
        int unit = c ? 2 : 4;
        if(c && p.idc == 3) unit *= 2;
        for(int i = tm.tid; i < nC * per; i += tm.n) {
            const int k = i / per, t = i - k * per;
            const int *sh = S.sh[k];
            if(!sh[SH_ON]) continue;
            Cw &W = p.cw[c0 + k];
            const long pic = sh[SH_PIC];
            const uint32_t *ms = p.map_scu + pic * p.map_pic;
            const uint8_t  *mt = p.map_tidx + pic * p.map_pic;
            const int x = sh[SH_X], y = sh[SH_Y], x_scu = x >> 2, y_scu = y >> 2, scup = y_scu * p.w_scu + x_scu;
            const int s = c ? p.s_mod_c : p.s_mod_l;
            const pel *src = (c == 0 ? p.mod[0] + pic * p.mod_pic_l : p.mod[c] + pic * p.mod_pic_c) + (c ? (long)(y >> p.hs) * s + (x >> p.ws) : (long)y * s + x);
            const pel grey = (pel)(1 << (p.bd - 1));
            pel *left = W.nb[c][0] + 1, *up = W.nb[c][1] + 1;
#define XW_USABLE(u) (XW_COD_COD(ms[u]) && (!p.cip || XW_COD_IF(ms[u])) && mt[scup] == mt[u])
            if(t == 2 * nline) { // the corner sample: avail_cu & AVAIL_UP_LE (xeve_util.c:753-755), then the constrained-intra test
                const bool ok = x_scu > 0 && y_scu > 0 && XW_USABLE(scup - p.w_scu - 1);
                const pel v = ok ? src[-s - 1] : grey;
                up[-1] = v, left[-1] = v;
            }
            else if(t < nline) {
                const int u = t / unit;
                const bool ok = y_scu > 0 && x_scu + u < p.w_scu && XW_USABLE(scup - p.w_scu + u);
                up[t] = ok ? src[-s + t] : grey;
            }
            else {
                const int q = t - nline, u = q / unit;
                const bool ok = x_scu > 0 && y_scu + u < p.h_scu && XW_USABLE(scup - 1 + u * p.w_scu);
                left[q] = ok ? src[(long)q * s - 1] : grey;
            }
            if(c == 0 && t == 0) { // xeve_get_mpm
                const int8_t *mi = p.map_ipm + pic * p.map_pic;
                int l = 0, u = 0;
                if(x_scu > 0 && XW_COD_IF(ms[scup - 1]) && XW_COD_COD(ms[scup - 1]) && mt[scup] == mt[scup - 1]) l = mi[scup - 1] + 1;
                if(y_scu > 0 && XW_COD_IF(ms[scup - p.w_scu]) && XW_COD_COD(ms[scup - p.w_scu]) && mt[scup] == mt[scup - p.w_scu]) u = mi[scup - p.w_scu] + 1;
                S.sh[k][SH_MPM] = l * 6 + u;
            }
#undef XW_USABLE
        }
    }
    sync(tm), mark(tm, p, S, PR_I_NBR);
    // C: the DC value of every component: (sum + w) >> (log2 w + 1) (xeve_ipred.c:133-150)
    for(int i = tm.tid; i < nC * ncomp; i += tm.n) {
        const int k = i / ncomp, c = i - k * ncomp, cw = c ? Nc : N, lw = c ? lc : log2n;
        if(!S.sh[k][SH_ON]) continue;
        const Cw &W = p.cw[c0 + k];
        const pel *left = W.nb[c][0] + 1, *up = W.nb[c][1] + 1;
        int dc = 0;
        for(int t = 0; t < cw; t++) dc += left[t] + up[t];
        S.sh[k][SH_DC + c] = (dc + cw) >> (lw + 1);
    }
    sync(tm);
    // D: the five luma predictors (xeve_ipred, :107-202)
    for(int i = tm.tid; i < nC * 5 * n0; i += tm.n) {
        const int px = i & (n0 - 1), km = i >> (2 * log2n), k = km / 5, m = km - k * 5;
        if(!S.sh[k][SH_ON]) continue;
        Cw &W = p.cw[c0 + k];
        const pel *left = W.nb[0][0] + 1, *up = W.nb[0][1] + 1;
        const int r = px >> log2n, q = px & (N - 1);
        int v;
        if(m == 0) v = S.sh[k][SH_DC];
        else if(m == 1) v = left[r];
        else if(m == 2) v = up[q];
        else if(m == 3) v = r > q ? left[r - q - 1] : (r == q ? up[-1] : up[q - r - 1]);
        else v = (up[r + q + 1] + left[r + q + 1]) >> 1;
        W.ipred[m][px] = (pel)v;
    }
    sync(tm), mark(tm, p, S, PR_I_PRED);
    // E: SATD of each predictor against the original (tile per lane); the bits of each mode index from the entry state (xeve_rdo_bit_cnt_intra_dir); the rate tables
    {
        const int ts = N == 4 ? 4 : 8, tl = N / ts, tiles = tl * tl;
        for(int i = tm.tid; i < nC * 5 * tiles; i += tm.n) {
            const int t = i % tiles, km = i / tiles, k = km / 5, m = km - k * 5;
            const int *sh = S.sh[k];
            if(!sh[SH_ON]) continue;
            const Cw &W = p.cw[c0 + k];
            const int ty = (t / tl) * ts, tx = (t % tl) * ts;
            const pel *o = p.org[0] + (long)sh[SH_PIC] * p.org_pic_l + (long)(sh[SH_Y] + ty) * p.s_org_l + sh[SH_X] + tx;
            aadd(&S.acc[k * XW_ACC + m], had_tile(o, p.s_org_l, W.ipred[m] + ty * N + tx, N, ts));
        }
        for(int i = tm.tid; i < nC * 28; i += tm.n) {
            const int k = i / 28, e = i - k * 28;
            if(S.sh[k][SH_ON]) est_entry(p, p.cw[c0 + k].curr[L], e, S.est[k]);
        }
    }
    sync(tm), mark(tm, p, S, PR_I_SATD);
    coder_stage<FULL>(
        tm, S, nC * 5,
        [&](int j, const Sbac *&in, Sbac *&out) {
            const int k = j / 5;
            in = &p.cw[c0 + k].curr[L], out = nullptr;
            return S.sh[k][SH_ON] != 0;
        },
        [&](int j, Cod &c) {
            const int k = j / 5, m = j - k * 5;
            cod_unary2<FULL>(c, (unsigned)mpm_rank(S.sh[k][SH_MPM], m), XEVE_HIP_CTX_INTRA_DIR);
            S.acc[k * XW_ACC + 8 + m] = (int)cod_bits<FULL>(c);
        });
    mark(tm, p, S, PR_I_BITS);
    // F: make_ipred_list (xeve_pintra.c:308-374) per chain; the luma blocks of the list
    for(int k = tm.tid; k < nC; k += tm.n) {
        if(!S.sh[k][SH_ON]) continue;
        int *sh = S.sh[k];
        int      lst[5];
        double   cc[5];
        unsigned cs[5];
        for(int i = 0; i < 5; i++) lst[i] = 0, cc[i] = XW_MAX_COST, cs[i] = 0xFFFFFFFFu;
        for(int m = 0; m < 5; m++) {
            const unsigned sa = (unsigned)(S.acc[k * XW_ACC + m] >> (p.bd - 8));
            const double cost = (double)sa + (double)S.acc[k * XW_ACC + 8 + m] * p.sqrt_lambda0;
            int shift = 0;
            while(shift < 5 && cost < cc[4 - shift]) shift++;
            if(shift) {
                for(int q = 1; q < shift; q++) lst[5 - q] = lst[4 - q], cc[5 - q] = cc[4 - q], cs[5 - q] = cs[4 - q];
                lst[5 - shift] = m, cc[5 - shift] = cost, cs[5 - shift] = sa;
            }
        }
        int cnt = 5;
        for(int i = 4; i >= 1; i--) {
            if((double)cs[i] > (double)(uint32_t)sh[SH_ISATD] * (1.2)) cnt--;
            else break;
        }
        sh[SH_CNT] = cnt;
        for(int i = 0; i < 5; i++) sh[SH_LIST + i] = lst[i];
    }
    sync(tm);
    for(int i = tm.tid; i < nC * 5; i += tm.n) {
        const int k = i / 5, sl = i - k * 5;
        const int *sh = S.sh[k];
        Cw  &W = p.cw[c0 + k];
        Blk &B = S.blk[i];
        B.on = sh[SH_ON] && sl < sh[SH_CNT];
        B.org = p.org[0] + (long)sh[SH_PIC] * p.org_pic_l + (long)sh[SH_Y] * p.s_org_l + sh[SH_X], B.s_org = p.s_org_l;
        B.pred = W.ipred[sh[SH_ON] ? sh[SH_LIST + sl] : 0], blk_slot(B, &W.slot[sl]), B.comp = 0, B.nnz = 0, B.nev = 0, B.k = k, B.is_intra = 1, B.ssd[0] = B.ssd[1] = 0;
    }
    sync(tm), mark(tm, p, S, PR_I_LIST);
    // G: the luma RDO of the list (:604-637; pintra_residue_rdo mode 0)
    blocks_chain(tm, p, S, S.blk, nC * 5, log2n, 0);
    if(p.rdo_dbk) { // rdo_dbk_switch (xeve_pintra.c:131-149): the loop filter's share of every candidate's luma distortion (an intra CU: the strongest filter class)
        dbk_stage(tm, p, nC * 5, log2n, [&](int j, DbkJob &J) {
            Blk &B = S.blk[j];
            const int *sh = S.sh[j / 5];
            J.on = B.on, J.pic = sh[SH_PIC], J.x = sh[SH_X], J.y = sh[SH_Y], J.intra = 1, J.cbf = B.nnz != 0, J.two = 0, J.refi[0] = J.refi[1] = -1;
            J.mv[0] = J.mv[1] = J.mv[2] = J.mv[3] = 0;
            J.a[0] = B.rec, J.a[1] = J.a[2] = nullptr, J.b[0] = J.b[1] = J.b[2] = nullptr, J.acc[0] = &B.ssd[1], J.acc[1] = J.acc[2] = nullptr;
        });
        sync(tm);
    }
    coder_stage<FULL>(
        tm, S, nC * 5,
        [&](int j, const Sbac *&in, Sbac *&out) {
            in = &p.cw[c0 + j / 5].curr[L], out = nullptr;
            return S.blk[j].on != 0;
        },
        [&](int j, Cod &c) {
            const Blk &B = S.blk[j];
            const int k = j / 5, sl = j - k * 5;
            cod_intra_head<FULL>(c, p, mpm_rank(S.sh[k][SH_MPM], S.sh[k][SH_LIST + sl]));
            CoefSet q;
            q.ev[0] = B.ev, q.nev[0] = B.nev, q.nnz[0] = B.nnz, q.ev[1] = q.ev[2] = nullptr, q.nev[1] = q.nev[2] = q.nnz[1] = q.nnz[2] = 0;
            cod_coef<FULL>(c, p.idc, q, 1, 1); // xeve_rdo_bit_cnt_cu_intra_luma (xeve_mode.c:81-117)
            S.acc[k * XW_ACC + 8 + sl] = (int)cod_bits<FULL>(c);
        });
    mark(tm, p, S, PR_I_BITS);
    // H: the luma decision (first strictly smallest cost); the chroma blocks of the winner's mode
    for(int k = tm.tid; k < nC; k += tm.n) {
        if(!S.sh[k][SH_ON]) continue;
        int *sh = S.sh[k];
        Cw  &W = p.cw[c0 + k];
        double best = XW_MAX_COST;
        int    bs = 0;
        for(int sl = 0; sl < sh[SH_CNT]; sl++) {
            double cost = 0;
            cost += (double)(int64_t)S.blk[k * 5 + sl].ssd[1];
            cost += (double)S.acc[k * XW_ACC + 8 + sl] * p.lambda[0];
            if(cost < best) best = cost, bs = sl;
        }
        sh[SH_BEST] = bs, sh[SH_IPD] = sh[SH_LIST + bs];
        W.ires.slot = bs, W.ires.ipm = sh[SH_IPD], W.ires.pred_cnt = sh[SH_CNT];
        W.ires.nnz[0] = S.blk[k * 5 + bs].nnz, W.ires.dist_cu = (int32_t)(double)(int64_t)S.blk[k * 5 + bs].ssd[1]; // (dist_y for now)
    }
    sync(tm), mark(tm, p, S, PR_I_PICK);
    Blk *cb = S.blk + XW_MAXC * 5; // chroma blocks: (chain, U / V)
    if(ncomp > 1) {
        for(int i = tm.tid; i < nC * 2; i += tm.n) {
            const int k = i >> 1, c = 1 + (i & 1);
            const int *sh = S.sh[k];
            Cw  &W = p.cw[c0 + k];
            Blk &B = cb[i];
            B.on = sh[SH_ON];
            B.org = p.org[c] + (long)sh[SH_PIC] * p.org_pic_c + (long)(sh[SH_Y] >> p.hs) * p.s_org_c + (sh[SH_X] >> p.ws), B.s_org = p.s_org_c;
            B.pred = W.cpred[c - 1], blk_slot(B, &W.slot[4 + c]), B.comp = c, B.nnz = 0, B.nev = 0, B.k = k, B.is_intra = 1, B.ssd[0] = B.ssd[1] = 0;
        }
        for(int i = tm.tid; i < nC * 2 * n1; i += tm.n) {
            const int px = i % n1, kc = i / n1, k = kc >> 1, c = 1 + (kc & 1);
            const int *sh = S.sh[k];
            if(!sh[SH_ON]) continue;
            Cw &W = p.cw[c0 + k];
            const pel *left = W.nb[c][0] + 1, *up = W.nb[c][1] + 1;
            const int r = px >> lc, q = px & (Nc - 1), m = sh[SH_IPD];
            int v;
            if(m == 0) v = sh[SH_DC + c];
            else if(m == 1) v = left[r];
            else if(m == 2) v = up[q];
            else if(m == 3) v = r > q ? left[r - q - 1] : (r == q ? up[-1] : up[q - r - 1]);
            else v = (up[r + q + 1] + left[r + q + 1]) >> 1;
            W.cpred[c - 1][px] = (pel)v;
        }
        sync(tm), mark(tm, p, S, PR_I_CPRED);
        blocks_chain(tm, p, S, cb, nC * 2, lc, 0); // pintra_residue_rdo mode 1 (:150-269)
        if(p.rdo_dbk) { // (:244-263) the chroma blocks' share; kept apart from their SSD: it enters the cost as delta x weight (ssd[0] of a chroma block is unused here)
            for(int i = tm.tid; i < nC * 2; i += tm.n) cb[i].ssd[0] = 0;
            sync(tm);
            dbk_stage(tm, p, nC, log2n, [&](int k, DbkJob &J) {
                const int *sh = S.sh[k];
                J.on = sh[SH_ON], J.pic = sh[SH_PIC], J.x = sh[SH_X], J.y = sh[SH_Y], J.intra = 1, J.cbf = 0, J.two = 0, J.refi[0] = J.refi[1] = -1;
                J.mv[0] = J.mv[1] = J.mv[2] = J.mv[3] = 0;
                J.a[0] = nullptr, J.a[1] = cb[k * 2].rec, J.a[2] = cb[k * 2 + 1].rec, J.b[0] = J.b[1] = J.b[2] = nullptr;
                J.acc[0] = nullptr, J.acc[1] = &cb[k * 2].ssd[0], J.acc[2] = &cb[k * 2 + 1].ssd[0];
            });
            sync(tm);
        }
    }
    // I: the CU's cost (:679-695): the whole syntax from the entry state; its exit state is core->s_temp_best
    coder_stage<FULL>(
        tm, S, nC,
        [&](int k, const Sbac *&in, Sbac *&out) {
            in = &p.cw[c0 + k].curr[L], out = &p.cw[c0 + k].sbest;
            return S.sh[k][SH_ON] != 0;
        },
        [&](int k, Cod &c) {
            Cw &W = p.cw[c0 + k];
            const Blk &Y = S.blk[k * 5 + S.sh[k][SH_BEST]];
            cod_intra_head<FULL>(c, p, mpm_rank(S.sh[k][SH_MPM], S.sh[k][SH_IPD]));
            CoefSet q;
            q.ev[0] = Y.ev, q.nev[0] = Y.nev, q.nnz[0] = Y.nnz;
            for(int cc = 1; cc < 3; cc++) {
                const Blk &B = cb[k * 2 + cc - 1];
                q.ev[cc] = ncomp > 1 ? B.ev : nullptr, q.nev[cc] = ncomp > 1 ? B.nev : 0, q.nnz[cc] = ncomp > 1 ? B.nnz : 0;
            }
            cod_coef<FULL>(c, p.idc, q, 7, 1);
            int dist_c = 0;
            if(ncomp > 1) { // (xeve_pintra.c:216-233, :266: the weighted sum as a double, then (s32))
                double d = 0;
                d += p.wgt[0] * (double)(int64_t)cb[k * 2].ssd[1];
                d += p.wgt[1] * (double)(int64_t)cb[k * 2 + 1].ssd[1];
                if(p.rdo_dbk) d += ((double)(int64_t)cb[k * 2].ssd[0] * p.wgt[0]) + ((double)(int64_t)cb[k * 2 + 1].ssd[0] * p.wgt[1]);
                dist_c = (int)d;
            }
            const int dist_y = W.ires.dist_cu;
            double cost = (double)(int)cod_bits<FULL>(c) * p.lambda[0];
            cost += dist_y;
            if(ncomp > 1) cost += dist_c;
            W.ires.cost = cost, W.ires.dist_cu = dist_y + (ncomp > 1 ? dist_c : 0);
            W.ires.nnz[1] = ncomp > 1 ? cb[k * 2].nnz : 0, W.ires.nnz[2] = ncomp > 1 ? cb[k * 2 + 1].nnz : 0;
        });
    mark(tm, p, S, PR_I_FINAL);
}

} // namespace xw
