// xeve_amd/csrc/walk_tree.h -- the tree operations of mode_coding_tree (src_base/xeve_mode.c:2007-2375) for the chains of a team, and the team's walk through the
// static schedule.  Per tree node (size 2^(L+2)):
//     ENTER(L)        the node's entry: coder state from the parent / the previous sibling, split_cu_flag = 0 priced, clear_map_scu (:1129)
//     [P / B: the inter analysis, MID(L)]  [the intra analysis]
//     LEAF(L)         copy_to_cu_data (:868) of the winning mode, mode_cpy_rec_to_ref (:797), the early terminations (:2162-2187), split_cu_flag = 1 priced
//     4 x { the quadrant's subtree; CHILD_DONE(L): its cost added, copy_cu_data (:430) into the parent, update_map_scu (:1036) }
//     EXIT(L)         the cheaper alternative kept (a split must win by more than 0.0001), picture + split mode + coder state of the winner
// Scalar decisions: thread k for chain k.  Bulk copies: all threads, chain after chain.
#pragma once
namespace xw {

XW void cud_init(const Tm &tm, CtuData *d, int log2)
{ // init_cu_data (:374-428): what the walk reads back -- split modes and luma / chroma modes cleared
    const int n = 1 << (2 * (log2 - 2));
    for(int u = tm.tid; u < n; u += tm.n) {
        for(int k = 0; k < XEVE_HIP_CU_DEPTHS; k++) d->split_mode[k][u] = 0;
        d->ipm[0][u] = 0, d->ipm[1][u] = 0;
    }
}
// copy_cu_data (:430-620): the sub-block (x, y; log2) of dst (pitch 1 << log2_cus) <- all of src, split modes from depth cud on
XW void cud_copy(const Tm &tm, const P &p, CtuData *dst, const CtuData *src, int x, int y, int log2, int log2_cus, int cud)
{
    const int n = 1 << (log2 - 2), cus = 1 << (log2_cus - 2), cw = 1 << log2, cs = 1 << log2_cus;
    for(int u = tm.tid; u < n * n; u += tm.n) {
        const int j = u / n, i = u - j * n, di = ((y >> 2) + j) * cus + (x >> 2) + i, si = u;
        for(int k = cud; k < XEVE_HIP_CU_DEPTHS; k++) dst->split_mode[k][di] = src->split_mode[k][si];
        dst->pred_mode[di] = src->pred_mode[si], dst->ipm[0][di] = src->ipm[0][si], dst->ipm[1][di] = src->ipm[1][si], dst->depth[di] = src->depth[si];
        dst->map_scu[di] = src->map_scu[si], dst->map_cu_mode[di] = src->map_cu_mode[si];
        for(int c = 0; c < 3; c++) dst->nnz[c][di] = src->nnz[c][si];
        for(int k = 0; k < 4; k++) (&dst->mv[di][0][0])[k] = (&src->mv[si][0][0])[k], (&dst->mvd[di][0][0])[k] = (&src->mvd[si][0][0])[k];
        for(int k = 0; k < 2; k++) dst->refi[di][k] = src->refi[si][k], dst->mvp_idx[di][k] = src->mvp_idx[si][k];
    }
    for(int t = tm.tid; t < cw * cw; t += tm.n) {
        const int j = t >> log2, i = t & (cw - 1), d = (y + j) * cs + x + i;
        dst->coef[0][d] = src->coef[0][t], dst->reco[0][d] = src->reco[0][t];
    }
    if(p.idc) {
        const int wc = cw >> p.ws, hc = cw >> p.hs, sc = cs >> p.ws;
        for(int t = tm.tid; t < wc * hc; t += tm.n) {
            const int j = t / wc, i = t - j * wc, d = ((y >> p.hs) + j) * sc + (x >> p.ws) + i;
            dst->coef[1][d] = src->coef[1][t], dst->reco[1][d] = src->reco[1][t];
            dst->coef[2][d] = src->coef[2][t], dst->reco[2][d] = src->reco[2][t];
        }
    }
}
XW void clear_map(const Tm &tm, const P &p, int pic, int x, int y, int cu)
{ // clear_map_scu (:1129-1155)
    const int w = (x + cu > p.pic_w ? p.pic_w - x : cu) >> 2, h = (y + cu > p.pic_h ? p.pic_h - y : cu) >> 2;
    uint32_t *ms = p.map_scu + (long)pic * p.map_pic, *mc = p.map_cu_mode + (long)pic * p.map_pic;
    for(int t = tm.tid; t < w * h; t += tm.n) {
        const int j = t / w, i = t - j * w, g = ((y >> 2) + j) * p.w_scu + (x >> 2) + i;
        ms[g] = 0, mc[g] = 0;
    }
}
XW void update_map(const Tm &tm, const P &p, int pic, const CtuData *d, int x, int y, int cu)
{ // update_map_scu (:1036-1127) + update_to_ctx_map (:2445-2516): the maps the analyses of later CUs read
    const int w = (x + cu > p.pic_w ? p.pic_w - x : cu) >> 2, h = (y + cu > p.pic_h ? p.pic_h - y : cu) >> 2, n = cu >> 2;
    uint32_t *ms = p.map_scu + (long)pic * p.map_pic, *mc = p.map_cu_mode + (long)pic * p.map_pic;
    int8_t   *mi = p.map_ipm + (long)pic * p.map_pic;
    for(int t = tm.tid; t < w * h; t += tm.n) {
        const int j = t / w, i = t - j * w, g = ((y >> 2) + j) * p.w_scu + (x >> 2) + i, u = j * n + i;
        ms[g] = d->map_scu[u], mc[g] = d->map_cu_mode[u], mi[g] = d->ipm[0][u];
        if(p.inter) {
            const long gm = (long)pic * p.map_pic + g;
            for(int k = 0; k < 4; k++) (&p.map_mv[gm][0][0])[k] = (&d->mv[u][0][0])[k];
            p.map_refi[gm][0] = d->refi[u][0], p.map_refi[gm][1] = d->refi[u][1];
        }
    }
}
XW void rec_to_pic(const Tm &tm, const P &p, int pic, const CtuData *d, int x, int y, int cu)
{ // mode_cpy_rec_to_ref (:797-866)
    const int w = x + cu > p.pic_w ? p.pic_w - x : cu, h = y + cu > p.pic_h ? p.pic_h - y : cu;
    pel *m = p.mod[0] + (long)pic * p.mod_pic_l;
    for(int t = tm.tid; t < w * h; t += tm.n) {
        const int j = t / w, i = t - j * w;
        m[(long)(y + j) * p.s_mod_l + x + i] = d->reco[0][j * cu + i];
    }
    if(p.idc) {
        const int wc = w >> p.ws, hc = h >> p.hs, sc = cu >> p.ws;
        pel *mu = p.mod[1] + (long)pic * p.mod_pic_c, *mv = p.mod[2] + (long)pic * p.mod_pic_c;
        for(int t = tm.tid; t < wc * hc; t += tm.n) {
            const int  j = t / wc, i = t - j * wc;
            const long g = (long)((y >> p.hs) + j) * p.s_mod_c + (x >> p.ws) + i;
            mu[g] = d->reco[1][j * sc + i], mv[g] = d->reco[2][j * sc + i];
        }
    }
}
// copy_to_cu_data (:868-1034) of the node's CU into its cu_data_temp: the unit fields, then the dense blocks
XW void unit_to_temp(const Tm &tm, const P &p, CtuData *t, int log2, int cud, int n, int cu_mode, int ipm, const int32_t *nnz, const InterRes *R, const int16_t *cy,
                     const int16_t *cu_, const int16_t *cv, const pel *ry, const pel *ru, const pel *rv, int n0, int n1)
{
    const uint32_t scu = ((uint32_t)p.slice_num & 0x7F) | ((uint32_t)p.slice_qp << 16) | (1u << 31) | (cu_mode == 0 ? 1u << 15 : 0) | (cu_mode == 2 ? 1u << 23 : 0); // _SN, _QP, _COD, _IF, _SF
    const uint32_t cum = ((uint32_t)log2 << 24) | ((uint32_t)log2 << 28);                                                                                        // MCU_SET_LOGW / LOGH
    for(int u = tm.tid; u < n; u += tm.n) {
        t->pred_mode[u] = (uint8_t)cu_mode, t->depth[u] = (int8_t)cud;
        if(cu_mode == 0) t->ipm[0][u] = (int8_t)ipm, t->ipm[1][u] = (int8_t)(p.idc ? ipm : 0);
        t->nnz[0][u] = nnz[0], t->nnz[1][u] = p.idc ? nnz[1] : 0, t->nnz[2][u] = p.idc ? nnz[2] : 0;
        t->map_scu[u] = scu, t->map_cu_mode[u] = cum;
        for(int k = 0; k < 4; k++) (&t->mv[u][0][0])[k] = R ? (&R->mv[0][0])[k] : 0, (&t->mvd[u][0][0])[k] = R ? (&R->mvd[0][0])[k] : 0;
        for(int k = 0; k < 2; k++) t->refi[u][k] = R ? R->refi[k] : -1, t->mvp_idx[u][k] = R ? R->mvp_idx[k] : 0;
    }
    for(int i = tm.tid; i < n0; i += tm.n) t->coef[0][i] = cy ? cy[i] : (int16_t)0, t->reco[0][i] = ry[i];
    if(p.idc)
        for(int i = tm.tid; i < n1; i += tm.n) t->coef[1][i] = cu_ ? cu_[i] : (int16_t)0, t->reco[1][i] = ru[i], t->coef[2][i] = cv ? cv[i] : (int16_t)0, t->reco[2][i] = rv[i];
}

enum { T_ACT = 0, T_LEAF, T_BND, T_X, T_Y, T_F0, T_F1 };

// body(k, sub) for every chain of the team: the lanes split evenly among the chains, each share copying its own chain's data side by side (chain after chain on a
// team smaller than its chains: the host build).  The bodies hold no barrier.
template <class F> XW void for_chains(const Tm &tm, int nC, F body)
{
    if(tm.n >= 2 * nC) {
        const int g = tm.n / nC, k = tm.tid / g;
        Tm sub;
        sub.tid = tm.tid - k * g, sub.n = g;
        if(k < nC) body(k, sub);
    }
    else
        for(int k = 0; k < nC; k++) body(k, tm);
}

template <bool FULL> XW void op_enter(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L, int part)
{
    const int log2 = L + 2, cu = 1 << log2;
    for(int k = tm.tid; k < nC; k += tm.n) {
        const int c = c0 + k;
        Cw &W = p.cw[c];
        const xeve_hip_ctu_job J = p.jobs[c];
        Node *nd = &W.node[L];
        int active, x0, y0;
        if(part < 0) {
            active = 1, x0 = J.x, y0 = J.y;
            W.curr[L] = p.states[J.sbac];
        }
        else {
            const Node *pn = &W.node[L + 1];
            x0 = pn->x0 + (part & 1) * cu, y0 = pn->y0 + (part >> 1) * cu;
            active = pn->active && pn->do_split && x0 < p.pic_w && y0 < p.pic_h;
            if(active) W.curr[L] = part == 0 ? W.curr[L + 1] : W.next[L]; // the state the previous quadrant's winner left (:2248-2262)
        }
        int leaf = 0, boundary = 0;
        if(active) {
            boundary = !(x0 + cu <= p.pic_w && y0 + cu <= p.pic_h);
            leaf = !boundary && cu <= p.max_cu;
            W.before[L] = W.curr[L];
            memset(&W.tdepth[L], 0, sizeof(Sbac));
            nd->cost_best = XW_MAX_COST, nd->best_split = 0, nd->do_split = 0, nd->dist_cu = 0;
            double cost_temp = 0.0;
            if(!boundary) {
                if(leaf) {
                    if(cu > p.min_cuwh) { // split_cu_flag = 0 (:2079-2091)
                        Sbac run;
                        cost_temp += (double)(int)split_flag_bits<FULL>(W.curr[L], run, 0) * p.lambda[0];
                        W.curr[L] = run;
                    }
                }
                else cost_temp = XW_MAX_COST;
            }
            nd->cost_temp = cost_temp;
        }
        nd->active = active, nd->x0 = x0, nd->y0 = y0, nd->leaf = leaf;
        if(p.inter) nd->try_intra = 0, nd->cu_mode = 0, nd->unit_cost = XW_MAX_COST;
        int *sh = S.sh[k];
        sh[T_ACT] = active, sh[T_LEAF] = leaf, sh[T_BND] = boundary, sh[T_X] = x0, sh[T_Y] = y0;
    }
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        const int *sh = S.sh[k];
        if(sh[T_ACT] && !sh[T_BND]) cud_init(sub, &p.cw[c0 + k].temp[L], log2);
        if(sh[T_LEAF]) clear_map(sub, p, p.jobs[c0 + k].pic, sh[T_X], sh[T_Y], cu);
    });
    sync(tm);
}

// P / B slices, after the inter analysis: mode_check_inter's store (:1199-1218) and what mode_check_intra needs (:1245-1262)
XW void op_mid(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L)
{
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (p.log2_ctu - log2), n = 1 << (2 * L), n0 = cu * cu, n1 = p.idc ? n0 >> (p.ws + p.hs) : 0;
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        Cw &W = p.cw[c0 + k];
        Node *nd = &W.node[L];
        if(!nd->leaf) return;
        const InterRes &R = W.eres;
        const Slot *sy = R.slot >= 0 ? &W.slot[R.slot] : nullptr;
        // (a winner without levels of a component leaves that component's coefficients zero: pi->coef[best] is cleared where cbf drops out, xeve_pinter.c:1264-1274)
        unit_to_temp(sub, p, &W.temp[L], log2, cud, n, R.cu_mode, 0, R.nnz, &R, sy && R.nnz[0] ? sy[0].lev : nullptr, sy && R.nnz[1] ? sy[1].lev : nullptr,
                     sy && R.nnz[2] ? sy[2].lev : nullptr, W.wrec[0], W.wrec[1], W.wrec[2], n0, n1);
    });
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        Node *nd = &W.node[L];
        if(nd->leaf) {
            const InterRes &R = W.eres;
            nd->unit_cost = R.cost, nd->cu_mode = R.cu_mode;
            nd->try_intra = R.nnz[0] != 0 || R.nnz[1] != 0 || R.nnz[2] != 0;
        }
    }
    sync(tm);
}

XW void op_leaf(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L, bool full)
{
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (p.log2_ctu - log2), n = 1 << (2 * L), n0 = cu * cu, n1 = p.idc ? n0 >> (p.ws + p.hs) : 0;
    // mode_coding_unit (:1310-1350): the intra analysis becomes the CU's mode in an I slice, and in a P / B slice where it is cheaper than the inter winner
    for(int k = tm.tid; k < nC; k += tm.n) {
        const Cw &W = p.cw[c0 + k];
        const Node &nd = W.node[L];
        S.sh[k][T_F0] = nd.leaf && (!p.inter || (nd.try_intra && W.ires.cost < nd.unit_cost));
    }
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        if(!S.sh[k][T_F0]) return;
        Cw &W = p.cw[c0 + k];
        const IntraRes &R = W.ires;
        unit_to_temp(sub, p, &W.temp[L], log2, cud, n, 0, R.ipm, R.nnz, nullptr, W.slot[R.slot].lev, W.slot[5].lev, W.slot[6].lev, W.slot[R.slot].rec, W.slot[5].rec,
                     W.slot[6].rec, n0, n1);
    });
    sync(tm);
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        Node *nd = &W.node[L];
        int better = 0;
        if(nd->leaf) {
            const int intra_wins = S.sh[k][T_F0];
            if(intra_wins) nd->unit_cost = W.ires.cost, nd->cu_mode = 0, nd->dist_cu = W.ires.dist_cu;
            else nd->dist_cu = 0x7FFFFFFF;
            const double cost_temp = nd->cost_temp + nd->unit_cost;
            better = nd->cost_best > cost_temp;
            if(better) nd->cost_best = cost_temp, nd->best_split = 0, W.tdepth[L] = intra_wins ? W.sbest : W.enext; // (:2116-2135)
            nd->cost_temp = nd->cost_best;
        }
        S.sh[k][T_F1] = better;
    }
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        if(S.sh[k][T_F1]) cud_copy(sub, p, &p.cw[c0 + k].best[L], &p.cw[c0 + k].temp[L], 0, 0, log2, log2, cud);
    });
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        if(S.sh[k][T_F1]) {
            const Node &nd = p.cw[c0 + k].node[L];
            rec_to_pic(sub, p, p.jobs[c0 + k].pic, &p.cw[c0 + k].best[L], nd.x0, nd.y0, cu);
        }
    });
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        Node *nd = &W.node[L];
        const int active = nd->active;
        int next_split = 1;
        if(active && nd->cost_best != XW_MAX_COST && p.inter && cud >= p.ecu_depth && nd->cu_mode == 2 /* MODE_SKIP */) next_split = 0; // early CU termination (:2162-2172)
        if(active && nd->cost_best != XW_MAX_COST && !p.inter) { // early termination in I pictures (:2174-2187)
            const int th = 1 << (2 * log2 + 7);
            if(nd->dist_cu < th) {
                const int bits_inc = (2 * log2 >= 6 ? 2 : 0) + 8;
                if(nd->dist_cu < p.lambda[0] * bits_inc) next_split = 0;
            }
        }
        const int do_split = active && cu > 4 && next_split && cu > p.min_cu && cu > p.min_cuwh;
        nd->do_split = do_split;
        if(do_split) { // SPLIT_QUAD (:2189-2329): split_cu_flag = 1 from the node's entry state
            Sbac run;
            const unsigned bits = full ? split_flag_bits<true>(W.before[L], run, 1) : split_flag_bits<false>(W.before[L], run, 1);
            nd->cost_temp = (double)(int)bits * p.lambda[0];
            W.curr[L] = run;
        }
        S.sh[k][T_F0] = do_split;
    }
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        if(S.sh[k][T_F0]) {
            const Node &nd = p.cw[c0 + k].node[L];
            cud_init(sub, &p.cw[c0 + k].temp[L], log2);
            clear_map(sub, p, p.jobs[c0 + k].pic, nd.x0, nd.y0, cu);
        }
    });
    sync(tm);
}

XW void op_child_done(const Tm &tm, const P &p, int c0, int nC, int L)
{ // L = the parent's level; the quadrant just left is node (L - 1)
    const int log2 = L + 2, cud = 2 * (p.log2_ctu - log2), half = 1 << (log2 - 1);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        Cw &W = p.cw[c0 + k];
        Node       *pn = &W.node[L];
        const Node *ch = &W.node[L - 1];
        if(!ch->active) return;
        if(sub.tid == 0) pn->cost_temp += ch->cost_best;
        cud_copy(sub, p, &W.temp[L], &W.best[L - 1], ch->x0 - pn->x0, ch->y0 - pn->y0, log2 - 1, log2, cud);
        update_map(sub, p, p.jobs[c0 + k].pic, &W.best[L - 1], ch->x0, ch->y0, half);
    });
    sync(tm);
}

XW void op_exit(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L)
{
    const int log2 = L + 2, cu = 1 << log2, cud = 2 * (p.log2_ctu - log2);
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        Node *nd = &W.node[L];
        int split_wins = 0;
        if(nd->active) {
            split_wins = nd->do_split && nd->cost_best - 0.0001 > nd->cost_temp;
            if(split_wins) nd->cost_best = nd->cost_temp, nd->best_split = 5 /* SPLIT_QUAD */, W.tdepth[L] = W.next[L - 1];
            W.next[L] = W.tdepth[L];
        }
        S.sh[k][T_F0] = split_wins;
    }
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        if(S.sh[k][T_F0]) cud_copy(sub, p, &p.cw[c0 + k].best[L], &p.cw[c0 + k].temp[L], 0, 0, log2, log2, cud);
    });
    sync(tm);
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        Cw &W = p.cw[c0 + k];
        const Node &nd = W.node[L];
        if(!nd.active) return;
        rec_to_pic(sub, p, p.jobs[c0 + k].pic, &W.best[L], nd.x0, nd.y0, cu);
        if(cu >= 8 && sub.tid == 0) W.best[L].split_mode[cud][((cu >> 1) >> 2) * (cu >> 2) + ((cu >> 1) >> 2)] = (int8_t)nd.best_split; // xeve_set_split_mode (xeve_util.c:1148-1161)
    });
    sync(tm);
}

XW void op_root_done(const Tm &tm, const P &p, int c0, int nC, int L)
{ // update_to_ctx_map + update_map_scu (:2455-2516), then the products the caller takes
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        const int c = c0 + k;
        Cw &W = p.cw[c];
        const Node    &nd = W.node[L];
        const CtuData *b = &W.best[L];
        update_map(sub, p, p.jobs[c].pic, b, nd.x0, nd.y0, 1 << (L + 2));
        const uint32_t *s = (const uint32_t *)b;
        uint32_t       *d = (uint32_t *)(p.out + c);
        for(int i = sub.tid; i < (int)(sizeof(CtuData) / 4); i += sub.n) d[i] = s[i];
        if(sub.tid == 0) p.out_next[c] = W.next[L], p.out_cost[c] = nd.cost_best;
    });
    sync(tm);
}

// the walk's own state starts from zero (a node the picture cuts leaves its outside part untouched)
XW void walk_clear(const Tm &tm, const P &p, int c0, int nC)
{
    for_chains(tm, nC, [&](int k, const Tm &sub) {
        Cw &W = p.cw[c0 + k];
        uint32_t *a = (uint32_t *)&W.node[0];
        const size_t head = (size_t)((char *)&W.nb[0][0][0] - (char *)&W.node[0]) / 4;
        for(size_t i = sub.tid; i < head; i += sub.n) a[i] = 0;
        uint32_t *b = (uint32_t *)&W.best[0];
        for(size_t i = sub.tid; i < 10 * sizeof(CtuData) / 4; i += sub.n) b[i] = 0;
    });
    sync(tm);
}

// one team: chains [c0, c0 + nC) through the whole schedule
template <bool FULL> XW void walk_team(const Tm &tm, const P &p, Lds &S, int team)
{
    const int c0 = team * p.C, nC = imin(p.C, p.nchains - c0);
    if(nC <= 0) return;
#if XW_DEVICE
    if(tm.tid == 0) S.t0 = clock64();
#endif
    if(tm.tid == 0) S.deal = p.deal;
    walk_clear(tm, p, c0, nC);
    mark(tm, p, S, PR_CLEAR);
    for(int i = 0; i < p.nops; i++) {
        const Op o = p.ops[i];
        switch(o.op) {
        case OP_ENTER: op_enter<FULL>(tm, p, S, c0, nC, o.lvl, o.part), mark(tm, p, S, PR_ENTER); break;
        case OP_LEAF: op_leaf(tm, p, S, c0, nC, o.lvl, FULL), mark(tm, p, S, PR_LEAF); break;
        case OP_CHILD_DONE: op_child_done(tm, p, c0, nC, o.lvl), mark(tm, p, S, PR_CHILD); break;
        case OP_EXIT: op_exit(tm, p, S, c0, nC, o.lvl), mark(tm, p, S, PR_EXIT); break;
        case OP_MID: op_mid(tm, p, S, c0, nC, o.lvl), mark(tm, p, S, PR_MID); break;
        case OP_INTRA: intra_node<FULL>(tm, p, S, c0, nC, o.lvl); break;
        case OP_INTER: inter_node<FULL>(tm, p, S, c0, nC, o.lvl); break;
        default: op_root_done(tm, p, c0, nC, o.lvl), mark(tm, p, S, PR_ROOT); break;
        }
    }
}

} // namespace xw
