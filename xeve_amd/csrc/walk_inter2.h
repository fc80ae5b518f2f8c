// xeve_amd/csrc/walk_inter2.h -- inter_node: the stages of xeve_pinter_analyze_cu for the team's chains (helpers: walk_inter.h)
#pragma once
namespace xw {

// the header of an inter CU in the estimate's syntax (xeve_rdo_bit_cnt_cu_inter, xeve_mode.c:201-274): skip flag 0, pred_mode inter, direct flag, then unless
// direct: inter_pred_idc, per used list refi + mvp_idx + mvd
template <bool FULL> XW void cod_inter_head(Cod &c, const P &p, int dir, const int8_t refi[2], const uint8_t mvpi[2], const int16_t mvd[2][2])
{
    cod_bin<FULL>(c, XEVE_HIP_CTX_SKIP_FLAG, 0);
    cod_bin<FULL>(c, XEVE_HIP_CTX_PRED_MODE, 0);
    cod_bin<FULL>(c, XEVE_HIP_CTX_DIRECT, dir);
    if(dir) return;
    const int v0 = refi[0] >= 0, v1 = refi[1] >= 0;
    if(v0 && v1) cod_bin<FULL>(c, XEVE_HIP_CTX_INTER_DIR, 0);
    else {
        if(p.slice_type == 0) cod_bin<FULL>(c, XEVE_HIP_CTX_INTER_DIR, 1);
        cod_bin<FULL>(c, XEVE_HIP_CTX_INTER_DIR + 1, v0 ? 0 : 1);
    }
    if(v0) {
        cod_refi<FULL>(c, p.nref[0], refi[0]);
        cod_mvp_idx<FULL>(c, mvpi[0]);
        cod_mvd1<FULL>(c, mvd[0][0]), cod_mvd1<FULL>(c, mvd[0][1]);
    }
    if(p.slice_type == 0 && v1) {
        cod_refi<FULL>(c, p.nref[1], refi[1]);
        cod_mvp_idx<FULL>(c, mvpi[1]);
        cod_mvd1<FULL>(c, mvd[1][0]), cod_mvd1<FULL>(c, mvd[1][1]);
    }
}

// pinter_residue_rdo (xeve_pinter.c:906-1336) of candidates cand0 .. cand0 + ncand - 1 (slots of cand_slot()) of every chain that goes on.  mode_of[i] = the PRED_* a
// candidate stands for.  Leaves cost_inter / nnz of the modes in the chains' ISt.
template <bool FULL> XW void residue_rdo(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L, const int *modes, int ncand)
{
    const int log2n = L + 2, N = 1 << log2n, lc = log2n - p.ws, ncomp = p.idc ? 3 : 1;
    // the predictions
    for(int ci = 0; ci < ncand; ci++) {
        const int m = modes[ci], sl = cand_slot(m);
        mc_cus(tm, p, S, nC, log2n, ncomp, [&](int k, int &on, int8_t refi[2], int16_t mv[2][2], pel *dst[3]) {
            const ISt &I = S.ist[k];
            on = I.on && I.go;
            if(!on) return;
            if(m == M_DIR) refi[0] = refi[1] = 0; // xeve_get_mv_dir: reference index 0 of both lists
            else refi[0] = I.refi[m][0], refi[1] = I.refi[m][1];
            for(int l = 0; l < 2; l++) mv[l][0] = I.mv[m][l][0], mv[l][1] = I.mv[m][l][1];
            Cw &W = p.cw[c0 + k];
            dst[0] = W.epred[sl][0], dst[1] = W.epred[sl][1], dst[2] = W.epred[sl][2];
        });
    }
    mark(tm, p, S, PR_E_MC);
    // the blocks: luma of every (chain, candidate), then chroma
    const int nl = nC * ncand;
    Blk *lb = S.blk, *cb = S.blk + nl;
    for(int i = tm.tid; i < nl * ncomp; i += tm.n) {
        const int c = i / nl, e = i - c * nl, k = e / ncand, ci = e - k * ncand, sl = cand_slot(modes[ci]);
        const ISt &I = S.ist[k];
        Cw  &W = p.cw[c0 + k];
        Blk &B = c == 0 ? lb[e] : cb[(c - 1) * nl + e];
        B.on = I.on && I.go;
        B.org = c == 0 ? p.org[0] + (long)I.pic * p.org_pic_l + (long)I.y * p.s_org_l + I.x : p.org[c] + (long)I.pic * p.org_pic_c + (long)(I.y >> p.hs) * p.s_org_c + (I.x >> p.ws);
        B.s_org = c ? p.s_org_c : p.s_org_l, B.pred = W.epred[sl][c], blk_slot(B, &W.slot[3 * sl + c]), B.comp = c, B.nnz = 0, B.nev = 0, B.k = k, B.is_intra = 0;
        B.ssd[0] = B.ssd[1] = 0;
    }
    for(int i = tm.tid; i < nC * 28; i += tm.n) {
        const int k = i / 28, e = i - k * 28;
        if(S.ist[k].on) est_entry(p, p.cw[c0 + k].curr[L], e, S.est[k]); // core->rdoq_est_* of mode_coding_unit (xeve_mode.c:792)
    }
    sync(tm);
    blocks_chain(tm, p, S, lb, nl, log2n, 1);
    if(ncomp > 1) blocks_chain(tm, p, S, cb, 2 * nl, lc, 1);
    if(p.rdo_dbk) { // rdo_dbk_switch (:1016-1095, 1290-1312): the loop filter's share -- of the prediction alone (no luma cbf) into ssd[0], of the reconstruction into ssd[1]
        dbk_stage(tm, p, 2 * nl, log2n, [&](int j, DbkJob &J) {
            const int v = j / nl, e = j - v * nl, k = e / ncand, ci = e - k * ncand, m = modes[ci], sl = cand_slot(m);
            const ISt &I = S.ist[k];
            Cw &W = p.cw[c0 + k];
            J.on = lb[e].on, J.pic = I.pic, J.x = I.x, J.y = I.y, J.intra = 0, J.cbf = v ? lb[e].nnz != 0 : 0, J.two = 0;
            J.refi[0] = m == M_DIR ? 0 : I.refi[m][0], J.refi[1] = m == M_DIR ? 0 : I.refi[m][1];
            for(int l = 0; l < 2; l++) J.mv[2 * l] = I.mv[m][l][0], J.mv[2 * l + 1] = I.mv[m][l][1];
            for(int c = 0; c < 3; c++) {
                const bool has = c < ncomp;
                Blk &B = c == 0 ? lb[e] : cb[(c - 1) * nl + e];
                J.a[c] = !has ? nullptr : v ? B.rec : W.epred[sl][c], J.b[c] = nullptr, J.acc[c] = has ? &B.ssd[v] : nullptr;
            }
        });
        sync(tm);
    }
    // the bit counts, round 1: lanes A (all zero), B (as quantised), C (the component tests) of every (chain, candidate)
    auto blk_of = [&](int e, int c) -> Blk & { return c == 0 ? lb[e] : cb[(c - 1) * nl + e]; };
    auto coefs_of = [&](int e, CoefSet &q, const int nz[3]) {
        for(int c = 0; c < 3; c++) {
            const bool has = c < ncomp;
            q.ev[c] = has ? blk_of(e, c).ev : nullptr, q.nev[c] = has ? blk_of(e, c).nev : 0, q.nnz[c] = has ? nz[c] : 0;
        }
    };
    int *res = S.acc; // per (chain, candidate) 8 ints: bits A, bits B, idx_best Y / U / V, bits of the chosen combination
    coder_stage<FULL>(
        tm, S, 3 * nl,
        [&](int j, const Sbac *&in, Sbac *&out) {
            const int lane = j / nl, e = j - lane * nl, k = e / ncand, ci = e - k * ncand;
            in = &p.cw[c0 + k].curr[L], out = lane < 2 ? &p.cw[c0 + k].cst[cand_slot(modes[ci])][lane] : nullptr; // (the winner's state is core->s_next_best)
            return lb[e].on != 0;
        },
        [&](int j, Cod &c) {
            const int lane = j / nl, e = j - lane * nl, k = e / ncand, ci = e - k * ncand, m = modes[ci];
            const ISt &I = S.ist[k];
            const int store[3] = {lb[e].nnz, ncomp > 1 ? cb[e].nnz : 0, ncomp > 1 ? cb[nl + e].nnz : 0};
            const int zero[3] = {0, 0, 0};
            CoefSet q;
            if(lane < 2) {
                cod_inter_head<FULL>(c, p, m == M_DIR, I.refi[m], I.mvpi[m], I.mvd[m]);
                coefs_of(e, q, lane ? store : zero);
                cod_coef<FULL>(c, p.idc, q, 7, 0);
                res[e * 8 + lane] = (int)cod_bits<FULL>(c);
                return;
            }
            // the component tests (:1180-1218): per coded component, without and with its coefficients from the state the previous component's winner left
            int nnz[3] = {store[0], store[1], store[2]};
            for(int i = 0; i < 3; i++) {
                res[e * 8 + 2 + i] = 0;
                if(store[i] <= 0) continue;
                double comp_best = XW_MAX_COST;
                const Cod c_in = c;
                uint16_t keep[6], with[6];
                const int t0 = i ? 2 : 0, ids[6] = {i == 0 ? XEVE_HIP_CTX_CBF_LUMA : i == 1 ? XEVE_HIP_CTX_CBF_CB : XEVE_HIP_CTX_CBF_CR, XEVE_HIP_CTX_RUN + t0, XEVE_HIP_CTX_RUN + t0 + 1,
                                                XEVE_HIP_CTX_LEVEL + t0, XEVE_HIP_CTX_LEVEL + t0 + 1, XEVE_HIP_CTX_LAST + (i ? 1 : 0)};
                for(int t = 0; t < 6; t++) keep[t] = XW_M(c, ids[t]);
                Cod c_best = c;
                int best_j = 0;
                for(int jj = 0; jj < 2; jj++) {
                    c = c_in;
                    for(int t = 0; t < 6; t++) XW_M(c, ids[t]) = keep[t];
                    cod_reset(c);
                    nnz[i] = jj ? store[i] : 0;
                    coefs_of(e, q, nnz);
                    cod_coef<FULL>(c, p.idc, q, 1 << i, 0);
                    const long long d = (long long)blk_of(e, i).ssd[jj];
                    double cost = (double)d * (i == 0 ? 1 : p.wgt[i - 1]);
                    cost += (double)(int)cod_bits<FULL>(c) * p.lambda[i];
                    if(cost < comp_best) {
                        comp_best = cost, best_j = jj, c_best = c;
                        for(int t = 0; t < 6; t++) with[t] = XW_M(c, ids[t]);
                    }
                }
                c = c_best;
                for(int t = 0; t < 6; t++) XW_M(c, ids[t]) = with[t];
                res[e * 8 + 2 + i] = best_j;
            }
        });
    mark(tm, p, S, PR_E_BITS);
    // round 2: the combination the component tests chose, where it is neither of the two already counted
    coder_stage<FULL>(
        tm, S, nl,
        [&](int e, const Sbac *&in, Sbac *&out) {
            const int k = e / ncand, ci = e - k * ncand;
            in = &p.cw[c0 + k].curr[L], out = &p.cw[c0 + k].cst[cand_slot(modes[ci])][2];
            if(!lb[e].on) return false;
            const int store[3] = {lb[e].nnz, ncomp > 1 ? cb[e].nnz : 0, ncomp > 1 ? cb[nl + e].nnz : 0};
            if(store[0] + store[1] + store[2] == 0) return false;
            const int *r = res + e * 8;
            int nz[3] = {store[0], store[1], store[2]};
            if(r[2] || r[3] || r[4])
                for(int c = 0; c < 3; c++) nz[c] = r[2 + c] ? store[c] : 0;
            return nz[0] != store[0] || nz[1] != store[1] || nz[2] != store[2];
        },
        [&](int e, Cod &c) {
            const int k = e / ncand, ci = e - k * ncand, m = modes[ci];
            const ISt &I = S.ist[k];
            const int store[3] = {lb[e].nnz, ncomp > 1 ? cb[e].nnz : 0, ncomp > 1 ? cb[nl + e].nnz : 0};
            int nz[3];
            for(int cc = 0; cc < 3; cc++) nz[cc] = res[e * 8 + 2 + cc] ? store[cc] : 0;
            CoefSet q;
            cod_inter_head<FULL>(c, p, m == M_DIR, I.refi[m], I.mvpi[m], I.mvd[m]);
            coefs_of(e, q, nz);
            cod_coef<FULL>(c, p.idc, q, 7, 0);
            res[e * 8 + 5] = (int)cod_bits<FULL>(c);
        });
    mark(tm, p, S, PR_E_BITS);
    // the coded-block-flag decision (:1103-1331) in the reference's expression order
    for(int e = tm.tid; e < nl; e += tm.n) {
        if(!lb[e].on) continue;
        const int k = e / ncand, ci = e - k * ncand, m = modes[ci];
        ISt &I = S.ist[k];
        const int store[3] = {lb[e].nnz, ncomp > 1 ? cb[e].nnz : 0, ncomp > 1 ? cb[nl + e].nnz : 0};
        long long dist[2][3];
        for(int c = 0; c < 3; c++) dist[0][c] = c < ncomp ? (long long)blk_of(e, c).ssd[0] : 0, dist[1][c] = c < ncomp ? (long long)blk_of(e, c).ssd[1] : 0;
        const int *r = res + e * 8;
        double cost, cost_best = XW_MAX_COST;
        int cbf[3] = {0, 0, 0};
#define XW_SUM_COST(IY, IU, IV) ((double)dist[IY][0] + (((double)dist[IU][1] * p.wgt[0]) + ((double)dist[IV][2] * p.wgt[1])))
        if(store[0] + store[1] + store[2]) {
            if(m != M_DIR) { // the all-zero alternative (:1103-1142)
                cost = XW_SUM_COST(0, 0, 0);
                cost += (double)r[0] * p.lambda[0];
                if(cost < cost_best) cost_best = cost, cbf[0] = cbf[1] = cbf[2] = 0, I.csel[m] = 0;
            }
            int iy = store[0] > 0, iu = store[1] > 0, iv = store[2] > 0;
            cost = XW_SUM_COST(iy, iu, iv);
            cost += (double)r[1] * p.lambda[0];
            if(cost < cost_best) cost_best = cost, cbf[0] = iy, cbf[1] = iu, cbf[2] = iv, I.csel[m] = 1;
            int nz[3] = {store[0], store[1], store[2]};
            if(r[2] || r[3] || r[4]) {
                iy = r[2], iu = r[3], iv = r[4];
                nz[0] = iy ? store[0] : 0, nz[1] = iu ? store[1] : 0, nz[2] = iv ? store[2] : 0;
            }
            if(nz[0] != store[0] || nz[1] != store[1] || nz[2] != store[2]) {
                cost = XW_SUM_COST(iy, iu, iv);
                cost += (double)r[5] * p.lambda[0];
                if(cost < cost_best) cost_best = cost, cbf[0] = iy, cbf[1] = iu, cbf[2] = iv, I.csel[m] = 2;
            }
            for(int c = 0; c < 3; c++) I.nnz[m][c] = cbf[c] ? store[c] : 0;
        }
        else { // nothing survived quantisation (:1276-1331)
            cost_best = (double)dist[0][0] + (p.wgt[0] * (double)dist[0][1]) + (p.wgt[1] * (double)dist[0][2]);
            cost_best += (double)r[0] * p.lambda[0];
            I.nnz[m][0] = I.nnz[m][1] = I.nnz[m][2] = 0, I.csel[m] = 0;
        }
#undef XW_SUM_COST
        I.cost_inter[m] = cost_best;
    }
    sync(tm);
    mark(tm, p, S, PR_E_GLUE);
}

template <bool FULL> XW void inter_node(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L)
{
    const int log2n = L + 2, N = 1 << log2n, n0 = N * N, Nc = p.idc ? N >> p.ws : 0, n1 = Nc * (p.idc ? N >> p.hs : 0), ncomp = p.idc ? 3 : 1, isb = p.isb;
    const int scuw = N >> 2;
    // ---- the candidates from the unit maps: xeve_get_avail_inter (left / up / up-right; xeve_util.c:652-714) + xeve_get_motion (:526-573) + the collocated vector
    for(int k = tm.tid; k < nC; k += tm.n) {
        Cw &W = p.cw[c0 + k];
        const Node &nd = W.node[L];
        ISt &I = S.ist[k];
        memset(&I, 0, sizeof(I));
        I.on = nd.leaf, I.x = nd.x0, I.y = nd.y0, I.pic = p.jobs[c0 + k].pic;
        for(int m = 0; m < M_NUM; m++) I.cost_inter[m] = XW_MAX_COST;
        W.eres.cu_mode = -1;
        if(!I.on) continue;
        const long mo = (long)I.pic * p.map_pic;
        const uint32_t *ms = p.map_scu + mo;
        const uint8_t  *mt = p.map_tidx + mo;
        const int x_scu = I.x >> 2, y_scu = I.y >> 2, scup = y_scu * p.w_scu + x_scu, t = mt[scup];
        bool ok[3] = {false, false, false};
        const int at[3] = {scup - 1, scup - p.w_scu, scup - p.w_scu + scuw};
        if(x_scu > 0) {
            const uint32_t m = ms[at[0]];
            ok[0] = !((m >> 15) & 1) && (m >> 31) && mt[at[0]] == t && !((m >> 26) & 1); // !IF && COD && same tile && !IBC
        }
        if(y_scu > 0) {
            const uint32_t m = ms[at[1]];
            ok[1] = !((m >> 15) & 1) && mt[at[1]] == t && !((m >> 26) & 1); // (no COD test for the unit above, :681-684)
            if(x_scu + scuw < p.w_scu) {
                const uint32_t r = ms[at[2]];
                ok[2] = (((r >> 15) & 0x10001u) == 0x10000u) && (r >> 31) && mt[at[2]] == t; // MCU_IS_COD_NIF && COD
            }
        }
        for(int l = 0; l <= isb; l++) {
            const int16_t(*col)[2][2] = (l ? p.col1 : p.col0) + mo;
            for(int q = 0; q < 3; q++) I.mvp[l][q][0] = ok[q] ? p.map_mv[mo + at[q]][l][0] : 1, I.mvp[l][q][1] = ok[q] ? p.map_mv[mo + at[q]][l][1] : 1;
            I.mvp[l][3][0] = col[scup][0][0], I.mvp[l][3][1] = col[scup][0][1]; // refp[0][l].map_mv[scup][0]
            int dup = 0;
            for(int q = 1; q < 4; q++)
                for(int e = 0; e < q; e++)
                    if(I.mvp[l][q][0] == I.mvp[l][e][0] && I.mvp[l][q][1] == I.mvp[l][e][1]) dup |= 1 << q;
            I.dup[l] = dup;
        }
        if(isb) {
            const long corner = mo + scup + (scuw - 1) + (long)(scuw - 1) * p.w_scu;
            I.mv_col[0] = p.col1[corner][0][0], I.mv_col[1] = p.col1[corner][0][1];
        }
    }
    sync(tm), mark(tm, p, S, PR_E_CAND);
    if(p.dbg == 1) return;
    // ---- skip / merge (xeve_analyze_skip, xeve_pinter.c:1337-1530): every candidate's uni-directional prediction once ...
    {
        const int ncl = isb ? 2 : 1, per = N + (ncomp > 1 ? 2 * Nc : 0), pk = ncl * p.max_cand * per;
        for(int i = tm.tid; i < nC * pk; i += tm.n) {
            const int k = i / pk, e = i - k * pk, li = e / per, cc = e - li * per, l = li / p.max_cand, idx = li - l * p.max_cand;
            const ISt &I = S.ist[k];
            if(!I.on || ((I.dup[l] >> idx) & 1)) continue;
            const int c = cc < N ? 0 : cc < N + Nc ? 1 : 2, col = c == 0 ? cc : c == 1 ? cc - N : cc - N - Nc;
            mc_uni_column(p, I.x, I.y, N, I.pic, l, 0, I.mvp[l][idx], c, col, p.cw[c0 + k].upred[l][idx][c]);
        }
        sync(tm);
        if(p.dbg == 21) return;
        // ... the SSD of every pair (idx0, idx1) against the original, per component (a lane per (chain, pair, component))
        const int np = p.max_cand * (isb ? p.max_cand : 1);
        for(int i = tm.tid; i < nC * np * ncomp; i += tm.n) {
            const int k = i / (np * ncomp), e = i - k * (np * ncomp), pr = e / ncomp, c = e - pr * ncomp, i0 = isb ? pr / p.max_cand : pr, i1 = isb ? pr - i0 * p.max_cand : 0;
            const ISt &I = S.ist[k];
            if(!I.on || ((I.dup[0] >> i0) & 1) || (isb && ((I.dup[1] >> i1) & 1))) continue;
            Cw &W = p.cw[c0 + k];
            const int w = c ? Nc : N, h = c ? N >> p.hs : N, so = c ? p.s_org_c : p.s_org_l, sh = (p.bd - 8) * 2;
            const pel *o = c == 0 ? p.org[0] + (long)I.pic * p.org_pic_l + (long)I.y * so + I.x : p.org[c] + (long)I.pic * p.org_pic_c + (long)(I.y >> p.hs) * so + (I.x >> p.ws);
            const pel *a = W.upred[0][i0][c], *b = W.upred[1][i1][c];
            const int8_t rf[2] = {0, (int8_t)(isb ? 0 : -1)};
            const int16_t mv[2][2] = {{I.mvp[0][i0][0], I.mvp[0][i0][1]}, {I.mvp[1][i1][0], I.mvp[1][i1][1]}};
            const bool two = isb && !mc_identical(p, I.x, I.y, N, rf, mv);
            u64 acc = 0;
            for(int yy = 0; yy < h; yy++)
                for(int xx = 0; xx < w; xx++) {
                    const int q = yy * w + xx, v = two ? (a[q] + b[q] + 1) >> 1 : a[q], d = v - (int)o[(long)yy * so + xx];
                    acc += (u64)((d * d) >> sh);
                }
            W.sk_ssd[pr][c] = acc, W.sk_dbk[pr][c] = 0;
        }
        if(p.rdo_dbk) { // (:1463-1485) every pair's prediction through the loop filter's estimate; pi->best_ssd stays without it
            sync(tm);
            dbk_stage(tm, p, nC * np, log2n, [&](int j, DbkJob &J) {
                const int k = j / np, pr = j - k * np, i0 = isb ? pr / p.max_cand : pr, i1 = isb ? pr - i0 * p.max_cand : 0;
                const ISt &I = S.ist[k];
                Cw &W = p.cw[c0 + k];
                J.on = I.on && !((I.dup[0] >> i0) & 1) && !(isb && ((I.dup[1] >> i1) & 1));
                J.pic = I.pic, J.x = I.x, J.y = I.y, J.intra = 0, J.cbf = 0;
                J.refi[0] = 0, J.refi[1] = (int8_t)(isb ? 0 : -1);
                J.mv[0] = I.mvp[0][i0][0], J.mv[1] = I.mvp[0][i0][1], J.mv[2] = I.mvp[1][i1][0], J.mv[3] = I.mvp[1][i1][1];
                const int16_t mv[2][2] = {{J.mv[0], J.mv[1]}, {J.mv[2], J.mv[3]}};
                J.two = isb && !mc_identical(p, I.x, I.y, N, J.refi, mv);
                for(int c = 0; c < 3; c++) J.a[c] = c < ncomp ? W.upred[0][i0][c] : nullptr, J.b[c] = c < ncomp ? W.upred[1][i1][c] : nullptr, J.acc[c] = (u64 *)&W.sk_dbk[pr][c];
            });
        }
        if(p.dbg == 22) { sync(tm); return; }
        // the pairs' bits: skip flag + candidate indices from the CU's entry state (xeve_rdo_bit_cnt_cu_skip, xeve_mode.c:276-295)
        int *sbits = S.acc; // [chain][pair]
        coder_stage<FULL>(
            tm, S, nC * np,
            [&](int j, const Sbac *&in, Sbac *&out) {
                const int k = j / np;
                in = &p.cw[c0 + k].curr[L], out = nullptr;
                return S.ist[k].on != 0;
            },
            [&](int j, Cod &c) {
                const int k = j / np, pr = j - k * np, i0 = isb ? pr / p.max_cand : pr, i1 = isb ? pr - i0 * p.max_cand : 0;
                cod_bin<FULL>(c, XEVE_HIP_CTX_SKIP_FLAG, 1);
                cod_mvp_idx<FULL>(c, i0);
                if(isb) cod_mvp_idx<FULL>(c, i1);
                sbits[k * 16 + pr] = (int)cod_bits<FULL>(c);
            });
        if(p.dbg == 23) return;
        // the first pair with the strictly smallest cost; does the CU go on (:1885-1887); the direct candidate (analyze_t_direct + xeve_get_mv_dir)
        for(int k = tm.tid; k < nC; k += tm.n) {
            ISt &I = S.ist[k];
            if(!I.on) continue;
            Cw &W = p.cw[c0 + k];
            double cost_best = XW_MAX_COST;
            long long best_ssd = (long long)1 << (2 * log2n + 16);
            int b0 = 0, b1 = 0;
            for(int i0 = 0; i0 < p.max_cand; i0++) {
                if((I.dup[0] >> i0) & 1) continue;
                for(int i1 = 0; i1 < (isb ? p.max_cand : 1); i1++) {
                    if(isb && ((I.dup[1] >> i1) & 1)) continue;
                    const int pr = isb ? i0 * p.max_cand + i1 : i0;
                    long long cy = (long long)W.sk_ssd[pr][0], cu = ncomp > 1 ? (long long)W.sk_ssd[pr][1] : 0, cv = ncomp > 1 ? (long long)W.sk_ssd[pr][2] : 0;
                    const long long temp_ssd = cy + cu + cv;
                    if(p.rdo_dbk) cy += (long long)W.sk_dbk[pr][0], cu += ncomp > 1 ? (long long)W.sk_dbk[pr][1] : 0, cv += ncomp > 1 ? (long long)W.sk_dbk[pr][2] : 0;
                    double cost = (double)cy + (p.wgt[0] * (double)cu) + (p.wgt[1] * (double)cv);
                    cost += (double)sbits[k * 16 + pr] * p.lambda[0];
                    if(cost < cost_best) cost_best = cost, b0 = i0, b1 = i1, best_ssd = temp_ssd;
                }
            }
            I.cost_inter[M_SKIP] = cost_best;
            for(int l = 0; l < 2; l++) I.mv[M_SKIP][l][0] = I.mvp[l][l ? b1 : b0][0], I.mv[M_SKIP][l][1] = I.mvp[l][l ? b1 : b0][1];
            if(!isb) I.mv[M_SKIP][1][0] = I.mv[M_SKIP][1][1] = 0; // (stale in the reference)
            I.refi[M_SKIP][0] = 0, I.refi[M_SKIP][1] = (int8_t)(isb ? 0 : -1), I.mvpi[M_SKIP][0] = (uint8_t)b0, I.mvpi[M_SKIP][1] = (uint8_t)b1;
            I.go = cost_best < XW_MAX_COST && (double)best_ssd > (double)((int64_t)1 << (2 * log2n + 2 * (p.bd - 8))) * p.skip_th;
            if(isb) {
                const int dpoc_co = p.refp[1].poc - p.col_list_poc0, dpoc_l0 = p.poc - p.refp[0].poc, dpoc_l1 = p.refp[1].poc - p.poc; // xeve_util.c:634-636
                if(dpoc_co != 0) {
                    I.mv[M_DIR][0][0] = (int16_t)(dpoc_l0 * I.mv_col[0] / dpoc_co), I.mv[M_DIR][0][1] = (int16_t)(dpoc_l0 * I.mv_col[1] / dpoc_co);
                    I.mv[M_DIR][1][0] = (int16_t)(-dpoc_l1 * I.mv_col[0] / dpoc_co), I.mv[M_DIR][1][1] = (int16_t)(-dpoc_l1 * I.mv_col[1] / dpoc_co);
                }
            }
        }
        sync(tm);
        if(p.dbg == 24) return;
        // the skip winner's prediction kept (pi->pred[PRED_SKIP])
        for(int i = tm.tid; i < nC * (n0 + 2 * n1); i += tm.n) {
            const int k = i / (n0 + 2 * n1), e = i - k * (n0 + 2 * n1), c = e < n0 ? 0 : e < n0 + n1 ? 1 : 2, q = c == 0 ? e : c == 1 ? e - n0 : e - n0 - n1;
            const ISt &I = S.ist[k];
            if(!I.on || I.cost_inter[M_SKIP] >= XW_MAX_COST) continue;
            Cw &W = p.cw[c0 + k];
            const int i0 = I.mvpi[M_SKIP][0], i1 = I.mvpi[M_SKIP][1];
            const int8_t  rf[2] = {I.refi[M_SKIP][0], I.refi[M_SKIP][1]};
            const int16_t mvv[2][2] = {{I.mv[M_SKIP][0][0], I.mv[M_SKIP][0][1]}, {I.mv[M_SKIP][1][0], I.mv[M_SKIP][1][1]}};
            const bool two = isb && !mc_identical(p, I.x, I.y, N, rf, mvv);
            const pel *a = W.upred[0][i0][c], *b = W.upred[1][isb ? i1 : 0][c];
            W.spred[c][q] = (pel)(two ? (a[q] + b[q] + 1) >> 1 : a[q]);
        }
        sync(tm);
    }
    mark(tm, p, S, PR_E_SKIP);
    if(p.dbg == 2) return;
    // ---- the motion search per list over every reference picture (:1906-1950)
    const int nl = 1 + isb, nrmax = imax(p.nref[0], p.nref[1]);
    {
        const int lists = nC * nl <= XW_MEJ ? nl : 1; // lists searched side by side in one pass
        for(int r = 0; r < nrmax; r++)
            for(int l0 = 0; l0 < nl; l0 += lists) {
                const int nj = nC * lists;
                for(int j = tm.tid; j < nj; j += tm.n) {
                    const int k = j / lists, l = l0 + (j - k * lists);
                    const ISt &I = S.ist[k];
                    MeJob &J = S.mej[j];
                    J.on = 0;
                    if(!I.on || !I.go || r >= p.nref[l]) continue;
                    const int idx = I.mvpi[M_SKIP][l]; // mvp_idx[lidx] = pi->mvp_idx[PRED_SKIP][lidx] (:1927)
                    const pel *o = p.org[0] + (long)I.pic * p.org_pic_l + (long)I.y * p.s_org_l + I.x;
                    const int16_t none[2] = {0, 0};
                    me_job_init(p, J, I, k, l, r, 0, N, o, p.s_org_l, I.mvp[l][idx], I.mvp[l][idx], none, p.refi_bits[l][r], 0);
                }
                sync(tm);
                me_run(tm, p, S, nj, log2n);
                for(int j = tm.tid; j < nj; j += tm.n) {
                    const MeJob &J = S.mej[j];
                    if(!J.on) continue;
                    ISt &I = S.ist[J.k];
                    I.mv_scale[J.l][r][0] = (int16_t)J.mv[0], I.mv_scale[J.l][r][1] = (int16_t)J.mv[1];
                    if(J.mot_bits > 0) I.mot_bits[J.l] = J.mot_bits;
                    // the best reference picture: first strictly smaller cost (:1945-1948)
                    if(r == 0 || J.cost_best < I.best_mecost_l[J.l]) I.best_mecost_l[J.l] = J.cost_best, I.refi[J.l][J.l] = (int8_t)r;
                }
                sync(tm);
            }
    }
    mark(tm, p, S, PR_E_ME);
    if(p.dbg == 3) return;
    // ---- check_best_mvp (:1773-1837): the bits of mvp_idx + mvd for the entry index, then for every index
    {
        for(int k = tm.tid; k < nC; k += tm.n) {
            ISt &I = S.ist[k];
            if(!I.on || !I.go) continue;
            for(int l = 0; l < nl; l++) {
                const int rsel = I.refi[l][l];
                I.mv[l][l][0] = I.mv_scale[l][rsel][0], I.mv[l][l][1] = I.mv_scale[l][rsel][1], I.refi[l][1 - l] = -1;
            }
        }
        sync(tm);
        int *mb = S.acc; // [chain][list][5]
        coder_stage<FULL>(
            tm, S, nC * nl * 5,
            [&](int j, const Sbac *&in, Sbac *&out) {
                const int k = j / (nl * 5);
                in = &p.cw[c0 + k].curr[L], out = nullptr;
                return S.ist[k].on && S.ist[k].go;
            },
            [&](int j, Cod &c) {
                const int k = j / (nl * 5), e = j - k * (nl * 5), l = e / 5, v = e - l * 5;
                const ISt &I = S.ist[k];
                const int idx = v == 0 ? I.mvpi[M_SKIP][l] : v - 1;
                // xeve_rdo_bit_cnt_mvp (xeve_mode.c:57-79): only list l is used
                if(l == 0 || p.slice_type == 0) {
                    cod_mvp_idx<FULL>(c, idx);
                    cod_mvd1<FULL>(c, (int16_t)(I.mv[l][l][0] - I.mvp[l][idx][0])), cod_mvd1<FULL>(c, (int16_t)(I.mv[l][l][1] - I.mvp[l][idx][1]));
                }
                mb[(k * 2 + l) * 5 + v] = (int)cod_bits<FULL>(c);
            });
        for(int k = tm.tid; k < nC; k += tm.n) {
            ISt &I = S.ist[k];
            if(!I.on || !I.go) continue;
            uint8_t pair[2] = {0, 0};
            for(int l = 0; l < nl; l++) {
                const int *b = mb + (k * 2 + l) * 5;
                const double best_cost = (double)b[0] * p.lambda[0]; // never updated: the LAST unpruned index cheaper than the entry index wins (:1829-1831)
                int best_idx = I.mvpi[M_SKIP][l];
                for(int idx = 0; idx < 4; idx++) {
                    if((I.dup[l] >> idx) & 1) continue;
                    if((double)b[1 + idx] * p.lambda[0] < best_cost) best_idx = idx;
                }
                pair[l] = (uint8_t)best_idx;
                I.mvd[l][l][0] = (int16_t)(I.mv[l][l][0] - I.mvp[l][best_idx][0]), I.mvd[l][l][1] = (int16_t)(I.mv[l][l][1] - I.mvp[l][best_idx][1]);
                I.mvpi[l][0] = pair[0], I.mvpi[l][1] = pair[1]; // (the local pair as it stands after each list is what pinter_residue_rdo is given)
            }
        }
        sync(tm);
    }
    mark(tm, p, S, PR_E_GLUE);
    if(p.dbg == 4) return;
    // ---- pinter_residue_rdo of direct + L0 + L1 side by side
    {
        const int mB[3] = {M_DIR, M_L0, M_L1}, mP[1] = {M_L0};
        residue_rdo<FULL>(tm, p, S, c0, nC, L, isb ? mB : mP, isb ? 3 : 1);
    }
    if(p.dbg == 5) return;
    // ---- analyze_bi (:1567-1714)
    if(isb) {
        const int nb = p.nref[1]; // pi->num_refp as the list-1 search left it
        for(int k = tm.tid; k < nC; k += tm.n) {
            ISt &I = S.ist[k];
            I.active = I.on && I.go;
            if(!I.active) continue;
            const int lref = I.cost_inter[M_L0] <= I.cost_inter[M_L1] ? 0 : 1;
            I.lidx_ref = lref, I.best_mecost = 0xFFFFFFFFu, I.refi_best = 0;
            I.mvpi[M_BI][0] = I.mvpi[M_L0][0], I.mvpi[M_BI][1] = I.mvpi[M_L1][1], I.refi[M_BI][0] = I.refi[M_L0][0], I.refi[M_BI][1] = I.refi[M_L1][1];
            I.mv[M_BI][0][0] = I.mv[M_L0][0][0], I.mv[M_BI][0][1] = I.mv[M_L0][0][1], I.mv[M_BI][1][0] = I.mv[M_L1][1][0], I.mv[M_BI][1][1] = I.mv[M_L1][1][1];
            I.rf[lref] = I.refi[M_BI][lref], I.rf[1 - lref] = -1;
        }
        sync(tm);
        for(int it = 0; it < 4; it++) { // BI_ITER
            // the prediction from the fixed list (luma is all get_org_bi reads), org_bi = 2 * org - pred (:143-156)
            for(int i = tm.tid; i < nC * N; i += tm.n) {
                const int k = i >> log2n, col = i & (N - 1);
                const ISt &I = S.ist[k];
                if(!I.active) continue;
                const int l = I.lidx_ref;
                mc_uni_column(p, I.x, I.y, N, I.pic, l, I.rf[l], I.mv[M_BI][l], 0, col, p.cw[c0 + k].bpred);
            }
            sync(tm);
            for(int i = tm.tid; i < nC * n0; i += tm.n) {
                const int k = i >> (2 * log2n), q = i & (n0 - 1);
                const ISt &I = S.ist[k];
                if(!I.active) continue;
                Cw &W = p.cw[c0 + k];
                const pel *o = p.org[0] + (long)I.pic * p.org_pic_l + (long)(I.y + (q >> log2n)) * p.s_org_l + I.x + (q & (N - 1));
                W.org_bi[q] = (int16_t)((o[0] << 1) - W.bpred[q]);
            }
            for(int k = tm.tid; k < nC; k += tm.n) { // SWAP(refi[lidx_ref], refi[lidx_cnd]), SWAP(lidx_ref, lidx_cnd) (:1626-1628)
                ISt &I = S.ist[k];
                if(!I.active) continue;
                const int8_t t = I.rf[0];
                I.rf[0] = I.rf[1], I.rf[1] = t, I.lidx_ref = 1 - I.lidx_ref, I.changed = 0;
            }
            sync(tm);
            for(int r = 0; r < nb; r++) {
                for(int j = tm.tid; j < nC; j += tm.n) {
                    const ISt &I = S.ist[j];
                    MeJob &J = S.mej[j];
                    J.on = 0;
                    if(!I.active) continue;
                    const int l = I.lidx_ref, idx = I.mvpi[M_BI][l];
                    me_job_init(p, J, I, j, l, r, 1, N, p.cw[c0 + j].org_bi, N, I.mvp[l][idx], I.mv_scale[l][r], I.mv_scale[l][r], p.refi_bits[1][r], I.mot_bits[1 - l]);
                }
                sync(tm);
                me_run(tm, p, S, nC, log2n);
                for(int j = tm.tid; j < nC; j += tm.n) {
                    const MeJob &J = S.mej[j];
                    if(!J.on) continue;
                    ISt &I = S.ist[j];
                    const int l = I.lidx_ref;
                    I.mv_scale[l][r][0] = (int16_t)J.mv[0], I.mv_scale[l][r][1] = (int16_t)J.mv[1]; // fn_me refines pi->mv_scale[lidx_ref][refi_cur] in place
                    if(J.cost_best < I.best_mecost) {
                        I.refi_best = r, I.best_mecost = J.cost_best, I.changed = 1, I.refi[M_BI][l] = (int8_t)r;
                        I.mv[M_BI][l][0] = (int16_t)J.mv[0], I.mv[M_BI][l][1] = (int16_t)J.mv[1];
                    }
                }
                sync(tm);
            }
            if(tm.tid == 0) S.flag[1] = 0;
            sync(tm);
            for(int k = tm.tid; k < nC; k += tm.n) {
                ISt &I = S.ist[k];
                if(!I.active) continue;
                const int l = I.lidx_ref;
                I.rf[l] = (int8_t)I.refi_best, I.rf[1 - l] = -1;
                if(!I.changed) I.active = 0;
                else aor(&S.flag[1], 1);
            }
            sync(tm);
            const int more = S.flag[1];
            sync(tm);
            if(!more) break;
        }
        for(int k = tm.tid; k < nC; k += tm.n) {
            ISt &I = S.ist[k];
            if(!I.on || !I.go) continue;
            for(int l = 0; l < 2; l++)
                for(int d = 0; d < 2; d++) I.mvd[M_BI][l][d] = (int16_t)(I.mv[M_BI][l][d] - I.mvp[l][I.mvpi[M_BI][l]][d]);
        }
        sync(tm);
        mark(tm, p, S, PR_E_ME);
        const int mBI[1] = {M_BI};
        residue_rdo<FULL>(tm, p, S, c0, nC, L, mBI, 1);
    }
    if(p.dbg == 6) return;
    // ---- the decision (:1872-2001): first strictly smaller cost in the order skip, direct, L0, L1, bi; the winner's data
    for(int k = tm.tid; k < nC; k += tm.n) {
        ISt &I = S.ist[k];
        if(!I.on) continue;
        Cw &W = p.cw[c0 + k];
        double cost_best = XW_MAX_COST;
        int best = M_SKIP, cu_mode = -1;
        const int order[5] = {M_SKIP, M_DIR, M_L0, M_L1, M_BI};
        for(int q = 0; q < 5; q++) {
            const int m = order[q];
            if(m != M_SKIP && !I.go) continue;
            if(I.cost_inter[m] < cost_best) cost_best = I.cost_inter[m], best = m, cu_mode = m == M_SKIP ? 2 : m == M_DIR ? 3 : 1;
        }
        I.best = (int8_t)best, I.cu_mode = (int8_t)cu_mode;
        InterRes &R = W.eres;
        memset(&R, 0, sizeof(R));
        R.cost = I.cost_inter[best];
        for(int m = 0; m < M_NUM; m++) R.cost_inter[m] = (m == M_SKIP || I.go) ? I.cost_inter[m] : XW_MAX_COST;
        R.cu_mode = cu_mode, R.best_idx = best, R.slot = best == M_SKIP ? -1 : 3 * cand_slot(best);
        const int8_t rd[2] = {0, 0};
        const int8_t *rfi = best == M_DIR ? rd : I.refi[best];
        for(int l = 0; l < 2; l++) {
            const bool lst = isb || l == 0, used = lst && rfi[l] >= 0;
            R.refi[l] = lst ? rfi[l] : -1;
            if(used) R.mv[l][0] = I.mv[best][l][0], R.mv[l][1] = I.mv[best][l][1], R.mvd[l][0] = I.mvd[best][l][0], R.mvd[l][1] = I.mvd[best][l][1], R.mvp_idx[l] = I.mvpi[best][l];
        }
        if(best == M_DIR) R.mvp_idx[0] = R.mvp_idx[1] = 0, R.mvd[0][0] = R.mvd[0][1] = R.mvd[1][0] = R.mvd[1][1] = 0;
        if(best != M_SKIP)
            for(int c = 0; c < 3; c++) R.nnz[c] = I.nnz[best][c];
        for(int t = 0; t < 4; t++) S.acc[k * XW_ACC + t] = 0; // (the SATD of the winner's luma prediction)
    }
    sync(tm);
    // the winner's reconstruction (:2004-2032): dequantisation + inverse transform of its levels on its prediction
    {
        Blk *wb = S.blk;
        for(int i = tm.tid; i < nC * ncomp; i += tm.n) {
            const int c = i / nC, k = i - c * nC;
            const ISt &I = S.ist[k];
            Cw  &W = p.cw[c0 + k];
            Blk &B = wb[i];
            B.on = I.on && I.cu_mode >= 0;
            const int best = I.best, sl = cand_slot(best);
            B.org = c == 0 ? p.org[0] + (long)I.pic * p.org_pic_l + (long)I.y * p.s_org_l + I.x : p.org[c] + (long)I.pic * p.org_pic_c + (long)(I.y >> p.hs) * p.s_org_c + (I.x >> p.ws);
            B.s_org = c ? p.s_org_c : p.s_org_l, B.pred = best == M_SKIP ? W.spred[c] : W.epred[sl][c];
            blk_slot(B, &W.slot[3 * (best == M_SKIP ? 0 : sl) + c]);
            B.rec = W.wrec[c], B.comp = c, B.nnz = best == M_SKIP ? 0 : I.nnz[best][c], B.nev = 0, B.k = k, B.is_intra = 0, B.ssd[0] = B.ssd[1] = 0;
        }
        sync(tm);
        st_dquant(tm, p, wb, nC, log2n);
        if(ncomp > 1) st_dquant(tm, p, wb + nC, 2 * nC, log2n - p.ws);
        sync(tm);
        for(int pass = 2; pass < 4; pass++) {
            st_tpass(tm, p, wb, nC, log2n, pass);
            if(ncomp > 1) st_tpass(tm, p, wb + nC, 2 * nC, log2n - p.ws, pass);
            sync(tm);
        }
        st_recon(tm, p, wb, nC, log2n);
        if(ncomp > 1) st_recon(tm, p, wb + nC, 2 * nC, log2n - p.ws);
        // core->inter_satd = xeve_satd_16b(original, mi->pred_y_best) (mode_check_intra, :1250-1262)
        const int tn = N >= 8 ? 8 : 4, tl = N / tn, tiles = tl * tl; // (a 4x4 CU is one 4x4 tile, xeve_had_4x4)
        for(int i = tm.tid; i < nC * tiles; i += tm.n) {
            const int k = i / tiles, t = i - k * tiles, ty = (t / tl) * tn, tx = (t % tl) * tn;
            const Blk &B = wb[k];
            if(!B.on) continue;
            aadd(&S.acc[k * XW_ACC], had_tile(B.org + (long)ty * B.s_org + tx, B.s_org, B.pred + ty * N + tx, N, tn));
        }
        sync(tm);
    }
    mark(tm, p, S, PR_E_FINAL);
    if(p.dbg == 7) return;
    // core->s_next_best: the state the winner's deciding count left (kept by that count); a skipped CU's three bins are counted again
    coder_stage<FULL>(
        tm, S, nC,
        [&](int k, const Sbac *&in, Sbac *&out) {
            in = &p.cw[c0 + k].curr[L], out = &p.cw[c0 + k].enext;
            return S.ist[k].on && S.ist[k].cu_mode >= 0 && S.ist[k].best == M_SKIP;
        },
        [&](int k, Cod &c) {
            const ISt &I = S.ist[k];
            cod_bin<FULL>(c, XEVE_HIP_CTX_SKIP_FLAG, 1);
            cod_mvp_idx<FULL>(c, I.mvpi[M_SKIP][0]);
            if(isb) cod_mvp_idx<FULL>(c, I.mvpi[M_SKIP][1]);
        });
    for(int i = tm.tid; i < nC * (int)(sizeof(Sbac) / 4); i += tm.n) {
        const int k = i / (int)(sizeof(Sbac) / 4), w = i - k * (int)(sizeof(Sbac) / 4);
        const ISt &I = S.ist[k];
        if(!I.on || I.cu_mode < 0) continue;
        Cw &W = p.cw[c0 + k];
        if(w == 0) W.eres.satd = (uint32_t)(S.acc[k * XW_ACC] >> (p.bd - 8));
        if(I.best != M_SKIP) ((uint32_t *)&W.enext)[w] = ((const uint32_t *)&W.cst[cand_slot(I.best)][I.csel[I.best]])[w];
    }
    sync(tm);
    mark(tm, p, S, PR_E_BITS);
}

} // namespace xw
