// xeve_amd/csrc/inter.hip -- the whole inter analysis of a batch of CUs of one size, composed on the device.
//
// reference: xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047) = ctx->fn_pinter_analyze_cu, Baseline (tool_admvp 0):
//   skip / merge analysis (xeve_analyze_skip)                                   -> xeve_hip_analyze_skip_jobs
//   temporal direct (analyze_t_direct + xeve_get_mv_dir)                         -> a candidate of xeve_hip_residue_rdo_jobs
//   per list: motion search over every reference picture (pi->fn_me = pinter_me_epzs), best reference picture, check_best_mvp
//   (CABAC bit counts of mvp_idx + mvd for every predictor), pinter_residue_rdo  -> xeve_hip_me_epzs_jobs, xeve_hip_cu_bits_jobs, ..residue_rdo_jobs
//   analyze_bi: up to BI_ITER rounds of {predict from the fixed list, org_bi = 2 * org - pred, search the other list over every reference
//   picture from where its last search ended}, pinter_residue_rdo                -> xeve_hip_mc_cu_jobs, xeve_hip_me_epzs_jobs_x, ..residue_rdo_jobs
//   the cheapest mode (first strictly smaller cost in the order skip, direct, L0, L1, bi), its coefficients, reconstruction
//   (xeve_itdq + xeve_recon), motion data and coder state.
// Everything heavy is one of the batched entry points; this file adds the per-CU glue kernels (one thread per CU: a few dozen scalar
// operations each) that turn the results of one stage into the jobs of the next, so that the whole analysis of all CUs of a level runs
// without a host round trip.  CUs whose skip residual is below the skip_th threshold stop after the skip mode in the reference; here they
// ride along (skip_th is 0 in every preset, so this only concerns CUs with a perfect skip prediction) and are masked in the decision.
#include <cstring>
#include <cstdlib>
#include <vector>
#include "xh_common.h"
#include "mc_cu.h"

#define MAXR XEVE_HIP_MAX_REFP
#define MAX_COST 1.7e+308
enum { M_L0 = 0, M_L1 = 1, M_BI = 2, M_SKIP = 3, M_DIR = 4, M_NUM = 5 }; // PRED_* (xeve_def.h:461-469)

struct InterK {
    int    n, isb, lw, n0, n1, ncomp, bd, max_cand, nref[2], nb, na, np; // np: planes per list in the search job / result arrays // na: candidates of the first pinter_residue_rdo batch (3n in B, n in P)
    int    s_org_l, s_org_c, ws, hs;
    int    dpoc_co, dpoc_l0, dpoc_l1;
    double thr, lambda0;
};

struct InterSt { // per CU, between the stages
    int16_t  mv[M_NUM][2][2], mvd[M_NUM][2][2];
    int8_t   refi[M_NUM][2];
    uint8_t  mvpi[M_NUM][2];
    int16_t  mv_scale[2][MAXR][2];
    int32_t  mot_bits[2];
    int32_t  go;
    // analyze_bi
    int32_t  lidx_ref, active, refi_best, bi_slot;
    uint32_t best_mecost;
    int8_t   rf[2];
    int8_t   pad_[2];
};
static_assert(sizeof(InterSt) % 4 == 0, "InterSt is cleared word by word");

__device__ __forceinline__ void copy_sbac(xeve_hip_sbac *d, const xeve_hip_sbac *s)
{
    const unsigned *a = (const unsigned *)s;
    unsigned       *b = (unsigned *)d;
#pragma unroll
    for(int i = 0; i < (int)(sizeof(xeve_hip_sbac) / 4); i++) b[i] = a[i];
}

__device__ __forceinline__ void rdo_job(xeve_hip_rdo_job &r, const xeve_hip_inter_job &J, const InterSt &S, int m, int dir)
{
    r.x = J.x, r.y = J.y;
    for(int l = 0; l < 2; l++) r.mv[l][0] = S.mv[m][l][0], r.mv[l][1] = S.mv[m][l][1], r.mvd[l][0] = S.mvd[m][l][0], r.mvd[l][1] = S.mvd[m][l][1];
    r.refi[0] = S.refi[m][0], r.refi[1] = S.refi[m][1], r.mvp_idx[0] = S.mvpi[m][0], r.mvp_idx[1] = S.mvpi[m][1];
    r.dir_flag = (uint8_t)dir, r.ctx_skip = J.ctx_skip, r.ctx_pred_mode = J.ctx_pred_mode, r.pad_ = 0, r.sbac = J.sbac;
}

// ---- the candidates of the jobs from the encoder's per-unit maps ---------------------------------------------------------------------
// xeve_get_avail_inter (left / up / up-right; xeve_util.c:652-714) + xeve_get_motion (xeve_util.c:526-573) + the collocated vector of the
// temporal direct mode.  One thread per CU; the maps are [unit][list][x, y] as the reference keeps them.
__device__ __forceinline__ void inter_candidates_one(const XhInterCand &C, xeve_hip_inter_job &J)
{
    const uint32_t *__restrict__ map_scu = C.map_scu;
    const uint8_t *__restrict__  map_tidx = C.map_tidx;
    const int16_t *__restrict__  map_mv = C.map_mv, *__restrict__ col0 = C.col0, *__restrict__ col1 = C.col1;
    const int w_scu = C.w_scu, scuw = C.scuw, scuh = C.scuh, isb = C.isb, vh = C.vh;
    // (a batch of pictures stacked vertically, xh_common.h: the maps are stacked like the planes, so the unit addresses follow from y as it is; only "is there a row
    // above" asks for the row inside the job's own picture)
    const int x_scu = J.x >> 2, y_scu = J.y >> 2, scup = y_scu * w_scu + x_scu, y_in_pic = (J.y - xh_vh_base(J.y, vh)) >> 2;
    auto tile = [&](int at) { return map_tidx ? (int)map_tidx[at] : 0; };
    const int t = tile(scup);
    bool ok[3] = {false, false, false};
    const int at[3] = {scup - 1, scup - w_scu, scup - w_scu + scuw};
    if(x_scu > 0) {
        const uint32_t m = map_scu[at[0]];
        ok[0] = !((m >> 15) & 1) && (m >> 31) && tile(at[0]) == t && !((m >> 26) & 1); // !IF && COD && same tile && !IBC
    }
    if(y_in_pic > 0) {
        const uint32_t m = map_scu[at[1]];
        ok[1] = !((m >> 15) & 1) && tile(at[1]) == t && !((m >> 26) & 1); // (no COD test for the unit above, :681-684)
        if(x_scu + scuw < w_scu) {
            const uint32_t r = map_scu[at[2]];
            ok[2] = (((r >> 15) & 0x10001u) == 0x10000u) && (r >> 31) && tile(at[2]) == t; // MCU_IS_COD_NIF && COD
        }
    }
    for(int l = 0; l < 2; l++) {
        const int16_t *col = l ? col1 : col0;
        for(int k = 0; k < 4; k++) {
            int vx = 0, vy = 0;
            if(l <= isb) {
                if(k < 3) vx = ok[k] ? map_mv[((size_t)at[k] * 2 + l) * 2] : 1, vy = ok[k] ? map_mv[((size_t)at[k] * 2 + l) * 2 + 1] : 1;
                else vx = col[((size_t)scup * 2 + 0) * 2], vy = col[((size_t)scup * 2 + 0) * 2 + 1]; // refp[0][l].map_mv[scup][0]
            }
            J.mvp[l][k][0] = (int16_t)vx, J.mvp[l][k][1] = (int16_t)vy;
        }
    }
    J.mv_col[0] = J.mv_col[1] = 0;
    if(isb) {
        const size_t corner = (size_t)scup + (scuw - 1) + (size_t)(scuh - 1) * w_scu;
        J.mv_col[0] = col1[(corner * 2 + 0) * 2], J.mv_col[1] = col1[(corner * 2 + 0) * 2 + 1];
    }
}
__global__ void k_inter_candidates(XhInterCand C, xeve_hip_inter_job *__restrict__ jobs, int njobs)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= njobs) return;
    xeve_hip_inter_job J = jobs[j];
    inter_candidates_one(C, J);
    jobs[j] = J;
}


// ---- skip --------------------------------------------------------------------------------------------------------------------
__global__ void k_inter_skip_jobs(xeve_hip_inter_job *__restrict__ jobs, InterK P, xeve_hip_skip_job *__restrict__ sj, XhInterCand C)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.n) return;
    xeve_hip_inter_job J = jobs[j];
    if(C.map_scu) { // (the walk: the merge / MVP candidates of the CU from the per-unit maps in the same launch -- xeve_hip_inter_candidates for the C-ABI's callers)
        inter_candidates_one(C, J);
        jobs[j] = J;
    }
    xeve_hip_skip_job s;
    s.x = J.x, s.y = J.y;
    for(int l = 0; l < 2; l++)
        for(int i = 0; i < 4; i++) s.mvp[l][i][0] = J.mvp[l][i][0], s.mvp[l][i][1] = J.mvp[l][i][1], s.refi_pred[l][i] = 0; // xeve_get_motion: always index 0
    s.ncand = P.max_cand, s.sbac = J.sbac, s.ctx_skip = J.ctx_skip, s.pad_[0] = s.pad_[1] = s.pad_[2] = 0;
    sj[j] = s;
}

// after the skip analysis: does the CU go on (:1885-1887); the direct candidate (analyze_t_direct, xeve_get_mv_dir); the search jobs
__global__ void k_inter_stage1(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const xeve_hip_skip_result *__restrict__ sres, InterSt *__restrict__ st,
                               xeve_hip_rdo_job *__restrict__ rj, xeve_hip_epzs_job *__restrict__ ej, int32_t *__restrict__ cnt)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j < 8) cnt[j] = 0; // (the slot counters of analyze_bi's rounds, two per round: k_bi_head)
    if(j >= P.n) return;
    const xeve_hip_inter_job   J = jobs[j];
    const xeve_hip_skip_result R = sres[j];
    InterSt S;
    for(int i = 0; i < (int)(sizeof(S) / 4); i++) ((int *)&S)[i] = 0;
    for(int l = 0; l < 2; l++) S.mv[M_SKIP][l][0] = R.mv[l][0], S.mv[M_SKIP][l][1] = R.mv[l][1], S.refi[M_SKIP][l] = R.refi[l];
    if(!P.isb) S.mv[M_SKIP][1][0] = S.mv[M_SKIP][1][1] = 0;
    S.mvpi[M_SKIP][0] = (uint8_t)R.idx0, S.mvpi[M_SKIP][1] = (uint8_t)R.idx1;
    S.go = R.cost < MAX_COST && (double)R.best_ssd > P.thr;
    S.active = 1, S.best_mecost = 0xFFFFFFFFu;
    if(P.isb) {
        if(P.dpoc_co != 0) {
            S.mv[M_DIR][0][0] = (int16_t)(P.dpoc_l0 * J.mv_col[0] / P.dpoc_co), S.mv[M_DIR][0][1] = (int16_t)(P.dpoc_l0 * J.mv_col[1] / P.dpoc_co);
            S.mv[M_DIR][1][0] = (int16_t)(-P.dpoc_l1 * J.mv_col[0] / P.dpoc_co), S.mv[M_DIR][1][1] = (int16_t)(-P.dpoc_l1 * J.mv_col[1] / P.dpoc_co);
        }
        rdo_job(rj[j], J, S, M_DIR, 1);
    }
    for(int l = 0; l <= P.isb; l++) {
        xeve_hip_epzs_job e;
        const int idx = S.mvpi[M_SKIP][l]; // mvp_idx[lidx] = pi->mvp_idx[PRED_SKIP][lidx] (:1927)
        e.y = J.y, e.org_off = 0, e.mvp[0] = J.mvp[l][idx][0], e.mvp[1] = J.mvp[l][idx][1], e.mv_start[0] = e.mv_start[1] = 0;
        for(int r = 0; r < P.np; r++) { // the same search against every reference picture of the list (planes the list does not hold: off)
            e.x = r < P.nref[l] ? J.x : -1;
            ej[((size_t)l * P.np + r) * P.n + j] = e;
        }
    }
    st[j] = S;
}

// ---- uni-directional modes -------------------------------------------------------------------------------------------------------
// after the searches of one list: best reference picture (first strictly smaller mecost, :1945-1948), pi->mot_bits as the LAST search left it,
// and the five bit-count jobs of check_best_mvp (entry index, then indices 0..3)
__global__ void k_inter_uni_a(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const xeve_hip_me_result *__restrict__ mres, InterSt *__restrict__ st,
                              xeve_hip_cu_bits_job *__restrict__ bj)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= P.n * (1 + P.isb)) return;
    const int l = t / P.n, j = t - l * P.n;
    const xeve_hip_inter_job J = jobs[j];
    InterSt &S = st[j];
    unsigned best = 0xFFFFFFFFu;
    int      rsel = 0;
    for(int r = 0; r < P.nref[l]; r++) {
        const xeve_hip_me_result m = mres[((size_t)l * P.np + r) * P.n + j];
        S.mv_scale[l][r][0] = m.mv[0], S.mv_scale[l][r][1] = m.mv[1];
        if(m.cost < best) best = m.cost, rsel = r;
        if(m.best_mv_bits > 0) S.mot_bits[l] = m.best_mv_bits;
    }
    const int mvx = S.mv_scale[l][rsel][0], mvy = S.mv_scale[l][rsel][1];
    S.mv[l][l][0] = (int16_t)mvx, S.mv[l][l][1] = (int16_t)mvy;
    S.refi[l][l] = (int8_t)rsel, S.refi[l][1 - l] = -1;
    const int entry = S.mvpi[M_SKIP][l];
    for(int k = 0; k < 5; k++) {
        const int idx = k == 0 ? entry : k - 1;
        xeve_hip_cu_bits_job b;
        b.coef_off[0] = b.coef_off[1] = b.coef_off[2] = 0, b.nnz[0] = b.nnz[1] = b.nnz[2] = 0, b.sbac = J.sbac;
        b.mvd[0][0] = b.mvd[0][1] = b.mvd[1][0] = b.mvd[1][1] = 0;
        b.mvd[l][0] = (int16_t)(mvx - J.mvp[l][idx][0]), b.mvd[l][1] = (int16_t)(mvy - J.mvp[l][idx][1]);
        b.refi[l] = (int8_t)rsel, b.refi[1 - l] = -1, b.mvp_idx[0] = b.mvp_idx[1] = (uint8_t)idx;
        b.mode = XEVE_HIP_BITS_MVP, b.dir_flag = 0, b.ctx_skip = 0, b.ctx_pred_mode = 0;
        bj[(size_t)t * 5 + k] = b;
    }
}

// check_best_mvp (:1773-1837): the LAST unpruned index cheaper than the entry index wins (best_cost is never updated); the candidates of
// pinter_residue_rdo for L0 / L1 with the local mvp_idx pair as it stands after each list
__global__ void k_inter_uni_b(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const unsigned *__restrict__ bits, InterSt *__restrict__ st,
                              xeve_hip_rdo_job *__restrict__ rj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.n) return;
    const xeve_hip_inter_job J = jobs[j];
    InterSt S = st[j];
    uint8_t pair[2] = {0, 0};
    for(int l = 0; l <= P.isb; l++) {
        const unsigned *b = bits + ((size_t)l * P.n + j) * 5;
        const double best_cost = (double)(int)b[0] * P.lambda0;
        int best_idx = S.mvpi[M_SKIP][l];
        for(int idx = 0; idx < 4; idx++) {
            bool same = false;
            for(int t = 0; t < idx; t++) same = same || (J.mvp[l][idx][0] == J.mvp[l][t][0] && J.mvp[l][idx][1] == J.mvp[l][t][1]);
            if(same) continue;
            if((double)(int)b[1 + idx] * P.lambda0 < best_cost) best_idx = idx;
        }
        pair[l] = (uint8_t)best_idx;
        S.mvd[l][l][0] = (int16_t)(S.mv[l][l][0] - J.mvp[l][best_idx][0]), S.mvd[l][l][1] = (int16_t)(S.mv[l][l][1] - J.mvp[l][best_idx][1]);
        S.mvpi[l][0] = pair[0], S.mvpi[l][1] = pair[1];
        rdo_job(rj[(size_t)(P.isb ? 1 + l : 0) * P.n + j], J, S, l, 0);
    }
    st[j] = S;
}

// ---- analyze_bi (:1567-1714) -------------------------------------------------------------------------------------------------------
// One kernel per round in front of the round's prediction and search (round 6: k_bi_init / k_bi_update, k_bi_mc_jobs, the prediction's own front half, k_bi_jobs_off and
// k_bi_me_jobs were five launches of one thread per CU each): the previous round's results taken in (round 0: the set-up), the prediction job from the fixed list with
// its interpolation jobs (mc_cu.h), the list swap and the search jobs of the round.
__device__ __forceinline__ void bi_init(const InterK &P, const xeve_hip_rdo_result *__restrict__ rres, InterSt &S, int j)
{
    const int lref = rres[P.n + j].cost <= rres[2 * (size_t)P.n + j].cost ? 0 : 1; // cost_inter[PRED_L0] <= cost_inter[PRED_L1]
    S.lidx_ref = lref;
    S.mvpi[M_BI][0] = S.mvpi[M_L0][0], S.mvpi[M_BI][1] = S.mvpi[M_L1][1], S.refi[M_BI][0] = S.refi[M_L0][0], S.refi[M_BI][1] = S.refi[M_L1][1];
    S.mv[M_BI][0][0] = S.mv[M_L0][0][0], S.mv[M_BI][0][1] = S.mv[M_L0][0][1], S.mv[M_BI][1][0] = S.mv[M_L1][1][0], S.mv[M_BI][1][1] = S.mv[M_L1][1][1];
    S.rf[lref] = S.refi[M_BI][lref], S.rf[1 - lref] = -1;
}
// one round, second half (:1633-1663): every reference picture of the searched list against the running best
__device__ __forceinline__ void bi_update(const InterK &P, const xeve_hip_me_result *__restrict__ mres, InterSt &S)
{
    if(!S.active) return;
    const int l = S.lidx_ref;
    int changed = 0;
    for(int r = 0; r < P.nb; r++) {
        const xeve_hip_me_result m = mres[(size_t)r * P.n + S.bi_slot];
        S.mv_scale[l][r][0] = m.mv[0], S.mv_scale[l][r][1] = m.mv[1]; // fn_me refines pi->mv_scale[lidx_ref][refi_cur] in place
        if(m.cost < S.best_mecost) {
            S.refi_best = r, S.best_mecost = m.cost, changed = 1;
            S.refi[M_BI][l] = (int8_t)r;
            S.mv[M_BI][l][0] = m.mv[0], S.mv[M_BI][l][1] = m.mv[1];
        }
    }
    S.rf[l] = (int8_t)S.refi_best, S.rf[1 - l] = -1;
    if(!changed) S.active = 0;
}
// cnt: this round's two counters (zero: k_inter_stage1 clears all of them).  The still-searching CUs are compacted to the FRONT of the search-job arrays (slots handed out
// per wave with one atomic; the order is arbitrary, results come back through bi_slot), the others take the slots from the BACK and switch them off: every slot is
// written exactly once, nothing has to be cleared beforehand.  A CU's jobs -- one per reference picture of the list it searches -- carry their plane (list, picture)
// in job_plane.
__global__ void k_bi_head(const xeve_hip_inter_job *__restrict__ jobs, InterK P, InterSt *__restrict__ st, const xeve_hip_rdo_result *__restrict__ rres,
                          const xeve_hip_me_result *__restrict__ mres, int first, CuMcPrep C, xeve_hip_epzs_job *__restrict__ ej, int32_t *__restrict__ extra,
                          int32_t *__restrict__ cnt, unsigned char *__restrict__ job_plane)
{
    const int  j = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool in = j < P.n;
    bool act = false;
    int  l = 0;
    xeve_hip_inter_job J;
    if(in) {
        InterSt &S = st[j];
        J = jobs[j];
        if(first) bi_init(P, rres, S, j);
        else bi_update(P, mres, S);
        // first half of the round: the prediction from the fixed list
        xeve_hip_cu_mc_job m;
        m.x = J.x, m.y = J.y, m.pad_[0] = m.pad_[1] = 0;
        for(int q = 0; q < 2; q++) m.mv[q][0] = S.mv[M_BI][q][0], m.mv[q][1] = S.mv[M_BI][q][1], m.refi[q] = S.active ? S.rf[q] : -1;
        xh_cu_mc_prep_one(m, j, C);
        act = S.active != 0;
        if(act) { // SWAP(refi[lidx_ref], refi[lidx_cnd]), SWAP(lidx_ref, lidx_cnd) (:1626-1628)
            const int8_t t = S.rf[0];
            S.rf[0] = S.rf[1], S.rf[1] = t;
            l = S.lidx_ref = 1 - S.lidx_ref;
        }
    }
    const unsigned long long ma = __ballot(act), mi = __ballot(in && !act);
    int ba = 0, bi = 0;
    if(lane == 0 && ma) ba = atomicAdd(&cnt[0], __popcll(ma));
    if(lane == 0 && mi) bi = atomicAdd(&cnt[1], __popcll(mi));
    ba = __shfl(ba, 0, 64), bi = __shfl(bi, 0, 64);
    if(!in) return;
    const unsigned long long below = (1ull << lane) - 1;
    const int k = act ? ba + __popcll(ma & below) : P.n - 1 - (bi + __popcll(mi & below));
    InterSt &S = st[j];
    if(act) S.bi_slot = k;
    const int idx = S.mvpi[M_BI][l];
    for(int r = 0; r < P.nb; r++) {
        xeve_hip_epzs_job e;
        e.x = act ? J.x : -1, e.y = J.y, e.org_off = j * P.n0, e.mvp[0] = J.mvp[l][idx][0], e.mvp[1] = J.mvp[l][idx][1];
        e.mv_start[0] = S.mv_scale[l][r][0], e.mv_start[1] = S.mv_scale[l][r][1];
        const size_t at = (size_t)r * P.n + k;
        ej[at] = e, extra[at] = S.mot_bits[1 - l], job_plane[at] = act ? (unsigned char)(l * P.np + r) : 0;
    }
}

// get_org_bi (:143-156): 2 * org - pred, the prediction taken from where the interpolation left it (list 0's buffer, or list 1's when the fixed list is list 1)
__global__ void k_bi_org(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const pel *__restrict__ org, const pel *__restrict__ pred0, const pel *__restrict__ pred1,
                         const uint8_t *__restrict__ mode, int16_t *__restrict__ org_bi)
{
    const int nbx = P.n0 >= 1024 ? 4 : 1, j = blockIdx.x / nbx, bx = blockIdx.x % nbx, w = 1 << P.lw;
    const xeve_hip_inter_job J = jobs[j];
    const pel *pred = mode[j] == 2 ? pred1 : pred0;
    for(int i = bx * blockDim.x + threadIdx.x; i < P.n0; i += nbx * blockDim.x) {
        const int yy = i >> P.lw, xx = i & (w - 1);
        org_bi[(size_t)j * P.n0 + i] = (int16_t)((org[(size_t)(J.y + yy) * P.s_org_l + J.x + xx] << 1) - pred[(size_t)j * P.n0 + i]);
    }
}

// after the last round: its results taken in, then the candidate of pinter_residue_rdo
__global__ void k_bi_tail(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const xeve_hip_me_result *__restrict__ mres, InterSt *__restrict__ st,
                          xeve_hip_rdo_job *__restrict__ rj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.n) return;
    const xeve_hip_inter_job J = jobs[j];
    InterSt &S = st[j];
    bi_update(P, mres, S);
    for(int l = 0; l < 2; l++)
        for(int d = 0; d < 2; d++) S.mvd[M_BI][l][d] = (int16_t)(S.mv[M_BI][l][d] - J.mvp[l][S.mvpi[M_BI][l]][d]);
    rdo_job(rj[j], J, S, M_BI, 0);
}

// ---- the decision (:1872-2001) and the winner's data ---------------------------------------------------------------------------------
__global__ void k_inter_decide(const xeve_hip_inter_job *__restrict__ jobs, InterK P, const InterSt *__restrict__ st, const xeve_hip_skip_result *__restrict__ sres,
                               const xeve_hip_rdo_result *__restrict__ ra, const xeve_hip_rdo_result *__restrict__ rb, xeve_hip_inter_result *__restrict__ res,
                               int *__restrict__ win, CuMcPrep C, xeve_hip_job *__restrict__ wl, xeve_hip_job *__restrict__ wc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.n) return;
    wl[j] = xh_make_job(jobs[j].y, P.s_org_l, jobs[j].x, j * P.n0); // (the winner's blocks for the reconstruction: original and dense prediction, as k_rdo_prep's)
    wc[j] = xh_make_job(jobs[j].y >> P.hs, P.s_org_c, jobs[j].x >> P.ws, j * P.n1);
    const InterSt &S = st[j];
    double ci[M_NUM];
    int    nz[M_NUM][3];
    for(int m = 0; m < M_NUM; m++) ci[m] = MAX_COST, nz[m][0] = nz[m][1] = nz[m][2] = 0;
    ci[M_SKIP] = sres[j].cost;
    if(S.go) {
        auto take = [&](int m, const xeve_hip_rdo_result &r) { ci[m] = r.cost, nz[m][0] = r.nnz[0], nz[m][1] = r.nnz[1], nz[m][2] = r.nnz[2]; };
        if(P.isb) take(M_DIR, ra[j]), take(M_L0, ra[P.n + j]), take(M_L1, ra[2 * (size_t)P.n + j]), take(M_BI, rb[j]);
        else take(M_L0, ra[j]);
    }
    double cost_best = MAX_COST;
    int    best = M_SKIP, cu_mode = -1;
    const int order[5] = {M_SKIP, M_DIR, M_L0, M_L1, M_BI};
    for(int k = 0; k < 5; k++) {
        const int m = order[k];
        if(ci[m] < cost_best) cost_best = ci[m], best = m, cu_mode = m == M_SKIP ? 2 : m == M_DIR ? 3 : 1;
    }
    xeve_hip_inter_result R;
    for(int i = 0; i < (int)(sizeof(R) / 4); i++) ((int *)&R)[i] = 0;
    R.cost = ci[best];
    for(int m = 0; m < M_NUM; m++) R.cost_inter[m] = ci[m];
    R.cu_mode = cu_mode, R.best_idx = best;
    for(int l = 0; l < 2; l++) {
        const bool lst = P.isb || l == 0, used = lst && S.refi[best][l] >= 0;
        R.refi[l] = lst ? S.refi[best][l] : -1;
        if(used) R.mv[l][0] = S.mv[best][l][0], R.mv[l][1] = S.mv[best][l][1], R.mvd[l][0] = S.mvd[best][l][0], R.mvd[l][1] = S.mvd[best][l][1], R.mvp_idx[l] = S.mvpi[best][l];
    }
    if(best == M_DIR) R.mvp_idx[0] = R.mvp_idx[1] = 0;
    R.nnz[0] = nz[best][0], R.nnz[1] = nz[best][1], R.nnz[2] = nz[best][2];
    res[j] = R;
    win[j] = best;
    xeve_hip_cu_mc_job m; // the winner's prediction (a skipped CU keeps the one the skip analysis produced)
    m.x = jobs[j].x, m.y = jobs[j].y, m.pad_[0] = m.pad_[1] = 0;
    for(int l = 0; l < 2; l++) m.mv[l][0] = S.mv[best][l][0], m.mv[l][1] = S.mv[best][l][1], m.refi[l] = best == M_SKIP ? -1 : S.refi[best][l];
    xh_cu_mc_prep_one(m, j, C); // (the prediction's per-list interpolation jobs: mc_cu.h)
}

// per (CU, component): the winner's coefficients out (zero for a skipped CU); the skip prediction where the CU is skipped; and, by the luma block's threads,
// core->s_next_best: the coder state the winning mode's evaluation left (SBAC_STORE(core->s_next_best, core->s_temp_best), :1888 / :1963 / :1997) -- the skip analysis and
// both pinter_residue_rdo batches hand out core->s_temp_best per candidate, the winner's is copied (round 6: was the winner's syntax once more through the coder, three
// launches and a CU's worth of bins at the end of every node's chain)
__global__ void k_inter_out(InterK P, const int *__restrict__ win, const xeve_hip_inter_result *__restrict__ res, const int16_t *__restrict__ coef_a,
                            const int16_t *__restrict__ coef_b, int16_t *__restrict__ coef_out, pel *__restrict__ pred_y,
                            pel *__restrict__ pred_u, pel *__restrict__ pred_v, const pel *__restrict__ sk_y, const pel *__restrict__ sk_u, const pel *__restrict__ sk_v,
                            const xeve_hip_sbac *__restrict__ st_s, const xeve_hip_sbac *__restrict__ st_a, const xeve_hip_sbac *__restrict__ st_b,
                            xeve_hip_sbac *__restrict__ next_best)
{
    const int j = blockIdx.x / 3, k = blockIdx.x % 3, best = win[j];
    if(k == 0 && res[j].cu_mode >= 0) {
        const xeve_hip_sbac *src = best == M_SKIP ? st_s + j : best == M_BI ? st_b + j : st_a + (P.isb ? (size_t)(best == M_DIR ? 0 : best == M_L0 ? 1 : 2) * P.n + j : j);
        const unsigned *a = (const unsigned *)src;
        unsigned       *b = (unsigned *)(next_best + j);
        for(int i = threadIdx.x; i < (int)(sizeof(xeve_hip_sbac) / 4); i += blockDim.x) b[i] = a[i];
    }
    if(k && P.ncomp == 1) return;
    const int    nk = k ? P.n1 : P.n0;
    const size_t n = P.n;
    // where the winner's block lives: batch A holds [DIR, L0, L1] (B slices) or [L0] (P slices), batch B the bi candidate
    const int      slot = best == M_BI ? j : (P.isb ? (best == M_DIR ? 0 : best == M_L0 ? 1 : 2) * P.n + j : j);
    const size_t   nn = best == M_BI ? n : (size_t)P.na;
    const int16_t *src = (best == M_BI ? coef_b : coef_a) + (k == 0 ? (size_t)slot * P.n0 : nn * P.n0 + (size_t)(k - 1) * nn * P.n1 + (size_t)slot * P.n1);
    int16_t       *dst = coef_out + (k == 0 ? (size_t)j * P.n0 : n * P.n0 + (size_t)(k - 1) * n * P.n1 + (size_t)j * P.n1);
    const bool     skip = best == M_SKIP;
    const bool     none = skip || res[j].nnz[k] == 0; // (a component the decision dropped: the batch's block still holds its levels, xh_residue_rdo_jobs_x keep_dropped)
    for(int i = threadIdx.x; i < nk; i += blockDim.x) dst[i] = none ? (int16_t)0 : src[i];
    if(skip) {
        const pel *s = (k == 0 ? sk_y : k == 1 ? sk_u : sk_v) + (size_t)j * nk;
        pel       *d = (k == 0 ? pred_y : k == 1 ? pred_u : pred_v) + (size_t)j * nk;
        for(int i = threadIdx.x; i < nk; i += blockDim.x) d[i] = s[i];
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------------------
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t max2(size_t a, size_t b) { return a > b ? a : b; }

struct InterLayout {
    size_t st, sj, sres, sk[3], st_s, st_a, st_b, est, ej, mres, bjm, bitsm, rja, rra, coef_a, rjb, rrb, coef_b, mc, pred[3], org_bi, extra, job_plane, cnt, win, wl, wc, wssd, wnnz,
        scratch, scratch_bytes, total;
};

static InterLayout inter_layout(int n, int nstates, const xeve_hip_inter_params *p, int s_org_l, int s_org_c)
{
    InterLayout L;
    const xeve_hip_rdo_params rp = p->rdo;
    const int    isb = rp.slice_type == 0, idc = rp.chroma_format_idc, ws = idc <= 2, hs = idc <= 1;
    const size_t N = n, n0 = (size_t)1 << (rp.log2_cuw + rp.log2_cuh), n1 = idc ? n0 >> (ws + hs) : 0, na = isb ? 3 * N : N, ne = n0 + 2 * n1;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    L.st = take(N * sizeof(InterSt)), L.sj = take(N * sizeof(xeve_hip_skip_job)), L.sres = take(N * sizeof(xeve_hip_skip_result));
    L.sk[0] = take(N * n0 * 2), L.sk[1] = take(N * n1 * 2 + 8), L.sk[2] = take(N * n1 * 2 + 8);
    L.st_s = take(N * sizeof(xeve_hip_sbac)), L.st_a = take(na * sizeof(xeve_hip_sbac)), L.st_b = take(N * sizeof(xeve_hip_sbac));
    L.est = take((size_t)nstates * sizeof(xeve_hip_rdoq_est_full));
    L.ej = take(2 * MAXR * N * sizeof(xeve_hip_epzs_job)), L.mres = take(2 * MAXR * N * sizeof(xeve_hip_me_result));
    L.bjm = take(10 * N * sizeof(xeve_hip_cu_bits_job)), L.bitsm = take(10 * N * 4);
    L.rja = take(na * sizeof(xeve_hip_rdo_job)), L.rra = take(na * sizeof(xeve_hip_rdo_result)), L.coef_a = take(na * ne * 2);
    L.rjb = take(N * sizeof(xeve_hip_rdo_job)), L.rrb = take(N * sizeof(xeve_hip_rdo_result)), L.coef_b = take(N * ne * 2);
    L.mc = take(N * sizeof(xeve_hip_cu_mc_job));
    L.pred[0] = take(N * n0 * 2), L.pred[1] = take(N * n1 * 2 + 8), L.pred[2] = take(N * n1 * 2 + 8);
    L.org_bi = take(N * n0 * 2), L.extra = take(2 * MAXR * N * 4), L.job_plane = take(MAXR * N), L.cnt = take(256), L.win = take(N * 4);
    L.wl = take(N * sizeof(xeve_hip_job)), L.wc = take(N * sizeof(xeve_hip_job)), L.wssd = take(N * 16), L.wnnz = take(N * 4);
    // the building blocks run one after the other on the stream: one scratch region, as large as the hungriest
    size_t s = xeve_hip_analyze_skip_workspace(n, &rp, p->max_cand);
    s = max2(s, xeve_hip_me_epzs_workspace(2 * (rp.num_refp[0] > rp.num_refp[1] ? rp.num_refp[0] : rp.num_refp[1]) * n));
    s = max2(s, xeve_hip_cu_bits_workspace(10 * n, N * ne));
    s = max2(s, xeve_hip_residue_rdo_workspace((int)na, nstates, &rp, s_org_l, s_org_c));
    s = max2(s, xeve_hip_mc_cu_workspace(n, 1 << rp.log2_cuw, 1 << rp.log2_cuh, rp.num_refp[0], rp.num_refp[1]));
    L.scratch = take(s), L.scratch_bytes = s;
    L.total = o;
    return L;
}

static bool inter_params_ok(const xeve_hip_inter_params *p)
{
    const xeve_hip_rdo_params &r = p->rdo;
    return r.log2_cuw == r.log2_cuh && r.log2_cuw >= 3 && r.log2_cuw <= 6 && (r.slice_type == 0 || r.slice_type == 1) && r.tool_iqt == 0 &&
           (r.chroma_format_idc == 0 || r.chroma_format_idc == 1 || r.chroma_format_idc == 3) && r.num_refp[0] >= 1 && r.num_refp[0] <= MAXR &&
           (r.slice_type == 1 || (r.num_refp[1] >= 1 && r.num_refp[1] <= r.num_refp[0])) && p->max_cand >= 1 && p->max_cand <= 4;
}

extern "C" size_t xeve_hip_pinter_analyze_cu_workspace(int njobs, int nstates, const xeve_hip_inter_params *p, int s_org_l, int s_org_c)
{
    if(!p || njobs <= 0 || nstates <= 0 || !inter_params_ok(p)) return 256;
    return inter_layout(njobs, nstates, p, s_org_l, s_org_c).total;
}

extern "C" int xeve_hip_pinter_analyze_cu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                                               const xeve_hip_sbac *states, int nstates, const xeve_hip_inter_params *p, const xeve_hip_inter_job *jobs,
                                               int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_inter_result *results,
                                               int16_t *coef, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y,
                                               xeve_hip_sbac *next_best, void *workspace, size_t workspace_bytes, void *stream)
{
    return xh_pinter_analyze_cu_jobs_x(org, s_org_l, s_org_c, refp, s_l, s_c, states, nstates, p, const_cast<xeve_hip_inter_job *>(jobs), njobs, coef_l, coef_c, results, coef,
                                       rec_y, rec_u, rec_v, pred_y, next_best, workspace, workspace_bytes, stream, nullptr, nullptr);
}

// cand: the jobs' candidates are derived here, from the encoder's per-unit maps, by the first kernel (jobs[] is then written); est_shared: core->rdoq_est_* of every entry
// state, made by the caller (xeve_hip_rdoq_bit_est over `states`) -- NULL: made here, once for both pinter_residue_rdo batches
int xh_pinter_analyze_cu_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, const xeve_hip_sbac *states,
                                int nstates, const xeve_hip_inter_params *p, xeve_hip_inter_job *jobs, int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4],
                                xeve_hip_inter_result *results, int16_t *coef, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y,
                                xeve_hip_sbac *next_best, void *workspace, size_t workspace_bytes, void *stream, const XhInterCand *cand, const void *est_shared)
{
    XH_ENTER();
    XH_REQUIRE(p && njobs >= 0 && inter_params_ok(p));
    if(njobs == 0) return XEVE_HIP_OK;
    XH_REQUIRE(org && refp && states && nstates > 0 && jobs && results && coef && rec_y && next_best && workspace && coef_l);
    const xeve_hip_rdo_params rp = p->rdo;
    const int idc = rp.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = rp.bit_depth, lw = rp.log2_cuw, w = 1 << lw;
    XH_REQUIRE(org[0] && (idc == 0 || (org[1] && org[2] && coef_c && rec_u && rec_v)));
    const InterLayout L = inter_layout(njobs, nstates, p, s_org_l, s_org_c);
    XH_REQUIRE(workspace_bytes >= L.total);
    InterK P;
    P.n = njobs, P.isb = rp.slice_type == 0, P.lw = lw, P.n0 = w * w, P.n1 = idc ? P.n0 >> (ws + hs) : 0, P.ncomp = idc ? 3 : 1, P.bd = bd;
    P.max_cand = p->max_cand, P.nref[0] = rp.num_refp[0], P.nref[1] = P.isb ? rp.num_refp[1] : 0, P.nb = rp.num_refp[1], P.na = P.isb ? 3 * njobs : njobs, P.np = rp.num_refp[0] > P.nref[1] ? rp.num_refp[0] : P.nref[1];
    P.s_org_l = s_org_l, P.s_org_c = s_org_c, P.ws = ws, P.hs = hs;
    P.dpoc_co = refp[0 * 2 + 1].poc - p->col_list_poc0, P.dpoc_l0 = p->poc - refp[0 * 2 + 0].poc, P.dpoc_l1 = refp[0 * 2 + 1].poc - p->poc; // xeve_util.c:634-636
    P.thr = (double)((int64_t)1 << (2 * lw + 2 * (bd - 8))) * p->skip_th, P.lambda0 = rp.lambda[0];
    char *W = (char *)workspace;
    auto *st = (InterSt *)(W + L.st);
    auto *sj = (xeve_hip_skip_job *)(W + L.sj);
    auto *sres = (xeve_hip_skip_result *)(W + L.sres);
    pel  *sk[3] = {(pel *)(W + L.sk[0]), (pel *)(W + L.sk[1]), (pel *)(W + L.sk[2])}, *pred[3] = {pred_y ? pred_y : (pel *)(W + L.pred[0]), (pel *)(W + L.pred[1]), (pel *)(W + L.pred[2])}; // (mi->pred_y_best, :2040: the winner's luma prediction is the last thing written there)
    auto *st_s = (xeve_hip_sbac *)(W + L.st_s), *st_a = (xeve_hip_sbac *)(W + L.st_a), *st_b = (xeve_hip_sbac *)(W + L.st_b);
    auto *ej = (xeve_hip_epzs_job *)(W + L.ej);
    auto *mres = (xeve_hip_me_result *)(W + L.mres);
    auto *bjm = (xeve_hip_cu_bits_job *)(W + L.bjm);
    auto *bitsm = (unsigned *)(W + L.bitsm);
    auto *rja = (xeve_hip_rdo_job *)(W + L.rja), *rjb = (xeve_hip_rdo_job *)(W + L.rjb);
    auto *rra = (xeve_hip_rdo_result *)(W + L.rra), *rrb = (xeve_hip_rdo_result *)(W + L.rrb);
    auto *coef_a = (int16_t *)(W + L.coef_a), *coef_b = (int16_t *)(W + L.coef_b), *org_bi = (int16_t *)(W + L.org_bi);
    auto *extra = (int32_t *)(W + L.extra), *win = (int32_t *)(W + L.win);
    auto *wl = (xeve_hip_job *)(W + L.wl), *wc = (xeve_hip_job *)(W + L.wc);
    auto *cnt = (int32_t *)(W + L.cnt);
    auto *job_plane = (unsigned char *)(W + L.job_plane);
    void *scr = W + L.scratch;
    hipStream_t s = (hipStream_t)stream;
    const int G = (njobs + 255) / 256;
    int rc;

    // skip mode
    XhInterCand C0;
    memset(&C0, 0, sizeof(C0));
    k_inter_skip_jobs<<<G, 256, 0, s>>>(jobs, P, sj, cand ? *cand : C0);
    if(!est_shared) { // the estimates of every entry state (xeve_mode.c:792), once for both batches
        rc = xeve_hip_rdoq_bit_est(states, nstates, (xeve_hip_rdoq_est_full *)(W + L.est), stream);
        if(rc != XEVE_HIP_OK) return rc;
        est_shared = W + L.est;
    }
    rc = xeve_hip_analyze_skip_jobs(org, s_org_l, s_org_c, refp, s_l, s_c, states, nstates, &rp, sj, njobs, p->max_cand, coef_l, coef_c, sres, sk[0], sk[1], sk[2],
                                    st_s, scr, L.scratch_bytes, stream);
    if(rc != XEVE_HIP_OK) return rc;
    k_inter_stage1<<<G, 256, 0, s>>>(jobs, P, sres, st, rja, ej, cnt);
    // motion search per list and reference picture (:1906-1950)
    // ONE launch chain over every (list, reference picture): the job arrays are laid out [list][plane][CU], the plane supplies the picture
    xeve_hip_epzs_params ep = p->me;
    XhSearchPlanes pl;
    pl.per_plane = njobs, pl.job_plane = nullptr;
    auto planes_for = [&](int nlists, int bi) {
        pl.n = nlists * P.np;
        for(int l = 0; l < nlists; l++)
            for(int r = 0; r < P.np; r++) {
                const int q = l * P.np + r, rr = r < rp.num_refp[l] ? r : 0;
                pl.ref[q] = refp[rr * 2 + l].y, pl.refi_bits[q] = bi ? p->refi_bits[1][rr] : p->refi_bits[l][rr], pl.range[q] = p->range_recentre[l][rr], pl.refi[q] = rr;
            }
    };
    for(int l = 0; l <= P.isb; l++)
        for(int r = 0; r < P.nref[l]; r++) XH_REQUIRE(refp[r * 2 + l].y);
    planes_for(1 + P.isb, 0);
    ep.me.bi = 0, ep.me.extra_bits = 0, ep.me.reserved = p->me.me.reserved & 1; // (bit 0: me_raster on; the plane supplies refi)
    rc = xh_me_epzs_jobs_planes(org[0], s_org_l, nullptr, nullptr, s_l, ej, pl.n * njobs, lw, lw, bd, coef_l, &ep, nullptr, mres, scr, L.scratch_bytes, stream, &pl);
    if(rc != XEVE_HIP_OK) return rc;
    // check_best_mvp, then pinter_residue_rdo of direct + L0 + L1 in one batch
    const int nl = 1 + P.isb;
    k_inter_uni_a<<<(nl * njobs + 255) / 256, 256, 0, s>>>(jobs, P, mres, st, bjm);
    xeve_hip_cu_bits_params bp;
    bp.log2_cuw = lw, bp.log2_cuh = lw, bp.slice_type = rp.slice_type, bp.num_refp[0] = rp.num_refp[0], bp.num_refp[1] = rp.num_refp[1], bp.cm_init = 0,
    bp.chroma_format_idc = idc;
    rc = xeve_hip_cu_bits_jobs(nullptr, 0, states, bjm, 5 * nl * njobs, &bp, scr, L.scratch_bytes, bitsm, nullptr, stream);
    if(rc != XEVE_HIP_OK) return rc;
    k_inter_uni_b<<<G, 256, 0, s>>>(jobs, P, bitsm, st, rja);
    rc = xh_residue_rdo_jobs_x(org, s_org_l, s_org_c, refp, s_l, s_c, states, nstates, &rp, rja, P.na, coef_l, coef_c, rra, coef_a, st_a, scr, L.scratch_bytes, stream,
                               est_shared, 1);
    if(rc != XEVE_HIP_OK) return rc;
    if(P.isb) { // analyze_bi
        CuMcPrep C; // (the prediction from the fixed list: luma alone, read by k_bi_org from whichever list's buffer holds it)
        rc = xh_mc_cu_prep_params(refp, rp.num_refp[0], rp.num_refp[1], rp.pic_w, rp.pic_h, njobs, w, w, idc, scr, L.scratch_bytes, &C);
        if(rc != XEVE_HIP_OK) return rc;
        for(int it = 0; it < 4; it++) { // BI_ITER
            k_bi_head<<<G, 256, 0, s>>>(jobs, P, st, rra, mres, it == 0, C, ej, extra, cnt + 2 * it, job_plane);
            rc = xh_mc_cu_jobs_x(refp, rp.num_refp[0], rp.num_refp[1], s_l, s_c, rp.pic_w, rp.pic_h, nullptr, njobs, w, w, bd, bd, idc, coef_l, coef_c, pred[0], pred[1],
                                 pred[2], scr, L.scratch_bytes, stream, XH_MC_PREPPED | XH_MC_LUMA_ONLY | XH_MC_NO_COMBINE);
            if(rc != XEVE_HIP_OK) return rc;
            k_bi_org<<<njobs * (P.n0 >= 1024 ? 4 : 1), P.n0 >= 256 ? 256 : 64, 0, s>>>(jobs, P, org[0], pred[0], C.p1[0], C.mode, org_bi);
            planes_for(2, 1);
            pl.job_plane = job_plane;
            ep.me.bi = 1, ep.me.extra_bits = 0;
            rc = xh_me_epzs_jobs_planes(org[0], s_org_l, (const pel *)org_bi, nullptr, s_l, ej, P.nb * njobs, lw, lw, bd, coef_l, &ep, extra, mres, scr, L.scratch_bytes, stream,
                                        &pl);
            if(rc != XEVE_HIP_OK) return rc;
        }
        k_bi_tail<<<G, 256, 0, s>>>(jobs, P, mres, st, rjb);
        rc = xh_residue_rdo_jobs_x(org, s_org_l, s_org_c, refp, s_l, s_c, states, nstates, &rp, rjb, njobs, coef_l, coef_c, rrb, coef_b, st_b, scr, L.scratch_bytes,
                                   stream, est_shared, 1);
        if(rc != XEVE_HIP_OK) return rc;
    }
    // the decision; the winner's prediction, coefficients, reconstruction (:2004-2032) and coder state
    CuMcPrep Cw;
    rc = xh_mc_cu_prep_params(refp, rp.num_refp[0], rp.num_refp[1], rp.pic_w, rp.pic_h, njobs, w, w, idc, scr, L.scratch_bytes, &Cw);
    if(rc != XEVE_HIP_OK) return rc;
    k_inter_decide<<<G, 256, 0, s>>>(jobs, P, st, sres, rra, rrb, results, win, Cw, wl, wc);
    rc = xh_mc_cu_jobs_x(refp, rp.num_refp[0], rp.num_refp[1], s_l, s_c, rp.pic_w, rp.pic_h, nullptr, njobs, w, w, bd, bd, idc, coef_l, coef_c, pred[0], pred[1], pred[2],
                         scr, L.scratch_bytes, stream, XH_MC_PREPPED);
    if(rc != XEVE_HIP_OK) return rc;
    k_inter_out<<<3 * njobs, 64, 0, s>>>(P, win, results, coef_a, coef_b, coef, pred[0], pred[1], pred[2], sk[0], sk[1], sk[2], st_s, st_a, st_b, next_best);
    // the winner's reconstruction (xeve_itdq + xeve_recon, :2004-2032): the back half of the fused residual chain, one launch per component (round 6: was dequantisation,
    // inverse transform and reconstruction as three); a component without coefficients comes out as its prediction
    static const int k_dq_scale[6] = {40, 45, 51, 57, 64, 71}; // xeve_tbl_dq_scale_b (xeve_tbl.c:237)
    pel *rec[3] = {rec_y, rec_u, rec_v};
    for(int k = 0; k < P.ncomp; k++) {
        const int lk = k ? lw - ws : lw, lhk = k ? lw - hs : lw, q = rp.qp[k];
        XH_REQUIRE(q >= 0 && q <= 51 + 6 * (bd - 8) && lk == lhk); // MAX_QUANT + the bit-depth offset; square blocks (4:2:0, 4:4:4)
        const int16_t *t = coef + (k == 0 ? 0 : (size_t)njobs * P.n0 + (size_t)(k - 1) * njobs * P.n1);
        rc = xh_residual_back(org[k], k ? s_org_c : s_org_l, pred[k], 1 << lk, k ? wc : wl, njobs, lk, lhk, bd, q, k_dq_scale[q % 6] << (q / 6), t, rec[k], -(1 << lk),
                              (int32_t *)(W + L.wnnz), (int64_t *)(W + L.wssd), s);
        if(rc != XEVE_HIP_OK) return rc;
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_inter_candidates(const uint32_t *map_scu, const uint8_t *map_tidx, const int16_t *map_mv, const int16_t *col_mv0, const int16_t *col_mv1,
                                         int w_scu, int h_scu, int log2_cuw, int log2_cuh, int slice_type, xeve_hip_inter_job *jobs, int njobs, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(njobs >= 0 && w_scu > 0 && h_scu > 0 && log2_cuw >= 2 && log2_cuw <= 7 && log2_cuh >= 2 && log2_cuh <= 7 && (slice_type == 0 || slice_type == 1));
    if(njobs == 0) return XEVE_HIP_OK;
    XH_REQUIRE(map_scu && map_mv && col_mv0 && jobs && (slice_type == 1 || col_mv1));
    const XhInterCand C = {map_scu, map_tidx, map_mv, col_mv0, col_mv1, w_scu, 1 << (log2_cuw - 2), 1 << (log2_cuh - 2), slice_type == 0, xh_vh()};
    k_inter_candidates<<<(njobs + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, jobs, njobs);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- per-thread device state of the host-memory form with resident pictures --------------------------------------------------------------
// One stream, one device arena (job | state | result | coefficients | reconstruction | prediction | exit state | workspace) and one pinned
// host mirror of the small records per encoder thread: a call is one upload of (job, state), the launches, one download of the outputs.
namespace {
// One CU's call is ~200 small dependent launches: issued one by one they cost the host 5-7 ms per CU (measured, 1280x720 and 1920x1080 encodes); the
// whole call -- upload of (job, state), every launch, download of the results -- is therefore captured ONCE per (CU size, picture parameters) into a HIP
// graph and replayed for every further CU of that size in the picture (all operands sit at fixed addresses of the per-thread arena; the planes are the
// picture's resident copies).  The key is everything the launches bake in: the parameter record, the plane table, strides, the coefficient tables.
struct HostGraph {
    std::vector<char> key;
    hipGraph_t        graph = nullptr;
    hipGraphExec_t    exec  = nullptr;
};
struct InterHostCtx {
    uint32_t    gen = 0;
    hipStream_t st  = nullptr;
    char       *dev = nullptr, *pin = nullptr;
    size_t      dev_bytes = 0;
    std::vector<HostGraph> graphs;
    void drop_graphs()
    {
        for(auto &g : graphs) {
            if(g.exec) (void)hipGraphExecDestroy(g.exec);
            if(g.graph) (void)hipGraphDestroy(g.graph);
        }
        graphs.clear();
    }
    static constexpr size_t IN_BYTES = 512, OUT_BYTES = 64 << 10; // in: job + state; out: result, exit state, coef, rec, pred (64x64: 12 + 12.1 + 8 KB)
    void release()
    {
        drop_graphs();
        if(st) (void)hipStreamDestroy(st);
        if(dev) (void)hipFree(dev);
        if(pin) (void)hipHostFree(pin);
        st = nullptr, dev = pin = nullptr, dev_bytes = 0;
    }
    int ensure(size_t ws_bytes)
    {
        if(gen != xh_generation()) release(), gen = xh_generation(); // the library was shut down or re-bound since
        if(!st) XH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if(!pin) XH_HIP(hipHostMalloc((void **)&pin, IN_BYTES + OUT_BYTES, hipHostMallocDefault));
        const size_t need = IN_BYTES + OUT_BYTES + 256 + ws_bytes;
        if(dev_bytes < need) {
            if(dev) {
                XH_HIP(hipStreamSynchronize(st));
                drop_graphs(); // (they hold the old arena's addresses)
                (void)hipFree(dev);
                dev = nullptr, dev_bytes = 0;
            }
            XH_HIP(hipMalloc((void **)&dev, need + (need >> 2)));
            dev_bytes = need + (need >> 2);
        }
        return XEVE_HIP_OK;
    }
    ~InterHostCtx() { release(); }
};
} // namespace

static int inter_host_resident(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, int pad_l, int pad_c,
                               const xeve_hip_sbac *state, const xeve_hip_inter_params *p, const xeve_hip_inter_job *job, const int16_t (*coef_l)[8],
                               const int16_t (*coef_c)[4], xeve_hip_inter_result *result, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, xeve_hip_pel *rec_y,
                               xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y, xeve_hip_sbac *next_best)
{
    static thread_local InterHostCtx C;
    const xeve_hip_rdo_params &rp = p->rdo;
    const int idc = rp.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, ncomp = idc ? 3 : 1, isb = rp.slice_type == 0;
    const size_t n0 = (size_t)1 << (2 * rp.log2_cuw), n1 = idc ? n0 >> (ws + hs) : 0;
    const size_t eo[3] = {(size_t)s_org_l * rp.pic_h, (size_t)s_org_c * (rp.pic_h >> hs), (size_t)s_org_c * (rp.pic_h >> hs)};
    const size_t er[3] = {(size_t)s_l * (rp.pic_h + 2 * pad_l), (size_t)s_c * ((rp.pic_h >> hs) + 2 * pad_c), (size_t)s_c * ((rp.pic_h >> hs) + 2 * pad_c)};
    const size_t orr[3] = {(size_t)pad_l * s_l + pad_l, (size_t)pad_c * s_c + pad_c, (size_t)pad_c * s_c + pad_c};
    const size_t wsb = xeve_hip_pinter_analyze_cu_workspace(1, 1, p, s_org_l, s_org_c);
    int rc = C.ensure(wsb);
    if(rc != XEVE_HIP_OK) return rc;
    // planes: the picture's resident copies (uploaded on first sight)
    const pel *dorg[3] = {nullptr, nullptr, nullptr};
    for(int c = 0; c < ncomp; c++) {
        dorg[c] = (const pel *)xh_resident(org[c], eo[c] * sizeof(pel));
        if(!dorg[c]) return XEVE_HIP_ERR_DEVICE;
    }
    xeve_hip_refpic tab[2 * MAXR];
    memset(tab, 0, sizeof(tab));
    const int nr[2] = {rp.num_refp[0], isb ? rp.num_refp[1] : 0};
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            const xeve_hip_refpic &e = refp[r * 2 + l];
            XH_REQUIRE(e.y && (idc == 0 || (e.u && e.v)));
            const xeve_hip_pel *hp[3] = {e.y, e.u, e.v};
            const pel *dp[3] = {nullptr, nullptr, nullptr};
            for(int c = 0; c < ncomp; c++) {
                const pel *d = (const pel *)xh_resident(hp[c] - orr[c], er[c] * sizeof(pel));
                if(!d) return XEVE_HIP_ERR_DEVICE;
                dp[c] = d + orr[c];
            }
            xeve_hip_refpic &t = tab[r * 2 + l];
            t.y = dp[0], t.u = dp[1], t.v = dp[2], t.poc = e.poc;
        }
    if(!isb) tab[0 * 2 + 1] = tab[0 * 2 + 0]; // (P slices never read list 1; keep the table addressable)
    // arena layout
    char *d = C.dev, *h = C.pin;
    const size_t o_job = 0, o_state = 256;                                                                       // in
    const size_t o_res = InterHostCtx::IN_BYTES, o_nb = o_res + 256, o_coef = o_nb + 256, o_rec = o_coef + (((n0 + 2 * n1) * 2 + 255) & ~(size_t)255),
                 o_pred = o_rec + (((n0 + 2 * n1 + 16) * 2 + 255) & ~(size_t)255), o_end = o_pred + n0 * 2;     // out
    XH_REQUIRE(o_end <= InterHostCtx::IN_BYTES + InterHostCtx::OUT_BYTES);
    const size_t o_ws = (InterHostCtx::IN_BYTES + InterHostCtx::OUT_BYTES + 255) & ~(size_t)255;
    xeve_hip_inter_job j0 = *job;
    j0.sbac = 0;
    memcpy(h + o_job, &j0, sizeof(j0)), memcpy(h + o_state, state, sizeof(*state));
    pel *drec = (pel *)(d + o_rec);
    auto enqueue = [&]() -> int { // the whole call on C.st
        XH_HIP(hipMemcpyAsync(d, h, InterHostCtx::IN_BYTES, hipMemcpyHostToDevice, C.st));
        int r = xeve_hip_pinter_analyze_cu_jobs(dorg, s_org_l, s_org_c, tab, s_l, s_c, (const xeve_hip_sbac *)(d + o_state), 1, p, (const xeve_hip_inter_job *)(d + o_job), 1,
                                                coef_l, coef_c, (xeve_hip_inter_result *)(d + o_res), (int16_t *)(d + o_coef), drec, drec + n0 + 8, drec + n0 + n1 + 16,
                                                (pel *)(d + o_pred), (xeve_hip_sbac *)(d + o_nb), d + o_ws, wsb, C.st);
        if(r != XEVE_HIP_OK) return r;
        XH_HIP(hipMemcpyAsync(h + o_res, d + o_res, o_end - o_res, hipMemcpyDeviceToHost, C.st));
        return XEVE_HIP_OK;
    };
    static const int use_graph = getenv("XEVE_HIP_HOST_GRAPH") ? atoi(getenv("XEVE_HIP_HOST_GRAPH")) : 1; // developer switch (measurement)
    if(use_graph && !xh_prof_on(XH_PROF_SEARCH) && !xh_prof_on(XH_PROF_CU_BITS)) {
        std::vector<char> key(sizeof(*p) + sizeof(tab) + sizeof(dorg) + 4 * sizeof(int) + 2 * sizeof(void *));
        char *k = key.data();
        memcpy(k, p, sizeof(*p)), k += sizeof(*p);
        memcpy(k, tab, sizeof(tab)), k += sizeof(tab);
        memcpy(k, dorg, sizeof(dorg)), k += sizeof(dorg);
        const int strides[4] = {s_org_l, s_org_c, s_l, s_c};
        memcpy(k, strides, sizeof(strides)), k += sizeof(strides);
        const void *ct[2] = {(const void *)coef_l, (const void *)coef_c};
        memcpy(k, ct, sizeof(ct));
        HostGraph *g = nullptr;
        for(auto &c : C.graphs)
            if(c.key == key) {
                g = &c;
                break;
            }
        if(!g) {
            if(C.graphs.size() >= 64) C.drop_graphs();
            HostGraph ng;
            XH_HIP(hipStreamBeginCapture(C.st, hipStreamCaptureModeThreadLocal));
            rc = enqueue();
            const hipError_t ec = hipStreamEndCapture(C.st, &ng.graph);
            if(rc != XEVE_HIP_OK) {
                if(ng.graph) (void)hipGraphDestroy(ng.graph);
                return rc;
            }
            XH_HIP(ec);
            XH_HIP(hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0));
            ng.key = key;
            C.graphs.push_back(std::move(ng));
            g = &C.graphs.back();
        }
        XH_HIP(hipGraphLaunch(g->exec, C.st));
    }
    else {
        rc = enqueue();
        if(rc != XEVE_HIP_OK) return rc;
    }
    XH_HIP(hipStreamSynchronize(C.st));
    memcpy(result, h + o_res, sizeof(*result)), memcpy(next_best, h + o_nb, sizeof(*next_best));
    memcpy(coef_y, h + o_coef, n0 * 2), memcpy(rec_y, h + o_rec, n0 * 2);
    if(pred_y) memcpy(pred_y, h + o_pred, n0 * 2);
    if(idc) {
        memcpy(coef_u, h + o_coef + n0 * 2, n1 * 2), memcpy(coef_v, h + o_coef + (n0 + n1) * 2, n1 * 2);
        memcpy(rec_u, h + o_rec + (n0 + 8) * 2, n1 * 2), memcpy(rec_v, h + o_rec + (n0 + n1 + 16) * 2, n1 * 2);
    }
    return XEVE_HIP_OK;
}

// ---- host-memory form of one xeve_pinter_analyze_cu call (the table layer's style: synchronous, every plane staged per call) ----------
// What ctx->fn_pinter_analyze_cu can be pointed at (tests/test_integration_ref.py does, through shim/xeve_hip_shim.c).  org / refp: HOST pointers to
// sample (0, 0); the reference planes extend pad_l / pad_c samples around the picture.
extern "C" int xeve_hip_pinter_analyze_cu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, int pad_l,
                                               int pad_c, const xeve_hip_sbac *state, const xeve_hip_inter_params *p, const xeve_hip_inter_job *job,
                                               const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_inter_result *result, int16_t *coef_y,
                                               int16_t *coef_u, int16_t *coef_v, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v,
                                               xeve_hip_pel *pred_y, xeve_hip_sbac *next_best)
{
    XH_ENTER();
    XH_REQUIRE(org && org[0] && refp && state && p && job && result && coef_y && rec_y && next_best && inter_params_ok(p) && pad_l >= 0 && pad_c >= 0);
    const xeve_hip_rdo_params &rp = p->rdo;
    const int idc = rp.chroma_format_idc, ws = idc <= 2, hs = idc <= 1, ncomp = idc ? 3 : 1, isb = rp.slice_type == 0;
    XH_REQUIRE(idc == 0 || (org[1] && org[2] && coef_u && coef_v && rec_u && rec_v));
    if(xh_resident_on()) // the caller announces its pictures: planes are resident, the call moves a job and a CU's worth of results
        return inter_host_resident(org, s_org_l, s_org_c, refp, s_l, s_c, pad_l, pad_c, state, p, job, coef_l, coef_c, result, coef_y, coef_u, coef_v, rec_y, rec_u, rec_v,
                                   pred_y, next_best);
    const size_t n0 = (size_t)1 << (2 * rp.log2_cuw), n1 = idc ? n0 >> (ws + hs) : 0;
    const size_t eo[3] = {(size_t)s_org_l * rp.pic_h, (size_t)s_org_c * (rp.pic_h >> hs), (size_t)s_org_c * (rp.pic_h >> hs)};
    const size_t er[3] = {(size_t)s_l * (rp.pic_h + 2 * pad_l), (size_t)s_c * ((rp.pic_h >> hs) + 2 * pad_c), (size_t)s_c * ((rp.pic_h >> hs) + 2 * pad_c)};
    const size_t orr[3] = {(size_t)pad_l * s_l + pad_l, (size_t)pad_c * s_c + pad_c, (size_t)pad_c * s_c + pad_c};
    // the pictures both lists hold (a picture may sit in both: staged once)
    const int nr[2] = {rp.num_refp[0], isb ? rp.num_refp[1] : 0};
    const xeve_hip_pel *uniq[2 * MAXR];
    int nu = 0, which[2 * MAXR];
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            const xeve_hip_refpic &e = refp[r * 2 + l];
            XH_REQUIRE(e.y && (idc == 0 || (e.u && e.v)));
            int k = 0;
            while(k < nu && uniq[k] != e.y) k++;
            if(k == nu) uniq[nu++] = e.y;
            which[r * 2 + l] = k;
        }
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    size_t o_org[3], o_ref[2 * MAXR][3];
    for(int c = 0; c < ncomp; c++) o_org[c] = take(eo[c] * sizeof(pel));
    for(int k = 0; k < nu; k++)
        for(int c = 0; c < ncomp; c++) o_ref[k][c] = take(er[c] * sizeof(pel));
    const size_t o_state = take(sizeof(xeve_hip_sbac)), o_job = take(sizeof(xeve_hip_inter_job)), o_res = take(sizeof(xeve_hip_inter_result));
    const size_t o_coef = take((n0 + 2 * n1) * 2), o_rec = take((n0 + 2 * n1 + 16) * 2), o_pred = take(n0 * 2), o_nb = take(sizeof(xeve_hip_sbac));
    const size_t wsb = xeve_hip_pinter_analyze_cu_workspace(1, 1, p, s_org_l, s_org_c), o_ws = take(wsb);
    char *d = nullptr;
    XH_HIP(hipMalloc((void **)&d, o));
    int rc = XEVE_HIP_OK;
    auto up = [&](size_t off, const void *src, size_t bytes) { if(rc == XEVE_HIP_OK && hipMemcpy(d + off, src, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE; };
    auto down = [&](void *dst, size_t off, size_t bytes) { if(rc == XEVE_HIP_OK && hipMemcpy(dst, d + off, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE; };
    for(int c = 0; c < ncomp; c++) up(o_org[c], org[c], eo[c] * sizeof(pel));
    xeve_hip_refpic tab[2 * MAXR];
    memset(tab, 0, sizeof(tab));
    bool done[2 * MAXR] = {};
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            const xeve_hip_refpic &e = refp[r * 2 + l];
            const int k = which[r * 2 + l];
            const xeve_hip_pel *hp[3] = {e.y, e.u, e.v};
            if(!done[k])
                for(int c = 0; c < ncomp; c++) up(o_ref[k][c], hp[c] - orr[c], er[c] * sizeof(pel));
            done[k] = true;
            xeve_hip_refpic &t = tab[r * 2 + l];
            t.y = (pel *)(d + o_ref[k][0]) + orr[0], t.poc = e.poc;
            if(idc) t.u = (pel *)(d + o_ref[k][1]) + orr[1], t.v = (pel *)(d + o_ref[k][2]) + orr[2];
        }
    if(!isb) tab[0 * 2 + 1] = tab[0 * 2 + 0]; // (P slices never read list 1; keep the table addressable)
    xeve_hip_inter_job j0 = *job;
    j0.sbac = 0;
    up(o_state, state, sizeof(*state)), up(o_job, &j0, sizeof(j0));
    if(rc != XEVE_HIP_OK) xh_set_error("xeve_hip_pinter_analyze_cu_host: staging failed");
    if(rc == XEVE_HIP_OK) {
        const pel *dorg[3] = {(pel *)(d + o_org[0]), idc ? (pel *)(d + o_org[1]) : nullptr, idc ? (pel *)(d + o_org[2]) : nullptr};
        pel *drec = (pel *)(d + o_rec);
        rc = xeve_hip_pinter_analyze_cu_jobs(dorg, s_org_l, s_org_c, tab, s_l, s_c, (const xeve_hip_sbac *)(d + o_state), 1, p, (const xeve_hip_inter_job *)(d + o_job), 1,
                                             coef_l, coef_c, (xeve_hip_inter_result *)(d + o_res), (int16_t *)(d + o_coef), drec, drec + n0 + 8, drec + n0 + n1 + 16,
                                             (pel *)(d + o_pred), (xeve_hip_sbac *)(d + o_nb), d + o_ws, wsb, nullptr);
        if(rc == XEVE_HIP_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE, xh_set_error("xeve_hip_pinter_analyze_cu_host: device error");
        if(rc == XEVE_HIP_OK) {
            down(result, o_res, sizeof(*result)), down(coef_y, o_coef, n0 * 2), down(rec_y, o_rec, n0 * 2), down(next_best, o_nb, sizeof(*next_best));
            if(pred_y) down(pred_y, o_pred, n0 * 2);
            if(idc) {
                down(coef_u, o_coef + n0 * 2, n1 * 2), down(coef_v, o_coef + (n0 + n1) * 2, n1 * 2);
                down(rec_u, o_rec + (n0 + 8) * 2, n1 * 2), down(rec_v, o_rec + (n0 + n1 + 16) * 2, n1 * 2);
            }
            if(rc != XEVE_HIP_OK) xh_set_error("xeve_hip_pinter_analyze_cu_host: copy back failed");
        }
    }
    (void)hipFree(d);
    return rc;
}
