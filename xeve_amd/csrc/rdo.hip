// xeve_amd/csrc/rdo.hip -- pinter_residue_rdo for a batch of inter CU candidates, composed on the device.
//
// reference: pinter_residue_rdo (src_base/xeve_pinter.c:906-1336): prediction (xeve_mc), residual + SSD, transform + RDOQ with the
// estimates of the entry coder state (xeve_rdoq_bit_est, xeve_mode.c:792), reconstruction + SSD, and the coded-block-flag decision
// from CABAC bit counts: all-zero alternative, as quantised, every component with / without its coefficients (the coder state is
// handed from one component test to the next), the combination those tests chose.  rdo_dbk_switch = 0 (presets fast / medium).
//
// Everything heavy already exists as a batched entry point (xeve_hip_mc_cu_jobs, the fused residual chain, xeve_hip_rdoq_dev,
// xeve_hip_cu_bits_jobs); this file strings them together and adds the per-candidate decision kernels (double precision, the
// reference's expression order; compiled with -ffp-contract=off).  The decision is a dependent chain of four bit-count rounds:
//   round 1: {all-zero, as-is, Y without, Y with} from the entry state         -> best so far, Y's choice, state after Y
//   round 2: {U without, U with} from the state after Y                         -> U's choice, state after U
//   round 3: {V without, V with} from the state after U                         -> V's choice
//   round 4: {the chosen combination} from the entry state (when it differs from "as is")
// Jobs of a round run in parallel over all candidates.  Rounds 2 and 3 always use the count-only kernel (their states are only ever
// loaded into further counts: range and models suffice); rounds 1 and 4 carry the complete coder state when the caller asks for
// core->s_temp_best.
#include <cstdlib>
#include "xh_common.h"
#include "mc_cu.h"

extern "C" int xeve_hip_residual_rdoq_dev(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h,
                                          int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, double lambda, int ch_type, int tool_iqt,
                                          const xeve_hip_rdoq_est_full *est, const int32_t *est_idx, int16_t *coef, pel *rec, int s_rec, int32_t *nnz,
                                          int64_t *ssd, void *stream);

struct RdoK {
    int    n0, n1, ncomp, njobs, dir_unused;
    int    w, h, ws, hs, s_org_l, s_org_c;
    double lambda[3], wgt[2];
};

struct Cand { // per candidate, between the rounds
    double cost_best;
    int    nnz_store[3], idx_best[3], cbf_idx[3], iy, iu, iv, tnnz, round4, win; // win: 0 all-zero, 1 as quantised, 2 the combination
};

#define MAX_COST 1.7e+308

__device__ __forceinline__ void copy_state(xeve_hip_sbac *d, const xeve_hip_sbac *s)
{
    const unsigned *a = (const unsigned *)s;
    unsigned       *b = (unsigned *)d;
#pragma unroll
    for(int i = 0; i < (int)(sizeof(xeve_hip_sbac) / 4); i++) b[i] = a[i];
}

__device__ __forceinline__ void fill_bits_job(xeve_hip_cu_bits_job &b, const xeve_hip_rdo_job &J, const RdoK &P, int j, int mode, int n0, int n1, int n2, int sbac)
{
    b.coef_off[0] = j * P.n0, b.coef_off[1] = P.njobs * P.n0 + j * P.n1, b.coef_off[2] = P.njobs * (P.n0 + P.n1) + j * P.n1;
    b.nnz[0] = n0, b.nnz[1] = n1, b.nnz[2] = n2, b.sbac = sbac;
    b.mvd[0][0] = J.mvd[0][0], b.mvd[0][1] = J.mvd[0][1], b.mvd[1][0] = J.mvd[1][0], b.mvd[1][1] = J.mvd[1][1];
    b.refi[0] = J.refi[0], b.refi[1] = J.refi[1], b.mvp_idx[0] = J.mvp_idx[0], b.mvp_idx[1] = J.mvp_idx[1];
    b.mode = (uint8_t)mode, b.dir_flag = J.dir_flag, b.ctx_skip = J.ctx_skip, b.ctx_pred_mode = J.ctx_pred_mode;
}

// jobs of the building blocks: prediction, residual chain (luma / chroma), estimate record per block
__global__ void k_rdo_prep(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, CuMcPrep C, xeve_hip_job *__restrict__ rl, xeve_hip_job *__restrict__ rc,
                           int *__restrict__ est_idx)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    xeve_hip_cu_mc_job m;
    m.x = J.x, m.y = J.y, m.mv[0][0] = J.mv[0][0], m.mv[0][1] = J.mv[0][1], m.mv[1][0] = J.mv[1][0], m.mv[1][1] = J.mv[1][1];
    m.refi[0] = J.refi[0], m.refi[1] = J.refi[1], m.pad_[0] = m.pad_[1] = 0;
    xh_cu_mc_prep_one(m, j, C); // (the prediction's per-list interpolation jobs: mc_cu.h)
    rl[j] = xh_make_job(J.y, P.s_org_l, J.x, j * P.n0);
    rc[j] = xh_make_job(J.y >> P.hs, P.s_org_c, J.x >> P.ws, j * P.n1);
    est_idx[j] = J.sbac;
}

// round 1 jobs: all-zero, as quantised, Y without / with its coefficients -- all from the entry state
__global__ void k_rdo_round1(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, const int *__restrict__ nnz_y, const int *__restrict__ nnz_u,
                             const int *__restrict__ nnz_v, Cand *__restrict__ cand, xeve_hip_cu_bits_job *__restrict__ bj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    Cand c;
    c.nnz_store[0] = nnz_y[j], c.nnz_store[1] = P.ncomp > 1 ? nnz_u[j] : 0, c.nnz_store[2] = P.ncomp > 1 ? nnz_v[j] : 0;
    c.tnnz = c.nnz_store[0] + c.nnz_store[1] + c.nnz_store[2];
    c.cost_best = MAX_COST, c.idx_best[0] = c.idx_best[1] = c.idx_best[2] = 0, c.cbf_idx[0] = c.cbf_idx[1] = c.cbf_idx[2] = 0;
    c.iy = c.nnz_store[0] > 0, c.iu = c.nnz_store[1] > 0, c.iv = c.nnz_store[2] > 0, c.round4 = 0, c.win = 0;
    cand[j] = c;
    // job arrays are KIND-MAJOR (kind k of candidate j at k * njobs + j): a wave of the bit counter then holds 64 jobs of one kind -- 64 whole-CU counts or 64
    // one-bin "without" counts -- instead of a mix in which the one-bin lanes idle while the wave waits for its longest string
    const int N = P.njobs;
    fill_bits_job(bj[0 * N + j], J, P, j, XEVE_HIP_BITS_CU_INTER, 0, 0, 0, J.sbac);
    fill_bits_job(bj[1 * N + j], J, P, j, XEVE_HIP_BITS_CU_INTER, c.nnz_store[0], c.nnz_store[1], c.nnz_store[2], J.sbac);
    fill_bits_job(bj[2 * N + j], J, P, j, XEVE_HIP_BITS_COMP_Y, 0, c.nnz_store[1], c.nnz_store[2], J.sbac);
    fill_bits_job(bj[3 * N + j], J, P, j, XEVE_HIP_BITS_COMP_Y, c.nnz_store[0], c.nnz_store[1], c.nnz_store[2], J.sbac);
}

__device__ __forceinline__ double sum_cost(const long *d0, const long *d1, int iy, int iu, int iv, const RdoK &P)
{ // (double)dist[idx_y][Y] + (((double)dist[idx_u][U] * w0) + ((double)dist[idx_v][V] * w1))  (xeve_pinter.c:1112-1113)
    return (double)(iy ? d1[0] : d0[0]) + (((double)(iu ? d1[1] : d0[1]) * P.wgt[0]) + ((double)(iv ? d1[2] : d0[2]) * P.wgt[1]));
}

__device__ __forceinline__ void load_dist(const long *ssd_y, const long *ssd_u, const long *ssd_v, int j, const RdoK &P, long *d0, long *d1)
{
    d0[0] = ssd_y[2 * j], d1[0] = ssd_y[2 * j + 1];
    d0[1] = P.ncomp > 1 ? ssd_u[2 * j] : 0, d1[1] = P.ncomp > 1 ? ssd_u[2 * j + 1] : 0;
    d0[2] = P.ncomp > 1 ? ssd_v[2 * j] : 0, d1[2] = P.ncomp > 1 ? ssd_v[2 * j + 1] : 0;
}

// after round 1: best so far; Y's choice; jobs of round 2 (U without / with) from the state after Y
__global__ void k_rdo_decide1(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, const long *ssd_y, const long *ssd_u, const long *ssd_v,
                              const unsigned *__restrict__ bits, const xeve_hip_sbac *__restrict__ st_out, const xeve_hip_sbac *__restrict__ entry,
                              Cand *__restrict__ cand, xeve_hip_sbac *__restrict__ prev, xeve_hip_cu_bits_job *__restrict__ bj,
                              xeve_hip_sbac *__restrict__ best)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    Cand c = cand[j];
    long d0[3], d1[3];
    load_dist(ssd_y, ssd_u, ssd_v, j, P, d0, d1);
    const int N = P.njobs; // (kind-major arrays, see k_rdo_round1)
    const unsigned b[4] = {bits[j], bits[N + j], bits[2 * N + j], bits[3 * N + j]};
    auto so = [&](int k) { return st_out + (size_t)k * N + j; };
    if(c.tnnz == 0) { // nothing survived quantisation (xeve_pinter.c:1276-1331)
        c.cost_best = (double)d0[0] + (P.wgt[0] * (double)d0[1]) + (P.wgt[1] * (double)d0[2]);
        c.cost_best += (double)(int)b[0] * P.lambda[0];
        c.win = 0;
        if(best) copy_state(best + j, so(0));
        copy_state(prev + j, entry + J.sbac);
    }
    else {
        if(!J.dir_flag) { // all-zero alternative (:1103-1142)
            double cost = sum_cost(d0, d1, 0, 0, 0, P);
            cost += (double)(int)b[0] * P.lambda[0];
            if(cost < c.cost_best) {
                c.cost_best = cost, c.cbf_idx[0] = c.cbf_idx[1] = c.cbf_idx[2] = 0, c.win = 0;
                if(best) copy_state(best + j, so(0));
            }
        }
        { // as quantised (:1144-1178)
            double cost = sum_cost(d0, d1, c.iy, c.iu, c.iv, P);
            cost += (double)(int)b[1] * P.lambda[0];
            if(cost < c.cost_best) {
                c.cost_best = cost, c.cbf_idx[0] = c.iy, c.cbf_idx[1] = c.iu, c.cbf_idx[2] = c.iv, c.win = 1;
                if(best) copy_state(best + j, so(1));
            }
        }
        // Y with / without its coefficients (:1180-1218, i = 0)
        if(c.nnz_store[0] > 0) {
            double c0 = (double)d0[0] * 1.0, c1 = (double)d1[0] * 1.0;
            c0 += (double)(int)b[2] * P.lambda[0], c1 += (double)(int)b[3] * P.lambda[0];
            const int pick = c1 < c0; // j = 0 is taken first, j = 1 must be strictly smaller
            c.idx_best[0] = pick;
            copy_state(prev + j, so(2 + pick));
        }
        else copy_state(prev + j, entry + J.sbac);
    }
    cand[j] = c;
    fill_bits_job(bj[j], J, P, j, XEVE_HIP_BITS_COMP_U, c.nnz_store[0], 0, c.nnz_store[2], j);
    fill_bits_job(bj[N + j], J, P, j, XEVE_HIP_BITS_COMP_U, c.nnz_store[0], c.nnz_store[1], c.nnz_store[2], j);
}

// after round 2 (comp = 1) / round 3 (comp = 2): that component's choice; next round's jobs
__global__ void k_rdo_decide_comp(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, int comp, const long *ssd_y, const long *ssd_u, const long *ssd_v,
                                  const unsigned *__restrict__ bits, const xeve_hip_sbac *__restrict__ st_out, Cand *__restrict__ cand,
                                  xeve_hip_sbac *__restrict__ prev, xeve_hip_cu_bits_job *__restrict__ bj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    Cand c = cand[j];
    long d0[3], d1[3];
    load_dist(ssd_y, ssd_u, ssd_v, j, P, d0, d1);
    if(c.tnnz != 0 && c.nnz_store[comp] > 0) {
        double c0 = (double)d0[comp] * P.wgt[comp - 1], c1 = (double)d1[comp] * P.wgt[comp - 1];
        c0 += (double)(int)bits[j] * P.lambda[comp], c1 += (double)(int)bits[P.njobs + j] * P.lambda[comp];
        const int pick = c1 < c0;
        c.idx_best[comp] = pick;
        copy_state(prev + j, st_out + (size_t)pick * P.njobs + j);
    }
    if(comp == 1) {
        fill_bits_job(bj[j], J, P, j, XEVE_HIP_BITS_COMP_V, c.nnz_store[0], c.nnz_store[1], 0, j);
        fill_bits_job(bj[P.njobs + j], J, P, j, XEVE_HIP_BITS_COMP_V, c.nnz_store[0], c.nnz_store[1], c.nnz_store[2], j);
    }
    else { // the combination the component tests chose (:1220-1262)
        int n[3] = {c.nnz_store[0], c.nnz_store[1], c.nnz_store[2]};
        if(c.tnnz != 0 && (c.idx_best[0] || c.idx_best[1] || c.idx_best[2])) {
            c.iy = c.idx_best[0], c.iu = c.idx_best[1], c.iv = c.idx_best[2];
            n[0] = c.iy ? n[0] : 0, n[1] = c.iu ? n[1] : 0, n[2] = c.iv ? n[2] : 0;
        }
        c.round4 = c.tnnz != 0 && (n[0] != c.nnz_store[0] || n[1] != c.nnz_store[1] || n[2] != c.nnz_store[2]);
        fill_bits_job(bj[j], J, P, j, c.round4 ? XEVE_HIP_BITS_CU_INTER : XEVE_HIP_BITS_CU_SKIP, n[0], n[1], n[2], J.sbac);
    }
    cand[j] = c;
}

// after round 4: final comparison, results, dropped coefficient blocks zeroed (:1264-1275)
__global__ void k_rdo_finish(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, const long *ssd_y, const long *ssd_u, const long *ssd_v,
                             const unsigned *__restrict__ bits, Cand *__restrict__ cand, xeve_hip_rdo_result *__restrict__ res,
                             unsigned char *__restrict__ drop, const xeve_hip_sbac *__restrict__ st_out, xeve_hip_sbac *__restrict__ best)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    Cand c = cand[j];
    long d0[3], d1[3];
    load_dist(ssd_y, ssd_u, ssd_v, j, P, d0, d1);
    if(c.round4) {
        double cost = sum_cost(d0, d1, c.iy, c.iu, c.iv, P);
        cost += (double)(int)bits[j] * P.lambda[0];
        if(cost < c.cost_best) {
            c.cost_best = cost, c.cbf_idx[0] = c.iy, c.cbf_idx[1] = c.iu, c.cbf_idx[2] = c.iv, c.win = 2;
            if(best) copy_state(best + j, st_out + j);
        }
    }
    xeve_hip_rdo_result r;
    r.cost = c.cost_best, r.pad_ = 0;
    for(int k = 0; k < 3; k++) {
        r.nnz[k] = c.tnnz != 0 && c.cbf_idx[k] ? c.nnz_store[k] : 0;
        drop[3 * j + k] = r.nnz[k] == 0 && c.nnz_store[k] != 0;
        r.dist[0][k] = d0[k], r.dist[1][k] = c.tnnz != 0 ? d1[k] : 0;
    }
    res[j] = r;
}

__global__ void k_rdo_zero_dropped(int16_t *__restrict__ coef, const unsigned char *__restrict__ drop, RdoK P)
{ // one wave per (candidate, component), four to a workgroup (most have nothing to do: keep the launch small)
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if(item >= 3 * P.njobs) return;
    const int j = item / 3, k = item % 3;
    if(!drop[3 * j + k]) return;
    const int n = k ? P.n1 : P.n0;
    int16_t *b = coef + (k == 0 ? (size_t)j * P.n0 : (size_t)P.njobs * P.n0 + (size_t)(k - 1) * P.njobs * P.n1 + (size_t)j * P.n1);
    for(int i = lane; i < n; i += 64) b[i] = 0;
}

// ---- the same decision in ONE bit-count round (batches small enough to be latency-bound) ----------------------------------------------
// The four rounds above are a dependent chain: on a level with few, large CUs (64x64: 2 040 per 4K picture) each round is a handful of waves waiting
// for their longest bin string, and the chain is the critical path of the whole level.  Every count the chain can ever ask for is known up front,
// though: the all-zero and as-quantised CU, the per-component tests under each of the four possible (Y choice, U choice) outcomes (k_cu_bits_chain: one
// lane per outcome), and the whole CU under each of the six combinations of kept components the tests can choose (CU_INTER jobs).  They are all
// launched together -- 8 whole-CU jobs + 4 chain lanes per candidate, against at most 8 jobs in four rounds -- and one kernel then walks the reference's
// decision over the counts.  More bins in total (the large levels have the lanes to spare), a third of the serial length.
// whole-CU jobs: kind k = (keep Y) | (keep U) << 1 | (keep V) << 2 of candidate j at k * njobs + j; kind 0 = all-zero, kind 7 = as quantised.
__global__ void k_rdo_spec_jobs(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, const int *__restrict__ nnz_y, const int *__restrict__ nnz_u,
                                const int *__restrict__ nnz_v, Cand *__restrict__ cand, xeve_hip_cu_bits_job *__restrict__ bj, xeve_hip_cu_bits_job *__restrict__ cj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    Cand c;
    c.nnz_store[0] = nnz_y[j], c.nnz_store[1] = P.ncomp > 1 ? nnz_u[j] : 0, c.nnz_store[2] = P.ncomp > 1 ? nnz_v[j] : 0;
    c.tnnz = c.nnz_store[0] + c.nnz_store[1] + c.nnz_store[2];
    c.cost_best = MAX_COST, c.idx_best[0] = c.idx_best[1] = c.idx_best[2] = 0, c.cbf_idx[0] = c.cbf_idx[1] = c.cbf_idx[2] = 0;
    c.iy = c.nnz_store[0] > 0, c.iu = c.nnz_store[1] > 0, c.iv = c.nnz_store[2] > 0, c.round4 = 0, c.win = 0;
    cand[j] = c;
    const int N = P.njobs, all = c.iy | (c.iu << 1) | (c.iv << 2);
    for(int k = 0; k < 8; k++) {
        // a combination is a job of its own only when it keeps nothing but stored components and is neither all-zero (kind 0) nor everything stored
        // (kind 7 stands for that one whatever `all` is); the others are switched off (a skip-mode job: a handful of header bins)
        const bool on = k == 0 || k == 7 || ((k & ~all) == 0 && k != all);
        const int  kk = k == 7 ? all : k;
        fill_bits_job(bj[(size_t)k * N + j], J, P, j, on ? XEVE_HIP_BITS_CU_INTER : XEVE_HIP_BITS_CU_SKIP, (kk & 1) ? c.nnz_store[0] : 0, (kk & 2) ? c.nnz_store[1] : 0,
                      (kk & 4) ? c.nnz_store[2] : 0, J.sbac);
    }
    for(int l = 0; l < 4; l++) { // chain lane l assumes (Y kept = l & 1, U kept = l >> 1); impossible assumptions are switched off
        const bool on = c.tnnz != 0 && (!(l & 1) || c.iy) && (!(l & 2) || c.iu);
        xeve_hip_cu_bits_job &b = cj[(size_t)l * N + j];
        fill_bits_job(b, J, P, j, on ? XEVE_HIP_BITS_COMP_Y : XEVE_HIP_BITS_CU_SKIP, c.nnz_store[0], c.nnz_store[1], c.nnz_store[2], J.sbac);
        b.dir_flag = (uint8_t)l;
    }
}

// the reference's decision (xeve_pinter.c:1103-1275) over the counts of the one round; also zeroes nothing: k_rdo_zero_dropped follows
__global__ void k_rdo_spec_decide(const xeve_hip_rdo_job *__restrict__ jobs, RdoK P, const long *ssd_y, const long *ssd_u, const long *ssd_v,
                                  const unsigned *__restrict__ bits, const unsigned *__restrict__ cbits, const xeve_hip_sbac *__restrict__ st_out,
                                  Cand *__restrict__ cand, xeve_hip_rdo_result *__restrict__ res, unsigned char *__restrict__ drop, xeve_hip_sbac *__restrict__ best)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_rdo_job J = jobs[j];
    Cand c = cand[j];
    long d0[3], d1[3];
    load_dist(ssd_y, ssd_u, ssd_v, j, P, d0, d1);
    const size_t N = P.njobs;
    auto wb = [&](int k) { return (double)(int)bits[(size_t)k * N + j]; };                        // whole-CU count of combination k
    auto cb = [&](int lane, int which) { return (double)(int)cbits[4 * ((size_t)lane * N + j) + which]; }; // chain lane: 0 Y, 1 U, 2 V without, 3 V with
    if(c.tnnz == 0) { // nothing survived quantisation (:1276-1331)
        c.cost_best = (double)d0[0] + (P.wgt[0] * (double)d0[1]) + (P.wgt[1] * (double)d0[2]);
        c.cost_best += wb(0) * P.lambda[0];
        c.win = 0;
        if(best) copy_state(best + j, st_out + j);
    }
    else {
        if(!J.dir_flag) { // all-zero alternative (:1103-1142)
            double cost = sum_cost(d0, d1, 0, 0, 0, P);
            cost += wb(0) * P.lambda[0];
            if(cost < c.cost_best) {
                c.cost_best = cost, c.cbf_idx[0] = c.cbf_idx[1] = c.cbf_idx[2] = 0, c.win = 0;
                if(best) copy_state(best + j, st_out + j);
            }
        }
        { // as quantised (:1144-1178)
            double cost = sum_cost(d0, d1, c.iy, c.iu, c.iv, P);
            cost += wb(7) * P.lambda[0];
            if(cost < c.cost_best) {
                c.cost_best = cost, c.cbf_idx[0] = c.iy, c.cbf_idx[1] = c.iu, c.cbf_idx[2] = c.iv, c.win = 1;
                if(best) copy_state(best + j, st_out + 7 * N + j);
            }
        }
        // the component tests (:1180-1218): without is taken first, with must be strictly cheaper; each test reads the lane that assumed the picks so far
        int lane = 0;
        if(c.nnz_store[0] > 0) {
            double c0 = (double)d0[0] * 1.0, c1 = (double)d1[0] * 1.0;
            c0 += cb(0, 0) * P.lambda[0], c1 += cb(1, 0) * P.lambda[0];
            c.idx_best[0] = c1 < c0;
            lane |= c.idx_best[0];
        }
        if(c.nnz_store[1] > 0) {
            double c0 = (double)d0[1] * P.wgt[0], c1 = (double)d1[1] * P.wgt[0];
            c0 += cb(lane, 1) * P.lambda[1], c1 += cb(lane | 2, 1) * P.lambda[1];
            c.idx_best[1] = c1 < c0;
            lane |= c.idx_best[1] << 1;
        }
        if(c.nnz_store[2] > 0) {
            double c0 = (double)d0[2] * P.wgt[1], c1 = (double)d1[2] * P.wgt[1];
            c0 += cb(lane, 2) * P.lambda[2], c1 += cb(lane, 3) * P.lambda[2];
            c.idx_best[2] = c1 < c0;
        }
        // the combination the tests chose (:1220-1262)
        int n[3] = {c.nnz_store[0], c.nnz_store[1], c.nnz_store[2]};
        if(c.idx_best[0] || c.idx_best[1] || c.idx_best[2]) {
            c.iy = c.idx_best[0], c.iu = c.idx_best[1], c.iv = c.idx_best[2];
            n[0] = c.iy ? n[0] : 0, n[1] = c.iu ? n[1] : 0, n[2] = c.iv ? n[2] : 0;
        }
        c.round4 = n[0] != c.nnz_store[0] || n[1] != c.nnz_store[1] || n[2] != c.nnz_store[2];
        if(c.round4) {
            const int k = (n[0] ? 1 : 0) | (n[1] ? 2 : 0) | (n[2] ? 4 : 0);
            double cost = sum_cost(d0, d1, c.iy, c.iu, c.iv, P);
            cost += wb(k) * P.lambda[0];
            if(cost < c.cost_best) {
                c.cost_best = cost, c.cbf_idx[0] = c.iy, c.cbf_idx[1] = c.iu, c.cbf_idx[2] = c.iv, c.win = 2;
                if(best) copy_state(best + j, st_out + (size_t)k * N + j);
            }
        }
    }
    cand[j] = c;
    xeve_hip_rdo_result r;
    r.cost = c.cost_best, r.pad_ = 0;
    for(int k = 0; k < 3; k++) {
        r.nnz[k] = c.tnnz != 0 && c.cbf_idx[k] ? c.nnz_store[k] : 0;
        drop[3 * j + k] = r.nnz[k] == 0 && c.nnz_store[k] != 0;
        r.dist[0][k] = d0[k], r.dist[1][k] = c.tnnz != 0 ? d1[k] : 0;
    }
    res[j] = r;
}

// Candidates per batch up to which the one-round form is used.  MEASURED (3840x2160 i.i.d. picture, all four levels side by side on four streams): with the form
// on for the 64x64 and 32x32 levels (limit 26 000) a step takes 56.2 ms, with the four-round form everywhere 49.1 ms -- once the four levels share the chip the
// extra bins (667 M against 600 M per picture) cost more than the shorter chain saves.  It stays for what IS latency-bound: the per-CU calls of the host form
// (one to three candidates; three fewer dependent bit-count launches per batch).
static int rdo_spec_limit()
{
    static const int v = getenv("XEVE_HIP_RDO_SPEC") ? atoi(getenv("XEVE_HIP_RDO_SPEC")) : 256; // developer switch: 0 = never
    return v;
}
static bool rdo_use_spec(int njobs) { return njobs <= rdo_spec_limit(); }

// ---- host ------------------------------------------------------------------------------------------------------------------
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct RdoLayout {
    size_t mc, rl, rc, est_idx, est, pred[3], rec[3], nnz[3], ssd[3], cand, prev, bj, bits, st_out, drop, mcws, bitws, total;
};
static RdoLayout rdo_layout(int njobs, int n0, int n1, int nstates, size_t rec_l, size_t rec_c, int w, int h, int nr0, int nr1)
{
    RdoLayout L;
    size_t o = 0, n = njobs;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    L.mc = take(n * sizeof(xeve_hip_cu_mc_job)), L.rl = take(n * sizeof(xeve_hip_job)), L.rc = take(n * sizeof(xeve_hip_job));
    L.est_idx = take(n * 4), L.est = take((size_t)nstates * sizeof(xeve_hip_rdoq_est_full));
    L.pred[0] = take(n * n0 * 2), L.pred[1] = take(n * n1 * 2 + 8), L.pred[2] = take(n * n1 * 2 + 8);
    (void)rec_l, (void)rec_c;
    L.rec[0] = take(n * n0 * 2), L.rec[1] = take(n * n1 * 2 + 8), L.rec[2] = take(n * n1 * 2 + 8); // dense blocks (nothing reads them: pinter_residue_rdo's reconstruction only feeds the SSD)
    for(int k = 0; k < 3; k++) L.nnz[k] = take(n * 4), L.ssd[k] = take(n * 16);
    L.cand = take(n * sizeof(Cand)), L.prev = take(n * sizeof(xeve_hip_sbac));
    const size_t kinds = rdo_use_spec(njobs) ? 8 : 4; // whole-CU job kinds per candidate (the one-round form adds 4 chain lanes with 4 counts each)
    L.bj = take((kinds + 4) * n * sizeof(xeve_hip_cu_bits_job)), L.bits = take((kinds + 16) * n * 4), L.st_out = take(kinds * n * sizeof(xeve_hip_sbac));
    L.drop = take(3 * n);
    L.mcws = take(xeve_hip_mc_cu_workspace(njobs, w, h, nr0, nr1));
    L.bitws = take(xeve_hip_cu_bits_workspace((int)kinds * njobs, n * ((size_t)n0 + 2 * (size_t)n1)));
    L.total = o;
    return L;
}

extern "C" size_t xeve_hip_residue_rdo_workspace(int njobs, int nstates, const xeve_hip_rdo_params *p, int s_org_l, int s_org_c)
{
    if(!p || njobs <= 0) return 256;
    const int ws = p->chroma_format_idc <= 2, hs = p->chroma_format_idc <= 1;
    const int n0 = 1 << (p->log2_cuw + p->log2_cuh), n1 = p->chroma_format_idc ? n0 >> (ws + hs) : 0;
    return rdo_layout(njobs, n0, n1, nstates > 0 ? nstates : 1, (size_t)s_org_l * p->pic_h, (size_t)s_org_c * (p->pic_h >> hs), 1 << p->log2_cuw, 1 << p->log2_cuh,
                      p->num_refp[0], p->num_refp[1]).total;
}

static const int k_q_scale[6]  = {26214, 23302, 20560, 18396, 16384, 14764}; // xeve_quant_scale[0] (xeve_tq.c:37)
static const int k_dq_scale[6] = {40, 45, 51, 57, 64, 71};                   // xeve_tbl_dq_scale_b (xeve_tbl.c:237)

extern "C" int xeve_hip_residue_rdo_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                                         const xeve_hip_sbac *states, int nstates, const xeve_hip_rdo_params *p, const xeve_hip_rdo_job *jobs, int njobs,
                                         const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_rdo_result *results, int16_t *coef,
                                         xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream)
{
    return xh_residue_rdo_jobs_x(org, s_org_l, s_org_c, refp, s_l, s_c, states, nstates, p, jobs, njobs, coef_l, coef_c, results, coef, best, workspace, workspace_bytes, stream,
                                 nullptr, 0);
}

// est_shared: core->rdoq_est_* of every entry state, made by the caller (xeve_hip_rdoq_bit_est over the same states) -- NULL: made here; keep_dropped: the coefficient
// blocks of components the decision drops are NOT zeroed (the caller goes by results[].nnz) -- the inter analysis' two launches less per batch
int xh_residue_rdo_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, const xeve_hip_sbac *states, int nstates,
                          const xeve_hip_rdo_params *p, const xeve_hip_rdo_job *jobs, int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4],
                          xeve_hip_rdo_result *results, int16_t *coef, xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream,
                          const void *est_shared, int keep_dropped)
{
    XH_ENTER();
    XH_REQUIRE(org && refp && states && nstates > 0 && p && jobs && njobs >= 0 && results && coef && workspace && coef_l);
    XH_REQUIRE(p->log2_cuw >= 2 && p->log2_cuw <= 6 && p->log2_cuh >= 2 && p->log2_cuh <= 6 && p->tool_iqt == 0);
    XH_REQUIRE(p->chroma_format_idc == 0 || p->chroma_format_idc == 1 || p->chroma_format_idc == 3);
    XH_REQUIRE(org[0] && (p->chroma_format_idc == 0 || (org[1] && org[2] && coef_c)));
    if(njobs == 0) return XEVE_HIP_OK;
    XH_REQUIRE(workspace_bytes >= xeve_hip_residue_rdo_workspace(njobs, nstates, p, s_org_l, s_org_c));
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth;
    const int lw[3] = {p->log2_cuw, p->log2_cuw - ws, p->log2_cuw - ws}, lh[3] = {p->log2_cuh, p->log2_cuh - hs, p->log2_cuh - hs};
    RdoK P;
    P.n0 = 1 << (lw[0] + lh[0]), P.n1 = idc ? 1 << (lw[1] + lh[1]) : 0, P.ncomp = idc ? 3 : 1, P.njobs = njobs, P.dir_unused = 0;
    P.w = 1 << lw[0], P.h = 1 << lh[0], P.ws = ws, P.hs = hs, P.s_org_l = s_org_l, P.s_org_c = s_org_c;
    for(int k = 0; k < 3; k++) P.lambda[k] = p->lambda[k];
    P.wgt[0] = p->dist_chroma_weight[0], P.wgt[1] = p->dist_chroma_weight[1];
    const RdoLayout L = rdo_layout(njobs, P.n0, P.n1, nstates, (size_t)s_org_l * p->pic_h, (size_t)s_org_c * (p->pic_h >> hs), P.w, P.h, p->num_refp[0], p->num_refp[1]);
    char *W = (char *)workspace;
    auto *rl = (xeve_hip_job *)(W + L.rl), *rc = (xeve_hip_job *)(W + L.rc);
    int  *est_idx = (int *)(W + L.est_idx);
    auto *est = est_shared ? (xeve_hip_rdoq_est_full *)est_shared : (xeve_hip_rdoq_est_full *)(W + L.est);
    pel  *pred[3] = {(pel *)(W + L.pred[0]), (pel *)(W + L.pred[1]), (pel *)(W + L.pred[2])};
    pel  *rec[3]  = {(pel *)(W + L.rec[0]), (pel *)(W + L.rec[1]), (pel *)(W + L.rec[2])};
    int  *nnz[3]  = {(int *)(W + L.nnz[0]), (int *)(W + L.nnz[1]), (int *)(W + L.nnz[2])};
    long *ssd[3]  = {(long *)(W + L.ssd[0]), (long *)(W + L.ssd[1]), (long *)(W + L.ssd[2])};
    auto *cand = (Cand *)(W + L.cand);
    auto *prev = (xeve_hip_sbac *)(W + L.prev), *st_out = (xeve_hip_sbac *)(W + L.st_out);
    auto *bj = (xeve_hip_cu_bits_job *)(W + L.bj);
    auto *bits = (unsigned *)(W + L.bits);
    auto *drop = (unsigned char *)(W + L.drop);
    hipStream_t st = (hipStream_t)stream;
    const int   G  = (njobs + 255) / 256;
    int rc_;

    CuMcPrep C;
    rc_ = xh_mc_cu_prep_params(refp, p->num_refp[0], p->num_refp[1], p->pic_w, p->pic_h, njobs, P.w, P.h, idc, W + L.mcws, L.bitws - L.mcws, &C);
    if(rc_ != XEVE_HIP_OK) return rc_;
    k_rdo_prep<<<G, 256, 0, st>>>(jobs, P, C, rl, rc, est_idx);
    // prediction (:962)
    rc_ = xh_mc_cu_jobs_x(refp, p->num_refp[0], p->num_refp[1], s_l, s_c, p->pic_w, p->pic_h, nullptr, njobs, P.w, P.h, bd, bd, idc, coef_l, coef_c, pred[0],
                          pred[1], pred[2], W + L.mcws, L.bitws - L.mcws, stream, XH_MC_PREPPED);
    if(rc_ != XEVE_HIP_OK) return rc_;
    // the estimates of every entry state (xeve_mode.c:792)
    if(!est_shared) {
        rc_ = xeve_hip_rdoq_bit_est(states, nstates, est, stream);
        if(rc_ != XEVE_HIP_OK) return rc_;
    }
    // residual, SSD, transform, zero pre-test + RDOQ, reconstruction, SSD (:969-1051)
    int16_t *cf[3] = {coef, coef + (size_t)njobs * P.n0, coef + (size_t)njobs * (P.n0 + P.n1)};
    for(int k = 0; k < P.ncomp; k++) {
        const int q = p->qp[k];
        XH_REQUIRE(q >= 0 && q <= 51 + 6 * (bd - 8)); // MAX_QUANT + the bit-depth offset
        rc_ = xeve_hip_residual_rdoq_dev(org[k], k ? s_org_c : s_org_l, pred[k], 1 << lw[k], k ? rc : rl, njobs, lw[k], lh[k], bd, q, k_q_scale[q % 6],
                                         k_dq_scale[q % 6] << (q / 6), p->slice_type == 2, p->lambda[k], k, p->tool_iqt, est, est_idx, cf[k], rec[k],
                                         -(1 << lw[k]), nnz[k], (int64_t *)ssd[k], stream);
        if(rc_ != XEVE_HIP_OK) return rc_;
    }
    // the decision
    xeve_hip_cu_bits_params bp;
    bp.log2_cuw = p->log2_cuw, bp.log2_cuh = p->log2_cuh, bp.slice_type = p->slice_type, bp.num_refp[0] = p->num_refp[0], bp.num_refp[1] = p->num_refp[1];
    bp.cm_init = 0, bp.chroma_format_idc = idc;
    const size_t coef_elems = (size_t)njobs * (P.n0 + 2 * (size_t)P.n1), bws = workspace_bytes - L.bitws;
    if(rdo_use_spec(njobs)) { // one bit-count round: every count the decision can ask for, side by side
        xeve_hip_cu_bits_job *cj = bj + 8 * (size_t)njobs;
        unsigned             *cbits = bits + 8 * (size_t)njobs;
        k_rdo_spec_jobs<<<G, 256, 0, st>>>(jobs, P, nnz[0], nnz[1], nnz[2], cand, bj, cj);
        // (events and bin strings from the as-quantised jobs alone: they code every stored block exactly once)
        rc_ = xh_cu_bits_jobs_round(coef, coef_elems, states, bj, 8 * njobs, &bp, W + L.bitws, bws, bits, best ? st_out : nullptr, best != nullptr, 0, stream, 7 * njobs, njobs);
        if(rc_ != XEVE_HIP_OK) return rc_;
        rc_ = xh_cu_bits_chain_round(coef_elems, states, cj, 4 * njobs, &bp, W + L.bitws, bws, cbits, stream);
        if(rc_ != XEVE_HIP_OK) return rc_;
        k_rdo_spec_decide<<<G, 256, 0, st>>>(jobs, P, ssd[0], ssd[1], ssd[2], bits, cbits, st_out, cand, results, drop, best);
        if(!keep_dropped) k_rdo_zero_dropped<<<(3 * njobs + 3) / 4, 256, 0, st>>>(coef, drop, P);
        XH_HIP(hipGetLastError());
        return XEVE_HIP_OK;
    }
    k_rdo_round1<<<G, 256, 0, st>>>(jobs, P, nnz[0], nnz[1], nnz[2], cand, bj);
    // (the complete coder state is carried only where core->s_temp_best can come from -- the whole-CU counts of rounds 1 and 4 -- and only
    // when the caller wants it; the component tests' states are only ever loaded into further counts: range and models suffice)
    // (the event lists of the coefficient blocks are made in round 1 -- its "as quantised" jobs code every non-zero block -- and reused after)
    rc_ = xh_cu_bits_jobs_round(coef, coef_elems, states, bj, 4 * njobs, &bp, W + L.bitws, bws, bits, st_out, best != nullptr, 0, stream, njobs, njobs); // (events from the as-quantised jobs)
    if(rc_ != XEVE_HIP_OK) return rc_;
    k_rdo_decide1<<<G, 256, 0, st>>>(jobs, P, ssd[0], ssd[1], ssd[2], bits, st_out, states, cand, prev, bj, best);
    if(P.ncomp > 1) {
        for(int comp = 1; comp <= 2; comp++) {
            rc_ = xh_cu_bits_jobs_round(coef, coef_elems, prev, bj, 2 * njobs, &bp, W + L.bitws, bws, bits, st_out, 0, 1, stream);
            if(rc_ != XEVE_HIP_OK) return rc_;
            k_rdo_decide_comp<<<G, 256, 0, st>>>(jobs, P, comp, ssd[0], ssd[1], ssd[2], bits, st_out, cand, prev, bj);
        }
    }
    else k_rdo_decide_comp<<<G, 256, 0, st>>>(jobs, P, 2, ssd[0], ssd[1], ssd[2], bits, st_out, cand, prev, bj); // (no chroma: only builds the round 4 job)
    rc_ = xh_cu_bits_jobs_round(coef, coef_elems, states, bj, njobs, &bp, W + L.bitws, bws, bits, best ? st_out : nullptr, best != nullptr, 1, stream);
    if(rc_ != XEVE_HIP_OK) return rc_;
    k_rdo_finish<<<G, 256, 0, st>>>(jobs, P, ssd[0], ssd[1], ssd[2], bits, cand, results, drop, st_out, best);
    if(!keep_dropped) k_rdo_zero_dropped<<<(3 * njobs + 3) / 4, 256, 0, st>>>(coef, drop, P);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// =============================================================================================================================
// xeve_analyze_skip (src_base/xeve_pinter.c:1337-1530) for a batch of CUs of one size: every (idx0, idx1) pair of the merge candidate
// list that survives the encoder side pruning is predicted (xeve_mc), measured (SSD Y + weighted U, V) and priced (skip flag + the two
// candidate indices through the CABAC counter, from the CU's entry state); the first pair with the strictly smallest cost wins.
// One slot per pair (max_cand^2 in B slices, max_cand in P slices): slots run in parallel through the batched building blocks, pruned /
// unusable slots are switched off, and one thread per CU walks its slots in the reference's loop order for the decision.
// =============================================================================================================================
struct SkipK {
    int    njobs, S, mc, isb, n0, n1, ncomp, ws, hs, s_org_l, s_org_c, best_shift;
    double lambda0, wgt[2];
};

__global__ void k_skip_prep(const xeve_hip_skip_job *__restrict__ jobs, SkipK P, CuMcPrep C, xeve_hip_cu_mc_job *__restrict__ mc, xeve_hip_job *__restrict__ rl,
                            xeve_hip_job *__restrict__ rc, xeve_hip_cu_bits_job *__restrict__ bj, unsigned char *__restrict__ valid, int *__restrict__ zero)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t < 64) zero[t] = 0; // (the one-candidate offset table of the SSD launches below)
    if(t >= P.njobs * P.S) return;
    const int j = t / P.S, s = t % P.S;
    const int i0 = P.isb ? s / P.mc : s, i1 = P.isb ? s % P.mc : 0;
    const xeve_hip_skip_job J = jobs[j];
    const int cnt1 = P.isb ? J.ncand : 1;
    bool ok = i0 < J.ncand && i1 < cnt1;
    // encoder side pruning (:1396-1409, :1420-1433): an earlier candidate of the same list with the same vector
    for(int k = 0; k < 3; k++) {
        if(k < i0 && J.mvp[0][k][0] == J.mvp[0][i0][0] && J.mvp[0][k][1] == J.mvp[0][i0][1]) ok = false;
        if(k < i1 && J.mvp[1][k][0] == J.mvp[1][i1][0] && J.mvp[1][k][1] == J.mvp[1][i1][1]) ok = false;
    }
    xeve_hip_cu_mc_job m;
    m.x = J.x, m.y = J.y, m.mv[0][0] = J.mvp[0][i0][0], m.mv[0][1] = J.mvp[0][i0][1], m.mv[1][0] = J.mvp[1][i1][0], m.mv[1][1] = J.mvp[1][i1][1];
    m.refi[0] = J.refi_pred[0][i0], m.refi[1] = P.isb ? J.refi_pred[1][i1] : -1, m.pad_[0] = m.pad_[1] = 0;
    if(m.refi[0] < 0 && m.refi[1] < 0) ok = false; // (:1444)
    if(!ok) m.refi[0] = m.refi[1] = -1;            // no prediction work for a slot that is not evaluated
    mc[t] = m;
    xh_cu_mc_prep_one(m, t, C); // (the prediction's per-list interpolation jobs: mc_cu.h)
    rl[t] = xh_make_job(J.y, P.s_org_l, J.x, t * P.n0);
    rc[t] = xh_make_job(J.y >> P.hs, P.s_org_c, J.x >> P.ws, t * P.n1);
    xeve_hip_cu_bits_job b;
    b.coef_off[0] = b.coef_off[1] = b.coef_off[2] = 0, b.nnz[0] = b.nnz[1] = b.nnz[2] = 0, b.sbac = J.sbac;
    b.mvd[0][0] = b.mvd[0][1] = b.mvd[1][0] = b.mvd[1][1] = 0, b.refi[0] = b.refi[1] = 0;
    b.mvp_idx[0] = (uint8_t)i0, b.mvp_idx[1] = (uint8_t)i1, b.mode = XEVE_HIP_BITS_CU_SKIP, b.dir_flag = 0, b.ctx_skip = J.ctx_skip, b.ctx_pred_mode = 0;
    bj[t] = b;
    valid[t] = ok;
}

// the walk over the slots in (idx0, idx1) order (:1391-1521) by the block's first thread, then pi->pred[PRED_SKIP][0] and core->s_temp_best of the winner (both untouched
// when no pair was usable) by all of them: one block per CU (round 6: was a decision kernel and a copy kernel)
__global__ void k_skip_decide_copy(const xeve_hip_skip_job *__restrict__ jobs, SkipK P, const xeve_hip_cu_mc_job *__restrict__ mc, const unsigned char *__restrict__ valid,
                                   const long *__restrict__ ssd_y, const long *__restrict__ ssd_u, const long *__restrict__ ssd_v, const unsigned *__restrict__ bits,
                                   xeve_hip_skip_result *__restrict__ res, int *__restrict__ win, const pel *__restrict__ sy, const pel *__restrict__ su,
                                   const pel *__restrict__ sv, pel *__restrict__ py, pel *__restrict__ pu, pel *__restrict__ pv, const xeve_hip_sbac *__restrict__ st,
                                   xeve_hip_sbac *__restrict__ best)
{
    __shared__ int s_win;
    const int j = blockIdx.x;
    if(threadIdx.x == 0) {
        double cost_best = MAX_COST;
        int    sb = -1;
        long   ssd_best = 1L << P.best_shift;
        for(int s = 0; s < P.S; s++) {
            const int t = j * P.S + s;
            if(!valid[t]) continue;
            const long cy = ssd_y[t], cu = P.ncomp > 1 ? ssd_u[t] : 0, cv = P.ncomp > 1 ? ssd_v[t] : 0;
            double cost = (double)cy + (P.wgt[0] * (double)cu) + (P.wgt[1] * (double)cv); // (:1467-1473)
            cost += (double)(int)bits[t] * P.lambda0;                                      // RATE_TO_COST_LAMBDA (:1488)
            if(cost < cost_best) cost_best = cost, sb = s, ssd_best = cy + cu + cv;
        }
        xeve_hip_skip_result r;
        r.cost = cost_best, r.best_ssd = ssd_best, r.idx0 = r.idx1 = 0;
        r.mv[0][0] = r.mv[0][1] = r.mv[1][0] = r.mv[1][1] = 0, r.refi[0] = r.refi[1] = 0;
        for(int k = 0; k < 6; k++) r.pad_[k] = 0;
        if(sb >= 0) {
            const xeve_hip_cu_mc_job m = mc[j * P.S + sb];
            r.idx0 = P.isb ? sb / P.mc : sb, r.idx1 = P.isb ? sb % P.mc : 0;
            r.mv[0][0] = m.mv[0][0], r.mv[0][1] = m.mv[0][1], r.mv[1][0] = m.mv[1][0], r.mv[1][1] = m.mv[1][1], r.refi[0] = m.refi[0], r.refi[1] = m.refi[1];
        }
        res[j] = r;
        win[j] = sb;
        s_win = sb;
    }
    __syncthreads();
    const int sb = s_win;
    if(sb < 0) return;
    const size_t t = (size_t)j * P.S + sb;
    for(int k = 0; k < P.ncomp; k++) {
        const int  n = k ? P.n1 : P.n0;
        const pel *s = (k == 0 ? sy : k == 1 ? su : sv) + t * n;
        pel       *d = (k == 0 ? py : k == 1 ? pu : pv) + (size_t)j * n;
        for(int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
    }
    if(best) { // (the slot's state after its skip flag and candidate indices: SBAC_STORE(core->s_temp_best, *sbac), :1519)
        const unsigned *a = (const unsigned *)(st + t);
        unsigned       *b = (unsigned *)(best + j);
        for(int i = threadIdx.x; i < (int)(sizeof(xeve_hip_sbac) / 4); i += blockDim.x) b[i] = a[i];
    }
}

struct SkipLayout {
    size_t mc, rl, rc, bj, valid, pred[3], ssd[3], bits, win, bjw, bitsw, stw, zero, mcws, bitws, total;
};
static SkipLayout skip_layout(int njobs, int S, int n0, int n1, int w, int h, int nr0, int nr1)
{
    SkipLayout L;
    size_t o = 0, n = (size_t)njobs * S;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    L.mc = take(n * sizeof(xeve_hip_cu_mc_job)), L.rl = take(n * sizeof(xeve_hip_job)), L.rc = take(n * sizeof(xeve_hip_job));
    L.bj = take(n * sizeof(xeve_hip_cu_bits_job)), L.valid = take(n);
    L.pred[0] = take(n * n0 * 2), L.pred[1] = take(n * n1 * 2 + 8), L.pred[2] = take(n * n1 * 2 + 8);
    for(int k = 0; k < 3; k++) L.ssd[k] = take(n * 8);
    L.bits = take(n * 4), L.win = take((size_t)njobs * 4), L.bjw = take((size_t)njobs * sizeof(xeve_hip_cu_bits_job));
    L.bitsw = take((size_t)njobs * 4), L.stw = take(n * sizeof(xeve_hip_sbac)), L.zero = take(256);
    L.mcws = take(xeve_hip_mc_cu_workspace((int)n, w, h, nr0, nr1));
    L.bitws = take(xeve_hip_cu_bits_workspace((int)n, 64));
    L.total = o;
    return L;
}

static int skip_slots(const xeve_hip_rdo_params *p, int max_cand) { return p->slice_type == 0 ? max_cand * max_cand : max_cand; }

extern "C" size_t xeve_hip_analyze_skip_workspace(int njobs, const xeve_hip_rdo_params *p, int max_cand)
{
    if(!p || njobs <= 0 || max_cand < 1 || max_cand > 4) return 256;
    const int ws = p->chroma_format_idc <= 2, hs = p->chroma_format_idc <= 1;
    const int n0 = 1 << (p->log2_cuw + p->log2_cuh), n1 = p->chroma_format_idc ? n0 >> (ws + hs) : 0;
    return skip_layout(njobs, skip_slots(p, max_cand), n0, n1, 1 << p->log2_cuw, 1 << p->log2_cuh, p->num_refp[0], p->num_refp[1]).total;
}

extern "C" int xeve_hip_analyze_skip_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c,
                                          const xeve_hip_sbac *states, int nstates, const xeve_hip_rdo_params *p, const xeve_hip_skip_job *jobs, int njobs,
                                          int max_cand, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], xeve_hip_skip_result *results,
                                          xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, xeve_hip_sbac *best, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(p && njobs >= 0 && max_cand >= 1 && max_cand <= 4 && (p->slice_type == 0 || p->slice_type == 1));
    if(njobs == 0) return XEVE_HIP_OK;
    XH_REQUIRE(org && refp && states && nstates > 0 && jobs && results && pred_y && workspace && coef_l);
    XH_REQUIRE(p->log2_cuw >= 2 && p->log2_cuw <= 6 && p->log2_cuh >= 2 && p->log2_cuh <= 6);
    XH_REQUIRE(p->chroma_format_idc == 0 || p->chroma_format_idc == 1 || p->chroma_format_idc == 3);
    XH_REQUIRE(org[0] && (p->chroma_format_idc == 0 || (org[1] && org[2] && coef_c && pred_u && pred_v)));
    if(njobs == 0) return XEVE_HIP_OK;
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth;
    SkipK P;
    P.njobs = njobs, P.mc = max_cand, P.isb = p->slice_type == 0, P.S = skip_slots(p, max_cand);
    XH_REQUIRE((long)njobs * P.S < (1L << 30) / 64);
    P.n0 = 1 << (p->log2_cuw + p->log2_cuh), P.n1 = idc ? P.n0 >> (ws + hs) : 0, P.ncomp = idc ? 3 : 1, P.ws = ws, P.hs = hs;
    P.s_org_l = s_org_l, P.s_org_c = s_org_c, P.best_shift = p->log2_cuw + p->log2_cuh + 16; // pi->best_ssd's reset value
    P.lambda0 = p->lambda[0], P.wgt[0] = p->dist_chroma_weight[0], P.wgt[1] = p->dist_chroma_weight[1];
    const int w = 1 << p->log2_cuw, h = 1 << p->log2_cuh, nt = njobs * P.S;
    const SkipLayout L = skip_layout(njobs, P.S, P.n0, P.n1, w, h, p->num_refp[0], p->num_refp[1]);
    XH_REQUIRE(workspace_bytes >= L.total);
    char *W  = (char *)workspace;
    auto *mc = (xeve_hip_cu_mc_job *)(W + L.mc);
    auto *rl = (xeve_hip_job *)(W + L.rl), *rc = (xeve_hip_job *)(W + L.rc);
    auto *bj = (xeve_hip_cu_bits_job *)(W + L.bj);
    auto *valid = (unsigned char *)(W + L.valid);
    pel  *pred[3] = {(pel *)(W + L.pred[0]), (pel *)(W + L.pred[1]), (pel *)(W + L.pred[2])};
    long *ssd[3]  = {(long *)(W + L.ssd[0]), (long *)(W + L.ssd[1]), (long *)(W + L.ssd[2])};
    auto *bits = (unsigned *)(W + L.bits);
    int  *win = (int *)(W + L.win), *zero = (int *)(W + L.zero);
    auto *stw = (xeve_hip_sbac *)(W + L.stw);
    hipStream_t st = (hipStream_t)stream;
    int rc_;

    CuMcPrep C;
    rc_ = xh_mc_cu_prep_params(refp, p->num_refp[0], p->num_refp[1], p->pic_w, p->pic_h, nt, w, h, idc, W + L.mcws, L.bitws - L.mcws, &C);
    if(rc_ != XEVE_HIP_OK) return rc_;
    k_skip_prep<<<(nt + 255) / 256, 256, 0, st>>>(jobs, P, C, mc, rl, rc, bj, valid, zero);
    // xeve_mc of every pair (:1453)
    rc_ = xh_mc_cu_jobs_x(refp, p->num_refp[0], p->num_refp[1], s_l, s_c, p->pic_w, p->pic_h, nullptr, nt, w, h, bd, bd, idc, coef_l, coef_c, pred[0], pred[1],
                          pred[2], W + L.mcws, L.bitws - L.mcws, stream, XH_MC_PREPPED);
    if(rc_ != XEVE_HIP_OK) return rc_;
    // xeve_ssd per component (:1455-1465)
    for(int k = 0; k < P.ncomp; k++) {
        rc_ = xeve_hip_ssd_jobs(org[k], k ? s_org_c : s_org_l, pred[k], k ? w >> ws : w, k ? rc : rl, nt, zero, 1, k ? w >> ws : w, k ? h >> hs : h, bd,
                                (int64_t *)ssd[k], stream);
        if(rc_ != XEVE_HIP_OK) return rc_;
    }
    // skip flag + candidate indices from the entry state (:1479-1487)
    xeve_hip_cu_bits_params bp;
    bp.log2_cuw = p->log2_cuw, bp.log2_cuh = p->log2_cuh, bp.slice_type = p->slice_type, bp.num_refp[0] = p->num_refp[0], bp.num_refp[1] = p->num_refp[1];
    bp.cm_init = 0, bp.chroma_format_idc = idc;
    // (with `best`: every slot's coder state comes out of the same launch -- a handful of header bins per lane -- and k_skip_copy takes the winner's: round 6, was a
    // second count of the winners alone on the dependent chain)
    rc_ = xeve_hip_cu_bits_jobs(nullptr, 0, states, bj, nt, &bp, W + L.bitws, workspace_bytes - L.bitws, bits, best ? stw : nullptr, stream);
    if(rc_ != XEVE_HIP_OK) return rc_;
    k_skip_decide_copy<<<njobs, P.n0 >= 256 ? 256 : 64, 0, st>>>(jobs, P, mc, valid, ssd[0], ssd[1], ssd[2], bits, results, win, pred[0], pred[1], pred[2], pred_y, pred_u,
                                                                 pred_v, stw, best);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
