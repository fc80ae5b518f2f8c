// xeve_amd/csrc/intra.hip -- the intra analysis of a batch of CUs of one size (pintra_analyze_cu, src_base/xeve_pintra.c:544-698 = ctx->fn_pintra_analyze_cu),
// composed on the device.  Baseline profile, rdo_dbk_switch 0, no delta QP, CU <= 64x64.
//
//   1. neighbours of every component from the picture being reconstructed + the 4x4-unit maps (xeve_get_nbr, xeve_ipred.c:32-105), the rank row of the
//      most probable modes (xeve_get_mpm, :229-252)                                                                         k_intra_nbr
//   2. the five Baseline predictors of the luma block (xeve_ipred, :107-202)                                                  k_intra_pred
//   3. SATD of each against the original (xeve_hip_satd_jobs) and the bits of each mode index from the CU's entry coder state
//      (bit counter, job mode XEVE_HIP_BITS_INTRA_DIR); the insertion-sorted candidate list and its cut against the best inter
//      prediction's SATD (make_ipred_list, xeve_pintra.c:308-374)                                                            k_intra_list
//   4. the luma RDO of the list (pintra_residue_rdo mode 0, :102-148): one SLOT per (CU, list position) -- residual, DCT, RDOQ with the entry
//      state's estimates, reconstruction, SSD (the fused residual chain) and the bits of the intra luma syntax; the first slot with the strictly
//      smallest cost wins                                                                                                    k_intra_pick
//   5. chroma with the winner's mode (mode 1, :150-269): predictors, chain for U and V, weighted SSD                            k_intra_pred (chroma)
//   6. the CU's cost: the whole intra syntax from the entry state (xeve_rdo_bit_cnt_cu_intra), core->s_temp_best               k_intra_finish
// All candidates of a CU start from the same entry state, so the slots of step 4 are independent; the bit count of the chroma RDO (which the reference
// runs on the state the last luma candidate left) never reaches an output -- cost_t of that call is discarded (:643-646) -- and is not computed.
// Slots past the cut are computed and ignored (the list always holds five modes).
#include <algorithm>
#include <cstring>
#include "xh_common.h"

extern "C" int xeve_hip_satd_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs, const int32_t *cand_off, int ncand, int w, int h,
                                  int bit_depth, int32_t *out, void *stream);
extern "C" int xeve_hip_rdoq_bit_est(const xeve_hip_sbac *sbac, int nstates, xeve_hip_rdoq_est_full *est, void *stream);
int xh_residual_rdoq(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h, int bit_depth, int qp,
                     int qscale, int dqscale, int is_intra_slice, int is_intra_cu, double lambda, int ch_type, int tool_iqt, const xeve_hip_rdoq_est_full *est,
                     const int32_t *est_idx, int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st); // tq.hip

#define NB 136 // one neighbour line: element 0 = the corner sample [-1], then up to 2 * 64 samples (+ slack)
#define SLOTS 5
#define MAX_COST 1.7e+308

struct IntraK {
    int    njobs, w, h, n0, n1, ncomp, ws, hs, idc;
    int    s_org_l, s_org_c, s_mod_l, s_mod_c, w_scu, h_scu, cip, bd, slice_type, rdo_cnt;
    long   org_pic_l, org_pic_c, mod_pic_l, mod_pic_c, map_pic;
    double lambda0, sqrt_lambda0, wgt[2];
};

// xeve_tbl_mpm (xeve_tbl.c:40-48): [left mode + 1 | 0][up mode + 1 | 0] -> rank of every mode
__constant__ unsigned char c_mpm[36][5] = {
    {0, 2, 3, 1, 4}, {0, 2, 1, 3, 4}, {0, 2, 1, 3, 4}, {1, 2, 0, 3, 4}, {0, 2, 1, 3, 4}, {0, 1, 2, 3, 4}, {1, 0, 2, 3, 4}, {0, 1, 2, 3, 4}, {0, 1, 2, 3, 4},
    {1, 2, 0, 3, 4}, {0, 1, 3, 2, 4}, {0, 2, 1, 4, 3}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {1, 0, 2, 3, 4}, {2, 0, 1, 3, 4}, {1, 0, 3, 2, 4}, {0, 1, 2, 4, 3},
    {1, 0, 2, 3, 4}, {0, 2, 1, 3, 4}, {1, 0, 2, 3, 4}, {1, 2, 0, 3, 4}, {0, 1, 2, 3, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 3, 2, 1, 4}, {1, 0, 2, 3, 4},
    {1, 2, 0, 3, 4}, {1, 2, 3, 0, 4}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}, {0, 1, 2, 4, 3}, {0, 2, 1, 4, 3}, {0, 1, 2, 3, 4}, {0, 1, 2, 4, 3}};

#define SCU_COD(m) (((m) >> 31) & 1u)
#define SCU_IF(m)  (((m) >> 15) & 1u)

// ---- 1. neighbours ---------------------------------------------------------------------------------------------------------------------------------------
// nb[((job * 3 + comp) * 2 + line) * NB + 1 + i]: line 0 = left, 1 = up; [.. + 0] = the corner sample.  One workgroup per (job, component).
__global__ void k_intra_nbr(const pel *__restrict__ mod_y, const pel *__restrict__ mod_u, const pel *__restrict__ mod_v, const uint32_t *__restrict__ map_scu,
                            const int8_t *__restrict__ map_ipm, const uint8_t *__restrict__ map_tidx, const xeve_hip_intra_job *__restrict__ jobs, IntraK P,
                            pel *__restrict__ nb, unsigned char *__restrict__ mpm_row)
{
    const int j = blockIdx.x, c = blockIdx.y;
    if(c >= P.ncomp) return;
    const xeve_hip_intra_job J = jobs[j];
    const uint32_t *ms = map_scu + (long)J.pic * P.map_pic;
    const uint8_t  *mt = map_tidx + (long)J.pic * P.map_pic;
    const int x_scu = J.x >> 2, y_scu = J.y >> 2, scup = y_scu * P.w_scu + x_scu;
    const int cw = c ? P.w >> P.ws : P.w, ch = c ? P.h >> P.hs : P.h;
    int scuw = c ? cw >> (2 - P.ws) : cw >> 2, scuh = c ? ch >> (2 - P.hs) : ch >> 2, unit = c ? 2 : 4;
    if(c && P.idc == 2) scuh *= 2;
    if(c && P.idc == 3) unit *= 2;
    const int  s   = c ? P.s_mod_c : P.s_mod_l;
    const pel *src = (c == 0 ? mod_y + (long)J.pic * P.mod_pic_l : (c == 1 ? mod_u : mod_v) + (long)J.pic * P.mod_pic_c) +
                     (c ? (long)(J.y >> P.hs) * s + (J.x >> P.ws) : (long)J.y * s + J.x);
    const pel grey = (pel)(1 << (P.bd - 1));
    pel *left = nb + ((long)(j * 3 + c) * 2) * NB + 1, *up = left + NB;
    auto usable = [&](int u) { return SCU_COD(ms[u]) && (!P.cip || SCU_IF(ms[u])) && mt[scup] == mt[u]; };
    const int nline = (scuw + scuh) * unit; // = cw + ch
    for(int t = threadIdx.x; t < 2 * nline + 1; t += blockDim.x) {
        if(t == 2 * nline) { // the corner (up-left) sample: avail_cu & AVAIL_UP_LE (xeve_util.c:753-755), then the constrained-intra test
            const bool ok = x_scu > 0 && y_scu > 0 && usable(scup - P.w_scu - 1);
            const pel v = ok ? src[-s - 1] : grey;
            up[-1] = v, left[-1] = v;
        }
        else if(t < nline) { // up
            const int i = t / unit;
            const bool ok = y_scu > 0 && x_scu + i < P.w_scu && usable(scup - P.w_scu + i);
            up[t] = ok ? src[-s + t] : grey;
        }
        else { // left
            const int k = t - nline, i = k / unit;
            const bool ok = x_scu > 0 && y_scu + i < P.h_scu && usable(scup - 1 + i * P.w_scu);
            left[k] = ok ? src[(long)k * s - 1] : grey;
        }
    }
    if(c == 0 && threadIdx.x == 0) { // xeve_get_mpm
        const int8_t *mi = map_ipm + (long)J.pic * P.map_pic;
        int l = 0, u = 0;
        if(x_scu > 0 && SCU_IF(ms[scup - 1]) && SCU_COD(ms[scup - 1]) && mt[scup] == mt[scup - 1]) l = mi[scup - 1] + 1;
        if(y_scu > 0 && SCU_IF(ms[scup - P.w_scu]) && SCU_COD(ms[scup - P.w_scu]) && mt[scup] == mt[scup - P.w_scu]) u = mi[scup - P.w_scu] + 1;
        mpm_row[j] = (unsigned char)(l * 6 + u);
    }
}

// ---- 2 / 5. predictors -----------------------------------------------------------------------------------------------------------------------------------
// luma (comp0 = 0): grid (njobs, 5), block (job, mode) -> pred[(job * 5 + mode) * n0]; chroma (comp0 = 1): grid (njobs, 2), the winner's mode -> pred[(c - 1) * njobs * n1 + job * n1]
// component c of job j with mode ipm into dst; every thread of the block calls it (the DC predictor's reduction synchronises)
__device__ __forceinline__ void intra_pred_block(const pel *__restrict__ nb, const IntraK &P, int j, int c, int ipm, pel *__restrict__ dst)
{
    const int w = c ? P.w >> P.ws : P.w, h = c ? P.h >> P.hs : P.h, n = w * h;
    const pel *left = nb + ((long)(j * 3 + c) * 2) * NB + 1, *up = left + NB;
    __shared__ int s_dc;
    if(ipm == 0) { // DC: (sum + w) >> (log2 w + 1) (xeve_ipred.c:133-150)
        __syncthreads(); // (a second call in the same kernel: nobody still reads the first one's sum)
        if(threadIdx.x == 0) s_dc = 0;
        __syncthreads();
        int a = 0;
        for(int t = threadIdx.x; t < w + h; t += blockDim.x) a += t < h ? left[t] : up[t - h];
        for(int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if((threadIdx.x & 63) == 0) atomicAdd(&s_dc, a);
        __syncthreads();
        const pel dc = (pel)((s_dc + w) >> (31 - __clz(w) + 1));
        for(int t = threadIdx.x; t < n; t += blockDim.x) dst[t] = dc;
        return;
    }
    const int lw = 31 - __clz(w);
    for(int t = threadIdx.x; t < n; t += blockDim.x) {
        const int i = t >> lw, k = t & (w - 1);
        int v;
        if(ipm == 1) v = left[i];
        else if(ipm == 2) v = up[k];
        else if(ipm == 3) v = i > k ? left[i - k - 1] : (i == k ? up[-1] : up[k - i - 1]);
        else v = (up[i + k + 1] + left[i + k + 1]) >> 1;
        dst[t] = (pel)v;
    }
}

// ---- 3. jobs of the SATD and of the mode-index bit count ---------------------------------------------------------------------------------------------------
// (element offsets into the stacked originals; xh_make_job halves and marks the even ones: xh_common.h XH_OFF2_HALF)
__device__ __forceinline__ size_t org_off_l(const xeve_hip_intra_job &J, const IntraK &P) { return (size_t)((long)J.pic * P.org_pic_l + (long)J.y * P.s_org_l + J.x); }
__device__ __forceinline__ size_t org_off_c(const xeve_hip_intra_job &J, const IntraK &P) { return (size_t)((long)J.pic * P.org_pic_c + (long)(J.y >> P.hs) * P.s_org_c + (J.x >> P.ws)); }

// the five luma predictors: grid (njobs, 5), block (job, mode) -> pred[(job * 5 + mode) * n0]; the block's first thread also writes the slot's SATD job and its
// mode-index bit-count job (round 6: was a launch of its own behind this one)
__global__ void k_intra_pred_luma(const pel *__restrict__ nb, const xeve_hip_intra_job *__restrict__ jobs, IntraK P, const unsigned char *__restrict__ mpm_row,
                                  pel *__restrict__ pred, xeve_hip_job *__restrict__ sj, xeve_hip_cu_bits_job *__restrict__ bj, int32_t *__restrict__ zero)
{
    const int j = blockIdx.x, m = blockIdx.y, t = j * SLOTS + m;
    if(threadIdx.x == 0) {
        if(t == 0) zero[0] = 0;
        const xeve_hip_intra_job J = jobs[j];
        sj[t] = xh_make_job(org_off_l(J, P), t * P.n0);
        xeve_hip_cu_bits_job b;
        b.coef_off[0] = b.coef_off[1] = b.coef_off[2] = 0, b.nnz[0] = b.nnz[1] = b.nnz[2] = 0, b.sbac = J.sbac;
        b.mvd[0][0] = b.mvd[0][1] = b.mvd[1][0] = b.mvd[1][1] = 0, b.refi[0] = b.refi[1] = -1, b.mvp_idx[0] = c_mpm[mpm_row[j]][m], b.mvp_idx[1] = 0;
        b.mode = XEVE_HIP_BITS_INTRA_DIR, b.dir_flag = 0, b.ctx_skip = J.ctx_skip, b.ctx_pred_mode = J.ctx_pred_mode;
        bj[t] = b;
    }
    intra_pred_block(nb, P, j, 0, m, pred + (long)t * P.n0);
}

// make_ipred_list (xeve_pintra.c:308-374) per CU; then the slots of the luma RDO: chain jobs and estimate indices
__global__ void k_intra_list(const xeve_hip_intra_job *__restrict__ jobs, IntraK P, const int32_t *__restrict__ satd, const unsigned *__restrict__ bits,
                             int *__restrict__ list, int *__restrict__ pred_cnt, xeve_hip_job *__restrict__ sj, int *__restrict__ est_idx)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_intra_job J = jobs[j];
    const int R = P.rdo_cnt;
    int      lst[SLOTS];
    double   cc[SLOTS];
    unsigned cs[SLOTS];
    for(int i = 0; i < SLOTS; i++) lst[i] = 0, cc[i] = MAX_COST, cs[i] = 0xFFFFFFFFu;
    for(int m = 0; m < 5; m++) {
        const unsigned sa = (unsigned)satd[j * SLOTS + m];
        const double cost = (double)sa + (double)(int)bits[j * SLOTS + m] * P.sqrt_lambda0;
        int shift = 0;
        while(shift < R && cost < cc[R - 1 - shift]) shift++;
        if(shift) {
            for(int k = 1; k < shift; k++) lst[R - k] = lst[R - 1 - k], cc[R - k] = cc[R - 1 - k], cs[R - k] = cs[R - 1 - k];
            lst[R - shift] = m, cc[R - shift] = cost, cs[R - shift] = sa;
        }
    }
    int cnt = R;
    for(int i = R - 1; i >= 1; i--) {
        if((double)cs[i] > (double)J.inter_satd * (1.2)) cnt--;
        else break;
    }
    pred_cnt[j] = cnt;
    for(int k = 0; k < SLOTS; k++) {
        const int t = j * SLOTS + k;
        list[t] = lst[k];
        sj[t] = xh_make_job(org_off_l(J, P), (j * SLOTS + lst[k]) * P.n0);
        est_idx[t] = J.sbac;
    }
}

// bit-count jobs of the luma slots (xeve_rdo_bit_cnt_cu_intra_luma); the slot's coefficient block is block t of the slot buffer
__global__ void k_intra_jobs2(const xeve_hip_intra_job *__restrict__ jobs, IntraK P, const unsigned char *__restrict__ mpm_row, const int *__restrict__ list,
                              const int *__restrict__ nnz, xeve_hip_cu_bits_job *__restrict__ bj)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= P.njobs * SLOTS) return;
    const int j = t / SLOTS;
    const xeve_hip_intra_job J = jobs[j];
    xeve_hip_cu_bits_job b;
    b.coef_off[0] = t * P.n0, b.coef_off[1] = b.coef_off[2] = 0, b.nnz[0] = nnz[t], b.nnz[1] = b.nnz[2] = 0, b.sbac = J.sbac;
    b.mvd[0][0] = b.mvd[0][1] = b.mvd[1][0] = b.mvd[1][1] = 0, b.refi[0] = b.refi[1] = -1, b.mvp_idx[0] = c_mpm[mpm_row[j]][list[t]], b.mvp_idx[1] = 0;
    b.mode = XEVE_HIP_BITS_INTRA_LUMA, b.dir_flag = 0, b.ctx_skip = J.ctx_skip, b.ctx_pred_mode = J.ctx_pred_mode;
    bj[t] = b;
}

// ---- 4. the luma decision (xeve_pintra.c:604-637); chroma chain jobs ---------------------------------------------------------------------------------------
__global__ void k_intra_pick_copy(const xeve_hip_intra_job *__restrict__ jobs, IntraK P, const int *__restrict__ list, const int *__restrict__ pred_cnt,
                                  const long *__restrict__ ssd, const unsigned *__restrict__ bits, const int *__restrict__ nnz, int *__restrict__ best_ipd,
                                  int *__restrict__ dist_y, int *__restrict__ nnz_y, xeve_hip_job *__restrict__ cj, int *__restrict__ est_idx_c,
                                  const int16_t *__restrict__ coef_s, const pel *__restrict__ rec_s, int16_t *__restrict__ coef, pel *__restrict__ rec,
                                  const pel *__restrict__ nb, pel *__restrict__ predc)
{   // one block per CU (round 6: the pick, the copy of the winner's blocks and the chroma prediction with its mode were three launches)
    __shared__ int s_slot, s_ipd;
    const int j = blockIdx.x;
    if(threadIdx.x == 0) {
        const xeve_hip_intra_job J = jobs[j];
        double best = MAX_COST;
        int    bs = 0;
        for(int k = 0; k < pred_cnt[j]; k++) {
            const int t = j * SLOTS + k;
            double cost = 0;
            cost += (double)ssd[2 * t + 1];
            cost += (double)(int)bits[t] * P.lambda0;
            if(cost < best) best = cost, bs = k;
        }
        const int t = j * SLOTS + bs;
        best_ipd[j] = list[t], dist_y[j] = (int)(double)ssd[2 * t + 1], nnz_y[j] = nnz[t];
        cj[j] = xh_make_job(org_off_c(J, P), j * P.n1), est_idx_c[j] = J.sbac;
        s_slot = t, s_ipd = list[t];
    }
    __syncthreads();
    // winner's coefficients and reconstruction -> the output blocks
    const long src = (long)s_slot * P.n0, dst = (long)j * P.n0;
    for(int t = threadIdx.x; t < P.n0; t += blockDim.x) coef[dst + t] = coef_s[src + t], rec[dst + t] = rec_s[src + t];
    // 5: chroma with the winner's mode -> predc[(c - 1) * njobs * n1 + job * n1]
    if(P.ncomp > 1) {
        const int ipd = s_ipd;
        for(int c = 1; c <= 2; c++) intra_pred_block(nb, P, j, c, ipd, predc + (long)(c - 1) * P.njobs * P.n1 + (long)j * P.n1);
    }
}

// ---- 6. the CU's bit-count job (xeve_rdo_bit_cnt_cu_intra) and the result --------------------------------------------------------------------------------
__global__ void k_intra_jobs3(const xeve_hip_intra_job *__restrict__ jobs, IntraK P, const unsigned char *__restrict__ mpm_row, const int *__restrict__ best_ipd,
                              const int *__restrict__ nnz_y, const int *__restrict__ nnz_u, const int *__restrict__ nnz_v, xeve_hip_cu_bits_job *__restrict__ bj)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    const xeve_hip_intra_job J = jobs[j];
    xeve_hip_cu_bits_job b;
    b.coef_off[0] = j * P.n0, b.coef_off[1] = P.njobs * P.n0 + j * P.n1, b.coef_off[2] = P.njobs * (P.n0 + P.n1) + j * P.n1;
    b.nnz[0] = nnz_y[j], b.nnz[1] = P.ncomp > 1 ? nnz_u[j] : 0, b.nnz[2] = P.ncomp > 1 ? nnz_v[j] : 0, b.sbac = J.sbac;
    b.mvd[0][0] = b.mvd[0][1] = b.mvd[1][0] = b.mvd[1][1] = 0, b.refi[0] = b.refi[1] = -1, b.mvp_idx[0] = c_mpm[mpm_row[j]][best_ipd[j]], b.mvp_idx[1] = 0;
    b.mode = XEVE_HIP_BITS_CU_INTRA, b.dir_flag = 0, b.ctx_skip = J.ctx_skip, b.ctx_pred_mode = J.ctx_pred_mode;
    bj[j] = b;
}

__global__ void k_intra_finish(IntraK P, const unsigned *__restrict__ bits, const int *__restrict__ best_ipd, const int *__restrict__ pred_cnt,
                               const int *__restrict__ dist_y, const int *__restrict__ nnz_y, const int *__restrict__ nnz_u, const int *__restrict__ nnz_v,
                               const long *__restrict__ ssd_u, const long *__restrict__ ssd_v, xeve_hip_intra_result *__restrict__ res)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= P.njobs) return;
    int dist_c = 0;
    if(P.ncomp > 1) { // (xeve_pintra.c:216-233, :266: the weighted sum as a double, then (s32))
        double c = 0;
        c += P.wgt[0] * (double)ssd_u[2 * j + 1];
        c += P.wgt[1] * (double)ssd_v[2 * j + 1];
        dist_c = (int)c;
    }
    double cost = (double)(int)bits[j] * P.lambda0; // (:684-693)
    cost += dist_y[j];
    if(P.ncomp > 1) cost += dist_c;
    xeve_hip_intra_result r;
    r.cost = cost, r.dist_cu = dist_y[j] + (P.ncomp > 1 ? dist_c : 0);
    r.nnz[0] = nnz_y[j], r.nnz[1] = P.ncomp > 1 ? nnz_u[j] : 0, r.nnz[2] = P.ncomp > 1 ? nnz_v[j] : 0;
    r.pred_cnt = pred_cnt[j], r.ipm[0] = (int8_t)best_ipd[j], r.ipm[1] = (int8_t)(P.ncomp > 1 ? best_ipd[j] : 0), r.pad_[0] = r.pad_[1] = 0;
    res[j] = r;
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------------------
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
struct IntraLayout {
    size_t nb, mpm, pred, predc, zero, sj, cj, satd, est_idx, est_idx_c, est, bj, bits, coef_s, rec_s, nnz_s, ssd_s, nnz_c[2], ssd_c[2], list, cnt, slot, ipd, dist_y, nnz_y, bitws, total;
};
static IntraLayout intra_layout(int njobs, int nstates, int n0, int n1)
{
    IntraLayout L;
    const size_t N = (size_t)njobs, S = N * SLOTS;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    L.nb = take(N * 3 * 2 * NB * sizeof(pel)), L.mpm = take(N), L.pred = take(S * n0 * sizeof(pel)), L.predc = take(2 * N * (size_t)n1 * sizeof(pel) + 16);
    L.zero = take(16), L.sj = take(S * sizeof(xeve_hip_job)), L.cj = take(N * sizeof(xeve_hip_job)), L.satd = take(S * 4), L.est_idx = take(S * 4), L.est_idx_c = take(N * 4);
    L.est = take((size_t)nstates * sizeof(xeve_hip_rdoq_est_full)), L.bj = take(S * sizeof(xeve_hip_cu_bits_job)), L.bits = take(S * 4);
    L.coef_s = take(S * n0 * 2 + 64), L.rec_s = take(S * n0 * sizeof(pel)), L.nnz_s = take(S * 4), L.ssd_s = take(S * 16);
    for(int k = 0; k < 2; k++) L.nnz_c[k] = take(N * 4), L.ssd_c[k] = take(N * 16);
    L.list = take(S * 4), L.cnt = take(N * 4), L.slot = take(N * 4), L.ipd = take(N * 4), L.dist_y = take(N * 4), L.nnz_y = take(N * 4);
    L.bitws = o;
    const size_t b1 = xeve_hip_cu_bits_workspace(njobs * SLOTS, S * n0), b2 = xeve_hip_cu_bits_workspace(njobs, N * ((size_t)n0 + 2 * (size_t)n1));
    L.total = o + al(b1 > b2 ? b1 : b2);
    return L;
}

static bool intra_params_ok(const xeve_hip_intra_params *p)
{
    return p && p->log2_cuw >= 2 && p->log2_cuw <= 6 && p->log2_cuh == p->log2_cuw && p->tool_iqt == 0 && p->bit_depth >= 8 && p->bit_depth <= 14 &&
           (p->chroma_format_idc == 0 || p->chroma_format_idc == 1 || p->chroma_format_idc == 3) && p->slice_type >= 0 && p->slice_type <= 2 && p->w_scu > 0 &&
           p->h_scu > 0;
}

extern "C" size_t xeve_hip_pintra_analyze_cu_workspace(int njobs, int nstates, const xeve_hip_intra_params *p)
{
    if(!intra_params_ok(p) || njobs < 0 || nstates < 0) return 0;
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, n0 = 1 << (2 * p->log2_cuw), n1 = idc ? n0 >> (ws + hs) : 0;
    return intra_layout(njobs, nstates, n0, n1).total;
}

static const int k_q_scale[6]  = {26214, 23302, 20560, 18396, 16384, 14764}; // xeve_quant_scale[0] (xeve_tq.c:37)
static const int k_dq_scale[6] = {40, 45, 51, 57, 64, 71};                   // xeve_tbl_dq_scale_b (xeve_tbl.c:237)

extern "C" int xeve_hip_pintra_analyze_cu_jobs(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l,
                                               int s_mod_c, const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, const int64_t *pic_elems,
                                               const xeve_hip_sbac *states, int nstates, const xeve_hip_intra_params *p, const xeve_hip_intra_job *jobs, int njobs,
                                               xeve_hip_intra_result *results, int16_t *coef, xeve_hip_pel *rec, xeve_hip_sbac *best, void *workspace,
                                               size_t workspace_bytes, void *stream)
{
    return xh_pintra_analyze_cu_jobs_x(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, pic_elems, states, nstates, p, jobs, njobs, results, coef, rec, best,
                                       workspace, workspace_bytes, stream, nullptr);
}

// est_shared: core->rdoq_est_* of every entry state, made by the caller (xeve_hip_rdoq_bit_est over `states`; the walk makes them once per node) -- NULL: made here
int xh_pintra_analyze_cu_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, const uint32_t *map_scu,
                                const int8_t *map_ipm, const uint8_t *map_tidx, const int64_t *pic_elems, const xeve_hip_sbac *states, int nstates,
                                const xeve_hip_intra_params *p, const xeve_hip_intra_job *jobs, int njobs, xeve_hip_intra_result *results, int16_t *coef, xeve_hip_pel *rec,
                                xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream, const void *est_shared)
{
    XH_ENTER();
    XH_REQUIRE(org && mod && map_scu && map_ipm && map_tidx && states && nstates > 0 && jobs && njobs >= 0 && results && coef && rec && workspace);
    XH_REQUIRE(intra_params_ok(p));
    XH_REQUIRE(org[0] && mod[0] && (p->chroma_format_idc == 0 || (org[1] && org[2] && mod[1] && mod[2])));
    if(njobs == 0) return XEVE_HIP_OK;
    XH_REQUIRE(workspace_bytes >= xeve_hip_pintra_analyze_cu_workspace(njobs, nstates, p));
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, bd = p->bit_depth, lw = p->log2_cuw;
    IntraK P;
    P.njobs = njobs, P.w = P.h = 1 << lw, P.n0 = 1 << (2 * lw), P.n1 = idc ? P.n0 >> (ws + hs) : 0, P.ncomp = idc ? 3 : 1, P.ws = ws, P.hs = hs, P.idc = idc;
    P.s_org_l = s_org_l, P.s_org_c = s_org_c, P.s_mod_l = s_mod_l, P.s_mod_c = s_mod_c, P.w_scu = p->w_scu, P.h_scu = p->h_scu;
    P.cip = p->constrained_intra_pred != 0, P.bd = bd, P.slice_type = p->slice_type, P.rdo_cnt = SLOTS; // (square CUs: IPD_RDO_CNT candidates)
    P.org_pic_l = pic_elems ? pic_elems[0] : 0, P.org_pic_c = pic_elems ? pic_elems[1] : 0, P.mod_pic_l = pic_elems ? pic_elems[2] : 0;
    P.mod_pic_c = pic_elems ? pic_elems[3] : 0, P.map_pic = pic_elems ? pic_elems[4] : 0;
    P.lambda0 = p->lambda[0], P.sqrt_lambda0 = p->sqrt_lambda0, P.wgt[0] = p->dist_chroma_weight[0], P.wgt[1] = p->dist_chroma_weight[1];
    for(int k = 0; k < P.ncomp; k++) XH_REQUIRE(p->qp[k] >= 0 && p->qp[k] <= 51 + 6 * (bd - 8));
    const IntraLayout L = intra_layout(njobs, nstates, P.n0, P.n1);
    char *W = (char *)workspace;
    pel  *nb = (pel *)(W + L.nb), *pred = (pel *)(W + L.pred), *predc = (pel *)(W + L.predc), *rec_s = (pel *)(W + L.rec_s);
    auto *mpm = (unsigned char *)(W + L.mpm);
    auto *zero = (int32_t *)(W + L.zero), *satd = (int32_t *)(W + L.satd);
    auto *sj = (xeve_hip_job *)(W + L.sj), *cj = (xeve_hip_job *)(W + L.cj);
    int  *est_idx = (int *)(W + L.est_idx), *est_idx_c = (int *)(W + L.est_idx_c), *nnz_s = (int *)(W + L.nnz_s), *list = (int *)(W + L.list), *cnt = (int *)(W + L.cnt);
    int  *ipd = (int *)(W + L.ipd), *dist_y = (int *)(W + L.dist_y), *nnz_y = (int *)(W + L.nnz_y);
    int  *nnz_c[2] = {(int *)(W + L.nnz_c[0]), (int *)(W + L.nnz_c[1])};
    long *ssd_s = (long *)(W + L.ssd_s), *ssd_c[2] = {(long *)(W + L.ssd_c[0]), (long *)(W + L.ssd_c[1])};
    auto *est = (xeve_hip_rdoq_est_full *)(W + L.est);
    auto *bj = (xeve_hip_cu_bits_job *)(W + L.bj);
    auto *bits = (unsigned *)(W + L.bits);
    auto *coef_s = (int16_t *)(W + L.coef_s);
    hipStream_t st = (hipStream_t)stream;
    const int   S = njobs * SLOTS, GS = (S + 255) / 256, GJ = (njobs + 255) / 256;
    const size_t bws = workspace_bytes - L.bitws;
    int rc;
    xeve_hip_cu_bits_params bp;
    bp.log2_cuw = bp.log2_cuh = lw, bp.slice_type = p->slice_type, bp.num_refp[0] = bp.num_refp[1] = 0, bp.cm_init = 0, bp.chroma_format_idc = idc;

    // 1, 2: neighbours, rank row, the five predictors
    k_intra_nbr<<<dim3(njobs, 3), 128, 0, st>>>(mod[0], mod[1], mod[2], map_scu, map_ipm, map_tidx, jobs, P, nb, mpm);
    k_intra_pred_luma<<<dim3(njobs, 5), P.n0 >= 256 ? 256 : 64, 0, st>>>(nb, jobs, P, mpm, pred, sj, bj, zero);
    // 3: SATD, mode bits, the list
    rc = xeve_hip_satd_jobs(org[0], s_org_l, pred, P.w, sj, S, zero, 1, P.w, P.h, bd, satd, stream);
    if(rc != XEVE_HIP_OK) return rc;
    rc = xh_cu_bits_jobs_round(nullptr, 0, states, bj, S, &bp, W + L.bitws, bws, bits, nullptr, 0, 0, stream);
    if(rc != XEVE_HIP_OK) return rc;
    k_intra_list<<<GJ, 256, 0, st>>>(jobs, P, satd, bits, list, cnt, sj, est_idx);
    // 4: the luma RDO of the slots
    if(est_shared) est = (xeve_hip_rdoq_est_full *)est_shared;
    else {
        rc = xeve_hip_rdoq_bit_est(states, nstates, est, stream); // core->rdoq_est_* of mode_coding_unit (xeve_mode.c:792)
        if(rc != XEVE_HIP_OK) return rc;
    }
    {
        const int q = p->qp[0];
        rc = xh_residual_rdoq(org[0], s_org_l, pred, P.w, sj, S, lw, lw, bd, q, k_q_scale[q % 6], k_dq_scale[q % 6] << (q / 6), p->slice_type == 2, 1, p->lambda[0], 0,
                              p->tool_iqt, est, est_idx, coef_s, rec_s, -P.w, nnz_s, (int64_t *)ssd_s, st);
        if(rc != XEVE_HIP_OK) return rc;
    }
    k_intra_jobs2<<<GS, 256, 0, st>>>(jobs, P, mpm, list, nnz_s, bj);
    rc = xh_cu_bits_jobs_round(coef_s, (size_t)S * P.n0, states, bj, S, &bp, W + L.bitws, bws, bits, nullptr, 0, 0, stream);
    if(rc != XEVE_HIP_OK) return rc;
    k_intra_pick_copy<<<njobs, P.n0 >= 256 ? 256 : 64, 0, st>>>(jobs, P, list, cnt, ssd_s, bits, nnz_s, ipd, dist_y, nnz_y, cj, est_idx_c, coef_s, rec_s, coef, rec, nb, predc);
    // 5: chroma with the winner's mode (predicted by the launch above)
    if(P.ncomp > 1) {
        for(int k = 1; k <= 2; k++) {
            const int q = p->qp[k];
            rc = xh_residual_rdoq(org[k], s_org_c, predc + (size_t)(k - 1) * njobs * P.n1, P.w >> ws, cj, njobs, lw - ws, lw - hs, bd, q, k_q_scale[q % 6],
                                  k_dq_scale[q % 6] << (q / 6), p->slice_type == 2, 1, p->lambda[k], k, p->tool_iqt, est, est_idx_c,
                                  coef + (size_t)njobs * (P.n0 + (size_t)(k - 1) * P.n1), rec + (size_t)njobs * (P.n0 + (size_t)(k - 1) * P.n1), -(P.w >> ws),
                                  nnz_c[k - 1], (int64_t *)ssd_c[k - 1], st);
            if(rc != XEVE_HIP_OK) return rc;
        }
    }
    // 6: the CU's cost and core->s_temp_best
    k_intra_jobs3<<<GJ, 256, 0, st>>>(jobs, P, mpm, ipd, nnz_y, nnz_c[0], nnz_c[1], bj);
    rc = xh_cu_bits_jobs_round(coef, (size_t)njobs * (P.n0 + 2 * (size_t)P.n1), states, bj, njobs, &bp, W + L.bitws, bws, bits, best, best != nullptr, 0, stream);
    if(rc != XEVE_HIP_OK) return rc;
    k_intra_finish<<<GJ, 256, 0, st>>>(P, bits, ipd, cnt, dist_y, nnz_y, nnz_c[0], nnz_c[1], ssd_c[0], ssd_c[1], results);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- host-memory form of ONE call of ctx->fn_pintra_analyze_cu (the table layer's style: stage, launch, synchronise) ---------------------------------------------
// Every pointer is HOST memory: org / mod = sample (0, 0) of the original picture's planes and of the picture being reconstructed (pi->o, pi->m), the maps =
// ctx->map_scu / map_ipm / map_tidx.  Only what the analysis reads is moved: the CU's block of the original, the line above and the column left of the CU in the
// mode picture (cuw + cuh samples each, clipped to the picture), the map entries of the 4x4 units those samples lie in.  They are laid out as a small local
// picture (the CU at unit (1, 1), or on the edge where the real CU is on the picture's edge) so that the batched entry point runs unchanged on it.

extern "C" int xeve_hip_pintra_analyze_cu_host(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c,
                                               const uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, const xeve_hip_sbac *state,
                                               const xeve_hip_intra_params *p, const xeve_hip_intra_job *job, xeve_hip_intra_result *result, int16_t *coef_y,
                                               int16_t *coef_u, int16_t *coef_v, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_sbac *best)
{
    XH_ENTER();
    XH_REQUIRE(org && mod && map_scu && map_ipm && map_tidx && state && job && result && coef_y && rec_y && best && intra_params_ok(p));
    const int idc = p->chroma_format_idc, ws = idc <= 2, hs = idc <= 1, ncomp = idc ? 3 : 1, lw = p->log2_cuw;
    XH_REQUIRE(org[0] && mod[0] && (!idc || (org[1] && org[2] && mod[1] && mod[2] && coef_u && coef_v && rec_u && rec_v)));
    const int cu = 1 << lw, n = cu >> 2, x_scu = job->x >> 2, y_scu = job->y >> 2, scup = y_scu * p->w_scu + x_scu;
    XH_REQUIRE(job->x >= 0 && job->y >= 0 && (job->x & 3) == 0 && (job->y & 3) == 0 && x_scu + n <= p->w_scu && y_scu + n <= p->h_scu);
    const int lx = x_scu > 0, ly = y_scu > 0, nw = std::min(2 * n, p->w_scu - x_scu), nh = std::min(2 * n, p->h_scu - y_scu), Wl = lx + nw, Hl = ly + nh;
    const size_t n0 = (size_t)cu * cu, n1 = idc ? n0 >> (ws + hs) : 0, nmap = (size_t)Wl * Hl;
    // staging layout (identical in the pinned buffer and in the device arena)
    const int    pw[3] = {Wl * 4, Wl * (4 >> ws), Wl * (4 >> ws)}, ph[3] = {Hl * 4, Hl * (4 >> hs), Hl * (4 >> hs)};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 63) & ~(size_t)63; return at; };
    const size_t o_job = take(sizeof(*job)), o_st = take(sizeof(*state)), o_scu = take(nmap * 4), o_ipm = take(nmap), o_tidx = take(nmap);
    size_t o_org[3] = {0, 0, 0}, o_mod[3] = {0, 0, 0};
    for(int c = 0; c < ncomp; c++) o_org[c] = take((size_t)pw[c] * ph[c] * 2), o_mod[c] = take((size_t)pw[c] * ph[c] * 2);
    const size_t in_bytes = o;
    const size_t o_res = take(sizeof(*result)), o_best = take(sizeof(*best)), o_coef = take((n0 + 2 * n1) * 2), o_rec = take((n0 + 2 * n1) * 2);
    const size_t io_bytes = o;
    xeve_hip_intra_params pl = *p;
    pl.w_scu = Wl, pl.h_scu = Hl;
    const size_t wsb = xeve_hip_pintra_analyze_cu_workspace(1, 1, &pl);
    static thread_local XhHostArena C;
    int rc = C.ensure(io_bytes, wsb);
    if(rc != XEVE_HIP_OK) return rc;
    char *H = C.pin, *D = C.dev;
    // the job in local coordinates, the entry state, the maps
    xeve_hip_intra_job jl = *job;
    jl.x = 4 * lx, jl.y = 4 * ly, jl.sbac = 0, jl.pic = 0;
    memcpy(H + o_job, &jl, sizeof(jl)), memcpy(H + o_st, state, sizeof(*state));
    memset(H + o_scu, 0, o_org[0] - o_scu);
    auto *ls = (uint32_t *)(H + o_scu);
    auto *li = (int8_t *)(H + o_ipm);
    auto *lt = (uint8_t *)(H + o_tidx);
    auto unit = [&](int lu, int gu) { ls[lu] = map_scu[gu], li[lu] = map_ipm[gu], lt[lu] = map_tidx[gu]; };
    unit(ly * Wl + lx, scup);
    if(ly) for(int i = -lx; i < nw; i++) unit((ly - 1) * Wl + lx + i, scup - p->w_scu + i);
    if(lx) for(int i = 0; i < nh; i++) unit((ly + i) * Wl + lx - 1, scup - 1 + i * p->w_scu);
    // the planes: the original's block; the line above and the column to the left of the mode picture
    for(int c = 0; c < ncomp; c++) {
        const int sx = c ? ws : 0, sy = c ? hs : 0, uw = 4 >> sx, uh = 4 >> sy, bw = cu >> sx, bh = cu >> sy;
        const int so = c ? s_org_c : s_org_l, sm = c ? s_mod_c : s_mod_l, x = job->x >> sx, y = job->y >> sy, xl = lx * uw, yl = ly * uh;
        pel *lo = (pel *)(H + o_org[c]), *lm = (pel *)(H + o_mod[c]);
        memset(lo, 0, (size_t)pw[c] * ph[c] * 2), memset(lm, 0, (size_t)pw[c] * ph[c] * 2);
        for(int r = 0; r < bh; r++) memcpy(lo + (size_t)(yl + r) * pw[c] + xl, org[c] + (size_t)(y + r) * so + x, sizeof(pel) * bw);
        if(ly) memcpy(lm + (size_t)(yl - 1) * pw[c], mod[c] + (size_t)(y - 1) * sm + (x - xl), sizeof(pel) * (size_t)(xl + nw * uw));
        if(lx) for(int r = 0; r < nh * uh; r++) lm[(size_t)(yl + r) * pw[c] + xl - 1] = mod[c][(size_t)(y + r) * sm + x - 1];
    }
    XH_HIP(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, C.st));
    const pel *d_org[3] = {(const pel *)(D + o_org[0]), idc ? (const pel *)(D + o_org[1]) : nullptr, idc ? (const pel *)(D + o_org[2]) : nullptr};
    const pel *d_mod[3] = {(const pel *)(D + o_mod[0]), idc ? (const pel *)(D + o_mod[1]) : nullptr, idc ? (const pel *)(D + o_mod[2]) : nullptr};
    char *d_ws = D + ((io_bytes + 255) & ~(size_t)255);
    rc = xeve_hip_pintra_analyze_cu_jobs(d_org, pw[0], pw[1], d_mod, pw[0], pw[1], (const uint32_t *)(D + o_scu), (const int8_t *)(D + o_ipm), (const uint8_t *)(D + o_tidx),
                                         nullptr, (const xeve_hip_sbac *)(D + o_st), 1, &pl, (const xeve_hip_intra_job *)(D + o_job), 1,
                                         (xeve_hip_intra_result *)(D + o_res), (int16_t *)(D + o_coef), (pel *)(D + o_rec), (xeve_hip_sbac *)(D + o_best), d_ws,
                                         C.dev_bytes - (size_t)(d_ws - D), C.st);
    if(rc != XEVE_HIP_OK) return rc;
    XH_HIP(hipMemcpyAsync(H + o_res, D + o_res, io_bytes - o_res, hipMemcpyDeviceToHost, C.st));
    XH_HIP(hipStreamSynchronize(C.st));
    memcpy(result, H + o_res, sizeof(*result)), memcpy(best, H + o_best, sizeof(*best));
    const int16_t *hc = (const int16_t *)(H + o_coef);
    const pel     *hr = (const pel *)(H + o_rec);
    memcpy(coef_y, hc, n0 * 2), memcpy(rec_y, hr, n0 * 2);
    if(idc) {
        memcpy(coef_u, hc + n0, n1 * 2), memcpy(coef_v, hc + n0 + n1, n1 * 2);
        memcpy(rec_u, hr + n0, n1 * 2), memcpy(rec_v, hr + n0 + n1, n1 * 2);
    }
    return XEVE_HIP_OK;
}
