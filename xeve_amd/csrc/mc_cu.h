// xeve_amd/csrc/mc_cu.h -- the per-CU front half of xeve_mc (src_base/xeve_mc.c:465-610) as a device function, so that the kernel that PRODUCES a CU's prediction job
// (k_rdo_prep, k_skip_prep, k_bi_head, k_inter_decide, ...) also clips its vectors and writes the per-list interpolation jobs: one launch less per prediction on the
// walk's dependent launch chain (round 6; the stand-alone kernel k_cu_mc_prep of mc.hip calls the same function for the C-ABI's own callers).
#pragma once
#include "xh_common.h"

#define XH_MAX_REF 8
struct CuMcK {
    int pic_w, pic_h, w, h, cw, ch, wfac, hfac, nref[2], poc[2][XH_MAX_REF];
    int vh; // a batch of pictures stacked vertically (xh_common.h): 0 = one picture
};
// what a producer kernel needs to prepare the interpolation jobs of xh_mc_cu_jobs_x(..., XH_MC_PREPPED): parameters + where in the prediction workspace they go
struct CuMcPrep {
    CuMcK            P;
    int              njobs;
    uint8_t         *mode; // per job: 0 list 0 alone (or nothing), 1 average the two lists, 2 list 1 alone
    xeve_hip_mc_job *jl, *jc; // [list][job] luma / chroma interpolation jobs
    pel             *p1[3];   // where list 1's predictions land before the last kernel averages them into / copies them over the caller's buffers
};
enum { XH_MC_PREPPED = 1, XH_MC_LUMA_ONLY = 2, XH_MC_NO_COMBINE = 4 };
// host (mc.hip): fills *out for a call of xh_mc_cu_jobs_x with the same arguments and workspace
int xh_mc_cu_prep_params(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int pic_w, int pic_h, int njobs, int w, int h, int chroma_format_idc, void *workspace,
                         size_t workspace_bytes, CuMcPrep *out);
int xh_mc_cu_jobs_x(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h, const xeve_hip_cu_mc_job *jobs, int njobs, int w, int h,
                    int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], pel *pred_y, pel *pred_u, pel *pred_v,
                    void *workspace, size_t workspace_bytes, void *stream, int flags);

#ifdef __HIPCC__
// per job: clip both vectors (xeve_mv_clip), filter variant from the UNCLIPPED vector's fraction, position from the clipped one, drop list 1 when it repeats list 0
// (same POC, same clipped vector)
__device__ __forceinline__ void xh_cu_mc_prep_one(const xeve_hip_cu_mc_job &J, int j, const CuMcPrep &C)
{
    const CuMcK &P = C.P;
    const int njobs = C.njobs;
    const int yb = xh_vh_base(J.y, P.vh); // the vectors are clipped against the job's own picture; the rows of its picture in the stack are added to the positions below
    const int x4 = J.x << 2, y4 = (J.y - yb) << 2, w4 = P.w << 2, h4 = P.h << 2;
    const int min_c = -(128 << 2), max_x = (P.pic_w - 1 + 128) << 2, max_y = (P.pic_h - 1 + 128) << 2; // MAX_CU_SIZE margin
    int  mvt[2][2];
    bool valid[2];
#pragma unroll
    for(int l = 0; l < 2; l++) {
        valid[l] = J.refi[l] >= 0;
        int mx = J.mv[l][0], my = J.mv[l][1];
        if(valid[l]) {
            if(x4 + J.mv[l][0] < min_c) mx = (int16_t)(min_c - x4);
            if(y4 + J.mv[l][1] < min_c) my = (int16_t)(min_c - y4);
            if(x4 + J.mv[l][0] + w4 - 4 > max_x) mx = (int16_t)(max_x - x4 - w4 + 4);
            if(y4 + J.mv[l][1] + h4 - 4 > max_y) my = (int16_t)(max_y - y4 - h4 + 4);
        }
        mvt[l][0] = mx, mvt[l][1] = my;
    }
    bool use1 = valid[1];
    if(valid[0] && valid[1] && P.poc[0][J.refi[0]] == P.poc[1][J.refi[1]] && mvt[0][0] == mvt[1][0] && mvt[0][1] == mvt[1][1]) use1 = false;
    C.mode[j] = (uint8_t)(use1 ? (valid[0] ? 1 : 2) : 0); // 1 average the two, 2 list 1 alone: copy it over
#pragma unroll
    for(int l = 0; l < 2; l++) {
        const bool on = l == 0 ? valid[0] : use1;
        const int  gx = (x4 + mvt[l][0]) << 2, gy = (y4 + (yb << 2) + mvt[l][1]) << 2;
        xeve_hip_mc_job a, c;
        a.gmv_x = gx, a.gmv_y = gy, a.pred_off = j * P.w * P.h;
        a.frac = ((J.mv[l][0] & 3) ? 1 : 0) | ((J.mv[l][1] & 3) ? 2 : 0);
        c.gmv_x = gx * P.wfac, c.gmv_y = gy * P.hfac, c.pred_off = j * P.cw * P.ch;
        c.frac = ((J.mv[l][0] & 7) ? 1 : 0) | ((J.mv[l][1] & 7) ? 2 : 0);
        // one job array per list; the reference picture rides in frac bits 3.. (PlaneTab); bit 2 switches a job off
        const int sel = (on && J.refi[l] < P.nref[l]) ? (J.refi[l] << 3) : 4;
        a.frac |= sel, c.frac |= sel;
        C.jl[(size_t)l * njobs + j] = a, C.jc[(size_t)l * njobs + j] = c;
    }
}
#endif
