// xeve_amd/csrc/affine.hip -- Main profile: affine motion compensation of a batch of CUs on reference pictures resident in HBM ("affine MC").
// reference: xeve_affine_mc (src_main/xevem_mc.c:2236-2339) = derive_affine_subblock_size_bi (xevem_util.c:1203-1272), per list in use xeve_affine_mc_lc (:1671-1915: the
// CU through the Main 8- / 4-tap filters at one vector, or -- sub-blocks below 8 -- the enhanced interpolation filter xeve_eif_mc, :2123-2234), the average of two lists.
// What the lanes compute is affine_core.h (pinned on the host: tests/native/affine_host.cpp); this file is staging and data movement.
//   One workgroup per CU.  Per list and component the reference samples the CU can touch are staged ONCE in LDS -- the translated window with its filter margin (at most
//   135 x 135), or the enhanced filter's bilinear samples of the positions -1 .. w / h (at most 130 x 130: every output reads nine of them) -- and the lanes take the CU's
//   samples in turn; the second list's value is averaged into what the SAME lane wrote for the first.  HBM traffic = the window once + the prediction once.
#include <cstring>
#include "affine_core.h"
#include "xh_common.h"

namespace {

constexpr int LDS_PITCH = 136, MAXREF = XEVE_HIP_MAX_REFP;

// the Main filters: luma 1/16 sample, chroma 1/32 sample (ISO/IEC 23094-1 8.5.4.3.2 / .3; the reference's copy: xevem_tbl_mc_l_coeff / _c_coeff, xevem_mc.c:48-104)
__device__ __constant__ int16_t c_main_l[16][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0},        {0, 1, -3, 63, 4, -2, 1, 0},      {-1, 2, -5, 62, 8, -3, 1, 0},     {-1, 3, -8, 60, 13, -4, 1, 0},
    {-1, 4, -10, 58, 17, -5, 1, 0},   {-1, 4, -11, 52, 26, -8, 3, -1},  {-1, 3, -9, 47, 31, -10, 4, -1},  {-1, 4, -11, 45, 34, -10, 4, -1},
    {-1, 4, -11, 40, 40, -11, 4, -1}, {-1, 4, -10, 34, 45, -11, 4, -1}, {-1, 4, -10, 31, 47, -9, 3, -1},  {-1, 3, -8, 26, 52, -11, 4, -1},
    {0, 1, -5, 17, 58, -10, 4, -1},   {0, 1, -4, 13, 60, -8, 3, -1},    {0, 1, -3, 8, 62, -5, 2, -1},     {0, 1, -2, 4, 63, -3, 1, 0}};
__device__ __constant__ int16_t c_main_c[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2}, {-4, 56, 14, -2}, {-4, 55, 15, -2},
    {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3}, {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4},
    {-4, 36, 36, -4}, {-4, 33, 39, -4}, {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},  {0, 4, 62, -2},   {0, 2, 63, -1}};

struct RefTab {
    xeve_hip_refpic r[2 * MAXREF]; // [refi * 2 + list]
};
struct GlobalAt { // a component's reference samples relative to the CU's first sample
    const pel *p;
    long       s;
    __device__ int operator()(int dy, int dx) const { return p[(long)dy * s + dx]; }
};
struct LdsAt {
    const pel *p;
    int        pitch;
    __device__ int operator()(int r, int c) const { return p[r * pitch + c]; }
};

// one component (c: 0 luma, 1 / 2 chroma) of one list's prediction of the CU at (x, y): the reference samples it can touch are staged in LDS (buf) once, then the lanes of the
// block take the component's samples in turn and write dst (dense); nth: the value is averaged into what the same lane wrote for the first list.  org: the component's
// reference sample at the CU's first sample, s its pitch.
__device__ void mc_component(const xaff::Model &m, bool eif, bool mem_ok, int sub_w, int sub_h, int x, int y, int w, int h, int pic_w, int pic_h, int c, const pel *org, long s,
                             pel *dst, int nth, int bit_depth, pel *buf)
{
    const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h;
    __syncthreads(); // (the previous component's readers are done with buf)
    if(eif) {
        int mx[2], mn[2];
        xaff::eif_range(m, mem_ok, x, y, w, h, pic_w, pic_h, mx, mn);
        const xaff::Eif e = xaff::eif_component(m, mx, mn, c != 0);
        const int pitch = cw + 2;
        for(int i = threadIdx.x; i < pitch * (ch + 2); i += blockDim.x) {
            const int r = i / pitch, q = i - r * pitch;
            buf[r * pitch + q] = (pel)xaff::eif_bilinear(GlobalAt{org, s}, e, q - 1, r - 1, bit_depth);
        }
        __syncthreads();
        for(int i = threadIdx.x; i < cw * ch; i += blockDim.x) {
            const int py = i / cw, px = i - py * cw, v = xaff::eif_out(LdsAt{buf, pitch}, px, py, bit_depth);
            dst[i] = (pel)(nth ? (dst[i] + v + 1) >> 1 : v);
        }
    }
    else {
        int th, tv, oh, ov;
        xaff::block_vector(m, sub_w, sub_h, x, y, w, h, pic_w, pic_h, th, tv, oh, ov);
        const int fs = c ? 5 : 4, fm = (1 << fs) - 1, taps = c ? 4 : 8, back = taps / 2 - 1, pitch = cw + taps - 1;
        const pel *win = org + (long)((tv >> fs) - back) * s + (th >> fs) - back; // the translated CU's first sample, `back` rows / columns earlier
        for(int i = threadIdx.x; i < pitch * (ch + taps - 1); i += blockDim.x) {
            const int r = i / pitch, q = i - r * pitch;
            buf[r * pitch + q] = win[(long)r * s + q];
        }
        __syncthreads();
        const bool     fx = (oh & fm) != 0, fy = (ov & fm) != 0;
        const int16_t *cx = c ? c_main_c[th & fm] : c_main_l[th & fm], *cy = c ? c_main_c[tv & fm] : c_main_l[tv & fm];
        for(int i = threadIdx.x; i < cw * ch; i += blockDim.x) {
            const int   py = i / cw, px = i - py * cw;
            const LdsAt at{buf + (py + back) * pitch + px + back, pitch};
            const int   v = c ? xaff::mc_sample<4>(at, fx, fy, cx, cy, bit_depth) : xaff::mc_sample<8>(at, fx, fy, cx, cy, bit_depth);
            dst[i] = (pel)(nth ? (dst[i] + v + 1) >> 1 : v);
        }
    }
}

__global__ void __launch_bounds__(256) k_affine_mc(RefTab tab, long s_l, long s_c, int pic_w, int pic_h, const xeve_hip_affine_job *__restrict__ jobs, int w, int h, int bit_depth,
                                                   pel *__restrict__ pred_y, pel *__restrict__ pred_u, pel *__restrict__ pred_v)
{
    __shared__ pel buf[LDS_PITCH * LDS_PITCH];
    const xeve_hip_affine_job jb = jobs[blockIdx.x];
    int  sub_w, sub_h;
    bool mem_ok;
    xaff::subblock_size(jb.refi, jb.mv, jb.vertex_num, w, h, sub_w, sub_h, mem_ok);
    const bool eif = sub_w < 8 || sub_h < 8;
    int nth = 0;
    for(int l = 0; l < 2; l++) {
        if(jb.refi[l] < 0) continue;
        const xeve_hip_refpic rp = tab.r[jb.refi[l] * 2 + l];
        const xaff::Model m = xaff::model(jb.mv[l], w, h, jb.vertex_num);
        for(int c = 0; c < 3; c++) {
            const int  cw = c ? w >> 1 : w, ch = c ? h >> 1 : h;
            const long s = c ? s_c : s_l;
            const pel *org = (c == 0 ? rp.y : c == 1 ? rp.u : rp.v) + (long)(c ? jb.y >> 1 : jb.y) * s + (c ? jb.x >> 1 : jb.x);
            mc_component(m, eif, mem_ok, sub_w, sub_h, jb.x, jb.y, w, h, pic_w, pic_h, c, org, s, (c == 0 ? pred_y : c == 1 ? pred_u : pred_v) + (size_t)blockIdx.x * cw * ch, nth,
                         bit_depth, buf);
        }
        nth++;
    }
}

// ---- the affine gradient search (pinter_affine_me_gradient, src_main/xevem_pinter.c:4290-4501): ONE workgroup carries a whole search -- the start compensation, then per
// round the normal equations, the update and the next compensation -- with the CU's prediction resident in LDS from the compensation that writes it to the SATD, the error and
// the Sobel derivatives that read it (the reference keeps pred / error / two derivative planes in memory and makes five passes over them per round; here a round is the
// compensation's pass, one pass for the SATD and ONE fused pass that forms error, both derivatives and the 27 (14) sums of the normal equations in registers).  The scalar
// steps (vector bits, solve_equal in double, the rounded update, the comparison) are lane 0's: affine_core.h me_*, held to the reference's goldens on the host as well.
//   HBM traffic per round: the reference window ((w + 7) x (h + 7) samples) + the original twice (SATD, error); algorithmic bytes per search in DESIGN.md 5e.
struct AffMeArgs {
    RefTab         tab;
    long           s_l, s_org;
    const int16_t *org;        // the picture's original luma plane (sample (0, 0)); with org_dense: unused
    const int16_t *org_blocks; // dense w x h block per job: the bi-prediction target of the jobs that have bi set (every job's original with org_dense)
    xeve_hip_affine_me_job *jobs;
    int      pic_w, pic_h, w, h, bit_depth, org_dense, num_refp[2];
    uint32_t lambda_mv;
};
__device__ __forceinline__ long long wave_sum64(long long v)
{
#pragma unroll
    for(int m = 1; m < 64; m <<= 1) {
        const int lo = __shfl_xor((int)(unsigned)(unsigned long long)v, m, 64), hi = __shfl_xor((int)((unsigned long long)v >> 32), m, 64);
        v += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    return v;
}
// SATD of the CU (xeve_had's tiling for CUs of 16 and more, xeve_sad.c:1051-1135: TW x TH = 16x8 when wider than high, 8x16 when higher, 8x8 when square): a lane per tile
// ROW -- the row's Hadamard in registers, the columns' across the TH lanes of the tile by lane exchanges --, the tile's sum normalised on its first lane.  Returns the block's
// total on every lane.  red: 4 ints of LDS.
template <int TW, int TH> __device__ int satd_block(const int16_t *org, long s_org, const pel *pred, int w, int h, int *red)
{
    const int tiles_x = w / TW, rows = (w / TW) * h; // rows of tiles in all: a multiple of TH (and lanes of a tile are consecutive)
    int acc = 0;
    for(int base = 0; base < rows; base += blockDim.x) {
        const int  item = base + threadIdx.x;
        const bool on = item < rows;
        const int  tile = item / TH, trow = item % TH, ty = tile / tiles_x, tx = tile - ty * tiles_x, r = ty * TH + trow, c0 = tx * TW;
        int v[TW];
#pragma unroll
        for(int k = 0; k < TW; k++) v[k] = on ? (int)org[(long)r * s_org + c0 + k] - (int)pred[r * w + c0 + k] : 0;
#pragma unroll
        for(int len = 1; len < TW; len <<= 1)
#pragma unroll
            for(int i = 0; i < TW; i++)
                if(!(i & len)) {
                    const int a = v[i], b = v[i + len];
                    v[i] = a + b, v[i + len] = a - b;
                }
#pragma unroll
        for(int len = 1; len < TH; len <<= 1)
#pragma unroll
            for(int k = 0; k < TW; k++) {
                const int o = __shfl_xor(v[k], len, 64);
                v[k] = (trow & len) ? o - v[k] : v[k] + o;
            }
        int sa = 0;
#pragma unroll
        for(int k = 0; k < TW; k++) {
            int a = v[k] < 0 ? -v[k] : v[k];
            if(k == 0 && trow == 0) a >>= 2; // the tile's DC
            sa += a;
        }
#pragma unroll
        for(int len = 1; len < TH; len <<= 1) sa += __shfl_xor(sa, len, 64);
        if(on && trow == 0) acc += TW == TH ? (sa + 2) >> 2 : (int)(sa / (2.0 * 2.8284271247461903)); // (xeve_sad.c:602; :748, :885: / (2 * sqrt(8)) in double)
    }
#pragma unroll
    for(int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
    __syncthreads(); // (red's previous readers)
    if((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
// solve_equal (xevem_pinter.c:4213-4255) on the matrix in LDS by the block's first lanes: per pivot the column's largest row (every lane finds the same), the two rows swapped by
// NP + 1 lanes, then every element below the pivot row eliminated by a lane of its own -- the same expression eq[j][k] - eq[i][k] * eq[j][i - 1] / eq[i][i - 1] on the same
// operands as the reference's loops (within a pivot step no element it reads is one it writes) -- and the back substitution by lane 0.  A serial solve is 76 dependent
// double divisions for three control points; this is 5 + 6.
template <int NP> __device__ void solve_by_lanes(double (*eq)[7], double *para)
{
    const int t = threadIdx.x;
    for(int i = 1; i < NP; i++) {
        double best = xaff::me_abs(eq[i][i - 1]);
        int    at = i;
        for(int j = i + 1; j < NP + 1; j++) {
            const double v = xaff::me_abs(eq[j][i - 1]);
            if(v > best) best = v, at = j;
        }
        const bool sw = at != i && t <= NP;
        double a = 0, b = 0;
        if(sw) a = eq[i][t], b = eq[at][t];
        __syncthreads();
        if(sw) eq[i][t] = b, eq[at][t] = a;
        __syncthreads();
        const int  nk = NP + 1 - i, nj = NP - i; // columns i .. NP of rows i + 1 .. NP
        const bool on = t < nj * nk;
        const int  j = i + 1 + (on ? t / nk : 0), k = i + (on ? t % nk : 0);
        double v = 0;
        if(on) v = eq[j][k] - eq[i][k] * eq[j][i - 1] / eq[i][i - 1];
        __syncthreads();
        if(on) eq[j][k] = v;
        __syncthreads();
    }
    if(t == 0) {
        para[NP - 1] = eq[NP][NP] / eq[NP][NP - 1];
        for(int i = NP - 2; i >= 0; i--) {
            double acc = 0;
            for(int j = i + 1; j < NP; j++) acc += eq[i + 1][j] * para[j];
            para[i] = (eq[i + 1][NP] - acc) / eq[i + 1][i];
        }
    }
}
template <int VN> __global__ void __launch_bounds__(256) k_affine_me(AffMeArgs A)
{
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr int NP = 2 * VN, NS = NP * (NP + 1) / 2 + NP; // the upper triangle of the symmetric part + the right-hand side
    const int w = A.w, h = A.h, n = w * h;
    pel       *pred = reinterpret_cast<pel *>(lds), *buf = pred + n;
    const int  nbuf = (w + 7) * (h + 7);
    long long *red64 = reinterpret_cast<long long *>(lds + (((size_t)(n + nbuf) * sizeof(pel) + 15) & ~(size_t)15)); // [4][NS]
    __shared__ int     red[4], s_stop;
    __shared__ int16_t s_mvt[3][2];
    __shared__ double  s_eq[7][7], s_para[6];
    xeve_hip_affine_me_job *const jp = A.jobs + blockIdx.x;
    const xeve_hip_affine_me_job  jb = *jp;
    if(jb.vertex_num != VN) return; // (the other model's launch carries it)
    const int  bi = jb.bi, ri = jb.refi, num_refp = A.num_refp[jb.list & 1];
    const bool dense = bi || A.org_dense;
    const int16_t *org = dense ? A.org_blocks + (size_t)blockIdx.x * n : A.org + (long)jb.y * A.s_org + jb.x;
    const long     s_org = dense ? w : A.s_org;
    const pel     *ref = A.tab.r[ri * 2 + (jb.list & 1)].y + (long)jb.y * A.s_l + jb.x;
    int16_t mvt[3][2], best[3][2];
    for(int v = 0; v < 3; v++) mvt[v][0] = best[v][0] = jb.mv[v][0], mvt[v][1] = best[v][1] = jb.mv[v][1];
    auto compensate = [&]() { // xeve_affine_mc_l (xevem_mc.c:1532-1669): the sub-block size of ONE list's model, the luma plane alone
        const int8_t one[2] = {0, -1};
        int16_t two[2][3][2];
        for(int v = 0; v < 3; v++) two[0][v][0] = mvt[v][0], two[0][v][1] = mvt[v][1], two[1][v][0] = two[1][v][1] = 0;
        int  sub_w, sub_h;
        bool mem_ok;
        xaff::subblock_size(one, two, VN, w, h, sub_w, sub_h, mem_ok);
        mc_component(xaff::model(mvt, w, h, VN), sub_w < 8 || sub_h < 8, mem_ok, sub_w, sub_h, jb.x, jb.y, w, h, A.pic_w, A.pic_h, 0, ref, A.s_l, pred, 0, A.bit_depth, buf);
        __syncthreads();
    };
    auto satd = [&]() { return w > h ? satd_block<16, 8>(org, s_org, pred, w, h, red) : w < h ? satd_block<8, 16>(org, s_org, pred, w, h, red) : satd_block<8, 8>(org, s_org, pred, w, h, red); };
    compensate();
    int      best_bits = xaff::me_mv_bits(mvt, jb.mvp, num_refp, ri, VN) + (bi ? jb.mot_bits_other : 0);
    uint32_t cost_best = xaff::me_mv_cost(A.lambda_mv, best_bits) + (uint32_t)((satd() >> (A.bit_depth - 8)) >> bi);
    const int rounds = (bi ? 5 : 7) - (VN == 3 ? 2 : 0); // AF_ITER_BI / AF_ITER_UNI (xeve_def.h:168-169), two fewer with three control points (:4380-4382)
    for(int it = 0; it < rounds; it++) {
        // error, both Sobel derivatives and the sample's products in one pass (xeve_diff_16b, xevem_scaled_*_sobel_filter, xevem_equal_coeff_computer: :4385-4404)
        long long acc[NS];
#pragma unroll
        for(int q = 0; q < NS; q++) acc[q] = 0;
        for(int i = threadIdx.x; i < n; i += blockDim.x) {
            const int j = i / w, k = i - j * w;
            int32_t   c[6];
            xaff::me_terms(LdsAt{pred, w}, w, h, j, k, VN, c);
            const long long e8 = (long long)(int16_t)((int)org[(long)j * s_org + k] - (int)pred[i]) * 8;
            int q = 0;
#pragma unroll
            for(int col = 0; col < NP; col++) {
#pragma unroll
                for(int row = col; row < NP; row++) acc[q++] += (long long)c[col] * c[row];
                acc[q++] += (long long)c[col] * e8;
            }
        }
#pragma unroll
        for(int q = 0; q < NS; q++) acc[q] = wave_sum64(acc[q]);
        if((threadIdx.x & 63) == 0)
#pragma unroll
            for(int q = 0; q < NS; q++) red64[(threadIdx.x >> 6) * NS + q] = acc[q];
        __syncthreads();
        if(threadIdx.x < (NP + 1) * (NP + 1)) { // equal_coeff_t -> equal_coeff (:4405-4409): a lane per element; the sums are symmetric, stored as the upper triangle + right-hand side
            const int r = threadIdx.x / (NP + 1), c = threadIdx.x - r * (NP + 1);
            long long v = 0;
            if(r > 0) {
                const int col = r - 1, lo = c < NP ? (c < col ? c : col) : col, hi = c < NP ? (c < col ? col : c) : NP; // (column NP: the right-hand side, behind its row's triangle)
                const int q = lo * (NP + 1) - lo * (lo - 1) / 2 + (hi - lo);
                v = red64[q] + red64[NS + q] + red64[2 * NS + q] + red64[3 * NS + q];
            }
            s_eq[r][c] = (double)v;
        }
        __syncthreads();
        solve_by_lanes<NP>(s_eq, s_para);
        if(threadIdx.x == 0) {
            int16_t mvd[3][2];
            s_stop = xaff::me_step(s_para, VN, w, h, mvd);
            for(int v = 0; v < 3; v++) s_mvt[v][0] = (int16_t)(mvt[v][0] + mvd[v][0]), s_mvt[v][1] = (int16_t)(mvt[v][1] + mvd[v][1]);
        }
        __syncthreads();
        if(s_stop) break; // (:4449-4458)
        for(int v = 0; v < VN; v++) mvt[v][0] = s_mvt[v][0], mvt[v][1] = s_mvt[v][1];
        compensate();
        const int      bits = xaff::me_mv_bits(mvt, jb.mvp, num_refp, ri, VN) + (bi ? jb.mot_bits_other : 0);
        const uint32_t cost = xaff::me_mv_cost(A.lambda_mv, bits) + (uint32_t)((satd() >> (A.bit_depth - 8)) >> bi);
        if(cost < cost_best) {
            cost_best = cost, best_bits = bits;
            for(int v = 0; v < VN; v++) best[v][0] = mvt[v][0], best[v][1] = mvt[v][1];
        }
    }
    if(threadIdx.x == 0) {
        for(int v = 0; v < VN; v++) jp->mv[v][0] = best[v][0], jp->mv[v][1] = best[v][1];
        jp->cost = cost_best - xaff::me_mv_cost(A.lambda_mv, best_bits);
    }
}

} // namespace

extern "C" int xeve_hip_affine_mc_jobs(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h, const xeve_hip_affine_job *jobs,
                                       int njobs, int w, int h, int bit_depth, xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(refp && num_refp0 >= 0 && num_refp1 >= 0 && num_refp0 <= MAXREF && num_refp1 <= MAXREF && num_refp0 + num_refp1 > 0 && s_l > 0 && s_c > 0 && pic_w > 0 && pic_h > 0);
    XH_REQUIRE(njobs >= 0 && (njobs == 0 || jobs) && xh_pow2(w) && xh_pow2(h) && w >= 8 && h >= 8 && w <= 128 && h <= 128 && bit_depth >= 8 && bit_depth <= 12);
    XH_REQUIRE(pred_y && pred_u && pred_v);
    if(njobs == 0) return XEVE_HIP_OK;
    RefTab tab;
    memset(&tab, 0, sizeof(tab));
    const int nr = num_refp0 > num_refp1 ? num_refp0 : num_refp1;
    for(int r = 0; r < nr; r++)
        for(int l = 0; l < 2; l++)
            if(r < (l ? num_refp1 : num_refp0)) {
                tab.r[r * 2 + l] = refp[r * 2 + l];
                XH_REQUIRE(tab.r[r * 2 + l].y && tab.r[r * 2 + l].u && tab.r[r * 2 + l].v);
            }
    k_affine_mc<<<njobs, 256, 0, (hipStream_t)stream>>>(tab, s_l, s_c, pic_w, pic_h, jobs, w, h, bit_depth, pred_y, pred_u, pred_v);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- host-memory form of ONE xeve_affine_mc call (src_main/xevem_mc.c:2236-2339): what the Main-profile encoder's by-name calls can be routed to (the affine merge
// analysis, every round of the affine gradient search, the affine bi-prediction: xevem_pinter.c:1947, 4659, 4918 -- oracle/ref_shim_affine.c does).
// refp: table [refi * 2 + list] of HOST plane pointers (sample (0, 0)) of the pictures the CU uses; the planes extend pad_l / pad_c samples around the picture; pred_y /
// pred_u / pred_v: HOST blocks of w * h and (w / 2) * (h / 2) samples, what the reference leaves in pred[0][Y_C / U_C / V_C].  Per calling thread one stream and one device
// arena (grow-only): a call is the upload of the (at most two) pictures it uses, one launch, the download of three blocks.
extern "C" int xeve_hip_affine_mc_host(int x, int y, int pic_w, int pic_h, int w, int h, const int8_t refi[2], const int16_t mv[2][3][2], const xeve_hip_refpic *refp,
                                       int num_refp0, int num_refp1, int s_l, int s_c, int pad_l, int pad_c, xeve_hip_pel *pred_y, xeve_hip_pel *pred_u, xeve_hip_pel *pred_v,
                                       int vertex_num, int bit_depth)
{
    XH_ENTER();
    XH_REQUIRE(refi && mv && refp && pred_y && pred_u && pred_v && (vertex_num == 2 || vertex_num == 3) && pad_l >= 0 && pad_c >= 0 && s_l > 0 && s_c > 0 && pic_w > 0 && pic_h > 0);
    XH_REQUIRE(num_refp0 >= 0 && num_refp1 >= 0 && num_refp0 <= MAXREF && num_refp1 <= MAXREF && (refi[0] < 0 || refi[0] < num_refp0) && (refi[1] < 0 || refi[1] < num_refp1) &&
               (refi[0] >= 0 || refi[1] >= 0));
    const size_t el = (size_t)s_l * (pic_h + 2 * pad_l), ec = (size_t)s_c * ((pic_h >> 1) + 2 * pad_c), ol = (size_t)pad_l * s_l + pad_l, oc = (size_t)pad_c * s_c + pad_c;
    const size_t n0 = (size_t)w * h, n1 = n0 >> 2, plane = (el + 2 * ec) * sizeof(pel), o_planes = 256, o_pred = o_planes + 2 * ((plane + 255) & ~(size_t)255);
    static thread_local XhHostArena A;
    const int rc0 = A.ensure(o_pred + (n0 + 2 * n1) * sizeof(pel), 0);
    if(rc0 != XEVE_HIP_OK) return rc0;
    xeve_hip_affine_job J;
    memset(&J, 0, sizeof(J));
    J.x = x, J.y = y, J.vertex_num = (int8_t)vertex_num;
    xeve_hip_refpic tab[2 * MAXREF];
    memset(tab, 0, sizeof(tab));
    pel *first = nullptr;
    for(int l = 0; l < 2; l++) {
        J.refi[l] = refi[l];
        for(int v = 0; v < 3; v++) J.mv[l][v][0] = mv[l][v][0], J.mv[l][v][1] = mv[l][v][1];
        if(refi[l] < 0) continue;
        const xeve_hip_refpic &src = refp[refi[l] * 2 + l];
        XH_REQUIRE(src.y && src.u && src.v);
        pel *d = (pel *)(A.dev + o_planes + (size_t)l * ((plane + 255) & ~(size_t)255));
        XH_HIP(hipMemcpyAsync(d, src.y - ol, el * sizeof(pel), hipMemcpyHostToDevice, A.st));
        XH_HIP(hipMemcpyAsync(d + el, src.u - oc, ec * sizeof(pel), hipMemcpyHostToDevice, A.st));
        XH_HIP(hipMemcpyAsync(d + el + ec, src.v - oc, ec * sizeof(pel), hipMemcpyHostToDevice, A.st));
        if(!first) first = d;
        // every picture index a list is asked to hold must be addressable: all of them alias the staged one (the job names refi[l] alone)
        for(int r = 0; r < (l ? num_refp1 : num_refp0); r++) tab[r * 2 + l].y = d + ol, tab[r * 2 + l].u = d + el + oc, tab[r * 2 + l].v = d + el + ec + oc, tab[r * 2 + l].poc = src.poc;
    }
    for(int l = 0; l < 2; l++) // (a list the CU does not use: valid pointers for the table's check, never read)
        if(refi[l] < 0)
            for(int r = 0; r < (l ? num_refp1 : num_refp0); r++) tab[r * 2 + l].y = first + ol, tab[r * 2 + l].u = first + el + oc, tab[r * 2 + l].v = first + el + ec + oc;
    XH_HIP(hipMemcpyAsync(A.dev, &J, sizeof(J), hipMemcpyHostToDevice, A.st));
    pel *dp = (pel *)(A.dev + o_pred);
    const int rc = xeve_hip_affine_mc_jobs(tab, num_refp0, num_refp1, s_l, s_c, pic_w, pic_h, (const xeve_hip_affine_job *)A.dev, 1, w, h, bit_depth, dp, dp + n0, dp + n0 + n1, A.st);
    if(rc != XEVE_HIP_OK) return rc;
    XH_HIP(hipMemcpyAsync(pred_y, dp, n0 * sizeof(pel), hipMemcpyDeviceToHost, A.st));
    XH_HIP(hipMemcpyAsync(pred_u, dp + n0, n1 * sizeof(pel), hipMemcpyDeviceToHost, A.st));
    XH_HIP(hipMemcpyAsync(pred_v, dp + n0 + n1, n1 * sizeof(pel), hipMemcpyDeviceToHost, A.st));
    XH_HIP(hipStreamSynchronize(A.st));
    return XEVE_HIP_OK;
}

// ---- the affine gradient search: every search of one CU size in ONE launch (a workgroup per search) ------------------------------------------------------------------------
static int affine_me_launch(const AffMeArgs &A, int njobs, hipStream_t st)
{
    const int    n = A.w * A.h, nbuf = (A.w + 7) * (A.h + 7);
    const size_t lds = (((size_t)(n + nbuf) * sizeof(pel) + 15) & ~(size_t)15) + 4 * 27 * sizeof(long long);
    static bool  raised = false; // CUs of 128 need more than the 64 KB a kernel gets by default (gfx950: 160 KB per workgroup)
    if(lds > 60000 && !raised) {
        XH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_affine_me<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        XH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_affine_me<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        raised = true;
    }
    // a job belongs to the launch of its model (two / three control points); the other launch's workgroup returns at once
    k_affine_me<2><<<njobs, 256, lds, st>>>(A);
    XH_HIP(hipGetLastError());
    k_affine_me<3><<<njobs, 256, lds, st>>>(A);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_affine_me_jobs(const xeve_hip_refpic *refp, int ntab0, int ntab1, int s_l, int pic_w, int pic_h, const xeve_hip_pel *org, int s_org, const int16_t *org_bi,
                                       xeve_hip_affine_me_job *jobs, int njobs, int w, int h, int bit_depth, uint32_t lambda_mv, int num_refp0, int num_refp1, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(refp && ntab0 >= 0 && ntab1 >= 0 && ntab0 <= MAXREF && ntab1 <= MAXREF && ntab0 + ntab1 > 0 && s_l > 0 && pic_w > 0 && pic_h > 0 && org && s_org > 0);
    XH_REQUIRE(njobs >= 0 && (njobs == 0 || jobs) && xh_pow2(w) && xh_pow2(h) && w >= 16 && h >= 16 && w <= 128 && h <= 128 && bit_depth >= 8 && bit_depth <= 12);
    XH_REQUIRE(num_refp0 >= 0 && num_refp1 >= 0 && num_refp0 <= 16 && num_refp1 <= 16);
    if(njobs == 0) return XEVE_HIP_OK;
    AffMeArgs A;
    memset(&A, 0, sizeof(A));
    const int nr = ntab0 > ntab1 ? ntab0 : ntab1;
    for(int r = 0; r < nr; r++)
        for(int l = 0; l < 2; l++)
            if(r < (l ? ntab1 : ntab0)) {
                A.tab.r[r * 2 + l] = refp[r * 2 + l];
                XH_REQUIRE(A.tab.r[r * 2 + l].y);
            }
    A.s_l = s_l, A.s_org = s_org, A.org = org, A.org_blocks = org_bi, A.jobs = jobs, A.pic_w = pic_w, A.pic_h = pic_h, A.w = w, A.h = h, A.bit_depth = bit_depth, A.org_dense = 0;
    A.num_refp[0] = num_refp0, A.num_refp[1] = num_refp1, A.lambda_mv = lambda_mv;
    return affine_me_launch(A, njobs, (hipStream_t)stream);
}

// ---- host-memory form of ONE pinter_affine_me_gradient call (src_main/xevem_pinter.c:4290-4501): what the Main-profile encoder's pi->fn_affine_me can be bound to
// (oracle/ref_shim_affine.c does).  ref_y: HOST luma plane (sample (0, 0)) of refp[refi][list].pic, extending pad_l samples around the picture; org: the CU's
// first original sample with its pitch (bi: pi->org_bi, pitch w); mv: in the start vectors, out the best ones (vertex_num of them); *cost: the function's value.  Per calling
// thread one stream and one device arena (grow-only): a call is the upload of the picture and of the CU's original, two launches, the download of the job record.
extern "C" int xeve_hip_affine_me_host(int x, int y, int pic_w, int pic_h, int w, int h, int refi, int list, const int16_t mvp[3][2], int16_t mv[3][2], int bi, int vertex_num,
                                       const xeve_hip_pel *ref_y, int s_l, int pad_l, const int16_t *org, int s_org, int bit_depth, uint32_t lambda_mv, int num_refp,
                                       int mot_bits_other, uint32_t *cost)
{
    XH_ENTER();
    XH_REQUIRE(mvp && mv && ref_y && org && cost && (vertex_num == 2 || vertex_num == 3) && pad_l >= 0 && s_l > 0 && s_org > 0 && pic_w > 0 && pic_h > 0);
    XH_REQUIRE(refi >= 0 && refi < MAXREF && (list == 0 || list == 1) && num_refp > refi && num_refp <= 16 && xh_pow2(w) && xh_pow2(h) && w >= 16 && h >= 16 && w <= 128 && h <= 128);
    const size_t el = (size_t)s_l * (pic_h + 2 * pad_l), ol = (size_t)pad_l * s_l + pad_l, n = (size_t)w * h;
    const size_t o_plane = 256, o_org = o_plane + ((el * sizeof(pel) + 255) & ~(size_t)255), total = o_org + n * sizeof(int16_t);
    static thread_local XhHostArena A;
    const int rc0 = A.ensure(total, 0);
    if(rc0 != XEVE_HIP_OK) return rc0;
    xeve_hip_affine_me_job *J = reinterpret_cast<xeve_hip_affine_me_job *>(A.pin);
    memset(J, 0, sizeof(*J));
    J->x = x, J->y = y, J->refi = (int8_t)refi, J->list = (int8_t)list, J->bi = (int8_t)(bi != 0), J->vertex_num = (int8_t)vertex_num, J->mot_bits_other = mot_bits_other;
    for(int v = 0; v < 3; v++)
        for(int c = 0; c < 2; c++) J->mvp[v][c] = v < vertex_num ? mvp[v][c] : (int16_t)0, J->mv[v][c] = v < vertex_num ? mv[v][c] : (int16_t)0;
    int16_t *ho = reinterpret_cast<int16_t *>(A.pin + o_org);
    for(int r = 0; r < h; r++) memcpy(ho + (size_t)r * w, org + (size_t)r * s_org, (size_t)w * sizeof(int16_t));
    XH_HIP(hipMemcpyAsync(A.dev, J, sizeof(*J), hipMemcpyHostToDevice, A.st));
    XH_HIP(hipMemcpyAsync(A.dev + o_org, ho, n * sizeof(int16_t), hipMemcpyHostToDevice, A.st));
    XH_HIP(hipMemcpyAsync(A.dev + o_plane, ref_y - ol, el * sizeof(pel), hipMemcpyHostToDevice, A.st));
    AffMeArgs K;
    memset(&K, 0, sizeof(K));
    K.tab.r[refi * 2 + list].y = reinterpret_cast<const pel *>(A.dev + o_plane) + ol;
    K.s_l = s_l, K.s_org = w, K.org = nullptr, K.org_blocks = reinterpret_cast<const int16_t *>(A.dev + o_org), K.jobs = reinterpret_cast<xeve_hip_affine_me_job *>(A.dev);
    K.pic_w = pic_w, K.pic_h = pic_h, K.w = w, K.h = h, K.bit_depth = bit_depth, K.org_dense = 1, K.num_refp[0] = K.num_refp[1] = num_refp, K.lambda_mv = lambda_mv;
    const int rc = affine_me_launch(K, 1, A.st);
    if(rc != XEVE_HIP_OK) return rc;
    XH_HIP(hipMemcpyAsync(J, A.dev, sizeof(*J), hipMemcpyDeviceToHost, A.st));
    XH_HIP(hipStreamSynchronize(A.st));
    for(int v = 0; v < vertex_num; v++) mv[v][0] = J->mv[v][0], mv[v][1] = J->mv[v][1];
    *cost = J->cost;
    return XEVE_HIP_OK;
}
