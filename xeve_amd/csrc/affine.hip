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
            pel *dst = (c == 0 ? pred_y : c == 1 ? pred_u : pred_v) + (size_t)blockIdx.x * cw * ch;
            __syncthreads(); // (the previous component's readers are done with buf)
            if(eif) {
                int mx[2], mn[2];
                xaff::eif_range(m, mem_ok, jb.x, jb.y, w, h, pic_w, pic_h, mx, mn);
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
                xaff::block_vector(m, sub_w, sub_h, jb.x, jb.y, w, h, pic_w, pic_h, th, tv, oh, ov);
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
        nth++;
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
