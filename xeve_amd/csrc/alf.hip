// xeve_amd/csrc/alf.hip -- Main profile: the adaptive loop filter's sample kernels on planes resident in HBM.  reference: src_main/xevem_alf.c --
// alf_copy_and_extend (:91-168), alf_derive_classification / _blk (:463-654), alf_filter_blk_7 / _5 (:656-882; the ADAPTIVE_LOOP_FILTER object's filter_7x7_blk /
// filter_5x5_blk pointers, :52-53), xeve_alf_get_blk_stats + xeve_alf_clac_covariance (:3836-3952).  What a lane computes is alf_core.h (pinned on the host:
// tests/native/alf_host.cpp); this file is tiling and data movement.
//   All three sample kernels work on 64x64 tiles: a workgroup stages the tile and the 3 samples around it in LDS (70 x 70 x 2 B = 9.8 KB) once -- every sample is read
//   13 .. 100 times by the lanes around it -- and gives each lane ONE 4x4 block (256 lanes = the 256 blocks of a tile): the classifier byte, the transposed
//   coefficients, the class's statistics row are per block.  HBM traffic = the tile once + the outputs: all three are bandwidth-bound by construction.
//   The statistics are integer sums (products of 12-bit sums, 64-bit accumulators) turned into doubles at the end: the reference adds the same integers into doubles
//   sample by sample, exactly (a picture's sums stay below 2^53), so the order of accumulation cannot show.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "alf_core.h"
#include "xh_common.h"

namespace {

constexpr int TS = 64, MG = 3, TP = TS + 2 * MG; // tile, margin, tile pitch in LDS

struct TileAt { // alf_core.h's `at`: an element's neighbourhood in the staged tile
    const pel *p;
    __device__ int operator()(int dy, int dx) const { return p[dy * TP + dx]; }
};

// the tile whose first sample is src[0] (w x h of it inside the area) with its margin -> LDS.  Every sample the area's kernels may read exists in the caller's plane (the
// planes carry the ALF margin: alf_copy_and_extend's output, or a CTU-sized window cut out of one)
__device__ void stage_tile(pel *tile, const pel *src, long s_src, int w, int h)
{
    for(int i = threadIdx.x; i < (h + 2 * MG) * (w + 2 * MG); i += blockDim.x) {
        const int r = i / (w + 2 * MG), c = i - r * (w + 2 * MG);
        tile[r * TP + c] = src[(long)(r - MG) * s_src + c - MG];
    }
    __syncthreads();
}

// ---- alf_copy_and_extend: dst(y, x) = rec(clamp y, clamp x) over the area and m samples around it ----------------------------------------------------------------------
__global__ void k_alf_extend(pel *__restrict__ tmp, long s_tmp, const pel *__restrict__ rec, long s_rec, int w, int h, int m)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, W = w + 2 * m;
    if(i >= W * (h + 2 * m)) return;
    const int y = (int)(i / W) - m, x = (int)(i % W) - m, cy = y < 0 ? 0 : y >= h ? h - 1 : y, cx = x < 0 ? 0 : x >= w ? w - 1 : x;
    tmp[(long)y * s_tmp + x] = rec[(long)cy * s_rec + cx];
}

// ---- alf_derive_classification: one 64x64 tile per workgroup, one 4x4 block per lane --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_alf_classify(uint8_t *__restrict__ cls, long s_cls, const pel *__restrict__ src, long s_src, int ax, int ay, int aw, int ah, int bit_depth)
{
    __shared__ pel tile[TP * TP];
    const int tx = blockIdx.x * TS, ty = blockIdx.y * TS, w = min(TS, aw - tx), h = min(TS, ah - ty);
    stage_tile(tile, src + (long)(ay + ty) * s_src + ax + tx, s_src, w, h);
    const int bx = (threadIdx.x & 15) * 4, by = (threadIdx.x >> 4) * 4;
    if(bx >= w || by >= h) return;
    const uint8_t v = xalf::block_class(TileAt{tile + (by + MG) * TP + bx + MG}, bit_depth);
    const uint32_t v4 = v * 0x01010101u;
    for(int a = 0; a < 4; a++) *reinterpret_cast<uint32_t *>(cls + (long)(ay + ty + by + a) * s_cls + ax + tx + bx) = v4; // (areas start on multiples of 4: aligned)
}

// ---- alf_filter_blk_7 / _5 over a list of areas: one job per blockIdx.x, its 64x64 tiles in sequence, one 4x4 block per lane ------------------------------------------
struct FilterSet {
    int16_t c[25 * 13];
};
template <int TAPS>
__global__ void __launch_bounds__(256) k_alf_filter(pel *__restrict__ dst, long s_dst, const pel *__restrict__ src, long s_src, const uint8_t *__restrict__ cls, long s_cls,
                                                    const xeve_hip_alf_filter_job *__restrict__ jobs, FilterSet fs, int clip_min, int clip_max)
{
    __shared__ pel tile[TP * TP];
    const xeve_hip_alf_filter_job jb = jobs[blockIdx.x];
    const int bx = (threadIdx.x & 15) * 4, by = (threadIdx.x >> 4) * 4;
    for(int ty = 0; ty < jb.h; ty += TS)
        for(int tx = 0; tx < jb.w; tx += TS) {
            const int w = min(TS, jb.w - tx), h = min(TS, jb.h - ty);
            __syncthreads(); // (the previous tile's readers are done)
            stage_tile(tile, src + jb.src_off + (long)ty * s_src + tx, s_src, w, h);
            if(bx >= w || by >= h) continue;
            int16_t c[TAPS == 7 ? 13 : 7];
            if(TAPS == 7) {
                const uint8_t cl = cls[(long)(jb.y + ty + by) * s_cls + jb.x + tx + bx];
                const int16_t *set = fs.c + min((cl >> 2) & 0x1F, 24) * 13; // (25 classes: a stale classifier byte cannot index past the filter set)
#pragma unroll
                for(int k = 0; k < 13; k++) c[k] = set[xalf::order7(cl & 3, k)];
            }
            else {
#pragma unroll
                for(int k = 0; k < 7; k++) c[k] = fs.c[k];
            }
            pel *d = dst + jb.dst_off + (long)(ty + by) * s_dst + tx + bx;
            for(int a = 0; a < 4; a++) {
                pel o[4];
#pragma unroll
                for(int b = 0; b < 4; b++) o[b] = (pel)xalf::filter_sample<TAPS>(TileAt{tile + (by + a + MG) * TP + bx + b + MG}, c, clip_min, clip_max);
#pragma unroll
                for(int b = 0; b < 4; b++) d[(long)a * s_dst + b] = o[b];
            }
        }
}

// ---- xeve_alf_get_blk_stats over a list of areas: one job per blockIdx.x.  A row of 4x4 blocks at a time (16 blocks = 256 samples): every lane forms ONE sample's local
// sums and error in LDS, then lane t < nent, which owns entry t of the record (an E[k][l], a y[k] or the energy), adds each block's 16 products to the row of the
// block's class -- no atomics, no conflicts ------------------------------------------------------------------------------------------------------------------------------
template <int TAPS>
__global__ void __launch_bounds__(256) k_alf_stats(const uint8_t *__restrict__ cls, long s_cls, const pel *__restrict__ org, long s_org, const pel *__restrict__ rec, long s_rec,
                                                   const xeve_hip_alf_area *__restrict__ jobs, int nclasses, double *__restrict__ E, double *__restrict__ yv,
                                                   double *__restrict__ pix)
{
    constexpr int NC = TAPS * TAPS / 4 + 1, NENT = NC * (NC + 1) / 2 + NC + 1;
    __shared__ pel       tile[TP * TP];
    __shared__ int       loc[256][NC + 1]; // per sample of the row of blocks: the NC local sums, then org - rec
    __shared__ uint8_t   bcls[16];         // the blocks' classifier bytes
    __shared__ long long acc[25][NENT];
    const xeve_hip_alf_area jb = jobs[blockIdx.x];
    for(int i = threadIdx.x; i < 25 * NENT; i += blockDim.x) (&acc[0][0])[i] = 0;
    int k = -1, l = -1;
    if(threadIdx.x < NENT) xalf::stat_entry(NC, threadIdx.x, k, l);
    const int ia = k < 0 ? NC : k, ib = k < 0 ? NC : l < 0 ? NC : l; // the entry's two factors among loc[][0 .. NC]
    const int sb = threadIdx.x >> 4, sa = (threadIdx.x >> 2) & 3, sc = threadIdx.x & 3; // this lane's sample: block of the row, row and column inside it
    for(int ty = 0; ty < jb.h; ty += TS)
        for(int tx = 0; tx < jb.w; tx += TS) {
            const int w = min(TS, jb.w - tx), h = min(TS, jb.h - ty);
            __syncthreads();
            stage_tile(tile, rec + (long)(jb.y + ty) * s_rec + jb.x + tx, s_rec, w, h);
            for(int by = 0; by < h; by += 4) {
                if(threadIdx.x < 16) bcls[threadIdx.x] = cls && threadIdx.x * 4 < w ? cls[(long)(jb.y + ty + by) * s_cls + jb.x + tx + threadIdx.x * 4] : 0;
                __syncthreads();
                if(sb * 4 < w) {
                    const pel *t = tile + (by + sa + MG) * TP + sb * 4 + sc + MG;
                    int e[NC];
                    xalf::local_sums<TAPS>(TileAt{t}, bcls[sb] & 3, e);
#pragma unroll
                    for(int q = 0; q < NC; q++) loc[threadIdx.x][q] = e[q];
                    loc[threadIdx.x][NC] = org[(long)(jb.y + ty + by + sa) * s_org + jb.x + tx + sb * 4 + sc] - t[0];
                }
                __syncthreads();
                if(threadIdx.x < NENT)
                    for(int b = 0; b * 4 < w; b++) {
                        long long part = 0;
#pragma unroll
                        for(int q = 0; q < 16; q++) part += loc[b * 16 + q][ia] * loc[b * 16 + q][ib];
                        acc[min((bcls[b] >> 2) & 0x1F, 24)][threadIdx.x] += part;
                    }
                __syncthreads();
            }
        }
    // the record: E full and symmetric in [13][13] (zero beyond NC), y, the energy
    for(int i = threadIdx.x; i < nclasses * (13 * 13 + 13 + 1); i += blockDim.x) {
        const int c = i / (13 * 13 + 13 + 1), r = i - c * (13 * 13 + 13 + 1);
        const long jc = (long)blockIdx.x * nclasses + c;
        if(r < 13 * 13) {
            int a = r / 13, b = r - a * 13;
            if(a > b) { const int t = a; a = b, b = t; }
            // entry index of (a, b), a <= b: rows of the upper triangle one after the other
            E[jc * 169 + r] = b < NC ? (double)acc[c][a * NC - a * (a - 1) / 2 + (b - a)] : 0.0;
        }
        else if(r < 13 * 13 + 13) yv[jc * 13 + r - 169] = r - 169 < NC ? (double)acc[c][NC * (NC + 1) / 2 + r - 169] : 0.0;
        else pix[jc] = (double)acc[c][NENT - 1];
    }
}

} // namespace

extern "C" int xeve_hip_alf_copy_and_extend(xeve_hip_pel *tmp, int s_tmp, const xeve_hip_pel *rec, int s_rec, int w, int h, int m, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(tmp && rec && w > 0 && h > 0 && m >= 0 && m <= 16 && s_tmp >= w + 2 * m && s_rec >= w);
    const long n = (long)(w + 2 * m) * (h + 2 * m);
    k_alf_extend<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(tmp, s_tmp, rec, s_rec, w, h, m);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_alf_classify(uint8_t *classifier, int s_cls, const xeve_hip_pel *src_luma, int s_src, const xeve_hip_alf_area *area, int bit_depth, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(classifier && src_luma && area && area->w > 0 && area->h > 0 && ((area->x | area->y | area->w | area->h) & 3) == 0 && area->x >= 0 && area->y >= 0);
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14 && s_cls >= area->x + area->w && (s_cls & 3) == 0 && s_src > 0);
    XH_REQUIRE((reinterpret_cast<uintptr_t>(classifier) & 3) == 0); // (a block's four bytes of a row are written as one word)
    const dim3 grid((area->w + TS - 1) / TS, (area->h + TS - 1) / TS);
    k_alf_classify<<<grid, 256, 0, (hipStream_t)stream>>>(classifier, s_cls, src_luma, s_src, area->x, area->y, area->w, area->h, bit_depth);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_alf_filter_jobs(int taps, xeve_hip_pel *dst, int s_dst, const xeve_hip_pel *src, int s_src, const uint8_t *classifier, int s_cls,
                                        const xeve_hip_alf_filter_job *jobs, int njobs, const int16_t *filter_set, int clip_min, int clip_max, void *stream)
{
    XH_ENTER();
    XH_REQUIRE((taps == 5 || taps == 7) && dst && src && dst != src && filter_set && njobs >= 0 && (njobs == 0 || jobs) && s_dst > 0 && s_src > 0 && clip_min <= clip_max);
    XH_REQUIRE(taps == 5 || (classifier && s_cls > 0));
    if(njobs == 0) return XEVE_HIP_OK;
    FilterSet fs;
    memset(&fs, 0, sizeof(fs));
    memcpy(fs.c, filter_set, sizeof(int16_t) * (taps == 7 ? 25 * 13 : 7));
    if(taps == 7) k_alf_filter<7><<<njobs, 256, 0, (hipStream_t)stream>>>(dst, s_dst, src, s_src, classifier, s_cls, jobs, fs, clip_min, clip_max);
    else k_alf_filter<5><<<njobs, 256, 0, (hipStream_t)stream>>>(dst, s_dst, src, s_src, classifier, s_cls, jobs, fs, clip_min, clip_max);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_alf_blk_stats_jobs(int taps, const uint8_t *classifier, int s_cls, const xeve_hip_pel *org, int s_org, const xeve_hip_pel *rec, int s_rec,
                                           const xeve_hip_alf_area *jobs, int njobs, double *E, double *y, double *pix_acc, void *stream)
{
    XH_ENTER();
    XH_REQUIRE((taps == 5 || taps == 7) && org && rec && s_org > 0 && s_rec > 0 && njobs >= 0 && (njobs == 0 || jobs) && E && y && pix_acc && (!classifier || s_cls > 0));
    if(njobs == 0) return XEVE_HIP_OK;
    const int nclasses = classifier ? 25 : 1;
    if(taps == 7) k_alf_stats<7><<<njobs, 256, 0, (hipStream_t)stream>>>(classifier, s_cls, org, s_org, rec, s_rec, jobs, nclasses, E, y, pix_acc);
    else k_alf_stats<5><<<njobs, 256, 0, (hipStream_t)stream>>>(classifier, s_cls, org, s_org, rec, s_rec, jobs, nclasses, E, y, pix_acc);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- host-memory forms with the reference's signatures (the ADAPTIVE_LOOP_FILTER object's function pointers) -----------------------------------------------------------
namespace {
[[noreturn]] void alf_die(const char *what)
{ // (these signatures have no status channel, and the callers check nothing: a failure must not pass for a result)
    fprintf(stderr, "libxeve_hip: %s failed: %s\n", what, xeve_hip_last_error());
    abort();
}
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes, const char *what)
    {
        if(hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) xh_set_error("device allocation of %zu bytes", bytes), alf_die(what);
    }
    ~DevBuf() { (void)hipFree(p); }
};
// rows [y0, y0 + rows) x columns [x0, x0 + cols) of a host plane (pointer at its sample (0, 0), either sign of offsets) <-> a dense device buffer
void rows_to_device(void *dev, const void *host00, long stride, int elem, int y0, int x0, int rows, int cols, const char *what)
{
    if(hipMemcpy2D(dev, (size_t)cols * elem, (const char *)host00 + ((long)y0 * stride + x0) * elem, (size_t)stride * elem, (size_t)cols * elem, rows, hipMemcpyHostToDevice) != hipSuccess)
        xh_set_error("copy to the device"), alf_die(what);
}
void filter_host(int taps, uint8_t **classifier, pel *rec_dst, int dst_stride, const pel *rec_src, int src_stride, const xeve_hip_alf_area *blk, short *filter_set,
                 const xeve_hip_alf_clip_range *cr, const char *what)
{
    if(!xh_ready()) xh_set_error("xeve_hip_init() has not been called"), alf_die(what);
    if(!(rec_dst && rec_src && blk && filter_set && cr && blk->w > 0 && blk->h > 0 && ((blk->w | blk->h) & 3) == 0 && (taps == 5 || classifier)))
        xh_set_error("invalid argument"), alf_die(what);
    const int w = blk->w, h = blk->h, sw = w + 2 * MG;
    DevBuf src((size_t)sw * (h + 2 * MG) * sizeof(pel), what), dst((size_t)w * h * sizeof(pel), what), cls((size_t)w * h, what), job(sizeof(xeve_hip_alf_filter_job), what);
    rows_to_device(src.p, rec_src, src_stride, sizeof(pel), -MG, -MG, h + 2 * MG, sw, what);
    if(taps == 7) { // the classifier is a table of row pointers, indexed with the area's own coordinates
        std::vector<uint8_t> c((size_t)w * h);
        for(int i = 0; i < h; i++) memcpy(c.data() + (size_t)i * w, classifier[blk->y + i] + blk->x, (size_t)w);
        if(hipMemcpy(cls.p, c.data(), c.size(), hipMemcpyHostToDevice) != hipSuccess) xh_set_error("copy to the device"), alf_die(what);
    }
    const xeve_hip_alf_filter_job jb = {0, 0, w, h, 0, (int64_t)MG * sw + MG};
    if(hipMemcpy(job.p, &jb, sizeof(jb), hipMemcpyHostToDevice) != hipSuccess) xh_set_error("copy to the device"), alf_die(what);
    if(xeve_hip_alf_filter_jobs(taps, (pel *)dst.p, w, (const pel *)src.p, sw, (const uint8_t *)cls.p, w, (const xeve_hip_alf_filter_job *)job.p, 1, filter_set, cr->min, cr->max,
                                nullptr) != XEVE_HIP_OK)
        alf_die(what);
    if(hipMemcpy2D(rec_dst, (size_t)dst_stride * sizeof(pel), dst.p, (size_t)w * sizeof(pel), (size_t)w * sizeof(pel), h, hipMemcpyDeviceToHost) != hipSuccess)
        xh_set_error("copy from the device"), alf_die(what);
}
} // namespace

extern "C" void xeve_hip_alf_derive_classification_blk_host(uint8_t **classifier, const xeve_hip_pel *src_luma, int src_stride, const xeve_hip_alf_area *blk, int shift,
                                                            int bit_depth)
{
    const char *what = "xeve_hip_alf_derive_classification_blk_host";
    (void)shift; // (unused by the reference as well, :488-493)
    if(!xh_ready()) xh_set_error("xeve_hip_init() has not been called"), alf_die(what);
    if(!(classifier && src_luma && blk && blk->w > 0 && blk->h > 0 && ((blk->w | blk->h) & 3) == 0)) xh_set_error("invalid argument"), alf_die(what);
    const int w = blk->w, h = blk->h, sw = w + 2 * MG;
    DevBuf src((size_t)sw * (h + 2 * MG) * sizeof(pel), what), cls((size_t)w * h, what);
    rows_to_device(src.p, src_luma, src_stride, sizeof(pel), blk->y - MG, blk->x - MG, h + 2 * MG, sw, what);
    const xeve_hip_alf_area a = {0, 0, w, h};
    if(xeve_hip_alf_classify((uint8_t *)cls.p, w, (const pel *)src.p + (size_t)MG * sw + MG, sw, &a, bit_depth, nullptr) != XEVE_HIP_OK) alf_die(what);
    std::vector<uint8_t> c((size_t)w * h);
    if(hipMemcpy(c.data(), cls.p, c.size(), hipMemcpyDeviceToHost) != hipSuccess) xh_set_error("copy from the device"), alf_die(what);
    for(int i = 0; i < h; i++) memcpy(classifier[blk->y + i] + blk->x, c.data() + (size_t)i * w, (size_t)w);
}
extern "C" void xeve_hip_alf_filter_blk_7_host(uint8_t **classifier, xeve_hip_pel *rec_dst, int dst_stride, const xeve_hip_pel *rec_src, int src_stride,
                                               const xeve_hip_alf_area *blk, uint8_t comp_id, short *filter_set, const xeve_hip_alf_clip_range *clip_range)
{
    (void)comp_id;
    filter_host(7, classifier, rec_dst, dst_stride, rec_src, src_stride, blk, filter_set, clip_range, "xeve_hip_alf_filter_blk_7_host");
}
extern "C" void xeve_hip_alf_filter_blk_5_host(uint8_t **classifier, xeve_hip_pel *rec_dst, int dst_stride, const xeve_hip_pel *rec_src, int src_stride,
                                               const xeve_hip_alf_area *blk, uint8_t comp_id, short *filter_set, const xeve_hip_alf_clip_range *clip_range)
{
    (void)comp_id;
    filter_host(5, classifier, rec_dst, dst_stride, rec_src, src_stride, blk, filter_set, clip_range, "xeve_hip_alf_filter_blk_5_host");
}
extern "C" void xeve_hip_alf_get_blk_stats_host(int taps, xeve_hip_alf_covariance *alf_cov, uint8_t **classifier, const xeve_hip_pel *org0, int org_stride,
                                                const xeve_hip_pel *rec0, int rec_stride, int x, int y, int width, int height)
{
    const char *what = "xeve_hip_alf_get_blk_stats_host";
    if(!xh_ready()) xh_set_error("xeve_hip_init() has not been called"), alf_die(what);
    if(!((taps == 5 || taps == 7) && alf_cov && org0 && rec0 && width > 0 && height > 0 && ((width | height) & 3) == 0)) xh_set_error("invalid argument"), alf_die(what);
    const int w = width, h = height, sw = w + 2 * MG, nclasses = classifier ? 25 : 1, ncoef = taps * taps / 4 + 1;
    for(int c = 0; c < nclasses; c++)
        if(alf_cov[c].num_coef < ncoef || !alf_cov[c].E || !alf_cov[c].y) xh_set_error("a covariance record smaller than the filter shape"), alf_die(what);
    DevBuf rec((size_t)sw * (h + 2 * MG) * sizeof(pel), what), org((size_t)w * h * sizeof(pel), what), cls((size_t)w * h, what), job(sizeof(xeve_hip_alf_area), what);
    DevBuf out((size_t)nclasses * (169 + 13 + 1) * sizeof(double), what);
    rows_to_device(rec.p, rec0, rec_stride, sizeof(pel), y - MG, x - MG, h + 2 * MG, sw, what);
    rows_to_device(org.p, org0, org_stride, sizeof(pel), y, x, h, w, what);
    if(classifier) {
        std::vector<uint8_t> c((size_t)w * h);
        for(int i = 0; i < h; i++) memcpy(c.data() + (size_t)i * w, classifier[y + i] + x, (size_t)w);
        if(hipMemcpy(cls.p, c.data(), c.size(), hipMemcpyHostToDevice) != hipSuccess) xh_set_error("copy to the device"), alf_die(what);
    }
    const xeve_hip_alf_area a = {0, 0, w, h};
    if(hipMemcpy(job.p, &a, sizeof(a), hipMemcpyHostToDevice) != hipSuccess) xh_set_error("copy to the device"), alf_die(what);
    double *dE = (double *)out.p, *dy = dE + (size_t)nclasses * 169, *dp = dy + (size_t)nclasses * 13;
    if(xeve_hip_alf_blk_stats_jobs(taps, classifier ? (const uint8_t *)cls.p : nullptr, w, (const pel *)org.p, w, (const pel *)rec.p + (size_t)MG * sw + MG, sw,
                                   (const xeve_hip_alf_area *)job.p, 1, dE, dy, dp, nullptr) != XEVE_HIP_OK)
        alf_die(what);
    std::vector<double> r((size_t)nclasses * (169 + 13 + 1));
    if(hipMemcpy(r.data(), out.p, r.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) xh_set_error("copy from the device"), alf_die(what);
    const double *E = r.data(), *yv = E + (size_t)nclasses * 169, *pp = yv + (size_t)nclasses * 13;
    for(int c = 0; c < nclasses; c++) {
        for(int k = 0; k < ncoef; k++) {
            for(int l = k; l < ncoef; l++) alf_cov[c].E[k][l] += E[(c * 13 + k) * 13 + l];
            alf_cov[c].y[k] += yv[c * 13 + k];
        }
        alf_cov[c].pix_acc += pp[c];
        for(int k = 1; k < ncoef; k++)
            for(int l = 0; l < k; l++) alf_cov[c].E[k][l] = alf_cov[c].E[l][k];
    }
}
