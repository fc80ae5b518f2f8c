// xeve_amd/csrc/mc.hip -- sub-pel motion-compensation interpolation for gfx950.
//
//   luma   8-tap, 1/16-pel positions   reference: xeve_mc_l_{00,n0,0n,nn}  src_base/xeve_mc.c:99-254
//   chroma 4-tap, 1/32-pel positions   reference: xeve_mc_c_{00,n0,0n,nn}  src_base/xeve_mc.c:259-381
//   average (bi-prediction)            reference: xeve_average_16b_no_clip src_base/xeve_mc.c:449-463
//
// Rounding rules that must be kept bit-exact (xeve_mc.h:36-62, xeve_mc.c:211-214):
//   n0 / 0n : (sum + 0) >> 6, then clip to [0, 2^bd - 1]
//   nn      : horizontal pass over h + taps - 1 rows, (sum + 0) >> shift1 stored as int16
//             (shift1 = min(4, bd - 8)); vertical pass (sum + 2^(shift2-1)) >> shift2, clip
//             (shift2 = max(8, 20 - bd)).
//
// One workgroup per job (one predicted block, or one tile of a whole-picture phase plane).  Every
// thread produces SEG horizontally adjacent pels from two vector loads; the nn case stages the
// horizontal pass in LDS ((h + taps - 1) x w int16, <= 9 KB) and reads it back column-aligned.
#include <algorithm>
#include <cstring>
#include "xh_common.h"
#include "mc_cu.h"

template <int TAPS> struct CoefTab {
    int16_t c[(TAPS == 8 ? 16 : 32)][TAPS]; // passed by value in the kernarg segment (256 B)
};

template <int SEG> struct SegIO;
template <> struct SegIO<8> {
    typedef u32x4 vec;
    static __device__ __forceinline__ vec ld(const pel *p) { return xh_ld8(p); }
    static __device__ __forceinline__ void st(pel *p, vec v) { xh_st8(p, v); }
};
template <> struct SegIO<4> {
    typedef u32x2 vec;
    static __device__ __forceinline__ vec ld(const pel *p) { return xh_ld4(p); }
    static __device__ __forceinline__ void st(pel *p, vec v) { xh_st4(p, v); }
};

template <int SEG, typename V> __device__ __forceinline__ void unpack(V v, int *dst)
{
#pragma unroll
    for(int k = 0; k < SEG / 2; k++) {
        dst[2 * k]     = xh_lo16(v[k]);
        dst[2 * k + 1] = xh_hi16(v[k]);
    }
}
template <int SEG> __device__ __forceinline__ typename SegIO<SEG>::vec pack(const int *src)
{
    typename SegIO<SEG>::vec v;
#pragma unroll
    for(int k = 0; k < SEG / 2; k++) v[k] = xh_pack16(src[2 * k], src[2 * k + 1]);
    return v;
}

// horizontal FIR of SEG outputs starting at row pointer r (already moved back by TAPS/2-1 pels)
template <int TAPS, int SEG> __device__ __forceinline__ void hfir(const pel *r, const int16_t *c, int *acc)
{
    int px[2 * SEG];
    unpack<SEG>(SegIO<SEG>::ld(r), px);
    unpack<SEG>(SegIO<SEG>::ld(r + SEG), px + SEG);
#pragma unroll
    for(int i = 0; i < SEG; i++) {
        int a = 0;
#pragma unroll
        for(int t = 0; t < TAPS; t++) a += (int)c[t] * px[i + t];
        acc[i] = a;
    }
}

__device__ __forceinline__ int clipi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// Jobs of one launch may read different reference pictures: bits 3.. of job.frac then index this table (n == 0: every job reads `ref`).
// Used by the CU prediction driver and the sub-pel searches, which would otherwise need one launch per reference picture.
struct PlaneTab {
    const pel *p[XH_MAX_PLANES];
    int        n;
};
__device__ __forceinline__ const pel *plane_of(const PlaneTab &pt, const pel *ref, int frac) { return pt.n ? pt.p[(frac >> 3) & (XH_MAX_PLANES - 1)] : ref; }
// Several interpolation passes of one shape in ONE launch: blockIdx.y picks the pass -- its job array, destination and plane table (the CU prediction
// driver: both lists of luma in one launch, both lists x {Cb, Cr} in another, instead of six).  n == 0: the single pass described by the plain arguments.
#define XH_MC_VARS 4
struct McMulti {
    const xeve_hip_mc_job *jobs[XH_MC_VARS];
    pel                   *pred[XH_MC_VARS];
    PlaneTab               pt[XH_MC_VARS];
    int                    n;
};

// `jpb` jobs per workgroup: small blocks (8x8 luma = 8 row units) are packed so that all 256 threads have a unit.
// Phase 1 (only for jobs with both filters): horizontal pass of h + TAPS - 1 rows into LDS.  Phase 2: every output
// unit, by the job's own variant.  Jobs of one launch may mix variants (quarter-pel merge / final MC).
//
// OUT = 0: the predicted block is stored (pred + job.pred_off).  OUT = 1 / 2: it is compared on the fly with the original
// block at org + job.pred_off and only the SAD / SSD leaves the CU (me_spel_pattern's xeve_mc_l + xeve_sad_16b,
// xeve_pinter.c:593-627; skip/merge analysis' MC + xeve_ssd_16b, xeve_pinter.c:1437-1458).
template <int TAPS, int SEG, int OUT>
__global__ __launch_bounds__(256) void k_mc(const pel *__restrict__ ref, int s_ref, int s_pred, int njobs, int jpb, int w, int h,
                                            int bit_depth, CoefTab<TAPS> tab, const pel *__restrict__ org, int s_org, int dshift,
                                            void *__restrict__ dist_out, McMulti mv)
{
    static_assert(SEG + TAPS - 1 <= 2 * SEG, "two vector loads must cover the FIR footprint");
    // the pass of this workgroup (always read straight out of the kernel-argument segment: a local copy of a pass's table would live in scratch)
    const PlaneTab &ptv = mv.pt[blockIdx.y];
    const xeve_hip_mc_job *__restrict__ jobs = mv.jobs[blockIdx.y];
    pel *__restrict__ pred = mv.pred[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) int16_t hbuf[]; // jpb * (h + TAPS - 1) * w
    constexpr int FS = TAPS == 8 ? 4 : 5, FM = (1 << FS) - 1, BACK = TAPS / 2 - 1;
    typedef typename SegIO<SEG>::vec vec;
    const int j0 = xh_xcd_block(blockIdx.x, gridDim.x) * jpb;
    const int nj = min(jpb, njobs - j0);
    if(nj <= 0) return;
    const int maxv = (1 << bit_depth) - 1, segs = w / SEG, rows = h + TAPS - 1;
    // per-job distortion accumulators live behind the horizontal-pass buffer
    unsigned long long *dacc = reinterpret_cast<unsigned long long *>(hbuf + (((size_t)jpb * rows * w + 3) & ~(size_t)3));
    if(OUT != 0)
        for(int i = threadIdx.x; i < nj; i += blockDim.x) dacc[i] = 0;
    const int shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4;
    const int shift2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8;
    const int round2 = 1 << (shift2 - 1);

    const int per1 = rows * segs;
    for(int u = threadIdx.x; u < nj * per1; u += blockDim.x) {
        const int jl = u / per1, r = u - jl * per1, y = r / segs, x0 = (r - y * segs) * SEG;
        const xeve_hip_mc_job jb = jobs[j0 + jl];
        if((jb.frac & 7) != 3) continue; // bit 2 = job switched off (xeve_hip_mc_cu_jobs)
        const pel *src = plane_of(ptv, ref, jb.frac) + (long)((jb.gmv_y >> FS) + y - BACK) * s_ref + (jb.gmv_x >> FS) + x0 - BACK;
        int acc[SEG];
        hfir<TAPS, SEG>(src, tab.c[jb.gmv_x & FM], acc);
#pragma unroll
        for(int i = 0; i < SEG; i++) acc[i] = (int)(int16_t)(acc[i] >> shift1);
        *reinterpret_cast<vec *>(hbuf + (jl * rows + y) * w + x0) = pack<SEG>(acc);
    }
    __syncthreads();
    const int per2 = h * segs;
    for(int u = threadIdx.x; u < nj * per2; u += blockDim.x) {
        const int jl = u / per2, r = u - jl * per2, y = r / segs, x0 = (r - y * segs) * SEG;
        const xeve_hip_mc_job jb = jobs[j0 + jl];
        if(jb.frac & 4) continue;
        const bool hx = (jb.frac & 1) != 0, vy = (jb.frac & 2) != 0;
        const pel *src = plane_of(ptv, ref, jb.frac) + (long)((jb.gmv_y >> FS) + y) * s_ref + (jb.gmv_x >> FS) + x0;
        pel *out = pred + jb.pred_off + y * s_pred + x0;
        int acc[SEG];
        if(!hx && !vy) {
            if(OUT == 0) {
                SegIO<SEG>::st(out, SegIO<SEG>::ld(src));
                continue;
            }
            unpack<SEG>(SegIO<SEG>::ld(src), acc);
        }
        else
        if(hx && !vy) {
            hfir<TAPS, SEG>(src - BACK, tab.c[jb.gmv_x & FM], acc);
#pragma unroll
            for(int i = 0; i < SEG; i++) acc[i] = clipi(acc[i] >> 6, maxv);
        }
        else {
            const int16_t *cy = tab.c[jb.gmv_y & FM];
#pragma unroll
            for(int i = 0; i < SEG; i++) acc[i] = 0;
            if(!hx) {
#pragma unroll
                for(int t = 0; t < TAPS; t++) {
                    int px[SEG];
                    unpack<SEG>(SegIO<SEG>::ld(src + (t - BACK) * s_ref), px);
#pragma unroll
                    for(int i = 0; i < SEG; i++) acc[i] += (int)cy[t] * px[i];
                }
#pragma unroll
                for(int i = 0; i < SEG; i++) acc[i] = clipi(acc[i] >> 6, maxv);
            }
            else {
#pragma unroll
                for(int t = 0; t < TAPS; t++) {
                    int px[SEG];
                    unpack<SEG>(*reinterpret_cast<const vec *>(hbuf + (jl * rows + y + t) * w + x0), px);
#pragma unroll
                    for(int i = 0; i < SEG; i++) acc[i] += (int)cy[t] * px[i];
                }
#pragma unroll
                for(int i = 0; i < SEG; i++) acc[i] = clipi((acc[i] + round2) >> shift2, maxv);
            }
        }
        if(OUT == 0) SegIO<SEG>::st(out, pack<SEG>(acc));
        else {
            int o[SEG];
            unpack<SEG>(SegIO<SEG>::ld(org + ((jb.frac & XH_FRAC_HALF) ? (size_t)(uint32_t)jb.pred_off << 1 : xh_u(jb.pred_off)) + (long)y * s_org + x0), o);
            unsigned long long dsum = 0;
#pragma unroll
            for(int i = 0; i < SEG; i++) {
                const int df = o[i] - acc[i];
                dsum += OUT == 1 ? (unsigned)(df < 0 ? -df : df) : (unsigned)((df * df) >> dshift);
            }
            atomicAdd(&dacc[jl], dsum);
        }
    }
    if(OUT != 0) {
        __syncthreads();
        for(int i = threadIdx.x; i < nj; i += blockDim.x) {
            if(OUT == 1) static_cast<int32_t *>(dist_out)[j0 + i] = (int32_t)(dacc[i] >> dshift);
            else static_cast<int64_t *>(dist_out)[j0 + i] = (int64_t)dacc[i];
        }
    }
}

// Fallback for widths that are not a multiple of 4 (or luma width 4): one thread per output pel.
template <int TAPS>
__global__ void k_mc_any(const pel *__restrict__ ref, int s_ref, pel *__restrict__ pred, int s_pred,
                         const xeve_hip_mc_job *__restrict__ jobs, int w, int h, int bit_depth, CoefTab<TAPS> tab, PlaneTab pt)
{
    constexpr int FS = TAPS == 8 ? 4 : 5, FM = (1 << FS) - 1, BACK = TAPS / 2 - 1;
    const xeve_hip_mc_job jb = jobs[blockIdx.x];
    if(jb.frac & 4) return; // job switched off
    const int ix = jb.gmv_x >> FS, iy = jb.gmv_y >> FS;
    const int16_t *cx = tab.c[jb.gmv_x & FM], *cy = tab.c[jb.gmv_y & FM];
    const bool hx = (jb.frac & 1) != 0, vy = (jb.frac & 2) != 0;
    const int  maxv = (1 << bit_depth) - 1;
    const int  shift1 = bit_depth - 8 < 4 ? bit_depth - 8 : 4, shift2 = 20 - bit_depth > 8 ? 20 - bit_depth : 8;
    for(int u = threadIdx.x; u < w * h; u += blockDim.x) {
        int y = u / w, x = u % w, v;
        const pel *p = plane_of(pt, ref, jb.frac) + (long)(iy + y) * s_ref + ix + x;
        if(!hx && !vy) v = p[0];
        else if(hx && !vy) {
            int a = 0;
            for(int t = 0; t < TAPS; t++) a += (int)cx[t] * p[t - BACK];
            v = clipi(a >> 6, maxv);
        }
        else if(!hx && vy) {
            int a = 0;
            for(int t = 0; t < TAPS; t++) a += (int)cy[t] * p[(t - BACK) * s_ref];
            v = clipi(a >> 6, maxv);
        }
        else {
            int a = 0;
            for(int t = 0; t < TAPS; t++) {
                int hsum = 0;
                for(int k = 0; k < TAPS; k++) hsum += (int)cx[k] * p[(t - BACK) * s_ref + k - BACK];
                a += (int)cy[t] * (int)(int16_t)(hsum >> shift1);
            }
            v = clipi((a + (1 << (shift2 - 1))) >> shift2, maxv);
        }
        pred[jb.pred_off + y * s_pred + x] = (pel)v;
    }
}

__global__ void k_avg(const int16_t *__restrict__ a, const int16_t *__restrict__ b, int16_t *__restrict__ d, long n)
{
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if(i + 8 <= n) {
        u32x4 va = xh_ld8(a + i), vb = xh_ld8(b + i), vd;
#pragma unroll
        for(int k = 0; k < 4; k++)
            vd[k] = xh_pack16((xh_lo16(va[k]) + xh_lo16(vb[k]) + 1) >> 1, (xh_hi16(va[k]) + xh_hi16(vb[k]) + 1) >> 1);
        xh_st8(d + i, vd);
    }
    else {
        for(; i < n; i++) d[i] = (int16_t)(((int)a[i] + (int)b[i] + 1) >> 1);
    }
}

template <int TAPS, int OUT>
static int mc_launch(const pel *ref, int s_ref, pel *pred, int s_pred, const xeve_hip_mc_job *jobs, int njobs, int w, int h,
                     int bit_depth, const int16_t *coef, hipStream_t st, const pel *org = nullptr, int s_org = 0, void *dist_out = nullptr,
                     const PlaneTab *planes = nullptr, const McMulti *multi = nullptr)
{
    XH_ENTER();
    XH_REQUIRE((ref || (planes && planes->n > 0) || multi) && (jobs || multi) && coef && njobs >= 0 && w >= 1 && h >= 1 && w <= 128 && h <= 128);
    XH_REQUIRE(OUT != 0 ? (org && dist_out) : (pred != nullptr || multi != nullptr));
    XH_REQUIRE(!multi || (OUT == 0 && multi->n >= 1 && multi->n <= XH_MC_VARS && (w % (TAPS == 8 ? 8 : 4)) == 0));
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14);
    if(njobs == 0) return XEVE_HIP_OK;
    CoefTab<TAPS> tab;
    memcpy(tab.c, coef, sizeof(tab.c));
    PlaneTab pt;
    if(planes) pt = *planes;
    else pt.n = 0;
    McMulti mv;
    if(multi) mv = *multi;
    else mv.n = 1, mv.jobs[0] = jobs, mv.pred[0] = pred, mv.pt[0] = pt; // the single pass described by the plain arguments
    const unsigned gy = (unsigned)mv.n;
    const int dshift = OUT == 1 ? bit_depth - 8 : (bit_depth - 8) * 2;
    bool done = false;
    if(w % 8 == 0) {
        const int jpb = std::max(1, 256 / (h * (w / 8)));
        const size_t lds = ((sizeof(int16_t) * (size_t)jpb * (h + TAPS - 1) * w + 7) & ~(size_t)7) + 8 * (size_t)jpb;
        k_mc<TAPS, 8, OUT><<<dim3((njobs + jpb - 1) / jpb, gy), 256, lds, st>>>(ref, s_ref, s_pred, njobs, jpb, w, h, bit_depth, tab, org, s_org, dshift, dist_out, mv);
        done = true;
    }
    if constexpr(TAPS == 4) {
        if(!done && w % 4 == 0) {
            const int jpb = std::max(1, 256 / (h * (w / 4)));
            const size_t lds = ((sizeof(int16_t) * (size_t)jpb * (h + TAPS - 1) * w + 7) & ~(size_t)7) + 8 * (size_t)jpb;
            k_mc<4, 4, OUT><<<dim3((njobs + jpb - 1) / jpb, gy), 256, lds, st>>>(ref, s_ref, s_pred, njobs, jpb, w, h, bit_depth, tab, org, s_org, dshift, dist_out, mv);
            done = true;
        }
    }
    if(!done) {
        if(OUT != 0) {
            xh_set_error("fused MC + distortion needs w %% %d == 0 (got %dx%d); use the unfused calls", TAPS == 8 ? 8 : 4, w, h);
            return XEVE_HIP_ERR_ARG;
        }
        k_mc_any<TAPS><<<njobs, 64, 0, st>>>(ref, s_ref, pred, s_pred, jobs, w, h, bit_depth, tab, pt);
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_mc_l_jobs(const pel *ref, int s_ref, pel *pred, int s_pred, const xeve_hip_mc_job *jobs, int njobs,
                                  int w, int h, int bit_depth, const int16_t (*coef)[8], void *stream)
{
    return mc_launch<8, 0>(ref, s_ref, pred, s_pred, jobs, njobs, w, h, bit_depth, coef ? &coef[0][0] : nullptr, (hipStream_t)stream);
}

extern "C" int xeve_hip_mc_c_jobs(const pel *ref, int s_ref, pel *pred, int s_pred, const xeve_hip_mc_job *jobs, int njobs,
                                  int w, int h, int bit_depth, const int16_t (*coef)[4], void *stream)
{
    return mc_launch<4, 0>(ref, s_ref, pred, s_pred, jobs, njobs, w, h, bit_depth, coef ? &coef[0][0] : nullptr, (hipStream_t)stream);
}

extern "C" int xeve_hip_mc_l_sad_jobs(const pel *ref, int s_ref, const pel *org, int s_org, const xeve_hip_mc_job *jobs, int njobs, int w,
                                      int h, int bit_depth, const int16_t (*coef)[8], int32_t *sad, void *stream)
{
    return mc_launch<8, 1>(ref, s_ref, nullptr, 0, jobs, njobs, w, h, bit_depth, coef ? &coef[0][0] : nullptr, (hipStream_t)stream, org, s_org, sad);
}

extern "C" int xeve_hip_mc_ssd_jobs(int luma, const pel *ref, int s_ref, const pel *org, int s_org, const xeve_hip_mc_job *jobs, int njobs,
                                    int w, int h, int bit_depth, const void *coef, int64_t *ssd, void *stream)
{
    if(luma) return mc_launch<8, 2>(ref, s_ref, nullptr, 0, jobs, njobs, w, h, bit_depth, (const int16_t *)coef, (hipStream_t)stream, org, s_org, ssd);
    return mc_launch<4, 2>(ref, s_ref, nullptr, 0, jobs, njobs, w, h, bit_depth, (const int16_t *)coef, (hipStream_t)stream, org, s_org, ssd);
}

extern "C" int xeve_hip_avg(const int16_t *a, const int16_t *b, int16_t *dst, int64_t n, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(a && b && dst && n >= 0);
    if(n == 0) return XEVE_HIP_OK;
    const long threads = (n + 7) / 8;
    k_avg<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, (hipStream_t)stream>>>(a, b, dst, n);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// =========================================================================================================
// me_spel_pattern on the GPU (xeve_hip_me_spel_pattern_jobs): per stage, ALL candidates of ALL jobs go through one
// fused interpolation + SAD launch (k_mc<8, 8, 1>); two elementwise kernels build the candidate jobs and pick the
// winner in the reference's evaluation order.  reference: src_base/xeve_pinter.c:553-697.
// =========================================================================================================
__device__ __constant__ int8_t c_pat_hpel[8][2] = {{-2, 0}, {-2, 2}, {0, 2}, {2, 2}, {2, 0}, {2, -2}, {0, -2}, {-2, -2}}; // xeve_pinter.c:67-70
__device__ __constant__ int8_t c_pat_qpel[8][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, 1}, {1, 1}, {-1, -1}, {1, -1}}; // xeve_pinter.c:50-55

// stage 0: centre = mvi; stage 1: centre = the half-pel winner stored in res[j].mv
__global__ void k_spel_make(const xeve_hip_spel_job *__restrict__ jobs, int njobs, int cnt, int stage, int s_org, int blk_elems, int bi,
                            const xeve_hip_me_result *__restrict__ res, xeve_hip_mc_job *__restrict__ mc, int per_plane, const unsigned char *__restrict__ job_plane,
                            int vh)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= njobs * cnt) return;
    const int j = t / cnt, i = t - j * cnt;
    const xeve_hip_spel_job jb = jobs[j];
    const int mvx = stage ? res[j].mv[0] : jb.mvi[0], mvy = stage ? res[j].mv[1] : jb.mvi[1];
    const int8_t(*pat)[2] = stage ? c_pat_qpel : c_pat_hpel;
    const int mx = mvx + (jb.x << 2) + pat[i][0], my = mvy + (jb.y << 2) + pat[i][1]; // quarter pel, picture coordinates (of the stacked batch: jb.y carries the picture)
    xeve_hip_mc_job m;
    m.gmv_x = mx << 2, m.gmv_y = my << 2; // 1/16 pel, as the reference passes (mv_x << 2), xeve_pinter.c:608
    (void)vh;
    const size_t oo = (size_t)((long)jb.y * s_org + jb.x); // the block in the stacked originals: halved and marked when even (xh_common.h XH_FRAC_HALF)
    m.pred_off = bi ? jb.org_off : (int)(uint32_t)((oo & 1) ? oo : oo >> 1);
    m.frac = ((mx & 3) != 0 ? 1 : 0) | ((my & 3) != 0 ? 2 : 0);
    if(per_plane) m.frac |= xh_plane_of_job(job_plane, per_plane, j) << 3; // the job's reference picture (PlaneTab)
    if(!bi && !(oo & 1)) m.frac |= XH_FRAC_HALF;
    if(jb.x < 0) m.gmv_x = m.gmv_y = 0, m.pred_off = 0, m.frac = 4; // job switched off
    (void)blk_elems;
    mc[t] = m;
}

struct SpelBits {
    int refi_bits[XH_MAX_PLANES], per_plane;
    const unsigned char *job_plane;
};
__global__ void k_spel_select(const xeve_hip_spel_job *__restrict__ jobs, int njobs, int cnt, int stage, xeve_hip_spel_params P,
                              const int32_t *__restrict__ extra, const int32_t *__restrict__ sad, xeve_hip_me_result *__restrict__ res, SpelBits sb, int vh,
                              XhSpelFinish fin)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= njobs) return;
    if(sb.per_plane) P.refi_bits = sb.refi_bits[xh_plane_of_job(sb.job_plane, sb.per_plane, j)];
    xeve_hip_spel_job jb = jobs[j];
    jb.y -= xh_vh_base(jb.y, vh); // (the predictor jb.gmvp is in the picture's own coordinates)
    xeve_hip_me_result r;
    if(stage) r = res[j];
    else r.mv[0] = jb.mvi[0], r.mv[1] = jb.mvi[1], r.cost = 0xFFFFFFFFu, r.beststep = 0, r.best_mv_bits = 0;
    const int cx = r.mv[0] + (jb.x << 2), cy = r.mv[1] + (jb.y << 2);
    const int8_t(*pat)[2] = stage ? c_pat_qpel : c_pat_hpel;
    for(int i = 0; i < cnt; i++) {
        const int mx = cx + pat[i][0], my = cy + pat[i][1];
        int bits = xh_mvd_bits(mx - jb.gmvp[0]) + xh_mvd_bits(my - jb.gmvp[1]) + P.refi_bits;
        if(P.bi) bits += extra ? extra[j] : P.extra_bits;
        const int s = sad[j * cnt + i];
        const unsigned cost = ((P.lambda_mv * (unsigned)bits + (1u << 15)) >> 16) + (unsigned)(P.bi ? s >> 1 : s);
        if(cost < r.cost) {
            r.mv[0] = (int16_t)(mx - (jb.x << 2)), r.mv[1] = (int16_t)(my - (jb.y << 2)), r.cost = cost;
            if(stage) r.best_mv_bits = bits; // only the quarter-pel stage records the bits (xeve_pinter.c:683)
        }
    }
    res[j] = r;
    if(fin.out) { // the last stage of a pinter_me_epzs call: the search's final result (xeve_pinter.c:828-833; me_spel_pattern's side effect on pi->mot_bits, :690-692)
        const EpzsState z = fin.state[j];
        xeve_hip_me_result o;
        o.cost = z.cost, o.mv[0] = z.mv[0], o.mv[1] = z.mv[1], o.beststep = 0, o.best_mv_bits = z.mot_bits;
        if(!P.bi && r.best_mv_bits > 0) o.best_mv_bits = r.best_mv_bits;
        if(r.cost < o.cost) o.cost = r.cost, o.mv[0] = r.mv[0], o.mv[1] = r.mv[1];
        fin.out[j] = o;
    }
}

extern "C" size_t xeve_hip_me_spel_workspace(int njobs) { return (size_t)(njobs > 0 ? njobs : 0) * 8 * (sizeof(xeve_hip_mc_job) + sizeof(int32_t)); }

extern "C" int xeve_hip_me_spel_pattern_jobs(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref,
                                             const xeve_hip_spel_job *jobs, int njobs, int log2w, int log2h, int bit_depth,
                                             const int16_t (*coef)[8], const xeve_hip_spel_params *params, xeve_hip_me_result *results,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    return xh_me_spel_pattern_jobs_x(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, log2w, log2h, bit_depth, coef, params, nullptr, results, workspace,
                                     workspace_bytes, stream, nullptr);
}

// extra: pi->mot_bits[other list] per job (device memory) instead of params->extra_bits; NULL = the common value
int xh_me_spel_pattern_jobs_x(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_spel_job *jobs, int njobs, int log2w,
                              int log2h, int bit_depth, const int16_t (*coef)[8], const xeve_hip_spel_params *params, const int32_t *extra,
                              xeve_hip_me_result *results, void *workspace, size_t workspace_bytes, void *stream, const XhSearchPlanes *planes, const XhSpelFinish *finish)
{
    XH_ENTER();
    XH_REQUIRE(org0 && (ref0 || (planes && planes->n > 0)) && jobs && coef && params && results && workspace && njobs >= 0);
    PlaneTab pt;
    SpelBits sb;
    pt.n = 0, sb.per_plane = 0, sb.job_plane = nullptr;
    if(planes && planes->n > 0) {
        XH_REQUIRE(planes->n <= XH_MAX_PLANES && planes->per_plane > 0);
        pt.n = planes->n, sb.per_plane = planes->per_plane, sb.job_plane = planes->job_plane;
        for(int i = 0; i < XH_MAX_PLANES; i++) pt.p[i] = planes->ref[i < planes->n ? i : 0], sb.refi_bits[i] = planes->refi_bits[i < planes->n ? i : 0];
    }
    XH_REQUIRE(log2w >= 3 && log2w <= 6 && log2h >= 3 && log2h <= 6);
    XH_REQUIRE(params->hpel_cnt >= 1 && params->hpel_cnt <= 8 && params->qpel_cnt >= 0 && params->qpel_cnt <= 8);
    XH_REQUIRE(params->bi == 0 || org_bi != nullptr);
    XH_REQUIRE(workspace_bytes >= xeve_hip_me_spel_workspace(njobs));
    if(njobs == 0) return XEVE_HIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const xeve_hip_spel_params P = *params;
    const int w = 1 << log2w, h = 1 << log2h;
    xeve_hip_mc_job *mc  = static_cast<xeve_hip_mc_job *>(workspace);
    int32_t         *sad = reinterpret_cast<int32_t *>(mc + (size_t)njobs * 8);
    const pel *cmp = P.bi ? org_bi : org0;
    const int  s_c = P.bi ? w : s_org;
    const int  vh = xh_vh();
    for(int stage = 0; stage < 2; stage++) {
        const int cnt = stage ? P.qpel_cnt : P.hpel_cnt;
        if(cnt == 0) break;
        const int items = njobs * cnt;
        k_spel_make<<<(items + 255) / 256, 256, 0, st>>>(jobs, njobs, cnt, stage, s_org, w * h, P.bi, results, mc, sb.per_plane, sb.job_plane, vh);
        XH_HIP(hipGetLastError());
        int rc = mc_launch<8, 1>(ref0, s_ref, nullptr, 0, mc, items, w, h, bit_depth, &coef[0][0], st, cmp, s_c, sad, pt.n ? &pt : nullptr);
        if(rc != XEVE_HIP_OK) return rc;
        const bool last = stage == 1 || P.qpel_cnt == 0;
        XhSpelFinish fin = {nullptr, nullptr};
        if(finish && last) fin = *finish;
        k_spel_select<<<(njobs + 255) / 256, 256, 0, st>>>(jobs, njobs, cnt, stage, P, extra, sad, results, sb, vh, fin);
        XH_HIP(hipGetLastError());
    }
    return XEVE_HIP_OK;
}

// ---- a8: the CU driver xeve_mc (src_base/xeve_mc.c:465-610) --------------------------------------------------------------
// per job: clip both vectors (xeve_mv_clip), interpolate Y / U / V from every used list -- filter variant from the UNCLIPPED
// vector's fraction, position from the clipped one --, drop list 1 when it repeats list 0 (same POC, same clipped vector),
// average when two predictions remain.  One small kernel turns the jobs into per-(list, reference picture) interpolation jobs
// (switched off where a job does not use that picture); list 0 lands in the caller's buffers, list 1 in the workspace; a
// last kernel averages / copies per job.
__global__ void k_cu_mc_prep(const xeve_hip_cu_mc_job *__restrict__ jobs, CuMcPrep C)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= C.njobs) return;
    xh_cu_mc_prep_one(jobs[j], j, C);
}

__global__ void k_cu_mc_combine(pel *__restrict__ p0, const pel *__restrict__ p1, const uint8_t *__restrict__ mode, int njobs, int n)
{ // n = samples per job (a multiple of 4)
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if(i >= (long)njobs * n) return;
    const int m = mode[i / n];
    if(!m) return;
    u32x2 a = xh_ld4(p0 + i), b = xh_ld4(p1 + i);
    if(m == 1) {
#pragma unroll
        for(int k = 0; k < 2; k++) a[k] = xh_pack16((xh_lo16(a[k]) + xh_lo16(b[k]) + 1) >> 1, (xh_hi16(a[k]) + xh_hi16(b[k]) + 1) >> 1); // xeve_average_16b_no_clip
    }
    else a = b;
    xh_st4(p0 + i, a);
}

// the three components of k_cu_mc_combine in one launch (blockIdx.y = component)
__global__ __launch_bounds__(256) void k_cu_mc_combine3(pel *py, const pel *qy, pel *pu, const pel *qu, pel *pv, const pel *qv, const uint8_t *__restrict__ mode, int njobs,
                                                        int n0, int n1)
{
    const int  k = blockIdx.y, n = k ? n1 : n0;
    pel       *p0 = k == 0 ? py : (k == 1 ? pu : pv);
    const pel *p1 = k == 0 ? qy : (k == 1 ? qu : qv);
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if(i >= (long)njobs * n) return;
    const int m = mode[i / n];
    if(!m) return;
    u32x2 a = xh_ld4(p0 + i), b = xh_ld4(p1 + i);
    if(m == 1) {
#pragma unroll
        for(int q = 0; q < 2; q++) a[q] = xh_pack16((xh_lo16(a[q]) + xh_lo16(b[q]) + 1) >> 1, (xh_hi16(a[q]) + xh_hi16(b[q]) + 1) >> 1); // xeve_average_16b_no_clip
    }
    else a = b;
    xh_st4(p0 + i, a);
}

extern "C" size_t xeve_hip_mc_cu_workspace(int njobs, int w, int h, int num_refp0, int num_refp1)
{
    const size_t n = njobs > 0 ? njobs : 0, q = 2; // one job array per list and plane type
    (void)num_refp0, (void)num_refp1;
    return ((n + 15) & ~(size_t)15) + 2 * q * n * sizeof(xeve_hip_mc_job) + 3 * n * (size_t)w * h * sizeof(pel);
}

// the layout of the prediction workspace and the parameters of the per-CU front half (mc_cu.h)
int xh_mc_cu_prep_params(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int pic_w, int pic_h, int njobs, int w, int h, int chroma_format_idc, void *workspace,
                         size_t workspace_bytes, CuMcPrep *out)
{
    XH_REQUIRE(refp && njobs >= 0 && workspace && out);
    XH_REQUIRE(num_refp0 >= 0 && num_refp0 <= XH_MAX_REF && num_refp1 >= 0 && num_refp1 <= XH_MAX_REF && num_refp0 + num_refp1 > 0);
    XH_REQUIRE(w >= 4 && h >= 4 && w <= 128 && h <= 128 && (w & 3) == 0 && (h & 3) == 0 && chroma_format_idc >= 0 && chroma_format_idc <= 3);
    XH_REQUIRE(workspace_bytes >= xeve_hip_mc_cu_workspace(njobs, w, h, num_refp0, num_refp1));
    const int ws = chroma_format_idc <= 2, hs = chroma_format_idc <= 1; // XEVE_GET_CHROMA_{W,H}_SHIFT
    CuMcK &P = out->P;
    P.vh = xh_vh(), P.pic_w = pic_w, P.pic_h = pic_h, P.w = w, P.h = h, P.cw = w >> ws, P.ch = h >> hs, P.wfac = 2 / (ws + 1), P.hfac = 2 / (hs + 1);
    P.nref[0] = num_refp0, P.nref[1] = num_refp1;
    const int nmax = num_refp0 > num_refp1 ? num_refp0 : num_refp1;
    for(int r = 0; r < XH_MAX_REF; r++)
        for(int l = 0; l < 2; l++) P.poc[l][r] = r < nmax ? refp[r * 2 + l].poc : 0;
    const size_t n = njobs, q = 2;
    out->njobs = njobs;
    out->mode = (uint8_t *)workspace;
    out->jl = (xeve_hip_mc_job *)(out->mode + ((n + 15) & ~(size_t)15)), out->jc = out->jl + q * n;
    out->p1[0] = (pel *)(out->jc + q * n), out->p1[1] = out->p1[0] + n * w * h, out->p1[2] = out->p1[1] + n * P.cw * P.ch;
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_mc_cu_jobs(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h,
                                   const xeve_hip_cu_mc_job *jobs, int njobs, int w, int h, int bit_depth_luma, int bit_depth_chroma,
                                   int chroma_format_idc, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], pel *pred_y, pel *pred_u,
                                   pel *pred_v, void *workspace, size_t workspace_bytes, void *stream)
{
    return xh_mc_cu_jobs_x(refp, num_refp0, num_refp1, s_l, s_c, pic_w, pic_h, jobs, njobs, w, h, bit_depth_luma, bit_depth_chroma, chroma_format_idc, coef_l, coef_c, pred_y,
                           pred_u, pred_v, workspace, workspace_bytes, stream, 0);
}

// flags: XH_MC_PREPPED -- the caller's own kernel has run xh_cu_mc_prep_one for every job (xh_mc_cu_prep_params with the same arguments): no front-half launch, `jobs` is not
// read; XH_MC_LUMA_ONLY -- the luma prediction alone (analyze_bi's prediction from the fixed list, xeve_pinter.c:1618-1624: get_org_bi reads Y only); XH_MC_NO_COMBINE --
// no last kernel: list 0's prediction stays in pred_*, list 1's in CuMcPrep::p1, CuMcPrep::mode says per job what the reader has to make of them
int xh_mc_cu_jobs_x(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pic_w, int pic_h, const xeve_hip_cu_mc_job *jobs, int njobs, int w, int h,
                    int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], pel *pred_y, pel *pred_u, pel *pred_v,
                    void *workspace, size_t workspace_bytes, void *stream, int flags)
{
    XH_ENTER();
    const bool prepped = (flags & XH_MC_PREPPED) != 0, luma_only = (flags & XH_MC_LUMA_ONLY) != 0;
    XH_REQUIRE(refp && (jobs || prepped) && njobs >= 0 && coef_l && pred_y && workspace);
    XH_REQUIRE(chroma_format_idc == 0 || luma_only || (pred_u && pred_v && coef_c));
    if(njobs == 0) return XEVE_HIP_OK;
    CuMcPrep C;
    int rc0 = xh_mc_cu_prep_params(refp, num_refp0, num_refp1, pic_w, pic_h, njobs, w, h, chroma_format_idc, workspace, workspace_bytes, &C);
    if(rc0 != XEVE_HIP_OK) return rc0;
    const CuMcK &P = C.P;
    const size_t n = njobs, q = 2;
    uint8_t         *mode = C.mode;
    xeve_hip_mc_job *jl = C.jl, *jc = C.jc;
    pel             *p1[3] = {C.p1[0], C.p1[1], C.p1[2]};
    hipStream_t st = (hipStream_t)stream;
    XhProf prof(XH_PROF_MC, st);
    if(!prepped) {
        k_cu_mc_prep<<<(njobs + 255) / 256, 256, 0, st>>>(jobs, C);
        XH_HIP(hipGetLastError());
    }
    const bool chroma = chroma_format_idc && !luma_only;
    // two interpolation launches: luma of both lists, then {Cb, Cr} of both lists (blockIdx.y picks list / plane; the jobs pick their reference
    // picture from the pass's table).  Shapes the packed kernel cannot take (chroma width not a multiple of 4) go list by list, plane by plane.
    PlaneTab ty[2], tu[2], tv[2];
    int nl = 0, lists[2] = {0, 0};
    for(int l = 0; l < 2; l++) {
        if(!P.nref[l]) continue;
        lists[nl++] = l;
        ty[l].n = tu[l].n = tv[l].n = P.nref[l];
        for(int r = 0; r < XH_MAX_PLANES; r++) {
            const xeve_hip_refpic &R = refp[(r < P.nref[l] ? r : 0) * 2 + l];
            XH_REQUIRE(R.y && (!chroma || (R.u && R.v)));
            ty[l].p[r] = R.y, tu[l].p[r] = R.u, tv[l].p[r] = R.v;
        }
    }
    const bool packed_l = w % 8 == 0, packed_c = chroma && P.cw % 4 == 0;
    if(packed_l) {
        McMulti mv;
        mv.n = nl;
        for(int i = 0; i < nl; i++) {
            const int l = lists[i];
            mv.jobs[i] = jl + (size_t)l * n, mv.pred[i] = l ? p1[0] : pred_y, mv.pt[i] = ty[l];
        }
        int rc = mc_launch<8, 0>(nullptr, s_l, nullptr, w, nullptr, njobs, w, h, bit_depth_luma, &coef_l[0][0], st, nullptr, 0, nullptr, nullptr, &mv);
        if(rc != XEVE_HIP_OK) return rc;
    }
    if(packed_c) {
        McMulti mv;
        mv.n = 2 * nl;
        for(int i = 0; i < nl; i++) {
            const int l = lists[i];
            mv.jobs[2 * i] = mv.jobs[2 * i + 1] = jc + (size_t)l * n;
            mv.pred[2 * i] = l ? p1[1] : pred_u, mv.pred[2 * i + 1] = l ? p1[2] : pred_v, mv.pt[2 * i] = tu[l], mv.pt[2 * i + 1] = tv[l];
        }
        int rc = mc_launch<4, 0>(nullptr, s_c, nullptr, P.cw, nullptr, njobs, P.cw, P.ch, bit_depth_chroma, &coef_c[0][0], st, nullptr, 0, nullptr, nullptr, &mv);
        if(rc != XEVE_HIP_OK) return rc;
    }
    for(int i = 0; i < nl; i++) {
        const int l = lists[i];
        if(!packed_l) {
            int rc = mc_launch<8, 0>(nullptr, s_l, l ? p1[0] : pred_y, w, jl + (size_t)l * n, njobs, w, h, bit_depth_luma, &coef_l[0][0], st, nullptr, 0, nullptr, &ty[l]);
            if(rc != XEVE_HIP_OK) return rc;
        }
        if(chroma && !packed_c) {
            int rc = mc_launch<4, 0>(nullptr, s_c, l ? p1[1] : pred_u, P.cw, jc + (size_t)l * n, njobs, P.cw, P.ch, bit_depth_chroma, &coef_c[0][0], st, nullptr, 0, nullptr, &tu[l]);
            if(rc != XEVE_HIP_OK) return rc;
            rc = mc_launch<4, 0>(nullptr, s_c, l ? p1[2] : pred_v, P.cw, jc + (size_t)l * n, njobs, P.cw, P.ch, bit_depth_chroma, &coef_c[0][0], st, nullptr, 0, nullptr, &tv[l]);
            if(rc != XEVE_HIP_OK) return rc;
        }
    }
    if(num_refp1 > 0 && !(flags & XH_MC_NO_COMBINE)) {
        const long tl = ((long)njobs * w * h) / 4;
        if(chroma)
            k_cu_mc_combine3<<<dim3((unsigned)((tl + 255) / 256), 3), 256, 0, st>>>(pred_y, p1[0], pred_u, p1[1], pred_v, p1[2], mode, njobs, w * h, P.cw * P.ch);
        else k_cu_mc_combine<<<(unsigned)((tl + 255) / 256), 256, 0, st>>>(pred_y, p1[0], mode, njobs, w * h);
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// ---- host-memory form of one xeve_mc call (the table layer's style): what pi->fn_mc (pinter_mc, xeve_pinter.c:2058-2085) can be pointed at ------
// refp: table [refi * 2 + list] of HOST plane pointers (sample (0, 0)); the planes extend pad_l / pad_c samples around the picture.  Only the
// (at most two) pictures the job uses are staged.
extern "C" int xeve_hip_mc_cu_host(const xeve_hip_refpic *refp, int num_refp0, int num_refp1, int s_l, int s_c, int pad_l, int pad_c, int pic_w, int pic_h,
                                   const xeve_hip_cu_mc_job *job, int w, int h, int bit_depth_luma, int bit_depth_chroma, int chroma_format_idc,
                                   const int16_t (*coef_l)[8], const int16_t (*coef_c)[4], pel *pred_y, pel *pred_u, pel *pred_v)
{
    XH_ENTER();
    XH_REQUIRE(refp && job && pred_y && num_refp0 >= 0 && num_refp0 <= XH_MAX_REF && num_refp1 >= 0 && num_refp1 <= XH_MAX_REF);
    const int idc = chroma_format_idc, ws = idc <= 2, hs = idc <= 1, nmax = num_refp0 > num_refp1 ? num_refp0 : num_refp1;
    const size_t el = (size_t)s_l * (pic_h + 2 * pad_l), ec = idc ? (size_t)s_c * ((pic_h >> hs) + 2 * pad_c) : 0;
    const size_t ol = (size_t)pad_l * s_l + pad_l, oc = (size_t)pad_c * s_c + pad_c;
    const size_t n0 = (size_t)w * h, n1 = idc ? n0 >> (ws + hs) : 0;
    xeve_hip_refpic tab[2 * XH_MAX_REF];
    pel *staged[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    memset(tab, 0, sizeof(tab));
    int rc = XEVE_HIP_OK;
    for(int r = 0; r < nmax; r++)
        for(int l = 0; l < 2; l++) tab[r * 2 + l].poc = refp[r * 2 + l].poc;
    for(int l = 0; l < 2 && rc == XEVE_HIP_OK; l++) {
        const int ri = job->refi[l];
        if(ri < 0) continue;
        XH_REQUIRE(ri < (l ? num_refp1 : num_refp0));
        const xeve_hip_refpic &src = refp[ri * 2 + l];
        const pel *hp[3] = {src.y, src.u, src.v};
        for(int c = 0; c < (idc ? 3 : 1) && rc == XEVE_HIP_OK; c++) {
            const size_t e = c ? ec : el, o = c ? oc : ol;
            if(hipMalloc((void **)&staged[l][c], e * sizeof(pel)) != hipSuccess || hipMemcpy(staged[l][c], hp[c] - o, e * sizeof(pel), hipMemcpyHostToDevice) != hipSuccess) {
                xh_set_error("xeve_hip_mc_cu_host: staging a reference plane failed");
                rc = XEVE_HIP_ERR_DEVICE;
            }
        }
        // every picture a list holds must be addressable for the per-picture launches; unused ones alias the staged one (their jobs are switched off)
        for(int r = 0; r < (l ? num_refp1 : num_refp0); r++)
            tab[r * 2 + l].y = staged[l][0] + ol, tab[r * 2 + l].u = idc ? staged[l][1] + oc : nullptr, tab[r * 2 + l].v = idc ? staged[l][2] + oc : nullptr;
    }
    // a list the job does not use still needs valid pointers for its (switched-off) launches: borrow the other list's planes
    for(int l = 0; l < 2; l++)
        if(job->refi[l] < 0)
            for(int r = 0; r < (l ? num_refp1 : num_refp0); r++) tab[r * 2 + l].y = tab[(job->refi[1 - l]) * 2 + (1 - l)].y, tab[r * 2 + l].u = tab[(job->refi[1 - l]) * 2 + (1 - l)].u, tab[r * 2 + l].v = tab[(job->refi[1 - l]) * 2 + (1 - l)].v;
    const size_t wsb = xeve_hip_mc_cu_workspace(1, w, h, num_refp0, num_refp1), o_pred = 256, o_ws = (o_pred + (n0 + 2 * n1) * sizeof(pel) + 255) & ~(size_t)255;
    char *d = nullptr;
    if(rc == XEVE_HIP_OK && (hipMalloc((void **)&d, o_ws + wsb) != hipSuccess || hipMemcpy(d, job, sizeof(*job), hipMemcpyHostToDevice) != hipSuccess)) {
        xh_set_error("xeve_hip_mc_cu_host: staging failed");
        rc = XEVE_HIP_ERR_DEVICE;
    }
    pel *dp = (pel *)(d + o_pred);
    if(rc == XEVE_HIP_OK)
        rc = xeve_hip_mc_cu_jobs(tab, num_refp0, num_refp1, s_l, s_c, pic_w, pic_h, (const xeve_hip_cu_mc_job *)d, 1, w, h, bit_depth_luma, bit_depth_chroma, idc, coef_l,
                                 coef_c, dp, dp + n0, dp + n0 + n1, d + o_ws, wsb, nullptr);
    if(rc == XEVE_HIP_OK) {
        if(hipMemcpy(pred_y, dp, n0 * sizeof(pel), hipMemcpyDeviceToHost) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE;
        if(idc && (hipMemcpy(pred_u, dp + n0, n1 * sizeof(pel), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(pred_v, dp + n0 + n1, n1 * sizeof(pel), hipMemcpyDeviceToHost) != hipSuccess))
            rc = XEVE_HIP_ERR_DEVICE;
        if(rc != XEVE_HIP_OK) xh_set_error("xeve_hip_mc_cu_host: copy back failed");
    }
    (void)hipFree(d);
    for(int l = 0; l < 2; l++)
        for(int c = 0; c < 3; c++) (void)hipFree(staged[l][c]);
    return rc;
}
