// xeve_amd/csrc/sad.hip -- block distortion kernels for gfx950 (wave64).
//
//   SAD   reference semantics: sad_16b     src_base/xeve_sad.c:40-61
//   SSD                        ssd_16b     src_base/xeve_sad.c:275-297   (shift per pixel, s64 sum)
//   SATD                       xeve_had    src_base/xeve_sad.c:394-1140  (per-tile Hadamard, DC >> 2)
//   DIFF                       diff_16b    src_base/xeve_sad.c:160-178
//
// Work decomposition (all three distortions): ONE WAVE PER JOB.  A job is one block of plane 1
// (the original) against `ncand` displaced blocks of plane 2 (the candidates of one motion-search
// round around the job's centre).  Lanes are laid out so that every vector-memory instruction is a
// 16-byte (8-pel) row segment per lane and consecutive lanes walk along the row first, then down the
// rows: an 8x8 block takes 8 lanes, so 8 candidates are evaluated per wave pass; a 64x64 block takes
// the whole wave for 8 passes.  The original's rows stay in registers for all candidates; the
// reference rows come through L1/L2, where neighbouring candidates and neighbouring jobs overlap.
// The per-candidate sum is a DPP butterfly over the lanes of the candidate's group.
#include "xh_common.h"

// ------------------------------------------------------------------------------------------------
// geometry of the fast (square, S = 8..64) path
// ------------------------------------------------------------------------------------------------
template <int S> struct Geo {
    static constexpr int LPR   = S / 8;                       // lanes per block row (8 pels each)
    static constexpr int RPP   = XH_WAVE / LPR;               // block rows covered by one wave pass
    static constexpr int CPP   = RPP >= S ? RPP / S : 1;      // candidates per pass
    static constexpr int NP    = RPP >= S ? 1 : S / RPP;      // passes per candidate
    static constexpr int GROUP = XH_WAVE / CPP;               // lanes that share a candidate
};

// abs-diff accumulation of 8 packed pels.  v_sad_u16 is an UNSIGNED 16-bit SAD; for operands that
// may be negative both sides are biased by 0x8000 first (|a-b| is translation invariant).
template <bool SIGNED> __device__ __forceinline__ int sad8(u32x4 a, u32x4 b, int acc)
{
    if(SIGNED) {
        a ^= 0x80008000u;
        b ^= 0x80008000u;
    }
    acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc);
    acc = __builtin_amdgcn_sad_u16(a.y, b.y, acc);
    acc = __builtin_amdgcn_sad_u16(a.z, b.z, acc);
    acc = __builtin_amdgcn_sad_u16(a.w, b.w, acc);
    return acc;
}

// SPLIT waves share one job (each takes every SPLIT-th group of candidates): large blocks have few jobs
// per picture, so the candidates -- not the jobs -- must supply the waves that hide L2 latency.
// UNROLL candidate groups are in flight per wave (independent loads and accumulators).
//
// Alignment: a vector load whose address is not dword-aligned (odd pel position) runs at roughly a
// third of the aligned rate on gfx950 (tools/probe_align.py).  When the caller supplies `p2s`, a copy of
// plane 2 shifted by one element (p2s[i] == p2[i + 1], xeve_hip_plane_shift1), odd positions are read
// from it at the even address one element to the left, so EVERY load is dword-aligned.
template <int S, bool SIGNED, int SPLIT, int UNROLL, int MODE>
__global__ __launch_bounds__(256) void k_sad_sq(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2,
                                                const pel *__restrict__ p2s, int s2,
                                                const xeve_hip_job *__restrict__ jobs, int njobs,
                                                const int32_t *__restrict__ cand_off, int ncand, int shift,
                                                int32_t *__restrict__ out)
{
    using G        = Geo<S>;
    const int lane = threadIdx.x & 63;
    const int widx = xh_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int job  = widx / SPLIT, part = widx % SPLIT;
    if(job >= njobs) return;
    const XhJob jb = xh_job(jobs[job]);

    const int slot = lane / G::GROUP;                  // which of the CPP candidates of a pass
    const int gl   = lane % G::GROUP;                  // lane inside the candidate group
    const int row0 = gl / G::LPR;                      // block row handled in pass 0
    const int col  = (gl % G::LPR) * 8;                // first pel of this lane's segment

    u32x4 org[G::NP];
#pragma unroll
    for(int p = 0; p < G::NP; p++) org[p] = xh_ld8(p1 + xh_u(jb.off1) + (row0 + p * G::RPP) * s1 + col);

    const long base2 = (long)jb.off2 + row0 * s2 + col; // element index into plane 2
    int32_t   *o     = out + (size_t)job * ncand;
    constexpr int STEP = G::CPP * SPLIT;
    for(int c0 = part * G::CPP; c0 < ncand; c0 += STEP * UNROLL) {
        int  acc[UNROLL];
        bool ok[UNROLL];
        long e[UNROLL];
#pragma unroll
        for(int u = 0; u < UNROLL; u++) {
            const int c = c0 + u * STEP + slot;
            ok[u]       = c < ncand;
            e[u]        = base2 + cand_off[ok[u] ? c : 0];
        }
        u32x4 v[UNROLL][G::NP];
#pragma unroll
        for(int u = 0; u < UNROLL; u++)
#pragma unroll
            for(int p = 0; p < G::NP; p++) {
                const long ei = e[u] + (long)p * G::RPP * s2;
                if(MODE == 1) { // dual plane
                    const pel *r = (ei & 1) ? p2s + (ei - 1) : p2 + ei;
                    v[u][p]      = xh_ld8(r);
                }
                else if(MODE == 2) { // aligned load + funnel shift; the 9th pel comes from the next lane of the row
                    const int  odd = (int)(ei & 1);
                    const pel *r   = p2 + (ei - odd);
                    const u32x4 a  = xh_ld8(r);
                    uint32_t nxt   = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.x, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
                    if((gl % G::LPR) == G::LPR - 1) nxt = *reinterpret_cast<const uint32_t *>(r + 8);
                    const uint32_t sh = odd * 16;
                    v[u][p].x = __builtin_amdgcn_alignbit(a.y, a.x, sh);
                    v[u][p].y = __builtin_amdgcn_alignbit(a.z, a.y, sh);
                    v[u][p].z = __builtin_amdgcn_alignbit(a.w, a.z, sh);
                    v[u][p].w = __builtin_amdgcn_alignbit(nxt, a.w, sh);
                }
                else v[u][p] = xh_ld8(p2 + ei);
            }
#pragma unroll
        for(int u = 0; u < UNROLL; u++) {
            acc[u] = 0;
#pragma unroll
            for(int p = 0; p < G::NP; p++) acc[u] = sad8<SIGNED>(org[p], v[u][p], acc[u]);
            acc[u] = xh_group_sum<G::GROUP>(acc[u]);
            if(gl == 0 && ok[u]) o[c0 + u * STEP + slot] = acc[u] >> shift;
        }
    }
}

// Generic w x h (any table entry, 1..128): one wave per (job, candidate), 2-byte accesses.
__global__ void k_sad_any(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                          const xeve_hip_job *__restrict__ jobs, int njobs, const int32_t *__restrict__ cand_off,
                          int ncand, int w, int h, int shift, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(item >= njobs * ncand) return;
    const XhJob jb = xh_job(jobs[item / ncand]);
    const pel *a = p1 + xh_u(jb.off1), *b = p2 + jb.off2 + cand_off[item % ncand];
    int acc = 0;
    for(int i = lane; i < w * h; i += 64) {
        int y = i / w, x = i - y * w;
        int d = (int)a[y * s1 + x] - (int)b[y * s2 + x];
        acc += d < 0 ? -d : d;
    }
    acc = xh_group_sum<64>(acc);
    if(lane == 0) out[item] = acc >> shift;
}

// Tiny blocks (w, h <= 4: 4x4 chroma of an 8x8 CU, 2x2 / 4x4 intra): one THREAD per (job, candidate).
template <bool SSD>
__global__ void k_dist_tiny(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                            const xeve_hip_job *__restrict__ jobs, long items, const int32_t *__restrict__ cand_off, int ncand,
                            int w, int h, int shift, void *__restrict__ out)
{
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(item >= items) return;
    const XhJob jb = xh_job(jobs[item / ncand]);
    const pel *a = p1 + xh_u(jb.off1), *b = p2 + jb.off2 + cand_off[item % ncand];
    long acc = 0;
    for(int y = 0; y < h; y++)
        for(int x = 0; x < w; x++) {
            const int d = (int)a[y * s1 + x] - (int)b[y * s2 + x];
            acc += SSD ? (long)((d * d) >> shift) : (long)(d < 0 ? -d : d);
        }
    if(SSD) static_cast<int64_t *>(out)[item] = acc;
    else static_cast<int32_t *>(out)[item] = (int32_t)(acc >> shift);
}

// ------------------------------------------------------------------------------------------------
// SSD
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ssd8(u32x4 a, u32x4 b, int sh, uint64_t acc)
{
#pragma unroll
    for(int k = 0; k < 4; k++) {
        int d0 = xh_lo16(a[k]) - xh_lo16(b[k]);
        int d1 = xh_hi16(a[k]) - xh_hi16(b[k]);
        acc += (uint32_t)((d0 * d0) >> sh);
        acc += (uint32_t)((d1 * d1) >> sh);
    }
    return acc;
}

__device__ __forceinline__ uint64_t wave_group_sum64(uint64_t v, int group)
{
    for(int m = 1; m < group; m <<= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

template <int S>
__global__ __launch_bounds__(256) void k_ssd_sq(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                                                const xeve_hip_job *__restrict__ jobs, int njobs,
                                                const int32_t *__restrict__ cand_off, int ncand, int sh,
                                                int64_t *__restrict__ out)
{
    using G        = Geo<S>;
    const int lane = threadIdx.x & 63;
    const int job  = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(job >= njobs) return;
    const XhJob jb = xh_job(jobs[job]);
    const int slot = lane / G::GROUP, gl = lane % G::GROUP, row0 = gl / G::LPR, col = (gl % G::LPR) * 8;
    u32x4 org[G::NP];
#pragma unroll
    for(int p = 0; p < G::NP; p++) org[p] = xh_ld8(p1 + xh_u(jb.off1) + (row0 + p * G::RPP) * s1 + col);
    const pel *base2 = p2 + jb.off2 + row0 * s2 + col;
    for(int c0 = 0; c0 < ncand; c0 += G::CPP) {
        const int c   = c0 + slot;
        uint64_t  acc = 0;
        if(c < ncand) {
            const pel *r = base2 + cand_off[c];
#pragma unroll
            for(int p = 0; p < G::NP; p++) acc = ssd8(org[p], xh_ld8(r + p * G::RPP * s2), sh, acc);
        }
        acc = wave_group_sum64(acc, G::GROUP);
        if(gl == 0 && c < ncand) out[(size_t)job * ncand + c] = (int64_t)acc;
    }
}

__global__ void k_ssd_any(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                          const xeve_hip_job *__restrict__ jobs, int njobs, const int32_t *__restrict__ cand_off,
                          int ncand, int w, int h, int sh, int64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(item >= njobs * ncand) return;
    const XhJob jb = xh_job(jobs[item / ncand]);
    const pel *a = p1 + xh_u(jb.off1), *b = p2 + jb.off2 + cand_off[item % ncand];
    uint64_t acc = 0;
    for(int i = lane; i < w * h; i += 64) {
        int y = i / w, x = i - y * w;
        int d = (int)a[y * s1 + x] - (int)b[y * s2 + x];
        acc += (int64_t)((d * d) >> sh);
    }
    acc = wave_group_sum64(acc, 64);
    if(lane == 0) out[item] = (int64_t)acc;
}

// ------------------------------------------------------------------------------------------------
// SATD
// ------------------------------------------------------------------------------------------------
// Fast path, square S x S with S a multiple of 8: the block decomposes into 8x8 tiles
// (xeve_sad.c:1103-1113).  With the lane layout above a lane owns 8 horizontally adjacent
// differences of ONE tile row, and the 8 rows of a tile sit LPR lanes apart: horizontal WHT in
// registers, vertical WHT as three lane-xor butterflies.
__device__ __forceinline__ void wht8_regs(int (&v)[8])
{
#pragma unroll
    for(int len = 1; len < 8; len <<= 1) {
#pragma unroll
        for(int i = 0; i < 8; i++) {
            if((i & len) == 0) {
                int a = v[i], b = v[i + len];
                v[i]       = a + b;
                v[i + len] = a - b;
            }
        }
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_satd_sq(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                                                 const xeve_hip_job *__restrict__ jobs, int njobs,
                                                 const int32_t *__restrict__ cand_off, int ncand, int shift,
                                                 int32_t *__restrict__ out)
{
    using G        = Geo<S>;
    const int lane = threadIdx.x & 63;
    const int job  = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(job >= njobs) return;
    const XhJob jb = xh_job(jobs[job]);
    const int slot = lane / G::GROUP, gl = lane % G::GROUP, row0 = gl / G::LPR, col = (gl % G::LPR) * 8;
    u32x4 org[G::NP];
#pragma unroll
    for(int p = 0; p < G::NP; p++) org[p] = xh_ld8(p1 + xh_u(jb.off1) + (row0 + p * G::RPP) * s1 + col);
    const pel *base2  = p2 + jb.off2 + row0 * s2 + col;
    const int  trow   = row0 & 7; // row inside the 8x8 tile (RPP is a multiple of 8 or equals S=8)
    for(int c0 = 0; c0 < ncand; c0 += G::CPP) {
        const int c   = c0 + slot;
        int       acc = 0;
        // all lanes run the butterflies (the partner lanes must be live); inactive slots use zeros
        const pel *r = base2 + (c < ncand ? cand_off[c] : 0);
#pragma unroll
        for(int p = 0; p < G::NP; p++) {
            u32x4 cur = c < ncand ? xh_ld8(r + p * G::RPP * s2) : org[p];
            int   v[8];
#pragma unroll
            for(int k = 0; k < 4; k++) {
                v[2 * k]     = xh_lo16(org[p][k]) - xh_lo16(cur[k]);
                v[2 * k + 1] = xh_hi16(org[p][k]) - xh_hi16(cur[k]);
            }
            wht8_regs(v); // horizontal
#pragma unroll
            for(int m = 1; m < 8; m <<= 1) { // vertical: partner row = trow ^ m, LPR lanes apart per row
                const bool upper = (trow & m) != 0;
#pragma unroll
                for(int k = 0; k < 8; k++) {
                    int other = __shfl_xor(v[k], m * G::LPR, 64);
                    v[k]      = upper ? other - v[k] : v[k] + other;
                }
            }
            int s = 0;
#pragma unroll
            for(int k = 0; k < 8; k++) {
                int a = v[k] < 0 ? -v[k] : v[k];
                if(k == 0 && trow == 0) a >>= 2; // DC of the tile
                s += a;
            }
            // tile total over its 8 rows (lanes LPR apart), normalise per tile, keep it on the tile's row 0
#pragma unroll
            for(int m = 1; m < 8; m <<= 1) s += __shfl_xor(s, m * G::LPR, 64);
            if(trow == 0) acc += (s + 2) >> 2;
        }
        acc = xh_group_sum<G::GROUP>(acc);
        if(gl == 0 && c < ncand) out[(size_t)job * ncand + c] = acc >> shift;
    }
}

// Generic path: one THREAD per tile, any tile kind the reference defines (xeve_sad.c:1051-1135),
// partial sums combined with atomics.  tw/th in {2,4,8,16}; the non-square tiles divide by an
// irrational constant in double precision exactly like the reference (xeve_sad.c:748,885,964,1038).
__global__ void k_satd_tiles(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                             const xeve_hip_job *__restrict__ jobs, int njobs, const int32_t *__restrict__ cand_off,
                             int ncand, int w, int h, int tw, int th, int32_t *__restrict__ acc_out)
{
    const int tiles_x = w / tw, tiles = tiles_x * (h / th);
    const long t      = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= (long)njobs * ncand * tiles) return;
    const int item = (int)(t / tiles), tile = (int)(t % tiles);
    const XhJob jb = xh_job(jobs[item / ncand]);
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const pel *a = p1 + xh_u(jb.off1) + ty * th * s1 + tx * tw;
    const pel *b = p2 + jb.off2 + cand_off[item % ncand] + ty * th * s2 + tx * tw;
    int v[128];
    for(int y = 0; y < th; y++)
        for(int x = 0; x < tw; x++) v[y * tw + x] = (int)a[y * s1 + x] - (int)b[y * s2 + x];
    for(int y = 0; y < th; y++)
        for(int len = 1; len < tw; len <<= 1)
            for(int i = 0; i < tw; i++)
                if((i & len) == 0) {
                    int p = v[y * tw + i], q = v[y * tw + i + len];
                    v[y * tw + i] = p + q, v[y * tw + i + len] = p - q;
                }
    for(int x = 0; x < tw; x++)
        for(int len = 1; len < th; len <<= 1)
            for(int i = 0; i < th; i++)
                if((i & len) == 0) {
                    int p = v[i * tw + x], q = v[(i + len) * tw + x];
                    v[i * tw + x] = p + q, v[(i + len) * tw + x] = p - q;
                }
    int s = (v[0] < 0 ? -v[0] : v[0]) >> 2;
    for(int i = 1; i < tw * th; i++) s += v[i] < 0 ? -v[i] : v[i];
    int r;
    if(tw == 2 && th == 2) r = s;
    else if(tw == 4 && th == 4) r = (s + 1) >> 1;
    else if(tw == 8 && th == 8) r = (s + 2) >> 2;
    else if(tw * th == 128) r = (int)((double)s / (2.0 * 2.8284271247461903)); // 2*sqrt(8)
    else r = (int)((double)s / 2.8284271247461903);                            // sqrt(8)
    atomicAdd(&acc_out[item], r);
}

__global__ void k_shift_inplace(int32_t *v, int n, int shift)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) v[i] >>= shift;
}

// ------------------------------------------------------------------------------------------------
// DIFF
// ------------------------------------------------------------------------------------------------
__global__ void k_diff(const pel *__restrict__ p1, int s1, const pel *__restrict__ p2, int s2,
                       const xeve_hip_job *__restrict__ jobs, int njobs, int w, int h, int16_t *__restrict__ diff)
{
    // one thread per 8-pel (or narrower) row segment
    const int  segs = (w + 7) / 8, per = segs * h;
    const long t    = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= (long)njobs * per) return;
    const int j = (int)(t / per), r = (int)(t % per), y = r / segs, x0 = (r % segs) * 8;
    const XhJob jb = xh_job(jobs[j]);
    const pel *a = p1 + xh_u(jb.off1) + y * s1 + x0, *b = p2 + jb.off2 + y * s2 + x0;
    int16_t   *d = diff + (size_t)j * w * h + y * w + x0;
    if(w - x0 >= 8) {
        u32x4 va = xh_ld8(a), vb = xh_ld8(b), vd;
#pragma unroll
        for(int k = 0; k < 4; k++) vd[k] = xh_pack16(xh_lo16(va[k]) - xh_lo16(vb[k]), xh_hi16(va[k]) - xh_hi16(vb[k]));
        xh_st8(d, vd);
    }
    else {
        for(int x = 0; x < w - x0; x++) d[x] = (int16_t)((int)a[x] - (int)b[x]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the batched API
// ------------------------------------------------------------------------------------------------
static inline dim3 wave_grid(long waves) { return dim3((unsigned)((waves + 3) / 4)); }

#define XH_JOB_ARGS_OK()                                                                          \
    XH_ENTER();                                                                                   \
    XH_REQUIRE(p1 && p2 && jobs && cand_off && out);                                              \
    XH_REQUIRE(njobs >= 0 && ncand >= 1 && w >= 1 && h >= 1 && w <= 128 && h <= 128);            \
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 16);                                                \
    if(njobs == 0) return XEVE_HIP_OK

static int sad_jobs_impl(const pel *p1, int s1, const pel *p2, const pel *p2s, int s2, const xeve_hip_job *jobs, int njobs,
                         const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int flags, int32_t *out,
                         void *stream)
{
    XH_JOB_ARGS_OK();
    XH_REQUIRE(p2s == nullptr || (((uintptr_t)p2 & 3) == 0 && ((uintptr_t)p2s & 3) == 0));
    hipStream_t st    = (hipStream_t)stream;
    const int   shift = bit_depth - 8;
    const bool  sg    = (flags & XEVE_HIP_SRC1_SIGNED) != 0;
    // how odd-pel (non-dword-aligned) reference rows are fetched: 0 plain, 1 dual plane (needs p2s), 2 aligned + funnel
    // flags bits 4-5 (developer override): 0 auto, 1 dual, 2 funnel, 3 plain
    int mode = (flags >> 4) & 3;
    if(mode == 3) mode = 0;
    else if(mode == 0) mode = p2s ? 1 : 0; // measured: the funnel form never beats plain loads (tools/probe_align.py)
    if(mode == 1 && !p2s) mode = 0;
    if(mode == 2 && ((uintptr_t)p2 & 3) != 0) mode = 0;
#define LAUNCH_SQ2(S, SPLIT, UNROLL, SG, MD)                                                                        \
    k_sad_sq<S, SG, SPLIT, UNROLL, MD><<<wave_grid((long)njobs * SPLIT), 256, 0, st>>>(p1, s1, p2, p2s, s2, jobs, njobs, cand_off, ncand, shift, out)
#define LAUNCH_SQ(S, SPLIT, UNROLL)                                                                               \
    do {                                                                                                         \
        if(sg) { if(mode == 1) LAUNCH_SQ2(S, SPLIT, UNROLL, true, 1); else if(mode == 2) LAUNCH_SQ2(S, SPLIT, UNROLL, true, 2); else LAUNCH_SQ2(S, SPLIT, UNROLL, true, 0); } \
        else   { if(mode == 1) LAUNCH_SQ2(S, SPLIT, UNROLL, false, 1); else if(mode == 2) LAUNCH_SQ2(S, SPLIT, UNROLL, false, 2); else LAUNCH_SQ2(S, SPLIT, UNROLL, false, 0); } \
    } while(0)
    // one-candidate calls (the table layer, sub-pel rounds) take the unsplit form
    const bool many = ncand >= 32;
    const int  tune = (flags >> 8) & 15; // developer override of the (SPLIT, UNROLL) choice, tools/probe_tune.py
    // defaults from the sweeps of tools/probe_tune.py on the 90-candidate search round (more loads in flight per wave,
    // and for the large blocks more waves per job, until the L1/TA path saturates)
    if(w == h && w == 8) { if(!many) LAUNCH_SQ(8, 1, 1); else if(tune == 1) LAUNCH_SQ(8, 1, 6); else LAUNCH_SQ(8, 1, 12); }
    else if(w == h && w == 16) { if(!many) LAUNCH_SQ(16, 1, 1); else if(tune == 1) LAUNCH_SQ(16, 2, 8); else LAUNCH_SQ(16, 4, 6); }
    else if(w == h && w == 32) { if(!many) LAUNCH_SQ(32, 1, 1); else if(tune == 1) LAUNCH_SQ(32, 2, 8); else LAUNCH_SQ(32, 4, 8); }
    else if(w == h && w == 64) { if(!many) LAUNCH_SQ(64, 1, 1); else if(tune == 1) LAUNCH_SQ(64, 4, 2); else LAUNCH_SQ(64, 4, 4); }
    else if(w <= 4 && h <= 4) {
        const long items = (long)njobs * ncand;
        k_dist_tiny<false><<<dim3((unsigned)((items + 255) / 256)), 256, 0, st>>>(p1, s1, p2, s2, jobs, items, cand_off, ncand, w, h, shift, out);
    }
    else k_sad_any<<<wave_grid((long)njobs * ncand), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, w, h, shift, out);
#undef LAUNCH_SQ2
#undef LAUNCH_SQ
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_sad_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                                 const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int flags, int32_t *out,
                                 void *stream)
{
    return sad_jobs_impl(p1, s1, p2, nullptr, s2, jobs, njobs, cand_off, ncand, w, h, bit_depth, flags, out, stream);
}

extern "C" int xeve_hip_sad_jobs_dual(const pel *p1, int s1, const pel *p2, const pel *p2_shift1, int s2,
                                      const xeve_hip_job *jobs, int njobs, const int32_t *cand_off, int ncand, int w, int h,
                                      int bit_depth, int flags, int32_t *out, void *stream)
{
    XH_REQUIRE(p2_shift1 != nullptr);
    return sad_jobs_impl(p1, s1, p2, p2_shift1, s2, jobs, njobs, cand_off, ncand, w, h, bit_depth, flags, out, stream);
}

// dst[i] = src[i + 1] for i in [0, n - 1), dst[n - 1] = 0
__global__ void k_shift1(const pel *__restrict__ src, pel *__restrict__ dst, long n)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if(i + 2 < n) {
        const uint32_t a = *reinterpret_cast<const uint32_t *>(src + i), b = *reinterpret_cast<const uint32_t *>(src + i + 2);
        *reinterpret_cast<uint32_t *>(dst + i) = (a >> 16) | (b << 16);
    }
    else {
        for(long k = i; k < n; k++) dst[k] = k + 1 < n ? src[k + 1] : (pel)0;
    }
}

extern "C" int xeve_hip_plane_shift1(const pel *src, pel *dst, int64_t n, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(src && dst && n >= 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0);
    if(n == 0) return XEVE_HIP_OK;
    const long threads = (n + 1) / 2;
    k_shift1<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, (hipStream_t)stream>>>(src, dst, n);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_ssd_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                                 const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int64_t *out, void *stream)
{
    XH_JOB_ARGS_OK();
    hipStream_t st = (hipStream_t)stream;
    const int   sh = (bit_depth - 8) * 2;
    if(w == h && w == 8) k_ssd_sq<8><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, sh, out);
    else if(w == h && w == 16) k_ssd_sq<16><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, sh, out);
    else if(w == h && w == 32) k_ssd_sq<32><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, sh, out);
    else if(w == h && w == 64) k_ssd_sq<64><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, sh, out);
    else if(w <= 4 && h <= 4) {
        const long items = (long)njobs * ncand;
        k_dist_tiny<true><<<dim3((unsigned)((items + 255) / 256)), 256, 0, st>>>(p1, s1, p2, s2, jobs, items, cand_off, ncand, w, h, sh, out);
    }
    else k_ssd_any<<<wave_grid((long)njobs * ncand), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, w, h, sh, out);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_satd_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs,
                                  const int32_t *cand_off, int ncand, int w, int h, int bit_depth, int32_t *out, void *stream)
{
    XH_JOB_ARGS_OK();
    hipStream_t st    = (hipStream_t)stream;
    const int   shift = bit_depth - 8;
    if(w == h && (w == 8 || w == 16 || w == 32 || w == 64)) {
        if(w == 8) k_satd_sq<8><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, shift, out);
        else if(w == 16) k_satd_sq<16><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, shift, out);
        else if(w == 32) k_satd_sq<32><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, shift, out);
        else k_satd_sq<64><<<wave_grid(njobs), 256, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, shift, out);
        XH_HIP(hipGetLastError());
        return XEVE_HIP_OK;
    }
    // tile selection: same precedence as xeve_had (xeve_sad.c:1051-1135)
    int tw, th;
    if(w > h && (h & 7) == 0 && (w & 15) == 0) tw = 16, th = 8;
    else if(w < h && (w & 7) == 0 && (h & 15) == 0) tw = 8, th = 16;
    else if(w > h && (h & 3) == 0 && (w & 7) == 0) tw = 8, th = 4;
    else if(w < h && (w & 3) == 0 && (h & 7) == 0) tw = 4, th = 8;
    else if((w % 8 == 0) && (h % 8 == 0)) tw = 8, th = 8;
    else if((w % 4 == 0) && (h % 4 == 0)) tw = 4, th = 4;
    else if((w % 2 == 0) && (h % 2 == 0)) tw = 2, th = 2;
    else {
        xh_set_error("satd: %dx%d has no Hadamard tiling (the reference asserts here, xeve_sad.c:1136)", w, h);
        return XEVE_HIP_ERR_ARG;
    }
    const long items = (long)njobs * ncand, threads = items * (w / tw) * (h / th);
    XH_HIP(hipMemsetAsync(out, 0, sizeof(int32_t) * items, st));
    k_satd_tiles<<<dim3((unsigned)((threads + 127) / 128)), 128, 0, st>>>(p1, s1, p2, s2, jobs, njobs, cand_off, ncand, w, h, tw, th, out);
    if(shift) k_shift_inplace<<<dim3((unsigned)((items + 255) / 256)), 256, 0, st>>>(out, (int)items, shift);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_diff_jobs(const pel *p1, int s1, const pel *p2, int s2, const xeve_hip_job *jobs, int njobs, int w,
                                  int h, int16_t *diff, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(p1 && p2 && jobs && diff && njobs >= 0 && w >= 1 && h >= 1 && w <= 128 && h <= 128);
    if(njobs == 0) return XEVE_HIP_OK;
    const long threads = (long)njobs * ((w + 7) / 8) * h;
    k_diff<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, (hipStream_t)stream>>>(p1, s1, p2, s2, jobs, njobs, w, h, diff);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
