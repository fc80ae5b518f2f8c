// xeve_amd/csrc/tq.hip -- integer DCT (forward / inverse), quantisation, dequantisation, reconstruction.
//
//   forward 1-D   tx_pb{2..64}b        reference: src_base/xeve_tq.c:40-392     (partial butterflies)
//   forward 2-D   xeve_trans           reference: src_base/xeve_tq.c:396-404
//   inverse 1-D   xeve_itx_pb{2..64}b  reference: src_base/xeve_itdq.c:34-430
//   inverse 2-D   xeve_itrans          reference: src_base/xeve_itdq.c:435-440
//   quant         xeve_quant_nnz       reference: src_base/xeve_tq.c:651-727    (zero pre-test + plain branch)
//   dequant       xeve_dquant/itdq_cu  reference: src_base/xeve_itdq.c:442-475
//   recon         xeve_recon_blk       reference: src_base/xeve_recon.c:34-57
//
// The reference's partial butterflies are an exact integer factorisation of y = M x with the EVC
// DCT-II matrix M (src_base/xeve_tbl.c:83-236), and pass 1 of the 2-D transforms uses shift 0, so a
// 2-D transform is ONE exact integer double product followed by ONE rounding shift:
//   forward:  C = (Mh * X * Mw^T + 2^(s-1)) >> s   (64-point: rows/cols >= 32 forced to 0)
//   inverse:  X = clip16((Mw^T-side sum of clip32(Mh^T-side sum) + 2^(s-1)) >> s)
// Any exact evaluation order is bit-identical; we evaluate the products directly.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "xh_common.h"

int xh_dct_mfma_init(const int8_t *m32, const int8_t *m64);
int xh_dct_mfma(bool fwd, int16_t *coef, int nblk, int n, int shift, hipStream_t st);
static bool g_use_mfma = true, g_use_rows = true;

// All six matrices, row-major [k][x], concatenated; offset of size 2^l is XH_TM_OFF(l).
__device__ __constant__ int8_t c_tm[4 + 16 + 64 + 256 + 1024 + 4096];
// the 4-, 8- and 16-point matrices widened to int32: read with wave-uniform indices (scalar loads) by k_rdo_rows
__device__ __constant__ int32_t c_tm32[4 + 16 + 64 + 256];
__host__ __device__ constexpr int xh_tm_off(int log2n) { return ((1 << (2 * log2n)) - 4) / 3; } // 0,4,20,84,340,1364

// EVC integer DCT-II: M_N[k][x] = +-g[fold((2x+1) * k * 64/N mod 256)], g[j] = round(64*sqrt(2)*cos(j*pi/128)).
int xh_tq_init()
{
    static int8_t tm[4 + 16 + 64 + 256 + 1024 + 4096];
    int g[65];
    g[0] = 64, g[64] = 0;
    for(int j = 1; j < 64; j++) g[j] = (int)floor(64.0 * 1.4142135623730951 * cos(j * 3.14159265358979323846 / 128.0) + 0.5);
    for(int l = 1; l <= 6; l++) {
        const int n = 1 << l;
        for(int k = 0; k < n; k++)
            for(int x = 0; x < n; x++) {
                int th = ((2 * x + 1) * k * (64 / n)) % 256, sg = 1;
                if(th > 128) th = 256 - th;
                if(th > 64) sg = -1, th = 128 - th;
                tm[xh_tm_off(l) + k * n + x] = (int8_t)(sg * g[th]);
            }
    }
    XH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tm), tm, sizeof(tm)));
    static int32_t tm32[4 + 16 + 64 + 256];
    for(int i = 0; i < 4 + 16 + 64 + 256; i++) tm32[i] = tm[i];
    XH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tm32), tm32, sizeof(tm32)));
    const char *e = getenv("XEVE_HIP_DCT"); // developer switch: "valu" forces the LDS/VALU path for 32/64 too
    g_use_mfma    = !(e && strcmp(e, "valu") == 0);
    g_use_rows    = g_use_mfma; // "valu" also selects the generic LDS form of the fused chain for every size
    return xh_dct_mfma_init(tm + xh_tm_off(5), tm + xh_tm_off(6));
}

// ---- 1-D (table-layer granularity): one thread per output -------------------------------------------
template <bool FWD>
__global__ void k_tx1d(const void *__restrict__ src, void *__restrict__ dst, int log2n, int shift, int line, int step)
{
    const int n = 1 << log2n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n * line) return;
    const int8_t *m = c_tm + xh_tm_off(log2n);
    const int64_t add = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    int64_t acc = 0;
    if(FWD) { // dst[k*line + j] = sum_x M[k][x] * src[j*n + x]
        const int k = i / line, j = i % line;
        if(!(n == 64 && k >= 32)) {
            for(int x = 0; x < n; x++) {
                int64_t v = step != 1 ? (int64_t)((const int16_t *)src)[j * n + x] : (int64_t)((const int32_t *)src)[j * n + x];
                acc += (int64_t)m[k * n + x] * v;
            }
            acc = (acc + add) >> shift;
        }
        if(step == 0) ((int32_t *)dst)[i] = (int32_t)acc;
        else ((int16_t *)dst)[i] = (int16_t)acc;
    }
    else { // dst[j*n + x] = clip(sum_k M[k][x] * src[k*line + j])
        const int j = i / n, x = i % n;
        for(int k = 0; k < n; k++) {
            int64_t v = step != 1 ? (int64_t)((const int16_t *)src)[k * line + j] : (int64_t)((const int32_t *)src)[k * line + j];
            acc += (int64_t)m[k * n + x] * v;
        }
        acc = (acc + add) >> shift;
        if(step == 0) {
            acc = acc < INT32_MIN ? INT32_MIN : (acc > INT32_MAX ? INT32_MAX : acc);
            ((int32_t *)dst)[i] = (int32_t)acc;
        }
        else {
            acc = acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc);
            ((int16_t *)dst)[i] = (int16_t)acc;
        }
    }
}

// ---- 2-D, one workgroup per block, both passes through LDS (VALU path) ----------------------------
template <bool FWD>
__global__ __launch_bounds__(256) void k_trans2d(int16_t *__restrict__ coef, int log2w, int log2h, int shift)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = 1 << log2w, h = 1 << log2h, n = w * h;
    int32_t *T = reinterpret_cast<int32_t *>(smem);          // [kx][y]   (n int32)
    int16_t *X = reinterpret_cast<int16_t *>(smem + 4 * n);  // input block (n int16)
    int16_t *blk = coef + (size_t)blockIdx.x * n;
    const int8_t *mw = c_tm + xh_tm_off(log2w), *mh = c_tm + xh_tm_off(log2h);
    for(int i = threadIdx.x; i < n; i += blockDim.x) X[i] = blk[i];
    __syncthreads();
    const int64_t add = (int64_t)1 << (shift - 1);
    if(FWD) {
        for(int i = threadIdx.x; i < n; i += blockDim.x) { // T[kx][y] = sum_x Mw[kx][x] * X[y][x]
            const int kx = i / h, y = i % h;
            int acc = 0;
            if(!(w == 64 && kx >= 32))
                for(int x = 0; x < w; x++) acc += (int)mw[kx * w + x] * (int)X[y * w + x];
            T[i] = acc;
        }
        __syncthreads();
        for(int i = threadIdx.x; i < n; i += blockDim.x) { // C[ky][kx] = sum_y Mh[ky][y] * T[kx][y]
            const int ky = i / w, kx = i % w;
            int64_t acc = 0;
            if(!(h == 64 && ky >= 32)) {
                for(int y = 0; y < h; y++) acc += (int64_t)mh[ky * h + y] * (int64_t)T[kx * h + y];
                acc = (acc + add) >> shift;
            }
            blk[i] = (int16_t)acc;
        }
    }
    else {
        for(int i = threadIdx.x; i < n; i += blockDim.x) { // T[kx][y] = sum_ky Mh[ky][y] * C[ky][kx]
            const int kx = i / h, y = i % h;
            int acc = 0;
            for(int ky = 0; ky < h; ky++) acc += (int)mh[ky * h + y] * (int)X[ky * w + kx];
            T[i] = acc;
        }
        __syncthreads();
        for(int i = threadIdx.x; i < n; i += blockDim.x) { // X[y][x] = sum_kx Mw[kx][x] * T[kx][y]
            const int y = i / w, x = i % w;
            int64_t acc = 0;
            for(int kx = 0; kx < w; kx++) acc += (int64_t)mw[kx * w + x] * (int64_t)T[kx * h + y];
            acc = (acc + add) >> shift;
            acc = acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc);
            blk[i] = (int16_t)acc;
        }
    }
}

// ---- quantisation family: one workgroup per block -------------------------------------------------
__device__ __forceinline__ int block_sum(int v, int *scratch)
{
    v = xh_group_sum<64>(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    int t = 0;
    for(int i = 0; i < nw; i++) t += scratch[i];
    return t;
}

__global__ void k_quant(int16_t *__restrict__ coef, int n, int scale, int shift, int offset, int32_t *__restrict__ nnz)
{
    __shared__ int scratch[4];
    int16_t *blk = coef + (size_t)blockIdx.x * n;
    int cnt = 0;
    for(int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = blk[i], neg = c < 0;
        int lev = (neg ? -c : c) * scale;
        lev     = (int)(int16_t)((lev + offset) >> shift);
        const int16_t q = (int16_t)(neg ? -lev : lev);
        blk[i] = q;
        cnt += q != 0;
    }
    if(nnz) {
        cnt = block_sum(cnt, scratch);
        if(threadIdx.x == 0) nnz[blockIdx.x] = cnt;
    }
}

__global__ void k_rdoq_zero_test(int16_t *__restrict__ coef, int n, int64_t scale_ns, int64_t thr, int32_t *__restrict__ coded)
{
    __shared__ int scratch[4];
    int16_t *blk = coef + (size_t)blockIdx.x * n;
    int hit = 0;
    for(int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = blk[i];
        hit |= ((int64_t)(c < 0 ? -c : c) * scale_ns) >= thr;
    }
    hit = block_sum(hit, scratch);
    if(!hit)
        for(int i = threadIdx.x; i < n; i += blockDim.x) blk[i] = 0;
    if(threadIdx.x == 0) coded[blockIdx.x] = hit != 0;
}

__global__ void k_dquant(int16_t *__restrict__ coef, long total, int64_t scale_ns, int offset, int shift)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= total) return;
    int64_t lev = ((int64_t)coef[i] * scale_ns + offset) >> shift;
    lev         = lev < -32768 ? -32768 : (lev > 32767 ? 32767 : lev);
    coef[i]     = (int16_t)lev;
}

__global__ void k_recon(const int16_t *__restrict__ coef, const pel *__restrict__ pred, const uint8_t *__restrict__ is_coef,
                        int cuw, int cuh, const int32_t *__restrict__ rec_off, int s_rec, pel *__restrict__ rec, int maxv)
{
    const int b = blockIdx.x, n = cuw * cuh;
    const bool add = is_coef ? is_coef[b] != 0 : true;
    pel *r = rec + rec_off[b];
    for(int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / cuw, x = i % cuw;
        // the reference forms coef + pred in an int16 (wraps) before clipping, xeve_recon.c:50-51
        const int16_t t = add ? (int16_t)(coef[(size_t)b * n + i] + pred[(size_t)b * n + i]) : pred[(size_t)b * n + i];
        r[y * s_rec + x] = (pel)(t < 0 ? 0 : (t > maxv ? maxv : t));
    }
}

// ---- host side ----------------------------------------------------------------------------------------
int xh_tx1d(bool fwd, const void *src, void *dst, int log2n, int shift, int line, int step, hipStream_t st)
{
    const int total = (1 << log2n) * line;
    if(fwd) k_tx1d<true><<<(total + 255) / 256, 256, 0, st>>>(src, dst, log2n, shift, line, step);
    else k_tx1d<false><<<(total + 255) / 256, 256, 0, st>>>(src, dst, log2n, shift, line, step);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

static int trans_common(bool fwd, int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(coef && nblk >= 0 && log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6 && bit_depth >= 8 && bit_depth <= 16);
    if(nblk == 0) return XEVE_HIP_OK;
    const int n = 1 << (log2w + log2h);
    const int threads = n >= 256 ? 256 : 64;
    const size_t lds = (size_t)n * 6;
    // forward: TX_SHIFT1 + TX_SHIFT2 (xeve_util.c:34-35); inverse: ITX_SHIFT1 + ITX_SHIFT2 (xeve_itdq.h:38-39)
    const int shift = fwd ? (log2w - 1 + bit_depth - 8) + (log2h + 6) : 7 + (12 - (bit_depth - 8));
    // 32x32 and 64x64: matrix cores (dct_mfma.hip); everything else: the LDS/VALU kernel
    if(g_use_mfma && log2w == log2h && log2w >= 5) return xh_dct_mfma(fwd, coef, nblk, 1 << log2w, shift, st);
    if(fwd) k_trans2d<true><<<nblk, threads, lds, st>>>(coef, log2w, log2h, shift);
    else k_trans2d<false><<<nblk, threads, lds, st>>>(coef, log2w, log2h, shift);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_trans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream)
{
    return trans_common(true, coef, nblk, log2w, log2h, bit_depth, (hipStream_t)stream);
}
extern "C" int xeve_hip_itrans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream)
{
    return trans_common(false, coef, nblk, log2w, log2h, bit_depth, (hipStream_t)stream);
}

#define XH_Q_ARGS_OK()                                                                                               \
    XH_ENTER();                                                                                                      \
    XH_REQUIRE(coef && nblk >= 0 && log2w >= 1 && log2w <= 7 && log2h >= 1 && log2h <= 7 && bit_depth >= 8 && bit_depth <= 14); \
    if(nblk == 0) return XEVE_HIP_OK

extern "C" int xeve_hip_quant(int16_t *coef, int nblk, int log2w, int log2h, int qp, int scale, int is_intra_slice,
                              int bit_depth, int32_t *nnz, void *stream)
{
    XH_Q_ARGS_OK();
    XH_REQUIRE(qp >= 0 && qp <= 87 && scale > 0 && scale < 65536);
    // xeve_tq.c:716-718 with MAX_TX_DYNAMIC_RANGE 15, QUANT_SHIFT 14 (xeve_def.h:793-797)
    const int log2_size = (log2w + log2h) >> 1;
    const int shift     = 14 + (15 - bit_depth - log2_size) + qp / 6;
    XH_REQUIRE(shift >= 9 && shift <= 30);
    const int offset = (is_intra_slice ? 171 : 85) << (shift - 9);
    const int n      = 1 << (log2w + log2h);
    k_quant<<<nblk, n >= 256 ? 256 : 64, 0, (hipStream_t)stream>>>(coef, n, scale, shift, offset, nnz);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_rdoq_zero_test(int16_t *coef, int nblk, int log2w, int log2h, int qp, int scale, int is_intra_slice,
                                       int bit_depth, int32_t *coded, void *stream)
{
    XH_Q_ARGS_OK();
    XH_REQUIRE(coded && qp >= 0 && qp <= 87 && scale > 0 && scale < 65536);
    // xeve_tq.c:673-683
    const int odd       = (log2w + log2h) & 1;
    const int log2_size = (log2w + log2h) >> 1;
    const int shift     = 14 + (15 - bit_depth - log2_size + (odd ? 7 : 0)) + qp / 6;
    XH_REQUIRE(shift >= 9 && shift <= 40);
    const int64_t offset = (int64_t)(is_intra_slice ? 201 : 153) << (shift - 9);
    const int64_t thr    = ((int64_t)1 << shift) - offset;
    const int     n      = 1 << (log2w + log2h);
    k_rdoq_zero_test<<<nblk, n >= 256 ? 256 : 64, 0, (hipStream_t)stream>>>(coef, n, (int64_t)scale * (odd ? 181 : 1), thr, coded);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_dquant(int16_t *coef, int nblk, int log2w, int log2h, int scale, int bit_depth, void *stream)
{
    XH_Q_ARGS_OK();
    XH_REQUIRE(scale > 0);
    // itdq_cu, xeve_itdq.c:457-473 with QUANT_IQUANT_SHIFT 20, QUANT_SHIFT 14
    const int odd       = (log2w + log2h) & 1;
    const int log2_size = (log2w + log2h) >> 1;
    const int shift     = (uint8_t)(20 - 14 - (15 - bit_depth - log2_size) + (odd ? 8 : 0));
    const int offset    = shift == 0 ? 0 : 1 << (shift - 1);
    const long total    = (long)nblk << (log2w + log2h);
    k_dquant<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(coef, total, (int64_t)scale * (odd ? 181 : 1), offset, shift);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_recon(const int16_t *coef, const pel *pred, const uint8_t *is_coef, int nblk, int cuw, int cuh,
                              const int32_t *rec_off, int s_rec, pel *rec, int bit_depth, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(coef && pred && rec_off && rec && nblk >= 0 && cuw >= 1 && cuh >= 1 && cuw <= 128 && cuh <= 128);
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14);
    if(nblk == 0) return XEVE_HIP_OK;
    k_recon<<<nblk, cuw * cuh >= 256 ? 256 : 64, 0, (hipStream_t)stream>>>(coef, pred, is_coef, cuw, cuh, rec_off, s_rec, rec, (1 << bit_depth) - 1);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// =========================================================================================================
// Fused residual chain (xeve_hip_residual_rdo), LDS/VALU form: any w, h in 2..64.  A workgroup of 256 threads
// holds 256/n blocks (n = w*h <= 256) or one block; every step of pinter_residue_rdo's arithmetic core runs
// on the block while it sits in LDS.  32x32 / 64x64 go to the matrix-core form in dct_mfma.hip instead.
// =========================================================================================================
extern "C" int xeve_hip_rdoq_zt(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                                const xeve_hip_rdoq_est *est, int zero_test, int is_intra_slice, int32_t *nnz, void *stream);
int xh_rdo_mfma(int n, const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, const void *params,
                int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st);

__global__ __launch_bounds__(256) void k_rdo_valu(const pel *__restrict__ org, int s_org, const pel *__restrict__ pred, int s_pred,
                                                  const xeve_hip_job *__restrict__ jobs, int njobs, int log2w, int log2h, RdoParams P,
                                                  int16_t *__restrict__ coef, pel *__restrict__ rec, int s_rec,
                                                  int32_t *__restrict__ nnz_out, int64_t *__restrict__ ssd_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = 1 << log2w, h = 1 << log2h, n = w * h;
    const int tpb = n < 256 ? n : 256, bpw = 256 / tpb;
    const int bl = threadIdx.x / tpb, t = threadIdx.x - bl * tpb;
    const int j  = blockIdx.x * bpw + bl;
    const bool live = j < njobs;
    int32_t *X  = reinterpret_cast<int32_t *>(smem) + (size_t)bl * 2 * n;
    int32_t *Tm = X + n;
    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(smem + (size_t)bpw * 8 * n) + 2 * bl; // [ssd_pred, ssd_rec]
    int *acc32 = reinterpret_cast<int *>(smem + (size_t)bpw * 8 * n + (size_t)bpw * 16) + 2 * bl;              // [zero-test hit, nnz]
    // the two DCT matrices, widened to int32, staged in LDS once per workgroup (lane-varying constant-memory reads
    // would go through the vector memory path)
    int32_t *mw = reinterpret_cast<int32_t *>(smem + (size_t)bpw * 8 * n + (size_t)bpw * 24);
    int32_t *mh = mw + w * w;
    for(int i = threadIdx.x; i < w * w; i += blockDim.x) mw[i] = c_tm[xh_tm_off(log2w) + i];
    for(int i = threadIdx.x; i < h * h; i += blockDim.x) mh[i] = c_tm[xh_tm_off(log2h) + i];
    const XhJob jb = live ? xh_job(jobs[j]) : XhJob{0, 0};
    if(t == 0) acc64[0] = acc64[1] = 0, acc32[0] = acc32[1] = 0;
    __syncthreads();
    // 1. residual + SSD(org, pred)
    if(live && P.stage != 2) {
        unsigned long long s = 0;
        for(int i = t; i < n; i += tpb) {
            const int y = i >> log2w, x = i & (w - 1);
            const int d = (int)org[xh_u(jb.off1) + y * s_org + x] - (int)pred[jb.off2 + y * s_pred + x];
            X[i] = d;
            s += (unsigned)((d * d) >> P.ssd_shift);
        }
        atomicAdd(&acc64[0], s);
    }
    __syncthreads();
    // 2. forward pass 1: Tm[kx][y] = sum_x Mw[kx][x] X[y][x]
    if(live && P.stage != 2)
        for(int i = t; i < n; i += tpb) {
            const int kx = i >> log2h, y = i & (h - 1);
            int a = 0;
            if(!(w == 64 && kx >= 32))
                for(int x = 0; x < w; x++) a += mw[kx * w + x] * X[y * w + x];
            Tm[i] = a;
        }
    __syncthreads();
    // 3. forward pass 2 + zero pre-test: X[ky][kx] = coefficient
    if(live && P.stage != 2) {
        const int64_t add = (int64_t)1 << (P.shift_fwd - 1);
        int hit = 0;
        for(int i = t; i < n; i += tpb) {
            const int ky = i >> log2w, kx = i & (w - 1);
            int64_t a = 0;
            if(!(h == 64 && ky >= 32)) {
                for(int y = 0; y < h; y++) a += (int64_t)mh[ky * h + y] * (int64_t)Tm[kx * h + y];
                a = (a + add) >> P.shift_fwd;
            }
            const int c = (int)(int16_t)a;
            X[i] = c;
            if(P.stage == 1) coef[(size_t)j * n + i] = (int16_t)c;
            hit |= ((int64_t)(c < 0 ? -c : c) * P.z_scale) >= P.z_thr;
        }
        if(P.z_thr < 0) hit = 1;
        if(hit) atomicOr(&acc32[0], 1);
    }
    __syncthreads();
    if(P.stage == 1) { // front half ends here
        if(live && t == 0) ssd_out[2 * j] = (int64_t)acc64[0];
        return;
    }
    // 4. quant, levels out, dequant
    if(live) {
        const bool hit = acc32[0] != 0;
        int cnt = 0;
        for(int i = t; i < n; i += tpb) {
            int lev = 0;
            if(P.stage == 2) lev = coef[(size_t)j * n + i];
            else if(hit) {
                const int c = X[i], neg = c < 0;
                lev = (int)(int16_t)((((neg ? -c : c) * P.q_scale) + P.q_offset) >> P.q_shift);
                lev = (int)(int16_t)(neg ? -lev : lev);
            }
            cnt += lev != 0;
            if(P.stage == 0) coef[(size_t)j * n + i] = (int16_t)lev;
            int64_t dq = ((int64_t)lev * P.dq_scale + P.dq_offset) >> P.dq_shift;
            X[i] = (int)(dq < -32768 ? -32768 : (dq > 32767 ? 32767 : dq));
        }
        if(cnt) atomicAdd(&acc32[1], cnt);
    }
    __syncthreads();
    // 5. inverse pass 1: Tm[kx][y] = sum_ky Mh[ky][y] C[ky][kx]
    if(live)
        for(int i = t; i < n; i += tpb) {
            const int kx = i >> log2h, y = i & (h - 1);
            int a = 0;
            for(int ky = 0; ky < h; ky++) a += (int)mh[ky * h + y] * X[ky * w + kx];
            Tm[i] = a;
        }
    __syncthreads();
    // 6. inverse pass 2, recon, SSD(org, rec)
    if(live) {
        const int64_t add = (int64_t)1 << (P.shift_inv - 1);
        unsigned long long s = 0;
        for(int i = t; i < n; i += tpb) {
            const int y = i >> log2w, x = i & (w - 1);
            int64_t a = 0;
            for(int kx = 0; kx < w; kx++) a += (int64_t)mw[kx * w + x] * (int64_t)Tm[kx * h + y];
            a = (a + add) >> P.shift_inv;
            a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a);
            int r = (int)(int16_t)((int)a + (int)pred[jb.off2 + y * s_pred + x]);
            r     = r < 0 ? 0 : (r > P.maxv ? P.maxv : r);
            rec[(s_rec > 0 ? (long)jb.off1 + (long)y * s_rec : (long)j * n - (long)y * s_rec) + x] = (pel)r; // (s_rec < 0: dense blocks, block j at j * n, pitch -s_rec)
            const int e = (int)org[xh_u(jb.off1) + y * s_org + x] - r;
            s += (unsigned)((e * e) >> P.ssd_shift);
        }
        atomicAdd(&acc64[1], s);
    }
    __syncthreads();
    if(live && t == 0) {
        if(P.stage == 0) nnz_out[j] = acc32[1], ssd_out[2 * j] = (int64_t)acc64[0];
        ssd_out[2 * j + 1] = (int64_t)acc64[1];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused residual chain for square 4x4 / 8x8 / 16x16 blocks, ROW-PER-LANE form: N lanes hold the N rows of a block in
// registers (64/N blocks per wave), the 1-D transforms are fully unrolled MAC chains whose matrix entries come from
// scalar loads, and the only exchanges are two N x N transposes through a wave-private LDS tile (no workgroup
// barriers).  64-bit sums of pass 2 are formed exactly as two 32-bit chains over the 16-bit halves of the operand.
// ---------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ int tm32_at(int k, int x) { return c_tm32[xh_tm_off(N == 4 ? 2 : (N == 8 ? 3 : 4)) + k * N + x]; }

// out[k] = sum_x M[k][x] * in[x]   (TRANSPOSED = false)   or   sum_k M[k][x] * in[k]   (TRANSPOSED = true), 32-bit
template <int N, bool TRANSPOSED> __device__ __forceinline__ void mat32(const int (&in)[N], int (&out)[N])
{
#pragma unroll
    for(int o = 0; o < N; o++) {
        int a = 0;
#pragma unroll
        for(int i = 0; i < N; i++) a += (TRANSPOSED ? tm32_at<N>(i, o) : tm32_at<N>(o, i)) * in[i];
        out[o] = a;
    }
}
// same with an exact 64-bit result: in = hi * 65536 + lo (lo unsigned 16 bit), both chains fit int32
template <int N, bool TRANSPOSED> __device__ __forceinline__ void mat64(const int (&in)[N], long (&out)[N])
{
    int hi[N], lo[N];
#pragma unroll
    for(int i = 0; i < N; i++) hi[i] = in[i] >> 16, lo[i] = in[i] & 0xffff;
#pragma unroll
    for(int o = 0; o < N; o++) {
        int ah = 0, al = 0;
#pragma unroll
        for(int i = 0; i < N; i++) {
            const int m = TRANSPOSED ? tm32_at<N>(i, o) : tm32_at<N>(o, i);
            ah += m * hi[i], al += m * lo[i];
        }
        out[o] = (long)ah * 65536L + (long)al;
    }
}

template <int N>
__global__ __launch_bounds__(256) void k_rdo_rows(const pel *__restrict__ org, int s_org, const pel *__restrict__ pred, int s_pred,
                                                  const xeve_hip_job *__restrict__ jobs, int njobs, RdoParams P,
                                                  int16_t *__restrict__ coef, pel *__restrict__ rec, int s_rec,
                                                  int32_t *__restrict__ nnz_out, int64_t *__restrict__ ssd_out)
{
    constexpr int BPW = 64 / N, PITCH = N + 1;
    __shared__ int tile[4][BPW * N * PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bl = lane / N, y = lane % N;
    int *T = tile[wave] + bl * N * PITCH;
    const int  j    = (xh_xcd_block(blockIdx.x, gridDim.x) * 4 + wave) * BPW + bl;
    const bool live = j < njobs;
    const XhJob jb = live ? xh_job(jobs[j]) : XhJob{0, 0};
    // 1. my row of the original and of the prediction; residual; SSD(org, pred)
    int o[N], p[N], v[N];
    unsigned long long ssd_p = 0, ssd_r = 0;
    if(live) { // one row = N pels: 8-byte (N = 4) or 16-byte vector loads at any 2-byte alignment
        const pel *po = org + xh_u(jb.off1) + (long)y * s_org, *pp = pred + jb.off2 + (long)y * s_pred;
        if(N == 4) {
            const u32x2 a = xh_ld4(po), b = xh_ld4(pp);
#pragma unroll
            for(int k = 0; k < 2; k++) o[2 * k] = xh_lo16(a[k]), o[2 * k + 1] = xh_hi16(a[k]), p[2 * k] = xh_lo16(b[k]), p[2 * k + 1] = xh_hi16(b[k]);
        }
        else {
#pragma unroll
            for(int q = 0; q < N / 8; q++) {
                const u32x4 a = xh_ld8(po + 8 * q), b = xh_ld8(pp + 8 * q);
#pragma unroll
                for(int k = 0; k < 4; k++) {
                    o[8 * q + 2 * k] = xh_lo16(a[k]), o[8 * q + 2 * k + 1] = xh_hi16(a[k]);
                    p[8 * q + 2 * k] = xh_lo16(b[k]), p[8 * q + 2 * k + 1] = xh_hi16(b[k]);
                }
            }
        }
    }
    else {
#pragma unroll
        for(int x = 0; x < N; x++) o[x] = p[x] = 0;
    }
#pragma unroll
    for(int x = 0; x < N; x++) {
        v[x] = o[x] - p[x];
        ssd_p += (unsigned)((v[x] * v[x]) >> P.ssd_shift);
    }
    int  t[N], c[N];
    long c64[N];
    int  cnt = 0;
    if(P.stage != 2) {
    // 2. forward: rows (lane = y), transpose, columns (lane = kx)
    mat32<N, false>(v, t); // t[kx] = sum_x Mw[kx][x] d[y][x]
#pragma unroll
    for(int k = 0; k < N; k++) T[k * PITCH + y] = t[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int k = 0; k < N; k++) t[k] = T[y * PITCH + k]; // lane kx (= y) now holds T[kx][0..N-1]
    __builtin_amdgcn_wave_barrier();
    mat64<N, false>(t, c64); // c[ky] = sum_yy Mh[ky][yy] T[kx][yy]
    bool hit = P.z_thr < 0;
    const long addf = 1L << (P.shift_fwd - 1);
#pragma unroll
    for(int k = 0; k < N; k++) {
        c[k] = (int)(int16_t)((c64[k] + addf) >> P.shift_fwd);
        hit |= ((long)(c[k] < 0 ? -c[k] : c[k]) * P.z_scale) >= P.z_thr;
    }
    // 3. zero pre-test over the block (N lanes), quant, levels out (lane kx holds column kx), dequant
    const unsigned long long blk_mask = (N == 64 ? ~0ull : ((1ull << N) - 1)) << (bl * N);
    hit = (__ballot(hit) & blk_mask) != 0;
    if(P.stage == 1) { // front half: raw coefficients out (lane kx holds column kx), SSD(pred), done
#pragma unroll
        for(int k = 0; k < N; k++)
            if(live) coef[(size_t)j * N * N + k * N + y] = (int16_t)c[k];
        for(int m = 1; m < N; m <<= 1)
            ssd_p += ((unsigned long long)__shfl_xor((unsigned)(ssd_p >> 32), m, 64) << 32) | __shfl_xor((unsigned)ssd_p, m, 64);
        if(live && y == 0) ssd_out[2 * j] = (int64_t)ssd_p;
        return;
    }
#pragma unroll
    for(int k = 0; k < N; k++) {
        int lev = 0;
        if(hit) {
            const int neg = c[k] < 0;
            lev = (int)(int16_t)((((neg ? -c[k] : c[k]) * P.q_scale) + P.q_offset) >> P.q_shift);
            lev = (int)(int16_t)(neg ? -lev : lev);
        }
        cnt += lev != 0;
        if(live) coef[(size_t)j * N * N + k * N + y] = (int16_t)lev;
        c[k] = lev;
    }
    cnt = xh_group_sum<N>(cnt);
    }
    else { // back half: the levels are already in `coef`
#pragma unroll
        for(int k = 0; k < N; k++) c[k] = live ? coef[(size_t)j * N * N + k * N + y] : 0;
    }
#pragma unroll
    for(int k = 0; k < N; k++) {
        long dq = ((long)c[k] * P.dq_scale + P.dq_offset) >> P.dq_shift;
        c[k]    = (int)(dq < -32768 ? -32768 : (dq > 32767 ? 32767 : dq));
    }
    // 4. inverse: columns (lane = kx): t[yy] = sum_ky Mh[ky][yy] C[ky][kx]; transpose; rows (lane = y)
    mat32<N, true>(c, t);
#pragma unroll
    for(int k = 0; k < N; k++) T[k * PITCH + y] = t[k]; // T[yy][kx]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int k = 0; k < N; k++) t[k] = T[y * PITCH + k]; // lane y holds T'[y][kx = 0..N-1]
    mat64<N, true>(t, c64); // r[x] = sum_kx Mw[kx][x] T'[y][kx]
    const long addi = 1L << (P.shift_inv - 1);
#pragma unroll
    for(int x = 0; x < N; x++) {
        long r = (c64[x] + addi) >> P.shift_inv;
        r      = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
        int q  = (int)(int16_t)((int)r + p[x]);
        q      = q < 0 ? 0 : (q > P.maxv ? P.maxv : q);
        v[x]   = q;
        const int e = o[x] - q;
        ssd_r += (unsigned)((e * e) >> P.ssd_shift);
    }
    if(live) {
        pel *pr = s_rec > 0 ? rec + jb.off1 + (long)y * s_rec : rec + (long)j * (N * N) - (long)y * s_rec; // (s_rec < 0: dense blocks)
        if(N == 4) {
            u32x2 w2;
            w2[0] = xh_pack16(v[0], v[1]), w2[1] = xh_pack16(v[2], v[3]);
            xh_st4(pr, w2);
        }
        else {
#pragma unroll
            for(int q = 0; q < N / 8; q++) {
                u32x4 w4;
#pragma unroll
                for(int k = 0; k < 4; k++) w4[k] = xh_pack16(v[8 * q + 2 * k], v[8 * q + 2 * k + 1]);
                xh_st8(pr + 8 * q, w4);
            }
        }
    }
    // 5. per-block sums over the N lanes
    for(int m = 1; m < N; m <<= 1) {
        ssd_p += ((unsigned long long)__shfl_xor((unsigned)(ssd_p >> 32), m, 64) << 32) | __shfl_xor((unsigned)ssd_p, m, 64);
        ssd_r += ((unsigned long long)__shfl_xor((unsigned)(ssd_r >> 32), m, 64) << 32) | __shfl_xor((unsigned)ssd_r, m, 64);
    }
    if(live && y == 0) {
        if(P.stage == 0) nnz_out[j] = cnt, ssd_out[2 * j] = (int64_t)ssd_p;
        ssd_out[2 * j + 1] = (int64_t)ssd_r;
    }
}

static int residual_launch(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h,
                           int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, int zero_test, int stage, int16_t *coef,
                           pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(org && pred && jobs && coef && rec && nnz && ssd && njobs >= 0);
    XH_REQUIRE(log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6 && bit_depth >= 8 && bit_depth <= 14);
    XH_REQUIRE(qp >= 0 && qp <= 87 && qscale > 0 && qscale < 65536 && dqscale > 0);
    if(njobs == 0) return XEVE_HIP_OK;
    const int odd = (log2w + log2h) & 1, log2_size = (log2w + log2h) >> 1;
    RdoParams P;
    P.stage     = stage;
    P.shift_fwd = (log2w - 1 + bit_depth - 8) + (log2h + 6); // xeve_util.c:34-35
    P.shift_inv = 7 + (12 - (bit_depth - 8));                // xeve_itdq.h:38-39
    P.q_scale   = qscale;
    P.q_shift   = 14 + (15 - bit_depth - log2_size) + qp / 6; // xeve_tq.c:716-717
    XH_REQUIRE(P.q_shift >= 9 && P.q_shift <= 30);
    P.q_offset = (is_intra_slice ? 171 : 85) << (P.q_shift - 9);
    if(zero_test) { // xeve_tq.c:673-683
        const int zs = 14 + (15 - bit_depth - log2_size + (odd ? 7 : 0)) + qp / 6;
        P.z_scale    = (long)qscale * (odd ? 181 : 1);
        P.z_thr      = (1L << zs) - ((long)(is_intra_slice ? 201 : 153) << (zs - 9));
    }
    else P.z_scale = 0, P.z_thr = -1;
    P.dq_scale  = (long)dqscale * (odd ? 181 : 1);            // xeve_itdq.c:442-475
    P.dq_shift  = (uint8_t)(20 - 14 - (15 - bit_depth - log2_size) + (odd ? 8 : 0));
    P.dq_offset = P.dq_shift == 0 ? 0 : 1 << (P.dq_shift - 1);
    P.ssd_shift = (bit_depth - 8) * 2;
    P.maxv      = (1 << bit_depth) - 1;
    XhProf prof(XH_PROF_RESID, st);
    if(g_use_mfma && log2w == log2h && log2w >= 5)
        return xh_rdo_mfma(1 << log2w, org, s_org, pred, s_pred, jobs, njobs, &P, coef, rec, s_rec, nnz, ssd, st);
    if(g_use_rows && log2w == log2h && log2w >= 2 && log2w <= 4) { // 4x4, 8x8, 16x16: row-per-lane register form
        const int bpw = 4 * (64 >> log2w);
        const dim3 grid((njobs + bpw - 1) / bpw);
        if(log2w == 2) k_rdo_rows<4><<<grid, 256, 0, st>>>(org, s_org, pred, s_pred, jobs, njobs, P, coef, rec, s_rec, nnz, ssd);
        else if(log2w == 3) k_rdo_rows<8><<<grid, 256, 0, st>>>(org, s_org, pred, s_pred, jobs, njobs, P, coef, rec, s_rec, nnz, ssd);
        else k_rdo_rows<16><<<grid, 256, 0, st>>>(org, s_org, pred, s_pred, jobs, njobs, P, coef, rec, s_rec, nnz, ssd);
        XH_HIP(hipGetLastError());
        return XEVE_HIP_OK;
    }
    const int n = 1 << (log2w + log2h), tpb = n < 256 ? n : 256, bpw = 256 / tpb;
    const size_t lds = (size_t)bpw * 8 * n + (size_t)bpw * 24 + 4 * ((size_t)(1 << (2 * log2w)) + (size_t)(1 << (2 * log2h)));
    k_rdo_valu<<<(njobs + bpw - 1) / bpw, 256, lds, st>>>(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, P, coef, rec, s_rec, nnz, ssd);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_residual_rdo(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w,
                                     int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, int zero_test,
                                     int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, void *stream)
{
    return residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, zero_test, 0, coef,
                           rec, s_rec, nnz, ssd, (hipStream_t)stream);
}

// the back half alone: dequantisation, inverse transform and reconstruction of the levels in `coef` (xeve_itdq + xeve_recon of a decided CU, xeve_pinter.c:2004-2032), one
// launch per component instead of xeve_hip_dquant + xeve_hip_itrans + xeve_hip_recon; ssd: scratch of 2 * njobs entries (the kernel's by-product)
int xh_residual_back(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h, int bit_depth, int qp, int dqscale,
                     const int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st)
{
    return residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, 16384, dqscale, 0, 0, 2, const_cast<int16_t *>(coef), rec, s_rec, nnz, ssd, st);
}

// The chain as the presets actually run it (rdoq = 1, xeve_enc.c:2452,2469): front half (DIFF, SSD, DCT), then
// xeve_quant_nnz's zero pre-test + xeve_rdoq_run_length_cc (rdoq.hip), then the back half (dequant, IDCT, recon, SSD).
extern "C" int xeve_hip_residual_rdoq(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w,
                                      int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, double lambda, int is_luma,
                                      int tool_iqt, const xeve_hip_rdoq_est *est, int16_t *coef, pel *rec, int s_rec, int32_t *nnz,
                                      int64_t *ssd, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, 0, 1, coef, rec,
                             s_rec, nnz, ssd, st);
    if(rc != XEVE_HIP_OK) return rc;
    rc = xeve_hip_rdoq_zt(coef, njobs, log2w, log2h, qp, lambda, is_luma, bit_depth, tool_iqt, est, 1, is_intra_slice, nnz, st);
    if(rc != XEVE_HIP_OK) return rc;
    return residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, 0, 2, coef, rec,
                           s_rec, nnz, ssd, st);
}

// ... and with the estimates the reference derives from the CU's entry coder state, one record per block, all on the device
// (xeve_hip_rdoq_bit_est -> xeve_hip_rdoq_dev); ch_type 0 Y / 1 U / 2 V selects the context set and the cbf pair of an inter CU
extern "C" int xeve_hip_rdoq_dev(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int ch_type, int bit_depth, int tool_iqt,
                                 const xeve_hip_rdoq_est_full *est, const int32_t *est_idx, int zero_test, int is_intra_slice, int is_intra_cu,
                                 int32_t *nnz, void *stream);
// the fused chain with the estimates picked per block on the device; is_intra_cu selects the cbf pair an intra CU's luma block is priced with (xeve_tq.c:565-583);
// s_rec < 0: reconstruction stored as dense blocks (block j at j * n, pitch -s_rec) instead of in a plane laid out like the original
int xh_residual_rdoq(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h, int bit_depth, int qp,
                     int qscale, int dqscale, int is_intra_slice, int is_intra_cu, double lambda, int ch_type, int tool_iqt, const xeve_hip_rdoq_est_full *est,
                     const int32_t *est_idx, int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st)
{
    int rc = residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, 0, 1, coef, rec,
                             s_rec, nnz, ssd, st);
    if(rc != XEVE_HIP_OK) return rc;
    rc = xeve_hip_rdoq_dev(coef, njobs, log2w, log2h, qp, lambda, ch_type, bit_depth, tool_iqt, est, est_idx, 1, is_intra_slice, is_intra_cu, nnz, st);
    if(rc != XEVE_HIP_OK) return rc;
    return residual_launch(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, 0, 2, coef, rec,
                           s_rec, nnz, ssd, st);
}
extern "C" int xeve_hip_residual_rdoq_dev(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w,
                                          int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, double lambda, int ch_type,
                                          int tool_iqt, const xeve_hip_rdoq_est_full *est, const int32_t *est_idx, int16_t *coef, pel *rec, int s_rec,
                                          int32_t *nnz, int64_t *ssd, void *stream)
{
    return xh_residual_rdoq(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, bit_depth, qp, qscale, dqscale, is_intra_slice, 0, lambda, ch_type, tool_iqt, est,
                            est_idx, coef, rec, s_rec, nnz, ssd, (hipStream_t)stream);
}
