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
static bool g_use_mfma = true;

// All six matrices, row-major [k][x], concatenated; offset of size 2^l is XH_TM_OFF(l).
__device__ __constant__ int8_t c_tm[4 + 16 + 64 + 256 + 1024 + 4096];
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
    const char *e = getenv("XEVE_HIP_DCT"); // developer switch: "valu" forces the LDS/VALU path for 32/64 too
    g_use_mfma    = !(e && strcmp(e, "valu") == 0);
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
                int64_t v = step == 0 ? (int64_t)((const int16_t *)src)[j * n + x] : (int64_t)((const int32_t *)src)[j * n + x];
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
            int64_t v = step == 0 ? (int64_t)((const int16_t *)src)[k * line + j] : (int64_t)((const int32_t *)src)[k * line + j];
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
    XH_REQUIRE(qp >= 0 && qp <= 63 && scale > 0 && scale < 65536);
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
    XH_REQUIRE(coded && qp >= 0 && qp <= 63 && scale > 0 && scale < 65536);
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
struct RdoParams { // must match dct_mfma.hip
    int shift_fwd, shift_inv;
    int q_scale, q_shift, q_offset;
    long z_scale, z_thr;
    long dq_scale; int dq_shift, dq_offset;
    int ssd_shift, maxv;
};
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
    const xeve_hip_job jb = live ? jobs[j] : xeve_hip_job{0, 0};
    if(t == 0) acc64[0] = acc64[1] = 0, acc32[0] = acc32[1] = 0;
    __syncthreads();
    // 1. residual + SSD(org, pred)
    if(live) {
        unsigned long long s = 0;
        for(int i = t; i < n; i += tpb) {
            const int y = i >> log2w, x = i & (w - 1);
            const int d = (int)org[jb.off1 + y * s_org + x] - (int)pred[jb.off2 + y * s_pred + x];
            X[i] = d;
            s += (unsigned)((d * d) >> P.ssd_shift);
        }
        atomicAdd(&acc64[0], s);
    }
    __syncthreads();
    // 2. forward pass 1: Tm[kx][y] = sum_x Mw[kx][x] X[y][x]
    if(live)
        for(int i = t; i < n; i += tpb) {
            const int kx = i >> log2h, y = i & (h - 1);
            int a = 0;
            if(!(w == 64 && kx >= 32))
                for(int x = 0; x < w; x++) a += mw[kx * w + x] * X[y * w + x];
            Tm[i] = a;
        }
    __syncthreads();
    // 3. forward pass 2 + zero pre-test: X[ky][kx] = coefficient
    if(live) {
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
            hit |= ((int64_t)(c < 0 ? -c : c) * P.z_scale) >= P.z_thr;
        }
        if(P.z_thr < 0) hit = 1;
        if(hit) atomicOr(&acc32[0], 1);
    }
    __syncthreads();
    // 4. quant, levels out, dequant
    if(live) {
        const bool hit = acc32[0] != 0;
        int cnt = 0;
        for(int i = t; i < n; i += tpb) {
            int lev = 0;
            if(hit) {
                const int c = X[i], neg = c < 0;
                lev = (int)(int16_t)((((neg ? -c : c) * P.q_scale) + P.q_offset) >> P.q_shift);
                lev = (int)(int16_t)(neg ? -lev : lev);
            }
            cnt += lev != 0;
            coef[(size_t)j * n + i] = (int16_t)lev;
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
            rec[jb.off1 + y * s_rec + x] = (pel)r;
            const int e = (int)org[jb.off1 + y * s_org + x] - r;
            s += (unsigned)((e * e) >> P.ssd_shift);
        }
        atomicAdd(&acc64[1], s);
    }
    __syncthreads();
    if(live && t == 0) {
        nnz_out[j]         = acc32[1];
        ssd_out[2 * j]     = (int64_t)acc64[0];
        ssd_out[2 * j + 1] = (int64_t)acc64[1];
    }
}

extern "C" int xeve_hip_residual_rdo(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w,
                                     int log2h, int bit_depth, int qp, int qscale, int dqscale, int is_intra_slice, int zero_test,
                                     int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(org && pred && jobs && coef && rec && nnz && ssd && njobs >= 0);
    XH_REQUIRE(log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6 && bit_depth >= 8 && bit_depth <= 14);
    XH_REQUIRE(qp >= 0 && qp <= 63 && qscale > 0 && qscale < 65536 && dqscale > 0);
    if(njobs == 0) return XEVE_HIP_OK;
    const int odd = (log2w + log2h) & 1, log2_size = (log2w + log2h) >> 1;
    RdoParams P;
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
    hipStream_t st = (hipStream_t)stream;
    if(g_use_mfma && log2w == log2h && log2w >= 5)
        return xh_rdo_mfma(1 << log2w, org, s_org, pred, s_pred, jobs, njobs, &P, coef, rec, s_rec, nnz, ssd, st);
    const int n = 1 << (log2w + log2h), tpb = n < 256 ? n : 256, bpw = 256 / tpb;
    const size_t lds = (size_t)bpw * 8 * n + (size_t)bpw * 24 + 4 * ((size_t)(1 << (2 * log2w)) + (size_t)(1 << (2 * log2h)));
    k_rdo_valu<<<(njobs + bpw - 1) / bpw, 256, lds, st>>>(org, s_org, pred, s_pred, jobs, njobs, log2w, log2h, P, coef, rec, s_rec, nnz, ssd);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
