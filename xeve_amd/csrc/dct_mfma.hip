// xeve_amd/csrc/dct_mfma.hip -- 32x32 and 64x64 integer DCT / IDCT on the gfx950 matrix cores.
//
// reference semantics: xeve_trans (src_base/xeve_tq.c:396-404) / xeve_itrans (src_base/xeve_itdq.c:435-440)
// over tx_pb32b/tx_pb64b and xeve_itx_pb32b/xeve_itx_pb64b.  Both are an exact integer double product with ONE
// rounding shift at the end (pass 1 uses shift 0), so any exact evaluation order is bit-identical:
//     forward  C = (Mh * X * Mw^T + r) >> s          (64-point: only the 32x32 low-frequency corner is non-zero)
//     inverse  X = clip16((Mh^T * C * Mw + r) >> s)
//
// Exactness on v_mfma_i32_32x32x32_i8 (signed 8-bit x signed 8-bit -> i32):
//   * the DCT matrices are s8 already (|m| <= 90);
//   * 16-bit data is split into its two BYTES: x = 256*hi + lo + 128 with hi = (s8)(x >> 8) and
//     lo = (s8)((x & 0xff) ^ 0x80); the "+128" turns into a per-output constant 128 * (row/column sum of M);
//   * the 32-bit intermediate is split the same way into four bytes (three of them offset by 128):
//     t = b3*2^24 + (b2' + 128)*2^16 + (b1' + 128)*2^8 + (b0' + 128), constant 128*(1+2^8+2^16) * sum(M).
//   Every partial product sum fits i32 (|sum| <= 64*128*90), the partials are recombined in 64-bit VALU.
//
// Fragment layout used (CDNA4: 16 consecutive k per lane for A and B; C/D col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  The D tile of pass 1 is fed to pass 2 as the B operand WITHOUT any
// transposition: the contraction index of pass 2 is enumerated in exactly the order the D registers hold it,
// f(half, reg) = (reg&3) + 8*(reg>>2) + 4*half, and the A operand (rows of the DCT matrix) is gathered with the
// same f -- the hardware only requires that A and B enumerate k identically.
//
// One wave per block, four blocks per workgroup; HBM traffic = 2*N*N bytes in + 2*N*N out per block.
#include <cmath>
#include <cstring>
#include "xh_common.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// row-major [k][x] matrices and their transposes [x][k] for N = 32 and N = 64, plus row / column sums
struct DctTabs {
    int8_t  m32[32 * 32], t32[32 * 32], m64[64 * 64], t64[64 * 64];
    int32_t rs32[32], cs32[32], rs64[64], cs64[64], cs64lo[64];
};
__device__ DctTabs g_dct;

int xh_dct_mfma_init(const int8_t *m32, const int8_t *m64)
{
    static DctTabs h;
    memcpy(h.m32, m32, sizeof(h.m32));
    memcpy(h.m64, m64, sizeof(h.m64));
    for(int k = 0; k < 32; k++) for(int x = 0; x < 32; x++) h.t32[x * 32 + k] = m32[k * 32 + x];
    for(int k = 0; k < 64; k++) for(int x = 0; x < 64; x++) h.t64[x * 64 + k] = m64[k * 64 + x];
    for(int i = 0; i < 32; i++) {
        h.rs32[i] = h.cs32[i] = 0;
        for(int j = 0; j < 32; j++) h.rs32[i] += m32[i * 32 + j], h.cs32[i] += m32[j * 32 + i];
    }
    for(int i = 0; i < 64; i++) {
        h.rs64[i] = h.cs64[i] = 0;
        for(int j = 0; j < 64; j++) h.rs64[i] += m64[i * 64 + j], h.cs64[i] += m64[j * 64 + i];
        h.cs64lo[i] = 0;
        for(int j = 0; j < 32; j++) h.cs64lo[i] += m64[j * 64 + i];
    }
    XH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dct), &h, sizeof(h)));
    return XEVE_HIP_OK;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int frow(int half, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// 16 int16 (8 dwords, two pels per dword) -> byte planes: hi = high bytes, lo = low bytes ^ 0x80
__device__ __forceinline__ void split16(const u32x4 &a, const u32x4 &b, v4i &lo, v4i &hi)
{
    const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for(int q = 0; q < 4; q++) {
        lo[q] = (int)(__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x06040200u) ^ 0x80808080u);
        hi[q] = (int)__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07050301u);
    }
}

// 16 int32 (one D tile) -> four byte planes; planes 0..2 offset by 0x80, plane 3 (top byte) signed as is
__device__ __forceinline__ void split32(const v16i &t, v4i (&b)[4])
{
#pragma unroll
    for(int q = 0; q < 4; q++) {
        const uint32_t t0 = t[4 * q], t1 = t[4 * q + 1], t2 = t[4 * q + 2], t3 = t[4 * q + 3];
#pragma unroll
        for(int l = 0; l < 4; l++) {
            const uint32_t s01 = 0x0c0c0000u | ((4u + l) << 8) | (uint32_t)l;           // bytes: t0.l, t1.l, 0, 0
            const uint32_t p01 = __builtin_amdgcn_perm(t1, t0, s01);
            const uint32_t p23 = __builtin_amdgcn_perm(t3, t2, s01);
            uint32_t v = __builtin_amdgcn_perm(p23, p01, 0x05040100u);                   // p01.b0, p01.b1, p23.b0, p23.b1
            if(l < 3) v ^= 0x80808080u;
            b[l][q] = (int)v;
        }
    }
}

__device__ __forceinline__ v16i shl8(v16i v)
{
#pragma unroll
    for(int i = 0; i < 16; i++) v[i] = (int)((uint32_t)v[i] << 8);
    return v;
}

// A operand of pass 2: 16 bytes of matrix row `row`, columns base + f(kg, 0..15)
__device__ __forceinline__ v4i load_a2(const int8_t *mat, int n, int row, int base, int kg)
{
    const int32_t *p = reinterpret_cast<const int32_t *>(mat + row * n + base);
    v4i a;
#pragma unroll
    for(int q = 0; q < 4; q++) a[q] = p[2 * q + kg];
    return a;
}
__device__ __forceinline__ v4i load16(const int8_t *p) { return *reinterpret_cast<const v4i *>(p); }

constexpr long K3 = 128L * (1 + 256 + 65536);

// ------------------------------------------------------------------------------------------------------
// N = 32 or 64, square.  FWD: in = residual, out = coefficients.  !FWD: in = coefficients, out = residual.
// ------------------------------------------------------------------------------------------------------
template <int N, bool FWD>
__global__ __launch_bounds__(256) void k_dct_mfma(int16_t *__restrict__ coef, int nblk, int shift)
{
    constexpr int NT = N / 32;            // 32-wide tiles per dimension
    constexpr int OT = (FWD && N == 64) ? 1 : NT; // output tiles per dimension (64-point forward keeps 32x32)
    const int lane = threadIdx.x & 63, l32 = lane & 31, kg = lane >> 5;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(b >= nblk) return;
    int16_t *blk = coef + (size_t)b * N * N;
    const int8_t  *M  = N == 32 ? g_dct.m32 : g_dct.m64;   // [k][x]
    const int8_t  *MT = N == 32 ? g_dct.t32 : g_dct.t64;   // [x][k]
    const int32_t *RS = N == 32 ? g_dct.rs32 : g_dct.rs64, *CS = N == 32 ? g_dct.cs32 : g_dct.cs64;
    const long add = 1L << (shift - 1);

    // ---- A operands of pass 1: the data block, byte-split.  a1[mt][ks]: rows 32*mt + l32, cols 32*ks + 16*kg .. +15
    v4i alo[NT][NT], ahi[NT][NT];
#pragma unroll
    for(int mt = 0; mt < NT; mt++)
#pragma unroll
        for(int ks = 0; ks < NT; ks++) {
            const int16_t *p = blk + (32 * mt + l32) * N + 32 * ks + 16 * kg;
            split16(*reinterpret_cast<const u32x4 *>(p), *reinterpret_cast<const u32x4 *>(p + 8), alo[mt][ks], ahi[mt][ks]);
        }

#pragma unroll
    for(int nt = 0; nt < OT; nt++) { // output column tile (kx for FWD, x for inverse)
        // ---- pass 1: FWD  T[y][kx] = sum_x X[y][x] Mw[kx][x]   (B = rows of M)
        //              INV  U[ky][x] = sum_kx C[ky][kx] Mw[kx][x] (B = rows of M^T)
        v4i b1[NT];
#pragma unroll
        for(int ks = 0; ks < NT; ks++) b1[ks] = load16((FWD ? M : MT) + (32 * nt + l32) * N + 32 * ks + 16 * kg);
        const int c1 = 128 * (FWD ? RS : CS)[32 * nt + l32]; // same for every row of this lane's column
        v4i tb[NT][4];                                       // pass-1 result, byte planes, per row tile
#pragma unroll
        for(int mt = 0; mt < NT; mt++) {
            v16i dl = {0}, dh = {0};
#pragma unroll
            for(int ks = 0; ks < NT; ks++) {
                dl = MFMA(alo[mt][ks], b1[ks], dl);
                dh = MFMA(ahi[mt][ks], b1[ks], dh);
            }
            v16i t;
#pragma unroll
            for(int i = 0; i < 16; i++) t[i] = (int)((uint32_t)dh[i] << 8) + dl[i] + c1;
            split32(t, tb[mt]);
        }
        // ---- pass 2: FWD  C[ky][kx] = sum_y Mh[ky][y] T[y][kx]   (A = rows of M, gathered with f)
        //              INV  X[y][x]  = sum_ky Mh[ky][y] U[ky][x]   (A = rows of M^T)
#pragma unroll
        for(int ot = 0; ot < OT; ot++) { // output row tile (ky for FWD, y for inverse)
            v4i a2[NT];
#pragma unroll
            for(int ks = 0; ks < NT; ks++) a2[ks] = load_a2(FWD ? M : MT, N, 32 * ot + l32, 32 * ks, kg);
            v16i hacc = {0}, lacc = {0};
#pragma unroll
            for(int ks = 0; ks < NT; ks++) hacc = MFMA(a2[ks], tb[ks][3], hacc);
            hacc = shl8(hacc);
#pragma unroll
            for(int ks = 0; ks < NT; ks++) hacc = MFMA(a2[ks], tb[ks][2], hacc);
#pragma unroll
            for(int ks = 0; ks < NT; ks++) lacc = MFMA(a2[ks], tb[ks][1], lacc);
            lacc = shl8(lacc);
#pragma unroll
            for(int ks = 0; ks < NT; ks++) lacc = MFMA(a2[ks], tb[ks][0], lacc);
#pragma unroll
            for(int r = 0; r < 16; r++) {
                const int row = 32 * ot + frow(kg, r);
                long v = (long)hacc[r] * 65536L + (long)lacc[r] + K3 * (long)(FWD ? RS : CS)[row];
                v = (v + add) >> shift;
                if(!FWD) v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
                blk[row * N + 32 * nt + l32] = (int16_t)v;
            }
        }
    }
    if(FWD && N == 64) { // tx_pb64b forces outputs k >= 32 to zero in both dimensions (xeve_tq.c:321-381)
        const u32x4 z = {0, 0, 0, 0};
        for(int i = lane; i < 64 * 8; i += 64) { // 16-byte chunks; row = i / 8, chunk = i % 8 (8 pels each)
            const int row = i >> 3, ch = i & 7;
            if(row >= 32 || ch >= 4) *reinterpret_cast<u32x4 *>(blk + row * 64 + ch * 8) = z;
        }
    }
}

int xh_dct_mfma(bool fwd, int16_t *coef, int nblk, int n, int shift, hipStream_t st)
{
    const dim3 grid((nblk + 3) / 4);
    if(n == 32) {
        if(fwd) k_dct_mfma<32, true><<<grid, 256, 0, st>>>(coef, nblk, shift);
        else k_dct_mfma<32, false><<<grid, 256, 0, st>>>(coef, nblk, shift);
    }
    else {
        if(fwd) k_dct_mfma<64, true><<<grid, 256, 0, st>>>(coef, nblk, shift);
        else k_dct_mfma<64, false><<<grid, 256, 0, st>>>(coef, nblk, shift);
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// ======================================================================================================
// Fused residual chain for 32x32 / 64x64 blocks (xeve_hip_residual_rdo): one wave per block.
//   DIFF + SSD(pred) -> DCT (MFMA) -> zero pre-test + quant -> levels out -> dequant -> [LDS transpose of the 32x32
//   coefficient corner] -> IDCT (MFMA, K = 32 only: a 64-point forward transform leaves nothing outside the corner)
//   -> [LDS transpose back to row layout] -> recon + SSD(rec).
// ======================================================================================================

__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return xh_pack16(xh_lo16(a) - xh_lo16(b), xh_hi16(a) - xh_hi16(b));
}
__device__ __forceinline__ long wave_sum64(long v)
{
#pragma unroll
    for(int m = 1; m < 64; m <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)((unsigned long)v >> 32), m, 64);
        v += (long)(((unsigned long)hi << 32) | lo);
    }
    return v;
}

template <int N>
__global__ __launch_bounds__(256) void k_rdo_mfma(const pel *__restrict__ org, int s_org, const pel *__restrict__ pred, int s_pred,
                                                  const xeve_hip_job *__restrict__ jobs, int njobs, RdoParams P,
                                                  int16_t *__restrict__ coef, pel *__restrict__ rec, int s_rec,
                                                  int32_t *__restrict__ nnz_out, int64_t *__restrict__ ssd_out)
{
    constexpr int NT = N / 32;
    __shared__ __attribute__((aligned(16))) int16_t tile[4][32 * 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l32 = lane & 31, kg = lane >> 5;
    const int j = blockIdx.x * 4 + wave;
    if(j >= njobs) return;
    const XhJob jb = xh_job(jobs[j]);
    int16_t *T = tile[wave];
    const int8_t  *M  = N == 32 ? g_dct.m32 : g_dct.m64;
    const int8_t  *MT = N == 32 ? g_dct.t32 : g_dct.t64;
    const int32_t *RS = N == 32 ? g_dct.rs32 : g_dct.rs64, *CSP = N == 32 ? g_dct.cs32 : g_dct.cs64lo;

    // ---- a/b: rows of org and pred in A layout, residual, SSD(org, pred)
    u32x4 o[NT][NT][2], p[NT][NT][2];
    v4i   alo[NT][NT], ahi[NT][NT];
    long  ssd_pred = 0;
#pragma unroll
    for(int mt = 0; mt < NT; mt++)
#pragma unroll
        for(int ks = 0; ks < NT; ks++) {
            const int y = 32 * mt + l32, x = 32 * ks + 16 * kg;
            const pel *po = org + xh_u(jb.off1) + (long)y * s_org + x, *pp = pred + jb.off2 + (long)y * s_pred + x;
            u32x4 d[2];
#pragma unroll
            for(int hq = 0; hq < 2; hq++) {
                o[mt][ks][hq] = xh_ld8(po + 8 * hq);
                p[mt][ks][hq] = xh_ld8(pp + 8 * hq);
#pragma unroll
                for(int k = 0; k < 4; k++) {
                    d[hq][k] = pk_sub(o[mt][ks][hq][k], p[mt][ks][hq][k]);
                    const int d0 = xh_lo16(d[hq][k]), d1 = xh_hi16(d[hq][k]);
                    ssd_pred += (uint32_t)((d0 * d0) >> P.ssd_shift) + (uint32_t)((d1 * d1) >> P.ssd_shift);
                }
            }
            split16(d[0], d[1], alo[mt][ks], ahi[mt][ks]);
        }

    int16_t *cb = coef + (size_t)j * N * N;
    int  cnt = 0;
    v4i  clo, chi;
    if(P.stage != 2) {
    // ---- c: forward DCT, low-frequency 32x32 corner (the whole spectrum for N = 32)
    v4i b1[NT];
#pragma unroll
    for(int ks = 0; ks < NT; ks++) b1[ks] = load16(M + l32 * N + 32 * ks + 16 * kg);
    const int c1 = 128 * RS[l32];
    v4i tb[NT][4];
#pragma unroll
    for(int mt = 0; mt < NT; mt++) {
        v16i dl = {0}, dh = {0};
#pragma unroll
        for(int ks = 0; ks < NT; ks++) {
            dl = MFMA(alo[mt][ks], b1[ks], dl);
            dh = MFMA(ahi[mt][ks], b1[ks], dh);
        }
        v16i t;
#pragma unroll
        for(int i = 0; i < 16; i++) t[i] = (int)((uint32_t)dh[i] << 8) + dl[i] + c1;
        split32(t, tb[mt]);
    }
    int cf[16]; // C[ky = frow(kg, r)][kx = l32]
    {
        v4i a2[NT];
#pragma unroll
        for(int ks = 0; ks < NT; ks++) a2[ks] = load_a2(M, N, l32, 32 * ks, kg);
        v16i hacc = {0}, lacc = {0};
#pragma unroll
        for(int ks = 0; ks < NT; ks++) hacc = MFMA(a2[ks], tb[ks][3], hacc);
        hacc = shl8(hacc);
#pragma unroll
        for(int ks = 0; ks < NT; ks++) hacc = MFMA(a2[ks], tb[ks][2], hacc);
#pragma unroll
        for(int ks = 0; ks < NT; ks++) lacc = MFMA(a2[ks], tb[ks][1], lacc);
        lacc = shl8(lacc);
#pragma unroll
        for(int ks = 0; ks < NT; ks++) lacc = MFMA(a2[ks], tb[ks][0], lacc);
        const long add = 1L << (P.shift_fwd - 1);
#pragma unroll
        for(int r = 0; r < 16; r++) {
            const long v = (long)hacc[r] * 65536L + (long)lacc[r] + K3 * (long)RS[frow(kg, r)];
            cf[r] = (int)(int16_t)((v + add) >> P.shift_fwd);
        }
    }
    // ---- d: zero pre-test, quant, levels out, dequant
    bool hit = true;
    if(P.z_thr >= 0) {
        bool h = false;
#pragma unroll
        for(int r = 0; r < 16; r++) h |= ((long)(cf[r] < 0 ? -cf[r] : cf[r]) * P.z_scale) >= P.z_thr;
        hit = __any(h);
    }
    if(P.stage == 1) { // front half: raw coefficients (and the zero rows / columns of a 64-point block) out
#pragma unroll
        for(int r = 0; r < 16; r++) cb[frow(kg, r) * N + l32] = (int16_t)cf[r];
        if(N == 64) {
            const u32x4 z = {0, 0, 0, 0};
            for(int i = lane; i < 64 * 8; i += 64) {
                const int row = i >> 3, ch = i & 7;
                if(row >= 32 || ch >= 4) *reinterpret_cast<u32x4 *>(cb + row * 64 + ch * 8) = z;
            }
        }
        ssd_pred = wave_sum64(ssd_pred);
        if(lane == 0) ssd_out[2 * j] = ssd_pred;
        return;
    }
#pragma unroll
    for(int r = 0; r < 16; r++) {
        int lev = 0;
        if(hit) {
            const int c = cf[r], neg = c < 0;
            lev = (int)(int16_t)((((neg ? -c : c) * P.q_scale) + P.q_offset) >> P.q_shift);
            lev = (int)(int16_t)(neg ? -lev : lev);
        }
        cnt += lev != 0;
        cb[frow(kg, r) * N + l32] = (int16_t)lev;
        long dq = ((long)lev * P.dq_scale + P.dq_offset) >> P.dq_shift;
        dq      = dq < -32768 ? -32768 : (dq > 32767 ? 32767 : dq);
        T[frow(kg, r) * 32 + l32] = (int16_t)dq; // e: transpose through LDS
    }
    if(N == 64) {
        const u32x4 z = {0, 0, 0, 0};
        for(int i = lane; i < 64 * 8; i += 64) {
            const int row = i >> 3, ch = i & 7;
            if(row >= 32 || ch >= 4) *reinterpret_cast<u32x4 *>(cb + row * 64 + ch * 8) = z;
        }
    }
    cnt = xh_group_sum<64>(cnt);
    __builtin_amdgcn_wave_barrier();
    // ---- f: inverse DCT of the 32x32 coefficient tile (K = 32)
    {
        const int16_t *row = T + l32 * 32 + 16 * kg;
        split16(*reinterpret_cast<const u32x4 *>(row), *reinterpret_cast<const u32x4 *>(row + 8), clo, chi);
    }
    __builtin_amdgcn_wave_barrier();
    }
    else { // back half: levels of row ky = l32, columns 16*kg .. +15 straight from memory (A layout), dequantised in place
        const int16_t *row = cb + l32 * N + 16 * kg;
        u32x4 q[2] = {*reinterpret_cast<const u32x4 *>(row), *reinterpret_cast<const u32x4 *>(row + 8)};
#pragma unroll
        for(int hq = 0; hq < 2; hq++)
#pragma unroll
            for(int k = 0; k < 4; k++) {
                long a = ((long)xh_lo16(q[hq][k]) * P.dq_scale + P.dq_offset) >> P.dq_shift, b = ((long)xh_hi16(q[hq][k]) * P.dq_scale + P.dq_offset) >> P.dq_shift;
                a = a < -32768 ? -32768 : (a > 32767 ? 32767 : a), b = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
                q[hq][k] = xh_pack16((int)a, (int)b);
            }
        split16(q[0], q[1], clo, chi);
    }
    long ssd_rec = 0;
#pragma unroll
    for(int nt = 0; nt < NT; nt++) { // x tile
        const v4i bi = load16(MT + (32 * nt + l32) * N + 16 * kg); // Mw[kx = 16kg + j][x]
        v16i dl = {0}, dh = {0};
        dl = MFMA(clo, bi, dl);
        dh = MFMA(chi, bi, dh);
        const int ci = 128 * CSP[32 * nt + l32];
        v16i u;
#pragma unroll
        for(int i = 0; i < 16; i++) u[i] = (int)((uint32_t)dh[i] << 8) + dl[i] + ci;
        v4i ub[4];
        split32(u, ub);
#pragma unroll
        for(int ot = 0; ot < NT; ot++) { // y tile
            const v4i a2 = load_a2(MT, N, 32 * ot + l32, 0, kg); // Mh[ky = f(kg, j)][y]
            v16i hacc = {0}, lacc = {0};
            hacc = MFMA(a2, ub[3], hacc);
            hacc = shl8(hacc);
            hacc = MFMA(a2, ub[2], hacc);
            lacc = MFMA(a2, ub[1], lacc);
            lacc = shl8(lacc);
            lacc = MFMA(a2, ub[0], lacc);
            const long add = 1L << (P.shift_inv - 1);
            // ---- g: residual tile (lane = x, regs = y) back to row layout through LDS, then recon + SSD(rec)
#pragma unroll
            for(int r = 0; r < 16; r++) {
                long v = (long)hacc[r] * 65536L + (long)lacc[r] + K3 * (long)CSP[32 * ot + frow(kg, r)];
                v      = (v + add) >> P.shift_inv;
                v      = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
                T[frow(kg, r) * 32 + l32] = (int16_t)v;
            }
            __builtin_amdgcn_wave_barrier();
            const int16_t *row = T + l32 * 32 + 16 * kg;
            const u32x4 r0 = *reinterpret_cast<const u32x4 *>(row), r1 = *reinterpret_cast<const u32x4 *>(row + 8);
            __builtin_amdgcn_wave_barrier();
            pel *pr = (s_rec > 0 ? rec + jb.off1 + (long)(32 * ot + l32) * s_rec : rec + (long)j * (N * N) - (long)(32 * ot + l32) * s_rec) + 32 * nt + 16 * kg; // (s_rec < 0: dense blocks)
#pragma unroll
            for(int hq = 0; hq < 2; hq++) {
                const u32x4 rr = hq ? r1 : r0;
                u32x4 out;
#pragma unroll
                for(int k = 0; k < 4; k++) {
                    const uint32_t pw = p[ot][nt][hq][k], ow = o[ot][nt][hq][k];
                    int t0 = (int)(int16_t)(xh_lo16(rr[k]) + xh_lo16(pw)), t1 = (int)(int16_t)(xh_hi16(rr[k]) + xh_hi16(pw));
                    t0 = t0 < 0 ? 0 : (t0 > P.maxv ? P.maxv : t0);
                    t1 = t1 < 0 ? 0 : (t1 > P.maxv ? P.maxv : t1);
                    out[k] = xh_pack16(t0, t1);
                    const int e0 = xh_lo16(ow) - t0, e1 = xh_hi16(ow) - t1;
                    ssd_rec += (uint32_t)((e0 * e0) >> P.ssd_shift) + (uint32_t)((e1 * e1) >> P.ssd_shift);
                }
                xh_st8(pr + 8 * hq, out);
            }
        }
    }
    ssd_pred = wave_sum64(ssd_pred);
    ssd_rec  = wave_sum64(ssd_rec);
    if(lane == 0) {
        if(P.stage == 0) nnz_out[j] = cnt, ssd_out[2 * j] = ssd_pred;
        ssd_out[2 * j + 1] = ssd_rec;
    }
}

int xh_rdo_mfma(int n, const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, const void *params,
                int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st)
{
    const RdoParams P = *static_cast<const RdoParams *>(params);
    const dim3 grid((njobs + 3) / 4);
    if(n == 32) k_rdo_mfma<32><<<grid, 256, 0, st>>>(org, s_org, pred, s_pred, jobs, njobs, P, coef, rec, s_rec, nnz, ssd);
    else k_rdo_mfma<64><<<grid, 256, 0, st>>>(org, s_org, pred, s_pred, jobs, njobs, P, coef, rec, s_rec, nnz, ssd);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
