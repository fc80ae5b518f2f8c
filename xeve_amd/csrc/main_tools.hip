// xeve_amd/csrc/main_tools.hip -- Main profile: the dispatch entries of xevem_platform_init_func (src_main/xevem_util.c:3917-3966) that are neither
// interpolation nor the 16-bit DCT-II passes: the inverse ATS passes (xeve_func_itrans: DCT-VIII / DST-VII, xevem_itdq.c:42-276) and the three kernels of
// the affine gradient search (Sobel derivatives of the prediction and the normal equations, xevem_mc.c:2341-2447).  Table-layer granularity: one call = one
// block / one CU; the batched affine analysis that would call them per CU in bulk is not built.
#include <cmath>
#include "xh_common.h"

// xevem_tbl_tr[DCT8 | DST7][log2 N - 2] (xevem_tbl.c:421-565), generated at init from the closed form
//   round(64 sqrt(N) sqrt(4 / (2N + 1)) { cos(pi (2k + 1)(2j + 1) / (4N + 2)) | sin(pi (2k + 1)(j + 1) / (2N + 1)) })
__device__ __constant__ int8_t c_ats[2][16 + 64 + 256 + 1024];
__host__ __device__ constexpr int ats_off(int log2n) { return ((1 << (2 * log2n)) - 16) / 3; } // 0, 16, 80, 336

int xh_main_tools_init()
{
    static int8_t t[2][16 + 64 + 256 + 1024];
    const double  pi = 3.14159265358979323846;
    for(int type = 0; type < 2; type++)
        for(int l = 2; l <= 5; l++) {
            const int    n  = 1 << l;
            const double sc = 64.0 * sqrt((double)n) * sqrt(4.0 / (2 * n + 1));
            for(int k = 0; k < n; k++)
                for(int j = 0; j < n; j++) {
                    const double v = sc * (type ? sin(pi * (2 * k + 1) * (j + 1) / (2 * n + 1)) : cos(pi * (2 * k + 1) * (2 * j + 1) / (4 * n + 2)));
                    t[type][ats_off(l) + k * n + j] = (int8_t)(v >= 0 ? (int)floor(v + 0.5) : -(int)floor(-v + 0.5));
                }
        }
    XH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_ats), t, sizeof(t)));
    return XEVE_HIP_OK;
}

// block[i * N + j] = clip16((sum_{k < cut} coef[k * line + i] * M[k][j] + rnd) >> shift) for i < line - skip_line, 0 below; one thread per output
__global__ void k_itrans_ats(const int16_t *__restrict__ coef, int16_t *__restrict__ block, int type, int log2n, int shift, int line, int skip_line, int cut)
{
    const int n = 1 << log2n, t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n * line) return;
    const int i = t >> log2n, j = t & (n - 1);
    int sum = 0;
    if(i < line - skip_line) {
        const int8_t *m = c_ats[type] + ats_off(log2n);
        for(int k = 0; k < cut; k++) sum += coef[k * line + i] * m[k * n + j];
        sum = (sum + (1 << (shift - 1))) >> shift;
        sum = sum < -32768 ? -32768 : (sum > 32767 ? 32767 : sum);
    }
    block[t] = (int16_t)sum;
}
int xh_itrans_ats(int type, int log2n, const int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(coef && block && (type == 0 || type == 1) && log2n >= 2 && log2n <= 5 && shift >= 1 && shift <= 24 && line >= 1 && line <= 64);
    XH_REQUIRE(skip_line >= 0 && skip_line <= line && skip_line_2 >= 0 && skip_line_2 <= (1 << log2n));
    const int n = 1 << log2n, total = n * line;
    k_itrans_ats<<<(total + 255) / 256, 256, 0, st>>>(coef, block, type, log2n, shift, line, skip_line, log2n == 2 ? 4 : n - skip_line_2); // (the 4-point forms use all four inputs)
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// the forward pass: coef[j * line + i] = (sum_k M[j][k] * block[i * N + k] + rnd) >> shift for i < line - skip_line, j < cut, 0 elsewhere (xeve_trans_DST7_B4 .. _DCT8_B32,
// xevem_tq.c:336-680: no clip, the sum is stored in 16 bits); one thread per output
__global__ void k_trans_ats(const int16_t *__restrict__ block, int16_t *__restrict__ coef, int type, int log2n, int shift, int line, int skip_line, int cut)
{
    const int n = 1 << log2n, t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n * line) return;
    const int j = t / line, i = t - j * line;
    int sum = 0;
    if(i < line - skip_line && j < cut) {
        const int8_t *m = c_ats[type] + ats_off(log2n) + j * n;
        for(int k = 0; k < n; k++) sum += m[k] * block[i * n + k];
        sum = (sum + (1 << (shift - 1))) >> shift;
    }
    coef[t] = (int16_t)sum;
}
int xh_trans_ats(int type, int log2n, const int16_t *block, int16_t *coef, int shift, int line, int skip_line, int skip_line_2, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(block && coef && (type == 0 || type == 1) && log2n >= 2 && log2n <= 5 && shift >= 1 && shift <= 24 && line >= 1 && line <= 64);
    XH_REQUIRE(skip_line >= 0 && skip_line <= line && skip_line_2 >= 0 && skip_line_2 <= (1 << log2n));
    const int n = 1 << log2n, total = n * line;
    k_trans_ats<<<(total + 255) / 256, 256, 0, st>>>(block, coef, type, log2n, shift, line, skip_line, log2n == 2 ? 4 : n - skip_line_2); // (the 4-point forms ignore skip_line_2)
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// 3x3 Sobel gradient (weights 1 2 1) at the sample clamped into the interior: the reference copies the nearest interior value to the border (xevem_mc.c:2341-2395)
__global__ void k_sobel(const pel *__restrict__ pred, int s_pred, int32_t *__restrict__ der, int s_der, int w, int h, int vertical)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= w * h) return;
    const int y = t / w, x = t - y * w;
    const int cy = min(max(y, 1), h - 2), cx = min(max(x, 1), w - 2);
    const pel *c = pred + (long)cy * s_pred + cx;
    der[(long)y * s_der + x] = vertical ? c[s_pred - 1] - c[-s_pred - 1] + 2 * c[s_pred] - 2 * c[-s_pred] + c[s_pred + 1] - c[-s_pred + 1]
                                        : c[1 - s_pred] - c[-1 - s_pred] + 2 * c[1] - 2 * c[-1] + c[1 + s_pred] - c[-1 + s_pred];
}
int xh_sobel(int vertical, const pel *pred, int s_pred, int32_t *der, int s_der, int w, int h, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(pred && der && w >= 3 && h >= 3 && w <= 128 && h <= 128 && s_pred >= w && s_der >= w);
    k_sobel<<<(w * h + 255) / 256, 256, 0, st>>>(pred, s_pred, der, s_der, w, h, vertical);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// The normal equations of the affine gradient search (xevem_mc.c:2397-2447): eq[col + 1][row] += sum c[col] * c[row], eq[col + 1][np] += 8 * sum c[col] * residue, with the
// 32-bit terms c[] of the 4- / 6-parameter model (products formed in 32 bits as the reference's int arithmetic does, sums in 64).  One workgroup: every thread
// keeps the 6 x 7 partial sums of its samples in registers, waves combine by shuffles, the four waves through LDS; the residual is read with the derivative pitch.
__global__ __launch_bounds__(256) void k_equal_coeff(const pel *__restrict__ residue, const int32_t *__restrict__ d0, const int32_t *__restrict__ d1, int s_der,
                                                     long *__restrict__ eq, int w, int h, int vertex_num)
{
    __shared__ long s_part[4][42];
    const int np = vertex_num << 1;
    long acc[42];
#pragma unroll
    for(int i = 0; i < 42; i++) acc[i] = 0;
    for(int t = threadIdx.x; t < w * h; t += 256) {
        const int j = t / w, k = t - j * w, i = j * s_der + k;
        const unsigned a = (unsigned)d0[i], b = (unsigned)d1[i];
        int c[6];
        if(vertex_num == 2) c[0] = (int)a, c[1] = (int)((unsigned)k * a + (unsigned)j * b), c[2] = (int)b, c[3] = (int)((unsigned)j * a - (unsigned)k * b), c[4] = c[5] = 0;
        else c[0] = (int)a, c[1] = (int)((unsigned)k * a), c[2] = (int)b, c[3] = (int)((unsigned)k * b), c[4] = (int)((unsigned)j * a), c[5] = (int)((unsigned)j * b);
        const long r8 = (long)residue[i] * 8;
#pragma unroll
        for(int col = 0; col < 6; col++) {
#pragma unroll
            for(int row = 0; row < 6; row++) acc[col * 7 + row] += (long)c[col] * c[row];
            acc[col * 7 + 6] += (long)c[col] * r8;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for(int i = 0; i < 42; i++) {
        long v = acc[i];
        for(int m = 32; m >= 1; m >>= 1) v += (long)(((unsigned long)__shfl_xor((unsigned)((unsigned long)v >> 32), m, 64) << 32) | __shfl_xor((unsigned)v, m, 64));
        if(lane == 0) s_part[wave][i] = v;
    }
    __syncthreads();
    if(threadIdx.x < 42) {
        const int col = threadIdx.x / 7, row = threadIdx.x % 7;
        if(col < np && (row < np || row == 6)) // (row 6 of the register layout is the right-hand side: column np of the reference's array)
            eq[(col + 1) * 7 + (row == 6 ? np : row)] += s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    }
}
int xh_equal_coeff(const pel *residue, const int32_t *d0, const int32_t *d1, int s_der, long *eq, int w, int h, int vertex_num, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(residue && d0 && d1 && eq && w >= 1 && h >= 1 && w <= 128 && h <= 128 && s_der >= w && (vertex_num == 2 || vertex_num == 3));
    k_equal_coeff<<<1, 256, 0, st>>>(residue, d0, d1, s_der, eq, w, h, vertex_num);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- angular intra prediction of the Main profile (xeve_tbl_intra_pred_ang[group][right], src_main/xevem_ipred.c:456-815) ------------------------------------
// Every sample = a 4-tap interpolation {32 - f, 64 - f, 32 + f, f} / 128 (xevem_tbl_ipred_adi) of one of the three neighbour lines at the position the mode's slopes
// (xevem_tbl_ipred_dxdy: {dx/dy, dy/dx} << 10) project it onto, positions clipped to [-1, w + h - 1]; the six table entries differ in which line and direction.
// lines: [3][w + h + 1] (left, up, right; element 0 = index -1); one thread per sample.
__constant__ int c_ipred_dxdy[33][2] = {
    {0, 0}, {0, 0}, {0, 0}, {2816, 372}, {2048, 512}, {1408, 744}, {1024, 1024}, {744, 1408}, {512, 2048}, {372, 2816}, {256, 4096},
    {128, 8192}, {0, 0}, {128, 8192}, {256, 4096}, {372, 2816}, {512, 2048}, {744, 1408}, {1024, 1024}, {1408, 744}, {2048, 512},
    {2816, 372}, {4096, 256}, {8192, 128}, {0, 0}, {8192, 128}, {4096, 256}, {2816, 372}, {2048, 512}, {1408, 744}, {1024, 1024}, {744, 1408}, {512, 2048}};
__global__ void k_ipred_ang(const pel *__restrict__ lines, pel *__restrict__ dst, int group, int right, int w, int h, int ipm, int maxv)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= w * h) return;
    const int j = t / w, i = t - j * w, pmax = w + h - 1, L = w + h + 1;
    const int m0 = c_ipred_dxdy[ipm][0], m1 = c_ipred_dxdy[ipm][1];
    int line, pos, dir, d; // taps at pos - dir, pos, pos + dir, pos + 2 dir of line (0 left, 1 up, 2 right); d = projected distance << 10
    if(group == 0) {
        d = (j + 1) * m0;
        if(!right || i < w - (d >> 10)) line = 1, pos = i + (d >> 10), dir = 1;
        else d = (w - i) * m1, line = 2, pos = j - (d >> 10), dir = -1;
    }
    else if(group == 1) {
        if(!right) d = (i + 1) * m1, line = 0, pos = j + (d >> 10), dir = 1;
        else {
            d = (w - i) * m1;
            if(j < (d >> 10)) d = (w - i) * m0, line = 1, pos = i + (d >> 10), dir = 1;
            else line = 2, pos = j - (d >> 10), dir = -1;
        }
    }
    else {
        d = (i + 1) * m1;
        if(j < (d >> 10)) d = (j + 1) * m0, line = 1, pos = i - (d >> 10), dir = -1;
        else if(!right) line = 0, pos = j - (d >> 10), dir = -1;
        else d = (w - i) * m1, line = 2, pos = j + (d >> 10), dir = 1;
    }
    const int f = (d >> 5) - ((d >> 10) << 5);
    const pel *src = lines + line * L + 1;
    auto at = [&](int p) { return (int)src[min(max(p, -1), pmax)]; };
    const int v = (int)(int16_t)((at(pos - dir) * (32 - f) + at(pos) * (64 - f) + at(pos + dir) * (32 + f) + at(pos + 2 * dir) * f + 64) >> 7);
    dst[t] = (pel)(v < 0 ? 0 : (v > maxv ? maxv : v));
}
int xh_ipred_ang(int group, int right, const pel *lines, pel *dst, int w, int h, int ipm, int bit_depth, hipStream_t st)
{
    XH_ENTER();
    XH_REQUIRE(lines && dst && group >= 0 && group <= 2 && (right == 0 || right == 1) && w >= 1 && h >= 1 && w <= 128 && h <= 128 && ipm >= 3 && ipm <= 32);
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14);
    k_ipred_ang<<<(w * h + 255) / 256, 256, 0, st>>>(lines, dst, group, right, w, h, ipm, (1 << bit_depth) - 1);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
