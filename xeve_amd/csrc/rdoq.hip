// xeve_amd/csrc/rdoq.hip -- rate-distortion optimised quantisation as a parallel scan.
//
// reference: xeve_rdoq_run_length_cc  src_base/xeve_tq.c:497-649 (+ get_coded_level_rl :458-490, get_ic_rate_cost_rl
// :425-456, xeve_init_err_scale :406-423, zig-zag scan xeve_tbl.c:625 / xeve_util.c:1289-1327).
//
// The reference is a sequential walk over the zig-zag scan with three carried quantities:
//   run        -- zeros since the last non-zero level; only "run == 0 or not" enters the rate (the context index is a
//                 per-component constant in Baseline), i.e. a TWO-STATE automaton driven by "was the previous level 0";
//   base cost  -- running sum of (coded - uncoded) cost and of the "not last" flag cost of every non-zero level;
//   best last  -- first position whose "I am the last coefficient" cost beats everything before (strictly).
// Per coefficient both automaton outcomes (level and cost for run == 0 and for run > 0) are computed independently;
// the states are then resolved by a prefix scan of 2-bit transition functions, the base cost by a 64-bit prefix sum and
// the best last position by an (cost, position) arg-min -- bit-identical to the walk.
#include <cmath>
#include <cstring>
#include <mutex>
#include "xh_common.h"

// the twelve estimates one component uses in Baseline: run / level models c, c + 1 (c = 0 luma, 2 chroma), the last-flag
// model of the component and the cbf pair (xeve_rdoq_set_ctx_cc, xeve_tq.c:492-495; cbf choice :565-583)
struct Est12 {
    int run[2][2], level[2][2], last[2], cbf[2];
};
struct RdoqK {
    int  n, log2n, q_value, q_bits;
    long lambda, err_scale;
    long z_scale, z_thr; // zero pre-test of xeve_quant_nnz (xeve_tq.c:666-699); z_thr < 0: off
    Est12 est;           // uniform estimates (host parameter) ...
    // ... or per block from device memory: record est_dev[est_idx[b]] of xeve_hip_rdoq_est_full, fields at these int offsets
    const int *est_dev, *est_idx;
    int        o_run, o_level, o_last, o_cbf;
};

__device__ __forceinline__ long rl_cost(unsigned abs_level, int rs, const RdoqK &P, const Est12 &E)
{
    unsigned rate;
    if(abs_level == 0) rate = (unsigned)E.run[rs][1];
    else {
        rate = 32768u + (unsigned)E.run[rs][0];
        if(abs_level == 1) rate += (unsigned)E.level[0][0];
        else rate += (unsigned)E.level[0][1] + (unsigned)E.level[1][1] * (abs_level - 2) + (unsigned)E.level[1][0];
    }
    return (long)(int)rate * P.lambda; // s32 rate as the reference, then GET_I_COST
}

// one coefficient: both automaton outcomes
struct Cand {
    unsigned lev[2]; // level if run == 0 / run > 0
    long     d[2];   // coded - uncoded cost for the two cases
    long     unc;
    unsigned maxabs;
    int      neg;
};
__device__ __forceinline__ Cand eval_coef(int v, const RdoqK &P, const Est12 &E)
{
    Cand c;
    const long t = (long)(v < 0 ? -v : v) * P.q_value, cap = (long)INT32_MAX - (1L << (P.q_bits - 1));
    const long ld = (long)(int)(t < cap ? t : cap);
    unsigned m = (unsigned)(ld >> P.q_bits);
    if(!((ld - ((long)m << P.q_bits)) < (1L << (P.q_bits - 1)))) m++;
    const long e1 = (ld * P.err_scale) >> 20;
    c.unc = e1 * e1, c.maxabs = m, c.neg = !(v > 0);
    const unsigned lo = m > 1 ? m - 1 : 1;
#pragma unroll
    for(int rs = 0; rs < 2; rs++) {
        long coded = c.unc + rl_cost(0, rs, P, E);
        unsigned best = 0;
        for(unsigned a = m; a >= lo; a--) {
            const long dd = ld - ((long)a << P.q_bits), e2 = (dd * P.err_scale) >> 20, cost = e2 * e2 + rl_cost(a, rs, P, E);
            if(cost < coded) best = a, coded = cost;
        }
        c.lev[rs] = best, c.d[rs] = coded - c.unc;
    }
    return c;
}

// ---- wave / block primitives (WPB waves of 64 lanes form one block's thread group) --------------------
__device__ __forceinline__ long shfl_up64(long v, int d, int width = 64)
{
    const unsigned lo = __shfl_up((unsigned)v, d, width), hi = __shfl_up((unsigned)((unsigned long)v >> 32), d, width);
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ long shfl_xor64(long v, int m)
{
    const unsigned lo = __shfl_xor((unsigned)v, m, 64), hi = __shfl_xor((unsigned)((unsigned long)v >> 32), m, 64);
    return (long)(((unsigned long)hi << 32) | lo);
}
// transition functions on {0,1}: bit s = next state from state s; identity = 0b10
__device__ __forceinline__ int fcompose(int later, int earlier) { return ((later >> (earlier & 1)) & 1) | (((later >> ((earlier >> 1) & 1)) & 1) << 1); }

// LPB = lanes per block: 64 normally; 16 for n <= 16 (4x4 chroma of an 8x8 CU), where four blocks share one wave and
// every wave-level scan / reduction below is segmented to LPB lanes (WPB must be 1 then).
template <int K, int WPB, int LPB = 64, bool DEV = false>
__global__ __launch_bounds__(256) void k_rdoq(int16_t *__restrict__ coef, int nblk, RdoqK P, const uint16_t *__restrict__ scan,
                                              int32_t *__restrict__ nnz_out)
{
    static_assert(LPB == 64 || WPB == 1, "sub-wave blocks cannot span waves");
    constexpr int SUB = 64 / LPB;                  // blocks per wave
    constexpr int BPW = (4 / WPB) * SUB;           // blocks per workgroup
    __shared__ long s_long[4][4];                  // [wave in workgroup][slot]
    __shared__ int  s_int[4][4];
    const int wave = threadIdx.x >> 6, wlane = threadIdx.x & 63, sub = wlane / LPB, lane = wlane % LPB;
    const int bl = (wave / WPB) * SUB + sub, wib = wave % WPB, t = wib * 64 + lane; // block-local thread id
    const int b = blockIdx.x * BPW + bl;
    const bool live = b < nblk;
    int16_t *blk = coef + (size_t)(live ? b : 0) * P.n;
    const int w0 = (wave / WPB) * WPB; // first wave of my block in the s_* arrays
    Est12 E = P.est;
    if(DEV) { // this block's estimates (xeve_rdoq_bit_est of the coder state its CU starts from)
        const int *rec = P.est_dev + (size_t)(sizeof(xeve_hip_rdoq_est_full) / sizeof(int)) * (P.est_idx ? P.est_idx[live ? b : 0] : 0);
#pragma unroll
        for(int i = 0; i < 2; i++)
#pragma unroll
            for(int j = 0; j < 2; j++) E.run[i][j] = rec[P.o_run + 2 * i + j], E.level[i][j] = rec[P.o_level + 2 * i + j];
        E.last[0] = rec[P.o_last], E.last[1] = rec[P.o_last + 1], E.cbf[0] = rec[P.o_cbf], E.cbf[1] = rec[P.o_cbf + 1];
    }

    // ---- phase 1: candidates of my K consecutive scan positions
    unsigned lev0[K], lev1[K];
    long     d0[K], d1[K];
    int      neg[K];
    long     unc_sum = 0;
    int      sum_all = 0, F = 2; // chunk transition function, starts as identity
    int      zhit = P.z_thr < 0 ? 1 : 0;
#pragma unroll
    for(int i = 0; i < K; i++) {
        const int p = t * K + i;
        int v = 0;
        if(live && p < P.n) v = blk[scan[p]];
        zhit |= ((long)(v < 0 ? -v : v) * P.z_scale) >= P.z_thr;
        const Cand c = eval_coef(v, P, E);
        lev0[i] = c.lev[0], lev1[i] = c.lev[1], d0[i] = c.d[0], d1[i] = c.d[1], neg[i] = c.neg;
        if(p < P.n) {
            unc_sum += c.unc, sum_all += (int)c.maxabs;
            F = fcompose((c.lev[0] == 0 ? 1 : 0) | ((c.lev[1] == 0 ? 1 : 0) << 1), F);
        }
    }
    // block totals: sum_all, uncoded cost
#pragma unroll
    for(int m = 1; m < LPB; m <<= 1) sum_all += __shfl_xor(sum_all, m, 64), unc_sum += shfl_xor64(unc_sum, m), zhit |= __shfl_xor(zhit, m, 64);
    if(LPB == 64) {
        if(lane == 0) s_int[wave][0] = sum_all, s_long[wave][0] = unc_sum, s_int[wave][3] = zhit;
        __syncthreads();
        sum_all = 0, unc_sum = 0, zhit = 0;
#pragma unroll
        for(int i = 0; i < WPB; i++) sum_all += s_int[w0 + i][0], unc_sum += s_long[w0 + i][0], zhit |= s_int[w0 + i][3];
    }
    if(!zhit) sum_all = 0; // the pre-test found nothing codable: the block is zeroed exactly like sum_all == 0
    __syncthreads();

    // ---- phase 2: incoming automaton state of my chunk = (F_{t-1} o ... o F_0)(0)
    int incl = F;
#pragma unroll
    for(int dlt = 1; dlt < LPB; dlt <<= 1) {
        const int o = __shfl_up(incl, dlt, LPB);
        if(lane >= dlt) incl = fcompose(incl, o);
    }
    if(LPB == 64 && lane == 63) s_int[wave][1] = incl;
    __syncthreads();
    int pre = 2; // functions of the earlier waves of my block
    for(int i = 0; i < wib; i++) pre = fcompose(s_int[w0 + i][1], pre);
    int excl = __shfl_up(incl, 1, LPB);
    if(lane == 0) excl = 2;
    const int state_in = fcompose(excl, pre) & 1; // applied to the initial state 0 (run = 0)
    __syncthreads();

    // ---- phase 3: resolve my chunk, running base cost
    const long lz = (long)E.last[0] * P.lambda, lone = (long)E.last[1] * P.lambda;
    unsigned lev[K];
    long     inc[K], dsel[K];
    long     chunk = 0;
    int      st = state_in;
#pragma unroll
    for(int i = 0; i < K; i++) {
        const int p = t * K + i;
        lev[i]  = st ? lev1[i] : lev0[i];
        dsel[i] = st ? d1[i] : d0[i];
        inc[i]  = p < P.n ? dsel[i] + (lev[i] ? lz : 0) : 0;
        if(p >= P.n) lev[i] = 0;
        chunk += inc[i];
        st = lev[i] == 0 ? 1 : 0;
    }
    long incl_sum = chunk;
#pragma unroll
    for(int dlt = 1; dlt < LPB; dlt <<= 1) {
        const long o = shfl_up64(incl_sum, dlt, LPB);
        if(lane >= dlt) incl_sum += o;
    }
    if(LPB == 64 && lane == 63) s_long[wave][1] = incl_sum;
    __syncthreads();
    long base = unc_sum + (long)E.cbf[1] * P.lambda + (incl_sum - chunk); // d64_base_cost before my chunk
    for(int i = 0; i < wib; i++) base += s_long[w0 + i][1];
    // best "I am last" candidate of my chunk: strictly smaller cost wins, earlier position on ties
    long best_cost = 0x7fffffffffffffffL;
    int  best_pos  = 0x7fffffff;
#pragma unroll
    for(int i = 0; i < K; i++) {
        if(lev[i]) {
            const long cur = base + dsel[i] + lone;
            if(cur < best_cost) best_cost = cur, best_pos = t * K + i;
        }
        base += inc[i];
    }
#pragma unroll
    for(int m = 1; m < LPB; m <<= 1) {
        const long oc = shfl_xor64(best_cost, m);
        const int  op = __shfl_xor(best_pos, m, 64);
        if(oc < best_cost || (oc == best_cost && op < best_pos)) best_cost = oc, best_pos = op;
    }
    __syncthreads();
    if(LPB == 64) {
        if(lane == 0) s_long[wave][2] = best_cost, s_int[wave][2] = best_pos;
        __syncthreads();
        best_cost = s_long[w0][2], best_pos = s_int[w0][2];
        for(int i = 1; i < WPB; i++)
            if(s_long[w0 + i][2] < best_cost || (s_long[w0 + i][2] == best_cost && s_int[w0 + i][2] < best_pos)) best_cost = s_long[w0 + i][2], best_pos = s_int[w0 + i][2];
    }
    const long best0 = unc_sum + (long)E.cbf[0] * P.lambda; // d64_best_cost: "code nothing"
    const int best_last = (sum_all != 0 && best_cost < best0) ? best_pos + 1 : 0;

    // ---- phase 4: levels out
    int cnt = 0;
#pragma unroll
    for(int i = 0; i < K; i++) {
        const int p = t * K + i;
        if(live && p < P.n) {
            const int q = p < best_last ? (int)lev[i] : 0;
            blk[scan[p]] = (int16_t)(neg[i] ? -q : q);
            cnt += q != 0;
        }
    }
#pragma unroll
    for(int m = 1; m < LPB; m <<= 1) cnt += __shfl_xor(cnt, m, 64);
    if(LPB == 64) {
        if(lane == 0) s_int[wave][3] = cnt;
        __syncthreads();
        if(live && t == 0) {
            int tot = 0;
            for(int i = 0; i < WPB; i++) tot += s_int[w0 + i][3];
            nnz_out[b] = tot;
        }
    }
    else if(live && lane == 0) nnz_out[b] = cnt;
}

// ---- host ---------------------------------------------------------------------------------------------------
// zig-zag scans (xeve_tbl_scan, generator xeve_util.c:1289-1327), built once per (log2w, log2h) and kept on the device
// All of them (log2 0..6 each way) are built by xeve_hip_init and freed by xeve_hip_shutdown (xh_rdoq_tables_init / _free below): entry points
// only look them up, so a first call inside a stream capture or on a latency-sensitive stream never allocates or synchronises.
static uint16_t *g_scan[7][7];
static int      *g_entropy; // entropy_bits[1024] of xeve_init_bits_est (xeve_mode.c:304-313), device copy
const int *xh_entropy_table() { return g_entropy; }
int xh_get_scan(int log2w, int log2h, const uint16_t **out)
{
    if(log2w < 0 || log2w > 6 || log2h < 0 || log2h > 6 || !g_scan[log2w][log2h]) {
        xh_set_error("xh_get_scan: no scan table for log2 %d x %d (xeve_hip_init builds them)", log2w, log2h);
        return XEVE_HIP_ERR_UNINIT;
    }
    *out = g_scan[log2w][log2h];
    return XEVE_HIP_OK;
}
void xh_rdoq_tables_free()
{
    for(auto &row : g_scan)
        for(auto &d : row)
            if(d) (void)hipFree(d), d = nullptr;
    if(g_entropy) (void)hipFree(g_entropy), g_entropy = nullptr;
}
int xh_rdoq_tables_init()
{
    xh_rdoq_tables_free();
    // one host image of every table, one allocation per table (they are handed out as separate pointers)
    uint16_t hs[64 * 64];
    for(int log2w = 0; log2w <= 6; log2w++)
        for(int log2h = 0; log2h <= 6; log2h++) {
            const int w = 1 << log2w, h = 1 << log2h;
            int pos = 0;
            for(int l = 0; l < w + h - 1; l++) { // odd anti-diagonals run down-left, even ones up-right
                if(l & 1) for(int x = l < w - 1 ? l : w - 1, y = l - x; x >= 0 && y < h; x--, y++) hs[pos++] = (uint16_t)(y * w + x);
                else for(int y = l < h - 1 ? l : h - 1, x = l - y; y >= 0 && x < w; x++, y--) hs[pos++] = (uint16_t)(y * w + x);
            }
            uint16_t *d = nullptr;
            if(hipMalloc((void **)&d, sizeof(uint16_t) * w * h) != hipSuccess || hipMemcpy(d, hs, sizeof(uint16_t) * w * h, hipMemcpyHostToDevice) != hipSuccess) {
                if(d) (void)hipFree(d);
                xh_rdoq_tables_free();
                xh_set_error("xeve_hip_init: building the scan tables failed");
                return XEVE_HIP_ERR_DEVICE;
            }
            g_scan[log2w][log2h] = d;
        }
    int he[1024];
    for(int i = 0; i < 1024; i++) {
        const double p = (512 * (i + 0.5)) / 1024;
        he[i] = (int)(-32768 * (log(p) / log(2.0) - 9));
    }
    if(hipMalloc((void **)&g_entropy, sizeof(he)) != hipSuccess || hipMemcpy(g_entropy, he, sizeof(he), hipMemcpyHostToDevice) != hipSuccess) {
        xh_rdoq_tables_free();
        xh_set_error("xeve_hip_init: building the entropy table failed");
        return XEVE_HIP_ERR_DEVICE;
    }
    return XEVE_HIP_OK;
}

static const int k_quant_scale[2][6] = {{26214, 23302, 20560, 18396, 16384, 14764}, {26214, 23302, 20560, 18396, 16384, 14564}}; // xeve_tq.c:37-38

extern "C" int xeve_hip_rdoq_zt(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                                const xeve_hip_rdoq_est *est, int zero_test, int is_intra_slice, int32_t *nnz, void *stream);

extern "C" int xeve_hip_rdoq(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                             const xeve_hip_rdoq_est *est, int32_t *nnz, void *stream)
{
    return xeve_hip_rdoq_zt(coef, nblk, log2w, log2h, qp, lambda, is_luma, bit_depth, tool_iqt, est, 0, 0, nnz, stream);
}

// common launcher: est != NULL -> uniform estimates; otherwise per block from est_dev / est_idx
template <bool DEV> static void rdoq_launch(int16_t *coef, int nblk, const RdoqK &P, const uint16_t *scan, int32_t *nnz, hipStream_t st)
{
    const int n = P.n;
    if(n <= 16) k_rdoq<1, 1, 16, DEV><<<(nblk + 15) / 16, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n <= 64) k_rdoq<1, 1, 64, DEV><<<(nblk + 3) / 4, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n == 128) k_rdoq<2, 1, 64, DEV><<<(nblk + 3) / 4, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n == 256) k_rdoq<1, 4, 64, DEV><<<nblk, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n == 512) k_rdoq<2, 4, 64, DEV><<<nblk, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n == 1024) k_rdoq<4, 4, 64, DEV><<<nblk, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else if(n == 2048) k_rdoq<8, 4, 64, DEV><<<nblk, 256, 0, st>>>(coef, nblk, P, scan, nnz);
    else k_rdoq<16, 4, 64, DEV><<<nblk, 256, 0, st>>>(coef, nblk, P, scan, nnz);
}

static int rdoq_common(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int ch_type, int bit_depth, int tool_iqt,
                       const xeve_hip_rdoq_est *est, const xeve_hip_rdoq_est_full *est_dev, const int32_t *est_idx, int zero_test,
                       int is_intra_slice, int is_intra_cu, int32_t *nnz, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(coef && (est || est_dev) && nnz && nblk >= 0 && log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6);
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14 && qp >= 0 && qp <= 51 + 6 * (bit_depth - 8) && (tool_iqt == 0 || tool_iqt == 1) && ch_type >= 0 && ch_type <= 2);
    if(nblk == 0) return XEVE_HIP_OK;
    const uint16_t *scan;
    int rc = xh_get_scan(log2w, log2h, &scan);
    if(rc != XEVE_HIP_OK) return rc;
    RdoqK P;
    const int odd = (log2w + log2h) & 1, ns_shift = odd ? 7 : 0, ns_scale = odd ? 181 : 1, ns_offset = odd ? 1 << (ns_shift - 1) : 0;
    const int log2_size = (log2w + log2h) >> 1;
    P.n = 1 << (log2w + log2h), P.log2n = log2w + log2h;
    P.q_value = (k_quant_scale[tool_iqt][qp % 6] * ns_scale + ns_offset) >> ns_shift;
    P.q_bits  = 14 + (15 - bit_depth - log2_size) + qp / 6;
    XH_REQUIRE(P.q_bits >= 1 && P.q_bits <= 30);
    const int c = ch_type == 0 ? 0 : 2, ctx_last = ch_type == 0 ? 0 : 1; // xeve_rdoq_set_ctx_cc with sps_cm_init_flag 0 (xeve_tq.c:492-495)
    P.lambda = (long)(lambda * (double)(1 << 15) + 0.5); // SCALE_BITS, xeve_tq.c:528
    { // ctx->err_scale[qp % 6][log2_size - 1], xeve_init_err_scale (xeve_tq.c:406-423)
        const int tr_shift = 15 - bit_depth - log2_size;
        double e = (double)(1 << 15) * pow(2.0, -tr_shift);
        e = e / k_quant_scale[tool_iqt][qp % 6] / (1 << (bit_depth - 8));
        P.err_scale = (long)(e * (double)(1 << 20));
    }
    if(zero_test) { // xeve_tq.c:673-683
        const int zs = 14 + (15 - bit_depth - log2_size + (odd ? 7 : 0)) + qp / 6;
        P.z_scale    = (long)k_quant_scale[tool_iqt][qp % 6] * (odd ? 181 : 1);
        P.z_thr      = (1L << zs) - ((long)(is_intra_slice ? 201 : 153) << (zs - 9));
    }
    else P.z_scale = 0, P.z_thr = -1;
    memset(&P.est, 0, sizeof(P.est));
    P.est_dev = nullptr, P.est_idx = nullptr, P.o_run = P.o_level = P.o_last = P.o_cbf = 0;
    hipStream_t st = (hipStream_t)stream;
    XhProf prof(XH_PROF_RDOQ, st);
    if(est) {
        for(int i = 0; i < 2; i++)
            for(int j = 0; j < 2; j++) P.est.run[i][j] = est->run[c + i][j], P.est.level[i][j] = est->level[c + i][j];
        P.est.last[0] = est->last[ctx_last][0], P.est.last[1] = est->last[ctx_last][1], P.est.cbf[0] = est->cbf[0], P.est.cbf[1] = est->cbf[1];
        rdoq_launch<false>(coef, nblk, P, scan, nnz, st);
    }
    else { // int offsets of the fields inside xeve_hip_rdoq_est_full; cbf pair as xeve_tq.c:565-583 picks it
        P.est_dev = (const int *)est_dev, P.est_idx = est_idx;
        P.o_run   = (int)(offsetof(xeve_hip_rdoq_est_full, run) / sizeof(int)) + 2 * c;
        P.o_level = (int)(offsetof(xeve_hip_rdoq_est_full, level) / sizeof(int)) + 2 * c;
        P.o_last  = (int)(offsetof(xeve_hip_rdoq_est_full, last) / sizeof(int)) + 2 * ctx_last;
        const size_t o = (!is_intra_cu && ch_type == 0) ? offsetof(xeve_hip_rdoq_est_full, cbf_all)
                         : ch_type == 0                   ? offsetof(xeve_hip_rdoq_est_full, cbf_luma)
                         : ch_type == 1                   ? offsetof(xeve_hip_rdoq_est_full, cbf_cb)
                                                          : offsetof(xeve_hip_rdoq_est_full, cbf_cr);
        P.o_cbf = (int)(o / sizeof(int));
        rdoq_launch<true>(coef, nblk, P, scan, nnz, st);
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_rdoq_zt(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int is_luma, int bit_depth, int tool_iqt,
                                const xeve_hip_rdoq_est *est, int zero_test, int is_intra_slice, int32_t *nnz, void *stream)
{
    XH_REQUIRE(est);
    return rdoq_common(coef, nblk, log2w, log2h, qp, lambda, is_luma ? 0 : 1, bit_depth, tool_iqt, est, nullptr, nullptr, zero_test, is_intra_slice, 0, nnz, stream);
}

extern "C" int xeve_hip_rdoq_dev(int16_t *coef, int nblk, int log2w, int log2h, int qp, double lambda, int ch_type, int bit_depth, int tool_iqt,
                                 const xeve_hip_rdoq_est_full *est, const int32_t *est_idx, int zero_test, int is_intra_slice, int is_intra_cu,
                                 int32_t *nnz, void *stream)
{
    XH_REQUIRE(est);
    return rdoq_common(coef, nblk, log2w, log2h, qp, lambda, ch_type, bit_depth, tool_iqt, nullptr, est, est_idx, zero_test, is_intra_slice, is_intra_cu, nnz, stream);
}

// ---- xeve_rdoq_bit_est (xeve_mode.c:326-372): estimates from a coder state ----------------------------------------------

__global__ void k_rdoq_bit_est(const xeve_hip_sbac *__restrict__ sbac, int n, const int *__restrict__ entropy, int *__restrict__ out)
{
    // one thread per (state, model): biari_no_bits for bin 0 and 1 (xeve_mode.c:315-324)
    constexpr int NM = sizeof(xeve_hip_rdoq_est_full) / sizeof(int) / 2; // 54 models
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n * NM) return;
    const int s = i / NM, m = i % NM;
    // order of xeve_hip_rdoq_est_full: cbf_all, cbf_luma, cbf_cb, cbf_cr, run[24], level[24], last[2]
    const int ci = m == 0 ? XEVE_HIP_CTX_CBF_ALL : m == 1 ? XEVE_HIP_CTX_CBF_LUMA : m == 2 ? XEVE_HIP_CTX_CBF_CB : m == 3 ? XEVE_HIP_CTX_CBF_CR
                   : m < 28 ? XEVE_HIP_CTX_RUN + (m - 4) : m < 52 ? XEVE_HIP_CTX_LEVEL + (m - 28) : XEVE_HIP_CTX_LAST + (m - 52);
    const unsigned cm = sbac[s].ctx[ci], mps = cm & 1, state = cm >> 1;
#pragma unroll
    for(unsigned b = 0; b < 2; b++) out[(size_t)s * 2 * NM + 2 * m + b] = entropy[((b != mps) ? state : 512 - state) << 1];
}

extern "C" int xeve_hip_rdoq_bit_est(const xeve_hip_sbac *sbac, int nstates, xeve_hip_rdoq_est_full *est, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(nstates >= 0);
    if(nstates == 0) return XEVE_HIP_OK;
    XH_REQUIRE(sbac && est);
    XH_REQUIRE(g_entropy != nullptr); // built by xeve_hip_init
    const int total = nstates * (int)(sizeof(xeve_hip_rdoq_est_full) / sizeof(int) / 2);
    k_rdoq_bit_est<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(sbac, nstates, g_entropy, (int *)est);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// ---- host-memory forms of one transform block (the table layer's style) ------------------------------------------------------------
// xeve_tq_nnz (xeve_tq.c:729-748) = xeve_trans + xeve_quant_nnz, and itdq_cu (xeve_itdq.c:454-497) = xeve_dquant + xeve_itrans, on a
// dense block in HOST memory: what the per-component loops of ctx->fn_tq (xeve_sub_block_tq) / ctx->fn_itdp (xeve_itdq) call.
extern "C" int xeve_hip_trans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream);
extern "C" int xeve_hip_itrans(int16_t *coef, int nblk, int log2w, int log2h, int bit_depth, void *stream);
extern "C" int xeve_hip_quant(int16_t *coef, int nblk, int log2w, int log2h, int qp, int scale, int is_intra_slice, int bit_depth, int32_t *nnz, void *stream);
extern "C" int xeve_hip_dquant(int16_t *coef, int nblk, int log2w, int log2h, int scale, int bit_depth, void *stream);

extern "C" int xeve_hip_tq_nnz_host(int16_t *coef, int log2w, int log2h, int qp, double lambda, int ch_type, int is_intra_cu, int is_intra_slice,
                                    int bit_depth, int tool_iqt, const xeve_hip_rdoq_est_full *est, int use_rdoq, int32_t *nnz)
{
    XH_ENTER();
    XH_REQUIRE(coef && nnz && (est || !use_rdoq) && log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6 && qp >= 0 && qp <= 87);
    const size_t n = (size_t)1 << (log2w + log2h), off_nnz = (n * 2 + 15) & ~(size_t)15, off_est = off_nnz + 16;
    char *d = nullptr;
    XH_HIP(hipMalloc((void **)&d, off_est + sizeof(*est)));
    int rc = XEVE_HIP_OK;
    if(hipMemcpy(d, coef, n * 2, hipMemcpyHostToDevice) != hipSuccess || (use_rdoq && hipMemcpy(d + off_est, est, sizeof(*est), hipMemcpyHostToDevice) != hipSuccess)) {
        xh_set_error("xeve_hip_tq_nnz_host: staging failed");
        rc = XEVE_HIP_ERR_DEVICE;
    }
    if(rc == XEVE_HIP_OK) rc = xeve_hip_trans((int16_t *)d, 1, log2w, log2h, bit_depth, nullptr);
    if(rc == XEVE_HIP_OK)
        rc = use_rdoq ? xeve_hip_rdoq_dev((int16_t *)d, 1, log2w, log2h, qp, lambda, ch_type, bit_depth, tool_iqt, (const xeve_hip_rdoq_est_full *)(d + off_est), nullptr,
                                          1, is_intra_slice, is_intra_cu, (int32_t *)(d + off_nnz), nullptr)
                      : xeve_hip_quant((int16_t *)d, 1, log2w, log2h, qp, k_quant_scale[tool_iqt][qp % 6], is_intra_slice, bit_depth, (int32_t *)(d + off_nnz), nullptr);
    if(rc == XEVE_HIP_OK && (hipMemcpy(coef, d, n * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(nnz, d + off_nnz, 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        xh_set_error("xeve_hip_tq_nnz_host: copy back failed");
        rc = XEVE_HIP_ERR_DEVICE;
    }
    (void)hipFree(d);
    return rc;
}

extern "C" int xeve_hip_itdq_host(int16_t *coef, int log2w, int log2h, int qp, int bit_depth)
{
    XH_ENTER();
    XH_REQUIRE(coef && log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6 && qp >= 0 && qp <= 87);
    static const int dq[6] = {40, 45, 51, 57, 64, 71}; // xeve_tbl_dq_scale_b (xeve_tbl.c:237), scale << (qp / 6) (xeve_itdq.c:549)
    const size_t n = (size_t)1 << (log2w + log2h);
    int16_t *d = nullptr;
    XH_HIP(hipMalloc((void **)&d, n * 2));
    int rc = hipMemcpy(d, coef, n * 2, hipMemcpyHostToDevice) == hipSuccess ? XEVE_HIP_OK : XEVE_HIP_ERR_DEVICE;
    if(rc == XEVE_HIP_OK) rc = xeve_hip_dquant(d, 1, log2w, log2h, dq[qp % 6] << (qp / 6), bit_depth, nullptr);
    if(rc == XEVE_HIP_OK) rc = xeve_hip_itrans(d, 1, log2w, log2h, bit_depth, nullptr);
    if(rc == XEVE_HIP_OK && hipMemcpy(coef, d, n * 2, hipMemcpyDeviceToHost) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE;
    if(rc == XEVE_HIP_ERR_DEVICE) xh_set_error("xeve_hip_itdq_host: staging failed");
    (void)hipFree(d);
    return rc;
}
