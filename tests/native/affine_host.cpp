// tests/native/affine_host.cpp -- TEST INFRASTRUCTURE: the per-CU set-up and per-sample code of the affine motion-compensation kernel (xeve_amd/csrc/affine_core.h, what
// the lanes of affine.hip run) compiled for the host and driven sample by sample in the kernel's own decomposition (per list and component: the CU's samples one by one,
// the second list averaged into the first), so that the CPU suite can hold it to the reference's goldens without a GPU.  Nothing of this is linked into the product library.
#include <cstddef>
#include <cstdint>
#include <vector>
#include "../../xeve_amd/csrc/affine_core.h"

typedef int16_t pel;
struct RefPic { // xo_refpic / xeve_hip_refpic
    const pel *y, *u, *v;
    int32_t    poc, pad_;
};
struct Job { // xo_affine_job / xeve_hip_affine_job
    int32_t x, y;
    int16_t mv[2][3][2];
    int8_t  refi[2], vertex_num, pad_;
};
// Main-profile filters (the product keeps its copy in affine.hip; the harness takes the table from the caller so that both are exercised against the same goldens)
struct PtrAt {
    const pel *p;
    ptrdiff_t  s;
    int operator()(int dy, int dx) const { return p[dy * s + dx]; }
};
struct BufAt {
    const int16_t *b;
    int            pitch;
    int operator()(int r, int c) const { return b[r * pitch + c]; }
};

// one component (c: 0 luma, 1 / 2 chroma) of one list's prediction of a CU, the kernel's decomposition: out = cw x ch samples
static void host_component(const xaff::Model &m, int sub_w, int sub_h, bool mem_ok, int x, int y, int w, int h, int pic_w, int pic_h, int c, const pel *org, ptrdiff_t s,
                           int bit_depth, const int16_t *coef_l, const int16_t *coef_c, std::vector<int16_t> &out)
{
    const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h;
    out.assign((size_t)cw * ch, 0);
    if(sub_w < 8 || sub_h < 8) { // the enhanced interpolation filter: the bilinear samples of positions -1 .. cw / ch first, then the outputs
        int mx[2], mn[2];
        xaff::eif_range(m, mem_ok, x, y, w, h, pic_w, pic_h, mx, mn);
        const xaff::Eif e = xaff::eif_component(m, mx, mn, c != 0);
        std::vector<int16_t> bl((size_t)(cw + 2) * (ch + 2));
        for(int py = -1; py <= ch; py++)
            for(int px = -1; px <= cw; px++) bl[(size_t)(py + 1) * (cw + 2) + px + 1] = (int16_t)xaff::eif_bilinear(PtrAt{org, s}, e, px, py, bit_depth);
        for(int py = 0; py < ch; py++)
            for(int px = 0; px < cw; px++) out[(size_t)py * cw + px] = (int16_t)xaff::eif_out(BufAt{bl.data(), cw + 2}, px, py, bit_depth);
    }
    else {
        int th, tv, oh, ov;
        xaff::block_vector(m, sub_w, sub_h, x, y, w, h, pic_w, pic_h, th, tv, oh, ov);
        const int fs = c ? 5 : 4, fm = (1 << fs) - 1;
        const int16_t *cx = c ? coef_c + (th & fm) * 4 : coef_l + (th & fm) * 8, *cy = c ? coef_c + (tv & fm) * 4 : coef_l + (tv & fm) * 8;
        for(int py = 0; py < ch; py++)
            for(int px = 0; px < cw; px++) {
                const PtrAt at{org + (ptrdiff_t)(py + (tv >> fs)) * s + px + (th >> fs), s};
                out[(size_t)py * cw + px] = (int16_t)(c ? xaff::mc_sample<4>(at, (oh & fm) != 0, (ov & fm) != 0, cx, cy, bit_depth)
                                                        : xaff::mc_sample<8>(at, (oh & fm) != 0, (ov & fm) != 0, cx, cy, bit_depth));
            }
    }
}

extern "C" void xa_host_affine_mc(const RefPic *refp, int s_l, int s_c, int pic_w, int pic_h, const Job *job, int w, int h, int bit_depth, const int16_t *coef_l /* [16][8] */,
                                  const int16_t *coef_c /* [32][4] */, pel *pred_y, pel *pred_u, pel *pred_v, int *path)
{
    int  sub_w, sub_h;
    bool mem_ok;
    xaff::subblock_size(job->refi, job->mv, job->vertex_num, w, h, sub_w, sub_h, mem_ok);
    if(path) path[0] = sub_w, path[1] = sub_h, path[2] = mem_ok;
    int nth = 0;
    for(int l = 0; l < 2; l++) {
        if(job->refi[l] < 0) continue;
        const RefPic &rp = refp[job->refi[l] * 2 + l];
        const xaff::Model m = xaff::model(job->mv[l], w, h, job->vertex_num);
        for(int c = 0; c < 3; c++) {
            const int s = c ? s_c : s_l;
            const pel *plane = c == 0 ? rp.y : c == 1 ? rp.u : rp.v, *org = plane + (ptrdiff_t)(c ? job->y >> 1 : job->y) * s + (c ? job->x >> 1 : job->x);
            pel *dst = c == 0 ? pred_y : c == 1 ? pred_u : pred_v;
            std::vector<int16_t> out;
            host_component(m, sub_w, sub_h, mem_ok, job->x, job->y, w, h, pic_w, pic_h, c, org, s, bit_depth, coef_l, coef_c, out);
            for(size_t i = 0; i < out.size(); i++) dst[i] = nth ? (pel)((dst[i] + out[i] + 1) >> 1) : out[i];
        }
        nth++;
    }
}

// ---- the affine gradient search in the kernel's decomposition (k_affine_me of affine.hip): the block-wide passes as plain loops over the CU's samples / Hadamard tiles, the
// scalar steps through the SAME functions the kernel's lane 0 runs (xaff::me_*) ----
struct MeJob { // xo_affine_me_job / xeve_hip_affine_me_job
    int32_t  x, y;
    int16_t  mvp[3][2], mv[3][2];
    int8_t   refi, list, bi, vertex_num;
    int32_t  mot_bits_other;
    uint32_t cost;
};
struct PredAt {
    const int16_t *p;
    int            w;
    int operator()(int r, int c) const { return p[r * w + c]; }
};
static int host_satd(const int16_t *org, ptrdiff_t s_org, const int16_t *pred, int w, int h, int bit_depth)
{ // xeve_had's tiling for CUs of 16 and more (xeve_sad.c:1051-1135): 16x8 tiles when wider than high, 8x16 when higher, 8x8 when square
    const int tw = w > h ? 16 : 8, th = w < h ? 16 : 8;
    long sum = 0;
    for(int ty = 0; ty < h; ty += th)
        for(int tx = 0; tx < w; tx += tw) {
            int t[16 * 16];
            for(int r = 0; r < th; r++)
                for(int c = 0; c < tw; c++) t[r * tw + c] = org[(ty + r) * s_org + tx + c] - pred[(ty + r) * w + tx + c];
            for(int pass = 0; pass < 2; pass++) {
                const int n = pass ? th : tw, lines = pass ? tw : th, st = pass ? tw : 1, ls = pass ? 1 : tw;
                for(int q = 0; q < lines; q++)
                    for(int len = 1; len < n; len <<= 1)
                        for(int i = 0; i < n; i++)
                            if(!(i & len)) {
                                const int a = t[q * ls + i * st], b = t[q * ls + (i + len) * st];
                                t[q * ls + i * st] = a + b, t[q * ls + (i + len) * st] = a - b;
                            }
            }
            int sa = (t[0] < 0 ? -t[0] : t[0]) >> 2;
            for(int i = 1; i < tw * th; i++) sa += t[i] < 0 ? -t[i] : t[i];
            sum += tw == th ? (sa + 2) >> 2 : (int)(sa / (2.0 * 2.8284271247461903)); // (2 * sqrt(8): xeve_sad.c:748, 885)
        }
    return (int)(sum >> (bit_depth - 8));
}
extern "C" void xa_host_affine_me(const RefPic *refp, int s_l, int pic_w, int pic_h, const int16_t *org_in, int s_org_in, MeJob *job, int w, int h, int bit_depth,
                                  uint32_t lambda_mv, int num_refp, const int16_t *coef_l, int *rounds_done)
{
    const int bi = job->bi, vn = job->vertex_num, ri = job->refi, np = vn << 1;
    const pel *ref = refp[ri * 2 + job->list].y + (ptrdiff_t)job->y * s_l + job->x;
    const int16_t *org = bi ? org_in : org_in + (ptrdiff_t)job->y * s_org_in + job->x;
    const ptrdiff_t s_org = bi ? w : s_org_in;
    std::vector<int16_t> pred;
    int16_t mvt[3][2], mvd[3][2];
    for(int v = 0; v < 3; v++) mvt[v][0] = job->mv[v][0], mvt[v][1] = job->mv[v][1];
    auto compensate = [&]() {
        const int8_t refi2[2] = {0, -1};
        int16_t mv2[2][3][2] = {};
        for(int v = 0; v < 3; v++) mv2[0][v][0] = mvt[v][0], mv2[0][v][1] = mvt[v][1];
        int  sub_w, sub_h;
        bool mem_ok;
        xaff::subblock_size(refi2, mv2, vn, w, h, sub_w, sub_h, mem_ok);
        host_component(xaff::model(mvt, w, h, vn), sub_w, sub_h, mem_ok, job->x, job->y, w, h, pic_w, pic_h, 0, ref, s_l, bit_depth, coef_l, nullptr, pred);
    };
    compensate();
    int best_bits = xaff::me_mv_bits(mvt, job->mvp, num_refp, ri, vn) + (bi ? job->mot_bits_other : 0);
    uint32_t cost_best = xaff::me_mv_cost(lambda_mv, best_bits) + (uint32_t)(host_satd(org, s_org, pred.data(), w, h, bit_depth) >> bi);
    int rounds = (bi ? 5 : 7) - (vn == 3 ? 2 : 0), it = 0;
    for(; it < rounds; it++) {
        int64_t sums[7][7] = {};
        for(int j = 0; j < h; j++)
            for(int k = 0; k < w; k++) {
                int32_t c[6];
                xaff::me_terms(PredAt{pred.data(), w}, w, h, j, k, vn, c);
                const int16_t e = (int16_t)(org[j * s_org + k] - pred[j * w + k]);
                for(int col = 0; col < np; col++) {
                    for(int row = 0; row < np; row++) sums[col + 1][row] += (int64_t)c[col] * c[row];
                    sums[col + 1][np] += (int64_t)c[col] * e * 8;
                }
            }
        if(xaff::me_update(sums, vn, w, h, mvd)) break;
        for(int v = 0; v < vn; v++) mvt[v][0] = (int16_t)(mvt[v][0] + mvd[v][0]), mvt[v][1] = (int16_t)(mvt[v][1] + mvd[v][1]);
        compensate();
        const int bits = xaff::me_mv_bits(mvt, job->mvp, num_refp, ri, vn) + (bi ? job->mot_bits_other : 0);
        const uint32_t cost = xaff::me_mv_cost(lambda_mv, bits) + (uint32_t)(host_satd(org, s_org, pred.data(), w, h, bit_depth) >> bi);
        if(cost < cost_best) {
            cost_best = cost, best_bits = bits;
            for(int v = 0; v < vn; v++) job->mv[v][0] = mvt[v][0], job->mv[v][1] = mvt[v][1];
        }
    }
    job->cost = cost_best - xaff::me_mv_cost(lambda_mv, best_bits);
    if(rounds_done) *rounds_done = it;
}
