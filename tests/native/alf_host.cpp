// tests/native/alf_host.cpp -- TEST INFRASTRUCTURE: the per-element code of the adaptive-loop-filter kernels (xeve_amd/csrc/alf_core.h, what one lane of alf.hip runs)
// compiled for the host and driven element by element in the kernels' own decomposition (a 4x4 block per lane for classification and filtering; one statistics entry per
// lane, blocks in sequence), so that the CPU suite can hold it to the oracle without a GPU.  Nothing of this is linked into the product library.
#include <cstddef>
#include <cstdint>
#include "../../xeve_amd/csrc/alf_core.h"

typedef int16_t pel;
struct PtrAt {
    const pel *p;
    ptrdiff_t  s;
    int operator()(int dy, int dx) const { return p[dy * s + dx]; }
};

extern "C" {
// src / cls at sample (0, 0) of the picture; the area in picture coordinates, multiples of 4
void xa_host_classify(uint8_t *cls, int s_cls, const pel *src, int s_src, int x, int y, int w, int h, int bit_depth)
{
    for(int by = y; by < y + h; by += 4)
        for(int bx = x; bx < x + w; bx += 4) {
            const uint8_t v = xalf::block_class(PtrAt{src + (ptrdiff_t)by * s_src + bx, s_src}, bit_depth);
            for(int a = 0; a < 4; a++)
                for(int b = 0; b < 4; b++) cls[(ptrdiff_t)(by + a) * s_cls + bx + b] = v;
        }
}
// dst / src at the area's first sample; (x, y): the area's position in the classifier plane (7 taps); filter_set 25 x 13 (7 taps) or 7 (5 taps)
void xa_host_filter(int taps, const uint8_t *cls, int s_cls, pel *dst, int s_dst, const pel *src, int s_src, int x, int y, int w, int h, const int16_t *filter_set, int clip_min,
                    int clip_max)
{
    for(int by = 0; by < h; by += 4)
        for(int bx = 0; bx < w; bx += 4) {
            int16_t c[13];
            if(taps == 7) {
                const uint8_t cl = cls[(ptrdiff_t)(y + by) * s_cls + x + bx];
                for(int k = 0; k < 13; k++) c[k] = filter_set[((cl >> 2) & 0x1F) * 13 + xalf::order7(cl & 3, k)];
            }
            else
                for(int k = 0; k < 7; k++) c[k] = filter_set[k];
            for(int a = 0; a < 4; a++)
                for(int b = 0; b < 4; b++) {
                    const PtrAt at{src + (ptrdiff_t)(by + a) * s_src + bx + b, s_src};
                    dst[(ptrdiff_t)(by + a) * s_dst + bx + b] = (pel)(taps == 7 ? xalf::filter_sample<7>(at, c, clip_min, clip_max) : xalf::filter_sample<5>(at, c, clip_min, clip_max));
                }
        }
}
// org / rec / cls at sample (0, 0); E [nclasses][13][13], yv [nclasses][13], pix [nclasses] are WRITTEN (zero beyond ncoef)
void xa_host_stats(int taps, const uint8_t *cls, int s_cls, const pel *org, int s_org, const pel *rec, int s_rec, int x, int y, int w, int h, double *E, double *yv, double *pix)
{
    const int ncoef = taps * taps / 4 + 1, nent = ncoef * (ncoef + 1) / 2 + ncoef + 1, nclasses = cls ? 25 : 1;
    for(int i = 0; i < nclasses * 13 * 13; i++) E[i] = 0;
    for(int i = 0; i < nclasses * 13; i++) yv[i] = 0;
    for(int i = 0; i < nclasses; i++) pix[i] = 0;
    for(int t = 0; t < nent; t++) { // (a lane of the kernel owns entry t of every class)
        int k, l;
        xalf::stat_entry(ncoef, t, k, l);
        long long acc[25] = {0};
        for(int by = y; by < y + h; by += 4)
            for(int bx = x; bx < x + w; bx += 4) {
                const uint8_t cl = cls ? cls[(ptrdiff_t)by * s_cls + bx] : 0;
                long long part = 0;
                for(int a = 0; a < 4; a++)
                    for(int b = 0; b < 4; b++) {
                        const pel *r = rec + (ptrdiff_t)(by + a) * s_rec + bx + b;
                        int e[13];
                        if(taps == 7) xalf::local_sums<7>(PtrAt{r, s_rec}, cl & 3, e);
                        else xalf::local_sums<5>(PtrAt{r, s_rec}, cl & 3, e);
                        const int d = org[(ptrdiff_t)(by + a) * s_org + bx + b] - r[0];
                        part += k < 0 ? d * d : l < 0 ? e[k] * d : e[k] * e[l];
                    }
                acc[(cl >> 2) & 0x1F] += part;
            }
        for(int c = 0; c < nclasses; c++) {
            if(k < 0) pix[c] = (double)acc[c];
            else if(l < 0) yv[c * 13 + k] = (double)acc[c];
            else E[(c * 13 + k) * 13 + l] = E[(c * 13 + l) * 13 + k] = (double)acc[c];
        }
    }
}
}
