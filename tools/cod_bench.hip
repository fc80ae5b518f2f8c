// microbenchmark behind DESIGN 5a: cycles per bin of the count-only coefficient coder (xw::cod_events<false>) for one wave with 1 / 8 / 64 active lanes.
// build + run on an MI355X: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/cod_bench.hip -o /tmp/cod_bench && /tmp/cod_bench   (measured: 172-195 cycles per bin, x1.36 with lanes on different events)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../xeve_amd/csrc/walk.h"
__global__ void k(const uint32_t *ev, int nev, const xw::Sbac *st, unsigned *out, long long *cyc, int active)
{
    __shared__ uint16_t ctx[72 * 64];
    if((int)threadIdx.x >= active) return;
    xw::Cod c;
    xw::cod_load(c, st[0], ctx + threadIdx.x, 64);
    xw::cod_reset(c);
    const long long t0 = clock64();
    xw::cod_events<false>(c, ev + (size_t)threadIdx.x * nev, nev, 0);
    const long long t1 = clock64();
    out[threadIdx.x] = xw::cod_bits<false>(c);
    if(threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    const int nev = 512;
    std::vector<uint32_t> ev(nev * 64);
    long bins = 0;
    unsigned seed = 12345;
    for(int i = 0; i < nev * 64; i++) {
        seed = seed * 1664525u + 1013904223u;
        const int run = (seed >> 8) % 3, lev1 = (seed >> 16) % 6;
        ev[i] = (uint32_t)lev1 | ((seed >> 30) & 1) << 15 | (uint32_t)run << 16;
        if(i < nev) bins += (run ? run + 1 : 1) + (lev1 ? lev1 + 1 : 1) + 2;
    }
    xw::Sbac s;
    memset(&s, 0, sizeof(s));
    s.range = 16384;
    for(int i = 0; i < 72; i++) s.ctx[i] = (256 << 1);
    uint32_t *dev; xw::Sbac *ds; unsigned *dout; long long *dc;
    hipMalloc(&dev, nev * 64 * 4), hipMalloc(&ds, sizeof(s)), hipMalloc(&dout, 256 * 4), hipMalloc(&dc, 8);
    hipMemcpy(dev, ev.data(), nev * 64 * 4, hipMemcpyHostToDevice), hipMemcpy(ds, &s, sizeof(s), hipMemcpyHostToDevice);
    for(int active : {1, 8, 64}) {
        for(int rep = 0; rep < 2; rep++) {
            k<<<1, 64>>>(dev, nev, ds, dout, dc, active);
            hipDeviceSynchronize();
        }
        long long c; unsigned b;
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost), hipMemcpy(&b, dout, 4, hipMemcpyDeviceToHost);
        printf("active lanes %2d: %lld cycles for %ld bins (%d events, %u bits) = %.1f cycles / bin\n", active, c, bins, nev, b, (double)c / bins);
    }
    return 0;
}
