// do two streams overlap on this device?  a long kernel on one, many short ones on the other
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while(0)
__global__ void k_spin(long long ticks, int *out)
{
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) {}
    if(threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
__global__ void k_spin_scratch(long long ticks, int *out, int n)
{
    int a[40];
    for(int i = 0; i < 40; i++) a[i] = i * n;
    const long long t0 = wall_clock64();
    int k = 0;
    while(wall_clock64() - t0 < ticks) { a[(k + n) % 40] += k; k++; }
    if(threadIdx.x == 0 && out) out[blockIdx.x] = a[k % 40];
}
__global__ void k_tiny(int *p, int v) { if(threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
__global__ void k_tiny_scratch(int *p, int v, int n)
{
    int a[40];
    for(int i = 0; i < 40; i++) a[i] = i * n + v;
    for(int i = 0; i < 8; i++) a[(i * n + v) % 40] += i;
    if(threadIdx.x == 0 && blockIdx.x == 0) p[0] = a[(v + n) % 40];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    int *d;
    CK(hipMalloc(&d, 1 << 20));
    char *big;
    CK(hipMalloc(&big, 64 << 20));
    int wc = 0;
    CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0)); // kHz
    const long long ms50 = (long long)wc * 50; // ticks in 50 ms
    printf("wall clock %d kHz\n", wc);
    for(int variant = 0; variant < 8; variant++) {
        hipStream_t a, b;
        if(variant == 4) {
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, lo));
            CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
        }
        else { CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); }
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        const int N = 10000;
        // B alone
        k_tiny<<<64, 256, 0, b>>>(d, 0);
        CK(hipStreamSynchronize(b));
        double t0 = now();
        for(int i = 0; i < N; i++) {
            if(variant == 5) k_tiny_scratch<<<64, 256, 0, b>>>(d, i, argc);
            else k_tiny<<<64, 256, 0, b>>>(d, i);
            if(variant == 2 && i % 10 == 0) CK(hipMemsetAsync(big, 0, 1 << 20, b));
            if(variant == 6 && i % 10 == 0) CK(hipMemcpyAsync(big, big + (32 << 20), 1 << 20, hipMemcpyDeviceToDevice, b));
        }
        CK(hipStreamSynchronize(b));
        const double alone = now() - t0;
        // A (4 x 50 ms spins) and B together
        t0 = now();
        if(variant == 7) { k_tiny<<<1, 64, 0, b>>>(d, 1); CK(hipEventRecord(ev, b)); CK(hipStreamWaitEvent(a, ev, 0)); }
        for(int r = 0; r < 4; r++) {
            if(variant == 1) k_spin_scratch<<<448, 64, 0, a>>>(ms50, d + 1024, argc);
            else if(variant == 3) k_spin<<<448, 64, 0, a>>>(ms50, d + 1024);
            else k_spin<<<1, 64, 0, a>>>(ms50, d + 1024);
        }
        for(int i = 0; i < N; i++) {
            if(variant == 5) k_tiny_scratch<<<64, 256, 0, b>>>(d, i, argc);
            else k_tiny<<<64, 256, 0, b>>>(d, i);
            if(variant == 2 && i % 10 == 0) CK(hipMemsetAsync(big, 0, 1 << 20, b));
            if(variant == 6 && i % 10 == 0) CK(hipMemcpyAsync(big, big + (32 << 20), 1 << 20, hipMemcpyDeviceToDevice, b));
        }
        const double issued = now() - t0;
        CK(hipStreamSynchronize(b));
        const double b_done = now() - t0;
        CK(hipStreamSynchronize(a));
        const double both = now() - t0;
        static const char *names[] = {"plain", "spin kernel with scratch, 448 blocks", "B with hipMemsetAsync", "spin 448 blocks", "priorities", "B kernels with scratch",
                                      "B with hipMemcpyAsync D2D", "A waits for an event of B first"};
        printf("variant %d (%s): B alone %.1f ms; together: issued %.1f ms, B done %.1f ms, all done %.1f ms (A alone = 200 ms)\n", variant, names[variant], alone * 1e3, issued * 1e3,
               b_done * 1e3, both * 1e3);
        CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b)); CK(hipEventDestroy(ev));
    }
    return 0;
}
