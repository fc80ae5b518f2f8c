// xeve_amd/csrc/resident.cpp -- resident pictures: device copies of the host planes the encoder works on, made ONCE per picture.
//
// The reference keeps its pictures in host memory (PICBUF_ALLOCATOR, src_base/xeve_def.h:893-914) and calls the hot path once per CU with
// pointers into them.  The host-memory entry points of this library (xeve_hip_pinter_analyze_cu_host ...) staged every plane on every call
// -- fine for a parity boundary, hopeless at real picture sizes (a 3840x2160 picture has 173 000 CU calls and 90 MB of planes).  With
// xeve_hip_picture_begin() a caller announces each new picture (the reference has the hook: ctx->fn_mode_analyze_frame, called once per
// picture before the CTU loop, src_base/xeve_enc.c:275); from then on a plane is uploaded the first time it is seen within the picture and
// served from HBM for every later call ("upload each finished recon once per picture and the original once per push").
// Contract: no host-memory call may be in flight while xeve_hip_picture_begin() runs, and the planes must not change between two of them.
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "xh_common.h"

namespace {
struct Entry {
    char    *dev   = nullptr;
    size_t   cap   = 0;      // bytes allocated
    size_t   bytes = 0;      // bytes valid
    uint64_t epoch = 0;      // picture this copy belongs to
};
std::mutex                       g_mu;
std::map<const void *, Entry>    g_live;   // host base pointer -> copy of the current picture
std::vector<Entry>               g_free;   // buffers of earlier pictures, reused
std::vector<Entry>               g_retired; // buffers replaced WITHIN the current picture: another thread may still read them, so they rest until the next picture
uint64_t                         g_epoch = 0, g_uploads = 0, g_hits = 0, g_upload_bytes = 0;
bool                             g_on = false; // between xeve_hip_picture_begin and xeve_hip_picture_end / shutdown
} // namespace

bool xh_resident_on()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_on;
}

// device copy of host[0 .. bytes); nullptr on a HIP error (message set)
void *xh_resident(const void *host, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(host);
    if(it != g_live.end() && it->second.epoch == g_epoch && it->second.bytes >= bytes) {
        g_hits++;
        return it->second.dev;
    }
    Entry e;
    if(it != g_live.end()) e = it->second, g_live.erase(it);
    if(e.dev && e.epoch == g_epoch) g_retired.push_back(e), e = Entry(); // a live copy of this picture is never recycled inside the picture (a reader may hold it)
    if(e.cap < bytes) {
        if(e.dev) g_free.push_back(e), e = Entry();
        for(size_t i = 0; i < g_free.size(); i++)
            if(g_free[i].cap >= bytes && g_free[i].cap <= bytes + (bytes >> 2)) {
                e = g_free[i], g_free.erase(g_free.begin() + i);
                break;
            }
        if(!e.dev) {
            if(hipMalloc((void **)&e.dev, bytes) != hipSuccess) {
                xh_set_error("resident pictures: hipMalloc of %zu bytes failed", bytes);
                return nullptr;
            }
            e.cap = bytes;
        }
    }
    if(hipMemcpy(e.dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        xh_set_error("resident pictures: upload of %zu bytes failed", bytes);
        g_free.push_back(e);
        return nullptr;
    }
    e.bytes = bytes, e.epoch = g_epoch;
    g_live[host] = e;
    g_uploads++, g_upload_bytes += bytes;
    return e.dev;
}

void xh_resident_free_all()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for(auto &kv : g_live) (void)hipFree(kv.second.dev);
    for(auto &e : g_free) (void)hipFree(e.dev);
    for(auto &e : g_retired) (void)hipFree(e.dev);
    g_live.clear(), g_free.clear(), g_retired.clear();
    g_epoch = 0, g_on = false;
}

extern "C" int xeve_hip_picture_begin(void)
{
    XH_ENTER();
    std::lock_guard<std::mutex> lk(g_mu);
    g_epoch++, g_on = true;
    for(auto &kv : g_live) g_free.push_back(kv.second); // every copy is stale now; the buffers are reused
    for(auto &e : g_retired) g_free.push_back(e);
    g_live.clear(), g_retired.clear();
    return XEVE_HIP_OK;
}

// Leaves resident mode: the host-memory entry points stage their planes per call again until the next xeve_hip_picture_begin().  For a caller that stops announcing
// pictures (another encoder instance in the same process, the end of a sequence): a stale hit on a re-used host address is impossible afterwards.
extern "C" int xeve_hip_picture_end(void)
{
    XH_ENTER();
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = false;
    for(auto &kv : g_live) g_free.push_back(kv.second);
    for(auto &e : g_retired) g_free.push_back(e);
    g_live.clear(), g_retired.clear();
    return XEVE_HIP_OK;
}

extern "C" int xeve_hip_resident_stats(uint64_t *pictures, uint64_t *uploads, uint64_t *upload_bytes, uint64_t *hits)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if(pictures) *pictures = g_epoch;
    if(uploads) *uploads = g_uploads;
    if(upload_bytes) *upload_bytes = g_upload_bytes;
    if(hits) *hits = g_hits;
    return XEVE_HIP_OK;
}
