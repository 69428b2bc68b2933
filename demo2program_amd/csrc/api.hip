// Library-level entry points of include/d2p.h: version, error string, device info.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void d2p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int d2p_version(void) { return 2; }

extern "C" const char* d2p_last_error(void) { return g_err; }

extern "C" int d2p_device_info(int device, char* name, int name_len, int* cus, int* wave,
                               size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    D2P_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (wave) *wave = prop.warpSize;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return D2P_OK;
}

// ---- optional HIP-event profiling (prof.h) ------------------------------------------------
#include <vector>

#include "prof.h"

struct ProfRec {
    hipEvent_t a, b;
    int key;
    double work;
};
static bool g_prof_enabled = false;
static thread_local int g_prof_tag = 0;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

bool d2p_prof_on() { return g_prof_enabled; }
int d2p_prof_tag() { return g_prof_tag; }

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void d2p_prof_begin(hipStream_t st, int family, double work) {
    ProfRec r;
    r.a = prof_event();
    r.b = prof_event();
    r.key = family * 8 + (g_prof_tag & 7);
    r.work = work;
    (void)hipEventRecord(r.a, st);
    g_prof_recs.push_back(r);
}

void d2p_prof_end(hipStream_t st) {
    if (!g_prof_recs.empty()) (void)hipEventRecord(g_prof_recs.back().b, st);
}

extern "C" int d2p_prof_enable(int on) {
    for (auto& r : g_prof_recs) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    g_prof_enabled = on != 0;
    return D2P_OK;
}

extern "C" int d2p_prof_set_tag(int tag) {
    g_prof_tag = tag;
    return D2P_OK;
}

extern "C" int d2p_prof_read(int key, int* count, double* total_ms, double* total_work) {
    int n = 0;
    double ms = 0.0, work = 0.0;
    for (auto& r : g_prof_recs) {
        if (r.key != key) continue;
        D2P_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D2P_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        work += r.work;
        ++n;
    }
    if (count) *count = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    return D2P_OK;
}
