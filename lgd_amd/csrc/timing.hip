// Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for
// the live roofline numbers).  Disabled by default: a launch then costs one relaxed flag read.
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace lgd {

struct Rec { const char* name; hipEvent_t a, b; };
static bool g_on = false;
static std::mutex g_mu;
static std::vector<Rec> g_recs;

KTimer::KTimer(const char* name, hipStream_t s) : name_(name), s_(s), a_(nullptr), b_(nullptr) {
    if (!g_on) return;
    if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { a_ = b_ = nullptr; return; }
    (void)hipEventRecord(a_, s_);
}
KTimer::~KTimer() {
    if (!a_) return;
    (void)hipEventRecord(b_, s_);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back({name_, a_, b_});
}

}  // namespace lgd

extern "C" {

int lgd_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(lgd::g_mu);
    lgd::g_on = on != 0;
    return LGD_OK;
}

static int collect(char* names, size_t names_len, double* total_ms, double* min_ms, double* max_ms, int32_t* launches,
                   int max_entries) {
    std::vector<lgd::Rec> recs;
    {
        std::lock_guard<std::mutex> lk(lgd::g_mu);
        recs.swap(lgd::g_recs);
    }
    struct Acc { double sum = 0, mn = 1e300, mx = 0; int n = 0; };
    std::map<std::string, Acc> acc;
    for (auto& r : recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = acc[r.name];
            e.sum += ms; e.n += 1;
            if (ms < e.mn) e.mn = ms;
            if (ms > e.mx) e.mx = ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    int n = 0;
    size_t off = 0;
    for (auto& kv : acc) {
        if (n >= max_entries || off + kv.first.size() + 1 > names_len) break;
        std::memcpy(names + off, kv.first.c_str(), kv.first.size() + 1);
        off += kv.first.size() + 1;
        total_ms[n] = kv.second.sum;
        if (min_ms) min_ms[n] = kv.second.mn;
        if (max_ms) max_ms[n] = kv.second.mx;
        launches[n] = kv.second.n;
        ++n;
    }
    return n;
}

int lgd_timing_collect(char* names, size_t names_len, double* total_ms, int32_t* launches, int max_entries) {
    return collect(names, names_len, total_ms, nullptr, nullptr, launches, max_entries);
}

int lgd_timing_collect_ex(char* names, size_t names_len, double* total_ms, double* min_ms, double* max_ms, int32_t* launches,
                          int max_entries) {
    return collect(names, names_len, total_ms, min_ms, max_ms, launches, max_entries);
}

}  // extern "C"
