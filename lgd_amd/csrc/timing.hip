// Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for
// the live roofline numbers).  Disabled by default: a launch then costs one relaxed flag read.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace lgd {

struct Rec { const char* name; hipEvent_t a, b; hipStream_t s; };
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
    g_recs.push_back({name_, a_, b_, s_});
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

// Stall diagnosis (tools/stall_repro.py): the launches recorded since the last collect whose END event has not completed, oldest first, as text
// lines "<RUNNING|QUEUED> <stream> <kernel>" -- RUNNING: its start event has completed (the kernel, or something in front of it on that stream that
// is not one of this library's launches, is what the stream is executing), QUEUED: not even that.  Never blocks.  Returns the number of pending
// launches (the text holds as many as fit).
int lgd_timing_pending(char* out, size_t out_len) {
    std::lock_guard<std::mutex> lk(lgd::g_mu);
    int n = 0;
    size_t off = 0;
    if (out && out_len) out[0] = 0;
    for (auto& r : lgd::g_recs) {
        if (hipEventQuery(r.b) == hipSuccess) continue;
        ++n;
        char line[160];
        const int len = snprintf(line, sizeof line, "%s %p %s\n", hipEventQuery(r.a) == hipSuccess ? "RUNNING" : "QUEUED", (void*)r.s, r.name);
        if (out && len > 0 && off + (size_t)len + 1 <= out_len) { std::memcpy(out + off, line, (size_t)len + 1); off += (size_t)len; }
    }
    (void)hipGetLastError();   // (hipErrorNotReady from the queries is not a launch failure)
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
