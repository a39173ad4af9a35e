// Library identification + thread-local error text (include/nerftex_hip.h).
#include "common.hpp"
#include "workspace.hpp"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace nerftex {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = 0; }

// ---- device scratch ------------------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 16;
struct Slot { void* ptr = nullptr; size_t bytes = 0; };
Slot g_ws[kMaxDevices][kWsSlots];
std::vector<void*> g_retired;  // outgrown buffers: kept until release_workspaces() because a captured HIP graph may still launch
                               // kernels that were recorded with the old pointer (growth is geometric, so there are only a few)
std::mutex g_ws_mutex;
}  // namespace

void* workspace(WorkspaceSlot slot, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("workspace: bad device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Slot& s = g_ws[dev][slot];
    if (s.bytes < bytes) {
        if (s.ptr) g_retired.push_back(s.ptr);
        s.ptr = nullptr;
        s.bytes = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
        if (hipMalloc(&s.ptr, want) != hipSuccess) {
            set_error("workspace: hipMalloc(%zu) failed", want);
            s.ptr = nullptr;
            return nullptr;
        }
        s.bytes = want;
    }
    return s.ptr;
}

void release_workspaces() {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (auto& d : g_ws)
        for (auto& s : d) {
            if (s.ptr) (void)hipFree(s.ptr);
            s = Slot{};
        }
    for (void* p : g_retired) (void)hipFree(p);
    g_retired.clear();
}

// ---- per-kernel timing ------------------------------------------------------------------------------
int g_profile_mode = 0;
namespace {
struct Span { const char* name; hipEvent_t a, b; };
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_free_events;
std::mutex g_profile_mutex;
hipEvent_t take_event() {
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void profile_begin(const char* name, hipStream_t st, int* slot) {
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    Span sp{name, take_event(), take_event()};
    (void)hipEventRecord(sp.a, st);
    g_spans.push_back(sp);
    *slot = (int)g_spans.size() - 1;
}
void profile_end(hipStream_t st, int slot) {
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    if (slot >= 0 && slot < (int)g_spans.size()) (void)hipEventRecord(g_spans[slot].b, st);
}

}  // namespace nerftex

extern "C" {

int nerftex_profile_enable(int on) { nerftex::g_profile_mode = on; return NERFTEX_OK; }

int nerftex_profile_reset(void) {
    std::lock_guard<std::mutex> lock(nerftex::g_profile_mutex);
    (void)hipDeviceSynchronize();
    for (auto& sp : nerftex::g_spans) { nerftex::g_free_events.push_back(sp.a); nerftex::g_free_events.push_back(sp.b); }
    nerftex::g_spans.clear();
    return NERFTEX_OK;
}

int nerftex_profile_report(char* buf, size_t n) {
    if (!buf || n == 0) return NERFTEX_ERR_INVALID;
    std::lock_guard<std::mutex> lock(nerftex::g_profile_mutex);
    (void)hipDeviceSynchronize();
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& sp : nerftex::g_spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { auto& a = agg[sp.name]; a.first++; a.second += ms * 1e3; }
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s\"%s\": {\"calls\": %ld, \"avg_us\": %.3f, \"total_us\": %.3f}", first ? "" : ", ", kv.first.c_str(), kv.second.first,
                 kv.second.second / (double)kv.second.first, kv.second.second);
        out += line;
        first = false;
    }
    out += "}";
    snprintf(buf, n, "%s", out.c_str());
    return NERFTEX_OK;
}

const char* nerftex_last_error(void) { return nerftex::g_err; }

const char* nerftex_version(void) { return "nerftex_hip 0.1.0 gfx950"; }

}
