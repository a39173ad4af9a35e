// Library identification + thread-local error text (include/nerftex_hip.h).
#include "common.hpp"
#include "workspace.hpp"

#include <atomic>

#include <cstdlib>
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

// ---- tuning knobs -------------------------------------------------------------------------------
long g_knobs[kKnobCount] = {0};
namespace {
const char* const kKnobNames[kKnobCount] = {"grid_fwd", "grid_bwd", "grid_bwd_sweep", "grid_bwd_items", "grid_bwd_slice", "grid_bwd_nomerge",
                                            "grid_bwd_probe", "march", "march_serial", "ffmlp_wg_per_cu",
                                            "ffmlp_bwd_split", "march_lean", "ffmlp_bwd_tr", "grid_bwd_stage", "composite_keep"};
int knob_index(const char* name, size_t n) {
    for (int k = 0; k < kKnobCount; k++)
        if (strlen(kKnobNames[k]) == n && strncmp(kKnobNames[k], name, n) == 0) return k;
    return -1;
}
// environment, read once at load: NERFTEX_TUNE="grid_bwd=1,march_serial=1" and the one-variable-per-switch names of round 1
struct KnobInit {
    KnobInit() {
        auto env = [](const char* n) { const char* v = getenv(n); return v ? v : ""; };
        const char* v;
        v = env("NERFTEX_GRID_FWD"); g_knobs[kKnobGridFwd] = v[0] == 'p' ? 1 : v[0] == 'l' ? 2 : 0;
        v = env("NERFTEX_GRID_BWD"); g_knobs[kKnobGridBwd] = v[0] == 'a' ? 1 : v[0] == 'o' ? 2 : 0;
        g_knobs[kKnobGridBwdSweep] = env("NERFTEX_GRID_BWD_ALGO")[0] == 's';
        g_knobs[kKnobGridBwdItems] = atol(env("NERFTEX_GRID_BWD_ITEMS"));
        g_knobs[kKnobGridBwdSlice] = atol(env("NERFTEX_GRID_BWD_SLICE"));
        g_knobs[kKnobGridBwdNoMerge] = getenv("NERFTEX_GRID_BWD_NOMERGE") != nullptr;
        v = env("NERFTEX_MARCH"); g_knobs[kKnobMarch] = v[0] == 'r' ? 1 : v[0] == 'l' ? 2 : 0;
        g_knobs[kKnobMarchSerial] = env("NERFTEX_MARCH_COUNT")[0] == 's';
        g_knobs[kKnobFfmlpWgPerCu] = atol(env("NERFTEX_FFMLP_WG_PER_CU"));
        g_knobs[kKnobFfmlpBwdSplit] = env("NERFTEX_FFMLP_BWD")[0] == 's';
        for (const char* p = env("NERFTEX_TUNE"); *p;) {
            const char* eq = strchr(p, '=');
            if (!eq) break;
            const int k = knob_index(p, (size_t)(eq - p));
            if (k >= 0) g_knobs[k] = atol(eq + 1);
            const char* c = strchr(eq, ',');
            if (!c) break;
            p = c + 1;
        }
    }
} g_knob_init;
}  // namespace

// ---- device scratch ------------------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 16;
struct Slot { void* ptr = nullptr; size_t bytes = 0; };
struct SlotSet { Slot slot[kWsSlots]; };
std::map<hipStream_t, SlotSet> g_ws[kMaxDevices];  // a handful of streams per device
const hipStream_t kCaptureSet = reinterpret_cast<hipStream_t>(~(uintptr_t)0);  // key of the set shared by all stream captures
std::vector<void*> g_retired;  // outgrown buffers: kept until release_workspaces() because a captured HIP graph may still launch
                               // kernels that were recorded with the old pointer (growth is geometric, so there are only a few)
std::mutex g_ws_mutex;
}  // namespace

std::atomic<unsigned> g_ws_touched{0};
thread_local int t_capture_set = 0;  // nerftex_workspace_capture_set: which scratch set the captures of this thread record against

void* workspace(WorkspaceSlot slot, size_t bytes, hipStream_t stream) {
    g_ws_touched.fetch_or(1u << (unsigned)slot, std::memory_order_relaxed);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("workspace: bad device");
        return nullptr;
    }
    // A stream under capture is a one-off handle; what it records is replayed later, on some other stream, against the pointers baked
    // in now.  All captures share one set (graphs that use it must not be replayed concurrently with each other: INTEGRATION.md) instead
    // of leaving one set per captured graph behind.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        stream = reinterpret_cast<hipStream_t>(reinterpret_cast<uintptr_t>(kCaptureSet) - (uintptr_t)t_capture_set);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Slot& s = g_ws[dev][stream].slot[slot];
    if (s.bytes < bytes) {
        if (s.ptr) g_retired.push_back(s.ptr);
        s.ptr = nullptr;
        s.bytes = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
        // an allocation while some stream of the process is being captured is legal only in the relaxed capture mode
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        const hipError_t err = hipMalloc(&s.ptr, want);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (err != hipSuccess) {
            set_error("workspace: hipMalloc(%zu) failed", want);
            s.ptr = nullptr;
            return nullptr;
        }
        s.bytes = want;
        // debugging aid (NERFTEX_POISON_WORKSPACE=<byte>, read once): fill fresh scratch with that byte -- 255 makes every float a NaN and every
        // index enormous -- so that a kernel which reads scratch nobody wrote shows up in the tests instead of depending on what the pages held
        static const int poison = [] { const char* v = getenv("NERFTEX_POISON_WORKSPACE"); return v && *v ? atoi(v) : -1; }();
        if (poison >= 0) {
            (void)hipThreadExchangeStreamCaptureMode(&mode);
            (void)hipMemset(s.ptr, poison & 255, want);
            (void)hipThreadExchangeStreamCaptureMode(&mode);
        }
    }
    return s.ptr;
}

bool peek_workspace(int slot, hipStream_t stream, void** ptr, size_t* bytes) {
    int dev = 0;
    if (slot < 0 || slot >= kWsSlots || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws[dev].find(stream);
    *ptr = it == g_ws[dev].end() ? nullptr : it->second.slot[slot].ptr;
    *bytes = it == g_ws[dev].end() ? 0 : it->second.slot[slot].bytes;
    return true;
}

void release_workspaces() {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (auto& d : g_ws) {
        for (auto& per_stream : d)
            for (auto& s : per_stream.second.slot)
                if (s.ptr) (void)hipFree(s.ptr);
        d.clear();
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

// A stream under capture gets NO timing events: a pair recorded into a graph cannot be read back after a replay (hipEventRecordWithFlags(...,
// hipEventRecordExternal), the form that could, fails with "invalid argument" during capture on ROCm 7.2 -- tried in round 4), so the spans of
// a captured launch would only ever report garbage.  bench.py times the kernels of the REPLAYED step with rocprofv3 instead.
static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}
void profile_begin(const char* name, hipStream_t st, int* slot) {
    if (capturing(st)) return;  // *slot stays -1: profile_end is not called
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

int nerftex_tune_set(const char* name, long value) {
    nerftex::clear_error();
    const int k = name ? nerftex::knob_index(name, strlen(name)) : -1;
    if (k < 0) {
        nerftex::set_error("nerftex_tune_set: unknown knob '%s'", name ? name : "(null)");
        return NERFTEX_ERR_INVALID;
    }
    nerftex::g_knobs[k] = value;
    return NERFTEX_OK;
}

int nerftex_release_workspaces(void) {
    nerftex::clear_error();
    int n = 0, cur = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return NERFTEX_OK;  // no device: nothing was allocated
    (void)hipGetDevice(&cur);
    for (int d = 0; d < n; d++)
        if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
    nerftex::release_workspaces();
    return NERFTEX_OK;
}

long nerftex_tune_get(const char* name) {
    const int k = name ? nerftex::knob_index(name, strlen(name)) : -1;
    return k < 0 ? -1 : nerftex::g_knobs[k];
}

// test aid: bit mask of the scratch slots (csrc/workspace.hpp WorkspaceSlot) library calls have asked for since the last call of this function
int nerftex_workspace_capture_set(int set) {
    nerftex::clear_error();
    if (set < 0 || set > 255) {
        nerftex::set_error("workspace_capture_set: 0 (the default set) .. 255");
        return NERFTEX_ERR_INVALID;
    }
    nerftex::t_capture_set = set;
    return NERFTEX_OK;
}

unsigned nerftex_workspace_slots_touched(void) {
    const unsigned m = nerftex::g_ws_touched.exchange(0u, std::memory_order_relaxed);
    return m;
}

int nerftex_debug_workspace(int slot, void* stream, void** ptr, size_t* bytes) {
    nerftex::clear_error();
    if (!ptr || !bytes || !nerftex::peek_workspace(slot, static_cast<hipStream_t>(stream), ptr, bytes)) {
        nerftex::set_error("nerftex_debug_workspace: bad slot / device / output pointers");
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}

const char* nerftex_last_error(void) { return nerftex::g_err; }

const char* nerftex_version(void) { return "nerftex_hip 0.1.0 gfx950"; }

}
