// Library identification + thread-local error text (include/nerftex_hip.h).
#include "common.hpp"
#include "workspace.hpp"

#include <cstring>
#include <mutex>

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
        if (s.ptr) (void)hipFree(s.ptr);  // hipFree synchronises the device: no kernel still reads the old buffer
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
}

}  // namespace nerftex

extern "C" {

const char* nerftex_last_error(void) { return nerftex::g_err; }

const char* nerftex_version(void) { return "nerftex_hip 0.1.0 gfx950"; }

}
