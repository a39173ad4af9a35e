// Shared host/device helpers of libnerftex_hip.so (gfx950 only; no CUDA / portability shims).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/nerftex_hip.h"

namespace nerftex {

// ---- error reporting: int status + thread-local text (include/nerftex_hip.h) -------------------
void set_error(const char* fmt, ...);
void clear_error();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return NERFTEX_ERR_HIP;
    }
    return NERFTEX_OK;
}

#define NERFTEX_HIP_TRY(expr, what)                                                   \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::nerftex::set_error("%s: %s", what, hipGetErrorString(_e));              \
            return NERFTEX_ERR_HIP;                                                   \
        }                                                                             \
    } while (0)

template <typename T>
constexpr T div_up(T a, T b) { return (a + b - 1) / b; }

// compute units of the current device (256 on an MI355X)
inline int device_cus() {
    static int cus[16] = {0};  // per device: a process may drive GPUs of different sizes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// ---- tuning knobs (runtime.hip) ----------------------------------------------------------------------------------------
// A/B switches and tuning parameters of the kernels.  Read ONCE when the library is loaded (NERFTEX_TUNE="name=value,..." plus
// the older one-variable-per-switch names), changed afterwards only through nerftex_tune_set(); a launch reads a plain array.
enum Knob {
    kKnobGridFwd,        // 0 auto, 1 thread-per-sample kernel, 2 level-per-XCD kernel
    kKnobGridBwd,        // 0 auto, 1 per-sample atomics, 2 tile owners (binned / sweep)
    kKnobGridBwdSweep,   // 1: the tile-owner sweep instead of binning
    kKnobGridBwdItems,   // sweep: work items per level (0 = auto)
    kKnobGridBwdSlice,   // binning: records per K4 work item (0 = default)
    kKnobGridBwdNoMerge, // binning: 1 = do not merge runs of samples that share a cell
    kKnobGridBwdProbe,   // phase ablation of K3 / K4 for profiling (0 = off; results are garbage when set)
    kKnobMarch,          // 0 auto, 1 replay, 2 log
    kKnobMarchSerial,    // 1: one-ray-per-lane DDA for the counting pass
    kKnobFfmlpWgPerCu,   // forward: workgroups per CU (0 = default)
    kKnobFfmlpBwdSplit,  // 1: dgrad kernel + wgrad kernel through backward_buffer instead of the fused backward
    kKnobMarchLean,      // 1: the training march's count pass compiled for 64 registers (spills; slower alone, a better neighbour on a shared CU)
    kKnobFfmlpBwdTr,     // 1: fused MLP backward builds its weight-gradient operands with ds_read_b64_tr_b16 instead of selection-matrix MFMAs (slower: A/B)
    kKnobGridBwdStage,   // binning: LDS record slots per fill workgroup (0 = default; the rest of a workgroup's block goes straight to memory)
    kKnobCompositeKeep,  // composite_step_kernel: 64-sample chunks a wave keeps in registers between its forward and backward walk (1..4; 0 = default 2)
    kKnobCount
};
extern long g_knobs[kKnobCount];
inline long knob(Knob k) { return g_knobs[k]; }

// ---- optional per-kernel timing (runtime.hip): { KernelTimer t("name", stream); hipLaunchKernelGGL(...); } --------------
extern int g_profile_mode;                   // 0 off, 1 every kernel, 2 hash-grid kernels only
constexpr int kTimeAny = 1, kTimeGrid = 2;  // timer groups
void profile_begin(const char* name, hipStream_t st, int* slot);
void profile_end(hipStream_t st, int slot);
struct KernelTimer {
    hipStream_t st;
    int slot = -1;
    KernelTimer(const char* name, hipStream_t s, int group = kTimeAny) : st(s) {
        if (g_profile_mode == 1 || (g_profile_mode == 2 && group == kTimeGrid)) profile_begin(name, s, &slot);
    }
    ~KernelTimer() { if (slot >= 0) profile_end(st, slot); }
};

// ---- device-side types --------------------------------------------------------------------------
using half_t = _Float16;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;  // gfx950 wavefront

// "rows per unit" of a launch sized by a device-side count (the inference loop: units = alive rays, rows per unit = n_step): a plain number,
// or -- bit 31 set, NERFTEX_ROWS_AUTO(N, F) of include/nerftex_hip.h -- "derive it from the count": n_step = clamp(F N / count, F, 8 F), the
// rule of nerf/renderer.py:470 with F N slots per iteration, evaluated by every kernel of the iteration from the same device word, so that
// an iteration can be recorded into a HIP graph without the host knowing how many rays are alive
__host__ __device__ __forceinline__ uint32_t unit_rows(uint32_t code, uint32_t count) {
    if (!(code >> 31)) return code;
    const uint32_t F = (code >> 24) & 127u, N = code & 0xffffffu;
    const uint32_t q = F * N / (count ? count : 1u);
    return q < F ? F : (q > 8u * F ? 8u * F : q);
}

}  // namespace nerftex
