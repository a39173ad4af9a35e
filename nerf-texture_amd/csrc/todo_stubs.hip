// TEMPORARY: entry points declared in include/nerftex_hip.h whose kernels are not written yet.
// They fail loudly (never fall back).  Each moves to its own .hip file as it is implemented.
#include "common.hpp"
using namespace nerftex;
#define NOT_YET(name) do { set_error(name ": not implemented yet"); return NERFTEX_ERR_INVALID; } while (0)
extern "C" {
int nerftex_create_raytracer(const float*, uint32_t, const uint32_t*, uint32_t, nerftex_raytracer**) { NOT_YET("create_raytracer"); }
int nerftex_destroy_raytracer(nerftex_raytracer*) { return NERFTEX_OK; }
int nerftex_raytracer_trace(const nerftex_raytracer*, const float*, const float*, float*, float*, float*, int64_t*, uint32_t, void*) { NOT_YET("raytracer_trace"); }
}
