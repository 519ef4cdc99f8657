// What mesh_host.hip / ordering.hip need from the rest of the library when they are built ALONE for the host sanitizers
// (tools/sanitize/run.sh): the error slot and the three operator entry points (never called by the drivers).
#include <cstdarg>
#include <cstdio>
#include "common.hpp"
namespace storm { char g_err[512]; void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); } }
extern "C" const char *storm_hip_last_error(void) { return storm::g_err; }
extern "C" int storm_hip_op_create_from_mesh(storm_hip_ctx *, int64_t, int64_t, int32_t, int64_t, const int64_t *, const int64_t *, const double *, const double *, int64_t, const int64_t *, const double *, const double *, const double *, storm_hip_op **) { return -6; }
extern "C" int storm_hip_op_set_halo(storm_hip_op *, int, const int32_t *, const int64_t *, const int64_t *, const int64_t *) { return -6; }
extern "C" int storm_hip_op_destroy(storm_hip_op *) { return 0; }
