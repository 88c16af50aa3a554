/* ccmp_debug.h — test and tool hooks of libccmp.  NOT part of the product ABI.
 *
 * The default library (lib/libccmp.so) exports none of these symbols and does not know the options named here; the same sources
 * built with -DCCMP_DEBUG_HOOKS (closed_chain_motion_planner_amd/build.py links them as lib/libccmp_debug.so beside the default
 * library — every kernel object is shared, only ccmp_api.cpp and ccmp_policy.cpp are compiled twice) do.  The GPU suite runs the
 * two tests that need a hook in a process of their own against the debug library (tests/test_gpu_debug_hooks.py); tools/ that
 * read the scout's predictions select it through CCMP_LIBRARY.
 *
 * The reference has no counterpart for any of this (no tracing, no fault injection). */
#ifndef CCMP_DEBUG_H
#define CCMP_DEBUG_H
#include "ccmp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* runs ccmp_detmath.h's sincos / atan2 / sqrt / div on the device: out[i] = {sin, cos, atan2_nn(|x|, |y|), sqrt(|x|), x / y} —
 * how the tests prove the device arithmetic bit-identical to the host's */
int ccmp_detmath_probe(ccmp_ctx *ctx, const double *x_dev, const double *y_dev, double *out_dev, size_t n, void *hip_stream);
/* an externally supplied processing order for the reference-arithmetic projector (device array of B sample indices, NULL = none) */
int ccmp_ctx_set_order_experimental(ccmp_ctx *ctx, const unsigned int *order_dev);
/* a copy of the FP32 scout's predicted iteration counts of the last call that ran one */
int ccmp_ctx_debug_lpt_pred(ccmp_ctx *ctx, uint16_t *host_out, size_t B);
/* fault injection: the next n compute entry points of this context return CCMP_EHIP before anything is launched */
int ccmp_debug_fail_calls(ccmp_ctx *ctx, int n);
/* option "fail_after_fork" (ccmp_ctx_set_option; listed by ccmp_ctx_option_info of the debug library only): 1 / 2 = the split
 * launches of the projector and of the extend step's bulk form report a failure in front of / behind their side-stream part */

#ifdef __cplusplus
}
#endif
#endif /* CCMP_DEBUG_H */
