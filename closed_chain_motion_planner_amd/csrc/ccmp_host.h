/* ccmp_host.h — what the two host translation units of libccmp share (not part of the public ABI). */
#ifndef CCMP_HOST_H
#define CCMP_HOST_H
#include "../../include/ccmp.h"
#include "ccmp_kin.h"

namespace ccmp_host {
/* kernel constants from the problem description: re-packing plus products of constants (ccmp_problem.cpp) */
void make_consts(const ccmp_problem &P, ccmp_consts &K);
}  // namespace ccmp_host
#endif
