/* hip_wrapper.c — the reference's Num_* backend routines (src/linalg/cublas_wrapper.c:162-987) as
 * forwards over the fused device layer; see include/primme_amd_wrapper.h.  Like the reference's GPU
 * backends each call synchronises when an operand lives on the host; the solver of this library does
 * not go through these (it chains the fused launches itself).  Both real stems (the reference's
 * _dprimme and _sprimme instantiations) are generated from hip_wrapper_body.h. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "primme_amd_wrapper.h"

#define WCHK(call) do { int rc_ = (call); if (rc_) return rc_ < 0 ? rc_ : PRIMME_UNEXPECTED_FAILURE; } while (0)
static int is_n(const char *t) { return *t == 'N' || *t == 'n'; }
void pa_larnv_uniform11(int64_t iseed[4], int64_t n, double *x);      /* eigs_dense.c: bit-exact DLARNV(2) stream */

#define WSCALAR double
#define WHSCALAR double
#define WDT HIPK_F64
#define WOPT primme_op_double
#define WN(name) name##_hip_dprimme
#include "hip_wrapper_body.h"
#undef WSCALAR
#undef WHSCALAR
#undef WDT
#undef WOPT
#undef WN

#define WSCALAR float
#define WHSCALAR float
#define WDT HIPK_F32
#define WOPT primme_op_float
#define WN(name) name##_hip_sprimme
#include "hip_wrapper_body.h"
