/* eigs_jd_z.c — the complex instantiation of eigs_jd.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_jd.c"
