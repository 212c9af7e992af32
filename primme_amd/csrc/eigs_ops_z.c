/* eigs_ops_z.c — the complex instantiation of eigs_ops.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_ops.c"
