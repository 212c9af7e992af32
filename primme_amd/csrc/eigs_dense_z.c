/* eigs_dense_z.c — the complex instantiation of eigs_dense.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_dense.c"
