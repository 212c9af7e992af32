/* eigs_harm_z.c — the complex objects of eigs_harm.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_harm.c"
