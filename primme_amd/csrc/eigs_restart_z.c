/* eigs_restart_z.c — the complex instantiation of eigs_restart.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_restart.c"
