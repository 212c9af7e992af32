/* eigs_main_z.c — the complex instantiation of eigs_main.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_main.c"
