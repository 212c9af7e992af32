/* eigs_conv_z.c — the complex instantiation of eigs_conv.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_conv.c"
