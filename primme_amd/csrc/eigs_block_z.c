/* eigs_block_z.c — the complex instantiation of eigs_block.c (see eigs_scalar.h) */
#define PA_COMPLEX 1
#include "eigs_block.c"
