/* eigs_scalar.h — the scalar type of the projected problem (the reference's HSCALAR, src/include/template.h).
 *
 * The host solver sources that touch coefficient-space data (eigs_main.c, eigs_ops.c, eigs_conv.c,
 * eigs_restart.c, eigs_block.c, eigs_dense.c, eigs_jd.c, eigs_harm.c) are compiled twice, like the reference's templated sources: once
 * as they are (HS = double: hip_dprimme / hip_sprimme) and once with PA_COMPLEX defined (HS = double complex:
 * the native path of hip_zprimme / hip_cprimme) through the one-line wrappers eigs_*_z.c.  In the complex
 * objects every external function of those files carries the suffix _z (the list below is checked by the
 * linker: a missing entry is a duplicate symbol).  eigs_dynamic.c (timings and cost ratios of the dynamic method
 * switch, nothing scalar-typed) and the parameter / callback plumbing exist once and serve both.
 *
 * Device-layer conventions for complex panels (include/primme_amd_kernels.h): inner products, projection
 * coefficients, Ritz coefficient vectors and axpy factors are (re, im) pairs; Ritz values, shifts, squared
 * norms and scale factors are real.  SD = doubles per scalar in the reduction / coefficient buffers.
 */
#ifndef EIGS_SCALAR_H
#define EIGS_SCALAR_H

#ifdef PA_COMPLEX
#include <complex.h>
typedef double _Complex HS;
#define SD 2
#define PA_IS_COMPLEX 1
#define HS_CONJ(x) conj(x)
#define HS_RE(x) creal(x)
#define HS_ABS(x) cabs(x)
#define HS_ABS2(x) (creal(x) * creal(x) + cimag(x) * cimag(x))

#define pa_sym_eig pa_sym_eig_z
#define pa_sym_eig_gen pa_sym_eig_gen_z
#define pa_potrf_upper pa_potrf_upper_z
#define pa_trsm_left_upper_trans pa_trsm_left_upper_trans_z
#define pa_trsm_left_upper pa_trsm_left_upper_z
#define pa_trsm_right_upper pa_trsm_right_upper_z
#define pa_permute_cols pa_permute_cols_z
#define pa_submatrix pa_submatrix_z
#define pa_update_cholesky pa_update_cholesky_z
#define pa_ortho_block_gram pa_ortho_block_gram_z
#define pa_random_col pa_random_col_z
#define pa_ortho_cgs pa_ortho_cgs_z
#define pa_ortho_local_vec pa_ortho_local_vec_z
#define pa_update_projection pa_update_projection_z
#define pa_solve_H_RR pa_solve_H_RR_z
#define pa_solve_H pa_solve_H_z
#define pa_push_coefficients pa_push_coefficients_z
#define pa_ritz_update pa_ritz_update_z
#define pa_project_once pa_project_once_z
#define pa_check_convergence pa_check_convergence_z
#define pa_map_vecs pa_map_vecs_z
#define pa_prepare_candidates pa_prepare_candidates_z
#define pa_block_first_reorder pa_block_first_reorder_z
#define pa_restart pa_restart_z
#define pa_eigs_solve pa_eigs_solve_z
#define pa_evecs_hat_init pa_evecs_hat_init_z
#define pa_evecs_hat_update pa_evecs_hat_update_z
#define pa_correction_jdqmr pa_correction_jdqmr_z
#define pa_svd pa_svd_z
#define pa_update_Q pa_update_Q_z
#define pa_update_QtV pa_update_QtV_z
#define pa_solve_H_harm pa_solve_H_harm_z
#define pa_solve_H_ref pa_solve_H_ref_z
#define pa_prepare_vecs pa_prepare_vecs_z
#define pa_restart_harmonic pa_restart_harmonic_z
#define pa_restart_refined pa_restart_refined_z
#else
typedef double HS;
#define SD 1
#define PA_IS_COMPLEX 0
#define HS_CONJ(x) (x)
#define HS_RE(x) (x)
#define HS_ABS(x) fabs(x)
#define HS_ABS2(x) ((x) * (x))
#endif

#endif
