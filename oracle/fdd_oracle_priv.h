/*
 * fdd_oracle_priv.h -- CPU ORACLE (test infrastructure, NOT product code).
 * The Subdomain<double> record shared by fdd_oracle_subdomain.c (solve path,
 * subdomain.tpp:3942-4646) and fdd_oracle_composite.c (the constructor's
 * composite, subdomain.tpp:86-2747).  See fdd_oracle.h.
 */
#ifndef FDD_ORACLE_PRIV_H
#define FDD_ORACLE_PRIV_H

#include "fdd_oracle.h"

typedef struct
{
    int num_points;
    int num_elements;
    int poly_degree;
    int offset;
} orc_level; /* subdomain.hpp:89-95 */

struct orc_subdomain
{
    int dim;
    int num_levels;
    int *poly_degree;
    orc_level *levels;

    double **D_hat; /* per level */
    double **J_cf;  /* J_cf[l]: level l+1 (coarse) -> level l (fine), n_f x n_c */

    /* subdomain_operator (subdomain.hpp:46-70) */
    int num_points;
    int num_dofs;
    int num_extended_dofs;
    orc_csr Q, Qt;
    double *geom_fact[ORC_NUM_GEOM_FACTS];
    int *offset, *vertex, *level;

    /* superdomain_operator (empty when the region holds the rank's own elements only) */
    int sup_num_dofs;
    int sup_num_extended_dofs;
    orc_csr sup_A, sup_Pt;
    int num_interface_dofs;
    int num_unique_dofs;    /* Subdomain::num_dofs, subdomain.tpp:2583 */
    int tree_done;          /* composite: f was filled by the world-level tree_operator (orc_fdd_tree_operator) */
    int own_points;         /* levels[0].num_points */

    orc_csr Qt_coarse;
    orc_csr Q_int, Qt_int, QQt_int;

    double *norm_weight;
    double *inner_weight;

    int num_values;
    int num_blocks;

    double *work[3];
    double *f, *u_k, *r_k, *r_kp1, *q_k, *z_k, *p_k;
    double **V, **Z;
    int cap_vectors;

    orc_amg *amg; /* low-order preconditioner hierarchy (not owned) */
    double *jacobi_dinv; /* 1 / diagonal over the unique dofs, built on first use (point-Jacobi option) */
};

void *orc_xcalloc(size_t n, size_t sz);
void orc_ranking(double *data, int size); /* the ranking lambda, subdomain.tpp:881-918 */
void orc_subdomain_alloc_solver(orc_subdomain *s, size_t work_size);


#endif
