/*
 * spfm_oracle.c -- CPU restatement (float64, single thread) of the reference's
 * proximal coordinate-descent hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (sparsepoly_amd/) never does and fails loudly without its HIP extension.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against fixtures in tests/golden/ that were produced by running the
 * reference's own source (neonnnnn/sparsepoly, /root/reference) under an
 * identity numba stub (oracle/gen_golden.py).
 *
 * Every function cites the reference file:line it restates.  Arithmetic is
 * written in the reference's evaluation order; build with
 *   gcc -O2 -fno-fast-math -ffp-contract=off
 * so no FMA contraction or re-association changes the rounding.
 *
 * Layouts (all row-major, as the reference's NumPy arrays):
 *   CSC: indptr int64[d+1], indices int32[nnz], data f64[nnz]
 *        (reference: sparsepoly/dataset.py:94-116, CSCDataset.get_column)
 *   pcd : P (k, d); A (n, a_cols) with a_cols = top_degree + 1
 *   pbcd: P (d, k); A (n, a_rows, k), dA (n, a_rows - 1, k), a_rows = top_degree + 1
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SPO_MAX_DEGREE 16

enum { SPO_LOSS_SQUARED = 0, SPO_LOSS_SQUARED_HINGE = 1, SPO_LOSS_LOGISTIC = 2 };
enum {
    SPO_REG_L1 = 0,
    SPO_REG_L21 = 1,
    SPO_REG_SQUAREDL12 = 2,
    SPO_REG_SQUAREDL21 = 3,
    SPO_REG_OMEGATI = 4,
    SPO_REG_OMEGACS = 5
};

/* ---------------------------------------------------------------- losses */

/* sparsepoly/loss.py:18,32,59 */
double spo_loss_mu(int loss) {
    if (loss == SPO_LOSS_SQUARED) return 1.0;
    if (loss == SPO_LOSS_LOGISTIC) return 0.25;
    return 2.0;
}

/* sparsepoly/loss.py:23-24 (Squared), :44-51 (Logistic), :67-71 (SquaredHinge) */
double spo_dloss(int loss, double p, double y) {
    if (loss == SPO_LOSS_SQUARED) return p - y;
    if (loss == SPO_LOSS_LOGISTIC) {
        double z = p * y;
        if (z > 18.0) return -y * exp(-z);
        if (z < -18.0) return -y;
        return -y / (exp(z) + 1.0);
    }
    {
        double z = 1 - p * y;
        if (z > 0) return -2 * y * z;
        return 0.0;
    }
}

/* sparsepoly/loss.py:20-21, :34-42, :61-65 */
double spo_loss(int loss, double p, double y) {
    if (loss == SPO_LOSS_SQUARED) return 0.5 * ((p - y) * (p - y));
    if (loss == SPO_LOSS_LOGISTIC) {
        double z = p * y;
        if (z > 18) return exp(-z);
        if (z < -18) return -z;
        return log(1.0 + exp(-z));
    }
    {
        double z = 1 - p * y;
        if (z > 0) return z * z;
        return 0.0;
    }
}

void spo_dloss_vec(int loss, int64_t n, const double* p, const double* y, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = spo_dloss(loss, p[i], y[i]);
}

double spo_loss_sum(int loss, int64_t n, const double* p, const double* y) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += spo_loss(loss, p[i], y[i]);
    return s;
}

/* ----------------------------------------------------------- regularizer */

typedef struct {
    int kind;
    int top_degree; /* degree passed to init_cache_* : sizes _cache/_dcache */
    int d, k;
    double* abs_p;  /* (d)  L12/OmegaTI  `_abs_p` */
    double* norms;  /* (d)  L21sq/OmegaCS `_norms` */
    double* cache;  /* (top_degree+1) ; SquaredL12: cache[0] ; SquaredL21: cache[0] is the scalar */
    double* dcache; /* (top_degree+1) */
    int ncache;
} spo_reg;

spo_reg* spo_reg_create(int kind) {
    spo_reg* r = (spo_reg*)calloc(1, sizeof(spo_reg));
    r->kind = kind;
    return r;
}

void spo_reg_destroy(spo_reg* r) {
    if (!r) return;
    free(r->abs_p);
    free(r->norms);
    free(r->cache);
    free(r->dcache);
    free(r);
}

static void reg_alloc(spo_reg* r, int degree, int d, int k) {
    free(r->abs_p);
    free(r->norms);
    free(r->cache);
    free(r->dcache);
    r->top_degree = degree;
    r->d = d;
    r->k = k;
    /* degree == -1 (all-subsets): the scalar _cache_all_subsets lives in cache[0] */
    r->ncache = (degree > 0) ? degree + 1 : 2;
    r->abs_p = (double*)calloc((size_t)d, sizeof(double));
    r->norms = (double*)calloc((size_t)d, sizeof(double));
    r->cache = (double*)calloc((size_t)r->ncache, sizeof(double));
    r->dcache = (double*)calloc((size_t)r->ncache, sizeof(double));
}

/* Returns 0 ok, -1 = the ValueError the reference raises.
 * l1.py:19-20 (no-op), squaredl12.py:24-31 (degree>2 raises), omegati.py:49-57.
 * L21/SquaredL21/OmegaCS have no init_cache_pcd (reference fails inside Numba):
 * reported as -2 = unsupported solver/regularizer pair. */
int spo_reg_init_cache_pcd(spo_reg* r, int degree, int d, int k) {
    if (degree + 1 > SPO_MAX_DEGREE) return -3;
    switch (r->kind) {
        case SPO_REG_L1:
            reg_alloc(r, degree, d, k);
            return 0;
        case SPO_REG_SQUAREDL12:
            if (degree > 2) return -1;
            reg_alloc(r, degree, d, k);
            return 0;
        case SPO_REG_OMEGATI:
            if (degree <= 0 && degree != -1) return -1;
            reg_alloc(r, degree, d, k);
            if (degree == -1) r->cache[0] = 1.0; /* omegati.py:54-55 */
            return 0;
        default:
            return -2;
    }
}

/* l1.py:35-36, l21.py:23-25, squaredl21.py:27-34, omegacs.py:41-50 */
int spo_reg_init_cache_pbcd(spo_reg* r, int degree, int d, int k) {
    if (degree + 1 > SPO_MAX_DEGREE) return -3;
    switch (r->kind) {
        case SPO_REG_L1:
        case SPO_REG_L21:
            reg_alloc(r, degree, d, k);
            return 0;
        case SPO_REG_SQUAREDL21:
            if (degree != 2) return -1;
            reg_alloc(r, degree, d, k);
            r->cache[0] = 0;
            return 0;
        case SPO_REG_OMEGACS:
            if (degree <= 0 && degree != -1) return -1;
            reg_alloc(r, degree, d, k);
            if (degree > 0)
                r->dcache[1] = 1.0; /* omegacs.py:46 */
            else
                r->cache[0] = 1.0; /* omegacs.py:47-48 */
            return 0;
        default:
            return -2;
    }
}

/* compute_cache_pcd_all: l1.py:22-23, squaredl12.py:33-40 (transpose=True: no-op),
 * omegati.py:59-60 -- all no-ops for the default configuration. */

/* squaredl12.py:42-45 ; omegati.py:62-74 ; l1.py:25-26 */
void spo_reg_compute_cache_pcd(spo_reg* r, const double* P, int degree, int s) {
    const int d = r->d;
    const double* ps = P + (size_t)s * d;
    if (r->kind == SPO_REG_SQUAREDL12) {
        double sum = 0.0;
        for (int j = 0; j < d; ++j) {
            r->abs_p[j] = fabs(ps[j]);
            sum += r->abs_p[j];
        }
        r->cache[0] = sum; /* np.sum is pairwise: equal to ~1e-16 relative */
    } else if (r->kind == SPO_REG_OMEGATI && degree == -1) { /* omegati.py:75-80 */
        r->cache[0] = 1.0;
        for (int j = 0; j < d; ++j) {
            double a = fabs(ps[j]);
            r->abs_p[j] = a;
            r->cache[0] *= 1.0 + a;
        }
    } else if (r->kind == SPO_REG_OMEGATI) {
        r->cache[0] = 1.0;
        for (int t = 1; t < r->ncache; ++t) r->cache[t] = 0.0;
        for (int t = 0; t < r->ncache; ++t) r->dcache[t] = 0.0;
        r->dcache[1] = 1.0;
        for (int j = 0; j < d; ++j) {
            double a = fabs(ps[j]);
            r->abs_p[j] = a;
            for (int deg = 0; deg < degree; ++deg)
                r->cache[degree - deg] += r->cache[degree - deg - 1] * a;
        }
    }
}

/* squaredl12.py:47-50 ; omegati.py:76-80 */
void spo_reg_update_cache_pcd(spo_reg* r, const double* P, int degree, int s, int j) {
    const double pv = P[(size_t)s * r->d + j];
    if (r->kind == SPO_REG_SQUAREDL12) {
        r->cache[0] -= r->abs_p[j];
        r->cache[0] += fabs(pv);
    } else if (r->kind == SPO_REG_OMEGATI) {
        double a = fabs(pv);
        if (degree > 0) {
            for (int deg = 1; deg < degree; ++deg)
                r->cache[deg] = r->dcache[deg + 1] + r->dcache[deg] * a;
        } else { /* all-subsets, omegati.py:87-88 */
            r->cache[0] *= 1.0 + a;
        }
        r->abs_p[j] = a;
    }
}

/* l1.py:32-33 ; squaredl12.py:52-57 ; omegati.py:82-99 */
double spo_reg_prox_cd(spo_reg* r, double p_sj, double strength, int degree, int j) {
    if (r->kind == SPO_REG_L1) {
        double sg = (p_sj > 0) ? 1.0 : ((p_sj < 0) ? -1.0 : 0.0); /* np.sign */
        double m = fabs(p_sj) - strength;
        return sg * (m > 0.0 ? m : 0.0);
    }
    if (r->kind == SPO_REG_SQUAREDL12) {
        double dcache = r->cache[0] - r->abs_p[j];
        p_sj /= 1 + 2 * strength;
        double sg = (p_sj > 0) ? 1.0 : -1.0;
        double m = fabs(p_sj) - 2 * strength * dcache / (1 + 2 * strength);
        return sg * (m > 0 ? m : 0);
    }
    /* OmegaTI */
    {
        double sg = (p_sj > 0) ? 1.0 : -1.0;
        if (degree > 0) {
            for (int deg = 2; deg <= degree; ++deg) {
                r->dcache[deg] = r->cache[deg - 1];
                r->dcache[deg] -= r->dcache[deg - 1] * r->abs_p[j];
                if (r->dcache[deg] < 0) r->dcache[deg] = 0.0;
            }
            strength *= r->dcache[degree];
        } else { /* all-subsets, omegati.py:100-102 */
            r->cache[0] /= 1.0 + r->abs_p[j];
            strength *= r->cache[0];
        }
        double m = fabs(p_sj) - strength;
        return sg * (m > 0 ? m : 0);
    }
}

/* regularizer/utils.py:14-18 : (sum |x|^2)^(1/2) */
static double row_l2_pow(const double* p, int k) {
    double s = 0.0;
    for (int t = 0; t < k; ++t) s += fabs(p[t]) * fabs(p[t]);
    return pow(s, 0.5);
}

static double row_l2_sqrt(const double* p, int k) {
    double s = 0.0;
    for (int t = 0; t < k; ++t) s += p[t] * p[t];
    return sqrt(s);
}

/* omegacs.py:52-62 (degree > 0 branch) */
static void omegacs_recompute(spo_reg* r, int degree) {
    if (degree <= 0) { /* all-subsets, omegacs.py:60-62 */
        r->cache[0] = 1.0;
        for (int j = 0; j < r->d; ++j) r->cache[0] *= 1.0 + r->norms[j];
        return;
    }
    for (int t = 1; t < r->ncache; ++t) r->cache[t] = 0.0;
    r->cache[0] = 1.0;
    for (int j = 0; j < r->d; ++j) {
        double nj = r->norms[j];
        for (int deg = 0; deg < degree; ++deg)
            r->cache[degree - deg] += r->cache[degree - deg - 1] * nj;
    }
}

/* squaredl21.py:36-38 ; omegacs.py:64-66 */
void spo_reg_compute_cache_pbcd(spo_reg* r, const double* P, int degree) {
    const int d = r->d, k = r->k;
    if (r->kind == SPO_REG_SQUAREDL21) {
        double sum = 0.0;
        for (int j = 0; j < d; ++j) {
            r->norms[j] = row_l2_pow(P + (size_t)j * k, k);
            sum += r->norms[j];
        }
        r->cache[0] = sum;
    } else if (r->kind == SPO_REG_OMEGACS) {
        for (int j = 0; j < d; ++j) r->norms[j] = row_l2_pow(P + (size_t)j * k, k);
        omegacs_recompute(r, degree);
    }
}

/* squaredl21.py:40-43 ; omegacs.py:68-76 */
void spo_reg_update_cache_pbcd(spo_reg* r, const double* P, int degree, int j) {
    const int k = r->k;
    const double* pj = P + (size_t)j * k;
    if (r->kind == SPO_REG_SQUAREDL21) {
        r->cache[0] -= r->norms[j];
        r->norms[j] = row_l2_sqrt(pj, k);
        r->cache[0] += r->norms[j];
    } else if (r->kind == SPO_REG_OMEGACS && degree == -1) { /* omegacs.py:77-81 */
        double l2 = row_l2_sqrt(pj, k);
        r->cache[0] *= 1.0 + l2;
        r->norms[j] = l2;
        if (r->cache[0] < 0) omegacs_recompute(r, -1);
    } else if (r->kind == SPO_REG_OMEGACS) {
        double l2 = row_l2_sqrt(pj, k);
        for (int deg = 1; deg <= degree; ++deg) {
            r->cache[deg] += r->dcache[deg] * l2;
            r->cache[deg] -= r->dcache[deg] * r->norms[j];
        }
        r->norms[j] = l2;
        double mn = r->cache[0];
        for (int t = 1; t < r->ncache; ++t)
            if (r->cache[t] < mn) mn = r->cache[t];
        if (mn < 0) omegacs_recompute(r, degree);
    }
}

/* l1.py:44-45 ; l21.py:33-38 ; squaredl21.py:45-55 ; omegacs.py:78-106 (in place on p_j) */
void spo_reg_prox_bcd(spo_reg* r, double* p_j, double strength, int degree, int j) {
    const int k = r->k;
    if (r->kind == SPO_REG_L1) {
        for (int s = 0; s < k; ++s) {
            double v = p_j[s];
            double sg = (v > 0) ? 1.0 : ((v < 0) ? -1.0 : 0.0);
            double m = fabs(v) - strength;
            p_j[s] = sg * (m > 0.0 ? m : 0.0);
        }
        return;
    }
    if (r->kind == SPO_REG_L21) {
        double l2 = row_l2_sqrt(p_j, k);
        if (l2 > strength) {
            double f = 1.0 - strength / l2;
            for (int s = 0; s < k; ++s) p_j[s] *= f;
        } else {
            for (int s = 0; s < k; ++s) p_j[s] = 0.0;
        }
        return;
    }
    if (r->kind == SPO_REG_SQUAREDL21) {
        double den = 1 + 2 * strength;
        for (int s = 0; s < k; ++s) p_j[s] /= den;
        double l2 = row_l2_sqrt(p_j, k);
        if (r->cache[0] < r->norms[j]) { /* "to avoid numerical error" */
            double sum = 0.0;
            for (int q = 0; q < r->d; ++q) sum += r->norms[q];
            r->cache[0] = sum;
        }
        double dcache = r->cache[0] - r->norms[j];
        strength = 2 * dcache * strength / (1.0 + 2 * strength);
        if (l2 > strength) {
            double f = 1.0 - strength / l2;
            for (int s = 0; s < k; ++s) p_j[s] *= f;
        } else {
            for (int s = 0; s < k; ++s) p_j[s] = 0.0;
        }
        return;
    }
    /* OmegaCS */
    if (degree == -1) { /* all-subsets, omegacs.py:99-101,103-106 */
        double l2 = row_l2_sqrt(p_j, k);
        r->cache[0] /= 1.0 + r->norms[j];
        strength *= r->cache[0];
        if (l2 > strength) {
            double f = 1 - strength / l2;
            for (int s = 0; s < k; ++s) p_j[s] *= f;
        } else {
            for (int s = 0; s < k; ++s) p_j[s] = 0.0;
        }
        return;
    }
    {
        double l2 = row_l2_sqrt(p_j, k);
        for (int deg = 2; deg <= degree; ++deg) {
            r->dcache[deg] = r->cache[deg - 1];
            r->dcache[deg] -= r->dcache[deg - 1] * r->norms[j];
        }
        double mn = r->dcache[0];
        for (int t = 1; t < r->ncache; ++t)
            if (r->dcache[t] < mn) mn = r->dcache[t];
        if (mn < 0) { /* omegacs.py:90-96 "numerical error" fallback */
            r->norms[j] = 0.0;
            omegacs_recompute(r, degree - 1);
            r->dcache[0] = 0.0;
            r->dcache[1] = 1.0;
            for (int deg = 2; deg <= degree; ++deg) r->dcache[deg] = r->cache[degree - 1];
        }
        strength *= r->dcache[degree];
        if (l2 > strength) {
            double f = 1 - strength / l2;
            for (int s = 0; s < k; ++s) p_j[s] *= f;
        } else {
            for (int s = 0; s < k; ++s) p_j[s] = 0.0;
        }
    }
}

/* raw state access for the per-call regularizer traces (golden G5) */
void spo_reg_get_state(const spo_reg* r, double* cache, double* dcache, double* abs_p,
                       double* norms) {
    if (cache) memcpy(cache, r->cache, sizeof(double) * (size_t)r->ncache);
    if (dcache) memcpy(dcache, r->dcache, sizeof(double) * (size_t)r->ncache);
    if (abs_p) memcpy(abs_p, r->abs_p, sizeof(double) * (size_t)r->d);
    if (norms) memcpy(norms, r->norms, sizeof(double) * (size_t)r->d);
}

void spo_reg_set_state(spo_reg* r, const double* cache, const double* dcache,
                       const double* abs_p, const double* norms) {
    if (cache) memcpy(r->cache, cache, sizeof(double) * (size_t)r->ncache);
    if (dcache) memcpy(r->dcache, dcache, sizeof(double) * (size_t)r->ncache);
    if (abs_p) memcpy(r->abs_p, abs_p, sizeof(double) * (size_t)r->d);
    if (norms) memcpy(r->norms, norms, sizeof(double) * (size_t)r->d);
}

/* ------------------------------------------------------------- cd_linear */

/* sparsepoly/optimizer/cd_linear.py:8-33 */
double spo_cd_linear_epoch(double* w, int64_t n, int d, const int64_t* indptr,
                           const int32_t* indices, const double* data, const double* y,
                           double* y_pred, const double* col_norm_sq, double alpha, int loss,
                           const int32_t* indices_feature, int n_feat) {
    (void)n;
    (void)d;
    double sum_viol = 0;
    const double mu = spo_loss_mu(loss);
    for (int jj = 0; jj < n_feat; ++jj) {
        const int j = indices_feature[jj];
        const int64_t b = indptr[j], e = indptr[j + 1];
        double update = 0;
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            const double val = data[ii];
            update += spo_dloss(loss, y_pred[i], y[i]) * val;
        }
        update += alpha * w[j];
        double inv_step_size = mu * col_norm_sq[j] + alpha;
        update /= inv_step_size;
        w[j] -= update;
        sum_viol += fabs(update);
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            y_pred[i] -= update * data[ii];
        }
    }
    return sum_viol;
}

/* ------------------------------------------------------------------- pcd */

/* sparsepoly/optimizer/pcd.py:15-30 */
void spo_pcd_precompute_A(int64_t n, int d, const int64_t* indptr, const int32_t* indices,
                          const double* data, const double* P, double* A, int a_cols, int s,
                          int degree) {
    for (int64_t i = 0; i < n; ++i) {
        A[i * a_cols] = 1.0;
        for (int t = 1; t < a_cols; ++t) A[i * a_cols + t] = 0.0;
    }
    for (int j = 0; j < d; ++j) {
        const double p_sj = P[(size_t)s * d + j];
        for (int64_t ii = indptr[j]; ii < indptr[j + 1]; ++ii) {
            double* Ai = A + (int64_t)indices[ii] * a_cols;
            const double x_ij = data[ii];
            for (int t = 0; t < degree; ++t)
                Ai[degree - t] += Ai[degree - t - 1] * p_sj * x_ij;
        }
    }
}

/* sparsepoly/optimizer/pcd.py:71-137 (with _update :33-68 and _grad_anova :8-12 inlined) */
double spo_pcd_epoch(double* P, int k, int64_t n, int d, const int64_t* indptr,
                     const int32_t* indices, const double* data, const double* y,
                     double* y_pred, const double* lams, int degree, double beta, double gamma,
                     double eta, spo_reg* reg, int loss, double* A, int a_cols,
                     const int32_t* indices_component, int n_comp,
                     const int32_t* indices_feature, int n_feat) {
    (void)k;
    double dA[SPO_MAX_DEGREE];
    double sum_viol = 0;
    const double mu = spo_loss_mu(loss);
    /* regularizer.compute_cache_pcd_all: no-op for l1 / squaredl12(transpose) / omegati */
    for (int ss = 0; ss < n_comp; ++ss) {
        const int s = indices_component[ss];
        spo_pcd_precompute_A(n, d, indptr, indices, data, P, A, a_cols, s, degree);
        spo_reg_compute_cache_pcd(reg, P, degree, s);
        const double lam = lams[s];
        for (int jj = 0; jj < n_feat; ++jj) {
            const int j = indices_feature[jj];
            const int64_t b = indptr[j], e = indptr[j + 1];
            const double p_sj_old = P[(size_t)s * d + j];
            /* _update, pcd.py:52-68 */
            double inv_step_size = 0;
            double update = 0;
            for (int64_t ii = b; ii < e; ++ii) {
                const int i = indices[ii];
                const double x_ij = data[ii];
                const double* Ai = A + (int64_t)i * a_cols;
                dA[0] = x_ij;
                for (int t = 1; t < degree; ++t) dA[t] = x_ij * (Ai[t] - p_sj_old * dA[t - 1]);
                update += spo_dloss(loss, y_pred[i], y[i]) * dA[degree - 1];
                inv_step_size += dA[degree - 1] * dA[degree - 1];
            }
            inv_step_size *= mu;
            inv_step_size += beta;
            update *= lam;
            update += beta * p_sj_old;
            update /= inv_step_size;
            double p_sj_new = p_sj_old - eta * update;
            p_sj_new = spo_reg_prox_cd(reg, p_sj_new, eta * gamma / inv_step_size, degree, j);
            /* pcd.py:119-133 */
            update = p_sj_old - p_sj_new;
            sum_viol += fabs(update);
            P[(size_t)s * d + j] = p_sj_new;
            for (int64_t ii = b; ii < e; ++ii) {
                const int i = indices[ii];
                const double x_ij = data[ii];
                double* Ai = A + (int64_t)i * a_cols;
                dA[0] = x_ij;
                for (int deg = 1; deg < degree; ++deg) {
                    dA[deg] = x_ij * (Ai[deg] - p_sj_old * dA[deg - 1]);
                    Ai[deg] -= update * dA[deg - 1];
                }
                Ai[degree] -= update * dA[degree - 1];
                y_pred[i] -= lam * update * dA[degree - 1];
            }
            spo_reg_update_cache_pcd(reg, P, degree, s, j);
        }
    }
    return sum_viol;
}

/* ------------------------------------------------------------------ pbcd */

/* sparsepoly/optimizer/pbcd.py:18-33 ; A is (n, a_rows, k) */
void spo_pbcd_precompute_A(int64_t n, int d, const int64_t* indptr, const int32_t* indices,
                           const double* data, const double* P, int k, double* A, int a_rows,
                           int degree) {
    const size_t slab = (size_t)a_rows * k;
    for (int64_t i = 0; i < n; ++i) {
        double* Ai = A + (size_t)i * slab;
        for (int s = 0; s < k; ++s) Ai[s] = 1.0;
        for (size_t q = (size_t)k; q < slab; ++q) Ai[q] = 0.0;
    }
    for (int j = 0; j < d; ++j) {
        const double* pj = P + (size_t)j * k;
        for (int64_t ii = indptr[j]; ii < indptr[j + 1]; ++ii) {
            double* Ai = A + (size_t)indices[ii] * slab;
            const double x_ij = data[ii];
            for (int t = 0; t < degree; ++t)
                for (int s = 0; s < k; ++s)
                    Ai[(degree - t) * k + s] += Ai[(degree - t - 1) * k + s] * pj[s] * x_ij;
        }
    }
}

/* sparsepoly/optimizer/pbcd.py:82-148 (with _update :36-79, _grad_anova :9-15 inlined).
 * scratch: grad, inv_step_sizes, p_j_old each (k). */
double spo_pbcd_epoch(double* P, int k, int64_t n, int d, const int64_t* indptr,
                      const int32_t* indices, const double* data, const double* y,
                      double* y_pred, const double* lams, int degree, double beta, double gamma,
                      double eta, spo_reg* reg, int loss, double* A, double* dA, int a_rows,
                      const int32_t* indices_feature, int n_feat) {
    double sum_viol = 0;
    const double mu = spo_loss_mu(loss);
    const size_t slabA = (size_t)a_rows * k;
    const size_t slabD = (size_t)(a_rows - 1) * k;
    double* grad = (double*)malloc(sizeof(double) * (size_t)k * 3);
    double* inv_step_sizes = grad + k;
    double* p_j_old = grad + 2 * k;

    spo_pbcd_precompute_A(n, d, indptr, indices, data, P, k, A, a_rows, degree);
    spo_reg_compute_cache_pbcd(reg, P, degree);
    for (int jj = 0; jj < n_feat; ++jj) {
        const int j = indices_feature[jj];
        const int64_t b = indptr[j], e = indptr[j + 1];
        double* p_j = P + (size_t)j * k;
        for (int s = 0; s < k; ++s) p_j_old[s] = p_j[s];
        /* _update, pbcd.py:56-79 */
        for (int s = 0; s < k; ++s) {
            grad[s] = 0.0;
            inv_step_sizes[s] = 0.0;
        }
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            const double x_ij = data[ii];
            const double* Ai = A + (size_t)i * slabA;
            double* dAi = dA + (size_t)i * slabD;
            for (int s = 0; s < k; ++s) dAi[s] = x_ij;
            for (int t = 1; t < degree; ++t)
                for (int s = 0; s < k; ++s)
                    dAi[t * k + s] = x_ij * (Ai[t * k + s] - p_j[s] * dAi[(t - 1) * k + s]);
            const double dl = spo_dloss(loss, y_pred[i], y[i]);
            for (int s = 0; s < k; ++s) {
                const double v = dAi[(degree - 1) * k + s];
                grad[s] += dl * v;
                inv_step_sizes[s] += v * v;
            }
        }
        double inv_step_size = 0;
        for (int s = 0; s < k; ++s) inv_step_size += inv_step_sizes[s];
        inv_step_size *= mu;
        inv_step_size += beta;
        for (int s = 0; s < k; ++s) grad[s] *= lams[s];
        for (int s = 0; s < k; ++s) grad[s] += beta * p_j[s];
        for (int s = 0; s < k; ++s) grad[s] /= inv_step_size;
        for (int s = 0; s < k; ++s) p_j[s] -= eta * grad[s];
        spo_reg_prox_bcd(reg, p_j, eta * gamma / inv_step_size, degree, j);
        /* pbcd.py:135-146 */
        double* updates = p_j_old;
        for (int s = 0; s < k; ++s) updates[s] -= p_j[s];
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            double* Ai = A + (size_t)i * slabA;
            const double* dAi = dA + (size_t)i * slabD;
            for (int deg = 1; deg <= degree; ++deg)
                for (int s = 0; s < k; ++s)
                    Ai[deg * k + s] -= updates[s] * dAi[(deg - 1) * k + s];
            for (int s = 0; s < k; ++s)
                y_pred[i] -= lams[s] * updates[s] * dAi[(degree - 1) * k + s];
        }
        spo_reg_update_cache_pbcd(reg, P, degree, j);
        double l1 = 0.0;
        for (int s = 0; s < k; ++s) l1 += fabs(updates[s]);
        sum_viol += l1;
    }
    free(grad);
    return sum_viol;
}

/* ------------------------------------------------------------ all-subsets */

/* sparsepoly/optimizer/pcd_all.py:8-18 */
void spo_pcd_all_precompute_A(int64_t n, int d, const int64_t* indptr, const int32_t* indices,
                              const double* data, const double* p_s, double* A) {
    for (int64_t i = 0; i < n; ++i) A[i] = 1.0;
    for (int j = 0; j < d; ++j) {
        const double p_sj = p_s[j];
        for (int64_t ii = indptr[j]; ii < indptr[j + 1]; ++ii)
            A[indices[ii]] *= 1.0 + p_sj * data[ii];
    }
}

/* sparsepoly/optimizer/pcd_all.py:44-102 (with _update :21-41 inlined); A is (n) */
double spo_pcd_all_epoch(double* P, int k, int64_t n, int d, const int64_t* indptr,
                         const int32_t* indices, const double* data, const double* y,
                         double* y_pred, const double* lams, double beta, double gamma,
                         double eta, spo_reg* reg, int loss, double* A,
                         const int32_t* indices_component, int n_comp,
                         const int32_t* indices_feature, int n_feat) {
    (void)k;
    double sum_viol = 0;
    const double mu = spo_loss_mu(loss);
    for (int ss = 0; ss < n_comp; ++ss) {
        const int s = indices_component[ss];
        spo_pcd_all_precompute_A(n, d, indptr, indices, data, P + (size_t)s * d, A);
        spo_reg_compute_cache_pcd(reg, P, -1, s);
        const double lam = lams[s];
        for (int jj = 0; jj < n_feat; ++jj) {
            const int j = indices_feature[jj];
            const int64_t b = indptr[j], e = indptr[j + 1];
            const double p_sj_old = P[(size_t)s * d + j];
            double update = 0, inv_step_size = 0;
            for (int64_t ii = b; ii < e; ++ii) {
                const int i = indices[ii];
                const double x_ij = data[ii];
                const double dA = x_ij * A[i] / (1.0 + x_ij * p_sj_old);
                update += spo_dloss(loss, y_pred[i], y[i]) * dA;
                inv_step_size += dA * dA;
            }
            inv_step_size *= mu;
            inv_step_size += beta;
            update *= lam;
            update += beta * p_sj_old;
            update /= inv_step_size;
            double p_sj_new = p_sj_old - eta * update;
            p_sj_new = spo_reg_prox_cd(reg, p_sj_new, eta * gamma / inv_step_size, -1, j);
            update = p_sj_old - p_sj_new;
            sum_viol += fabs(update);
            P[(size_t)s * d + j] = p_sj_new;
            for (int64_t ii = b; ii < e; ++ii) {
                const int i = indices[ii];
                const double x_ij = data[ii];
                y_pred[i] -= lam * A[i];
                A[i] /= 1.0 + x_ij * p_sj_old;
                A[i] *= 1.0 + x_ij * p_sj_new;
                y_pred[i] += lam * A[i];
            }
            spo_reg_update_cache_pcd(reg, P, -1, s, j);
        }
    }
    return sum_viol;
}

/* sparsepoly/optimizer/pbcd_all.py:68-132 (with _update :23-65 inlined); P (d,k), A (n,k) */
double spo_pbcd_all_epoch(double* P, int k, int64_t n, int d, const int64_t* indptr,
                          const int32_t* indices, const double* data, const double* y,
                          double* y_pred, const double* lams, double beta, double gamma,
                          double eta, spo_reg* reg, int loss, double* A,
                          const int32_t* indices_feature, int n_feat) {
    double sum_viol = 0;
    const double mu = spo_loss_mu(loss);
    double* grad = (double*)malloc(sizeof(double) * (size_t)k * 3);
    double* inv_step_sizes = grad + k;
    double* p_j_old = grad + 2 * k;
    for (int64_t i = 0; i < n * k; ++i) A[i] = 1.0; /* pbcd_all.py:9-20 */
    for (int j = 0; j < d; ++j)
        for (int64_t ii = indptr[j]; ii < indptr[j + 1]; ++ii) {
            double* Ai = A + (size_t)indices[ii] * k;
            const double x_ij = data[ii];
            for (int s = 0; s < k; ++s) Ai[s] *= 1.0 + P[(size_t)j * k + s] * x_ij;
        }
    spo_reg_compute_cache_pbcd(reg, P, -1);
    for (int jj = 0; jj < n_feat; ++jj) {
        const int j = indices_feature[jj];
        const int64_t b = indptr[j], e = indptr[j + 1];
        double* p_j = P + (size_t)j * k;
        for (int s = 0; s < k; ++s) {
            p_j_old[s] = p_j[s];
            grad[s] = 0.0;
            inv_step_sizes[s] = 0.0;
        }
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            const double x_ij = data[ii];
            const double dl = spo_dloss(loss, y_pred[i], y[i]);
            const double* Ai = A + (size_t)i * k;
            for (int s = 0; s < k; ++s) {
                const double dA = x_ij * Ai[s] / (1.0 + x_ij * p_j[s]);
                grad[s] += dl * dA;
                inv_step_sizes[s] += dA * dA;
            }
        }
        for (int s = 0; s < k; ++s) grad[s] *= lams[s];
        for (int s = 0; s < k; ++s) grad[s] += beta * p_j[s];
        double inv_step_size = 0;
        for (int s = 0; s < k; ++s) inv_step_size += inv_step_sizes[s];
        inv_step_size *= mu;
        inv_step_size += beta;
        for (int s = 0; s < k; ++s) grad[s] /= inv_step_size;
        for (int s = 0; s < k; ++s) p_j[s] -= eta * grad[s];
        spo_reg_prox_bcd(reg, p_j, eta * gamma / inv_step_size, -1, j);
        for (int64_t ii = b; ii < e; ++ii) {
            const int i = indices[ii];
            const double x_ij = data[ii];
            double* Ai = A + (size_t)i * k;
            double dot = 0.0;
            for (int s = 0; s < k; ++s) dot += lams[s] * Ai[s];
            y_pred[i] -= dot;
            for (int s = 0; s < k; ++s) {
                Ai[s] /= 1.0 + x_ij * p_j_old[s];
                Ai[s] *= 1.0 + x_ij * p_j[s];
            }
            dot = 0.0;
            for (int s = 0; s < k; ++s) dot += lams[s] * Ai[s];
            y_pred[i] += dot;
        }
        spo_reg_update_cache_pbcd(reg, P, -1, j);
        double l1 = 0.0;
        for (int s = 0; s < k; ++s) l1 += fabs(p_j_old[s] - p_j[s]);
        sum_viol += l1;
    }
    free(grad);
    return sum_viol;
}

/* kernels.py:117-137 all_subsets_kernel + poly_predict: out[i] = sum_s lams[s] prod_j (1 + x_ij p_sj) */
void spo_all_subsets_predict_csr(int64_t n, const int64_t* indptr, const int32_t* indices,
                                 const double* data, const double* P, int k, int d,
                                 const double* lams, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int s = 0; s < k; ++s) {
            double a = 1.0;
            for (int64_t ii = indptr[i]; ii < indptr[i + 1]; ++ii)
                a *= 1 + data[ii] * P[(size_t)s * d + indices[ii]];
            acc += a * lams[s];
        }
        out[i] = acc;
    }
}

/* -------------------------------------------------------- ANOVA predict */

/* Row-wise ANOVA kernel K_A(x_i, p_s) of order `degree`, then K @ lams.
 * The reference computes this with closed forms / Newton identities on dense
 * (n, k) intermediates (sparsepoly/kernels.py:71-115,140-153); the value is the
 * elementary symmetric polynomial e_degree(p_s . x_i), evaluated here by the
 * same DP as pcd.py:23-30.  Used for the large-size cpu_baseline leg only; the
 * small-size oracle (oracle.py:anova_kernel) follows kernels.py formula-for-formula.
 * CSR input: indptr int64[n+1], indices int32 (column ids), data f64. */
void spo_anova_predict_csr(int64_t n, const int64_t* indptr, const int32_t* indices,
                           const double* data, const double* P, int k, int d,
                           const double* lams, int degree, double* out_accumulate) {
    double a[SPO_MAX_DEGREE + 1];
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int s = 0; s < k; ++s) {
            const double* ps = P + (size_t)s * d;
            a[0] = 1.0;
            for (int t = 1; t <= degree; ++t) a[t] = 0.0;
            for (int64_t ii = indptr[i]; ii < indptr[i + 1]; ++ii) {
                const double px = ps[indices[ii]] * data[ii];
                for (int t = degree; t >= 1; --t) a[t] += a[t - 1] * px;
            }
            acc += a[degree] * lams[s];
        }
        out_accumulate[i] += acc;
    }
}

/* ------------------------------------------------------------------ psgd */

/* sparsepoly/regularizer/utils.py:27-70 (prox_squaredl12).  The reference finds
 * the support of the prox of strength * (sum_i |p_i|)^2 with a randomised-pivot
 * selection; the support {i : |p_i| >= tau} and S = sum of its |p_i| are unique,
 * so this restatement finds them with a descending sort (deterministic; S differs
 * from the reference's only in summation order) and applies the same final lines
 * :69-70:  S /= 1 + 2*strength*theta;  soft_thresholding(p, 2*strength*S). */
static int cmp_desc(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x < y) - (x > y);
}
void spo_prox_squaredl12(double* p, int64_t n, int64_t stride, double strength) {
    double* a = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) a[i] = fabs(p[i * stride]);
    qsort(a, (size_t)n, sizeof(double), cmp_desc);
    double S = 0.0, run = 0.0;
    int64_t theta = 0;
    for (int64_t i = 0; i < n; ++i) {
        run += a[i];
        /* utils.py:54-55: pivot >= 2*strength*(S + S_Gi) / (1 + 2*strength*(theta + n_greater)) */
        const double cond = 2 * strength * run / (1.0 + 2.0 * strength * (double)(i + 1));
        if (a[i] >= cond) {
            S = run;
            theta = i + 1;
        } else {
            break;
        }
    }
    free(a);
    S /= 1.0 + 2.0 * strength * (double)theta;
    const double thr = 2 * strength * S;
    for (int64_t i = 0; i < n; ++i) { /* utils.py:8-9 soft_thresholding */
        const double v = p[i * stride];
        const double m = fabs(v) - thr;
        const double sg = (v > 0) - (v < 0);
        p[i * stride] = sg * (m > 0.0 ? m : 0.0);
    }
}

/* regularizer.prox(P, strength, degree) on P (d, k) row-major, with the default
 * `transpose` of each class (base.py:27-34 constructs them without arguments):
 *   l1         l1.py:50-51
 *   l21        l21.py:43-48   (rows with norm <= strength are left UNCHANGED:
 *                              their norm is replaced by inf, so the factor is 1)
 *   squaredl12 squaredl12.py:66-75 (transpose=True: one prox per component column)
 *   squaredl21 squaredl21.py:63-74 (row norms, prox of the norm vector, rescale) */
int spo_reg_prox(int kind, double* P, int d, int k, double strength) {
    if (kind == SPO_REG_L1) {
        for (size_t e = 0; e < (size_t)d * k; ++e) {
            const double v = P[e];
            const double m = fabs(v) - strength;
            const double sg = (v > 0) - (v < 0);
            P[e] = sg * (m > 0.0 ? m : 0.0);
        }
        return 0;
    }
    if (kind == SPO_REG_L21) {
        for (int j = 0; j < d; ++j) {
            double* pj = P + (size_t)j * k;
            double q = 0.0;
            for (int s = 0; s < k; ++s) q += fabs(pj[s]) * fabs(pj[s]);
            double nr = sqrt(q);
            if (nr <= strength) nr = INFINITY;
            const double f = 1.0 - strength / nr;
            for (int s = 0; s < k; ++s) pj[s] *= f;
        }
        return 0;
    }
    if (kind == SPO_REG_SQUAREDL12) {
        for (int s = 0; s < k; ++s) spo_prox_squaredl12(P + s, d, k, strength);
        return 0;
    }
    if (kind == SPO_REG_SQUAREDL21) {
        double* norms = (double*)malloc(sizeof(double) * (size_t)d);
        for (int j = 0; j < d; ++j) {
            double* pj = P + (size_t)j * k;
            double q = 0.0;
            for (int s = 0; s < k; ++s) q += fabs(pj[s]) * fabs(pj[s]);
            norms[j] = sqrt(q);
            if (norms[j] > 0)
                for (int s = 0; s < k; ++s) pj[s] /= norms[j];
        }
        spo_prox_squaredl12(norms, d, 1, strength);
        for (int j = 0; j < d; ++j) {
            double* pj = P + (size_t)j * k;
            for (int s = 0; s < k; ++s) pj[s] *= norms[j];
        }
        free(norms);
        return 0;
    }
    return -1; /* omegati / omegacs define no prox (psgd unsupported) */
}

/* sparsepoly/optimizer/psgd.py:9-22 */
void spo_psgd_get_eta(int learning_rate, double eta0, double alpha, double beta,
                      double power_t, int64_t it, double* eta_P, double* eta_w) {
    if (learning_rate == 0) {
        *eta_P = eta0;
        *eta_w = eta0;
    } else if (learning_rate == 1) {
        const double eta_it = eta0 * (double)it;
        *eta_P = eta0 / pow(1.0 + eta_it * beta, power_t);
        *eta_w = eta0 / pow(1.0 + eta_it * alpha, power_t);
    } else if (learning_rate == 2) {
        *eta_P = 1.0 / (beta * (double)it);
        *eta_w = 1.0 / (alpha * (double)it);
    } else {
        const double eta = eta0 / pow((double)it, power_t);
        *eta_P = eta;
        *eta_w = eta;
    }
}

/* sparsepoly/optimizer/psgd.py:125-199 (psgd_epoch) with its helpers :25-122.
 * CSR input (rows): indptr int64[n+1], indices int32, data f64.
 * P (n_orders, d, k) row-major (the reference's "copy for fast training",
 * sparse_factorization_machines.py:113); w (d); order o has degree `degree - o`.
 * Returns sum_loss; *it is advanced once per parameter update. */
double spo_psgd_epoch(double* P, double* w, const double* lams, int n_orders, int k,
                      int64_t n, int d, const int64_t* indptr, const int32_t* indices,
                      const double* data, const double* y, int loss, int reg, int degree,
                      double alpha, double beta, double gamma,
                      const int32_t* indices_samples, int fit_linear, double eta0,
                      int learning_rate, double power_t, int64_t batch_size, int64_t* it) {
    const size_t np_ = (size_t)n_orders * d * k;
    double* grad_P = (double*)calloc(np_, sizeof(double));
    double* grad_w = (double*)calloc((size_t)d, sizeof(double));
    double* A = (double*)malloc(sizeof(double) * (size_t)n_orders * (degree + 1) * k);
    double* dA = (double*)malloc(sizeof(double) * (size_t)degree * k);
    double sum_loss = 0.0;
    int64_t b = 0;
    for (int64_t ii = 0; ii < n; ++ii) {
        const int64_t i = indices_samples[ii];
        const int64_t lo = indptr[i], hi = indptr[i + 1];
        /* _pred :47-57 */
        double y_pred = 0.0;
        for (int64_t jj = lo; jj < hi; ++jj) y_pred += data[jj] * w[indices[jj]];
        for (int o = 0; o < n_orders; ++o) {
            const int deg = degree - o;
            double* Ao = A + (size_t)o * (degree + 1) * k;
            const double* Po = P + (size_t)o * d * k;
            for (int s = 0; s < k; ++s) Ao[s] = 1.0;
            for (int t = 1; t <= degree; ++t)
                for (int s = 0; s < k; ++s) Ao[(size_t)t * k + s] = 0.0;
            for (int64_t jj = lo; jj < hi; ++jj) { /* _anova :34-44 */
                const double* pj = Po + (size_t)indices[jj] * k;
                const double x = data[jj];
                for (int t = 0; t < deg; ++t)
                    for (int s = 0; s < k; ++s)
                        Ao[(size_t)(deg - t) * k + s] += Ao[(size_t)(deg - t - 1) * k + s] * x * pj[s];
            }
            double dot = 0.0;
            for (int s = 0; s < k; ++s) dot += lams[s] * Ao[(size_t)deg * k + s];
            y_pred += dot;
        }
        sum_loss += spo_loss(loss, y_pred, y[i]);
        /* _update_grads :60-91 */
        const double dL = spo_dloss(loss, y_pred, y[i]);
        if (fit_linear)
            for (int64_t jj = lo; jj < hi; ++jj) grad_w[indices[jj]] += dL * data[jj];
        for (int o = 0; o < n_orders; ++o) {
            const int deg = degree - o;
            const double* Ao = A + (size_t)o * (degree + 1) * k;
            const double* Po = P + (size_t)o * d * k;
            double* Go = grad_P + (size_t)o * d * k;
            for (int64_t jj = lo; jj < hi; ++jj) {
                const int j = indices[jj];
                const double x = data[jj];
                const double* pj = Po + (size_t)j * k;
                for (int s = 0; s < k; ++s) dA[s] = x; /* _grad_anova :25-31 */
                for (int t = 1; t < deg; ++t)
                    for (int s = 0; s < k; ++s)
                        dA[(size_t)t * k + s] =
                            x * (Ao[(size_t)t * k + s] - pj[s] * dA[(size_t)(t - 1) * k + s]);
                for (int s = 0; s < k; ++s)
                    Go[(size_t)j * k + s] += dL * lams[s] * dA[(size_t)(deg - 1) * k + s];
            }
        }
        b += 1;
        if (b == batch_size || ii == n - 1) { /* :177-198 */
            double eta_P, eta_w;
            spo_psgd_get_eta(learning_rate, eta0, alpha, beta, power_t, *it, &eta_P, &eta_w);
            /* _update_params :94-122 */
            if (fit_linear) {
                const double cw = eta_w / (double)b;
                for (int j = 0; j < d; ++j) {
                    grad_w[j] *= cw;
                    w[j] -= grad_w[j];
                    w[j] /= 1 + eta_w * alpha;
                }
            }
            const double cp = eta_P / (double)b;
            for (size_t e = 0; e < np_; ++e) {
                grad_P[e] *= cp;
                P[e] -= grad_P[e];
                P[e] /= 1.0 + eta_P * beta;
            }
            for (int o = 0; o < n_orders; ++o)
                spo_reg_prox(reg, P + (size_t)o * d * k, d, k, gamma * eta_P / (1 + eta_P * beta));
            memset(grad_P, 0, sizeof(double) * np_);
            memset(grad_w, 0, sizeof(double) * (size_t)d);
            b = 0;
            *it += 1;
        }
    }
    free(grad_P);
    free(grad_w);
    free(A);
    free(dA);
    return sum_loss;
}
