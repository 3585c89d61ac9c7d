/* oracle/lsap.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * Plain-C restatement of the rectangular linear-sum-assignment solver the reference calls at
 * A2/models/matcher.py:6,246 (`scipy.optimize.linear_sum_assignment`; third-party, scipy 1.15.3 in this
 * image, unpinned in the reference's requirements).  Algorithm: D. F. Crouse, "On implementing 2D
 * rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- shortest augmenting paths with dual
 * variables, float64 throughout, the cost matrix transposed when it has more rows than columns, row
 * indices returned ascending.  Pinned against scipy itself by tests/test_lsap_oracle.py.
 *
 * int cdetr_oracle_lsap(nr, nc, cost[nr*nc] row-major double, row_ind[min], col_ind[min])
 *   returns min(nr,nc) on success, -1 infeasible, -2 invalid entry (NaN / -inf).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augmenting_path(int64_t nc, const double *cost, const double *u, const double *v, int64_t *path,
                               const int64_t *row4col, double *spc, int64_t i, char *SR, char *SC,
                               int64_t *remaining, double *p_min) {
    double min_val = 0.0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1; /* filled in reverse order */
    memset(SC, 0, (size_t)nc);
    for (int64_t j = 0; j < nc; ++j) spc[j] = INFINITY;
    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; ++it) {
            int64_t j = remaining[it];
            double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) {
                path[j] = i;
                spc[j] = r;
            }
            /* among equal minima prefer a column that is still unassigned (a new sink) */
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

static int cmp_pair(const void *a, const void *b) {
    const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
    return (x[0] > y[0]) - (x[0] < y[0]);
}

int cdetr_oracle_lsap(int64_t nr, int64_t nc, const double *cost_in, int64_t *row_ind, int64_t *col_ind) {
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
    if (transpose) {
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
    } else {
        memcpy(cost, cost_in, sizeof(double) * (size_t)(nr * nc));
    }
    for (int64_t k = 0; k < nr * nc; ++k)
        if (isnan(cost[k]) || cost[k] == -INFINITY) { free(cost); return -2; }

    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    char *SR = (char *)malloc((size_t)nr), *SC = (char *)malloc((size_t)nc);
    for (int64_t j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; ++i) col4row[i] = -1;
    int rc = 0;
    for (int64_t cur = 0; cur < nr; ++cur) {
        double min_val;
        memset(SR, 0, (size_t)nr);
        int64_t sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) { rc = -1; break; }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        for (;;) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) { /* rows of the transposed problem are original columns: sort by original row */
            int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)nr);
            for (int64_t i = 0; i < nr; ++i) { pairs[2 * i] = col4row[i]; pairs[2 * i + 1] = i; }
            qsort(pairs, (size_t)nr, 2 * sizeof(int64_t), cmp_pair);
            for (int64_t i = 0; i < nr; ++i) { row_ind[i] = pairs[2 * i]; col_ind[i] = pairs[2 * i + 1]; }
            free(pairs);
        } else {
            for (int64_t i = 0; i < nr; ++i) { row_ind[i] = i; col_ind[i] = col4row[i]; }
        }
        rc = (int)nr;
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining);
    free(SR); free(SC);
    return rc;
}
