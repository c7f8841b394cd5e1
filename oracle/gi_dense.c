/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/tph_ref.py header).  PARITY UNPINNED: quadprog is not
 * available in this image, so this file restates the published algorithm it implements:
 *
 *   D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly convex quadratic
 *   programs", Math. Programming 27 (1983) 1-33   (the method behind quadprog.solve_qp, which the
 *   reference reaches through tph.opt_min_curv, call sites [REF main_globaltraj.py:264-271, 344-350]).
 *
 * Conventions are quadprog's:   minimise 1/2 x'Gx - a'x   s.t.  C'x >= b,
 * the first `meq` constraints being equalities; dense G, dense C (one COLUMN per constraint);
 * entering constraint = the most negative slack after normalising by the column norm.
 *
 * KNOWN LIMITATION (found in round 5, tests/test_emu_gi.py::test_zero_width_rows_both_paths_against_the_second_route): a pair of exactly
 * dependent rows that must BOTH be respected -- the two box rows of a waypoint with w_r + w_l = w_veh, lo = hi -- can end in the exclusion list
 * (`excl`, the degenerate-full-step rule below) while still violated, and the solver then stops at a non-optimal point (stationarity 2e-4 in that
 * test).  No fixture of tests/golden/ has such rows; the engine is checked against the least-squares second route there instead.
 *
 * Dense on purpose: this is the "what the reference's CPU path costs" stand-in.  O(n^3) set-up
 * (Cholesky, J = L^-T), then per iteration O(n*m) slacks + O(n^2) Givens updates.  Single thread.
 *
 * Storage: G row-major n x n (only read); C column-major, constraint j at C[j*n .. j*n+n-1].
 * J and R column-major n x n.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JM(r, c) J[(size_t)(c) * n + (r)]
#define RM(r, c) R[(size_t)(c) * n + (r)]

enum { GI_OK = 0, GI_INFEASIBLE = 1, GI_NOT_PD = 2, GI_ITER_CAP = 3, GI_NOMEM = 4 };

static double dotn(const double* p, const double* q, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += p[i] * q[i];
    return s;
}

/* rotate columns (c0, c1) of column-major matrix M (n rows):  [m0 m1] <- [m0 m1] * [[cs, -sn], [sn, cs]] */
static void rot_cols(double* M, int n, int c0, int c1, double cs, double sn)
{
    double* p0 = M + (size_t)c0 * n;
    double* p1 = M + (size_t)c1 * n;
    for (int k = 0; k < n; ++k) {
        const double t0 = p0[k], t1 = p1[k];
        p0[k] = cs * t0 + sn * t1;
        p1[k] = cs * t1 - sn * t0;
    }
}

int gi_dense_solve(int n, int m, const double* G, const double* a, const double* C, const double* b, int meq,
                   double* x, double* lagr, int* iact, int* nact_out, int* iters, double* fval)
{
    double* L = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* J = (double*)calloc((size_t)n * n, sizeof(double));
    double* R = (double*)calloc((size_t)n * n, sizeof(double));
    double* wk = (double*)malloc(sizeof(double) * ((size_t)6 * n + 2 * (size_t)m));
    int* A = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    char* active = (char*)calloc((size_t)m, 1);
    char* excl = (char*)calloc((size_t)m, 1);
    if (!L || !J || !R || !wk || !A || !active || !excl) {
        free(L); free(J); free(R); free(wk); free(A); free(active); free(excl);
        return GI_NOMEM;
    }
    double* d = wk;            /* n */
    double* z = d + n;         /* n */
    double* r = z + n;         /* n */
    double* u = r + n;         /* n+... multipliers of the active set */
    double* np_ = u + n;       /* n */
    double* tmp = np_ + n;     /* n */
    double* slack = tmp + n;   /* m */
    double* cnorm = slack + m; /* m */
    int status = GI_OK, q = 0, n_add = 0, n_drop = 0;
    double f = 0.0;

    /* --- Cholesky G = L L' (lower, row-major) ------------------------------------------------------------------ */
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = G[(size_t)i * n + j];
            const double* li = L + (size_t)i * n;
            const double* lj = L + (size_t)j * n;
            for (int k = 0; k < j; ++k) s -= li[k] * lj[k];
            if (i == j) {
                if (!(s > 0.0)) { status = GI_NOT_PD; goto done; }
                L[(size_t)i * n + i] = sqrt(s);
            } else {
                L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
            }
        }
    }
    /* --- J = L^-T: column c of J solves L' y = e_c (upper triangular result) ---------------------------------- */
    for (int c = 0; c < n; ++c) {
        JM(c, c) = 1.0 / L[(size_t)c * n + c];
        for (int i = c - 1; i >= 0; --i) {
            double s = 0.0;
            for (int k = i + 1; k <= c; ++k) s += L[(size_t)k * n + i] * JM(k, c);
            JM(i, c) = -s / L[(size_t)i * n + i];
        }
    }
    /* --- unconstrained minimiser x = G^-1 a ------------------------------------------------------------------- */
    for (int i = 0; i < n; ++i) {
        double s = a[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * tmp[k];
        tmp[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = tmp[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * x[k];
        x[i] = s / L[(size_t)i * n + i];
    }
    f = -0.5 * dotn(a, x, n);
    for (int j = 0; j < m; ++j) cnorm[j] = sqrt(dotn(C + (size_t)j * n, C + (size_t)j * n, n));
    double rnorm = 1.0;
    const long iter_cap = 40L * (n + m) + 1000;
    long it = 0;

    for (;;) {
        /* step 1: choose the violated constraint */
        int p = -1;
        double best = 0.0;
        for (int j = 0; j < m; ++j) {
            if (active[j]) { slack[j] = 0.0; continue; }
            double s = dotn(C + (size_t)j * n, x, n) - b[j];
            if (j < meq) s = -fabs(s); /* equalities: oracle use is meq = 0; kept for interface completeness */
            slack[j] = s;
            if (excl[j] || cnorm[j] == 0.0) continue;
            if (s < best * cnorm[j]) { best = s / cnorm[j]; p = j; }
        }
        if (p < 0) break;
        memcpy(np_, C + (size_t)p * n, sizeof(double) * n);
        double up = 0.0;
        double sp = slack[p];

        for (;;) { /* step 2: (partial) steps until p becomes active or is found inconsistent */
            if (++it > iter_cap) { status = GI_ITER_CAP; goto done; }
            for (int i = 0; i < n; ++i) d[i] = dotn(J + (size_t)i * n, np_, n);
            memset(z, 0, sizeof(double) * n);
            for (int k = q; k < n; ++k) {
                const double dk = d[k];
                const double* jc = J + (size_t)k * n;
                for (int i = 0; i < n; ++i) z[i] += jc[i] * dk;
            }
            for (int i = q - 1; i >= 0; --i) {
                double s = d[i];
                for (int k = i + 1; k < q; ++k) s -= RM(i, k) * r[k];
                r[i] = s / RM(i, i);
            }
            /* step lengths */
            int l = -1;
            double t1 = INFINITY;
            for (int k = 0; k < q; ++k) {
                if (A[k] < meq) continue;
                if (r[k] > 0.0) {
                    const double cand = u[k] / r[k];
                    if (cand < t1) { t1 = cand; l = k; }
                }
            }
            int zzero = 1;
            for (int i = 0; i < n; ++i) if (fabs(z[i]) > DBL_MIN) { zzero = 0; break; }
            double t2 = INFINITY;
            double znp = 0.0;
            if (!zzero) {
                znp = dotn(z, np_, n);
                if (znp > 0.0) t2 = -sp / znp;
            }
            const double t = t1 < t2 ? t1 : t2;
            if (isinf(t)) { status = GI_INFEASIBLE; goto done; }

            if (isinf(t2)) { /* dual step only, drop l */
                for (int k = 0; k < q; ++k) u[k] -= t * r[k];
                up += t;
            } else {
                for (int i = 0; i < n; ++i) x[i] += t * z[i];
                f += t * znp * (0.5 * t + up);
                for (int k = 0; k < q; ++k) u[k] -= t * r[k];
                up += t;
                if (t2 <= t1) { /* full step: p joins the active set */
                    int degenerate = 0;
                    for (int j = n - 1; j > q; --j) {
                        const double hh = hypot(d[j - 1], d[j]);
                        if (hh == 0.0) continue;
                        const double cs = d[j - 1] / hh, sn = d[j] / hh;
                        d[j - 1] = hh;
                        d[j] = 0.0;
                        rot_cols(J, n, j - 1, j, cs, sn);
                    }
                    if (fabs(d[q]) <= DBL_EPSILON * rnorm) degenerate = 1;
                    if (degenerate) { excl[p] = 1; break; }
                    for (int i = 0; i <= q; ++i) RM(i, q) = d[i];
                    if (fabs(d[q]) > rnorm) rnorm = fabs(d[q]);
                    A[q] = p;
                    u[q] = up;
                    active[p] = 1;
                    ++q;
                    ++n_add;
                    memset(excl, 0, (size_t)m);
                    break;
                }
            }
            /* partial step: drop active constraint at position l */
            {
                active[A[l]] = 0;
                for (int k = l + 1; k < q; ++k) {
                    memcpy(R + (size_t)(k - 1) * n, R + (size_t)k * n, sizeof(double) * (size_t)(k + 1));
                    A[k - 1] = A[k];
                    u[k - 1] = u[k];
                }
                --q;
                ++n_drop;
                for (int j = l; j < q; ++j) { /* restore triangularity: rotate rows (j, j+1) of R, columns of J */
                    const double hh = hypot(RM(j, j), RM(j + 1, j));
                    if (hh == 0.0) continue;
                    const double cs = RM(j, j) / hh, sn = RM(j + 1, j) / hh;
                    for (int k = j; k < q; ++k) {
                        const double t0 = RM(j, k), t1_ = RM(j + 1, k);
                        RM(j, k) = cs * t0 + sn * t1_;
                        RM(j + 1, k) = cs * t1_ - sn * t0;
                    }
                    rot_cols(J, n, j, j + 1, cs, sn);
                }
                sp = dotn(np_, x, n) - b[p];
            }
        }
    }

done:
    if (lagr) {
        memset(lagr, 0, sizeof(double) * (size_t)m);
        for (int k = 0; k < q; ++k) lagr[A[k]] = u[k];
    }
    if (iact) for (int k = 0; k < q; ++k) iact[k] = A[k];
    if (nact_out) *nact_out = q;
    if (iters) { iters[0] = n_add; iters[1] = n_drop; }
    if (fval) *fval = f;
    free(L); free(J); free(R); free(wk); free(A); free(active); free(excl);
    return status;
}
