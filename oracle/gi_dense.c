/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/tph_ref.py header).  PARITY UNPINNED: quadprog is not
 * available in this image (a third-party dependency of trajectory_planning_helpers==0.76 [REF requirements.txt:3],
 * named at [REF Readme.md:40,44]; version hints 0.1.6 / 0.1.7), so this file restates the published algorithm it wraps:
 *
 *   D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly convex quadratic
 *   programs", Math. Programming 27 (1983) 1-33,
 *
 * in the form B. A. Turlach and A. Weingessel gave it in `solve.QP` / `qpgen2` (R package quadprog 1.5, which the Python
 * package compiles): the reference reaches it through tph.opt_min_curv [REF main_globaltraj.py:264-271, 344-350].
 *
 * The RULE SET is qpgen2's, statement by statement (round 6; rounds 1-5 followed the QuadProg++ variant with an exclusion
 * list and a "degenerate full step" test, which qpgen2 does not have):
 *
 *   set-up     Cholesky G = R'R (upper, LINPACK dpofa order), unconstrained minimiser by dposl, J = R^-1 by dpori.
 *   slacks     s_i = C_i'x - b_i for ALL m constraints every iteration; |s_i| < vsmall is set to 0; the slacks of the
 *              active constraints are set to 0 explicitly; equality rows (i < meq) enter as -|s_i| with their sign flipped
 *              when positive.
 *   entering   the constraint with the most negative s_i / |C_i| (strict <: the first one wins a tie).  No exclusion list.
 *   step       d = J'n+, z = J_2 d_2, r = R^-1 d_1.  t1 = min u_k / r_k over inequality rows with r_k > 0 (first minimum);
 *              z'z <= vsmall counts as z = 0: then "constraints are inconsistent" iff no r_k > 0, else a dual step and a drop.
 *              Otherwise t2 = -s_p / z'n+, step t = min(t1, t2) in primal and dual space; t2 <= t1 is a full step and the
 *              constraint is ADDED UNCONDITIONALLY (d_1 -> new column of R, Givens reflections (gc gs; gs -gc) on d_2 and the
 *              columns of J); t1 < t2 drops constraint it1 (reflections on rows of R / columns of J) and repeats with the
 *              same entering constraint after re-computing its slack.
 *   vsmall     the smallest 1e-60 * 2^k with 1 + 0.1 vsmall > 1 and 1 + 0.2 vsmall > 1 (1.4e-15), as qpgen2 computes it.
 *   results    x, the criterion value, lagr[m], iact[nact] (0-based here), iters = (main iterations = full steps + 1, drops).
 *
 * Consequence for exactly dependent row pairs that must both hold (the two box rows of a waypoint with w_r + w_l = w_veh,
 * which tph's `>` test lets through): after one of them has entered by a full step the other one's slack is a rounding
 * residue below vsmall, is set to 0 and never enters -- the solve ends at the optimum
 * (tests/test_emu_gi.py::test_zero_width_rows_both_paths_against_the_dense_oracle).
 *
 * Conventions are quadprog's:   minimise 1/2 x'Gx - a'x   s.t.  C'x >= b,
 * the first `meq` constraints being equalities; dense G, dense C (one COLUMN per constraint).
 *
 * Dense on purpose: this is the "what the reference's CPU path costs" stand-in.  O(n^3) set-up, then per iteration O(n m)
 * slacks + O(n^2) for d, z and the reflections.  Single thread.
 *
 * Storage: G row-major n x n (symmetric; only read); C column-major, constraint j at C[j*n .. j*n+n-1]; J column-major n x n;
 * R packed upper triangular by columns (column c at c (c + 1) / 2), as in qpgen2's work vector.
 * Not in qpgen2: the iteration cap (a safety net of this harness: GI_ITER_CAP has never been returned).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JM(r, c) J[(size_t)(c) * n + (r)]

enum { GI_OK = 0, GI_INFEASIBLE = 1, GI_NOT_PD = 2, GI_ITER_CAP = 3, GI_NOMEM = 4 };

static double dotn(const double* p, const double* q, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += p[i] * q[i];
    return s;
}

/* qpgen2's plane reflection (gc gs; gs -gc) that takes (p, q) to (temp, 0); returns 0 when there is nothing to do (gc == 1),
 * 1 for a plain swap (gc == 0), 2 otherwise */
static int reflection(double p, double q, double* gc, double* gs, double* temp)
{
    const double hi = fmax(fabs(p), fabs(q)), lo = fmin(fabs(p), fabs(q));
    *temp = copysign(hi * sqrt(1.0 + lo * lo / (hi * hi)), p);
    *gc = p / *temp;
    *gs = q / *temp;
    if (*gc == 1.0) return 0;
    return *gc == 0.0 ? 1 : 2;
}

static void reflect_cols(double* J, int n, int c0, int c1, double gc, double gs)
{
    double* p0 = J + (size_t)c0 * n;
    double* p1 = J + (size_t)c1 * n;
    const double nu = gs / (1.0 + gc);
    for (int k = 0; k < n; ++k) {
        const double t = gc * p0[k] + gs * p1[k];
        p1[k] = nu * (p0[k] + t) - p1[k];
        p0[k] = t;
    }
}

static void swap_cols(double* J, int n, int c0, int c1)
{
    double* p0 = J + (size_t)c0 * n;
    double* p1 = J + (size_t)c1 * n;
    for (int k = 0; k < n; ++k) { const double t = p0[k]; p0[k] = p1[k]; p1[k] = t; }
}

int gi_dense_solve(int n, int m, const double* G, const double* a, const double* C, const double* b, int meq,
                   double* x, double* lagr, int* iact_out, int* nact_out, int* iters, double* fval)
{
    const int rmax = n < m ? n : m;
    double* J = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* R = (double*)calloc((size_t)rmax * (rmax + 1) / 2 + 1, sizeof(double));
    double* wk = (double*)calloc((size_t)3 * n + 2 * (size_t)rmax + 2 + 2 * (size_t)m, sizeof(double));
    int* iact = (int*)calloc((size_t)rmax + 2, sizeof(int));
    signed char* sgn = (signed char*)malloc((size_t)(m > 0 ? m : 1));
    if (!J || !R || !wk || !iact || !sgn) {
        free(J); free(R); free(wk); free(iact); free(sgn);
        return GI_NOMEM;
    }
    double* d = wk;                  /* n */
    double* z = d + n;               /* n */
    double* np_ = z + n;             /* n: the entering normal with its sign */
    double* r = np_ + n;             /* rmax */
    double* u = r + rmax;            /* rmax + 2: multipliers of the active set, u[nact] = the entering constraint's */
    double* sv = u + rmax + 2;       /* m slacks */
    double* nb = sv + m;             /* m column norms */
    int status = GI_OK, nact = 0, nvl = -1, it1 = -1;
    int iter0 = 0, iter1 = 0;
    double crval = 0.0;
    memset(sgn, 1, (size_t)m);
    if (lagr) memset(lagr, 0, sizeof(double) * (size_t)m);

    double vsmall = 1e-60;
    {
        volatile double ta, tb;
        do { vsmall += vsmall; ta = 1.0 + 0.1 * vsmall; tb = 1.0 + 0.2 * vsmall; } while (ta <= 1.0 || tb <= 1.0);
    }

    /* --- dpofa: G = R'R, R upper triangular, kept column-major in J ----------------------------------------------------------- */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) JM(i, j) = G[(size_t)i * n + j];
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < j; ++k) {
            double t = JM(k, j) - dotn(&JM(0, k), &JM(0, j), k);
            t /= JM(k, k);
            JM(k, j) = t;
            s += t * t;
        }
        s = JM(j, j) - s;
        if (!(s > 0.0)) { status = GI_NOT_PD; goto done; }
        JM(j, j) = sqrt(s);
    }
    /* --- dposl: x = G^-1 a ------------------------------------------------------------------------------------------------------ */
    for (int k = 0; k < n; ++k) x[k] = a[k];
    for (int k = 0; k < n; ++k) x[k] = (x[k] - dotn(&JM(0, k), x, k)) / JM(k, k);
    for (int k = n - 1; k >= 0; --k) {
        x[k] /= JM(k, k);
        const double t = -x[k];
        for (int i = 0; i < k; ++i) x[i] += t * JM(i, k);
    }
    /* --- dpori: J = R^-1 in place (upper triangular), the strict lower triangle zeroed ----------------------------------------- */
    for (int k = 0; k < n; ++k) {
        JM(k, k) = 1.0 / JM(k, k);
        const double t = -JM(k, k);
        for (int i = 0; i < k; ++i) JM(i, k) *= t;
        for (int j = k + 1; j < n; ++j) {
            const double tj = JM(k, j);
            JM(k, j) = 0.0;
            for (int i = 0; i <= k; ++i) JM(i, j) += tj * JM(i, k);
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) JM(i, j) = 0.0;
    crval = -0.5 * dotn(a, x, n);
    for (int i = 0; i < m; ++i) nb[i] = sqrt(dotn(C + (size_t)i * n, C + (size_t)i * n, n));

    const long iter_cap = 40L * (n + m) + 1000;
    long it = 0;

    for (;;) { /* qpgen2 label 50: a new iteration */
        ++iter0;
        for (int i = 0; i < m; ++i) {
            double s = (double)sgn[i] * (dotn(C + (size_t)i * n, x, n) - b[i]);
            if (fabs(s) < vsmall) s = 0.0;
            if (i >= meq) sv[i] = s;
            else {
                sv[i] = -fabs(s);
                if (s > 0.0) sgn[i] = (signed char)-sgn[i];
            }
        }
        for (int k = 0; k < nact; ++k) sv[iact[k]] = 0.0;
        nvl = -1;
        {
            double temp = 0.0;
            for (int i = 0; i < m; ++i)
                if (sv[i] < temp * nb[i]) { nvl = i; temp = sv[i] / nb[i]; }
        }
        if (nvl < 0) break;
        for (int i = 0; i < n; ++i) np_[i] = (double)sgn[nvl] * C[(size_t)nvl * n + i];

        for (;;) { /* label 55: the step for constraint nvl; comes back here after a drop */
            if (++it > iter_cap) { status = GI_ITER_CAP; goto done; }
            for (int i = 0; i < n; ++i) d[i] = dotn(&JM(0, i), np_, n);
            memset(z, 0, sizeof(double) * (size_t)n);
            for (int j = nact; j < n; ++j) {
                const double dj = d[j];
                const double* jc = &JM(0, j);
                for (int i = 0; i < n; ++i) z[i] += jc[i] * dj;
            }
            int t1inf = 1;
            for (int i = nact - 1; i >= 0; --i) {
                double s = d[i];
                for (int j = i + 1; j < nact; ++j) s -= R[(size_t)j * (j + 1) / 2 + i] * r[j];
                s /= R[(size_t)i * (i + 1) / 2 + i];
                r[i] = s;
                if (iact[i] < meq || s <= 0.0) continue;
                t1inf = 0;
                it1 = i;
            }
            double t1 = 0.0;
            if (!t1inf) {
                t1 = u[it1] / r[it1];
                for (int i = 0; i < nact; ++i) {
                    if (iact[i] < meq || r[i] <= 0.0) continue;
                    const double temp = u[i] / r[i];
                    if (temp < t1) { t1 = temp; it1 = i; }
                }
            }
            int drop = 0;
            if (fabs(dotn(z, z, n)) <= vsmall) {
                /* no step in primal space makes the constraint feasible */
                if (t1inf) { status = GI_INFEASIBLE; goto done; }
                for (int i = 0; i < nact; ++i) u[i] -= t1 * r[i];
                u[nact] += t1;
                drop = 1;
            } else {
                const double znp = dotn(z, np_, n);
                double tt = -sv[nvl] / znp;
                int t2min = 1;
                if (!t1inf && t1 < tt) { tt = t1; t2min = 0; }
                for (int i = 0; i < n; ++i) x[i] += tt * z[i];
                crval += tt * znp * (0.5 * tt + u[nact]);
                for (int i = 0; i < nact; ++i) u[i] -= tt * r[i];
                u[nact] += tt;
                if (t2min) {
                    /* full step: nvl joins the active set, unconditionally */
                    double* rc = R + (size_t)nact * (nact + 1) / 2;
                    iact[nact] = nvl;
                    for (int i = 0; i < nact; ++i) rc[i] = d[i];
                    ++nact;
                    if (nact == n) rc[nact - 1] = d[n - 1];
                    else {
                        for (int i = n - 1; i >= nact; --i) {
                            if (d[i] == 0.0) continue;
                            double gc, gs, temp;
                            const int kind = reflection(d[i - 1], d[i], &gc, &gs, &temp);
                            if (kind == 0) continue;
                            if (kind == 1) { d[i - 1] = gs * temp; swap_cols(J, n, i - 1, i); }
                            else { d[i - 1] = temp; reflect_cols(J, n, i - 1, i, gc, gs); }
                        }
                        rc[nact - 1] = d[nact - 1];
                    }
                    break; /* -> label 50 */
                }
                /* partial step: the fit has moved, so has the violation of nvl */
                double s = (double)sgn[nvl] * (dotn(C + (size_t)nvl * n, x, n) - b[nvl]);
                if (nvl >= meq) sv[nvl] = s;
                else {
                    sv[nvl] = -fabs(s);
                    if (s > 0.0) {
                        sgn[nvl] = (signed char)-sgn[nvl];
                        for (int i = 0; i < n; ++i) np_[i] = -np_[i];
                    }
                }
                drop = 1;
            }
            if (drop) { /* label 700: drop the it1-th active constraint */
                for (; it1 < nact - 1; ++it1) {
                    /* column it1 + 1 of R: rows it1 and it1 + 1 meet on its diagonal */
                    size_t l = (size_t)(it1 + 1) * (it1 + 2) / 2; /* element (0, it1 + 1) */
                    size_t l1 = l + it1 + 1;                        /* element (it1 + 1, it1 + 1) */
                    if (R[l1] != 0.0) {
                        double gc, gs, temp;
                        const int kind = reflection(R[l1 - 1], R[l1], &gc, &gs, &temp);
                        if (kind == 1) {
                            for (int i = it1 + 1; i < nact; ++i) {
                                const double t = R[l1 - 1];
                                R[l1 - 1] = R[l1];
                                R[l1] = t;
                                l1 += (size_t)i + 1;
                            }
                            swap_cols(J, n, it1, it1 + 1);
                        } else if (kind == 2) {
                            const double nu = gs / (1.0 + gc);
                            for (int i = it1 + 1; i < nact; ++i) {
                                const double t = gc * R[l1 - 1] + gs * R[l1];
                                R[l1] = nu * (R[l1 - 1] + t) - R[l1];
                                R[l1 - 1] = t;
                                l1 += (size_t)i + 1;
                            }
                            reflect_cols(J, n, it1, it1 + 1, gc, gs);
                        }
                    }
                    /* the first it1 + 1 elements of column it1 + 1 become column it1 */
                    memmove(R + (size_t)it1 * (it1 + 1) / 2, R + l, sizeof(double) * (size_t)(it1 + 1));
                    u[it1] = u[it1 + 1];
                    iact[it1] = iact[it1 + 1];
                }
                u[nact - 1] = u[nact];
                u[nact] = 0.0;
                iact[nact - 1] = 0;
                --nact;
                ++iter1;
            }
        }
    }
    if (lagr) for (int k = 0; k < nact; ++k) lagr[iact[k]] = u[k];

done:
    if (iact_out) for (int k = 0; k < nact; ++k) iact_out[k] = iact[k];
    if (nact_out) *nact_out = nact;
    if (iters) { iters[0] = iter0; iters[1] = iter1; }
    if (fval) *fval = crval;
    free(J); free(R); free(wk); free(iact); free(sgn);
    return status;
}
