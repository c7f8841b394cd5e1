/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see the header of oracle/tph_ref.py).  PARITY UNPINNED BY THE REFERENCE.
 *
 * "CPU-B": the structure-exploiting CPU solver of BASELINE.md section 3 / SURVEY.md section 8d -- what a careful CPU
 * implementation of the reference's hot path costs once the gratuitous O((4N)^3) work of the dense formulation is removed.
 * Same QP as tph.opt_min_curv hands to quadprog (call sites [REF main_globaltraj.py:264-271, 344-350]; maths SURVEY.md
 * App. A), scalar fp64 C, one problem per thread (OpenMP over the batch):
 *
 *   assembly   closed cubic spline = cyclic tridiagonal system in the c-coefficients; rows of its inverse by one cyclic
 *              Thomas solve each (no truncation inside the solve), band of +-(BE+2) entries kept  ->  E_kappa band (half
 *              width BE = 36: wider than the GPU engine's 32, so that a band-truncation effect would show up as a difference),
 *              k_ref, H = E'E (cyclic band, half width 2 BE), f = 2 E'k_ref
 *   QP         Mehrotra predictor-corrector interior point on the box rows (one bordered-band Cholesky + two solves per
 *              iteration) -> active-set identification -> block principal pivoting with single-pivot backup on the vertex ->
 *              fp64 residual refinement through E.  The curvature rows are CHECKED at the result (status 6 if one is
 *              violated: this baseline does not carry them; on every workload it is used for they are inactive).
 *   post-check opt_min_curv's curvature error.
 *
 * Used by bench.py's cpu_baseline leg ("best-effort CPU", all host cores) and by tests as a third route to alpha at
 * N = 2000 for many problems (tests/test_oracle.py pins it against the dense-faithful oracle first).  Never shipped.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BE 36
#define EW (2 * BE + 1)
#define BH (2 * BE)
#define GWD (BE + 2)
#define F_SCALE 2.0

enum { BQ_OK = 0, BQ_INFEASIBLE = 1, BQ_NOT_PD = 2, BQ_ITER_CAP = 3, BQ_BAD_INPUT = 4, BQ_KAPPA = 6, BQ_NOMEM = 7 };

typedef struct {
    int n, ni, p;
    double *Eb;   /* [n][EW]   E[i, (i + o) mod n] at o + BE */
    double *Db;   /* [n][EW]   D (x'' = D x) the same way */
    double *Hb;   /* [n][BH+1] H[i, (i + k) mod n], k = 0..BH */
    double *kref, *f, *xp, *yp, *xpp, *ypp, *lo, *hi;
    /* factor of the masked matrix: interior band L (row i: L[i, i-k] at [k], k = 0..BH), W = L^-1 C (ni x p), L_S (p x p) */
    double *Lb, *W, *S;
} Bq;

static inline int cyc(int i, int n) { i %= n; return i < 0 ? i + n : i; }

/* cyclic tridiagonal  sub[m] x[m-1] + dg[m] x[m] + sup[m] x[m+1] = r[m]  factored once (Thomas + Sherman-Morrison) */
typedef struct { int n; double *cp, *dinv, *sub, *z; double gam, vz; double sup_last, sub0; } Ctri;

static int ctri_factor(Ctri* T, int n, const double* sub, const double* dg, const double* sup)
{
    T->n = n;
    T->cp = (double*)malloc(sizeof(double) * n * 4);
    if (!T->cp) return 1;
    T->dinv = T->cp + n; T->sub = T->dinv + n; T->z = T->sub + n;
    /* A' = A - u v',  u = (gam, 0, ..., 0, sup[n-1])',  v = (1, 0, ..., 0, sub[0] / gam)' */
    const double gam = -dg[0];
    T->gam = gam; T->sup_last = sup[n - 1]; T->sub0 = sub[0];
    double* dd = (double*)malloc(sizeof(double) * n);
    if (!dd) return 1;
    for (int i = 0; i < n; ++i) dd[i] = dg[i];
    dd[0] -= gam;
    dd[n - 1] -= sup[n - 1] * sub[0] / gam;
    /* Thomas factorisation of the plain tridiagonal */
    double piv = dd[0];
    T->dinv[0] = 1.0 / piv;
    T->sub[0] = 0.0;
    for (int i = 0; i < n - 1; ++i) {
        T->cp[i] = sup[i] * T->dinv[i];
        piv = dd[i + 1] - sub[i + 1] * T->cp[i];
        T->dinv[i + 1] = 1.0 / piv;
        T->sub[i + 1] = sub[i + 1];
    }
    free(dd);
    /* z = A'^-1 u */
    double* z = T->z;
    for (int i = 0; i < n; ++i) z[i] = 0.0;
    z[0] = gam; z[n - 1] = sup[n - 1];
    z[0] *= T->dinv[0];
    for (int i = 1; i < n; ++i) z[i] = (z[i] - T->sub[i] * z[i - 1]) * T->dinv[i];
    for (int i = n - 2; i >= 0; --i) z[i] -= T->cp[i] * z[i + 1];
    T->vz = 1.0 + z[0] + sub[0] / gam * z[n - 1];
    return 0;
}

static void ctri_solve(const Ctri* T, double* r)
{
    const int n = T->n;
    r[0] *= T->dinv[0];
    for (int i = 1; i < n; ++i) r[i] = (r[i] - T->sub[i] * r[i - 1]) * T->dinv[i];
    for (int i = n - 2; i >= 0; --i) r[i] -= T->cp[i] * r[i + 1];
    const double fac = (r[0] + T->sub0 / T->gam * r[n - 1]) / T->vz;
    for (int i = 0; i < n; ++i) r[i] -= fac * T->z[i];
}

/* y = E x  /  y = E' x  (cyclic band products) */
static void e_mul(const Bq* q, const double* x, double* y)
{
    const int n = q->n;
    for (int i = 0; i < n; ++i) {
        const double* e = q->Eb + (size_t)i * EW;
        double acc = 0.0;
        int j = cyc(i - BE, n);
        for (int o = 0; o < EW; ++o) { acc += e[o] * x[j]; j = j + 1 == n ? 0 : j + 1; }
        y[i] = acc;
    }
}
static void et_mul(const Bq* q, const double* x, double* y)
{
    const int n = q->n;
    for (int j = 0; j < n; ++j) y[j] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* e = q->Eb + (size_t)i * EW;
        const double xi = x[i];
        int j = cyc(i - BE, n);
        for (int o = 0; o < EW; ++o) { y[j] += e[o] * xi; j = j + 1 == n ? 0 : j + 1; }
    }
}
/* g = E'(E x + F_SCALE k_ref) */
static void gradient(const Bq* q, const double* x, double* tmp, double* g)
{
    e_mul(q, x, tmp);
    for (int i = 0; i < q->n; ++i) tmp[i] += F_SCALE * q->kref[i];
    et_mul(q, tmp, g);
}

static double hget(const Bq* q, int i, int j)
{
    const int n = q->n;
    int d = j - i;
    if (d < 0) d += n;
    if (d <= BH) return q->Hb[(size_t)i * (BH + 1) + d];
    if (n - d <= BH) return q->Hb[(size_t)j * (BH + 1) + (n - d)];
    return 0.0;
}

/* Cholesky of  M = H + diag(sig)  with rows / columns of masked variables replaced by identity; bordered band:
 * interior ni = n - p unknowns in a plain band of half width BH, the last p = BH unknowns a dense border. */
static int factor(Bq* q, const double* sig, const signed char* mk)
{
    const int n = q->n, ni = q->ni, p = q->p, b = BH;
    double* Lb = q->Lb; double* W = q->W; double* S = q->S;
    /* interior band, lower: Lb[i][k] = M[i, i-k] */
    for (int i = 0; i < ni; ++i) {
        double* row = Lb + (size_t)i * (b + 1);
        for (int k = 0; k <= b; ++k) {
            const int j = i - k;
            double v = 0.0;
            if (j >= 0) {
                const int pin = mk && (mk[i] || mk[j]);
                v = pin ? (k == 0 ? 1.0 : 0.0) : q->Hb[(size_t)j * (b + 1) + k];
                if (k == 0 && !pin && sig) v += sig[i];
            }
            row[k] = v;
        }
    }
    /* band Cholesky (row-oriented): L[i,j] = (M[i,j] - sum_{m} L[i,m] L[j,m]) / L[j,j] */
    for (int i = 0; i < ni; ++i) {
        double* ri = Lb + (size_t)i * (b + 1);
        const int j0 = i - b > 0 ? i - b : 0;
        for (int j = j0; j <= i; ++j) {
            const double* rj = Lb + (size_t)j * (b + 1);
            double s = ri[i - j];
            const int m0 = j0 > j - b ? j0 : (j - b > 0 ? j - b : 0);
            for (int m = m0; m < j; ++m) s -= ri[i - m] * rj[j - m];
            if (j < i) ri[i - j] = s / rj[0];
            else {
                if (!(s > 0.0)) return BQ_NOT_PD;
                ri[0] = sqrt(s);
            }
        }
    }
    /* W = L^-1 C,  C[i][jj] = M[i, ni + jj] */
    for (int i = 0; i < ni; ++i) {
        double* wi = W + (size_t)i * p;
        const double* ri = Lb + (size_t)i * (b + 1);
        const int near = (i < b) || (i >= ni - b);
        for (int jj = 0; jj < p; ++jj) {
            double v = 0.0;
            if (near) {
                const int j = ni + jj;
                if (!(mk && (mk[i] || mk[j]))) v = hget(q, i, j);
            }
            wi[jj] = v;
        }
        const int m0 = i - b > 0 ? i - b : 0;
        for (int m = m0; m < i; ++m) {
            const double l = ri[i - m];
            if (l == 0.0) continue;
            const double* wm = W + (size_t)m * p;
            for (int jj = 0; jj < p; ++jj) wi[jj] -= l * wm[jj];
        }
        const double inv = 1.0 / ri[0];
        for (int jj = 0; jj < p; ++jj) wi[jj] *= inv;
    }
    /* S = D - W'W, then its Cholesky (dense, lower) */
    for (int a = 0; a < p; ++a)
        for (int c = 0; c <= a; ++c) {
            const int i = ni + a, j = ni + c;
            double v;
            if (mk && (mk[i] || mk[j])) v = (a == c) ? 1.0 : 0.0;
            else { v = hget(q, j, i); if (a == c && sig) v += sig[i]; }
            S[(size_t)a * p + c] = v;
        }
    for (int i = 0; i < ni; ++i) {
        const double* wi = W + (size_t)i * p;
        for (int a = 0; a < p; ++a) {
            const double wa = wi[a];
            if (wa == 0.0) continue;
            double* sa = S + (size_t)a * p;
            for (int c = 0; c <= a; ++c) sa[c] -= wa * wi[c];
        }
    }
    for (int a = 0; a < p; ++a) {
        double* sa = S + (size_t)a * p;
        for (int c = 0; c <= a; ++c) {
            const double* sc = S + (size_t)c * p;
            double s = sa[c];
            for (int m = 0; m < c; ++m) s -= sa[m] * sc[m];
            if (c < a) sa[c] = s / sc[c];
            else { if (!(s > 0.0)) return BQ_NOT_PD; sa[a] = sqrt(s); }
        }
    }
    (void)n;
    return BQ_OK;
}

static void solve(const Bq* q, double* v)
{
    const int ni = q->ni, p = q->p, b = BH;
    const double* Lb = q->Lb; const double* W = q->W; const double* S = q->S;
    double* vd = v + ni;
    for (int i = 0; i < ni; ++i) {                       /* L y = v_B */
        const double* ri = Lb + (size_t)i * (b + 1);
        double s = v[i];
        const int m0 = i - b > 0 ? i - b : 0;
        for (int m = m0; m < i; ++m) s -= ri[i - m] * v[m];
        v[i] = s / ri[0];
    }
    for (int i = 0; i < ni; ++i) {                       /* t = v_D - W'y */
        const double* wi = W + (size_t)i * p;
        const double yi = v[i];
        for (int jj = 0; jj < p; ++jj) vd[jj] -= wi[jj] * yi;
    }
    for (int a = 0; a < p; ++a) {                        /* L_S L_S' x_D = t */
        double s = vd[a];
        for (int m = 0; m < a; ++m) s -= S[(size_t)a * p + m] * vd[m];
        vd[a] = s / S[(size_t)a * p + a];
    }
    for (int a = p - 1; a >= 0; --a) {
        double s = vd[a];
        for (int m = a + 1; m < p; ++m) s -= S[(size_t)m * p + a] * vd[m];
        vd[a] = s / S[(size_t)a * p + a];
    }
    for (int i = ni - 1; i >= 0; --i) {                  /* L' x_B = y - W x_D */
        const double* wi = W + (size_t)i * p;
        double s = v[i];
        for (int jj = 0; jj < p; ++jj) s -= wi[jj] * vd[jj];
        const int m1 = i + b < ni - 1 ? i + b : ni - 1;
        for (int m = i + 1; m <= m1; ++m) s -= Lb[(size_t)m * (b + 1) + (m - i)] * v[m];
        v[i] = s / Lb[(size_t)i * (b + 1)];
    }
}

static int assemble(Bq* q, const double* ref, const double* nv, const double* sc, double w_veh)
{
    const int n = q->n;
    double* S = (double*)malloc(sizeof(double) * n * 8);
    double* G = (double*)malloc(sizeof(double) * (size_t)n * (2 * GWD + 1));
    if (!S || !G) { free(S); free(G); return BQ_NOMEM; }
    double *sub = S + n, *dg = sub + n, *sup = dg + n, *rx = sup + n, *ry = rx + n, *cp = ry + n, *col = cp + n;
    int bad = 0, inf = 0;
    for (int i = 0; i < n; ++i) {
        S[i] = sc ? sc[i] : 1.0;
        const double lo = -(ref[4 * i + 3] - 0.5 * w_veh), hi = ref[4 * i + 2] - 0.5 * w_veh;
        if (!(isfinite(ref[4 * i]) && isfinite(ref[4 * i + 1]) && isfinite(lo) && isfinite(hi) && S[i] > 0.0)) bad = 1;
        if (hi < lo) inf = 1;
        q->lo[i] = lo; q->hi[i] = hi;
    }
    if (bad || inf) { free(S); free(G); return bad ? BQ_BAD_INPUT : BQ_INFEASIBLE; }
    /* centre m:  c_{m-1} + (2 s_{m-1}^2 + 2 s_{m-1}) c_m + s_{m-1} s_m^2 c_{m+1} = 3 (s_{m-1} D_m - D_{m-1}),  D_m = p_{m+1} - p_m */
    for (int m = 0; m < n; ++m) {
        const int mm = m == 0 ? n - 1 : m - 1, mp = m + 1 == n ? 0 : m + 1;
        const double s1 = S[mm];
        sub[m] = 1.0;
        dg[m] = 2.0 * s1 * s1 + 2.0 * s1;
        sup[m] = s1 * S[m] * S[m];
        rx[m] = 3.0 * (s1 * (ref[4 * mp] - ref[4 * m]) - (ref[4 * m] - ref[4 * mm]));
        ry[m] = 3.0 * (s1 * (ref[4 * mp + 1] - ref[4 * m + 1]) - (ref[4 * m + 1] - ref[4 * mm + 1]));
    }
    Ctri T;
    if (ctri_factor(&T, n, sub, dg, sup)) { free(S); free(G); return BQ_NOMEM; }
    /* band of T^-1 by columns: column j = T^-1 e_j; G[i][GWD + (j - i)] for |j - i| <= GWD (cyclic) */
    memset(G, 0, sizeof(double) * (size_t)n * (2 * GWD + 1));
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) col[i] = 0.0;
        col[j] = 1.0;
        ctri_solve(&T, col);
        for (int o = -GWD; o <= GWD; ++o) {             /* row i = j - o holds column j at offset +o */
            const int i = cyc(j - o, n);
            G[(size_t)i * (2 * GWD + 1) + GWD + o] = col[i];
        }
    }
    ctri_solve(&T, rx);                                  /* c-coefficients */
    ctri_solve(&T, ry);
    free(T.cp);
    for (int i = 0; i < n; ++i) { q->xpp[i] = 2.0 * rx[i]; q->ypp[i] = 2.0 * ry[i]; }
    for (int i = 0; i < n; ++i) {
        const int ip = i + 1 == n ? 0 : i + 1;
        const double s2 = S[i] * S[i];
        const double xp = (ref[4 * ip] - ref[4 * i]) - (q->xpp[i] + 0.5 * s2 * q->xpp[ip]) / 3.0;
        const double yp = (ref[4 * ip + 1] - ref[4 * i + 1]) - (q->ypp[i] + 0.5 * s2 * q->ypp[ip]) / 3.0;
        const double den = pow(xp * xp + yp * yp, 1.5);
        q->xp[i] = xp; q->yp[i] = yp;
        cp[i] = den != 0.0 ? 1.0 / den : 0.0;
        q->kref[i] = cp[i] * (xp * q->ypp[i] - yp * q->xpp[i]);
    }
    /* D[i, j] = 6 (G[i, j+1] - (1 + s_{j-1}) G[i, j] + s_{j-2} G[i, j-1]);  E[i, j] = D[i, j] cp_i (x'_i n_y,j - y'_i n_x,j) */
    for (int i = 0; i < n; ++i) {
        const double* g = G + (size_t)i * (2 * GWD + 1) + GWD;
        for (int o = -BE; o <= BE; ++o) {
            const int j = cyc(i + o, n), j1 = cyc(j - 1, n), j2 = cyc(j - 2, n);
            const double dv = 6.0 * (g[o + 1] - (1.0 + S[j1]) * g[o] + S[j2] * g[o - 1]);
            q->Db[(size_t)i * EW + o + BE] = dv;
            q->Eb[(size_t)i * EW + o + BE] = dv * cp[i] * (q->xp[i] * nv[2 * j + 1] - q->yp[i] * nv[2 * j]);
        }
    }
    /* H = E'E (upper cyclic band), f = F_SCALE E'k_ref */
    memset(q->Hb, 0, sizeof(double) * (size_t)n * (BH + 1));
    for (int i = 0; i < n; ++i) {
        const double* e = q->Eb + (size_t)i * EW;
        for (int oa = 0; oa < EW; ++oa) {
            const int ja = cyc(i - BE + oa, n);
            double* h = q->Hb + (size_t)ja * (BH + 1);
            const double ea = e[oa];
            for (int ob = oa; ob < EW; ++ob) h[ob - oa] += ea * e[ob];
        }
    }
    et_mul(q, q->kref, q->f);
    for (int i = 0; i < n; ++i) q->f[i] *= F_SCALE;
    free(S); free(G);
    return BQ_OK;
}

static int bq_alloc(Bq* q, int n)
{
    memset(q, 0, sizeof(*q));
    q->n = n; q->p = BH; q->ni = n - BH;
    const size_t nd = (size_t)n * (2 * EW + (BH + 1) + 8) + (size_t)q->ni * (BH + 1) + (size_t)q->ni * q->p + (size_t)q->p * q->p;
    double* m = (double*)malloc(sizeof(double) * nd);
    if (!m) return 1;
    q->Eb = m; m += (size_t)n * EW;
    q->Db = m; m += (size_t)n * EW;
    q->Hb = m; m += (size_t)n * (BH + 1);
    q->kref = m; m += n; q->f = m; m += n; q->xp = m; m += n; q->yp = m; m += n;
    q->xpp = m; m += n; q->ypp = m; m += n; q->lo = m; m += n; q->hi = m; m += n;
    q->Lb = m; m += (size_t)q->ni * (BH + 1);
    q->W = m; m += (size_t)q->ni * q->p;
    q->S = m;
    return 0;
}

/* One problem.  iters[0] interior-point iterations, iters[1] active-set rounds.  Returns a BQ_* status. */
int bqp_solve(int n, const double* ref, const double* nv, const double* sc, double kappa_bound, double w_veh, double* alpha,
              double* curv_err, int* iters)
{
    iters[0] = iters[1] = 0;
    *curv_err = 0.0;
    if (n < 4 * BH + 4) return BQ_BAD_INPUT;           /* the bordered band needs a ring much longer than the band */
    Bq q;
    if (bq_alloc(&q, n)) return BQ_NOMEM;
    int st = assemble(&q, ref, nv, sc, w_veh);
    double* wk = (double*)malloc(sizeof(double) * (size_t)n * 16);
    signed char* mk = (signed char*)calloc((size_t)n, 1);
    if (!wk || !mk) st = BQ_NOMEM;
    if (st != BQ_OK) { free(q.Eb); free(wk); free(mk); for (int i = 0; i < n; ++i) alpha[i] = 0.0; return st; }
    double *x = wk, *g = x + n, *zl = g + n, *zu = zl + n, *sig = zu + n, *rhs = sig + n, *dxa = rhs + n, *tmp = dxa + n,
           *dzl = tmp + n, *dzu = dzl + n, *ind = dzu + n, *xa = ind + n, *pv = xa + n, *t2 = pv + n, *xs = t2 + n;
    const double *lo = q.lo, *hi = q.hi;
    double wsum = 0.0, fsc = 0.0;
    int nfree = 0, any_fixed = 0;
    for (int i = 0; i < n; ++i) {
        const double w = hi[i] - lo[i];
        mk[i] = !(w > 1e-12) ? 2 : 0;
        if (mk[i]) any_fixed = 1; else { wsum += w; ++nfree; }
        x[i] = 0.5 * (lo[i] + hi[i]);
        fsc = fmax(fsc, fabs(q.f[i]));
    }
    const double wmean = nfree ? wsum / nfree : 1.0;
    gradient(&q, x, tmp, g);
    double zscale = 0.0;
    for (int i = 0; i < n; ++i) if (!mk[i]) zscale = fmax(zscale, fabs(g[i]));
    if (!(zscale > 0.0)) zscale = fsc > 0.0 ? fsc : 1.0;
    for (int i = 0; i < n; ++i) zl[i] = zu[i] = mk[i] ? 0.0 : zscale;

    /* ---- interior point ---- */
    const double TOL = 1e-10;
    double last_step = 0.0;
    int converged = nfree == 0;
    for (int it = 1; it <= 80 && !converged; ++it) {
        double mu = 0.0, rdm = 0.0;
        for (int i = 0; i < n; ++i) {
            if (mk[i]) { sig[i] = 0.0; continue; }
            const double sl = x[i] - lo[i], su = hi[i] - x[i];
            mu += sl * zl[i] + su * zu[i];
            rdm = fmax(rdm, fabs(g[i] - zl[i] + zu[i]));
            sig[i] = zl[i] / sl + zu[i] / su;
        }
        mu /= 2.0 * nfree;
        if (mu < TOL * zscale * wmean && rdm < TOL * zscale) { converged = 1; break; }
        iters[0] = it;
        st = factor(&q, sig, any_fixed ? mk : NULL);
        if (st != BQ_OK) break;
        for (int i = 0; i < n; ++i) rhs[i] = mk[i] ? 0.0 : -g[i];
        solve(&q, rhs);
        double ap = 1.0, ad = 1.0;
        for (int i = 0; i < n; ++i) {
            dxa[i] = mk[i] ? 0.0 : rhs[i];
            if (mk[i]) continue;
            const double dx = dxa[i], sl = x[i] - lo[i], su = hi[i] - x[i];
            const double a = -zl[i] - zl[i] * dx / sl, c = -zu[i] + zu[i] * dx / su;
            if (dx < 0.0) ap = fmin(ap, -sl / dx);
            if (dx > 0.0) ap = fmin(ap, su / dx);
            if (a < 0.0) ad = fmin(ad, -zl[i] / a);
            if (c < 0.0) ad = fmin(ad, -zu[i] / c);
        }
        double mua = 0.0;
        for (int i = 0; i < n; ++i) {
            if (mk[i]) continue;
            const double dx = dxa[i], sl = x[i] - lo[i], su = hi[i] - x[i];
            const double a = -zl[i] - zl[i] * dx / sl, c = -zu[i] + zu[i] * dx / su;
            mua += (sl + ap * dx) * (zl[i] + ad * a) + (su - ap * dx) * (zu[i] + ad * c);
        }
        mua /= 2.0 * nfree;
        const double ratio = mua / mu, smu = ratio * ratio * ratio * mu;
        for (int i = 0; i < n; ++i) {
            if (mk[i]) { rhs[i] = 0.0; continue; }
            const double dx = dxa[i], sl = x[i] - lo[i], su = hi[i] - x[i];
            const double a = -zl[i] - zl[i] * dx / sl, c = -zu[i] + zu[i] * dx / su;
            rhs[i] = -g[i] + (smu - dx * a) / sl - (smu + dx * c) / su;
        }
        solve(&q, rhs);
        const double gm = fmin(fmax(0.995, 1.0 - 10.0 * mu / (zscale * wmean)), 1.0 - 1e-9);
        double amax = 1.0 / gm;
        for (int i = 0; i < n; ++i) {
            if (mk[i]) continue;
            const double dx = rhs[i], da = dxa[i], sl = x[i] - lo[i], su = hi[i] - x[i];
            const double a = -zl[i] - zl[i] * da / sl, c = -zu[i] + zu[i] * da / su;
            dzl[i] = (-sl * zl[i] + smu - da * a - zl[i] * dx) / sl;
            dzu[i] = (-su * zu[i] + smu + da * c + zu[i] * dx) / su;
            if (dx < 0.0) amax = fmin(amax, -sl / dx);
            if (dx > 0.0) amax = fmin(amax, su / dx);
            if (dzl[i] < 0.0) amax = fmin(amax, -zl[i] / dzl[i]);
            if (dzu[i] < 0.0) amax = fmin(amax, -zu[i] / dzu[i]);
        }
        const double a = fmin(1.0, gm * amax);
        last_step = a;
        for (int i = 0; i < n; ++i) {
            if (mk[i]) continue;
            const double sl = x[i] - lo[i], su = hi[i] - x[i];
            const double rsl = (sl + a * rhs[i]) * zl[i], rzl = (zl[i] + a * dzl[i]) * sl;
            const double rsu = (su - a * rhs[i]) * zu[i], rzu = (zu[i] + a * dzu[i]) * su;
            const int al = rsl < 0.7 * rzl && sl + a * rhs[i] < 0.7 * sl, au = rsu < 0.7 * rzu && su - a * rhs[i] < 0.7 * su;
            ind[i] = al ? -1.0 : (au ? 1.0 : 0.0);
            x[i] += a * rhs[i];
            zl[i] += a * dzl[i];
            zu[i] += a * dzu[i];
        }
        gradient(&q, x, tmp, g);                          /* exact gradient every iteration (cheap on a CPU) */
    }
    if (st == BQ_OK && !converged) st = BQ_ITER_CAP;

    /* ---- active set: identification, block pivoting with single-pivot backup, one pinned row per neighbourhood ---- */
    if (st == BQ_OK && nfree > 0) {
        const int tapia = iters[0] >= 1 && last_step >= 0.9;
        for (int i = 0; i < n; ++i) {
            xs[i] = x[i];
            if (mk[i]) continue;
            const double w = hi[i] - lo[i], sl = x[i] - lo[i], su = hi[i] - x[i];
            signed char s = 0;
            if (sl * zscale < zl[i] * w) s = -1;
            else if (su * zscale < zu[i] * w) s = 1;
            if (tapia && s == 0) s = (signed char)ind[i];
            mk[i] = s;
        }
        const double TOLX = 1e-10, toly = 1e-10 * (fsc > 0.0 ? fsc : 1.0);
        int best = 2 * n + 1, pcnt = 3, done = 0;
        for (int it = 1; it <= 200 && !done; ++it) {
            iters[1] = it;
            for (int i = 0; i < n; ++i) xa[i] = mk[i] == 0 ? 0.0 : (mk[i] < 0 ? lo[i] : (mk[i] == 1 ? hi[i] : 0.5 * (lo[i] + hi[i])));
            gradient(&q, xa, tmp, t2);
            for (int i = 0; i < n; ++i) rhs[i] = mk[i] == 0 ? -t2[i] : xa[i];
            st = factor(&q, NULL, mk);
            if (st != BQ_OK) break;
            solve(&q, rhs);
            for (int i = 0; i < n; ++i) x[i] = mk[i] == 0 ? rhs[i] : xa[i];
            gradient(&q, x, tmp, g);
            int nv_ = 0, imax = -1;
            for (int i = 0; i < n; ++i) {
                int v = 0;
                double out = 0.0;
                if (mk[i] == 0) {
                    if (x[i] < lo[i] - TOLX) { v = -1; out = lo[i] - x[i]; }
                    else if (x[i] > hi[i] + TOLX) { v = 1; out = x[i] - hi[i]; }
                } else if (mk[i] == -1) { if (g[i] < -toly) v = 2; }
                else if (mk[i] == 1) { if (g[i] > toly) v = 2; }
                pv[i] = out;
                ind[i] = (double)v;
                if (v != 0) { ++nv_; imax = i; }
            }
            if (nv_ == 0) {
                for (int r = 0; r < 2; ++r) {              /* refinement through E on the final working set */
                    for (int i = 0; i < n; ++i) rhs[i] = mk[i] == 0 ? -g[i] : 0.0;
                    solve(&q, rhs);
                    double dm = 0.0;
                    for (int i = 0; i < n; ++i) if (mk[i] == 0) { x[i] += rhs[i]; dm = fmax(dm, fabs(rhs[i])); }
                    gradient(&q, x, tmp, g);
                    if (!(dm > 1e-8)) break;
                }
                done = 1;
                break;
            }
            int full;
            if (nv_ < best) { best = nv_; pcnt = 3; full = 1; }
            else if (pcnt > 0) { --pcnt; full = 1; }
            else full = 0;
            for (int i = 0; i < n; ++i) {
                const int v = (int)ind[i];
                if (v == 0 || !(full || i == imax)) continue;
                int take = 1;
                if (full && v != 2) {
                    for (int o = 1; o <= 8 && take; ++o)
                        if (pv[cyc(i - o, n)] >= pv[i] || pv[cyc(i + o, n)] > pv[i]) take = 0;
                }
                if (take) t2[i] = v == 2 ? 0.0 : (double)v; else t2[i] = (double)mk[i];
            }
            for (int i = 0; i < n; ++i) {
                const int v = (int)ind[i];
                if (v != 0 && (full || i == imax)) mk[i] = (signed char)t2[i];
            }
        }
        if (st == BQ_OK && !done) st = BQ_ITER_CAP;
    }

    /* ---- outputs: alpha, curvature rows checked, curvature-error post-check ---- */
    for (int i = 0; i < n; ++i) { x[i] = fmin(fmax(x[i], lo[i]), hi[i]); alpha[i] = x[i]; }
    e_mul(&q, x, tmp);
    double km = 0.0;
    for (int i = 0; i < n; ++i) km = fmax(km, fabs(tmp[i] + q.kref[i]));
    if (st == BQ_OK && km > kappa_bound * (1.0 + 1e-9)) st = BQ_KAPPA;
    {
        double *ax = dxa, *ay = dzl, *dx2 = dzu, *dy2 = ind;
        for (int i = 0; i < n; ++i) { ax[i] = nv[2 * i] * x[i]; ay[i] = nv[2 * i + 1] * x[i]; }
        for (int i = 0; i < n; ++i) {
            const double* d = q.Db + (size_t)i * EW;
            double sx = 0.0, sy = 0.0;
            int j = cyc(i - BE, n);
            for (int o = 0; o < EW; ++o) { sx += d[o] * ax[j]; sy += d[o] * ay[j]; j = j + 1 == n ? 0 : j + 1; }
            dx2[i] = sx; dy2[i] = sy;
        }
        double em = 0.0;
        for (int i = 0; i < n; ++i) {
            const int ip = i + 1 == n ? 0 : i + 1;
            const double s = sc ? sc[i] : 1.0, s2 = s * s;
            const double xpt = q.xp[i] + (ax[ip] - ax[i]) - (dx2[i] + 0.5 * s2 * dx2[ip]) / 3.0;
            const double ypt = q.yp[i] + (ay[ip] - ay[i]) - (dy2[i] + 0.5 * s2 * dy2[ip]) / 3.0;
            const double xpp = q.xpp[i] + dx2[i], ypp = q.ypp[i] + dy2[i];
            const double k0 = (q.xp[i] * ypp - q.yp[i] * xpp) / pow(q.xp[i] * q.xp[i] + q.yp[i] * q.yp[i], 1.5);
            const double k1 = (xpt * ypp - ypt * xpp) / pow(xpt * xpt + ypt * ypt, 1.5);
            em = fmax(em, fabs(k1 - k0));
        }
        *curv_err = em;
    }
    free(q.Eb); free(wk); free(mk);
    return st;
}

/* Uniform-n batch, one problem per thread over `nthreads` threads (<= 0: all).  ref [batch][n][4], nv [batch][n][2],
 * sc [batch][n] or NULL; alpha [batch][n], curv_err / status [batch], iters [batch][2].  Returns the number of threads used. */
int bqp_solve_batch(int batch, int n, const double* ref, const double* nv, const double* sc, double kappa_bound, double w_veh,
                    double* alpha, double* curv_err, int* status, int* iters, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > batch) nthreads = batch;
    used = nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int b = 0; b < batch; ++b)
        status[b] = bqp_solve(n, ref + (size_t)b * n * 4, nv + (size_t)b * n * 2, sc ? sc + (size_t)b * n : NULL, kappa_bound,
                              w_veh, alpha + (size_t)b * n, curv_err + b, iters + 2 * b);
    return used;
}
