// mcq_kernels.hip -- see mcq_kernels.h for the overview and memory layouts.
#include "mcq_kernels.h"

#include <math.h>

#define MCQ_NT 256
#define MCQ_NW (MCQ_NT / 64)
#define NSLOT (MCQ_BH_MAX + 2)
#define WLD (MCQ_BH_MAX + 1)
#define CLD MCQ_P_MAX
#define SLD (MCQ_P_MAX + 1)

// ---------------------------------------------------------------------------------------------------------------------
// shared-memory carve-up of the solver kernel (doubles)
// ---------------------------------------------------------------------------------------------------------------------
#define SM_RED 0
#define SM_XD (SM_RED + 64)
#define SM_PART (SM_XD + 64)
#define SM_S (SM_PART + MCQ_NW * 64)
#define SM_WIN (SM_S + MCQ_P_MAX * SLD)
#define SM_CW (SM_WIN + NSLOT * WLD)
#define SM_LRW (SM_CW + NSLOT * CLD)
#define SM_TOTAL (SM_LRW + NSLOT * WLD)

size_t mcq_solve_smem_bytes() { return sizeof(double) * SM_TOTAL; }

// ---------------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cyc(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

// signed cyclic difference a - b in (-n/2, n/2]
__device__ __forceinline__ int sdiff(int a, int b, int n)
{
    int d = cyc(a - b, n);
    return d > n / 2 ? d - n : d;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

// op: 0 sum, 1 min, 2 max.  All threads of the block must call; result returned to every thread.
__device__ double block_reduce(double v, int op, double* red)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = op == 0 ? wave_sum(v) : (op == 1 ? wave_min(v) : wave_max(v));
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = red[0];
    for (int k = 1; k < MCQ_NW; ++k) r = op == 0 ? r + red[k] : (op == 1 ? fmin(r, red[k]) : fmax(r, red[k]));
    return r;
}

__device__ McqWork mcq_work(const McqBatch& B, int pb, int& n, double& kb, double& wv)
{
    McqWork w;
    const size_t nm = (size_t)B.nmax;
    n = B.n_list ? B.n_list[pb] : B.n;
    kb = B.kappa_bound_list ? B.kappa_bound_list[pb] : B.kappa_bound;
    wv = B.w_veh_list ? B.w_veh_list[pb] : B.w_veh;
    w.ref = B.ref + (size_t)pb * nm * 4;
    w.nv = B.nv + (size_t)pb * nm * 2;
    w.sc = B.sc ? B.sc + (size_t)pb * nm : nullptr;
    w.Eb = B.Eb + (size_t)pb * nm * MCQ_ELD;
    w.Et = B.Et + (size_t)pb * nm * MCQ_ELD;
    w.Db = B.Db + (size_t)pb * nm * MCQ_ELD;
    w.H = B.H + (size_t)pb * nm * MCQ_HLD;
    w.L = B.L + (size_t)pb * nm * MCQ_HLD;
    w.vec = B.vec + (size_t)pb * nm * MCQ_NVEC;
    w.state = B.state + (size_t)pb * nm;
    w.alpha = B.alpha + (size_t)pb * nm;
    w.curv_err = B.curv_err + pb;
    w.status = B.status + pb;
    w.info = B.info ? B.info + pb : nullptr;
    return w;
}

#define VEC(w, nmax, id) ((w).vec + (size_t)(id) * (size_t)(nmax))

// dst_i = sum_{o=-bl..br} Mb[i][bl+o] * src[(i+o) mod n] + addc * add_i      (one wave per row, coalesced row read)
__device__ void band_matvec(const double* Mb, int bl, int br, int n, const double* src, const double* add, double addc,
                            double* dst)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ew = bl + br + 1;
    for (int i = wv; i < n; i += MCQ_NW) {
        const double* row = Mb + (size_t)i * MCQ_ELD;
        double acc = 0.0;
        if (lane < ew) acc = row[lane] * src[cyc(i + lane - bl, n)];
        if (lane == 0 && ew > 64) acc += row[64] * src[cyc(i + 64 - bl, n)];
        acc = wave_sum(acc);
        if (lane == 0) dst[i] = acc + (add ? addc * add[i] : 0.0);
    }
}

// =====================================================================================================================
// K1: assembly
// =====================================================================================================================
__global__ void __launch_bounds__(MCQ_NT) mcq_assemble_kernel(McqBatch B)
{
    __shared__ double red[64];
    const int tid = threadIdx.x;
    int n;
    double kb, wveh;
    const McqWork w = mcq_work(B, blockIdx.x, n, kb, wveh);
    const int nm = B.nmax;
    const McqDims d = mcq_dims(n < 3 ? 3 : n, B.band_e);
    double* LO = VEC(w, nm, V_LO);
    double* HI = VEC(w, nm, V_HI);
    double* S = VEC(w, nm, V_T0);    // spline scalings
    double* DE = VEC(w, nm, V_T1);   // periodic pivots, top-down
    double* EP = VEC(w, nm, V_T2);   // periodic pivots, bottom-up
    double* XP = VEC(w, nm, V_XP);
    double* YP = VEC(w, nm, V_YP);
    double* CP = VEC(w, nm, V_CP);
    double* KRF = VEC(w, nm, V_KREF);
    double* XPP = VEC(w, nm, V_XPP);
    double* YPP = VEC(w, nm, V_YPP);
    double* G = w.L;                 // T^-1 rows, leading dimension MCQ_GLD (scratch inside the L slab)

    // ---- phase 0: validate, box bounds  [-(w_l - w_veh/2), w_r - w_veh/2]   -----------------
    double flag_bad = 0.0, flag_inf = 0.0;
    if (n < 3) flag_bad = 1.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        const double x = w.ref[4 * i], y = w.ref[4 * i + 1], wr = w.ref[4 * i + 2], wl = w.ref[4 * i + 3];
        const double nx = w.nv[2 * i], ny = w.nv[2 * i + 1];
        const double s = w.sc ? w.sc[i] : 1.0;
        if (!(isfinite(x) && isfinite(y) && isfinite(wr) && isfinite(wl) && isfinite(nx) && isfinite(ny) && isfinite(s)
              && s > 0.0))
            flag_bad = 1.0;
        const double lo = -(wl - 0.5 * wveh), hi = wr - 0.5 * wveh;
        if (hi < lo) flag_inf = 1.0;
        LO[i] = lo;
        HI[i] = hi;
        S[i] = s;
    }
    flag_bad = block_reduce(flag_bad, 2, red);
    flag_inf = block_reduce(flag_inf, 2, red);
    const int st = flag_bad > 0.0 ? MCQ_BAD_INPUT : (flag_inf > 0.0 ? MCQ_INFEASIBLE : MCQ_OK);
    if (tid == 0) {
        *w.status = st;
        *w.curv_err = 0.0;
        if (w.info) {
            mcq_info z;
            z.ipm_iters = z.as_iters = z.n_active_box = z.n_active_kappa = 0;
            z.kappa_max = 0.0;
            z.kkt_res = 0.0;
            *w.info = z;
        }
    }
    if (st != MCQ_OK) {
        for (int i = tid; i < n; i += MCQ_NT) w.alpha[i] = 0.0;
        return;
    }
    __syncthreads();

    // ---- phase 1: periodic pivots of the cyclic tridiagonal system in the c-coefficients -----------------------------
    // centre m:  1*c_{m-1} + (2 s_{m-1}^2 + 2 s_{m-1}) c_m + (s_{m-1} s_m^2) c_{m+1} = 3 (s_{m-1} D_m - D_{m-1})
#define TDIAG(m) (2.0 * S[cyc((m) - 1, n)] * S[cyc((m) - 1, n)] + 2.0 * S[cyc((m) - 1, n)])
#define TSUP(m) (S[cyc((m) - 1, n)] * S[(m)] * S[(m)])
    {
        const int chunk = (n + MCQ_NT - 1) / MCQ_NT;
        const int m0 = tid * chunk;
        const int m1 = m0 + chunk < n ? m0 + chunk : n;
        if (m0 < n) {
            double dd = TDIAG(cyc(m0 - MCQ_PIVOT_WARMUP, n));
            for (int k = m0 - MCQ_PIVOT_WARMUP + 1; k < m1; ++k) {
                const int m = cyc(k, n);
                dd = TDIAG(m) - TSUP(cyc(m - 1, n)) / dd;
                if (k >= m0) DE[m] = dd;
            }
            double ee = TDIAG(cyc(m1 - 1 + MCQ_PIVOT_WARMUP, n));
            for (int k = m1 - 2 + MCQ_PIVOT_WARMUP; k >= m0; --k) {
                const int m = cyc(k, n);
                ee = TDIAG(m) - TSUP(m) / ee;
                if (k < m1) EP[m] = ee;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: rows of T^-1 (periodic Green's function, images folded onto the ring) and the c-coefficients -------
    const int W = d.bE + 2;
    // images of offset k land inside [-W, W] only if k >= n - W: run the recurrences further for short rings
    const int KR = (n - W > 96) ? W : 96;
    for (int i = tid; i < n; i += MCQ_NT) {
        double* g = G + (size_t)i * MCQ_GLD;
        for (int k = 0; k < MCQ_GLD; ++k) g[k] = 0.0;
        const double g0 = 1.0 / (DE[i] + EP[i] - TDIAG(i));
        for (int o = -W; o <= W; ++o) if (cyc(o, n) == 0) g[MCQ_GW + o] += g0;
        double cur = g0;
        for (int k = 1; k <= KR; ++k) {
            const int j = cyc(i + k - 1, n);
            cur = -(TSUP(j) / EP[cyc(j + 1, n)]) * cur;
            if (k <= W) g[MCQ_GW + k] += cur;
            if (k >= n - W)   // fold images: every offset o in [-W, W] with o == k (mod n), o != k
                for (int o = k - n; o >= -W; o -= n) if (o <= W) g[MCQ_GW + o] += cur;
        }
        cur = g0;
        for (int k = 1; k <= KR; ++k) {
            const int j = cyc(i - k + 1, n);
            cur = -cur / DE[cyc(j - 1, n)];
            if (k <= W) g[MCQ_GW - k] += cur;
            if (k >= n - W)
                for (int o = -k + n; o <= W; o += n) if (o >= -W) g[MCQ_GW + o] += cur;
        }
        double cx = 0.0, cy = 0.0;
        const int klo = -((W < (n - 1) / 2) ? W : (n - 1) / 2), khi = (W < n / 2) ? W : n / 2;   // each column once
        for (int k = klo; k <= khi; ++k) {
            const int m = cyc(i + k, n), mp = cyc(m + 1, n), mm = cyc(m - 1, n);
            const double sm1 = S[mm];
            const double rx = 3.0 * (sm1 * (w.ref[4 * mp] - w.ref[4 * m]) - (w.ref[4 * m] - w.ref[4 * mm]));
            const double ry = 3.0 * (sm1 * (w.ref[4 * mp + 1] - w.ref[4 * m + 1]) - (w.ref[4 * m + 1] - w.ref[4 * mm + 1]));
            cx += g[MCQ_GW + k] * rx;
            cy += g[MCQ_GW + k] * ry;
        }
        XPP[i] = 2.0 * cx;   // x''(0) of spline i
        YPP[i] = 2.0 * cy;
    }
    __syncthreads();

    // ---- phase 3a: x', y', curvature pre-factor, reference curvature ------------------------------------------------
    for (int i = tid; i < n; i += MCQ_NT) {
        const int ip = cyc(i + 1, n);
        const double s2 = S[i] * S[i];
        const double xp = (w.ref[4 * ip] - w.ref[4 * i]) - (XPP[i] + 0.5 * s2 * XPP[ip]) / 3.0;
        const double yp = (w.ref[4 * ip + 1] - w.ref[4 * i + 1]) - (YPP[i] + 0.5 * s2 * YPP[ip]) / 3.0;
        const double den = pow(xp * xp + yp * yp, 1.5);
        const double cp = den != 0.0 ? 1.0 / den : 0.0;
        XP[i] = xp;
        YP[i] = yp;
        CP[i] = cp;
        KRF[i] = cp * (xp * YPP[i] - yp * XPP[i]);
    }
    __syncthreads();

    // ---- phase 3b: D band (x'' = D x) and E_kappa band -----------------------------------------------------------------
    const int ew = d.ew;
    for (int idx = tid; idx < n * ew; idx += MCQ_NT) {
        const int i = idx / ew, oo = idx - i * ew, o = oo - d.bE;
        const int j = cyc(i + o, n);
        const double* g = G + (size_t)i * MCQ_GLD + MCQ_GW + o;
        const double dv = 6.0 * (g[1] - (1.0 + S[cyc(j - 1, n)]) * g[0] + S[cyc(j - 2, n)] * g[-1]);
        w.Db[(size_t)i * MCQ_ELD + oo] = dv;
        w.Eb[(size_t)i * MCQ_ELD + oo] = dv * CP[i] * (XP[i] * w.nv[2 * j + 1] - YP[i] * w.nv[2 * j]);
    }
    __syncthreads();
    // ---- phase 3c: transpose band  Et[j][bR+o] = E[(j+o) mod n][j],  -bR <= o <= bE ---------------------------------
    for (int idx = tid; idx < n * ew; idx += MCQ_NT) {
        const int j = idx / ew, oo = idx - j * ew, o = oo - d.bR;
        w.Et[(size_t)j * MCQ_ELD + oo] = w.Eb[(size_t)cyc(j + o, n) * MCQ_ELD + (d.bE - o)];
    }
#undef TDIAG
#undef TSUP
}

// =====================================================================================================================
// K2: H = E'E (bordered band) and f
// =====================================================================================================================
// H[i,j] = sum_r E[r,i] E[r,j] = sum_o Et[i][bR+o] * Et[j][bR+o+dd],  r = i+o,  dd = cyclic (i - j);  -bR <= o <= bE
__device__ __forceinline__ double h_entry(const double* Et, int bE, int bR, int n, int i, int j)
{
    const int dd = sdiff(i, j, n);
    const double* ri = Et + (size_t)i * MCQ_ELD + bR;
    const double* rj = Et + (size_t)j * MCQ_ELD + bR;
    double acc = 0.0;
    for (int o = -bR; o <= bE; ++o) {
        int t = o + dd;          // r - j, defined modulo n; the band holds every column at most once
        if (t > bE) t -= n;
        else if (t < -bR) t += n;
        if (t >= -bR && t <= bE) acc += ri[o] * rj[t];
    }
    return acc;
}

__global__ void __launch_bounds__(MCQ_NT) mcq_gram_kernel(McqBatch B)
{
    const int tid = threadIdx.x + blockIdx.y * MCQ_NT;
    const int nthreads = MCQ_NT * gridDim.y;
    int n;
    double kb, wveh;
    const McqWork w = mcq_work(B, blockIdx.x, n, kb, wveh);
    if (*w.status != MCQ_OK) return;
    const int nm = B.nmax;
    const McqDims d = mcq_dims(n, B.band_e);
    const double* KR = VEC(w, nm, V_KREF);
    double* F = VEC(w, nm, V_F);

    for (int j = tid; j < n; j += nthreads) {
        const double* r = w.Et + (size_t)j * MCQ_ELD + d.bR;
        double acc = 0.0;
        for (int o = -d.bR; o <= d.bE; ++o) acc += r[o] * KR[cyc(j + o, n)];
        F[j] = MCQ_F_SCALE * acc;
    }
    // interior rows: band part
    const int bw = MCQ_BH_MAX + 1;
    for (int idx = tid; idx < d.ni * bw; idx += nthreads) {
        const int i = idx / bw, k = idx - i * bw;
        double v = 0.0;
        if (k <= d.b && i + k < d.ni) v = h_entry(w.Et, d.bE, d.bR, n, i, i + k);
        w.H[(size_t)i * MCQ_HLD + k] = v;
    }
    // all rows: border part  H[i, ni+jj]
    for (int idx = tid; idx < n * MCQ_P_MAX; idx += nthreads) {
        const int i = idx / MCQ_P_MAX, jj = idx - i * MCQ_P_MAX;
        double v = 0.0;
        if (jj < d.p) {
            const int j = d.ni + jj;
            const int dist = abs(sdiff(i, j, n));
            if (dist <= d.bH) v = h_entry(w.Et, d.bE, d.bR, n, i, j);
        }
        w.H[(size_t)i * MCQ_HLD + MCQ_HBO + jj] = v;
    }
}

// =====================================================================================================================
// K3: solver
// =====================================================================================================================
struct SolveCtx {
    McqDims d;
    McqWork w;
    int nm;
    double* sm;
};

// ---- bordered-band Cholesky of  M = H + diag(sig)  with rows/cols of pinned variables replaced by identity ----------
// Right-looking, one column per step, LDS sliding window of b+2 columns (one __syncthreads per column).
// Output: L rows in w.L (row i: [0] = 1/L_ii, [k] = L[i,i-k]; [HBO+jj] = W[i][jj]); L_S (p x p, lower) in LDS SM_S.
// Returns 0 or MCQ_NOT_PD (uniform across the block).
__device__ int factor(const SolveCtx& c, const double* sig, const signed char* mk)
{
    const int tid = threadIdx.x;
    const int b = c.d.b, p = c.d.p, ni = c.d.ni;
    double* win = c.sm + SM_WIN;
    double* cwn = c.sm + SM_CW;
    double* lrw = c.sm + SM_LRW;
    double* Sm = c.sm + SM_S;
    const double* H = c.w.H;
    double* L = c.w.L;

    double sacc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) sacc[m] = 0.0;

    __syncthreads();
    for (int q = tid; q < NSLOT * WLD; q += MCQ_NT) lrw[q] = 0.0;
    // preload columns 0..min(b, ni-1) (and their border rows)
    const int npre = (b + 1 < ni ? b + 1 : ni);
    for (int q = tid; q < npre * (WLD + CLD); q += MCQ_NT) {
        const int cc = q / (WLD + CLD), e = q - cc * (WLD + CLD);
        const bool pc = mk && mk[cc] != 0;
        if (e < WLD) {
            const int k = e, r = cc + k;
            double v = 0.0;
            if (k <= b && r < ni) {
                v = H[(size_t)cc * MCQ_HLD + k];
                const bool pr = mk && mk[r] != 0;
                if (pc || pr) v = (k == 0) ? 1.0 : 0.0;
                else if (k == 0 && sig) v += sig[cc];
            }
            win[(cc % NSLOT) * WLD + k] = v;
        } else {
            const int jj = e - WLD;
            double v = 0.0;
            if (jj < p) {
                v = H[(size_t)cc * MCQ_HLD + MCQ_HBO + jj];
                if (pc || (mk && mk[ni + jj] != 0)) v = 0.0;
            }
            cwn[(cc % NSLOT) * CLD + jj] = v;
        }
    }
    __syncthreads();

    int fail = 0;
    for (int i = 0; i < ni; ++i) {
        const int slot = i % NSLOT;
        const double* ci = win + slot * WLD;
        const double* cwi = cwn + slot * CLD;
        const double piv = ci[0];
        if (!(piv > 0.0)) { fail = 1; break; }   // uniform: every thread reads the same LDS word
        const double rinv = 1.0 / piv;
        const double rs = 1.0 / sqrt(piv);
        const int nrem = (b < ni - 1 - i) ? b : ni - 1 - i;

        // (a) band part of the trailing update: M[r,c] -= a_r a_c / piv,  i < c <= r <= i + nrem
        {
            const int dc = 1 + (tid & 63);
            if (dc <= nrem) {
                const double ac = ci[dc] * rinv;
                double* colc = win + ((i + dc) % NSLOT) * WLD;
                for (int dr = dc + (tid >> 6); dr <= nrem; dr += MCQ_NW) colc[dr - dc] -= ci[dr] * ac;
            }
        }
        // (b) border coupling rows:  C[r][jj] -= a_r cw_i[jj] / piv
        {
            const int jj = tid & 63;
            if (jj < p) {
                const double cj = cwi[jj] * rinv;
                for (int dr = 1 + (tid >> 6); dr <= nrem; dr += MCQ_NW) cwn[((i + dr) % NSLOT) * CLD + jj] -= ci[dr] * cj;
            }
        }
        // (c) Schur complement accumulators  S[j1][j2] -= cw_i[j1] cw_i[j2] / piv   (j2 = lane, j1 = wave + 4 m)
        {
            const int j2 = tid & 63;
            const double cj = cwi[j2] * rinv;
#pragma unroll
            for (int m = 0; m < 16; ++m) sacc[m] -= cwi[(tid >> 6) + MCQ_NW * m] * cj;
        }
        // (d) emit column i of L into the row buffer, W row to global, flush finished row i-1, load column i+b+1
        {
            const int cn = i + b + 1;
            const int nA = nrem + 1, nB = p, nC = (i > 0) ? b + 1 : 0, nD = (cn < ni) ? (WLD + CLD) : 0;
            for (int q = tid; q < nA + nB + nC + nD; q += MCQ_NT) {
                if (q < nA) {
                    const int k = q;
                    lrw[((i + k) % NSLOT) * WLD + k] = (k == 0) ? rs : ci[k] * rs;
                } else if (q < nA + nB) {
                    const int jj = q - nA;
                    L[(size_t)i * MCQ_HLD + MCQ_HBO + jj] = cwi[jj] * rs;
                } else if (q < nA + nB + nC) {
                    const int k = q - nA - nB;
                    L[(size_t)(i - 1) * MCQ_HLD + k] = lrw[((i - 1) % NSLOT) * WLD + k];
                } else {
                    const int e = q - nA - nB - nC;
                    const bool pc = mk && mk[cn] != 0;
                    if (e < WLD) {
                        const int k = e, r = cn + k;
                        double v = 0.0;
                        if (k <= b && r < ni) {
                            v = H[(size_t)cn * MCQ_HLD + k];
                            const bool pr = mk && mk[r] != 0;
                            if (pc || pr) v = (k == 0) ? 1.0 : 0.0;
                            else if (k == 0 && sig) v += sig[cn];
                        }
                        win[(cn % NSLOT) * WLD + k] = v;
                    } else {
                        const int jj = e - WLD;
                        double v = 0.0;
                        if (jj < p) {
                            v = H[(size_t)cn * MCQ_HLD + MCQ_HBO + jj];
                            if (pc || (mk && mk[ni + jj] != 0)) v = 0.0;
                        }
                        cwn[(cn % NSLOT) * CLD + jj] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (fail) return MCQ_NOT_PD;
    for (int k = tid; k <= b; k += MCQ_NT) L[(size_t)(ni - 1) * MCQ_HLD + k] = lrw[((ni - 1) % NSLOT) * WLD + k];

    // ---- Schur complement of the border: S = D - W'W, dense Cholesky in LDS -------------------------------------------
    {
        const int j2 = tid & 63;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int j1 = (tid >> 6) + MCQ_NW * m;
            double v = 0.0;
            if (j1 < p && j2 < p) {
                const bool pj = mk && (mk[ni + j1] != 0 || mk[ni + j2] != 0);
                if (pj) v = (j1 == j2) ? 1.0 : 0.0;
                else {
                    v = H[(size_t)(ni + j1) * MCQ_HLD + MCQ_HBO + j2] + sacc[m];
                    if (j1 == j2 && sig) v += sig[ni + j1];
                }
            }
            Sm[j1 * SLD + j2] = v;
        }
    }
    __syncthreads();
    for (int j = 0; j < p; ++j) {
        const double piv = Sm[j * SLD + j];
        if (!(piv > 0.0)) { fail = 1; break; }
        const double rinv = 1.0 / piv;
        const int rem = p - 1 - j;
        // finalise column j-1 (scaled) while updating with column j (unscaled): disjoint elements
        if (j > 0) {
            const double pv = Sm[(j - 1) * SLD + (j - 1)];
            const double rs = 1.0 / sqrt(pv);
            for (int r = j + tid; r < p; r += MCQ_NT) Sm[r * SLD + (j - 1)] *= rs;
        }
        for (int e = tid; e < rem * rem; e += MCQ_NT) {
            const int r = j + 1 + e / rem, cc = j + 1 + e % rem;
            if (cc <= r) Sm[r * SLD + cc] -= Sm[r * SLD + j] * Sm[cc * SLD + j] * rinv;
        }
        __syncthreads();
        if (j > 0 && tid == 0) Sm[(j - 1) * SLD + (j - 1)] = sqrt(Sm[(j - 1) * SLD + (j - 1)]);
    }
    if (fail) return MCQ_NOT_PD;
    __syncthreads();
    if (tid == 0 && p > 0) Sm[(p - 1) * SLD + (p - 1)] = sqrt(Sm[(p - 1) * SLD + (p - 1)]);
    __syncthreads();
    return 0;
}

// ---- solve  M v = rhs  in place (v in global memory) with the factor produced by factor() ---------------------------
__device__ void solve(const SolveCtx& c, double* v)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = c.d.b, p = c.d.p, ni = c.d.ni;
    const double* L = c.w.L;
    double* Sm = c.sm + SM_S;
    double* xd = c.sm + SM_XD;
    double* part = c.sm + SM_PART;

    __syncthreads();
    // forward substitution, interior rows (wave 0; lane l keeps the newest y_j with j == l mod 64)
    if (wv == 0) {
        double ycur = 0.0;
        for (int i0 = 0; i0 < ni; i0 += 8) {
            double lv[8], dg[8], rv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                const int k = ((i - 1 - lane) & 63) + 1;
                lv[u] = 0.0; dg[u] = 0.0; rv[u] = 0.0;
                if (i < ni) {
                    if (k <= b) lv[u] = L[(size_t)i * MCQ_HLD + k];
                    dg[u] = L[(size_t)i * MCQ_HLD];
                    rv[u] = v[i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                if (i < ni) {
                    const double s = wave_sum(lv[u] * ycur);
                    const double yi = (rv[u] - s) * dg[u];
                    if (lane == (i & 63)) { ycur = yi; v[i] = yi; }
                }
            }
        }
    }
    __syncthreads();
    // border right-hand side  t = v_D - W' y_B
    {
        double acc = 0.0;
        if (lane < p)
            for (int i = wv; i < ni; i += MCQ_NW) acc += L[(size_t)i * MCQ_HLD + MCQ_HBO + lane] * v[i];
        part[wv * 64 + lane] = acc;
    }
    __syncthreads();
    if (wv == 0) {
        double t = 0.0;
        if (lane < p) {
            t = v[ni + lane];
            for (int q = 0; q < MCQ_NW; ++q) t -= part[q * 64 + lane];
        }
        // dense forward  L_S y = t
        for (int j = 0; j < p; ++j) {
            const double s = wave_sum(lane < j ? Sm[j * SLD + lane] * t : 0.0);
            const double tj = (__shfl(t, j) - s) / Sm[j * SLD + j];
            if (lane == j) t = tj;
        }
        // dense backward  L_S' x = y
        for (int j = p - 1; j >= 0; --j) {
            const double s = wave_sum((lane > j && lane < p) ? Sm[lane * SLD + j] * t : 0.0);
            const double xj = (__shfl(t, j) - s) / Sm[j * SLD + j];
            if (lane == j) t = xj;
        }
        if (lane < p) v[ni + lane] = t;
        xd[lane] = lane < p ? t : 0.0;
    }
    __syncthreads();
    // y_B -= W x_D   (one wave per row)
    for (int i = wv; i < ni; i += MCQ_NW) {
        const double s = wave_sum(lane < p ? L[(size_t)i * MCQ_HLD + MCQ_HBO + lane] * xd[lane] : 0.0);
        if (lane == 0) v[i] -= s;
    }
    __syncthreads();
    // backward substitution, interior rows (wave 0, axpy form; lane l owns the pending row j == l mod 64)
    if (wv == 0 && ni > 0) {
        int jown = ni - 1 - ((ni - 1 - lane) & 63);          // largest j <= ni-1 with j == lane (mod 64); may be < 0
        double acc = jown >= 0 ? v[jown] : 0.0;
        double vnext = jown - 64 >= 0 ? v[jown - 64] : 0.0;
        for (int i0 = ni - 1; i0 >= 0; i0 -= 8) {
            double lv[8], dg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 - u;
                const int k = ((i - 1 - lane) & 63) + 1;
                lv[u] = 0.0; dg[u] = 0.0;
                if (i >= 0) {
                    if (k <= b && i - k >= 0) lv[u] = L[(size_t)i * MCQ_HLD + k];
                    dg[u] = L[(size_t)i * MCQ_HLD];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 - u;
                if (i >= 0) {
                    const int owner = i & 63;
                    const double xi = __shfl(acc, owner) * dg[u];
                    if (lane == owner) {
                        v[i] = xi;
                        acc = vnext;
                        jown -= 64;
                        vnext = jown - 64 >= 0 ? v[jown - 64] : 0.0;
                    }
                    acc -= lv[u] * xi;
                }
            }
        }
    }
    __syncthreads();
}

// gradient  g = E'(E x + F k_ref)   (tmp: scratch vector)
__device__ void gradient(const SolveCtx& c, const double* x, double* tmp, double* g)
{
    const int n = c.d.n;
    __syncthreads();
    band_matvec(c.w.Eb, c.d.bE, c.d.bR, n, x, VEC(c.w, c.nm, V_KREF), MCQ_F_SCALE, tmp);
    __syncthreads();
    band_matvec(c.w.Et, c.d.bR, c.d.bE, n, tmp, nullptr, 0.0, g);
    __syncthreads();
}

__global__ void __launch_bounds__(MCQ_NT) mcq_solve_kernel(McqBatch B)
{
    HIP_DYNAMIC_SHARED(double, smem)   // the only LDS of this kernel: base is 16-byte aligned (guide, Guideline 17)
    const int tid = threadIdx.x;
    int n;
    double kbound, wveh;
    SolveCtx c;
    c.w = mcq_work(B, blockIdx.x, n, kbound, wveh);
    if (*c.w.status != MCQ_OK) return;
    c.nm = B.nmax;
    c.d = mcq_dims(n, B.band_e);
    c.sm = smem;
    double* red = smem + SM_RED;
    const int nm = B.nmax;

    const double* LO = VEC(c.w, nm, V_LO);
    const double* HI = VEC(c.w, nm, V_HI);
    const double* F = VEC(c.w, nm, V_F);
    double* X = VEC(c.w, nm, V_X);
    double* G = VEC(c.w, nm, V_G);
    double* ZL = VEC(c.w, nm, V_ZL);
    double* ZU = VEC(c.w, nm, V_ZU);
    double* SIG = VEC(c.w, nm, V_SIG);
    double* RHS = VEC(c.w, nm, V_RHS);
    double* DXA = VEC(c.w, nm, V_DXA);
    double* T0 = VEC(c.w, nm, V_T0);
    double* T1 = VEC(c.w, nm, V_T1);
    double* T2 = VEC(c.w, nm, V_T2);
    double* T3 = VEC(c.w, nm, V_T3);
    signed char* ST = c.w.state;

    const double FIX_TOL = 1e-12;
    int status = MCQ_OK;
    int ipm_iters = 0, as_iters = 0;

    // ---------------------------------------------------------------------------------------------------------------
    // interior point (Mehrotra predictor-corrector) on   min 1/2 x'Hx + f'x,  lo <= x <= hi
    // ---------------------------------------------------------------------------------------------------------------
    double wsum = 0.0, nfree_d = 0.0, fmaxl = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        const double wdt = HI[i] - LO[i];
        const bool fixed = !(wdt > FIX_TOL);
        ST[i] = fixed ? 2 : 0;
        X[i] = 0.5 * (LO[i] + HI[i]);
        if (!fixed) { wsum += wdt; nfree_d += 1.0; }
        fmaxl = fmax(fmaxl, fabs(F[i]));
    }
    wsum = block_reduce(wsum, 0, red);
    nfree_d = block_reduce(nfree_d, 0, red);
    const double fscale = block_reduce(fmaxl, 2, red);
    const double wmean = nfree_d > 0.0 ? wsum / nfree_d : 1.0;

    gradient(c, X, T0, G);
    double gm = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) gm = fmax(gm, fabs(G[i]));
    double zscale = block_reduce(gm, 2, red);
    if (!(zscale > 0.0)) zscale = fscale > 0.0 ? fscale : 1.0;
    for (int i = tid; i < n; i += MCQ_NT) { ZL[i] = ST[i] == 0 ? zscale : 0.0; ZU[i] = ZL[i]; }
    __syncthreads();

    const double IPM_TOL = 1e-10;
    if (nfree_d > 0.0) {
        for (int it = 1; it <= B.max_ipm_iter; ++it) {
            // residuals / duality measure
            double mu = 0.0, rdm = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) continue;
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                mu += sl * ZL[i] + su * ZU[i];
                rdm = fmax(rdm, fabs(G[i] - ZL[i] + ZU[i]));
                SIG[i] = ZL[i] / sl + ZU[i] / su;
            }
            mu = block_reduce(mu, 0, red) / (2.0 * nfree_d);
            rdm = block_reduce(rdm, 2, red);
            if (mu < IPM_TOL * zscale * wmean && rdm < IPM_TOL * zscale) break;
            ipm_iters = it;

            const int fs = factor(c, SIG, ST);
            if (fs != 0) { status = fs; break; }

            // predictor
            for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -G[i] : 0.0;
            solve(c, RHS);
            double ap = 1.0, ad = 1.0;
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) { DXA[i] = 0.0; continue; }
                const double dx = RHS[i];
                DXA[i] = dx;
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
                if (dx < 0.0) ap = fmin(ap, -sl / dx);
                if (dx > 0.0) ap = fmin(ap, su / dx);
                if (dzl < 0.0) ad = fmin(ad, -ZL[i] / dzl);
                if (dzu < 0.0) ad = fmin(ad, -ZU[i] / dzu);
            }
            ap = block_reduce(ap, 1, red);
            ad = block_reduce(ad, 1, red);
            double mua = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) continue;
                const double dx = DXA[i];
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
                mua += (sl + ap * dx) * (ZL[i] + ad * dzl) + (su - ap * dx) * (ZU[i] + ad * dzu);
            }
            mua = block_reduce(mua, 0, red) / (2.0 * nfree_d);
            const double ratio = mua / mu;
            const double sigma = ratio * ratio * ratio;
            const double smu = sigma * mu;

            // corrector
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) { RHS[i] = 0.0; continue; }
                const double dx = DXA[i];
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
                RHS[i] = -G[i] + (smu - dx * dzl) / sl - (smu + dx * dzu) / su;
            }
            solve(c, RHS);
            double amax = 1.0 / 0.995;
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) continue;
                const double dx = RHS[i], da = DXA[i];
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double dzla = -ZL[i] - ZL[i] * da / sl, dzua = -ZU[i] + ZU[i] * da / su;
                const double dzl = (-sl * ZL[i] + smu - da * dzla - ZL[i] * dx) / sl;
                const double dzu = (-su * ZU[i] + smu + da * dzua + ZU[i] * dx) / su;
                T1[i] = dzl;
                T2[i] = dzu;
                if (dx < 0.0) amax = fmin(amax, -sl / dx);
                if (dx > 0.0) amax = fmin(amax, su / dx);
                if (dzl < 0.0) amax = fmin(amax, -ZL[i] / dzl);
                if (dzu < 0.0) amax = fmin(amax, -ZU[i] / dzu);
            }
            amax = block_reduce(amax, 1, red);
            const double a = fmin(1.0, 0.995 * amax);
            for (int i = tid; i < n; i += MCQ_NT) {
                if (ST[i] != 0) continue;
                X[i] += a * RHS[i];
                ZL[i] += a * T1[i];
                ZU[i] += a * T2[i];
            }
            gradient(c, X, T0, G);
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // active-set identification + block principal pivoting on the vertex (exact KKT point)
    // ---------------------------------------------------------------------------------------------------------------
    double kkt = 0.0;
    if (status == MCQ_OK) {
        for (int i = tid; i < n; i += MCQ_NT) {
            if (ST[i] != 0) continue;
            const double wdt = HI[i] - LO[i];
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            signed char s = 0;
            if (sl * zscale < ZL[i] * wdt) s = -1;
            else if (su * zscale < ZU[i] * wdt) s = 1;
            ST[i] = s;
        }
        __syncthreads();
        const double TOLX = 1e-10;
        const double toly = 1e-10 * (fscale > 0.0 ? fscale : 1.0);
        int best = n + 1, pcnt = 3;
        bool converged = false;
        for (int it = 1; it <= B.max_as_iter; ++it) {
            as_iters = it;
            for (int i = tid; i < n; i += MCQ_NT) {
                const signed char s = ST[i];
                T1[i] = s == 0 ? 0.0 : (s < 0 ? LO[i] : (s == 1 ? HI[i] : 0.5 * (LO[i] + HI[i])));
            }
            gradient(c, T1, T0, T2);       // T2 = H x_A + f
            for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -T2[i] : T1[i];
            const int fs = factor(c, nullptr, ST);
            if (fs != 0) { status = fs; break; }
            solve(c, RHS);
            for (int i = tid; i < n; i += MCQ_NT) X[i] = ST[i] == 0 ? RHS[i] : T1[i];
            gradient(c, X, T0, G);
            // infeasibilities
            double nv = 0.0, imax = -1.0, kk = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) {
                const signed char s = ST[i];
                int v = 0;
                if (s == 0) { if (X[i] < LO[i] - TOLX) v = -1; else if (X[i] > HI[i] + TOLX) v = 1; kk = fmax(kk, fabs(G[i])); }
                else if (s == -1) { if (G[i] < -toly) v = 2; }
                else if (s == 1) { if (G[i] > toly) v = 2; }
                T3[i] = (double)v;
                if (v != 0) { nv += 1.0; imax = fmax(imax, (double)i); }
            }
            nv = block_reduce(nv, 0, red);
            imax = block_reduce(imax, 2, red);
            kkt = block_reduce(kk, 2, red);
            const int nvi = (int)nv;
            if (nvi == 0) { converged = true; break; }
            bool full;
            if (nvi < best) { best = nvi; pcnt = 3; full = true; }
            else if (pcnt > 0) { --pcnt; full = true; }
            else full = false;
            for (int i = tid; i < n; i += MCQ_NT) {
                const int v = (int)T3[i];
                if (v == 0) continue;
                if (!full && i != (int)imax) continue;
                ST[i] = v == 2 ? 0 : (signed char)v;
            }
            __syncthreads();
        }
        if (status == MCQ_OK && !converged) status = MCQ_ITER_CAP;

        // fp64 residual refinement through E on the final working set (same factor)
        if (status == MCQ_OK) {
            for (int r = 0; r < B.refine_steps; ++r) {
                for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -G[i] : 0.0;
                solve(c, RHS);
                for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) X[i] += RHS[i];
                gradient(c, X, T0, G);
            }
            double kk = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) kk = fmax(kk, fabs(G[i]));
            kkt = block_reduce(kk, 2, red);
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // outputs: alpha, curvature rows, opt_min_curv's curvature-error post-check
    // ---------------------------------------------------------------------------------------------------------------
    double nact = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        double a = X[i];
        a = fmin(fmax(a, LO[i]), HI[i]);
        X[i] = a;
        c.w.alpha[i] = a;
        if (ST[i] == -1 || ST[i] == 1) nact += 1.0;
    }
    nact = block_reduce(nact, 0, red);
    // kappa(alpha) = k_ref + E alpha
    band_matvec(c.w.Eb, c.d.bE, c.d.bR, n, X, VEC(c.w, nm, V_KREF), 1.0, T0);
    __syncthreads();
    double km = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) km = fmax(km, fabs(T0[i]));
    km = block_reduce(km, 2, red);
    if (status == MCQ_OK && B.check_kappa && km > kbound * (1.0 + 1e-9)) status = MCQ_KAPPA_ACTIVE;

    // curvature error: derivatives re-linearised at the solution
    {
        const double* XP = VEC(c.w, nm, V_XP);
        const double* YP = VEC(c.w, nm, V_YP);
        const double* XPP = VEC(c.w, nm, V_XPP);
        const double* YPP = VEC(c.w, nm, V_YPP);
        for (int i = tid; i < n; i += MCQ_NT) { T1[i] = c.w.nv[2 * i] * X[i]; T2[i] = c.w.nv[2 * i + 1] * X[i]; }
        __syncthreads();
        band_matvec(c.w.Db, c.d.bE, c.d.bR, n, T1, nullptr, 0.0, T0);   // D (n_x alpha)
        band_matvec(c.w.Db, c.d.bE, c.d.bR, n, T2, nullptr, 0.0, T3);   // D (n_y alpha)
        __syncthreads();
        double em = 0.0;
        for (int i = tid; i < n; i += MCQ_NT) {
            const int ip = cyc(i + 1, n);
            const double s = c.w.sc ? c.w.sc[i] : 1.0;
            const double s2 = s * s;
            const double xpt = XP[i] + (T1[ip] - T1[i]) - (T0[i] + 0.5 * s2 * T0[ip]) / 3.0;
            const double ypt = YP[i] + (T2[ip] - T2[i]) - (T3[i] + 0.5 * s2 * T3[ip]) / 3.0;
            const double xpp = XPP[i] + T0[i], ypp = YPP[i] + T3[i];
            const double xp = XP[i], yp = YP[i];
            const double k0 = (xp * ypp - yp * xpp) / pow(xp * xp + yp * yp, 1.5);
            const double k1 = (xpt * ypp - ypt * xpp) / pow(xpt * xpt + ypt * ypt, 1.5);
            em = fmax(em, fabs(k1 - k0));
        }
        em = block_reduce(em, 2, red);
        if (tid == 0) {
            *c.w.curv_err = em;
            *c.w.status = status;
            if (c.w.info) {
                mcq_info o;
                o.ipm_iters = ipm_iters;
                o.as_iters = as_iters;
                o.n_active_box = (int)nact;
                o.n_active_kappa = 0;
                o.kappa_max = km;
                o.kkt_res = fscale > 0.0 ? kkt / fscale : kkt;
                *c.w.info = o;
            }
        }
    }
}
