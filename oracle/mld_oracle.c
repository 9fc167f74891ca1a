/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, called from, or shipped with the product
 * path (pyhybridcontrol_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.
 *
 * Plain-C fp64 CPU restatement of the MPC hot path of michchr/pyhybridcontrol:
 *   (1) condensing an MLD model over the horizon
 *         controllers/components/mld_evolution_matrices.py:137-244, 253-332, 467-527
 *         utils/matrix_utils.py:55-81, 117-161
 *   (2) the constraint right-hand side  h = H_x x_k + H_omega omega + H_5
 *         controllers/controller_base.py:446-450
 *   (3) the mixed-integer solve the reference delegates to cvxpy -> Gurobi
 *         controllers/controller_base.py:509 ; variable layout controllers/components/variables.py:189-243
 *       Gurobi is third-party, proprietary, un-pinned and absent from /root/reference; what is
 *       restated here is the published textbook algorithm: bounded dual simplex on a dense
 *       dictionary with dual devex pricing (Forrest/Goldfarb 1992) and a Harris ratio test (Chvatal 1983;
 *       Harris 1973), Gomory mixed-integer cut
 *       rounds at the root (Gomory 1960; Balas/Ceria/Cornuejols/Natraj 1996) together with complemented
 *       mixed-integer rounding cuts on the original rows (Marchand/Wolsey 2001), row-activity bound
 *       propagation (Savelsbergh 1994), depth-first LP-based branch-and-bound (Land/Doig 1960) with
 *       iterative deepening on the bound (Korf 1985), a look-ahead dive + RINS (Danna/Rothberg/Le Pape
 *       2005) for degenerate instances, reduced-cost fixing, and convex-QP node relaxations by
 *       simplicial decomposition (von Hohenbalken 1977).
 *
 * Pinning: (1),(2) against golden vectors produced by the reference itself (tests/golden/, made by
 * oracle/gen_golden.py); (3) has no reference fixture (the reference has no tests) -> "parity
 * unpinned" at the solver boundary; it is cross-checked against exhaustive enumeration and
 * scipy.optimize.milp (HiGHS) in tests/test_oracle_solver.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_BIG 1.0e7
#define ORC_RESTART_ROWS 0        /* cut rows only the root restart may use (csrc/problem.inc S_RESTART_ROWS: 0 -- the restart takes what the root cut loop left) */
#define ORC_RESTART_ROUNDS 5       /* cut rounds of a root restart */
#define ORC_RESTART_GAPS 3.0      /* the restart runs when the incumbent is within this many gap tolerances of the proven bound (MIPGap 1e-2: 3 %) ... */
#define ORC_PTOL 1e-8
#define ORC_PTOL_SKIP 1e-6
#define ORC_DTOL 1e-9
#define ORC_CUT_PATIENCE 10     /* cut rounds without progress of the bound before the cut loop gives up (round 3: was 2; csrc/problem.inc S_CUT_PATIENCE).
                                * Paired runs, 2048 bench + 256 steady-state instances, patience 2 -> 10: row updates -12 % / -1 %, per-instance geometric mean -3 % on both,
                                * node-limited 5 -> 1 and 10 -> 5; environment ORC_PATIENCE overrides (study) */
#ifndef ORC_EAGER_START
#define ORC_EAGER_START 1      /* a MIP start is evaluated BEFORE the root LP (1, round 4) or only when the deepening passes end without an incumbent (0, round 3) */
#endif
#ifndef ORC_PSC_DEFAULT
#define ORC_PSC_DEFAULT 0      /* pseudocost branching: 0 off, k >= 1 = a direction's pseudocost is used once it has k observations (environment ORC_PSC) */
#endif
#ifndef ORC_BFRT_DEFAULT
#define ORC_BFRT_DEFAULT 1     /* long-step (bound flipping) dual ratio test: 0 off, 1 root LP only (csrc/problem.inc), 2 + cut rounds, 3 everywhere; environment ORC_BFRT overrides (study) */
#endif
#define ORC_PIV_ABS 1e-7
#define ORC_PIV_REL 1e-7
#define ORC_PIV_TINY 1e-5
#define ORC_INTTOL 1e-6
#define ORC_COEF_ZERO 1e-9
#define ORC_PURGE_SLACK 1e-3
#define ORC_RESID_TOL 1e-6
#define ORC_MIR_FMIN 0.005
#ifndef ORC_PEN_DEFAULT
#define ORC_PEN_DEFAULT 1
#endif

enum { ORC_OPTIMAL = 0, ORC_INFEASIBLE = 1, ORC_NODE_LIMIT = 2, ORC_NUMERICAL = 3, ORC_UNBOUNDED = 4 };
enum { LP_OPTIMAL = 0, LP_INFEASIBLE = 1, LP_CUTOFF = 2, LP_ITERLIMIT = 3 };

typedef struct {
    int nx, nu, ndelta, nz, nmu, nomega, ny, nc, nu_l, nmu_l;
} orc_dims;

typedef struct {
    double gap_abs, gap_rel;
    int max_nodes, cut_rounds, cuts_per_round, max_cuts, max_pivots, presolve;
    int mir_per_round;   /* c-MIR cuts on the original rows per cut round (0 = off) */
} orc_opts;

typedef struct {
    int nodes, pivots, cuts, refactors, status;
    double root_lp, root_bound, lower_bound;
    double work;         /* row updates summed over the pivots (rows with a non-zero multiplier): the proxy for the dense kernel's HBM bytes */
    double bland;        /* pivots taken under Bland's rule (the solve was stalling) */
    double rebuilds;     /* root rebuilt without cuts (the LP broke down in a cut round) */
    double phase_work[6];/* work split: root LP, cut rounds (+ MIP start), then the search phases IDS / DIVE / RINS / FINAL */
    double flips;        /* non-basic variables moved to their other bound by the long-step ratio test (no pivot) */
} orc_stats;

/* ------------------------------------------------------------------------------------------------
 * (1) condensing.  All matrices row-major.  Outputs may be NULL.  v = [u; delta; z; mu] per step.
 * ---------------------------------------------------------------------------------------------- */
static void matmul_acc(double *C, const double *A, const double *B, int m, int k, int n, int ldc, double alpha)
{ /* C[m x n] (ld ldc) += alpha * A[m x k] * B[k x n] */
    for (int i = 0; i < m; ++i)
        for (int p = 0; p < k; ++p) {
            double a = alpha * A[i * k + p];
            if (a == 0.0) continue;
            for (int j = 0; j < n; ++j) C[i * ldc + j] += a * B[p * n + j];
        }
}

int orc_condense(const orc_dims *d, int N, const double *A, const double *B1, const double *B2, const double *B3,
                 const double *B4, const double *b5, const double *C, const double *D1, const double *D2,
                 const double *D3, const double *D4, const double *d5, const double *E, const double *F1,
                 const double *F2, const double *F3, const double *F4, const double *f5, const double *G,
                 const double *Psi, double *Phi_x, double *Gamma_v, double *Gamma_w, double *Gamma_5, double *L_x,
                 double *L_v, double *L_w, double *L_5, double *H_x, double *H_v, double *H_w, double *H_5)
{
    const int nx = d->nx, nu = d->nu, nd = d->ndelta, nz = d->nz, nmu = d->nmu, nw = d->nomega, ny = d->ny,
              nc = d->nc;
    const int nv = nu + nd + nz + nmu;
    /* hstacks  Bv=[B1 B2 B3 0], Dv=[D1 D2 D3 0], Fv=[F1 F2 F3 Psi]  (mld_evolution_matrices.py:291,355,411) */
    double *Bv = calloc((size_t)nx * nv + 1, sizeof(double)), *Dv = calloc((size_t)ny * nv + 1, sizeof(double)),
           *Fv = calloc((size_t)nc * nv + 1, sizeof(double));
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nu; ++j) Bv[i * nv + j] = B1 ? B1[i * nu + j] : 0;
        for (int j = 0; j < nd; ++j) Bv[i * nv + nu + j] = B2 ? B2[i * nd + j] : 0;
        for (int j = 0; j < nz; ++j) Bv[i * nv + nu + nd + j] = B3 ? B3[i * nz + j] : 0;
    }
    for (int i = 0; i < ny; ++i) {
        for (int j = 0; j < nu; ++j) Dv[i * nv + j] = D1 ? D1[i * nu + j] : 0;
        for (int j = 0; j < nd; ++j) Dv[i * nv + nu + j] = D2 ? D2[i * nd + j] : 0;
        for (int j = 0; j < nz; ++j) Dv[i * nv + nu + nd + j] = D3 ? D3[i * nz + j] : 0;
    }
    for (int i = 0; i < nc; ++i) {
        for (int j = 0; j < nu; ++j) Fv[i * nv + j] = F1 ? F1[i * nu + j] : 0;
        for (int j = 0; j < nd; ++j) Fv[i * nv + nu + j] = F2 ? F2[i * nd + j] : 0;
        for (int j = 0; j < nz; ++j) Fv[i * nv + nu + nd + j] = F3 ? F3[i * nz + j] : 0;
        for (int j = 0; j < nmu; ++j) Fv[i * nv + nu + nd + nz + j] = Psi ? Psi[i * nmu + j] : 0;
    }
    /* per-lag blocks: Ak = A^k ; ABv[k] = A^k Bv ; ABw[k] = A^k B4 ; Ab5[k] = A^k b5 ; S5[k]=sum_{j<k} A^j b5 */
    double *Ak = calloc((size_t)N * nx * nx + 1, sizeof(double));
    double *ABv = calloc((size_t)N * nx * nv + 1, sizeof(double));
    double *ABw = calloc((size_t)N * nx * nw + 1, sizeof(double));
    double *S5 = calloc((size_t)(N + 1) * nx + 1, sizeof(double));
    for (int i = 0; i < nx; ++i) Ak[i * nx + i] = 1.0;
    for (int k = 1; k < N; ++k) /* A^k = A^{k-1} A   (mld_evolution_matrices.py:272 accumulate x@y) */
        if (A) matmul_acc(Ak + (size_t)k * nx * nx, Ak + (size_t)(k - 1) * nx * nx, A, nx, nx, nx, nx, 1.0);
    for (int k = 0; k < N; ++k) {
        matmul_acc(ABv + (size_t)k * nx * nv, Ak + (size_t)k * nx * nx, Bv, nx, nx, nv, nv, 1.0);
        if (B4) matmul_acc(ABw + (size_t)k * nx * nw, Ak + (size_t)k * nx * nx, B4, nx, nx, nw, nw, 1.0);
        /* Gamma_5 block row k = sum_{j=0}^{k-1} A^j b5  (toeplitz([0,A^0 b5,...]) @ ones, :331-332) */
        for (int i = 0; i < nx; ++i) S5[(k + 1) * nx + i] = S5[k * nx + i];
        if (b5) matmul_acc(S5 + (size_t)(k + 1) * nx, Ak + (size_t)k * nx * nx, b5, nx, nx, 1, 1, 1.0);
    }
    const int n = N * nv, nW = N * nw;
    /* state maps */
    if (Phi_x) memcpy(Phi_x, Ak, sizeof(double) * (size_t)N * nx * nx);
    if (Gamma_v) memset(Gamma_v, 0, sizeof(double) * (size_t)N * nx * n);
    if (Gamma_w) memset(Gamma_w, 0, sizeof(double) * (size_t)N * nx * nW);
    for (int i = 1; i < N; ++i)
        for (int j = 0; j < i; ++j) { /* block (i,j) = A^{i-j-1} B */
            const int lag = i - j - 1;
            if (Gamma_v)
                for (int r = 0; r < nx; ++r)
                    memcpy(Gamma_v + ((size_t)(i * nx + r)) * n + j * nv, ABv + ((size_t)lag * nx + r) * nv,
                           sizeof(double) * nv);
            if (Gamma_w)
                for (int r = 0; r < nx; ++r)
                    memcpy(Gamma_w + ((size_t)(i * nx + r)) * nW + j * nw, ABw + ((size_t)lag * nx + r) * nw,
                           sizeof(double) * nw);
        }
    if (Gamma_5)
        for (int k = 0; k < N; ++k)
            for (int r = 0; r < nx; ++r) Gamma_5[k * nx + r] = S5[k * nx + r];
    /* output and constraint maps, block by block:
     *   L(i,j)  = C Gamma(i,j) + [i==j] Dv           (:186-189)
     *   H_v(i,j)= E Gamma(i,j) + [i==j] Fv + G L(i,j) (:237-240)  etc. */
    double *Lb = calloc((size_t)ny * (nv + nw + nx + 1) + 1, sizeof(double));
    double *Hb = calloc((size_t)nc * (nv + nw + nx + 1) + 1, sizeof(double));
    if (L_v) memset(L_v, 0, sizeof(double) * (size_t)N * ny * n);
    if (L_w) memset(L_w, 0, sizeof(double) * (size_t)N * ny * nW);
    if (H_v) memset(H_v, 0, sizeof(double) * (size_t)N * nc * n);
    if (H_w) memset(H_w, 0, sizeof(double) * (size_t)N * nc * nW);
    for (int lag = -1; lag < N - 1; ++lag) { /* lag=-1 is the diagonal block (Gamma block = 0) */
        /* v part */
        memset(Lb, 0, sizeof(double) * (size_t)ny * nv);
        memset(Hb, 0, sizeof(double) * (size_t)nc * nv);
        if (lag >= 0) {
            if (C) matmul_acc(Lb, C, ABv + (size_t)lag * nx * nv, ny, nx, nv, nv, 1.0);
            if (E) matmul_acc(Hb, E, ABv + (size_t)lag * nx * nv, nc, nx, nv, nv, 1.0);
        } else {
            for (int t = 0; t < ny * nv; ++t) Lb[t] += Dv[t];
            for (int t = 0; t < nc * nv; ++t) Hb[t] += Fv[t];
        }
        if (G) matmul_acc(Hb, G, Lb, nc, ny, nv, nv, 1.0);
        for (int i = lag + 1; i < N; ++i) {
            const int j = i - lag - 1;
            if (L_v)
                for (int r = 0; r < ny; ++r)
                    memcpy(L_v + ((size_t)(i * ny + r)) * n + j * nv, Lb + (size_t)r * nv, sizeof(double) * nv);
            if (H_v)
                for (int r = 0; r < nc; ++r)
                    memcpy(H_v + ((size_t)(i * nc + r)) * n + j * nv, Hb + (size_t)r * nv, sizeof(double) * nv);
        }
        /* omega part (sign flipped for H, :239) */
        memset(Lb, 0, sizeof(double) * (size_t)ny * nw);
        memset(Hb, 0, sizeof(double) * (size_t)nc * nw);
        if (lag >= 0) {
            if (C) matmul_acc(Lb, C, ABw + (size_t)lag * nx * nw, ny, nx, nw, nw, 1.0);
            if (E) matmul_acc(Hb, E, ABw + (size_t)lag * nx * nw, nc, nx, nw, nw, 1.0);
        } else {
            if (D4) for (int t = 0; t < ny * nw; ++t) Lb[t] += D4[t];
            if (F4) for (int t = 0; t < nc * nw; ++t) Hb[t] += F4[t];
        }
        if (G) matmul_acc(Hb, G, Lb, nc, ny, nw, nw, 1.0);
        for (int i = lag + 1; i < N; ++i) {
            const int j = i - lag - 1;
            if (L_w)
                for (int r = 0; r < ny; ++r)
                    memcpy(L_w + ((size_t)(i * ny + r)) * nW + j * nw, Lb + (size_t)r * nw, sizeof(double) * nw);
            if (H_w)
                for (int r = 0; r < nc; ++r)
                    for (int c = 0; c < nw; ++c) H_w[((size_t)(i * nc + r)) * nW + j * nw + c] = -Hb[(size_t)r * nw + c];
        }
    }
    /* x and constant columns, one block row at a time */
    for (int k = 0; k < N; ++k) {
        memset(Lb, 0, sizeof(double) * (size_t)ny * (nx + 1));
        memset(Hb, 0, sizeof(double) * (size_t)nc * (nx + 1));
        if (C) matmul_acc(Lb, C, Ak + (size_t)k * nx * nx, ny, nx, nx, nx, 1.0); /* L_x = C A^k */
        if (E) matmul_acc(Hb, E, Ak + (size_t)k * nx * nx, nc, nx, nx, nx, 1.0);
        if (G) matmul_acc(Hb, G, Lb, nc, ny, nx, nx, 1.0);
        for (int r = 0; r < ny && L_x; ++r) memcpy(L_x + ((size_t)(k * ny + r)) * nx, Lb + (size_t)r * nx, sizeof(double) * nx);
        for (int r = 0; r < nc && H_x; ++r)
            for (int c = 0; c < nx; ++c) H_x[((size_t)(k * nc + r)) * nx + c] = -Hb[(size_t)r * nx + c];
        double *l5 = Lb + (size_t)ny * nx, *h5 = Hb + (size_t)nc * nx;
        for (int r = 0; r < ny; ++r) l5[r] = d5 ? d5[r] : 0.0;
        if (C) matmul_acc(l5, C, S5 + (size_t)k * nx, ny, nx, 1, 1, 1.0);
        for (int r = 0; r < nc; ++r) h5[r] = 0.0;
        if (E) matmul_acc(h5, E, S5 + (size_t)k * nx, nc, nx, 1, 1, 1.0);
        if (G) matmul_acc(h5, G, l5, nc, ny, 1, 1, 1.0);
        for (int r = 0; r < ny && L_5; ++r) L_5[k * ny + r] = l5[r];
        for (int r = 0; r < nc && H_5; ++r) H_5[k * nc + r] = (f5 ? f5[r] : 0.0) - h5[r];
    }
    free(Bv); free(Dv); free(Fv); free(Ak); free(ABv); free(ABw); free(S5); free(Lb); free(Hb);
    return 0;
}

/* (2)  h = H_x x0 + H_w w + H_5  (controller_base.py:446-450) */
void orc_rhs(int m, int nx, int nW, const double *H_x, const double *H_w, const double *H_5, const double *x0,
             const double *w, double *h)
{
    for (int i = 0; i < m; ++i) {
        double s = H_5[i];
        for (int j = 0; j < nx; ++j) s += H_x[(size_t)i * nx + j] * x0[j];
        for (int j = 0; j < nW; ++j) s += H_w[(size_t)i * nW + j] * w[j];
        h[i] = s;
    }
}

/* ------------------------------------------------------------------------------------------------
 * (3) MILP by cut-and-branch on one dense dictionary.
 *     xB[r] = D[r][n] - sum_c D[r][c] xN[c]   (column n holds beta);  row mcap is the cost row.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n, m0, m, mcap, ld, ntot;
    int cut_cap;        /* rows (original + cut) the cut generators may fill right now: m0 + max_cuts in the root cut loop, mcap in the root restart */
    double *D;          /* (mcap+1) x ld ; row mcap = reduced costs d (col n = z0) */
    double *Gx, *hx;    /* scaled original rows + cuts: mcap x n, mcap */
    double *q, *rs, *cs;
    double *lo, *hi;    /* ntot */
    double *clo, *chi;  /* n: bounds the CUT builders and the dead-row test may use for the continuous structurals (implied bounds of the presolve; == lo / hi without it) */
    double *xB, *xN;
    int *basic, *nonbasic, *where; /* where[id] = column index if nonbasic else -1-row */
    unsigned char *at_upper, *is_int, *skip;
    long pivots, max_pivots;
    double work, bland, flips;
    int bfrt_on;          /* long-step ratio test active (study: ORC_BFRT=1 root LP only, 2 root LP and cut rounds, 3 everywhere) */
    int perturbed;        /* the cost row carries the anti-stalling perturbation (dual_simplex_impl) */
    int refactors;
    double *tmp_col, *tmp_row;
    double *dw;           /* dual devex reference weights of the rows (Forrest & Goldfarb 1992), reset at every dual simplex entry */
    /* convex-QP relaxations by simplicial decomposition (P != NULL) */
    const double *P;      /* scaled Hessian n x n (NULL = linear cost) */
    double *Y, *PY, *Hm, *cm, *wm, *gcost, *vcur, *Pv;
    int pmax, qp_iters;
    int *partner;         /* study only (ORC_ELASTIC_STATS): row of the soft constraint a singleton penalty column belongs to, or -1 */
} dict_t;

/* ---- study: what implicit soft-constraint slacks would save.  A penalty column mu_i appears in ONE row with a negative
 * coefficient (E x + ... - mu_i <= f_i, cost c_i >= 0, mu_i >= 0): it and the row's slack s_i are never both basic, their dictionary
 * columns are negatives of each other when both are non-basic, and while exactly one of them is basic the other's column is a unit vector
 * (entry in the partner's row only, reduced cost exactly c_i) -- it needs no storage, and a pivot that exchanges the two partners touches
 * one row.  Counted per pivot when ORC_ELASTIC_STATS is set: pivots, partner-exchange pivots, penalty columns with an implicit (unit)
 * column, penalty columns whose partner is non-basic too (stored once for the pair), penalty columns in the model. */
static double g_el[5];
static int g_el_on = -1;
void orc_elastic_stats(double out[5]) { for (int k = 0; k < 5; ++k) { out[k] = g_el[k]; g_el[k] = 0.0; } }

static double *dalloc(size_t k) { return (double *)calloc(k + 1, sizeof(double)); }

static void equilibrate(const double *G, int m, int n, const unsigned char *is_int, double *rs, double *cs)
{
    for (int i = 0; i < m; ++i) rs[i] = 1.0;
    for (int j = 0; j < n; ++j) cs[j] = 1.0;
    double *cmax = dalloc(n);
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = 0; i < m; ++i) {
            double mx = 0;
            for (int j = 0; j < n; ++j) { double a = fabs(G[(size_t)i * n + j]) * rs[i] * cs[j]; if (a > mx) mx = a; }
            if (mx > 0) rs[i] /= mx;
        }
        for (int j = 0; j < n; ++j) cmax[j] = 0;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) { double a = fabs(G[(size_t)i * n + j]) * rs[i] * cs[j]; if (a > cmax[j]) cmax[j] = a; }
        for (int j = 0; j < n; ++j) if (!is_int[j] && cmax[j] > 0) cs[j] /= cmax[j];
    }
    for (int i = 0; i < m; ++i) rs[i] = exp2(rint(log2(rs[i])));
    for (int j = 0; j < n; ++j) cs[j] = exp2(rint(log2(cs[j])));
    free(cmax);
}

/* Per-instance presolve (opts.presolve bit 2; csrc/problem.inc s_presolve runs the same passes): row-activity bound propagation with integer
 * rounding (Savelsbergh 1994, section 3.2) on the SCALED rows of this instance -- the right-hand side carries the instance's x0 and omega, so a
 * bound that no per-model tightening can know follows here (the grid power of a step lies between omega_k and omega_k + sum P_i: z_k, free in the
 * model, gets finite bounds; a step whose omega_k decides the sign fixes its delta_k).  Jacobi passes: every pass reads the bounds of the pass
 * before (row statistics first, then every column takes the tightest bound its rows imply), so the result does not depend on the order rows are
 * visited in -- a thread per row / per column on the GPU computes the same numbers.  The implied bounds of the CONTINUOUS structurals go to
 * clo / chi only: the LP keeps its own (a free z stays free and basic -- making the implied bounds LP bounds doubles the dictionary's fill and
 * was measured slower), the c-MIR builder substitutes them (a row with a free continuous entry was unusable before) and the dead-row test adds
 * them up.  Binaries the propagation fixes are fixed in the LP.  Returns 0 when the rows are infeasible under the bounds. */
#define ORC_PRE_PASSES 8
static int presolve_instance(dict_t *t, int m0)
{
    const int n = t->n;
    double *tot = dalloc(m0), *nlo = dalloc(n), *nhi = dalloc(n);
    int *meta = (int *)calloc(m0 + 1, sizeof(int));      /* -1: every term finite, >= 0: the one column with an infinite term, -2: more than one */
    int ok = 1;
    for (int pass = 0; pass < ORC_PRE_PASSES && ok; ++pass) {
        for (int i = 0; i < m0; ++i) {
            const double *g = t->Gx + (size_t)i * n;
            double s = 0.0; int ninf = 0, jinf = -1;
            for (int j = 0; j < n; ++j) {
                if (g[j] == 0.0) continue;
                const double b = g[j] > 0.0 ? t->clo[j] : t->chi[j];
                if (fabs(b) >= 0.5 * ORC_BIG) { ninf++; jinf = j; } else s += g[j] * b;
            }
            tot[i] = s; meta[i] = ninf == 0 ? -1 : (ninf == 1 ? jinf : -2);
            if (t->hx[i] < 1.0e29 && ninf == 0 && s > t->hx[i] + 1e-6 * fmax(1.0, fabs(t->hx[i]))) ok = 0;
        }
        if (!ok) break;
        int changed = 0;
        for (int j = 0; j < n; ++j) {
            double lo = t->clo[j], hi = t->chi[j];
            const double lo0 = lo, hi0 = hi;
            for (int i = 0; i < m0; ++i) {
                const double gj = t->Gx[(size_t)i * n + j];
                if (gj == 0.0 || t->hx[i] >= 1.0e29) continue;
                const int mt = meta[i];
                if (mt == -2 || (mt >= 0 && mt != j)) continue;
                const double rest = mt == j ? tot[i] : tot[i] - gj * (gj > 0.0 ? lo0 : hi0);
                double b = (t->hx[i] - rest) / gj;
                if (gj > 0.0) {
                    if (t->is_int[j]) b = floor(b + 1e-6);
                    if (b < hi - 1e-9 * fmax(1.0, fabs(b))) hi = b;
                } else {
                    if (t->is_int[j]) b = ceil(b - 1e-6);
                    if (b > lo + 1e-9 * fmax(1.0, fabs(b))) lo = b;
                }
            }
            if (lo > hi + 1e-6 * fmax(1.0, fabs(lo))) ok = 0;
            if (hi < lo) { if (hi0 == hi) lo = hi; else hi = lo; }      /* (round-off: the bound that moved gives way) */
            if (lo != lo0 || hi != hi0) changed = 1;
            nlo[j] = lo; nhi[j] = hi;
        }
        memcpy(t->clo, nlo, sizeof(double) * n); memcpy(t->chi, nhi, sizeof(double) * n);
        if (!changed) break;
    }
    if (ok) for (int j = 0; j < n; ++j) if (t->is_int[j]) { t->lo[j] = t->clo[j]; t->hi[j] = t->chi[j]; }
    free(tot); free(nlo); free(nhi); free(meta);
    return ok;
}

/* Rows that cannot bind under the ROOT bounds (largest activity <= right-hand side): their slack is basic in the slack basis,
 * never violates its bound and so never leaves; nothing reads such a row again (cuts substitute rows of basic STRUCTURALS,
 * verification goes to the original rows), so the pivots need not maintain it: bit 1 of skip[].  Bounds only tighten below
 * the root.  Not with a quadratic cost: the primal simplex of the QP relaxations ratio-tests every row. */
static void mark_dead(dict_t *t, int m0, int enable)
{
    const int n = t->n;
    memset(t->skip, 0, (size_t)t->mcap);
    if (!enable) return;
    for (int i = 0; i < m0; ++i) {
        const double *g = t->Gx + (size_t)i * n;
        double act = 0.0; int inf = 0;
        for (int j = 0; j < n; ++j) {
            const double gj = g[j];
            if (gj > 0.0) { if (t->chi[j] < 0.5 * ORC_BIG) act += gj * t->chi[j]; else inf = 1; }
            else if (gj < 0.0) { if (t->clo[j] > -0.5 * ORC_BIG) act += gj * t->clo[j]; else inf = 1; }
        }
        if (!inf && act <= t->hx[i] - 1e-7) t->skip[i] = 2;
    }
}

static void reset_dictionary(dict_t *t)
{
    const int n = t->n, ld = t->ld;
    for (int i = 0; i < t->mcap; ++i) {
        memcpy(t->D + (size_t)i * ld, t->Gx + (size_t)i * n, sizeof(double) * n);
        t->D[(size_t)i * ld + n] = t->hx[i];
    }
    memcpy(t->D + (size_t)t->mcap * ld, t->q, sizeof(double) * n);
    t->D[(size_t)t->mcap * ld + n] = 0.0;
    for (int i = 0; i < t->mcap; ++i) { t->basic[i] = n + i; t->where[n + i] = -1 - i; }
    for (int j = 0; j < n; ++j) { t->nonbasic[j] = j; t->where[j] = j; }
}

static void refresh(dict_t *t)
{
    const int n = t->n, ld = t->ld;
    for (int r = 0; r < t->m; ++r) {
        const double *row = t->D + (size_t)r * ld;
        double s = row[n];
        for (int c = 0; c < n; ++c) s -= row[c] * t->xN[c];
        t->xB[r] = s;
    }
}

static double objective(const dict_t *t)
{
    const double *d = t->D + (size_t)t->mcap * t->ld;
    double s = d[t->n];
    for (int c = 0; c < t->n; ++c) s += d[c] * t->xN[c];
    return s;
}

static void place(dict_t *t, int c)
{
    const int j = t->nonbasic[c];
    const double dc = t->D[(size_t)t->mcap * t->ld + c];
    if (t->lo[j] == t->hi[j]) { t->at_upper[c] = 0; t->xN[c] = t->lo[j]; }
    else if (dc >= 0) { if (!isfinite(t->lo[j])) t->lo[j] = -ORC_BIG; t->at_upper[c] = 0; t->xN[c] = t->lo[j]; }
    else { if (!isfinite(t->hi[j])) t->hi[j] = ORC_BIG; t->at_upper[c] = 1; t->xN[c] = t->hi[j]; }
}

static void set_bounds(dict_t *t, int j, double lo, double hi)
{
    t->lo[j] = lo; t->hi[j] = hi;
    const int c = t->where[j];
    if (c < 0) return;
    const double old = t->xN[c];
    const double dc = t->D[(size_t)t->mcap * t->ld + c];
    double nw;
    if (lo == hi) { nw = lo; t->at_upper[c] = 0; }
    else if (dc >= 0) { nw = lo; t->at_upper[c] = 0; }
    else { nw = hi; t->at_upper[c] = 1; }
    if (nw != old) {
        const double dl = nw - old;
        for (int r = 0; r < t->m; ++r) t->xB[r] -= t->D[(size_t)r * t->ld + c] * dl;
        t->xN[c] = nw;
    }
}

static void pivot(dict_t *t, int r, int c, double leave_value)
{
    const int n = t->n, ld = t->ld, m = t->m;
    double *D = t->D;
    double *rowr = D + (size_t)r * ld;
    const double p = rowr[c];
    const double theta = (t->xB[r] - leave_value) / p;
    double *colc = t->tmp_col;
    for (int i = 0; i < m; ++i) colc[i] = (t->skip[i] & 2) ? 0.0 : D[(size_t)i * ld + c];   /* rows that can never bind are not maintained */
    for (int i = 0; i < m; ++i) t->xB[i] -= colc[i] * theta;
    const double enter_val = t->xN[c] + theta;
    const double inv = 1.0 / p;
    {   /* devex: w_i = max(w_i, (alpha_iq / alpha_rq)^2 w_r) on the rows the pivot touches, w_r = max(w_r / alpha_rq^2, 1) */
        const double wr = t->dw[r];
        for (int i = 0; i < m; ++i) { if (i == r || colc[i] == 0.0) continue; const double f = colc[i] * inv, c2 = f * f * wr; if (c2 > t->dw[i]) t->dw[i] = c2; }
        t->dw[r] = fmax(wr * inv * inv, 1.0);
    }
    for (int k = 0; k <= n; ++k) rowr[k] *= inv;
    rowr[c] = inv;
    colc[r] = 0.0;
    for (int i = 0; i < m; ++i) {
        if (i == r) continue;
        const double mu = colc[i];
        if (mu == 0.0) continue;
        t->work += 1.0;
        double *ri = D + (size_t)i * ld;
        for (int k = 0; k <= n; ++k) ri[k] -= mu * rowr[k];
        ri[c] = -mu * inv;
    }
    { /* cost row: obj = d[n] + sum_c d_c xN_c ; substituting xN_c = rowr[n] - sum_k rowr[k] xN_k */
        double *d = D + (size_t)t->mcap * ld;
        const double dc = d[c];
        if (dc != 0.0) {
            for (int k = 0; k < n; ++k) d[k] -= dc * rowr[k];
            d[c] = -dc * inv;
            d[n] += dc * rowr[n];
        }
    }
    const int jb = t->basic[r], jn = t->nonbasic[c];
    if (t->partner) {
        double cheap = 0.0, implicit = 0.0, both = 0.0, pairs = 0.0;
        if (jn < n && t->partner[jn] >= 0 && jb == n + t->partner[jn]) cheap = 1.0;
        if (jb < n && t->partner[jb] >= 0 && jn == n + t->partner[jb]) cheap = 1.0;
        for (int j = 0; j < n; ++j) {
            if (t->partner[j] < 0) continue;
            pairs += 1.0;
            const int bj = t->where[j] < 0, bs = t->where[n + t->partner[j]] < 0;
            if (bj != bs) implicit += 1.0; else if (!bj) both += 1.0;
        }
#pragma omp critical(orc_elastic)
        { g_el[0] += 1.0; g_el[1] += cheap; g_el[2] += implicit; g_el[3] += both; g_el[4] += pairs; }
    }
    t->basic[r] = jn; t->nonbasic[c] = jb;
    t->where[jn] = -1 - r; t->where[jb] = c;
    t->xB[r] = enter_val;
    t->xN[c] = leave_value;
    t->at_upper[c] = (leave_value == t->hi[jb]) && (t->lo[jb] != t->hi[jb]);
    t->pivots++;
}

static double check_residual(const dict_t *t)
{
    const int n = t->n;
    double *x = t->tmp_row;
    for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) x[t->nonbasic[c]] = t->xN[c];
    for (int r = 0; r < t->m; ++r) if (t->basic[r] < n) x[t->basic[r]] = t->xB[r];
    double worst = 0;
    for (int i = 0; i < t->m; ++i) {
        if (t->skip[i] & 2) continue;
        const double *g = t->Gx + (size_t)i * n;
        double s = t->hx[i];
        for (int j = 0; j < n; ++j) s -= g[j] * x[j];
        const int w = t->where[n + i];
        const double v = (w >= 0) ? t->xN[w] : t->xB[-1 - w];
        const double e = fabs(s - v);
        if (e > worst) worst = e;
    }
    return worst;
}

static void refactor(dict_t *t)
{
    t->perturbed = 0;     /* the dictionary is rebuilt with the true cost row */
    for (int i = 0; i < t->mcap; ++i) t->dw[i] = 1.0;
    const int n = t->n, m = t->m, ld = t->ld;
    t->refactors++;
    unsigned char *want = (unsigned char *)calloc(n + 1, 1);       /* structurals that must be basic */
    unsigned char *rowfree = (unsigned char *)calloc(m + 1, 1);    /* slacks that must be nonbasic */
    double *val = dalloc(t->ntot);
    unsigned char *up = (unsigned char *)calloc(t->ntot + 1, 1);
    int nwant = 0;
    for (int r = 0; r < m; ++r) if (t->basic[r] < n) { want[t->basic[r]] = 1; nwant++; }
    for (int c = 0; c < n; ++c) {
        const int j = t->nonbasic[c];
        val[j] = t->xN[c]; up[j] = t->at_upper[c];
        if (j >= n) rowfree[j - n] = 1;
    }
    reset_dictionary(t);
    for (int c = 0; c < n; ++c) t->xN[c] = 0;
    for (int r = 0; r < m; ++r) t->xB[r] = 0;
    const long saved = t->pivots;
    for (int k = 0; k < nwant; ++k) {
        double best = -1; int br = -1, bc = -1;
        for (int r = 0; r < m; ++r) {
            if (!rowfree[r]) continue;
            const double *row = t->D + (size_t)r * ld;
            for (int c = 0; c < n; ++c)
                if (want[c] && t->nonbasic[c] == c && fabs(row[c]) > best) { best = fabs(row[c]); br = r; bc = c; }
        }
        if (br < 0 || best <= 0) break;
        pivot(t, br, bc, 0.0);
        rowfree[br] = 0; want[bc] = 0;
    }
    t->pivots = saved;
    for (int c = 0; c < n; ++c) {
        const int j = t->nonbasic[c];
        if (j < n && want[j]) place(t, c);   /* could not be pivoted in (numerically singular basis): park at a bound */
        else { t->xN[c] = val[j]; t->at_upper[c] = up[j]; }
    }
    /* If a wanted variable could not be pivoted in, the basis is another one and reduced costs may have the wrong sign:
     * boxed columns move to the matching bound; a slack (or free) column cannot -- then the dictionary is restarted from
     * the slack basis with every column at its dual-feasible bound (the dual simplex re-solves from there). */
    {
        const double *dd = t->D + (size_t)t->mcap * ld;
        int bad = 0;
        for (int c = 0; c < n; ++c) {
            const int j = t->nonbasic[c];
            if (t->lo[j] == t->hi[j]) continue;
            const double viol = t->at_upper[c] ? dd[c] : -dd[c];
            if (viol <= 1e-7) continue;
            if (j < n) place(t, c); else bad = 1;
        }
        if (bad) {
            reset_dictionary(t);
            for (int c = 0; c < n; ++c) place(t, c);
        }
    }
    refresh(t);
    free(want); free(rowfree); free(val); free(up);
}

static void reprice(dict_t *t, const double *cost);
static int primal_simplex(dict_t *t);
#define LP_UNBOUNDED 4

/* Anti-stalling cost perturbation (the textbook remedy for dual degeneracy, e.g. Koberstein 2005 section 6.2): when a solve without an
 * objective cutoff stalls, every non-basic reduced cost is pushed away from zero in its feasible direction by a deterministic pseudo-random
 * amount of relative size 1e-6..2e-6, which breaks the ties of the dual ratio test.  When the perturbed problem is optimal the true cost row
 * is restored (reprice) and, if some reduced cost now has the wrong sign, the bounded primal simplex finishes from this primal-feasible basis. */
#define ORC_PERT 1e-6
static void perturb_costs(dict_t *t)
{
    const int n = t->n;
    double *d = t->D + (size_t)t->mcap * t->ld;
    for (int c = 0; c < n; ++c) {
        const int j = t->nonbasic[c];
        if (t->lo[j] == t->hi[j]) continue;
        const unsigned hsh = (unsigned)j * 2654435761u;
        const double u = (double)(hsh >> 8) * (1.0 / 16777216.0);          /* [0, 1) from the variable's id */
        const double eps = ORC_PERT * (1.0 + u) * (1.0 + fabs(d[c]));
        d[c] += t->at_upper[c] ? -eps : eps;
    }
    t->perturbed = 1;
}

/* leaves a perturbed solve: true reduced costs back, primal simplex if they are not dual feasible at an optimum */
static int unperturb(dict_t *t, int st)
{
    if (!t->perturbed) return st;
    reprice(t, t->q);
    t->perturbed = 0;
    if (st != LP_OPTIMAL) return st;
    const double *d = t->D + (size_t)t->mcap * t->ld;
    int bad = 0;
    for (int c = 0; c < t->n && !bad; ++c) {
        const int j = t->nonbasic[c];
        if (t->lo[j] == t->hi[j]) continue;
        if ((t->at_upper[c] ? d[c] : -d[c]) > ORC_DTOL) bad = 1;
    }
    if (!bad) return LP_OPTIMAL;
    const int ps = primal_simplex(t);
    return ps == LP_OPTIMAL ? LP_OPTIMAL : LP_ITERLIMIT;
}

static int dual_simplex_impl(dict_t *t, double cutoff)
{
    const int n = t->n, ld = t->ld;
    const int m = t->m;
    for (int i = 0; i < t->mcap; ++i) { t->skip[i] &= 2; t->dw[i] = 1.0; }     /* bit 1 (mark_dead) stays; new devex reference framework: the current basis */
    int stall = 0;
    double last_obj = -INFINITY;
    static int pert_env = -1;
    if (pert_env < 0) pert_env = getenv("ORC_NO_PERT") ? 0 : 1;
    int pert_ok = pert_env && !t->P;      /* one perturbation per solve; while the cost row is perturbed the objective cutoff is suspended (the caller compares the true value) */
    const double *d = t->D + (size_t)t->mcap * ld;
    long checked_at = t->pivots;     /* pivot count at the last verification against the original rows */
    for (;;) {
        if (t->pivots >= t->max_pivots) return unperturb(t, LP_ITERLIMIT);
        if (t->pivots - checked_at >= 512) {   /* a long solve never reaches the verification at an optimum: verify on the way */
            checked_at = t->pivots;
            if (check_residual(t) > ORC_RESID_TOL) { refactor(t); for (int i = 0; i < t->mcap; ++i) t->skip[i] &= 2; continue; }
        }
        const double cur = objective(t);
        if (cur > last_obj + 1e-12 * fmax(1.0, fabs(cur))) { stall = 0; last_obj = cur; } else stall++;
        if (stall > 30 && pert_ok && !t->perturbed) { pert_ok = 0; perturb_costs(t); stall = 0; last_obj = -INFINITY; continue; }
        const int bland = stall > 30;
        /* leaving row: dual devex pricing, largest violation^2 / weight (smallest variable id while stalling) */
        int r = -1; double best_sc = 0.0; int rb = -1; int rb_id = 0x7fffffff;
        for (int i = 0; i < m; ++i) {
            if (t->skip[i]) continue;
            const int j = t->basic[i];
            const double v = fmax(t->lo[j] - t->xB[i], t->xB[i] - t->hi[j]);
            if (v > ORC_PTOL) {
                const double sc = v * v / t->dw[i];
                if (sc > best_sc) { best_sc = sc; r = i; }
                if (j < rb_id) { rb_id = j; rb = i; }
            }
        }
        if (r < 0) {
            checked_at = t->pivots;
            if (check_residual(t) > ORC_RESID_TOL) { refactor(t); for (int i = 0; i < t->mcap; ++i) t->skip[i] &= 2; continue; }
            return unperturb(t, LP_OPTIMAL);
        }
        if (bland) r = rb;
        if (cur >= cutoff && !t->perturbed) {
            /* a cutoff is a claim about the bound: verify the dictionary first if it has moved since the last check */
            if (t->pivots - checked_at >= 64) {
                checked_at = t->pivots;
                if (check_residual(t) > ORC_RESID_TOL) { refactor(t); for (int i = 0; i < t->mcap; ++i) t->skip[i] &= 2; continue; }
            }
            return LP_CUTOFF;
        }
        const int jr = t->basic[r];
        const double vlo = t->lo[jr] - t->xB[r], vhi = t->xB[r] - t->hi[jr];
        const int below = vlo > vhi;
        const double viol = below ? vlo : vhi;
        const double *row = t->D + (size_t)r * ld;
        /* eligible columns and the row's largest eligible magnitude */
        double emax = 0;
        for (int c = 0; c < n; ++c) {
            const int j = t->nonbasic[c];
            if (t->lo[j] == t->hi[j]) continue;
            const double a = row[c];
            const int el = below ? (t->at_upper[c] ? a > 0 : a < 0) : (t->at_upper[c] ? a < 0 : a > 0);
            if (el && fabs(a) > emax) emax = fabs(a);
        }
        /* every eligible entry above the absolute floor takes part in the ratio test (excluding entries that are
           merely small RELATIVE to the row maximum lets their reduced costs change sign by O(1)) */
        const double ptol = fmax(ORC_PIV_ABS, ORC_PIV_REL * emax);
        double tmax = INFINITY, rmin = INFINITY;
        int any = 0;
        for (int c = 0; c < n; ++c) {
            const int j = t->nonbasic[c];
            if (t->lo[j] == t->hi[j]) continue;
            const double a = row[c];
            const int el = below ? (t->at_upper[c] ? a > 0 : a < 0) : (t->at_upper[c] ? a < 0 : a > 0);
            if (!el || fabs(a) <= ptol) continue;
            any = 1;
            const double da = fmax(t->at_upper[c] ? -d[c] : d[c], 0.0);
            const double r1 = (da + ORC_DTOL) / fabs(a), r0 = da / fabs(a);
            if (r1 < tmax) tmax = r1;
            if (r0 < rmin) rmin = r0;
        }
        if (!any) {
            if (viol <= ORC_PTOL_SKIP) { t->skip[r] |= 1; continue; }
            if (t->pivots > checked_at && viol <= 1e-3) {
                /* a SMALL violation without an eligible entry may be accumulated error (a degenerate basic variable
                 * drifting off its bound) rather than infeasibility: if the dictionary's discrepancy against the original
                 * rows is of the violation's size, re-derive it before believing the row */
                checked_at = t->pivots;
                if (check_residual(t) > 0.1 * viol) { refactor(t); for (int i = 0; i < t->mcap; ++i) t->skip[i] &= 2; continue; }
            }
            return unperturb(t, LP_INFEASIBLE);
        }
        int cbest = -1; double abest = -1; int idbest = 0x7fffffff;
        if (t->bfrt_on && !bland) {
            /* Long-step ("bound flipping") ratio test for boxed variables (Fourer 1994; Maros 2003): the dual objective rises along the step with slope =
             * the leaving row's violation; passing the breakpoint of a BOXED non-basic variable j lowers the slope by |a_rj| (hi_j - lo_j) and j simply
             * moves to its other bound -- no pivot.  Groups of breakpoints inside one Harris window are passed together while the slope stays positive;
             * the entering variable is the largest |a| of the first group that cannot be passed.  On for the root LP (ORC_BFRT_DEFAULT 1; ORC_BFRT=0 and the kernel's
             * opts.reserved bit 14 switch it off for an A/B) -- on the 2048 bench instances -15 % row updates (root LP -40 %), on the GPU -3 % bytes. */
            unsigned char *gone = (unsigned char *)t->tmp_row;
            memset(gone, 0, n);
            double slope = viol;
            for (;;) {
                double tm = INFINITY;
                for (int c = 0; c < n; ++c) {
                    if (gone[c]) continue;
                    const int j = t->nonbasic[c];
                    if (t->lo[j] == t->hi[j]) continue;
                    const double a = row[c];
                    const int el = below ? (t->at_upper[c] ? a > 0 : a < 0) : (t->at_upper[c] ? a < 0 : a > 0);
                    if (!el || fabs(a) <= ptol) continue;
                    const double da = fmax(t->at_upper[c] ? -d[c] : d[c], 0.0);
                    const double r1 = (da + ORC_DTOL) / fabs(a);
                    if (r1 < tm) tm = r1;
                }
                if (!(tm < INFINITY)) break;
                double drop = 0.0; int cb = -1; double ab = -1;
                for (int c = 0; c < n; ++c) {
                    if (gone[c]) continue;
                    const int j = t->nonbasic[c];
                    if (t->lo[j] == t->hi[j]) continue;
                    const double a = row[c];
                    const int el = below ? (t->at_upper[c] ? a > 0 : a < 0) : (t->at_upper[c] ? a < 0 : a > 0);
                    if (!el || fabs(a) <= ptol) continue;
                    const double da = fmax(t->at_upper[c] ? -d[c] : d[c], 0.0);
                    if (da / fabs(a) > tm) continue;
                    gone[c] |= 2;
                    const double range = (t->lo[j] > -0.5 * ORC_BIG && t->hi[j] < 0.5 * ORC_BIG) ? t->hi[j] - t->lo[j] : INFINITY;
                    drop += fabs(a) * range;
                    if (fabs(a) > ab) { ab = fabs(a); cb = c; }
                }
                cbest = cb;
                if (!(slope - drop > ORC_PTOL)) { for (int c = 0; c < n; ++c) gone[c] &= 1; break; }
                slope -= drop;
                for (int c = 0; c < n; ++c) if (gone[c] & 2) gone[c] = 1;
            }
            if (cbest >= 0) gone[cbest] = 0;
            for (int c = 0; c < n; ++c) {
                if (gone[c] != 1) continue;
                const int j = t->nonbasic[c];
                const double nw = t->at_upper[c] ? t->lo[j] : t->hi[j], dl = nw - t->xN[c];
                for (int i = 0; i < t->m; ++i) if (!(t->skip[i] & 2)) t->xB[i] -= t->D[(size_t)i * ld + c] * dl;
                t->xN[c] = nw; t->at_upper[c] = !t->at_upper[c];
                t->flips += 1.0;
            }
        } else
        for (int c = 0; c < n; ++c) {
            const int j = t->nonbasic[c];
            if (t->lo[j] == t->hi[j]) continue;
            const double a = row[c];
            const int el = below ? (t->at_upper[c] ? a > 0 : a < 0) : (t->at_upper[c] ? a < 0 : a > 0);
            if (!el || fabs(a) <= ptol) continue;
            const double da = fmax(t->at_upper[c] ? -d[c] : d[c], 0.0);
            const double r0 = da / fabs(a);
            if (bland) {
                if (r0 <= rmin * (1 + 1e-12) + 1e-300 && j < idbest) { idbest = j; cbest = c; }
            } else if (r0 <= tmax && fabs(a) > abest) { abest = fabs(a); cbest = c; }
        }
        if (fabs(row[cbest]) < ORC_PIV_TINY && viol <= ORC_PTOL_SKIP) {
            /* a violation within the skip tolerance whose only pivots are tiny: the dual step d_q / |a_q| would be huge and
             * the entries the ratio test ignored (|a| <= ptol) would carry it into their reduced costs -- dual feasibility,
             * and with it the bound, is lost (seen: violation 5e-8, pivot 1.7e-7, step 3e9).  The row counts as satisfied. */
            t->skip[r] |= 1; continue;
        }
        if (bland) t->bland += 1.0;
        pivot(t, r, cbest, below ? t->lo[jr] : t->hi[jr]);
    }
}


/* ---- primal side: re-pricing with a new cost vector, bounded primal simplex, simplicial decomposition ----------- */
#define SD_PMAX 32

/* reduced costs of the current basis for structural cost vector `cost` (slacks cost nothing) */
/* a single re-solve that needs more than four times the problem's size in pivots is cycling (the Harris tolerances void Bland's guarantee):
 * the dictionary is rebuilt from the original rows and the solve gets one more allowance (csrc/problem.inc s_dual_simplex) */
static int dual_simplex(dict_t *t, double cutoff)
{
    const long cap = 4L * (t->m0 + t->n) + 1000, save = t->max_pivots;
    t->max_pivots = save < t->pivots + cap ? save : t->pivots + cap;
    int st = dual_simplex_impl(t, cutoff);
    if (st == LP_ITERLIMIT && t->pivots < save) {
        refactor(t);
        t->max_pivots = save < t->pivots + cap ? save : t->pivots + cap;
        st = dual_simplex_impl(t, cutoff);
    }
    t->max_pivots = save;
    return st;
}

static void reprice(dict_t *t, const double *cost)
{
    const int n = t->n, ld = t->ld;
    double *d = t->D + (size_t)t->mcap * ld;
    for (int c = 0; c <= n; ++c) d[c] = 0.0;
    for (int c = 0; c < n; ++c) { const int j = t->nonbasic[c]; if (j < n) d[c] = cost[j]; }
    for (int r = 0; r < t->m; ++r) {
        const int j = t->basic[r];
        if (j >= n || cost[j] == 0.0) continue;
        const double cj = cost[j];
        const double *row = t->D + (size_t)r * ld;
        for (int c = 0; c < n; ++c) d[c] -= cj * row[c];
        d[n] += cj * row[n];
    }
}

/* bounded primal simplex from a primal-feasible basis (Dantzig pricing, Harris ratio test, Bland while stalling) */
static int primal_simplex(dict_t *t)
{
    const int n = t->n, ld = t->ld, m = t->m;
    const double *d = t->D + (size_t)t->mcap * ld;
    int stall = 0;
    double last_obj = INFINITY;
    for (;;) {
        if (t->pivots >= t->max_pivots) return LP_ITERLIMIT;
        const double cur = objective(t);
        if (cur < last_obj - 1e-12 * fmax(1.0, fabs(cur))) { stall = 0; last_obj = cur; } else stall++;
        /* cycling (round 4): Bland's rule on the entering side with Harris' ratio test is no guarantee, and this routine has no refactorisation of its own -- a
         * QP relaxation's return to the LP(q) vertex burnt 40 000 pivots at one objective value on a dictionary grown to 1e13 after a pivot of 1e-6.  The
         * caller re-derives the dictionary and goes on with the LP bound of the node (sd_relax) */
        if (stall > 2 * (n + m) + 500) return LP_ITERLIMIT;
        const int bland = stall > 30;
        int c = -1; double best = ORC_DTOL; int bid = 0x7fffffff;
        for (int k = 0; k < n; ++k) {
            const int j = t->nonbasic[k];
            if (t->lo[j] == t->hi[j]) continue;
            const double viol = t->at_upper[k] ? d[k] : -d[k];
            if (viol <= ORC_DTOL) continue;
            if (bland) { if (j < bid) { bid = j; c = k; } }
            else if (viol > best) { best = viol; c = k; }
        }
        if (c < 0) return LP_OPTIMAL;
        const int jc = t->nonbasic[c];
        const double dir = t->at_upper[c] ? -1.0 : 1.0;
        /* pass 1: Harris bound */
        double tmax = t->hi[jc] - t->lo[jc];
        double amax = 0;
        for (int i = 0; i < m; ++i) { if (t->skip[i] & 2) continue; const double a = fabs(t->D[(size_t)i * ld + c]); if (a > amax) amax = a; }
        const double ptol = fmax(ORC_PIV_ABS, ORC_PIV_REL * amax);
        for (int i = 0; i < m; ++i) {
            if (t->skip[i] & 2) continue;      /* rows that cannot bind / dropped cut rows are not maintained (linear-cost mode only) */
            const double a = t->D[(size_t)i * ld + c] * dir;
            if (fabs(a) <= ptol) continue;
            const int j = t->basic[i];
            const double room = a > 0 ? t->xB[i] - t->lo[j] : t->hi[j] - t->xB[i];
            const double r1 = (fmax(room, 0.0) + ORC_PTOL) / fabs(a);
            if (r1 < tmax) tmax = r1;
        }
        /* pass 2: largest pivot among rows whose exact ratio is within the bound */
        int r = -1; double abest = -1;
        for (int i = 0; i < m; ++i) {
            if (t->skip[i] & 2) continue;
            const double a = t->D[(size_t)i * ld + c] * dir;
            if (fabs(a) <= ptol) continue;
            const int j = t->basic[i];
            const double room = a > 0 ? t->xB[i] - t->lo[j] : t->hi[j] - t->xB[i];
            const double r0 = fmax(room, 0.0) / fabs(a);
            if (r0 <= tmax && fabs(a) > abest) { abest = fabs(a); r = i; }
        }
        if (r < 0) {
            const double range = t->hi[jc] - t->lo[jc];
            if (!isfinite(range)) return LP_UNBOUNDED;
            /* bound flip */
            const double nw = t->at_upper[c] ? t->lo[jc] : t->hi[jc];
            const double dl = nw - t->xN[c];
            for (int i = 0; i < m; ++i) if (!(t->skip[i] & 2)) t->xB[i] -= t->D[(size_t)i * ld + c] * dl;
            t->xN[c] = nw; t->at_upper[c] = !t->at_upper[c];
            t->pivots++;
            continue;
        }
        {
            const int jr = t->basic[r];
            const double a = t->D[(size_t)r * ld + c] * dir;
            pivot(t, r, c, a > 0 ? t->lo[jr] : t->hi[jr]);
        }
    }
}

static void structural_x(const dict_t *t, double *x)
{
    for (int c = 0; c < t->n; ++c) if (t->nonbasic[c] < t->n) x[t->nonbasic[c]] = t->xN[c];
    for (int r = 0; r < t->m; ++r) if (t->basic[r] < t->n) x[t->basic[r]] = t->xB[r];
}

static void matvecP(const dict_t *t, const double *x, double *y)
{
    const int n = t->n;
    for (int i = 0; i < n; ++i) { const double *pi = t->P + (size_t)i * n; double s = 0; for (int j = 0; j < n; ++j) s += pi[j] * x[j]; y[i] = s; }
}

/* min 1/2 w'Hw + c'w over the unit simplex (p <= SD_PMAX), primal active set started from w (feasible) with index
 * `enter` forced free.  H is p x p (ld SD_PMAX). */
static void master_qp(int p, const double *H, const double *c, double *w, int enter)
{
    unsigned char F[SD_PMAX];
    double K[(SD_PMAX + 1) * (SD_PMAX + 2)], sol[SD_PMAX + 1], wn[SD_PMAX];
    for (int i = 0; i < p; ++i) F[i] = w[i] > 0 || i == enter;
    for (int iter = 0; iter < 200; ++iter) {
        int idx[SD_PMAX], nf = 0;
        for (int i = 0; i < p; ++i) if (F[i]) idx[nf++] = i;
        /* KKT: [H_FF 1; 1' 0] [w; nu] = [-c_F; 1] */
        const int N1 = nf + 1;
        double tr = 0; for (int a = 0; a < nf; ++a) tr += H[idx[a] * SD_PMAX + idx[a]];
        const double ridge = 1e-14 * fmax(tr, 1e-300);
        for (int a = 0; a < nf; ++a) {
            for (int b = 0; b < nf; ++b) K[a * (N1 + 1) + b] = H[idx[a] * SD_PMAX + idx[b]] + (a == b ? ridge : 0.0);
            K[a * (N1 + 1) + nf] = 1.0; K[a * (N1 + 1) + N1] = -c[idx[a]];
        }
        for (int b = 0; b < nf; ++b) K[nf * (N1 + 1) + b] = 1.0;
        K[nf * (N1 + 1) + nf] = 0.0; K[nf * (N1 + 1) + N1] = 1.0;
        for (int k = 0; k < N1; ++k) {      /* Gaussian elimination with partial pivoting */
            int pv = k; double mx = fabs(K[k * (N1 + 1) + k]);
            for (int a = k + 1; a < N1; ++a) if (fabs(K[a * (N1 + 1) + k]) > mx) { mx = fabs(K[a * (N1 + 1) + k]); pv = a; }
            if (pv != k) for (int b = 0; b <= N1; ++b) { const double tmp = K[k * (N1 + 1) + b]; K[k * (N1 + 1) + b] = K[pv * (N1 + 1) + b]; K[pv * (N1 + 1) + b] = tmp; }
            const double piv = K[k * (N1 + 1) + k];
            if (fabs(piv) < 1e-300) continue;
            for (int a = k + 1; a < N1; ++a) {
                const double f = K[a * (N1 + 1) + k] / piv;
                if (f != 0.0) for (int b = k; b <= N1; ++b) K[a * (N1 + 1) + b] -= f * K[k * (N1 + 1) + b];
            }
        }
        for (int k = N1 - 1; k >= 0; --k) {
            double sacc = K[k * (N1 + 1) + N1];
            for (int b = k + 1; b < N1; ++b) sacc -= K[k * (N1 + 1) + b] * sol[b];
            const double piv = K[k * (N1 + 1) + k];
            sol[k] = fabs(piv) < 1e-300 ? 0.0 : sacc / piv;
        }
        for (int i = 0; i < p; ++i) wn[i] = 0.0;
        for (int a = 0; a < nf; ++a) wn[idx[a]] = sol[a];
        int blocking = -1; double alpha = 1.0;
        for (int a = 0; a < nf; ++a) {
            const int i = idx[a];
            if (wn[i] < -1e-13 && w[i] - wn[i] > 0) { const double al = w[i] / (w[i] - wn[i]); if (al < alpha) { alpha = al; blocking = i; } }
        }
        if (blocking >= 0) {
            for (int i = 0; i < p; ++i) w[i] += alpha * (wn[i] - w[i]);
            w[blocking] = 0.0; F[blocking] = 0;
            continue;
        }
        for (int i = 0; i < p; ++i) w[i] = wn[i] > 0 ? wn[i] : 0.0;
        /* multipliers of the inactive vertices: g_i + nu >= 0 with nu = sol[nf] */
        int add = -1; double worst = -1e-12;
        for (int i = 0; i < p; ++i) {
            if (F[i]) continue;
            double g = c[i]; for (int j = 0; j < p; ++j) g += H[i * SD_PMAX + j] * w[j];
            if (g + sol[nf] < worst) { worst = g + sol[nf]; add = i; }
        }
        if (add < 0) break;
        F[add] = 1;
    }
    double sw = 0; for (int i = 0; i < p; ++i) sw += w[i];
    if (sw > 0) for (int i = 0; i < p; ++i) w[i] /= sw;
}

/* Convex-QP relaxation of the current node by simplicial decomposition (von Hohenbalken 1977; fully corrective
 * Frank-Wolfe).  On entry the dictionary is LP(q)-optimal for the node; on exit it is LP(q)-optimal again.
 * Returns 0 and (lower bound *lb, relaxation value *fv, point v in t->vcur) -- or 1 if *lb already exceeds `cutoff`,
 * or -1 on an iteration limit. */
static int sd_relax(dict_t *t, double cutoff, double *lb_out, double *fv_out)
{
    const int n = t->n;
    double *v = t->vcur, *Pv = t->Pv, *g = t->gcost;
    double *Y = t->Y, *PY = t->PY, *H = t->Hm, *cm = t->cm, *w = t->wm;
    int p = 1, rc = 0;
    structural_x(t, Y);
    matvecP(t, Y, PY);
    { double s = 0, l = 0; for (int j = 0; j < n; ++j) { s += Y[j] * PY[j]; l += t->q[j] * Y[j]; } H[0] = s; cm[0] = l; }
    w[0] = 1.0;
    memcpy(v, Y, sizeof(double) * n); memcpy(Pv, PY, sizeof(double) * n);
    double LB = -INFINITY, fv = 0;
    for (int it = 0; it < 80; ++it) {
        fv = 0; double gv = 0;
        for (int j = 0; j < n; ++j) { g[j] = Pv[j] + t->q[j]; fv += v[j] * (0.5 * Pv[j] + t->q[j]); gv += g[j] * v[j]; }
        reprice(t, g);
        const int lp = primal_simplex(t);
        if (lp != LP_OPTIMAL) { rc = -2; break; }
        t->qp_iters++;
        double *ynew = Y + (size_t)p * n;
        if (p >= SD_PMAX) {   /* drop the lightest vertex */
            int k = 0; for (int i = 1; i < p; ++i) if (w[i] < w[k]) k = i;
            for (int i = k; i + 1 < p; ++i) {
                memcpy(Y + (size_t)i * n, Y + (size_t)(i + 1) * n, sizeof(double) * n);
                memcpy(PY + (size_t)i * n, PY + (size_t)(i + 1) * n, sizeof(double) * n);
                w[i] = w[i + 1]; cm[i] = cm[i + 1];
            }
            for (int a = 0; a < p; ++a) for (int b = k; b + 1 < p; ++b) H[a * SD_PMAX + b] = H[a * SD_PMAX + b + 1];
            for (int a = k; a + 1 < p; ++a) for (int b = 0; b < p; ++b) H[a * SD_PMAX + b] = H[(a + 1) * SD_PMAX + b];
            p--; ynew = Y + (size_t)p * n;
        }
        structural_x(t, ynew);
        double gy = 0; for (int j = 0; j < n; ++j) gy += g[j] * ynew[j];
        const double lbt = fv + gy - gv;
        if (lbt > LB) LB = lbt;
        if (fv - LB <= 1e-10 * fmax(1.0, fabs(fv))) break;
        if (LB > cutoff) { rc = 1; break; }
        matvecP(t, ynew, PY + (size_t)p * n);
        for (int i = 0; i <= p; ++i) {
            double sacc = 0; const double *yi = Y + (size_t)i * n, *pyn = PY + (size_t)p * n;
            for (int j = 0; j < n; ++j) sacc += yi[j] * pyn[j];
            H[i * SD_PMAX + p] = H[p * SD_PMAX + i] = sacc;
        }
        { double l = 0; for (int j = 0; j < n; ++j) l += t->q[j] * ynew[j]; cm[p] = l; }
        w[p] = 0.0;
        p++;
        master_qp(p, H, cm, w, p - 1);
        /* drop zero-weight vertices, rebuild v and Pv */
        int k2 = 0;
        for (int i = 0; i < p; ++i) {
            if (w[i] <= 0.0) continue;
            if (k2 != i) {
                memcpy(Y + (size_t)k2 * n, Y + (size_t)i * n, sizeof(double) * n);
                memcpy(PY + (size_t)k2 * n, PY + (size_t)i * n, sizeof(double) * n);
                w[k2] = w[i]; cm[k2] = cm[i];
                for (int a = 0; a < p; ++a) H[a * SD_PMAX + k2] = H[a * SD_PMAX + i];
                for (int b = 0; b < p; ++b) H[k2 * SD_PMAX + b] = H[i * SD_PMAX + b];
            }
            k2++;
        }
        p = k2;
        for (int j = 0; j < n; ++j) { double a = 0, b = 0; for (int i = 0; i < p; ++i) { a += w[i] * Y[(size_t)i * n + j]; b += w[i] * PY[(size_t)i * n + j]; } v[j] = a; Pv[j] = b; }
    }
    if (rc != -2) { reprice(t, t->q); if (primal_simplex(t) != LP_OPTIMAL) rc = -2; }
    if (rc == -2) {
        /* the primal simplex gave up (cycling on a dictionary that has lost its accuracy): re-derive the dictionary from the original rows and return to the
         * LP(q) vertex of the node with the dual simplex; the caller goes on with the LP bound and the LP point of this node (-2), or ends (-1) */
        refactor(t);
        if (dual_simplex(t, INFINITY) != LP_OPTIMAL) rc = -1;
    }
    *lb_out = LB; *fv_out = fv;
    return rc;
}

/* row-activity bound propagation with integer rounding (Savelsbergh 1994); returns 0 if infeasible */
static int propagate_bounds(const double *G, const double *h, int m, int n, double *lb, double *ub,
                            const unsigned char *is_int)
{
    for (int pass = 0; pass < 4; ++pass) {
        int changed = 0;
        for (int i = 0; i < m; ++i) {
            const double *g = G + (size_t)i * n;
            double tot = 0; int ninf = 0, jinf = -1, nnz = 0;
            for (int j = 0; j < n; ++j) {
                if (g[j] == 0) continue;
                nnz++;
                const double lc = g[j] > 0 ? g[j] * lb[j] : g[j] * ub[j];
                if (isinf(lc)) { ninf++; jinf = j; } else tot += lc;
            }
            if (!nnz) { if (h[i] < -1e-9) return 0; continue; }
            if (!ninf && tot > h[i] + 1e-7 * fmax(1.0, fabs(h[i]))) return 0;
            if (ninf > 1) continue;
            for (int j = 0; j < n; ++j) {
                if (g[j] == 0) continue;
                double rest;
                if (!ninf) rest = tot - (g[j] > 0 ? g[j] * lb[j] : g[j] * ub[j]);
                else if (j == jinf) rest = tot;
                else continue;
                double b = (h[i] - rest) / g[j];
                if (g[j] > 0) {
                    if (is_int[j]) b = floor(b + 1e-7);
                    if (b < ub[j] - 1e-9 * fmax(1.0, fabs(b))) { ub[j] = b; changed = 1; }
                } else {
                    if (is_int[j]) b = ceil(b - 1e-7);
                    if (b > lb[j] + 1e-9 * fmax(1.0, fabs(b))) { lb[j] = b; changed = 1; }
                }
            }
        }
        for (int j = 0; j < n; ++j) { if (lb[j] > ub[j] + 1e-7) return 0; if (ub[j] < lb[j]) ub[j] = lb[j]; }
        if (!changed) break;
    }
    return 1;
}

typedef struct { double key; int r; } frac_t;
static int frac_cmp(const void *a, const void *b)
{
    const frac_t *x = (const frac_t *)a, *y = (const frac_t *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->r - y->r;
}

static int gmi_round(dict_t *t, int max_cuts)
{
    const int n = t->n, ld = t->ld, m = t->m;
    frac_t *fr = (frac_t *)calloc(m + 1, sizeof(frac_t));
    int nf = 0;
    for (int r = 0; r < m; ++r) {
        const int j = t->basic[r];
        if (j < n && t->is_int[j]) {
            const double f0 = t->xB[r] - floor(t->xB[r]);
            if (f0 > 1e-3 && f0 < 1 - 1e-3) { fr[nf].key = fabs(f0 - 0.5); fr[nf].r = r; nf++; }
        }
    }
    qsort(fr, nf, sizeof(frac_t), frac_cmp);
    double *x = dalloc(n), *g = dalloc(n), *ax = dalloc(n);
    for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) x[t->nonbasic[c]] = t->xN[c];
    for (int r = 0; r < m; ++r) if (t->basic[r] < n) x[t->basic[r]] = t->xB[r];
    int added = 0;
    const int m_start = m;
    for (int f = 0; f < nf; ++f) {
        if (m_start + added >= t->cut_cap || added >= max_cuts) break;
        const int r = fr[f].r;
        const double f0 = t->xB[r] - floor(t->xB[r]);
        const double *row = t->D + (size_t)r * ld;
        double gmin = INFINITY, gmax = 0;
        int unsafe = 0;
        for (int c = 0; c < n; ++c) {
            const int j = t->nonbasic[c];
            double a = (t->lo[j] == t->hi[j]) ? 0.0 : (t->at_upper[c] ? -row[c] : row[c]);
            /* round-off sized coefficients are dropped; that strengthens the cut by |a| t_j, which is only harmless when t_j is O(1): a free
             * variable resting on the artificial box (t_j up to 2e7) makes the cut unsafe -- it is not derived (gmin = 0 fails the range test) */
            if (fabs(a) < ORC_COEF_ZERO) { if (a != 0.0 && j < n && fabs(t->at_upper[c] ? t->hi[j] : t->lo[j]) >= 0.5 * ORC_BIG) unsafe = 1; a = 0.0; }
            double gc;
            if (j < n && t->is_int[j]) {
                double fj = a - floor(a);
                if (fj < ORC_COEF_ZERO || fj > 1 - ORC_COEF_ZERO) fj = 0.0;
                gc = fj <= f0 ? fj / f0 : (1 - fj) / (1 - f0);
            } else gc = a > 0 ? a / f0 : -a / (1 - f0);
            g[c] = gc;
            if (gc > 0) { if (gc < gmin) gmin = gc; if (gc > gmax) gmax = gc; }
        }
        if (unsafe || gmax <= 0 || gmax / gmin > 1e6) continue;
        /* sum_c g_c t_c >= 1  over structural variables:  ax.x <= bx */
        for (int j = 0; j < n; ++j) ax[j] = 0;
        double bx = -1.0;
        for (int c = 0; c < n; ++c) {
            if (g[c] <= 0) continue;
            const int j = t->nonbasic[c];
            const double sg = t->at_upper[c] ? -1.0 : 1.0;
            if (j < n) { const double bnd = t->at_upper[c] ? t->hi[j] : t->lo[j]; ax[j] -= g[c] * sg; bx -= g[c] * sg * bnd; }
            else { const int i = j - n; const double *gi = t->Gx + (size_t)i * n; for (int k = 0; k < n; ++k) ax[k] += g[c] * gi[k]; bx += g[c] * t->hx[i]; }
        }
        double nrm = 0;
        for (int j = 0; j < n; ++j) if (fabs(ax[j]) > nrm) nrm = fabs(ax[j]);
        if (nrm <= 0) continue;
        if (gmax > nrm) nrm = gmax;   /* keep the dictionary row (coefficients g) O(1) as well as the structural row */
        const int k = m_start + added;
        double *gk = t->Gx + (size_t)k * n, *dk = t->D + (size_t)k * ld;
        double s_now = bx / nrm, dx = 0;
        for (int j = 0; j < n; ++j) { gk[j] = ax[j] / nrm; s_now -= gk[j] * x[j]; }
        t->hx[k] = bx / nrm;
        for (int c = 0; c < n; ++c) { const double sg = t->at_upper[c] ? -1.0 : 1.0; dk[c] = -(g[c] / nrm) * sg; dx += dk[c] * t->xN[c]; }
        dk[n] = s_now + dx;
        t->basic[k] = n + k; t->where[n + k] = -1 - k;
        added++;
    }
    t->m = m_start + added;
    if (added) refresh(t);
    free(fr); free(x); free(g); free(ax);
    return added;
}

/* Complemented mixed-integer rounding (Marchand & Wolsey 2001) on the ORIGINAL rows  g.x <= h  (scaled space).  The tank
 * rows of an MLD-MPC problem are knapsack rows over the horizon's binaries with one continuous slack (lot-sizing
 * structure): rounding them gives what a single Gomory cut of the current vertex cannot.  Binaries above 1/2 are
 * complemented, continuous variables are bound-substituted, the row is divided by delta in {|g_j| of fractional binaries}
 * x {1, 1/2, 1/4, 1/8} and the MIR inequality with the best efficacy is kept.  Up to max_cuts rows are appended. */
typedef struct { double eff; int row; double delta; } mir_cand_t;
static int mir_cmp(const void *a, const void *b) { const double x = ((const mir_cand_t *)a)->eff, y = ((const mir_cand_t *)b)->eff; return x > y ? -1 : (x < y ? 1 : ((const mir_cand_t *)a)->row - ((const mir_cand_t *)b)->row); }

/* builds the cut for (row, delta) into ax/bx (structural, "ax.x <= bx"); returns the violation at x (<= 0: no cut) */
static double mir_build(const dict_t *t, int i, double delta, const double *x, double *ax, double *bx_out, double *nrm2_out)
{
    const int n = t->n;
    const double *g = t->Gx + (size_t)i * n;
    double rhs = t->hx[i], sval = 0.0;
    for (int j = 0; j < n; ++j) ax[j] = 0.0;
    /* pass 1: right-hand side after complementing / bound substitution */
    for (int j = 0; j < n; ++j) {
        const double gj = g[j];
        if (gj == 0.0) continue;
        const double lo = t->is_int[j] ? t->lo[j] : t->clo[j], hi = t->is_int[j] ? t->hi[j] : t->chi[j];
        if (t->is_int[j]) { if (lo == hi) rhs -= gj * lo; else if (x[j] > 0.5) rhs -= gj; }
        else {
            const int lof = lo > -0.5 * ORC_BIG, hif = hi < 0.5 * ORC_BIG;
            if (gj > 0) { if (lof) rhs -= gj * lo; else if (hif) { rhs -= gj * hi; sval += gj * (hi - x[j]); } else return -1.0; }
            else { if (lof) { rhs -= gj * lo; sval += -gj * (x[j] - lo); } else if (hif) rhs -= gj * hi; else return -1.0; }
        }
    }
    const double b = rhs / delta, fb = floor(b), f = b - fb;
    /* the fractional part must lie in [ORC_MIR_FMIN, 1 - ORC_MIR_FMIN] (round 3: 0.005, was 0.05 -- the strongest roundings of the tank rows are the
     * "just barely needs one more heating step" ones, f ~ 0.003 .. 0.03; csrc/problem.inc S_MIR_FMIN) */
    { static double fmin_ = -1.0; if (fmin_ < 0) fmin_ = getenv("ORC_MIR_FMIN") ? atof(getenv("ORC_MIR_FMIN")) : ORC_MIR_FMIN;
      if (f < fmin_ || f > 1.0 - fmin_) return -1.0; }
    double lhs = 0.0, nrm2 = 0.0, bx = fb;
    const double kc = 1.0 / (delta * (1.0 - f));
    for (int j = 0; j < n; ++j) {
        const double gj = g[j];
        if (gj == 0.0) continue;
        const double lo = t->is_int[j] ? t->lo[j] : t->clo[j], hi = t->is_int[j] ? t->hi[j] : t->chi[j];
        if (t->is_int[j]) {
            if (lo == hi) continue;
            const int comp = x[j] > 0.5;
            const double a = (comp ? -gj : gj) / delta, fl = floor(a + 1e-12), fj = a - fl;
            const double cf = fl + (fj > f ? (fj - f) / (1.0 - f) : 0.0);
            lhs += cf * (comp ? 1.0 - x[j] : x[j]);
            if (comp) { ax[j] -= cf; bx -= cf; } else ax[j] += cf;
        } else {
            const int lof = lo > -0.5 * ORC_BIG, hif = hi < 0.5 * ORC_BIG;
            if (gj > 0) { if (!lof && hif) { ax[j] += kc * gj; bx += kc * gj * hi; } }          /* x = hi - x', coefficient -g on x' */
            else { if (lof) { ax[j] -= kc * (-gj); bx -= kc * (-gj) * lo; } }                   /* x = lo + x', coefficient g on x' */
        }
    }
    lhs -= sval * kc;
    for (int j = 0; j < n; ++j) nrm2 += ax[j] * ax[j];
    *bx_out = bx; *nrm2_out = nrm2;
    return lhs - fb;
}

static int mir_round(dict_t *t, int max_cuts)
{
    const int n = t->n, ld = t->ld, m = t->m, m0 = t->m0;
    double *x = dalloc(n), *ax = dalloc(n);
    for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) x[t->nonbasic[c]] = t->xN[c];
    for (int r = 0; r < m; ++r) if (t->basic[r] < n) x[t->basic[r]] = t->xB[r];
    mir_cand_t *cand = (mir_cand_t *)calloc(m0 + 1, sizeof(mir_cand_t));
    int nc = 0;
    static const double divs[4] = {1.0, 2.0, 4.0, 8.0};
    for (int i = 0; i < m0; ++i) {
        if (t->skip[i] & 2) continue;      /* a row that cannot bind under the bounds has no rounding that cuts a point inside them (csrc/problem.inc: same screen) */
        const double *g = t->Gx + (size_t)i * n;
        double best = 0.0, bdelta = 0.0;
        for (int j = 0; j < n; ++j) {
            if (g[j] == 0.0 || !t->is_int[j] || t->lo[j] == t->hi[j]) continue;
            if (x[j] < 1e-6 || x[j] > 1 - 1e-6) continue;
            for (int q = 0; q < 4; ++q) {
                const double delta = fabs(g[j]) / divs[q];
                if (delta < 1e-9) continue;
                double bx, nrm2;
                const double viol = mir_build(t, i, delta, x, ax, &bx, &nrm2);
                if (viol <= 1e-6 || nrm2 <= 0) continue;
                const double eff = viol / sqrt(nrm2);
                if (eff > best) { best = eff; bdelta = delta; }
            }
        }
        if (best > 1e-4) { cand[nc].eff = best; cand[nc].row = i; cand[nc].delta = bdelta; nc++; }
    }
    qsort(cand, nc, sizeof(mir_cand_t), mir_cmp);
    int added = 0;
    const int m_start = m;
    for (int q = 0; q < nc; ++q) {
        if (m_start + added >= t->cut_cap || added >= max_cuts) break;
        double bx, nrm2;
        if (mir_build(t, cand[q].row, cand[q].delta, x, ax, &bx, &nrm2) <= 1e-6) continue;
        double nrm = 0;
        for (int j = 0; j < n; ++j) if (fabs(ax[j]) > nrm) nrm = fabs(ax[j]);
        if (nrm <= 0) continue;
        const int k = m_start + added;
        double *gk = t->Gx + (size_t)k * n, *dk = t->D + (size_t)k * ld;
        for (int j = 0; j < n; ++j) gk[j] = ax[j] / nrm;
        t->hx[k] = bx / nrm;
        /* dictionary row of the new slack: s = hx - gk.x with the basic structurals substituted */
        for (int c = 0; c <= n; ++c) dk[c] = 0.0;
        dk[n] = t->hx[k];
        for (int c = 0; c < n; ++c) { const int j = t->nonbasic[c]; if (j < n) dk[c] += gk[j]; }
        for (int r = 0; r < m; ++r) {
            const int j = t->basic[r];
            if (j >= n || gk[j] == 0.0) continue;
            const double *dr = t->D + (size_t)r * ld;
            for (int c = 0; c < n; ++c) dk[c] -= gk[j] * dr[c];
            dk[n] -= gk[j] * dr[n];
        }
        t->basic[k] = n + k; t->where[n + k] = -1 - k;
        added++;
    }
    t->m = m_start + added;
    if (added) refresh(t);
    free(x); free(ax); free(cand);
    return added;
}

/* Driebeek (1966) / Tomlin penalties of a fractional basic binary in row r: lower bounds on the rise of the LP value when the
 * variable is branched down (x <= 0) or up (x >= 1), from one dual ratio test on its dictionary row each -- the first
 * dual simplex pivot of the child.  Every non-zero eligible entry takes part (no pivot tolerance: a smaller ratio only
 * weakens the bound), so obj + penalty never exceeds the child's LP value. */
static void penalties(const dict_t *t, int r, double *pd_out, double *pu_out)
{
    const int n = t->n;
    const double *row = t->D + (size_t)r * t->ld, *d = t->D + (size_t)t->mcap * t->ld;
    const double f = t->xB[r] - floor(t->xB[r]);
    double rd = INFINITY, ru = INFINITY;
    for (int c = 0; c < n; ++c) {
        const int j = t->nonbasic[c];
        if (t->lo[j] == t->hi[j]) continue;
        const double a = row[c];
        if (fabs(a) <= 1e-9) continue;
        const double da = fmax(t->at_upper[c] ? -d[c] : d[c], 0.0), ratio = da / fabs(a);
        if (t->at_upper[c] ? a < 0 : a > 0) { if (ratio < rd) rd = ratio; } else { if (ratio < ru) ru = ratio; }
    }
    *pd_out = isfinite(rd) ? rd * f : INFINITY;
    *pu_out = isfinite(ru) ? ru * (1.0 - f) : INFINITY;
}

/* Evaluation of a complete binary assignment (a leaf of the search, or a MIP start): every free binary is fixed at the rounded entry of
 * vals[], the remaining LP (QP) is solved on the same dictionary, the point is verified against the ORIGINAL rows and kept as the incumbent
 * when it is feasible and better; the bounds are restored.  Returns 1 when the rounded point is feasible. */
typedef struct {
    int n, m, nb; const int *bins;
    const double *Pq, *q, *G, *h, *lb, *ub;
    int *sv_j; double *sv_lo, *sv_hi, *xo;
    double *best; int *have; double *x_out; int *unbounded;
} leaf_ctx;

static int leaf_eval(dict_t *t, leaf_ctx *L, const double *vals)
{
    const int n = L->n, m = L->m, nb = L->nb; const int *bins = L->bins;
    double *xo = L->xo;
    int ns = 0;
    for (int k = 0; k < nb; ++k) {
        const int j = bins[k];
        if (t->lo[j] != t->hi[j]) { L->sv_j[ns] = j; L->sv_lo[ns] = t->lo[j]; L->sv_hi[ns] = t->hi[j]; ns++; const double v = fmin(fmax(rint(vals[j]), t->lo[j]), t->hi[j]); set_bounds(t, j, v, v); }
    }
    int leaf_ok = dual_simplex(t, INFINITY) == LP_OPTIMAL;
    if (leaf_ok && t->P) { double lbq, fvq; leaf_ok = sd_relax(t, INFINITY, &lbq, &fvq) == 0; }
    if (leaf_ok) {
        if (t->P) for (int j = 0; j < n; ++j) xo[j] = t->vcur[j] * t->cs[j];
        else {
            for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) xo[t->nonbasic[c]] = t->xN[c] * t->cs[t->nonbasic[c]];
            for (int r = 0; r < t->m; ++r) if (t->basic[r] < n) xo[t->basic[r]] = t->xB[r] * t->cs[t->basic[r]];
        }
        for (int k = 0; k < nb; ++k) xo[bins[k]] = rint(xo[bins[k]]);
        double ob = 0; for (int j = 0; j < n; ++j) ob += L->q[j] * xo[j];
        if (L->Pq) for (int i = 0; i < n; ++i) { const double *pi = L->Pq + (size_t)i * n; double sq = 0; for (int j = 0; j < n; ++j) sq += pi[j] * xo[j]; ob += 0.5 * xo[i] * sq; }
        int feas = 1;
        for (int i = 0; i < m && feas; ++i) {
            double sa = -L->h[i]; const double *gi = L->G + (size_t)i * n;
            for (int j = 0; j < n; ++j) sa += gi[j] * xo[j];
            if (sa * t->rs[i] > 1e-6) feas = 0;
        }
        if (feas && ob < *L->best) { *L->best = ob; *L->have = 1; memcpy(L->x_out, xo, sizeof(double) * n); }
        if (feas && t->P)   /* (QP relaxations: a free variable resting on the artificial box counts as no finite optimum) */
            for (int j = 0; j < n; ++j)
                if ((L->lb[j] == -INFINITY && xo[j] / t->cs[j] <= -0.999 * ORC_BIG) || (L->ub[j] == INFINITY && xo[j] / t->cs[j] >= 0.999 * ORC_BIG)) *L->unbounded = 1;
        if (feas && !t->P) {
            /* A free variable resting NON-BASIC on the artificial box (+-ORC_BIG) with a non-zero reduced cost: the value
             * still falls along its ray -- no finite optimum.  With a zero reduced cost the box is only where a
             * variable the objective does not depend on happens to sit: the optimum is finite (found by the fuzz
             * test: a zero-cost auxiliary z whose column only relaxes the rows; round 1 called that unbounded). */
            const double *dd = t->D + (size_t)t->mcap * t->ld;
            for (int c = 0; c < n; ++c) {
                const int j = t->nonbasic[c];
                if (j >= n || fabs(dd[c]) <= 1e-9) continue;
                if ((L->lb[j] == -INFINITY && t->xN[c] <= -0.999 * ORC_BIG) || (L->ub[j] == INFINITY && t->xN[c] >= 0.999 * ORC_BIG)) *L->unbounded = 1;
            }
        }
        leaf_ok = feas;
    }
    for (int k = 0; k < ns; ++k) set_bounds(t, L->sv_j[k], L->sv_lo[k], L->sv_hi[k]);
    return leaf_ok;
}

static double gtol(const orc_opts *o, double v) { return fmax(o->gap_abs, o->gap_rel * fabs(v)); }

static int solve_core(int n, int m, const double *Pq, const double *q, const double *G, const double *h, const double *lb_in,
                      const double *ub_in, const unsigned char *is_bin, const orc_opts *o, const double *x_start, double *x_out,
                      double *obj_out, orc_stats *st);

int orc_solve_miqp(int n, int m, const double *Pq, const double *q, const double *G, const double *h, const double *lb_in,
                   const double *ub_in, const unsigned char *is_bin, const orc_opts *o, double *x_out,
                   double *obj_out, orc_stats *st)
{
    return solve_core(n, m, Pq, q, G, h, lb_in, ub_in, is_bin, o, NULL, x_out, obj_out, st);
}

int orc_solve_milp(int n, int m, const double *q, const double *G, const double *h, const double *lb_in,
                   const double *ub_in, const unsigned char *is_bin, const orc_opts *o, double *x_out,
                   double *obj_out, orc_stats *st)
{
    return solve_core(n, m, NULL, q, G, h, lb_in, ub_in, is_bin, o, NULL, x_out, obj_out, st);
}

/* MIP start (the reference forwards warm_start=True to its backend, controllers/controller_base.py:493,509-512: the previous values of the
 * variables are the solver's start).  x_start (n values, NULL = none): its binaries, rounded, are fixed after the root LP, the remaining LP is
 * solved and, when the point satisfies the original rows, it is the initial incumbent: its value is the cutoff of everything that follows and the
 * cut loop stops as soon as the bound is within the gap of it. */
int orc_solve_miqp_start(int n, int m, const double *Pq, const double *q, const double *G, const double *h, const double *lb_in,
                         const double *ub_in, const unsigned char *is_bin, const orc_opts *o, const double *x_start, double *x_out,
                         double *obj_out, orc_stats *st)
{
    return solve_core(n, m, Pq, q, G, h, lb_in, ub_in, is_bin, o, x_start, x_out, obj_out, st);
}

/* min 1/2 x'Px + q'x  s.t. Gx <= h, lb <= x <= ub, x_i in {0,1} (i binary);  P symmetric PSD or NULL */
static int solve_core(int n, int m, const double *Pq, const double *q, const double *G, const double *h, const double *lb_in,
                      const double *ub_in, const unsigned char *is_bin, const orc_opts *o, const double *x_start, double *x_out,
                      double *obj_out, orc_stats *st)
{
    memset(st, 0, sizeof(*st));
    st->root_lp = st->root_bound = NAN; st->lower_bound = -INFINITY;
    *obj_out = INFINITY;
    double *lb = dalloc(n), *ub = dalloc(n);
    memcpy(lb, lb_in, sizeof(double) * n); memcpy(ub, ub_in, sizeof(double) * n);
    int status = ORC_INFEASIBLE;
    if ((o->presolve & 1) && !propagate_bounds(G, h, m, n, lb, ub, is_bin)) { st->status = status; free(lb); free(ub); return status; }
    dict_t T; dict_t *t = &T; memset(t, 0, sizeof(T));
    t->n = n; t->m0 = m; t->m = m; t->mcap = m + o->max_cuts + ((o->max_cuts > 0 && o->cut_rounds > 0) ? ORC_RESTART_ROWS : 0); t->cut_cap = m + o->max_cuts; t->ld = n + 1; t->ntot = n + t->mcap;
    t->max_pivots = o->max_pivots > 0 ? o->max_pivots : 2000000000L;
    t->D = dalloc((size_t)(t->mcap + 1) * t->ld);
    t->Gx = dalloc((size_t)t->mcap * n); t->hx = dalloc(t->mcap);
    t->q = dalloc(n); t->rs = dalloc(m); t->cs = dalloc(n);
    t->lo = dalloc(t->ntot); t->hi = dalloc(t->ntot); t->clo = dalloc(n); t->chi = dalloc(n);
    t->xB = dalloc(t->mcap); t->xN = dalloc(n);
    t->basic = (int *)calloc(t->mcap + 1, sizeof(int)); t->nonbasic = (int *)calloc(n + 1, sizeof(int));
    t->where = (int *)calloc(t->ntot + 1, sizeof(int));
    t->at_upper = (unsigned char *)calloc(n + 1, 1); t->is_int = (unsigned char *)calloc(n + 1, 1);
    t->skip = (unsigned char *)calloc(t->mcap + 1, 1);
    t->tmp_col = dalloc(t->mcap + 1); t->tmp_row = dalloc(n + 1); t->dw = dalloc(t->mcap + 1);
    memcpy(t->is_int, is_bin, n);
    equilibrate(G, m, n, t->is_int, t->rs, t->cs);
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) t->Gx[(size_t)i * n + j] = G[(size_t)i * n + j] * t->rs[i] * t->cs[j];
        t->hx[i] = h[i] * t->rs[i];
    }
    for (int j = 0; j < n; ++j) { t->q[j] = q[j] * t->cs[j]; t->lo[j] = lb[j] / t->cs[j]; t->hi[j] = ub[j] / t->cs[j]; t->clo[j] = t->lo[j]; t->chi[j] = t->hi[j]; }
    double *Ps = NULL;
    if (Pq) {
        Ps = dalloc((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ps[(size_t)i * n + j] = Pq[(size_t)i * n + j] * t->cs[i] * t->cs[j];
        t->P = Ps;
        t->Y = dalloc((size_t)(SD_PMAX + 1) * n); t->PY = dalloc((size_t)(SD_PMAX + 1) * n);
        t->Hm = dalloc(SD_PMAX * SD_PMAX); t->cm = dalloc(SD_PMAX); t->wm = dalloc(SD_PMAX);
        t->gcost = dalloc(n); t->vcur = dalloc(n); t->Pv = dalloc(n);
    }
    for (int i = 0; i < t->mcap; ++i) { t->lo[n + i] = 0; t->hi[n + i] = INFINITY; }
    if (g_el_on < 0) g_el_on = getenv("ORC_ELASTIC_STATS") != NULL;
    if (g_el_on && !Pq) {      /* study: singleton penalty columns and their rows */
        t->partner = (int *)calloc(n + 1, sizeof(int));
        for (int j = 0; j < n; ++j) {
            int cnt = 0, row = -1; double v = 0.0;
            for (int i = 0; i < m; ++i) if (G[(size_t)i * n + j] != 0.0) { cnt++; row = i; v = G[(size_t)i * n + j]; }
            t->partner[j] = (cnt == 1 && v < 0.0 && !is_bin[j] && lb[j] == 0.0 && ub[j] == INFINITY && q[j] >= 0.0) ? row : -1;
        }
    }
    const int pre_infeasible = (o->presolve & 4) && !presolve_instance(t, m);
    mark_dead(t, m, Pq == NULL);
    reset_dictionary(t);
    for (int c = 0; c < n; ++c) place(t, c);
    refresh(t);

    int *bins = (int *)calloc(n + 1, sizeof(int)); int nb = 0;
    for (int j = 0; j < n; ++j) if (is_bin[j]) bins[nb++] = j;
    double *root_lo = dalloc(n), *root_hi = dalloc(n), *xs = dalloc(n), *xo = dalloc(n);
    double best = INFINITY; int have = 0;
    int *stk_j = (int *)calloc(nb + 2, sizeof(int));
    (void)0;
    double *stk_first = dalloc(nb + 2);
    unsigned char *stk_second = (unsigned char *)calloc(nb + 2, 1);
    int *sv_j = (int *)calloc(nb + 2, sizeof(int)); double *sv_lo = dalloc(nb + 2), *sv_hi = dalloc(nb + 2);
    int lp = LP_OPTIMAL, root_ok = 0, unbounded = 0, started = 0, nodes_pre = 0;
    leaf_ctx L = { n, m, nb, bins, Pq, q, G, h, lb, ub, sv_j, sv_lo, sv_hi, xo, &best, &have, x_out, &unbounded };
    st->nodes = 1;
    if (pre_infeasible) { status = ORC_INFEASIBLE; goto done; }
    /* root LP + cut rounds; if the LP breaks down while cutting the root is rebuilt and solved without cuts */
    for (int attempt = 0; attempt < 2 && !root_ok; ++attempt) {
        const int use_cuts = attempt == 0 && o->cut_rounds > 0;
        if (attempt) {
            st->rebuilds += 1.0;
            t->m = m;
            reset_dictionary(t);
            for (int c = 0; c < n; ++c) place(t, c);
            refresh(t);
        }
        if (x_start && !started && !t->P && ORC_EAGER_START) {
            /* the MIP start FIRST (round 4, csrc/problem.inc: same rule): its leaf LP from the slack basis, then the root relaxation from the leaf's
             * basis; the incumbent is known before the first cut, and the cut loop below stops as soon as the bound is within the gap of it */
            started = 1;
            leaf_eval(t, &L, x_start);                  /* (a leaf, not a node of the tree: not counted) */
        }
        { const int bm = getenv("ORC_BFRT") ? atoi(getenv("ORC_BFRT")) : ORC_BFRT_DEFAULT; t->bfrt_on = bm >= 1;
        lp = dual_simplex(t, INFINITY);
        t->bfrt_on = bm >= 2; }
        if (lp != LP_OPTIMAL) { status = lp == LP_INFEASIBLE ? ORC_INFEASIBLE : ORC_NUMERICAL; goto done; }
        if (!attempt) st->root_lp = objective(t);
        st->phase_work[0] = t->work;
        root_ok = 1;
        st->cuts = 0;
        if (use_cuts) {
            int stalled = 0;
            const long saved_cap = t->max_pivots;
            for (int rnd = 0; rnd < o->cut_rounds; ++rnd) {
                const double before = objective(t);
                if (have && best - gtol(o, best) <= before) break;      /* the incumbent is within the gap of the bound: no more cuts needed */
                int k = gmi_round(t, o->cuts_per_round);
                const int kg = k;
                if (o->mir_per_round > 0) k += mir_round(t, o->mir_per_round);
                if (getenv("ORC_DEBUG_CUTS")) fprintf(stderr, "cut round %d: obj %.6f gmi %d mir %d rows %d\n", rnd, before, kg, k - kg, t->m);
                if (!k) break;
                st->cuts += k;
                const long cap = t->pivots + 4L * m + 200;
                t->max_pivots = cap < saved_cap ? cap : saved_cap;
                lp = dual_simplex(t, INFINITY);
                t->max_pivots = saved_cap;
                if (lp != LP_OPTIMAL) { root_ok = 0; break; }
                if (have && best - gtol(o, best) <= objective(t)) break;      /* (eager start: nothing left to prove) */
                if (objective(t) - before < 1e-6 * fmax(1.0, fabs(before))) { if (++stalled >= (getenv("ORC_PATIENCE") ? atoi(getenv("ORC_PATIENCE")) : ORC_CUT_PATIENCE)) break; } else stalled = 0;
            }
            /* cut rows whose slack still sits basic in its own row (the cut never had to leave) and is clearly positive when the cut loop ends
             * are not maintained below the root (csrc/problem.inc, same rule): dropping a cut is always valid, and the pivots get cheaper */
            if (root_ok && !t->P) for (int i = m; i < t->m; ++i) if (t->basic[i] == n + i && t->xB[i] > ORC_PURGE_SLACK) t->skip[i] |= 2;
        }
    }
    st->phase_work[1] = t->work - st->phase_work[0];
    t->bfrt_on = (getenv("ORC_BFRT") ? atoi(getenv("ORC_BFRT")) : ORC_BFRT_DEFAULT) >= 3;
    if (!root_ok) { status = ORC_NUMERICAL; goto done; }
    if (t->P) {   /* root bound of the QP relaxation */
        double lbq, fvq;
        const int rcq = sd_relax(t, INFINITY, &lbq, &fvq);
        if (rcq == -1) { status = ORC_NUMERICAL; goto done; }
        st->root_bound = rcq == -2 ? objective(t) : lbq;      /* (-2: the QP relaxation could not be finished -- the LP(q) value is a bound too) */
    } else
    st->root_bound = objective(t);
    if (getenv("ORC_DUMP_ROOT")) {   /* study only: the root point after cuts (unscaled), one value per line */
        FILE *fp = fopen(getenv("ORC_DUMP_ROOT"), "w");
        if (fp) {
            for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) xs[t->nonbasic[c]] = t->xN[c];
            for (int r = 0; r < t->m; ++r) if (t->basic[r] < n) xs[t->basic[r]] = t->xB[r];
            for (int j = 0; j < n; ++j) fprintf(fp, "%.17g\n", xs[j] * t->cs[j]);
            fclose(fp);
        }
    }
    memcpy(root_lo, t->lo, sizeof(double) * n); memcpy(root_hi, t->hi, sizeof(double) * n);
    {
        /* Depth-first branch-and-bound with iterative deepening on the LP bound (Korf 1985): pass k
         * explores every node whose bound is <= min(T_k, incumbent - tol).  T starts at the root bound
         * (finds an optimal vertex of the optimal face quickly when the root bound is tight), grows
         * geometrically while no incumbent exists and becomes +inf once one does.
         *
         * Primal side for the degenerate instances (phases): when the deepening passes have used a quarter of
         * the node budget without an incumbent, (DIVE) one look-ahead dive -- the fractional binary closest to
         * 1 goes up unless that raises the LP value, in which case the better child is taken -- produces an
         * incumbent; (RINS) binaries on which the incumbent and the root relaxation agree are fixed and the
         * remaining sub-problem is searched depth-first (Danna, Rothberg, Le Pape 2005), at most twice; (FINAL)
         * the plain search continues with the incumbent as cutoff for the rest of the node budget. */
        enum { PH_IDS = 0, PH_DIVE, PH_RINS, PH_FINAL };
        const double root_bound = st->root_bound;
        int nodes = nodes_pre, limit = 0, pass = 0, rescue = 0, node_budget = o->max_nodes;
        int phase = have ? PH_RINS : PH_IDS, rins_rounds = 0, nfix = 0, xroot_set = 0;      /* (an incumbent from an eager start: RINS around it first) */
        if (have) node_budget = o->max_nodes / 4 < o->max_nodes ? o->max_nodes / 4 : o->max_nodes;
        int ids_cap = o->max_nodes / 8 > 16 ? o->max_nodes / 8 : 16;
        if (getenv("ORC_IDS_CAP")) { const int c = atoi(getenv("ORC_IDS_CAP")); if (c < ids_cap) ids_cap = c; }
        const double dive_tol = 1e-2 * fmax(1.0, fabs(root_bound));
        double *xroot = dalloc(nb + 1), *fx_lo = dalloc(nb + 1), *fx_hi = dalloc(nb + 1);
        int *fx_j = (int *)calloc(nb + 1, sizeof(int));
        if (have) {      /* (the first pass is RINS: it needs the root relaxation's binaries now) */
            for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) xs[t->nonbasic[c]] = t->xN[c];
            for (int r = 0; r < t->m; ++r) if (t->basic[r] < n) xs[t->basic[r]] = t->xB[r];
            for (int k = 0; k < nb; ++k) xroot[k] = xs[bins[k]];
            xroot_set = 1;
        }
        double T = have ? INFINITY : root_bound + fmax(1e-7 * fmax(1.0, fabs(root_bound)), gtol(o, root_bound));
        const int pen_mode = getenv("ORC_PEN") ? atoi(getenv("ORC_PEN")) : ORC_PEN_DEFAULT;   /* 0: first fractional binary in index order (A/B) */
        /* pseudocosts (Benichou et al. 1971; reliability threshold of Achterberg, Koch, Martin 2005 with the Driebeek-Tomlin penalty as the unreliable
         * estimate): per binary and direction the average rise of the LP value per unit of change, learnt from every child LP of this search */
        const int psc_mode = getenv("ORC_PSC") ? atoi(getenv("ORC_PSC")) : ORC_PSC_DEFAULT;
        double *ps_sum = dalloc(2 * (size_t)n + 2); int *ps_cnt = (int *)calloc(2 * (size_t)n + 2, sizeof(int));
        double *stk_obj = dalloc(nb + 2), *stk_x = dalloc(nb + 2);
        int fresh_child = 0;          /* the node about to be evaluated is the child just created on top of the stack */
        double lbg = root_bound;      /* proven global lower bound: raised by every exhaustive pass */
        double restart_best = INFINITY;      /* incumbent value of the last root restart */
        status = ORC_NODE_LIMIT;
        for (;;) {
            const double work_at_pass = t->work; const int phase_at_pass = phase;
            int depth = 0;
            double t_next = INFINITY;   /* smallest bound among nodes pruned by T only */
            int finished = 0, dive_end = 0;
            const double best_at_start = best;
            pass++;
            if (phase == PH_FINAL && have && !x_start && !rescue && !t->P && o->cut_rounds > 0 && t->m < t->mcap && best < restart_best &&
                best - lbg <= ORC_RESTART_GAPS * gtol(o, best) && nodes >= o->max_nodes / 8 && !getenv("ORC_NO_RESTART")) {
                /* Root restart (round 4; csrc/problem.inc: same rule).  The final search is about to start from the root with an incumbent that is close to the proven
                 * bound but not within the gap -- under the flat mid-day tariff that is the whole unproven tail.  More cut rounds at the root, under the incumbent's
                 * cutoff and in the cut rows the root cut loop left free, move the bound the last percent for most of them (256 flat-tariff bench instances:
                 * 3 -> 1 unproven, -3 % row updates; instances that never get here are untouched). */
                restart_best = best;
                const double rcut = best - gtol(o, best) + 1e-12;
                t->cut_cap = t->mcap;
                int lpr = dual_simplex(t, rcut), stalled = 0;
                for (int rnd = 0; rnd < ORC_RESTART_ROUNDS && rnd < o->cut_rounds && lpr == LP_OPTIMAL && t->m < t->mcap; ++rnd) {
                    const double before = objective(t);
                    int k = gmi_round(t, o->cuts_per_round);
                    if (o->mir_per_round > 0) k += mir_round(t, o->mir_per_round);
                    if (!k) break;
                    st->cuts += k;
                    const long saved_cap = t->max_pivots, cap = t->pivots + 4L * m + 200;
                    t->max_pivots = cap < saved_cap ? cap : saved_cap;
                    lpr = dual_simplex(t, rcut);
                    t->max_pivots = saved_cap;
                    if (lpr != LP_OPTIMAL) break;
                    if (objective(t) - before < 1e-6 * fmax(1.0, fabs(before))) { if (++stalled >= 3) break; } else stalled = 0;
                }
                if (lpr == LP_CUTOFF) { lbg = fmax(lbg, best - gtol(o, best)); status = ORC_OPTIMAL; break; }      /* the root is closed by its bound under the cutoff */
                if (lpr == LP_OPTIMAL) {
                    for (int i = m; i < t->m; ++i) if (t->basic[i] == n + i && t->xB[i] > ORC_PURGE_SLACK) t->skip[i] |= 2;
                    if (objective(t) > lbg) lbg = objective(t);
                }
            }
            if (phase == PH_RINS) {   /* fix the binaries on which incumbent and root relaxation agree */
                nfix = 0;
                for (int k = 0; k < nb; ++k) {
                    const int j = bins[k];
                    if (root_lo[j] != root_hi[j] && fabs(xroot[k] - x_out[j]) <= ORC_INTTOL) {
                        fx_j[nfix] = j; fx_lo[nfix] = root_lo[j]; fx_hi[nfix] = root_hi[j]; nfix++;
                        root_lo[j] = root_hi[j] = x_out[j];
                        set_bounds(t, j, x_out[j], x_out[j]);
                    }
                }
            }
            for (;;) {
                /* ---- evaluate the current node */
                nodes++;
                int branch_j = -1; double branch_x = 0;
                int force_first = -1, second_done = 0;      /* penalty branching: preferred side, other side already excluded */
                /* RINS is a heuristic on a sub-problem: it keeps ANY improvement (round 4).  With the proof's cutoff -- "better by more than the gap" -- it threw away
                 * exactly the points that matter at the cfg5 size: the dive's leaf is 0.5-1 % above the best point, the root bound within 1 % of THAT one
                 * (32 cfg5 goldens: 25 -> 29 proven at NodeLimit 400, every incumbent within 1 % of HiGHS'; cfg4: +1-4 % row updates, same incumbents) */
                const double inc_cut = have ? (phase == PH_RINS ? best - 1e-6 * fmax(1.0, fabs(best)) : best - gtol(o, best)) : INFINITY;
                const double cut = fmin(T, inc_cut);
                double node_obj = INFINITY;
                lp = dual_simplex(t, cut + 1e-12);
                if (psc_mode && fresh_child && depth > 0 && (lp == LP_OPTIMAL || lp == LP_CUTOFF || lp == LP_INFEASIBLE)) {
                    const int jb = stk_j[depth - 1];
                    const int up = t->lo[jb] > 0.5;
                    const double dist = up ? 1.0 - stk_x[depth - 1] : stk_x[depth - 1];
                    double child = lp == LP_INFEASIBLE ? INFINITY : objective(t);
                    if (lp != LP_OPTIMAL && child < cut) child = cut;          /* (a cut-off child rose at least to the cutoff) */
                    if (!(child < 1e300)) child = fmax(cut < 1e300 ? cut : stk_obj[depth - 1], stk_obj[depth - 1]) + fabs(stk_obj[depth - 1]) * 0.1 + 1e-3;
                    if (dist > 1e-9 && stk_obj[depth - 1] < 1e300) { ps_sum[2 * jb + up] += fmax(0.0, child - stk_obj[depth - 1]) / dist; ps_cnt[2 * jb + up]++; }
                }
                fresh_child = 0;
                if (getenv("ORC_DEBUG")) fprintf(stderr, "node %d ph %d depth %d lp=%d obj=%.12g cut=%.12g T=%.12g pivots=%ld\n", nodes, phase, depth, lp, objective(t), cut, T, t->pivots);
                if (lp == LP_ITERLIMIT) limit = 1;
                else if (lp == LP_OPTIMAL || lp == LP_CUTOFF) {
                    double obj = objective(t);       /* LP(q) value: a valid bound also when P is PSD */
                    int pruned = (lp == LP_CUTOFF || obj > cut), qp_lp = 0;
                    node_obj = obj;
                    if (!pruned && t->P) {
                        double lbq, fvq;
                        const int rc = sd_relax(t, cut, &lbq, &fvq);
                        if (rc == -1) { limit = 1; pruned = 1; obj = INFINITY; }
                        else if (rc == -2) qp_lp = 1;      /* the QP relaxation could not be finished: this node goes on with its LP(q) bound and LP point */
                        else { obj = lbq; pruned = (rc == 1 || obj > cut); }
                    }
                    if (pruned) { if (obj <= inc_cut && obj < t_next) t_next = obj; }
                    else {
                        if (t->P && !qp_lp) memcpy(xs, t->vcur, sizeof(double) * n);
                        else {
                        for (int c = 0; c < n; ++c) if (t->nonbasic[c] < n) xs[t->nonbasic[c]] = t->xN[c];
                        for (int r = 0; r < t->m; ++r) if (t->basic[r] < n) xs[t->basic[r]] = t->xB[r];
                        }
                        if (!xroot_set) { for (int k = 0; k < nb; ++k) xroot[k] = xs[bins[k]]; xroot_set = 1; }
                        if (depth == 0 && have && (phase == PH_IDS || phase == PH_FINAL) && !t->P) {
                            /* reduced-cost fixing at the true root (no temporary fixings active): a non-basic binary whose
                             * reduced cost exceeds the room below the cutoff cannot leave its bound in any solution that
                             * still matters; the fixing is permanent */
                            const double *dd = t->D + (size_t)t->mcap * t->ld;
                            const double room = inc_cut - obj;
                            int nfx = 0;
                            for (int c = 0; c < n; ++c) {
                                const int j = t->nonbasic[c];
                                if (j >= n || !is_bin[j] || t->lo[j] == t->hi[j]) continue;
                                const double rc = t->at_upper[c] ? -dd[c] : dd[c];
                                if (rc > room + 1e-9) { const double v = t->xN[c]; root_lo[j] = root_hi[j] = v; set_bounds(t, j, v, v); nfx++; }
                            }
                            if (getenv("ORC_DEBUG")) fprintf(stderr, "  rc-fixed %d (room %.6g)\n", nfx, room);
                        }
                        if (phase == PH_DIVE) {   /* fractional binary closest to 1 (first such index) */
                            double bv = -1.0;
                            for (int k = 0; k < nb; ++k) {
                                const int j = bins[k];
                                /* a FIXED binary is never a candidate (round 4): basic with lo == hi it can drift off its value by more than the integrality tolerance
                                 * while the primal simplex of a QP relaxation ignores its tiny column entries; picked again and again it made a dive of 700 levels
                                 * on a 200-binary instance and the stack arrays overflowed (found under AddressSanitizer; csrc/problem.inc: same rule) */
                                if (t->lo[j] != t->hi[j] && fabs(xs[j] - rint(xs[j])) > ORC_INTTOL && xs[j] > bv) { bv = xs[j]; branch_j = j; branch_x = xs[j]; }
                            }
                        } else if (pen_mode && !t->P) {
                            /* penalty branching: the fractional binary with the largest product of up / down penalties (first index on
                             * ties); a side whose penalty lifts the bound over the cutoff is excluded without a node of its own, and the
                             * node is pruned when some variable has both sides excluded */
                            const double eps = 1e-6 * fmax(1.0, fabs(obj));
                            double bscore = -1.0; int bforced = 0;
                            for (int k = 0; k < nb && !pruned; ++k) {
                                const int j = bins[k];
                                if (fabs(xs[j] - rint(xs[j])) <= ORC_INTTOL || t->lo[j] == t->hi[j]) continue;
                                const int w = t->where[j];
                                if (w >= 0) continue;       /* (a fractional binary is basic) */
                                double pd, pu;
                                penalties(t, -1 - w, &pd, &pu);
                                const int xd = obj + pd > cut, xu = obj + pu > cut;
                                if (xd && xu) {
                                    const double b = obj + fmin(pd, pu);
                                    pruned = 1; branch_j = -1;
                                    if (b <= inc_cut && b < t_next) t_next = b;
                                    break;
                                }
                                if (xd || xu) {
                                    const double b = obj + (xd ? pd : pu);
                                    if (b <= inc_cut && b < t_next) t_next = b;
                                    if (!bforced) { bforced = 1; branch_j = j; branch_x = xs[j]; force_first = xd ? 1 : 0; second_done = 1; }
                                    continue;
                                }
                                if (bforced) continue;
                                if (psc_mode) {     /* reliable pseudocosts replace the penalty (a lower bound from ONE dual ratio test, usually far too small) */
                                    const double f = xs[j] - floor(xs[j]);
                                    if (ps_cnt[2 * j] >= psc_mode) pd = fmax(pd, ps_sum[2 * j] / ps_cnt[2 * j] * f);
                                    if (ps_cnt[2 * j + 1] >= psc_mode) pu = fmax(pu, ps_sum[2 * j + 1] / ps_cnt[2 * j + 1] * (1.0 - f));
                                }
                                const double sc = fmax(fmin(pd, 1e30), eps) * fmax(fmin(pu, 1e30), eps);
                                if (sc > bscore) { bscore = sc; branch_j = j; branch_x = xs[j]; force_first = (pd == pu) ? -1 : (pd < pu ? 0 : 1); second_done = 0; }
                            }
                        } else
                        for (int k = 0; k < nb; ++k) {
                            const int j = bins[k];
                            if (t->lo[j] != t->hi[j] && fabs(xs[j] - rint(xs[j])) > ORC_INTTOL) { branch_j = j; branch_x = xs[j]; break; }
                        }
                        if (branch_j < 0 && !pruned) {
                            /* leaf: fix every binary at its rounded value, re-solve, verify, restore */
                            int leaf_ok = leaf_eval(t, &L, xs);
                            if (!leaf_ok) {
                                /* The rounded point is not feasible although every binary is within the integrality
                                 * tolerance: with big-M rows a binary at 1e-6 can carry a whole unit of the row.  The
                                 * node is then NOT a leaf -- branch on the least integral free binary. */
                                double fbest = 1e-12;
                                for (int k = 0; k < nb; ++k) {
                                    const int j = bins[k];
                                    const double f = fabs(xs[j] - rint(xs[j]));
                                    if (t->lo[j] != t->hi[j] && f > fbest) { fbest = f; branch_j = j; branch_x = xs[j]; }
                                }
                            }
                        }
                    }
                }
                if (have && (rescue || unbounded || best <= lbg + gtol(o, best))) { finished = 1; }
                if (nodes >= ((phase == PH_IDS && !have && !rescue) ? ids_cap : node_budget)) limit = 1;
                if (phase == PH_DIVE && !limit && !finished) {
                    /* no backtracking: the dive ends at its first leaf, or when the node is infeasible */
                    if (branch_j < 0) dive_end = 1;
                    else {
                        const double tgt = 1.0;
                        set_bounds(t, branch_j, tgt, tgt);          /* (look-ahead LPs are not nodes of the tree: not counted against NodeLimit, round 4) */
                        int la = dual_simplex(t, INFINITY);
                        const double oa = la == LP_OPTIMAL ? objective(t) : INFINITY;
                        double take = tgt;
                        if (la == LP_ITERLIMIT) limit = 1;
                        else if (!(oa <= node_obj + dive_tol)) {
                            set_bounds(t, branch_j, 1.0 - tgt, 1.0 - tgt);
                            const int lb2 = dual_simplex(t, INFINITY);
                            const double ob2 = lb2 == LP_OPTIMAL ? objective(t) : INFINITY;
                            if (lb2 == LP_ITERLIMIT) limit = 1;
                            else if (ob2 <= oa) { take = 1.0 - tgt; if (ob2 == INFINITY) dive_end = 1; }
                            else set_bounds(t, branch_j, tgt, tgt);
                        }
                        (void)take;
                        if (depth >= nb) limit = 1;      /* (cannot happen: every level holds a different free binary) */
                        else { stk_j[depth] = branch_j; stk_first[depth] = take; stk_second[depth] = 1; depth++; }
                        if (!limit && !dive_end) continue; /* evaluate the chosen child */
                    }
                }
                if (branch_j >= 0 && depth >= nb) limit = 1;
                if (branch_j >= 0 && !limit && !finished && !dive_end) {
                    double first = force_first >= 0 ? (double)force_first : (branch_x >= 0.5 ? 1.0 : 0.0);
                    if (have && phase == PH_FINAL && !second_done) first = x_out[branch_j];     /* guided (Danna et al. 2005): towards the incumbent first */
                    stk_j[depth] = branch_j; stk_first[depth] = first; stk_second[depth] = (unsigned char)second_done; stk_obj[depth] = node_obj; stk_x[depth] = branch_x; depth++;
                    set_bounds(t, branch_j, first, first);
                    fresh_child = 1;
                    continue; /* evaluate the child */
                }
                /* ---- backtrack */
                if (limit || finished || dive_end) { while (depth > 0) { depth--; set_bounds(t, stk_j[depth], root_lo[stk_j[depth]], root_hi[stk_j[depth]]); } break; }
                while (depth > 0 && stk_second[depth - 1]) { depth--; set_bounds(t, stk_j[depth], root_lo[stk_j[depth]], root_hi[stk_j[depth]]); }
                if (depth == 0) break;
                stk_second[depth - 1] = 1;
                { const int j = stk_j[depth - 1]; const double v = 1.0 - stk_first[depth - 1]; set_bounds(t, j, v, v); fresh_child = 1; }
            }
            st->phase_work[2 + phase_at_pass] += t->work - work_at_pass;
            if (phase == PH_RINS) {   /* release the fixings */
                for (int k = 0; k < nfix; ++k) { const int j = fx_j[k]; root_lo[j] = fx_lo[k]; root_hi[j] = fx_hi[k]; set_bounds(t, j, fx_lo[k], fx_hi[k]); }
                rins_rounds++;
            }
            if (finished) { if (unbounded) { status = ORC_UNBOUNDED; best = -INFINITY; } else if (!rescue) status = ORC_OPTIMAL; break; }
            if (lp == LP_ITERLIMIT) break;
            if (phase == PH_IDS && limit && !have && !rescue && nodes < o->max_nodes && x_start && !started) {
                /* MIP start, evaluated lazily: only an instance whose deepening passes found no incumbent pays for it (one leaf); a feasible
                 * start then takes the place of the dive and the search continues with RINS around it */
                started = 1;
                leaf_eval(t, &L, x_start);
                if (have) { phase = PH_RINS; limit = 0; T = INFINITY; node_budget = nodes + o->max_nodes / 4 < o->max_nodes ? nodes + o->max_nodes / 4 : o->max_nodes; continue; }
            }
            if (phase == PH_IDS && limit && !have && !rescue && nodes < o->max_nodes) {
                /* the dive may finish even when it outlasts the node budget (it is what guarantees a feasible point) */
                phase = PH_DIVE; limit = 0; T = INFINITY;
                node_budget = o->max_nodes > nodes + 3 * nb + 10 ? o->max_nodes : nodes + 3 * nb + 10; continue;
            }
            if (phase == PH_DIVE) {
                limit = nodes >= o->max_nodes;
                if (have && !limit) { phase = PH_RINS; node_budget = nodes + o->max_nodes / 4 < o->max_nodes ? nodes + o->max_nodes / 4 : o->max_nodes; continue; }
                if (!have && !limit) { phase = PH_FINAL; node_budget = o->max_nodes; continue; }
            } else if (phase == PH_RINS) {
                limit = nodes >= o->max_nodes;
                if (!limit) {
                    if (best < best_at_start && rins_rounds < 2) { node_budget = nodes + o->max_nodes / 4 < o->max_nodes ? nodes + o->max_nodes / 4 : o->max_nodes; continue; }
                    phase = PH_FINAL; node_budget = o->max_nodes; continue;
                }
            }
            if (limit) {
                /* node limit without an incumbent: one un-thresholded dive so that a feasible point is returned */
                if (!have && !rescue) { rescue = 1; limit = 0; T = INFINITY; phase = PH_FINAL; node_budget = nodes + 3 * nb + 10; continue; }
                break;
            }
            /* the pass was exhaustive for its threshold */
            if (have && best - gtol(o, best) <= T) { status = ORC_OPTIMAL; break; }
            if (!isfinite(t_next)) {
                if (!have && !rescue) {   /* every node "infeasible": re-derive the dictionary from the original rows and dive once more */
                    refactor(t);
                    rescue = 1; T = INFINITY; phase = PH_FINAL; node_budget = nodes + 3 * nb + 10;
                    continue;
                }
                status = have ? (rescue ? ORC_NODE_LIMIT : ORC_OPTIMAL) : ORC_INFEASIBLE;
                break;
            }
            if (t_next > lbg) lbg = t_next;
            if (have) T = INFINITY;
            else T = fmax(t_next + 1e-9 * fmax(1.0, fabs(t_next)), T + ldexp(2.5e-4, 2 * pass) * fmax(1.0, fabs(T)));
        }
        free(xroot); free(fx_lo); free(fx_hi); free(fx_j);
        st->nodes = nodes;
        /* every way of ending OPTIMAL has closed all nodes below best - gtol; otherwise the bound is what exhaustive passes proved */
        st->lower_bound = status == ORC_OPTIMAL ? fmin(best, fmax(lbg, best - gtol(o, best))) : lbg;
    }
done:
    if (Ps) { free(Ps); free(t->Y); free(t->PY); free(t->Hm); free(t->cm); free(t->wm); free(t->gcost); free(t->vcur); free(t->Pv); }
    st->pivots = (int)t->pivots; st->refactors = t->refactors; st->status = status; st->work = t->work; st->bland = t->bland; st->flips = t->flips;
    *obj_out = have ? best : INFINITY;
    free(lb); free(ub); free(t->clo); free(t->chi); free(t->D); free(t->Gx); free(t->hx); free(t->q); free(t->rs); free(t->cs); free(t->lo);
    free(t->hi); free(t->xB); free(t->xN); free(t->basic); free(t->nonbasic); free(t->where); free(t->at_upper);
    free(t->is_int); free(t->skip); free(t->tmp_col); free(t->tmp_row); free(t->dw); free(t->partner); free(bins); free(root_lo); free(root_hi);
    free(xs); free(xo); free(stk_j); free(stk_first); free(stk_second); free(sv_j); free(sv_lo); free(sv_hi);
    return status;
}

/* exhaustive enumeration over the binaries + LP per leaf, for known-answer tests on tiny problems */
int orc_enumerate_milp(int n, int m, const double *q, const double *G, const double *h, const double *lb,
                       const double *ub, const unsigned char *is_bin, double *x_out, double *obj_out)
{
    int bins[24], nb = 0;
    for (int j = 0; j < n; ++j) if (is_bin[j]) { if (nb >= 24) return -1; bins[nb++] = j; }
    double *l = dalloc(n), *u = dalloc(n), *x = dalloc(n);
    unsigned char *nob = (unsigned char *)calloc(n + 1, 1);
    orc_opts o = {1e-9, 0.0, 1, 0, 0, 0, 0, 0};
    orc_stats st;
    double best = INFINITY;
    for (long mask = 0; mask < (1L << nb); ++mask) {
        memcpy(l, lb, sizeof(double) * n); memcpy(u, ub, sizeof(double) * n);
        for (int k = 0; k < nb; ++k) { const double v = (mask >> k) & 1; l[bins[k]] = u[bins[k]] = v; }
        double ob;
        const int s = orc_solve_milp(n, m, q, G, h, l, u, nob, &o, x, &ob, &st);
        if (s == ORC_OPTIMAL && ob < best) { best = ob; memcpy(x_out, x, sizeof(double) * n); }
    }
    *obj_out = best;
    free(l); free(u); free(x); free(nob);
    return isfinite(best) ? ORC_OPTIMAL : ORC_INFEASIBLE;
}

/* Batch of independent instances over all host cores (OpenMP, dynamic schedule: the solves differ by orders of magnitude) --
 * the CPU baseline leg of bench.py ("OpenMP over instances on all host cores").  Instance i uses q[i], G[i], h[i];
 * bounds and the binary mask are shared.  threads <= 0: all cores OpenMP gives this process. */
#ifdef _OPENMP
#include <omp.h>
#endif
int orc_solve_milp_batch(int n_inst, int n, int m, const double *const *q, const double *const *G, const double *const *h,
                         const double *lb, const double *ub, const unsigned char *is_bin, const orc_opts *o, int threads,
                         double *obj_out, int *status_out, int *nodes_out, int *pivots_out, double *lb_out)
{
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int i = 0; i < n_inst; ++i) {
        double *x = (double *)calloc((size_t)n + 1, sizeof(double));
        orc_stats st;
        double obj;
        status_out[i] = orc_solve_milp(n, m, q[i], G[i], h[i], lb, ub, is_bin, o, x, &obj, &st);
        obj_out[i] = obj; nodes_out[i] = st.nodes; pivots_out[i] = st.pivots; lb_out[i] = st.lower_bound;
        free(x);
    }
    return used;
}

/* ------------------------------------------------------------------------------------------------
 * (4) LP relaxation by a REVISED bounded dual simplex on the working basis (the formulation of the LDS-resident kernel
 *     k_lp_lds, csrc/lp_lds.inc): instead of a dense (m x n) dictionary only the inverse of the working basis
 *         W = G[T, B]   (T: tight rows = non-basic slacks, B: basic structurals, |T| = |B| = k, measured k <= 75 at cfg3)
 *     is kept (k x k), and every dictionary row / column the simplex needs is formed on the fly from W^-1 and rows / columns of G:
 *         row of basic structural B_p :  D[p, j] = rho . G[T, j],  D[p, s_t] = rho_t            with rho = W^-1[p, :]
 *         row of basic slack i        :  D[i, j] = G[i, j] - u . G[T, j],  D[i, s_t] = -u_t     with u = G[i, B] W^-1
 *         reduced costs               :  d_j = q_j - pi . G[T, j],  d_{s_t} = -pi_t             with pi = W^-T q_B
 *     The four basis changes (structural / slack entering x structural / slack leaving) replace a column of W, replace a row,
 *     border it or delete a row and a column: O(k^2) updates of W^-1.  Same tolerances, Harris ratio test and Bland fallback as
 *     dual_simplex() above; W^-1 and all primal / dual values are recomputed from G every ORC_REV_REFRESH pivots.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_REV_REFRESH 50

typedef struct {
    int n, m, k, kcap;
    const double *G;            /* scaled m x n */
    const double *h, *q;        /* scaled */
    double *lo, *hi;            /* structural bounds (scaled); free variables are boxed at +-ORC_BIG when they rest non-basic */
    double *x, *s;              /* structural values (n), slack values (m; 0 when tight) */
    int *posB, *posT;           /* structural -> position in B or -1; row -> position in T or -1 */
    unsigned char *up;          /* non-basic structural rests at its upper bound */
    int *Bs, *T;
    double *Winv;               /* kcap x kcap: rows = positions of B, columns = positions of T */
    double *d, *dS;             /* reduced costs: non-basic structurals (n), non-basic slacks by position (kcap) */
    double *alpha, *alphaS, *rho, *w, *col;
    unsigned char *skipX, *skipS;   /* violations within the skip tolerance without an eligible pivot: left alone for this solve */
    long pivots, refreshes;
} rev_t;

static int rev_refresh(rev_t *r)
{   /* W^-1 by Gauss-Jordan with partial pivoting, then x_B, s, pi, d from the original data */
    const int n = r->n, m = r->m, k = r->k, kc = r->kcap;
    double *Wm = dalloc((size_t)k * 2 * k);
    for (int t = 0; t < k; ++t) {
        for (int p = 0; p < k; ++p) Wm[(size_t)t * 2 * k + p] = r->G[(size_t)r->T[t] * n + r->Bs[p]];
        for (int p = 0; p < k; ++p) Wm[(size_t)t * 2 * k + k + p] = (p == t) ? 1.0 : 0.0;
    }
    for (int c = 0; c < k; ++c) {
        int pv = c; double mx = fabs(Wm[(size_t)c * 2 * k + c]);
        for (int a = c + 1; a < k; ++a) if (fabs(Wm[(size_t)a * 2 * k + c]) > mx) { mx = fabs(Wm[(size_t)a * 2 * k + c]); pv = a; }
        if (mx < 1e-13) { free(Wm); return 0; }
        if (pv != c) for (int b = 0; b < 2 * k; ++b) { const double t2 = Wm[(size_t)c * 2 * k + b]; Wm[(size_t)c * 2 * k + b] = Wm[(size_t)pv * 2 * k + b]; Wm[(size_t)pv * 2 * k + b] = t2; }
        const double inv = 1.0 / Wm[(size_t)c * 2 * k + c];
        for (int b = 0; b < 2 * k; ++b) Wm[(size_t)c * 2 * k + b] *= inv;
        for (int a = 0; a < k; ++a) {
            if (a == c) continue;
            const double f = Wm[(size_t)a * 2 * k + c];
            if (f != 0.0) for (int b = 0; b < 2 * k; ++b) Wm[(size_t)a * 2 * k + b] -= f * Wm[(size_t)c * 2 * k + b];
        }
    }
    /* [W | I] -> [I | W^-1]: W^-1[p][t] sits at row p, column k + t */
    for (int p = 0; p < k; ++p) for (int t = 0; t < k; ++t) r->Winv[(size_t)p * kc + t] = Wm[(size_t)p * 2 * k + k + t];
    free(Wm);
    /* x_B = W^-1 (h_T - G[T, N] x_N) */
    double *rhs = dalloc(k);
    for (int t = 0; t < k; ++t) {
        const double *g = r->G + (size_t)r->T[t] * n;
        double a = r->h[r->T[t]];
        for (int j = 0; j < n; ++j) if (r->posB[j] < 0) a -= g[j] * r->x[j];
        rhs[t] = a;
    }
    for (int p = 0; p < k; ++p) { double a = 0; for (int t = 0; t < k; ++t) a += r->Winv[(size_t)p * kc + t] * rhs[t]; r->x[r->Bs[p]] = a; }
    for (int i = 0; i < m; ++i) {
        if (r->posT[i] >= 0) { r->s[i] = 0.0; continue; }
        const double *g = r->G + (size_t)i * n;
        double a = r->h[i];
        for (int j = 0; j < n; ++j) a -= g[j] * r->x[j];
        r->s[i] = a;
    }
    /* pi = W^-T q_B ; d_j = q_j - pi . G[T, j] ; dS_t = -pi_t */
    for (int t = 0; t < k; ++t) { double a = 0; for (int p = 0; p < k; ++p) a += r->Winv[(size_t)p * kc + t] * r->q[r->Bs[p]]; rhs[t] = a; }
    for (int j = 0; j < n; ++j) {
        if (r->posB[j] >= 0) { r->d[j] = 0.0; continue; }
        double a = r->q[j];
        for (int t = 0; t < k; ++t) a -= rhs[t] * r->G[(size_t)r->T[t] * n + j];
        r->d[j] = a;
    }
    for (int t = 0; t < k; ++t) r->dS[t] = -rhs[t];
    free(rhs);
    r->refreshes++;
    return 1;
}

/* returns LP_OPTIMAL / LP_INFEASIBLE / LP_ITERLIMIT */
static int rev_dual_simplex(rev_t *r, long max_pivots)
{
    const int n = r->n, m = r->m, kc = r->kcap;
    int stall = 0; double last_obj = -INFINITY; long since = 0;
    memset(r->skipX, 0, (size_t)n); memset(r->skipS, 0, (size_t)m);
    for (;;) {
        if (r->pivots >= max_pivots) return LP_ITERLIMIT;
        if (since >= ORC_REV_REFRESH) { since = 0; if (!rev_refresh(r)) return LP_ITERLIMIT; }
        const int k = r->k;
        double cur = 0; for (int j = 0; j < n; ++j) cur += r->q[j] * r->x[j];
        if (cur > last_obj + 1e-12 * fmax(1.0, fabs(cur))) { stall = 0; last_obj = cur; } else stall++;
        const int bland = stall > 30;
        /* ---- leaving variable: largest bound violation (smallest id while stalling); ids: structural j, slack n + i */
        int lv = -1, lv_bl = -1, id_bl = 0x7fffffff; double best = ORC_PTOL;
        for (int p = 0; p < k; ++p) {
            const int j = r->Bs[p];
            if (r->skipX[j]) continue;
            const double v = fmax(r->lo[j] - r->x[j], r->x[j] - r->hi[j]);
            if (v > ORC_PTOL) { if (v > best) { best = v; lv = j; } if (j < id_bl) { id_bl = j; lv_bl = j; } }
        }
        for (int i = 0; i < m; ++i) {
            if (r->posT[i] >= 0 || r->skipS[i]) continue;
            const double v = -r->s[i];
            if (v > ORC_PTOL) { if (v > best) { best = v; lv = n + i; } if (n + i < id_bl) { id_bl = n + i; lv_bl = n + i; } }
        }
        if (lv < 0) {
            if (since > 0) { since = 0; if (!rev_refresh(r)) return LP_ITERLIMIT; continue; }   /* verify at a clean state before declaring optimality */
            return LP_OPTIMAL;
        }
        if (bland) lv = lv_bl;
        /* ---- its dictionary row over the non-basic structurals (alpha) and the non-basic slacks (alphaS) */
        int below; double viol, xl;
        if (lv < n) {
            const int p = r->posB[lv];
            for (int t = 0; t < k; ++t) r->rho[t] = r->Winv[(size_t)p * kc + t];
            for (int j = 0; j < n; ++j) {
                if (r->posB[j] >= 0) { r->alpha[j] = 0.0; continue; }
                double a = 0; for (int t = 0; t < k; ++t) a += r->rho[t] * r->G[(size_t)r->T[t] * n + j];
                r->alpha[j] = a;
            }
            for (int t = 0; t < k; ++t) r->alphaS[t] = r->rho[t];
            xl = r->x[lv];
            const double vlo = r->lo[lv] - xl, vhi = xl - r->hi[lv];
            below = vlo > vhi; viol = below ? vlo : vhi;
        } else {
            const int i = lv - n;
            const double *gi = r->G + (size_t)i * n;
            for (int t = 0; t < k; ++t) { double a = 0; for (int p = 0; p < k; ++p) a += gi[r->Bs[p]] * r->Winv[(size_t)p * kc + t]; r->rho[t] = a; }
            for (int j = 0; j < n; ++j) {
                if (r->posB[j] >= 0) { r->alpha[j] = 0.0; continue; }
                double a = gi[j]; for (int t = 0; t < k; ++t) a -= r->rho[t] * r->G[(size_t)r->T[t] * n + j];
                r->alpha[j] = a;
            }
            for (int t = 0; t < k; ++t) r->alphaS[t] = -r->rho[t];
            xl = r->s[i]; below = 1; viol = -xl;
        }
        /* ---- ratio test (Harris two-pass; eligibility as in dual_simplex) */
        double emax = 0;
        for (int j = 0; j < n; ++j) {
            if (r->posB[j] >= 0 || r->lo[j] == r->hi[j]) continue;
            const double a = r->alpha[j];
            const int el = below ? (r->up[j] ? a > 0 : a < 0) : (r->up[j] ? a < 0 : a > 0);
            if (el && fabs(a) > emax) emax = fabs(a);
        }
        for (int t = 0; t < k; ++t) { const double a = r->alphaS[t]; const int el = below ? a < 0 : a > 0; if (el && fabs(a) > emax) emax = fabs(a); }
        const double ptol = fmax(ORC_PIV_ABS, ORC_PIV_REL * emax);
        double tmax = INFINITY, rmin = INFINITY; int any = 0;
        for (int j = 0; j < n; ++j) {
            if (r->posB[j] >= 0 || r->lo[j] == r->hi[j]) continue;
            const double a = r->alpha[j];
            const int el = below ? (r->up[j] ? a > 0 : a < 0) : (r->up[j] ? a < 0 : a > 0);
            if (!el || fabs(a) <= ptol) continue;
            any = 1;
            const double da = fmax(r->up[j] ? -r->d[j] : r->d[j], 0.0);
            tmax = fmin(tmax, (da + ORC_DTOL) / fabs(a)); rmin = fmin(rmin, da / fabs(a));
        }
        for (int t = 0; t < k; ++t) {
            const double a = r->alphaS[t];
            const int el = below ? a < 0 : a > 0;
            if (!el || fabs(a) <= ptol) continue;
            any = 1;
            const double da = fmax(r->dS[t], 0.0);
            tmax = fmin(tmax, (da + ORC_DTOL) / fabs(a)); rmin = fmin(rmin, da / fabs(a));
        }
        if (!any) {
            if (viol <= ORC_PTOL_SKIP) { if (lv < n) r->skipX[lv] = 1; else r->skipS[lv - n] = 1; continue; }
            if (since > 0) { since = 0; if (!rev_refresh(r)) return LP_ITERLIMIT; continue; }
            return LP_INFEASIBLE;
        }
        int q = -1; double abest = -1; int idbest = 0x7fffffff;      /* q: structural j, or n + position t for a slack */
        for (int j = 0; j < n; ++j) {
            if (r->posB[j] >= 0 || r->lo[j] == r->hi[j]) continue;
            const double a = r->alpha[j];
            const int el = below ? (r->up[j] ? a > 0 : a < 0) : (r->up[j] ? a < 0 : a > 0);
            if (!el || fabs(a) <= ptol) continue;
            const double r0 = fmax(r->up[j] ? -r->d[j] : r->d[j], 0.0) / fabs(a);
            if (bland) { if (r0 <= rmin * (1 + 1e-12) + 1e-300 && j < idbest) { idbest = j; q = j; } }
            else if (r0 <= tmax && fabs(a) > abest) { abest = fabs(a); q = j; }
        }
        for (int t = 0; t < k; ++t) {
            const double a = r->alphaS[t];
            const int el = below ? a < 0 : a > 0;
            if (!el || fabs(a) <= ptol) continue;
            const double r0 = fmax(r->dS[t], 0.0) / fabs(a);
            const int id = n + r->T[t];
            if (bland) { if (r0 <= rmin * (1 + 1e-12) + 1e-300 && id < idbest) { idbest = id; q = n + t; } }
            else if (r0 <= tmax && fabs(a) > abest) { abest = fabs(a); q = n + t; }
        }
        const double aq = q < n ? r->alpha[q] : r->alphaS[q - n];
        if (fabs(aq) < ORC_PIV_TINY && viol <= ORC_PTOL_SKIP) { if (lv < n) r->skipX[lv] = 1; else r->skipS[lv - n] = 1; continue; }
        /* ---- column of the entering variable: w over the basic structurals, col over the basic slacks */
        if (q < n) {
            for (int p = 0; p < k; ++p) { double a = 0; for (int t = 0; t < k; ++t) a += r->Winv[(size_t)p * kc + t] * r->G[(size_t)r->T[t] * n + q]; r->w[p] = a; }
        } else for (int p = 0; p < k; ++p) r->w[p] = r->Winv[(size_t)p * kc + (q - n)];
        const double leave_value = lv < n ? (below ? r->lo[lv] : r->hi[lv]) : 0.0;
        const double theta = (xl - leave_value) / aq;
        for (int p = 0; p < k; ++p) r->x[r->Bs[p]] -= r->w[p] * theta;
        for (int i = 0; i < m; ++i) {
            if (r->posT[i] >= 0) continue;
            const double *gi = r->G + (size_t)i * n;
            double c = q < n ? gi[q] : 0.0;
            for (int p = 0; p < k; ++p) c -= gi[r->Bs[p]] * r->w[p];
            r->col[i] = c;
            r->s[i] -= c * theta;
        }
        /* ---- reduced costs */
        const double dq = q < n ? r->d[q] : r->dS[q - n];
        const double tau = dq / aq;
        for (int j = 0; j < n; ++j) if (r->posB[j] < 0) r->d[j] -= tau * r->alpha[j];
        for (int t = 0; t < k; ++t) r->dS[t] -= tau * r->alphaS[t];
        /* ---- basis change */
        if (q < n && lv < n) {                   /* structural enters, structural leaves: column p of W replaced */
            const int p = r->posB[lv];
            const double inv = 1.0 / r->w[p];
            for (int t = 0; t < k; ++t) r->Winv[(size_t)p * kc + t] *= inv;
            for (int a = 0; a < k; ++a) { if (a == p) continue; const double f = r->w[a]; if (f != 0.0) for (int t = 0; t < k; ++t) r->Winv[(size_t)a * kc + t] -= f * r->Winv[(size_t)p * kc + t]; }
            r->x[q] += theta; r->x[lv] = leave_value;
            r->Bs[p] = q; r->posB[q] = p; r->posB[lv] = -1;
            r->up[lv] = (leave_value == r->hi[lv]) && (r->lo[lv] != r->hi[lv]);
            r->d[lv] = -tau; r->d[q] = 0.0;
        } else if (q >= n && lv >= n) {          /* slack enters (row leaves T), slack leaves (row becomes tight): row t of W replaced */
            const int t = q - n, i = lv - n, told = r->T[t];
            const double inv = 1.0 / r->rho[t];      /* rho = G[i, B] W^-1 */
            for (int p = 0; p < k; ++p) r->Winv[(size_t)p * kc + t] *= inv;
            for (int b = 0; b < k; ++b) { if (b == t) continue; const double f = r->rho[b]; if (f != 0.0) for (int p = 0; p < k; ++p) r->Winv[(size_t)p * kc + b] -= f * r->Winv[(size_t)p * kc + t]; }
            r->s[told] = theta; r->s[i] = 0.0;
            r->T[t] = i; r->posT[i] = t; r->posT[told] = -1;
            r->dS[t] = -tau;
        } else if (q < n && lv >= n) {           /* structural enters, slack leaves: W bordered by row i and column q (k + 1) */
            const int i = lv - n;
            if (k + 1 > kc) return LP_ITERLIMIT;
            const double sig = aq;                  /* Schur complement G[i, q] - G[i, B] W^-1 G[T, q] = the pivot element */
            const double inv = 1.0 / sig;
            for (int p = 0; p < k; ++p) for (int t = 0; t < k; ++t) r->Winv[(size_t)p * kc + t] += r->w[p] * r->rho[t] * inv;
            for (int p = 0; p < k; ++p) r->Winv[(size_t)p * kc + k] = -r->w[p] * inv;
            for (int t = 0; t < k; ++t) r->Winv[(size_t)k * kc + t] = -r->rho[t] * inv;
            r->Winv[(size_t)k * kc + k] = inv;
            r->x[q] += theta; r->s[i] = 0.0;
            r->Bs[k] = q; r->posB[q] = k; r->T[k] = i; r->posT[i] = k;
            r->dS[k] = -tau; r->d[q] = 0.0;
            r->k = k + 1;
        } else {                                 /* slack enters (row leaves T), structural leaves: row t and column p of W removed */
            const int t = q - n, p = r->posB[lv], told = r->T[t];
            const double inv = 1.0 / r->Winv[(size_t)p * kc + t];
            for (int a = 0; a < k; ++a) {
                if (a == p) continue;
                const double f = r->Winv[(size_t)a * kc + t] * inv;
                if (f != 0.0) for (int b = 0; b < k; ++b) if (b != t) r->Winv[(size_t)a * kc + b] -= f * r->Winv[(size_t)p * kc + b];
            }
            r->s[told] = theta; r->x[lv] = leave_value;
            r->posB[lv] = -1; r->posT[told] = -1;
            r->up[lv] = (leave_value == r->hi[lv]) && (r->lo[lv] != r->hi[lv]);
            r->d[lv] = -tau;
            /* compact: move the last position into the holes (row p of W^-1 <- last row, column t <- last column) */
            const int last = k - 1;
            if (p != last) { for (int b = 0; b < k; ++b) r->Winv[(size_t)p * kc + b] = r->Winv[(size_t)last * kc + b]; r->Bs[p] = r->Bs[last]; r->posB[r->Bs[p]] = p; }
            if (t != last) { for (int a = 0; a < k; ++a) r->Winv[(size_t)a * kc + t] = r->Winv[(size_t)a * kc + last]; r->T[t] = r->T[last]; r->posT[r->T[t]] = t; r->dS[t] = r->dS[last]; }
            r->k = k - 1;
        }
        r->pivots++; since++;
    }
}

/* min q'x  s.t. Gx <= h, lb <= x <= ub (no integers): the LP by the revised method; same scaling as orc_solve_miqp */
int orc_lp_revised(int n, int m, const double *q, const double *G, const double *h, const double *lb, const double *ub, int kcap, long max_pivots,
                   double *x_out, double *obj_out, orc_stats *st)
{
    memset(st, 0, sizeof(*st));
    *obj_out = INFINITY;
    unsigned char *noint = (unsigned char *)calloc(n + 1, 1);
    double *rs = dalloc(m), *cs = dalloc(n);
    equilibrate(G, m, n, noint, rs, cs);
    double *Gs = dalloc((size_t)m * n), *hs = dalloc(m), *qs = dalloc(n);
    for (int i = 0; i < m; ++i) { for (int j = 0; j < n; ++j) Gs[(size_t)i * n + j] = G[(size_t)i * n + j] * rs[i] * cs[j]; hs[i] = h[i] * rs[i]; }
    rev_t R; rev_t *r = &R; memset(r, 0, sizeof(R));
    r->n = n; r->m = m; r->k = 0; r->kcap = kcap; r->G = Gs; r->h = hs; r->q = qs;
    r->lo = dalloc(n); r->hi = dalloc(n); r->x = dalloc(n); r->s = dalloc(m);
    r->posB = (int *)calloc(n + 1, sizeof(int)); r->posT = (int *)calloc(m + 1, sizeof(int));
    r->up = (unsigned char *)calloc(n + 1, 1);
    r->Bs = (int *)calloc(kcap + 1, sizeof(int)); r->T = (int *)calloc(kcap + 1, sizeof(int));
    r->Winv = dalloc((size_t)kcap * kcap); r->d = dalloc(n); r->dS = dalloc(kcap);
    r->alpha = dalloc(n); r->alphaS = dalloc(kcap); r->rho = dalloc(kcap); r->w = dalloc(kcap); r->col = dalloc(m);
    r->skipX = (unsigned char *)calloc(n + 1, 1); r->skipS = (unsigned char *)calloc(m + 1, 1);
    for (int j = 0; j < n; ++j) {
        qs[j] = q[j] * cs[j]; r->lo[j] = lb[j] / cs[j]; r->hi[j] = ub[j] / cs[j];
        r->posB[j] = -1; r->d[j] = qs[j];
        /* slack basis: every structural non-basic at its dual-feasible bound (place() of the dense code) */
        if (r->lo[j] == r->hi[j]) { r->up[j] = 0; r->x[j] = r->lo[j]; }
        else if (qs[j] >= 0) { if (!isfinite(r->lo[j])) r->lo[j] = -ORC_BIG; r->up[j] = 0; r->x[j] = r->lo[j]; }
        else { if (!isfinite(r->hi[j])) r->hi[j] = ORC_BIG; r->up[j] = 1; r->x[j] = r->hi[j]; }
    }
    for (int i = 0; i < m; ++i) r->posT[i] = -1;
    rev_refresh(r);
    const int lp = rev_dual_simplex(r, max_pivots > 0 ? max_pivots : 2000000000L);
    int status = lp == LP_OPTIMAL ? ORC_OPTIMAL : (lp == LP_INFEASIBLE ? ORC_INFEASIBLE : ORC_NUMERICAL);
    if (lp == LP_OPTIMAL) {
        double ob = 0;
        for (int j = 0; j < n; ++j) { x_out[j] = r->x[j] * cs[j]; ob += q[j] * x_out[j]; }
        *obj_out = ob;
    }
    st->pivots = (int)r->pivots; st->refactors = (int)r->refreshes; st->status = status; st->nodes = r->k;
    free(noint); free(rs); free(cs); free(Gs); free(hs); free(qs); free(r->lo); free(r->hi); free(r->x); free(r->s); free(r->posB); free(r->posT);
    free(r->up); free(r->Bs); free(r->T); free(r->Winv); free(r->d); free(r->dS); free(r->alpha); free(r->alphaS); free(r->rho); free(r->w); free(r->col); free(r->skipX); free(r->skipS);
    return status;
}
