/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
 * (arap_flow_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the reference's ARAP Gauss-Newton / PCG solve, instantiated twice by
 * arap_oracle.c (REAL = float -> suffix _f32, REAL = double -> suffix _f64).
 *
 * Every function cites the reference file:line (paths relative to /root/reference) it follows.
 * The reference emits its kernels at run time from a symbolic-AD compiler (ARAP/API/src/o.t), so
 * evalJTF / applyJTJ / cost below are the closed forms of what o.t:2129-2172, o.t:2029-2089 and
 * o.t:2375-2385 generate for the energy in arap_plan.t:1-23.  tests/test_oracle.py pins those
 * closed forms against a finite-difference Jacobian of the residuals (this file's
 * oracle_residuals_*) and pins the whole schedule against the reference's golden vector
 * ARAP/warping/cat512_iFlo.flo.
 *
 * Arithmetic: one IEEE operation per operator (-ffp-contract=off) plus explicit FMA() at the sites DESIGN.md lists;
 * the HIP kernels use the same sites, which is what makes the float32 variant their bit-level twin (tier T3).
 *
 * Layout (o.t:376-387): row major, x fastest, index = x + W*y, channels interleaved.
 *   O, U, C : REAL[N][2]      A, M : REAL[N]      3-vectors (delta, r, p, ...) : REAL[N][3] = (Ox,Oy,A)
 */

#ifndef REAL
#error "include from arap_oracle.c with REAL and SUF defined"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)
/* fused multiply-add a*b + c with ONE rounding, at exactly the sites the HIP kernels fuse */
#undef FMA
/* -DORACLE_NO_FMA (liboracle_nofma.so): the same sites as separate multiply and add, two roundings -- one of the
 * arithmetic variants of tests/golden/make_t4_variants.py (how far does fusing move the unconverged 19/8/400 answer) */
#ifdef ORACLE_NO_FMA
#define FMA(a, b, c) ((REAL)((REAL)(a) * (REAL)(b)) + (REAL)(c))
#else
#define FMA(a, b, c) (sizeof(REAL) == 4 ? (REAL)fmaf((float)(a), (float)(b), (float)(c)) : (REAL)fma((double)(a), (double)(b), (double)(c)))
#endif

/* reduction mode for the three PCG dot products and the cost:
 *   0 : accumulate in REAL, sequentially in index order (one fixed instance of the order the
 *       reference leaves undefined: warp shuffle + atomicAdd, solverGPUGaussNewton.t:312-317)
 *   1 : per-vertex term evaluated in REAL, accumulated in double ("f32sum64" of SURVEY 8c).
 *       Order independent to ~1e-16, which is what the HIP path implements. */
typedef struct {
    int W, H;
    const REAL *U;   /* UrShape     [N][2]  arap_plan.t:4 */
    const REAL *C;   /* Constraints [N][2]  arap_plan.t:5 */
    const REAL *M;   /* Mask        [N]     arap_plan.t:6 */
    REAL wf, wr;     /* w_fitSqrt, w_regSqrt  arap_plan.t:7-8 */
    int trig;        /* 0: libm cos/sin, 1: arap_sincos_spec (bit-twin of the HIP routine) */
} FN(Prob);

/* stencil order of arap_plan.t:14 */
static const int FN(SX)[4] = {1, -1, 0, 0};
static const int FN(SY)[4] = {0, 0, 1, -1};

static inline void FN(sincos_)(const FN(Prob) * pb, REAL a, REAL *c, REAL *s)
{
    if (pb->trig == 1) {
        double cd, sd;
        arap_sincos_spec((double)a, &cd, &sd);
        *c = (REAL)cd;
        *s = (REAL)sd;
    } else if (sizeof(REAL) == 4) {
        *c = (REAL)cosf((float)a);
        *s = (REAL)sinf((float)a);
    } else {
        *c = (REAL)cos((double)a);
        *s = (REAL)sin((double)a);
    }
}

/* Exclude(Not(eq(Mask(0,0),0)))  arap_plan.t:11 */
static inline int FN(act)(const FN(Prob) * pb, int i) { return pb->M[i] == (REAL)0; }

/* valid = InBounds(x,y) * eq(Mask(x,y),0) * eq(Mask(0,0),0)   arap_plan.t:17
 * (centre always in bounds: o.t:1930-1934; out-of-bounds reads give 0: o.t:570-576) */
static inline int FN(edge)(const FN(Prob) * pb, int x, int y, int s, int *n)
{
    int nx = x + FN(SX)[s], ny = y + FN(SY)[s];
    if (nx < 0 || nx >= pb->W || ny < 0 || ny >= pb->H) return 0;
    *n = nx + pb->W * ny;
    return FN(act)(pb, x + pb->W * y) && FN(act)(pb, *n);
}

/* valid = All(greatereq(Constraints(0,0),0))   arap_plan.t:22, lib.t:13-19 */
static inline int FN(fit)(const FN(Prob) * pb, int i)
{
    return pb->C[2 * i] >= (REAL)0 && pb->C[2 * i + 1] >= (REAL)0;
}

/* Reductions are evaluated as per-row partial sums (x ascending) combined in row order, so the
 * result does not depend on the OpenMP thread count. */
static double FN(combine_rows)(const double *rowd, const REAL *rowr, int H, int mode)
{
    if (mode == 1) { double a = 0.0; for (int y = 0; y < H; ++y) a += rowd[y]; return (double)(REAL)a; }
    REAL a = (REAL)0; for (int y = 0; y < H; ++y) a = a + rowr[y]; return (double)a;
}

/* ---------------------------------------------------------------------------------------------
 * residuals, arap_plan.t:14-23 :  e_s(c) = w_r[(O(c)-O(n)) - R(A(c))(U(c)-U(n))],  f(c) = w_f(O(c)-C(c))
 * Rotate2D: lib.t:92-96.  out[N][10] = e_0.xy, e_1.xy, e_2.xy, e_3.xy, f.xy (0 where invalid or
 * where the centre is excluded).  Used for the cost and by the finite-difference pin test.
 * ------------------------------------------------------------------------------------------- */
static void FN(residuals_at)(const FN(Prob) * pb, const REAL *O, const REAL *A, int x, int y, REAL out[10])
{
    int i = x + pb->W * y, n;
    for (int k = 0; k < 10; ++k) out[k] = (REAL)0;
    if (!FN(act)(pb, i)) return;
    REAL c, s;
    FN(sincos_)(pb, A[i], &c, &s);
    for (int k = 0; k < 4; ++k) {
        if (!FN(edge)(pb, x, y, k, &n)) continue;
        REAL dx = pb->U[2 * i] - pb->U[2 * n], dy = pb->U[2 * i + 1] - pb->U[2 * n + 1];
        REAL rx = FMA(c, dx, -(s * dy)), ry = FMA(s, dx, c * dy);
        out[2 * k] = pb->wr * ((O[2 * i] - O[2 * n]) - rx);
        out[2 * k + 1] = pb->wr * ((O[2 * i + 1] - O[2 * n + 1]) - ry);
    }
    if (FN(fit)(pb, i)) {
        out[8] = pb->wf * (O[2 * i] - pb->C[2 * i]);
        out[9] = pb->wf * (O[2 * i + 1] - pb->C[2 * i + 1]);
    }
}

void FN(oracle_residuals)(int W, int H, const REAL *O, const REAL *A, const REAL *U, const REAL *C,
                          const REAL *M, REAL wf, REAL wr, REAL *out /* [N][10] */)
{
    FN(Prob) pb = {W, H, U, C, M, wf, wr, g_oracle_trig};
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) FN(residuals_at)(&pb, O, A, x, y, out + 10 * (size_t)(x + W * y));
}

/* cost: o.t:2375-2385 (0.5 * sum of squares of the residuals centred on a vertex), excluded centres
 * skipped (solverGPUGaussNewton.t:580-592); device float upconverted to double (:1179-1182). */
static double FN(cost_)(const FN(Prob) * pb, const REAL *O, const REAL *A, int mode)
{
    int W = pb->W, H = pb->H;
    double *rowd = (double *)calloc((size_t)H, sizeof(double));
    REAL *rowr = (REAL *)calloc((size_t)H, sizeof(REAL));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        double accd = 0.0;
        REAL accr = (REAL)0;
        for (int x = 0; x < W; ++x) {
            if (!FN(act)(pb, x + W * y)) continue;
            REAL e[10];
            FN(residuals_at)(pb, O, A, x, y, e);
            REAL t = (REAL)0;
            for (int k = 0; k < 10; ++k) t = FMA(e[k], e[k], t);
            t = (REAL)0.5 * t;
            if (mode == 1) accd += (double)t; else accr = accr + t;
        }
        rowd[y] = accd; rowr[y] = accr;
    }
    double out = FN(combine_rows)(rowd, rowr, H, mode);
    free(rowd); free(rowr);
    return out;
}

double FN(oracle_cost)(int W, int H, const REAL *O, const REAL *A, const REAL *U, const REAL *C,
                       const REAL *M, REAL wf, REAL wr, int mode)
{
    FN(Prob) pb = {W, H, U, C, M, wf, wr, g_oracle_trig};
    return FN(cost_)(&pb, O, A, mode);
}

/* ---------------------------------------------------------------------------------------------
 * evalJTF: gradient J^T F (factor 1.0) and diag(J^T J), o.t:2129-2172.  For unknown c the
 * residuals that contain it are its own four e_s(c), its f(c), and e_{-s}(n) of each neighbour n
 * (residualsincludingX00, o.t:2143).  Closed form: SURVEY Appendix A.
 * ------------------------------------------------------------------------------------------- */
static void FN(evalJTF_at)(const FN(Prob) * pb, const REAL *O, const REAL *A, int x, int y,
                           REAL g[3], REAL d[3])
{
    int i = x + pb->W * y, n;
    REAL wr = pb->wr, wf = pb->wf;
    REAL ci, si;
    FN(sincos_)(pb, A[i], &ci, &si);
    REAL gx = 0, gy = 0, ga = 0, dO = 0, dA = 0;
    for (int k = 0; k < 4; ++k) {
        if (!FN(edge)(pb, x, y, k, &n)) continue;
        REAL cn, sn;
        FN(sincos_)(pb, A[n], &cn, &sn);
        REAL dx = pb->U[2 * i] - pb->U[2 * n], dy = pb->U[2 * i + 1] - pb->U[2 * n + 1];
        REAL ox = O[2 * i] - O[2 * n], oy = O[2 * i + 1] - O[2 * n + 1];
        /* e_s(c) */
        REAL ex = wr * (ox - FMA(ci, dx, -(si * dy)));
        REAL ey = wr * (oy - FMA(si, dx, ci * dy));
        /* e_{-s}(n) = w_r[-(O(c)-O(n)) + R(A(n)) d_s(c)] */
        REAL fx = wr * (FMA(cn, dx, -(sn * dy)) - ox);
        REAL fy = wr * (FMA(sn, dx, cn * dy) - oy);
        /* q_s(c) = R'(A(c)) d_s(c) */
        REAL qx = FMA(-si, dx, -(ci * dy)), qy = FMA(ci, dx, -(si * dy));
        gx = FMA(wr, ex - fx, gx);
        gy = FMA(wr, ey - fy, gy);
        ga = FMA(-wr, FMA(qx, ex, qy * ey), ga);
        dO = dO + (wr * wr + wr * wr);
        dA = FMA(wr * wr, FMA(qx, qx, qy * qy), dA);
    }
    REAL dOf = dO;
    if (FN(fit)(pb, i)) {
        gx = FMA(wf, wf * (O[2 * i] - pb->C[2 * i]), gx);
        gy = FMA(wf, wf * (O[2 * i + 1] - pb->C[2 * i + 1]), gy);
        dOf = FMA(wf, wf, dO);
    }
    g[0] = gx; g[1] = gy; g[2] = ga;
    d[0] = dOf; d[1] = dOf; d[2] = dA;
}

void FN(oracle_evalJTF)(int W, int H, const REAL *O, const REAL *A, const REAL *U, const REAL *C,
                        const REAL *M, REAL wf, REAL wr, REAL *g /* [N][3] */, REAL *d /* [N][3] */)
{
    FN(Prob) pb = {W, H, U, C, M, wf, wr, g_oracle_trig};
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)(x + W * y);
            if (FN(act)(&pb, (int)i)) FN(evalJTF_at)(&pb, O, A, x, y, g + 3 * i, d + 3 * i);
            else for (int k = 0; k < 3; ++k) g[3 * i + k] = d[3 * i + k] = 0;
        }
}

/* ---------------------------------------------------------------------------------------------
 * applyJTJ: (J^T J P)(c), factor 1.0, o.t:2029-2089; A and U frozen (they are read from the
 * unknown images, which PCG does not touch until PCGLinearUpdate).
 * cs[N][2] = (cos A, sin A) cached once per Gauss-Newton step by the caller.
 * ------------------------------------------------------------------------------------------- */
static void FN(applyJTJ_at)(const FN(Prob) * pb, const REAL *cs, const REAL *P, int x, int y, REAL out[3])
{
    int i = x + pb->W * y, n;
    REAL wr2 = pb->wr * pb->wr;
    REAL ci = cs[2 * i], si = cs[2 * i + 1];
    REAL ax = 0, ay = 0, aa = 0;
    for (int k = 0; k < 4; ++k) {
        if (!FN(edge)(pb, x, y, k, &n)) continue;
        REAL cn = cs[2 * n], sn = cs[2 * n + 1];
        REAL dx = pb->U[2 * i] - pb->U[2 * n], dy = pb->U[2 * i + 1] - pb->U[2 * n + 1];
        REAL qx = FMA(-si, dx, -(ci * dy)), qy = FMA(ci, dx, -(si * dy));   /* R'(A(c)) d */
        REAL hx = FMA(-sn, dx, -(cn * dy)), hy = FMA(cn, dx, -(sn * dy));   /* R'(A(n)) d */
        REAL px = P[3 * i] - P[3 * n], py = P[3 * i + 1] - P[3 * n + 1];
        REAL pa = P[3 * i + 2], pn = P[3 * n + 2];
        REAL tx = FMA(-qx, pa, px), ty = FMA(-qy, pa, py);      /* dP - q P_A(c) */
        ax = FMA(wr2, FMA(-hx, pn, px + tx), ax);
        ay = FMA(wr2, FMA(-hy, pn, py + ty), ay);
        aa = FMA(-wr2, FMA(qx, tx, qy * ty), aa);
    }
    if (FN(fit)(pb, i)) {
        REAL wf2 = pb->wf * pb->wf;
        ax = FMA(wf2, P[3 * i], ax);
        ay = FMA(wf2, P[3 * i + 1], ay);
    }
    out[0] = ax; out[1] = ay; out[2] = aa;
}

static void FN(fill_cs)(const FN(Prob) * pb, const REAL *A, REAL *cs)
{
    int N = pb->W * pb->H;
    for (int i = 0; i < N; ++i) FN(sincos_)(pb, A[i], &cs[2 * i], &cs[2 * i + 1]);
}

void FN(oracle_applyJTJ)(int W, int H, const REAL *A, const REAL *U, const REAL *C, const REAL *M,
                         REAL wf, REAL wr, const REAL *P /* [N][3] */, REAL *out /* [N][3] */)
{
    FN(Prob) pb = {W, H, U, C, M, wf, wr, g_oracle_trig};
    size_t N = (size_t)W * H;
    REAL *cs = (REAL *)malloc(sizeof(REAL) * 2 * N);
    FN(fill_cs)(&pb, A, cs);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)(x + W * y);
            if (FN(act)(&pb, (int)i)) FN(applyJTJ_at)(&pb, cs, P, x, y, out + 3 * i);
            else out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0;
        }
    free(cs);
}

/* guardedInvert, CERES variant: solverGPUGaussNewton.t:323-332 (selected at :22) */
static inline REAL FN(ginv)(REAL d)
{
    REAL t = (REAL)1 + (sizeof(REAL) == 4 ? (REAL)sqrtf((float)d) : (REAL)sqrt((double)d));
    return (REAL)1 / (t * t);
}

static inline REAL FN(dot3)(const REAL *a, const REAL *b)
{
    return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0]));
}

/* solver state that the reference keeps in its plan (makePlan, solverGPUGaussNewton.t:1254-1284);
 * only the six images the GN path touches. */
typedef struct {
    size_t N;
    REAL *delta, *r, *z, *p, *Ap, *pre, *cs;
} FN(Plan);

static void FN(plan_alloc)(FN(Plan) * pl, size_t N)
{
    pl->N = N;
    pl->delta = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->r = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->z = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->p = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->Ap = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->pre = (REAL *)calloc(3 * N, sizeof(REAL));
    pl->cs = (REAL *)calloc(2 * N, sizeof(REAL));
}
static void FN(plan_free)(FN(Plan) * pl)
{
    free(pl->delta); free(pl->r); free(pl->z); free(pl->p); free(pl->Ap); free(pl->pre); free(pl->cs);
}

/* One Opt_ProblemSolve (o.t:2548-2551 = init + step until 0):
 *   init  solverGPUGaussNewton.t:956-1007   nIter = 0, prevCost = cost
 *   step  :1016-1177                        PCGInit1, lIterations x (PCGStep1, PCGStep2, PCGStep3),
 *                                           PCGLinearUpdate, computeCost
 * O and A are updated in place.  costs[0] = cost at init, costs[1+k] = cost after GN step k
 * (costs may be NULL).  Returns the final cost (Opt_ProblemCurrentCost, :1179-1182).
 * No convergence test on the GN path (the zeta break at :1093-1102 is LM only). */
static double FN(solve_)(const FN(Prob) * pb, FN(Plan) * pl, REAL *O, REAL *A, int nIterations,
                         int lIterations, int mode, double *costs)
{
    const int W = pb->W, H = pb->H;
    double *rowd = (double *)calloc((size_t)H, sizeof(double));
    REAL *rowr = (REAL *)calloc((size_t)H, sizeof(REAL));
    /* rows that contain at least one active vertex (pure speed-up: inactive vertices are skipped
     * by every kernel of the reference, so skipping whole rows changes nothing) */
    int y0 = H, y1 = -1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (FN(act)(pb, x + W * y)) { if (y < y0) y0 = y; if (y > y1) y1 = y; break; }
    double prevCost = FN(cost_)(pb, O, A, mode);
    if (costs) costs[0] = prevCost;
    for (int nIter = 0; nIter < nIterations; ++nIter) {
        /* ---- PCGInit1 (:361-397) ---- */
        FN(fill_cs)(pb, A, pl->cs);
        memset(rowd, 0, sizeof(double) * (size_t)H); memset(rowr, 0, sizeof(REAL) * (size_t)H);
#pragma omp parallel for schedule(static)
        for (int y = y0; y <= y1; ++y) {
            double sd = 0.0; REAL sr = 0;
            for (int x = 0; x < W; ++x) {
                size_t i = (size_t)(x + W * y);
                REAL *pre = pl->pre + 3 * i;
                if (!FN(act)(pb, (int)i)) { pre[0] = pre[1] = pre[2] = 0; continue; }
                REAL g[3], d[3];
                FN(evalJTF_at)(pb, O, A, x, y, g, d);
                REAL *r = pl->r + 3 * i, *p = pl->p + 3 * i, *dl = pl->delta + 3 * i;
                for (int k = 0; k < 3; ++k) {
                    dl[k] = 0;
                    r[k] = -g[k];
                    pre[k] = FN(ginv)(d[k]);
                    p[k] = pre[k] * r[k];
                }
                REAL t = FN(dot3)(r, p);
                if (mode == 1) sd += (double)t; else sr = sr + t;
            }
            rowd[y] = sd; rowr[y] = sr;
        }
        REAL rho = (REAL)FN(combine_rows)(rowd, rowr, H, mode);      /* scanAlphaNumerator */
        for (int l = 0; l < lIterations; ++l) {
            /* ---- PCGStep1 (:421-434): Ap = J^T J p ; sigma = p.Ap ---- */
#pragma omp parallel for schedule(static)
            for (int y = y0; y <= y1; ++y) {
                double sd = 0.0; REAL sr = 0;
                for (int x = 0; x < W; ++x) {
                    size_t i = (size_t)(x + W * y);
                    if (!FN(act)(pb, (int)i)) continue;
                    FN(applyJTJ_at)(pb, pl->cs, pl->p, x, y, pl->Ap + 3 * i);
                    REAL t = FN(dot3)(pl->p + 3 * i, pl->Ap + 3 * i);
                    if (mode == 1) sd += (double)t; else sr = sr + t;
                }
                rowd[y] = sd; rowr[y] = sr;
            }
            REAL sigma = (REAL)FN(combine_rows)(rowd, rowr, H, mode); /* scanAlphaDenominator */
            /* ---- PCGStep2 (:446-489) ---- */
            REAL alpha = 0;
            if (sigma > (REAL)0) alpha = rho / sigma;
#pragma omp parallel for schedule(static)
            for (int y = y0; y <= y1; ++y) {
                double sd = 0.0; REAL sr = 0;
                for (int x = 0; x < W; ++x) {
                    size_t i = (size_t)(x + W * y);
                    if (!FN(act)(pb, (int)i)) continue;
                    REAL *dl = pl->delta + 3 * i, *r = pl->r + 3 * i, *z = pl->z + 3 * i;
                    const REAL *p = pl->p + 3 * i, *Ap = pl->Ap + 3 * i, *pre = pl->pre + 3 * i;
                    for (int k = 0; k < 3; ++k) {
                        dl[k] = FMA(alpha, p[k], dl[k]);
                        r[k] = FMA(-alpha, Ap[k], r[k]);
                        z[k] = pre[k] * r[k];
                    }
                    REAL t = FN(dot3)(z, r);
                    if (mode == 1) sd += (double)t; else sr = sr + t;
                }
                rowd[y] = sd; rowr[y] = sr;
            }
            REAL rhoNew = (REAL)FN(combine_rows)(rowd, rowr, H, mode); /* scanBetaNumerator */
            /* ---- PCGStep3 (:537-550) ---- */
            REAL beta = 0;
            if (rho > (REAL)0) beta = rhoNew / rho;
#pragma omp parallel for schedule(static)
            for (int y = y0; y <= y1; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t i = (size_t)(x + W * y);
                    if (!FN(act)(pb, (int)i)) continue;
                    for (int k = 0; k < 3; ++k) pl->p[3 * i + k] = FMA(beta, pl->p[3 * i + k], pl->z[3 * i + k]);
                }
            rho = rhoNew;                                /* D2D copy alphaNum <- betaNum (:1091) */
        }
        /* ---- PCGLinearUpdate (:552-557) ---- */
#pragma omp parallel for schedule(static)
        for (int y = y0; y <= y1; ++y)
            for (int x = 0; x < W; ++x) {
                size_t i = (size_t)(x + W * y);
                if (!FN(act)(pb, (int)i)) continue;
                O[2 * i] = O[2 * i] + pl->delta[3 * i];
                O[2 * i + 1] = O[2 * i + 1] + pl->delta[3 * i + 1];
                A[i] = A[i] + pl->delta[3 * i + 2];
            }
        prevCost = FN(cost_)(pb, O, A, mode);            /* computeCost (:1117), prevCost = newCost (:1161) */
        if (costs) costs[1 + nIter] = prevCost;
    }
    free(rowd); free(rowr);
    return prevCost;
}

double FN(oracle_solve)(int W, int H, REAL *O, REAL *A, const REAL *U, const REAL *C, const REAL *M,
                        REAL wf, REAL wr, int nIterations, int lIterations, int mode, int trig,
                        double *costs)
{
    FN(Prob) pb = {W, H, U, C, M, wf, wr, trig};
    FN(Plan) pl;
    FN(plan_alloc)(&pl, (size_t)W * H);
    double c = FN(solve_)(&pb, &pl, O, A, nIterations, lIterations, mode, costs);
    FN(plan_free)(&pl);
    return c;
}

/* ---------------------------------------------------------------------------------------------
 * "LMGPU" solver kind of the same API: the Levenberg-Marquardt branch of
 * solverGPUGaussNewton.t (UsesLambda): PCGSaveSSq :622-627, PCGComputeCtC :616-621 with computeCtC =
 * diag(J^T J)/trust_region_radius (o.t:2255-2287), PCGFinalizeDiagonal :629-662, applyJTJ + CtC*P
 * (o.t:2076-2082), PCGStep2 with q :477-482, residual reset :491-535 (PCGStep2_1stHalf, computeAdelta,
 * PCGStep2_2ndHalf), zeta break :1093-1102, model cost (o.t:2180-2201, :665-678), accept/reject and trust
 * region update :1119-1157.  The application never selects it (CombinedSolverBase.h:75-77) and the reference
 * holds no LM output: parity of this branch is pinned to this restatement only ("parity unpinned").
 * lm[9] = min_relative_decrease, min_trust_region_radius, max_trust_region_radius, q_tolerance,
 *         function_tolerance, trust_region_radius, radius_decrease_factor, min_lm_diagonal, max_lm_diagonal
 * Returns the number of steps taken; costs[0..steps] as for the GN solve; *final_radius out. */
static REAL FN(clampr)(REAL x, REAL lo, REAL hi) { REAL m = x > lo ? x : lo; return m < hi ? m : hi; }

static void FN(apply_lm_at)(const FN(Prob) * pb, const REAL *cs, const REAL *CtC, const REAL *P, int x, int y, REAL out[3])
{
    size_t i = (size_t)(x + pb->W * y);
    FN(applyJTJ_at)(pb, cs, P, x, y, out);
    for (int k = 0; k < 3; ++k) out[k] = FMA(CtC[3 * i + k], P[3 * i + k], out[k]);
}

int FN(oracle_solve_lm)(int W, int H, REAL *O, REAL *A, const REAL *U, const REAL *C, const REAL *M, REAL wf,
                        REAL wr, int nIterations, int lIterations, int residual_reset_period, const REAL *lm,
                        int trig, double *costs, REAL *final_radius)
{
    FN(Prob) pbv = {W, H, U, C, M, wf, wr, trig};
    const FN(Prob) *pb = &pbv;
    const size_t N = (size_t)W * H;
    FN(Plan) plv; FN(Plan) *pl = &plv;
    FN(plan_alloc)(pl, N);
    REAL *b = (REAL *)calloc(3 * N, sizeof(REAL)), *CtC = (REAL *)calloc(3 * N, sizeof(REAL));
    REAL *SSq = (REAL *)calloc(3 * N, sizeof(REAL)), *Ad = (REAL *)calloc(3 * N, sizeof(REAL));
    REAL *pO = (REAL *)malloc(2 * N * sizeof(REAL)), *pA = (REAL *)malloc(N * sizeof(REAL));
    const REAL min_rel = lm[0], min_rad = lm[1], max_rad = lm[2], q_tol = lm[3], f_tol = lm[4];
    REAL radius = lm[5], dec = lm[6];
    const REAL min_diag = lm[7], max_diag = lm[8];
    const int mode = 1;
    double prevCost = FN(cost_)(pb, O, A, mode);
    costs[0] = prevCost;
    int steps = 0;
    for (int nIter = 0; nIter < nIterations; ++nIter) {
        FN(fill_cs)(pb, A, pl->cs);
        double sd = 0.0, sq = 0.0;
        const REAL inv_radius = (REAL)1 / radius;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t i = (size_t)(x + W * y);
                if (!FN(act)(pb, (int)i)) continue;
                REAL g[3], d[3];
                FN(evalJTF_at)(pb, O, A, x, y, g, d);
                for (int k = 0; k < 3; ++k) {
                    pl->delta[3 * i + k] = 0;
                    pl->r[3 * i + k] = -g[k];
                    pl->pre[3 * i + k] = FN(ginv)(d[k]);                       /* PCGInit1 */
                    if (nIter == 0) SSq[3 * i + k] = pl->pre[3 * i + k];      /* PCGSaveSSq, ONCE_PER_SOLVE */
                    const REAL unclamped = d[k] * inv_radius;                  /* PCGComputeCtC */
                    const REAL mult = ((REAL)1 / SSq[3 * i + k]) / radius;     /* PCGFinalizeDiagonal */
                    const REAL ctc = FN(clampr)(unclamped, min_diag * mult, max_diag * mult);
                    CtC[3 * i + k] = ctc;
                    pl->pre[3 * i + k] = (REAL)1 / FMA(radius, unclamped, ctc);
                    b[3 * i + k] = pl->r[3 * i + k];
                    pl->p[3 * i + k] = pl->pre[3 * i + k] * pl->r[3 * i + k];
                }
                sd += (double)FN(dot3)(pl->r + 3 * i, pl->p + 3 * i);
                {   REAL rr[3]; for (int k = 0; k < 3; ++k) rr[k] = pl->r[3 * i + k] + pl->r[3 * i + k];
                    sq += (double)((REAL)0.5 * FN(dot3)(pl->delta + 3 * i, rr)); }
            }
        REAL rho = (REAL)sd, Q0 = (REAL)sq;
        for (int l = 0; l < lIterations; ++l) {
            sd = 0.0;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t i = (size_t)(x + W * y);
                    if (!FN(act)(pb, (int)i)) continue;
                    FN(apply_lm_at)(pb, pl->cs, CtC, pl->p, x, y, pl->Ap + 3 * i);
                    sd += (double)FN(dot3)(pl->p + 3 * i, pl->Ap + 3 * i);
                }
            const REAL sigma = (REAL)sd;
            REAL alpha = 0;
            if (sigma > (REAL)0) alpha = rho / sigma;
            sd = 0.0; sq = 0.0;
            if (((l + 1) % residual_reset_period) == 0) {
                for (size_t i = 0; i < N; ++i) {                               /* PCGStep2_1stHalf */
                    if (!FN(act)(pb, (int)i)) continue;
                    for (int k = 0; k < 3; ++k) pl->delta[3 * i + k] = FMA(alpha, pl->p[3 * i + k], pl->delta[3 * i + k]);
                }
                for (int y = 0; y < H; ++y)                                    /* computeAdelta */
                    for (int x = 0; x < W; ++x) {
                        size_t i = (size_t)(x + W * y);
                        if (FN(act)(pb, (int)i)) FN(apply_lm_at)(pb, pl->cs, CtC, pl->delta, x, y, Ad + 3 * i);
                    }
                for (size_t i = 0; i < N; ++i) {                               /* PCGStep2_2ndHalf */
                    if (!FN(act)(pb, (int)i)) continue;
                    REAL rb[3];
                    for (int k = 0; k < 3; ++k) {
                        pl->r[3 * i + k] = b[3 * i + k] - Ad[3 * i + k];
                        pl->z[3 * i + k] = pl->pre[3 * i + k] * pl->r[3 * i + k];
                        rb[k] = pl->r[3 * i + k] + b[3 * i + k];
                    }
                    sd += (double)FN(dot3)(pl->z + 3 * i, pl->r + 3 * i);
                    sq += (double)((REAL)0.5 * FN(dot3)(pl->delta + 3 * i, rb));
                }
            } else {
                for (size_t i = 0; i < N; ++i) {                               /* PCGStep2 with q */
                    if (!FN(act)(pb, (int)i)) continue;
                    REAL rb[3];
                    for (int k = 0; k < 3; ++k) {
                        pl->delta[3 * i + k] = FMA(alpha, pl->p[3 * i + k], pl->delta[3 * i + k]);
                        pl->r[3 * i + k] = FMA(-alpha, pl->Ap[3 * i + k], pl->r[3 * i + k]);
                        pl->z[3 * i + k] = pl->pre[3 * i + k] * pl->r[3 * i + k];
                        rb[k] = pl->r[3 * i + k] + b[3 * i + k];
                    }
                    sd += (double)FN(dot3)(pl->z + 3 * i, pl->r + 3 * i);
                    sq += (double)((REAL)0.5 * FN(dot3)(pl->delta + 3 * i, rb));
                }
            }
            const REAL rhoNew = (REAL)sd;
            REAL beta = 0;
            if (rho > (REAL)0) beta = rhoNew / rho;
            for (size_t i = 0; i < N; ++i) {
                if (!FN(act)(pb, (int)i)) continue;
                for (int k = 0; k < 3; ++k) pl->p[3 * i + k] = FMA(beta, pl->p[3 * i + k], pl->z[3 * i + k]);
            }
            rho = rhoNew;
            const REAL Q1 = (REAL)sq;
            const REAL zeta = (REAL)(l + 1) * (Q1 - Q0) / Q1;
            if (zeta < q_tol) break;
            Q0 = Q1;
        }
        /* model cost: 0.5 * sum (F + J delta)^2 over the residuals centred on active vertices */
        double mc = 0.0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t i = (size_t)(x + W * y);
                int n;
                if (!FN(act)(pb, (int)i)) continue;
                REAL e[10];
                FN(residuals_at)(pb, O, A, x, y, e);
                const REAL ci = pl->cs[2 * i], si = pl->cs[2 * i + 1];
                REAL t = 0;
                for (int k = 0; k < 4; ++k) {
                    if (!FN(edge)(pb, x, y, k, &n)) continue;
                    REAL dx = U[2 * i] - U[2 * n], dy = U[2 * i + 1] - U[2 * n + 1];
                    REAL qx = FMA(-si, dx, -(ci * dy)), qy = FMA(ci, dx, -(si * dy));
                    REAL mx = FMA(wr, FMA(-qx, pl->delta[3 * i + 2], pl->delta[3 * i] - pl->delta[3 * n]), e[2 * k]);
                    REAL my = FMA(wr, FMA(-qy, pl->delta[3 * i + 2], pl->delta[3 * i + 1] - pl->delta[3 * n + 1]), e[2 * k + 1]);
                    t = FMA(mx, mx, t); t = FMA(my, my, t);
                }
                if (FN(fit)(pb, (int)i)) {
                    REAL mx = FMA(wf, pl->delta[3 * i], e[8]), my = FMA(wf, pl->delta[3 * i + 1], e[9]);
                    t = FMA(mx, mx, t); t = FMA(my, my, t);
                }
                mc += (double)((REAL)0.5 * t);
            }
        const REAL model_cost_change = (REAL)prevCost - (REAL)mc;
        memcpy(pO, O, 2 * N * sizeof(REAL)); memcpy(pA, A, N * sizeof(REAL));      /* savePreviousUnknowns */
        for (size_t i = 0; i < N; ++i) {
            if (!FN(act)(pb, (int)i)) continue;
            O[2 * i] = O[2 * i] + pl->delta[3 * i]; O[2 * i + 1] = O[2 * i + 1] + pl->delta[3 * i + 1];
            A[i] = A[i] + pl->delta[3 * i + 2];
        }
        const double newCost = FN(cost_)(pb, O, A, mode);
        const REAL cost_change = (REAL)prevCost - (REAL)newCost;
        const REAL relative_decrease = cost_change / model_cost_change;
        ++steps;
        if (cost_change >= 0 && relative_decrease > min_rel) {
            if (cost_change <= (REAL)prevCost * f_tol) { costs[steps] = prevCost; --steps; break; }   /* exits, cost stays */
            const double sqv = (double)relative_decrease, tmp = 1.0 - pow(2.0 * sqv - 1.0, 3.0);
            radius = (REAL)((double)radius / fmax(1.0 / 3.0, tmp));
            radius = (REAL)fmin((double)radius, (double)max_rad);
            dec = (REAL)2;
            prevCost = newCost;
        } else {
            memcpy(O, pO, 2 * N * sizeof(REAL)); memcpy(A, pA, N * sizeof(REAL)); /* revertUpdate */
            radius = radius / dec;
            dec = (REAL)2 * dec;
            if (radius <= min_rad) { costs[steps] = prevCost; --steps; break; }
        }
        costs[steps] = prevCost;
    }
    *final_radius = radius;
    FN(plan_free)(pl);
    free(b); free(CtC); free(SSq); free(Ad); free(pO); free(pA);
    return steps;
}

/* setConstraintImage(alpha): ARAP/deformation/src/CombinedSolver.h:223-242.  float arithmetic in
 * the reference regardless of solver precision: computed in float, then widened. */
static void FN(constraint_image)(int W, int H, const unsigned char *mask_red, const int *cons, int ncons,
                                 float alpha, REAL *C)
{
    size_t N = (size_t)W * H;
    for (size_t i = 0; i < N; ++i) C[2 * i] = C[2 * i + 1] = (REAL)-1.0f;
    for (int k = 0; k < ncons; ++k) {
        int x = cons[4 * k], y = cons[4 * k + 1];
        if (x < 0 || x >= W || y < 0 || y >= H) continue; /* reference would read out of bounds */
        if (mask_red[x + (size_t)W * y] == 0) {
            float nx = (1.0f - alpha) * (float)x + alpha * (float)cons[4 * k + 2];
            float ny = (1.0f - alpha) * (float)y + alpha * (float)cons[4 * k + 3];
            C[2 * (x + (size_t)W * y)] = (REAL)nx;
            C[2 * (x + (size_t)W * y) + 1] = (REAL)ny;
        }
    }
}

/* Full frame schedule of arap_deform:
 *   border pins        ARAP/deformation/src/main.cpp:130-136 (appended after the file constraints)
 *   resetGPU           CombinedSolver.h:207-221   U = O = (x,y), A = 0, Mask = (float)red
 *   weights            CombinedSolver.h:173-177   w_fit = sqrt(100), w_reg = sqrt(0.01)
 *   ramp + warm start  CombinedSolverBase.h:99-120, CombinedSolver.h:199-201: for i < numIter:
 *                      setConstraintImage((i+1)/numIter); solve(nIterations, lIterations)
 *   (numIter == 1 runs a single solve with alpha = 1: CombinedSolverBase.h:101-106)
 * cons: ncons x 4 ints (x1 y1 x2 y2) from the constraint file (main.cpp:26-50).
 * O_out[N][2], A_out[N]; final_costs[numIter] (may be NULL) = Opt_ProblemCurrentCost after each solve. */
void FN(oracle_frame)(int W, int H, const unsigned char *mask_red, const int *cons, int ncons,
                      int add_border_pins, int numIter, int nIterations, int lIterations, int mode,
                      int trig, REAL *O_out, REAL *A_out, double *final_costs)
{
    size_t N = (size_t)W * H;
    int nb = add_border_pins ? (2 * (W + H) - 4) : 0;
    if (add_border_pins && (W < 2 || H < 2)) nb = W * H;
    int *all = (int *)malloc(sizeof(int) * 4 * (size_t)(ncons + nb + 1));
    memcpy(all, cons, sizeof(int) * 4 * (size_t)ncons);
    int nall = ncons;
    if (add_border_pins)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                if (y == 0 || x == 0 || y == H - 1 || x == W - 1) {
                    all[4 * nall] = x; all[4 * nall + 1] = y; all[4 * nall + 2] = x; all[4 * nall + 3] = y;
                    ++nall;
                }
    REAL *U = (REAL *)malloc(sizeof(REAL) * 2 * N), *C = (REAL *)malloc(sizeof(REAL) * 2 * N);
    REAL *M = (REAL *)malloc(sizeof(REAL) * N);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)(x + W * y);
            U[2 * i] = O_out[2 * i] = (REAL)(float)x;
            U[2 * i + 1] = O_out[2 * i + 1] = (REAL)(float)y;
            A_out[i] = 0;
            M[i] = (REAL)(float)mask_red[i];
        }
    float wfit = sqrtf(100.0f), wreg = sqrtf(0.01f);
    FN(Prob) pb = {W, H, U, C, M, (REAL)wfit, (REAL)wreg, trig};
    FN(Plan) pl;
    FN(plan_alloc)(&pl, N);
    for (int i = 0; i < numIter; ++i) {
        float alpha = (float)(i + 1) / (float)numIter;
        FN(constraint_image)(W, H, mask_red, all, nall, alpha, C);
        double c = FN(solve_)(&pb, &pl, O_out, A_out, nIterations, lIterations, mode, NULL);
        if (final_costs) final_costs[i] = c;
    }
    FN(plan_free)(&pl);
    free(U); free(C); free(M); free(all);
}

/* exported so tests can build Constraints images exactly as the reference host does */
void FN(oracle_constraint_image)(int W, int H, const unsigned char *mask_red, const int *cons, int ncons,
                                 float alpha, REAL *C)
{
    FN(constraint_image)(W, H, mask_red, cons, ncons, alpha, C);
}

#undef CAT_
#undef CAT
#undef FN
