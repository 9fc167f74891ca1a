/*
 * libmldgpu -- MI355X-native (gfx950) batched MLD-MPC solve path.  Plain C ABI: caller-owned,
 * C-contiguous row-major `double` arrays; the library copies in/out and never keeps a host pointer
 * past the call.  Every function returns 0 on success and a negative mld_err on failure; the text of
 * the last error on the calling thread is mld_last_error().  A handle is not thread-safe; distinct
 * handles are.  There is no CPU fallback: without a HIP device every compute entry point fails
 * with MLD_ERR_NO_DEVICE.
 *
 * The reference (michchr/pyhybridcontrol) has no FFI: its seam for this path is the Python call
 * `ConstraintSolvedController.solve()` -> `cvx.Problem.solve()` (controllers/controller_base.py:491-540,
 * :509).  Each entry point below names the reference code it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 */
#ifndef MLDGPU_H
#define MLDGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mld_model mld_model_t;
typedef struct mld_problem mld_problem_t;

typedef enum {
    MLD_OK = 0,
    MLD_ERR_INVALID = -1,     /* bad argument / shape */
    MLD_ERR_NO_DEVICE = -2,   /* no HIP device: the product path has no CPU fallback */
    MLD_ERR_UNSUPPORTED = -3, /* e.g. a problem too large for the LDS staging */
    MLD_ERR_HIP = -4,         /* HIP runtime error (text in mld_last_error) */
    MLD_ERR_COMM = -5         /* RCCL error */
} mld_err;

/* per-instance solve status (status_out of mld_solve_batch) */
typedef enum {
    MLD_STATUS_OPTIMAL = 0,
    MLD_STATUS_INFEASIBLE = 1,
    MLD_STATUS_NODE_LIMIT = 2, /* incumbent (if any) returned, gap = obj - lower_bound */
    MLD_STATUS_NUMERICAL = 3,
    MLD_STATUS_UNBOUNDED = 4   /* a free variable rests on the solver's artificial box: objective -inf, the ray's point in v_out
                                  (counted under n_numerical in mld_stats) */
} mld_status;

/* MLD dimensions, models/mld_model.py:149-168 (MldInfo._mld_dim_map): nv = nu+ndelta+nz+nmu;
 * binaries are the trailing nu_l / nmu_l entries of u / mu, all of delta, none of z (:294-345). */
typedef struct {
    int32_t nx, nu, ndelta, nz, nmu, nomega, ny, nc, nu_l, nmu_l;
} mld_dims;

/* flags */
#define MLD_CONDENSE_DEFAULT 0
/* mld_opts.flags: the dense GEMMs of the path -- K3 constraint right-hand sides ([H_x | H_w] [x0 ; w], controllers/
 * controller_base.py:446-450) and K4 quadratic cost pull-back (Gamma' W Gamma, controllers/components/objective_atoms.py:
 * 321-331) -- run on the matrix cores in fp32 (v_mfma_f32_16x16x4_f32: inputs rounded to fp32, fp32 accumulate; relative
 * error ~1e-6 of the largest term) instead of fp64 (v_mfma_f64_16x16x4_f64).  Arrays at the boundary stay `double`; the
 * branch-and-cut itself always runs in fp64 and verifies every returned point against the original fp64 rows. */
#define MLD_F32 1

/* Solver options.  max_nodes / gap_rel / time-like limits mirror the Gurobi parameters the reference
 * forwards through **solver_kwargs (NodeLimit, MIPGap; micro_grid_control_simulation.py:232). */
typedef struct {
    double gap_abs;        /* absolute optimality tolerance (default 1e-9) */
    double gap_rel;        /* relative MIP gap (Gurobi MIPGap; default 0 = prove optimality) */
    int32_t max_nodes;     /* per instance (default 100000): nodes of the branch-and-bound tree(s) a search evaluates -- the look-ahead LPs of the dive, the
                              leaf evaluations and a MIP start are not nodes (round 4; rounds 1-3 counted every LP, which is not Gurobi's NodeLimit) */
    int32_t max_pivots;    /* per instance simplex iteration limit (default 50000) */
    int32_t cut_rounds;    /* cut rounds at the root (default -1 = max(10, min(30, binaries / 40)); 0 = no cuts) */
    int32_t cuts_per_round;/* Gomory cuts per round (default -1 = max(80, binaries / 5)) */
    int32_t max_cuts;      /* rows reserved for cuts (default -1 = max(300, rows / 4) up to 400 binaries, rows / 2 above; 0 = no cuts) */
    int32_t presolve;      /* default 6.  bit1: per-model probing-based big-M tightening (once, when the problem is created).  bit2 (round 4): per-INSTANCE
                              presolve before the root LP -- row-activity bound propagation with integer rounding on the instance's own right-hand side
                              (its x0 / omega); binaries it fixes are fixed, the implied bounds of the continuous variables are used by the rounding cuts
                              and by the test for rows that can never bind, the LP keeps the model's bounds; an instance whose rows are infeasible under
                              the bounds returns MLD_STATUS_INFEASIBLE without a pivot.  What a MIP backend's own presolve does for the
                              reference (controller_base.py:497-512 hands the instance to the solver as is).  bit0 reserved */
    int32_t n_slots;       /* solver slots = persistent workgroups (0 = auto: what is resident at once, one per CU) */
    int32_t mir_per_round; /* complemented mixed-integer rounding cuts on the original rows per cut round
                              (default -1 = 20 up to 400 binaries, binaries / 5 above; 0 = off) */
    int32_t flags;         /* MLD_F32 (default 0, see below) */
    int32_t reserved;      /* diagnostics, default 0.  bit0 solver trace (builds with -DMLD_TRACE only), bit1 refactor at every
                              verification, bit2 never refactor, bit3 no longest-first work queue, bit4 keep maintaining the rows
                              that cannot bind under the root bounds, bit5 Gomory cuts one at a time (A/B of the wave-parallel
                              round).  Results are the same up to rounding with every bit; only speed and traces change.
                              bit6 first-fractional branching instead of penalty branching, bit7 K3 / K4 on the vector ALUs (k_rhs, k_gemm)
                              instead of the matrix cores, bit8 relaxation-only batches (every binary fixed) on the dense-dictionary kernel instead of
                              the LDS-resident revised simplex (k_lp_lds), bit9 k_lp_lds with a working-basis capacity of 24 (its overflow fall-back to the
                              dense kernel then takes most instances), bit11 leave those instances at status -1 instead (counting only),
                              (bit 10 belonged to the LDS-resident branch-and-cut experiment of rounds 2-3, removed in round 4: DESIGN section 4c), bit12 no per-instance presolve
                              (A/B of presolve bit2 on the same problem handle),
                              bit13 no anti-stalling cost
                              perturbation in the dual simplex (A/B of round 3's change), bit14 no long-step (bound flipping) ratio test in the root LP (A/B),
                              bit15 rounding cuts built one at a time by the whole workgroup instead of a wave per cut (A/B of round 4's change),
                              bit16 no root restart (more cut rounds at the root of a cold instance once an incumbent leaves a gap of at most three tolerances; A/B of round 4's change),
                              bit17 a MIP start is evaluated lazily (round 3: only when the deepening passes end without an incumbent) instead of before the root LP (A/B). */
    double time_limit;     /* seconds per INSTANCE on the device clock (Gurobi TimeLimit; the reference passes TimeLimit=20 with every solve,
                              examples/residential_mg_with_pv_and_dewhs/micro_grid_control_simulation.py:232, forwarded by
                              controllers/controller_base.py:509-512): the branch-and-bound of an instance ends once it has run that long, like
                              max_nodes ends it -- status MLD_STATUS_NODE_LIMIT with the incumbent and the proven bound (an instance without an
                              incumbent still gets its one rescue dive).  The clock is read between nodes, between cut rounds and, once the search has started,
                              every 128 pivots inside an LP (the root LP always runs to its end -- without it there is no answer --, and so does the rescue
                              dive), so ONE instance ends within its limit plus its root LP and a few pivots; the limit is per instance, not per call: a batch larger than
                              the number of resident workgroups takes (instances / workgroups) x time_limit at worst.  0 (default) = no limit. */
} mld_opts;

/* Linear cost in tiled horizon form (the Python layer parses the reference's string-keyed atoms,
 * controllers/components/objective_atoms.py:453-521, and tiles per-step weights :118-137).
 * Each pointer is n_models x len (or NULL = zeros):
 *   lin_v  : N_tilde*nv   weight on v_tilde  (q_u, q_delta, q_z, q_mu, q_v atoms scattered into v order)
 *   lin_x  : N_tilde*nx   weight on x_tilde  (pulled back through Gamma_v: variables.py:259-265)
 *   lin_y  : N_tilde*ny   weight on y_tilde  (pulled back through L_v:     variables.py:269-275)
 * Quadratic weights (symmetric, full horizon size, NULL = none): quad_v (n x n), quad_x, quad_y --
 * cost  var' W var  (objective_atoms.py:321-331); they are assembled into P / Qx / Qw by kernel K4 and the
 * solver then runs a convex-QP relaxation (simplicial decomposition) at every branch-and-bound node. */
typedef struct {
    const double *lin_v, *lin_x, *lin_y;
    const double *quad_v, *quad_x, *quad_y;
} mld_cost;

typedef struct {
    int64_t nodes, pivots, cuts, refactors; /* totals over the batch */
    int32_t n_optimal, n_infeasible, n_node_limit, n_numerical;
    double solve_ms;                         /* device time of the solve kernel(s), HIP events */
    double rhs_ms;
} mld_stats;

/* ---- discovery / errors ------------------------------------------------------------------ */
int mld_device_count(void);
int mld_set_device(int device);
const char *mld_last_error(void);
const char *mld_version(void);
int mld_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes, int *lds_bytes);

/* ---- model --------------------------------------------------------------------------------
 * n_models same-shaped MLD systems  x+ = A x + B1 u + B2 d + B3 z + B4 w + b5 ; y = C x + D1 u + ... ;
 * E x + F1 u + F2 d + F3 z + F4 w + G y + Psi mu <= f5  (models/mld_model.py:456-463).
 * `mats` holds 20 pointers in the order A,B1,B2,B3,B4,b5,C,D1,D2,D3,D4,d5,E,F1,F2,F3,F4,f5,G,Psi, each
 * n_models x rows x cols row-major, NULL = zeros (the reference pads missing matrices with zeros,
 * mld_model.py:910-928; C must be passed explicitly -- the Python layer applies the C=I default :515-520). */
int mld_model_create(mld_model_t **out, const mld_dims *dims, int n_models, const double *const *mats);
/* Time-varying horizon (MldInfo / mld_numeric_tilde, models/mld_model.py:1210-1227; the time-varying branch of
 * controllers/components/mld_evolution_matrices.py:265-272): every horizon is N_tilde step models, step k acting on
 * (x(k), v(k)).  `mats` as above with n_horizons x N_tilde stacked models (horizon-major).  The handle then stands
 * for n_horizons condensed systems: mld_condense* accept only this N_tilde, and problems built on it
 * (mld_problem_create with the same N_tilde) index horizons through model_idx exactly as they index models. */
int mld_model_create_tv(mld_model_t **out, const mld_dims *dims, int n_horizons, int N_tilde, const double *const *mats);
int mld_model_destroy(mld_model_t *);

/* ---- condensing (kernels K1+K2) -----------------------------------------------------------
 * Replaces MldEvoMatrices.gen_mld_evo_matrices (controllers/components/mld_evolution_matrices.py:107-244)
 * and block_toeplitz / block_diag_dense (utils/matrix_utils.py:55-81,117-161).  Outputs (any may be
 * NULL) are n_models x the exact materialised layout of the reference's *_N_tilde matrices:
 *   Phi_x (N nx, nx)  Gamma_v (N nx, N nv)  Gamma_w (N nx, N nw)  Gamma_5 (N nx, 1)
 *   L_x   (N ny, nx)  L_v     (N ny, N nv)  L_w     (N ny, N nw)  L_5     (N ny, 1)
 *   H_x   (N nc, nx)  H_v     (N nc, N nv)  H_w     (N nc, N nw)  H_5     (N nc, 1)
 * mld_condense_device computes into device-resident buffers owned by the model (no host traffic)
 * and returns the device time of the condensing kernels; mld_condense = that + copies to host. */
int mld_condense_device(mld_model_t *, int N_tilde, int flags, double *kernel_ms);
int mld_condense(mld_model_t *, int N_tilde, int flags, double *Phi_x, double *Gamma_v, double *Gamma_w,
                 double *Gamma_5, double *L_x, double *L_v, double *L_w, double *L_5, double *H_x, double *H_v,
                 double *H_w, double *H_5);

/* fp32 materialisation (the MLD_F32 side of the condensing, SURVEY 8b / BASELINE configs[4] "fp32 condensing"): the same block
 * arithmetic in fp64 (the block products have inner dimension nx <= 15: nothing for the matrix cores), every output element
 * rounded once to fp32 on its way to HBM -- each value is the correctly rounded fp64 value, relative error <= 6e-8 -- so the
 * write stream that bounds the kernel is half as long.  Time-invariant models only. */
int mld_condense_device_f32(mld_model_t *, int N_tilde, int flags, double *kernel_ms);
int mld_condense_f32(mld_model_t *, int N_tilde, int flags, float *Phi_x, float *Gamma_v, float *Gamma_w, float *Gamma_5, float *L_x,
                     float *L_v, float *L_w, float *L_5, float *H_x, float *H_v, float *H_w, float *H_5);

/* ---- problem ------------------------------------------------------------------------------
 * Replaces MpcController.build (controllers/mpc_controller.py:76-101): condensed constraint maps
 * (on device, from the big-M-tightened model), cost pull-back (kernel K4), scaling, bounds
 * (mu >= 0, binaries in {0,1}: controllers/components/variables.py:189-243). */
/* sizeof(mld_opts) of the library build: bindings check it against their own layout before the first call that takes the struct (the struct has
 * grown: time_limit in round 3); mld_version() is bumped with every layout change. */
int mld_opts_size(void);
int mld_opts_default(mld_opts *);
int mld_problem_create(mld_problem_t **out, mld_model_t *model, int N_p, int N_tilde, const mld_cost *cost,
                       const mld_opts *opts);
int mld_problem_set_cost(mld_problem_t *, const mld_cost *cost); /* prices change every MPC step */
int mld_problem_destroy(mld_problem_t *);

/* Cost assembly (kernel K4) read-back, n_models x ...: P (n,n) q0 (n) Qx (n,nx) Qw (n,N nw); any NULL.
 * objective = 1/2 v'Pv + (q0 + Qx x_k + Qw w)'v + r(x_k,w)   (objective_atoms.py:308-331,523-532) */
int mld_cost_assemble(mld_problem_t *, double *P, double *q0, double *Qx, double *Qw);

/* ---- solve (kernels K3, K5, K6) -------------------------------------------------------------
 * Replaces the backend call `self._problem.solve(...)` (controllers/controller_base.py:509) for
 * `batch` independent instances:  model_idx[b] in [0,n_models) (NULL = all 0), x0 (batch, nx),
 * omega (batch, N_tilde*nomega) step-major, fixed_bin (batch, n_bin) with 0/1 = fixed, 255 = free
 * (NULL = all free; all fixed = relaxation-only mode).  Outputs: v_out (batch, N_tilde*nv) in the
 * reference's v_tilde order [u0;d0;z0;mu0;u1;...] (variables.py:226-241), obj_out (batch) including
 * the constant term, status_out (batch), lower_bound_out (batch, may be NULL), stats (may be NULL). */
int mld_solve_batch(mld_problem_t *, int batch, const int32_t *model_idx, const double *x0, const double *omega,
                    const uint8_t *fixed_bin, double *v_out, double *obj_out, int32_t *status_out,
                    double *lower_bound_out, mld_stats *stats_out);

/* The same in three steps so that a benchmark can time the device work with inputs resident in HBM. */
int mld_upload_batch(mld_problem_t *, int batch, const int32_t *model_idx, const double *x0, const double *omega,
                     const uint8_t *fixed_bin);

/* Extra constraint blocks of the uploaded batch.  Replaces `set_constraints(other_constraints=[gen_evo_constraints(
 * omega_scenarios_k=..., N_tilde=...), ...])` (controllers/controller_base.py:457-475 with :411-456; callers
 * examples/.../micro_grid_control_simulation.py:200-227: scenario-based and min-max controllers).  Every block
 * has the standard block's left-hand side H_v (a row prefix when its N_tilde is reduced), so the stacked system is the
 * standard one with the row-wise minimum right-hand side.  omega_cols (batch, n_cols, N_tilde*nomega): the
 * disturbance columns of all blocks (scenario columns, min / max profiles; pad reduced-horizon columns with zeros);
 * col_rows (n_cols) = number of leading constraint rows column c applies to (NULL = all rows).  Valid until the
 * next mld_upload_batch; n_cols = 0 clears. */
int mld_upload_constraint_blocks(mld_problem_t *, int n_cols, const double *omega_cols, const int32_t *col_rows);
/* The same with the state every column was generated with: the reference's gen_evo_constraints accepts an explicit x_k instead of the
 * controller's parameter (controllers/controller_base.py:411-416: `x_k = self._x_k if x_k is None else x_k`); the right-hand side of such a
 * block is H_x x_cols + H_omega omega_col + H_5.  x_cols: batch x n_cols x nx, NULL = every column uses the instance's x0. */
int mld_upload_constraint_blocks_x(mld_problem_t *, int n_cols, const double *omega_cols, const int32_t *col_rows, const double *x_cols);
int mld_solve_resident(mld_problem_t *, mld_stats *stats_out);

/* The resident solve in two halves, for callers that keep several problems busy (a fleet of independent microgrids: the reference solves
 * them one after another, controllers/controller_base.py:509).  mld_problem_use_stream gives the problem its own HIP stream;
 * mld_solve_launch queues K3 + K5/K6 on it and returns; mld_solve_finish waits, reports the statistics and learns the work-queue order.
 * While one problem's stragglers finish, the workgroups of the next problem's launch move onto the freed CUs.  Results are those of
 * mld_solve_resident, bit for bit.  (Batches on the LDS-resident kernels complete inside mld_solve_launch.) */
int mld_problem_use_stream(mld_problem_t *p);
int mld_solve_launch(mld_problem_t *p);
int mld_solve_finish(mld_problem_t *p, mld_stats *stats_out);
int mld_download_results(mld_problem_t *, double *v_out, double *obj_out, int32_t *status_out,
                         double *lower_bound_out, int32_t *nodes_out, int32_t *pivots_out);

/* Limits and tolerances of an existing problem (gap_abs, gap_rel, max_nodes, max_pivots, cut_rounds, cuts_per_round,
 * mir_per_round, reserved; negative cut fields keep the current value).  The reference passes MIPGap / NodeLimit / TimeLimit as
 * per-call **solver_kwargs of solve() (controllers/controller_base.py:491-512): they must not need a rebuild.  max_cuts,
 * n_slots and presolve shape the workspace / the tightened model and are ignored here. */
int mld_problem_set_opts(mld_problem_t *, const mld_opts *opts);
/* the options in effect, every size-scaled default (-1 fields of mld_opts_default) resolved */
int mld_problem_get_opts(mld_problem_t *, mld_opts *opts_out);

/* Receding horizon on device -- the plant update the reference performs after every solve, ControllerBase.sim_step_k ->
 * MldModel.lsim_k (controllers/controller_base.py:229-253, models/mld_model.py:647-699): for every instance of the uploaded
 * batch  x0 <- A x0 + B1 u + B2 delta + B3 z + B4 omega_0 + b5  with (u, delta, z) the step-0 slice of the last solution, and the
 * disturbance forecast moved on by one step (the first step re-enters at the end of the horizon).  The next
 * mld_solve_resident then solves the NEXT MPC step without any host traffic.  mld_download_inputs reads the current inputs. */
int mld_advance_batch(mld_problem_t *);
/* The same with the count of instances that were NOT advanced (n_skipped_out may be NULL).  Equivalence with the reference's plant step:
 * sim_step_k hands only u_k to lsim_k (controller_base.py:229-239), which re-derives delta_k, z_k (and mu_k) from (x, u, omega) by a feasibility
 * problem (_compute_aux, mld_model.py:683-686, 701-766); this entry applies the PLANNED (delta_0, z_0) instead.  The two agree when the MLD
 * model is well posed (x, u, omega determine delta, z), the plan uses no soft-constraint slack in step 0 (or delta, z do not drive the state:
 * B2 = B3 = 0, as in the example's models) and the simulated model is the controller's model.  Instances that have no usable plan -- status
 * other than OPTIMAL / NODE_LIMIT, or no incumbent -- keep their state and forecast and are counted; time-varying models
 * (mld_model_create_tv) and a batch that has not been solved since its upload are refused. */
int mld_advance_batch2(mld_problem_t *, int32_t *n_skipped_out);

/* MIP start of the resident batch (the reference calls its backend with warm_start=True, controllers/controller_base.py:493,509-512: the
 * previous values of the variables are the solver's start).  bin_start: batch x n_bin bytes, the values (0 / 1) of the binaries in the order of
 * the variable layout (step-major, controllers/components/variables.py:189-243); an instance whose first byte is 255 has no start; NULL
 * clears the start.  The start is evaluated FIRST (binaries fixed, one LP from the slack basis, verified against the original rows; the root
 * relaxation is then solved from that leaf's basis): a feasible start becomes the incumbent, the cut loop stops as soon as the bound is within
 * the gap of it, and the search -- if one is still needed -- starts around it (RINS, then the guided depth-first search).  (Round 3 evaluated it only when the first passes ended without an incumbent:
 * opts.reserved bit 17.)  Any upload / selection of new inputs clears the start. */
int mld_set_warm_start(mld_problem_t *, const uint8_t *bin_start);
/* The start built on the device from the last solution of the resident batch: shift = 0 takes the plan as it is (what warm_start=True means
 * to the reference's backend: the variables' previous values), shift = k > 0 moves it k steps towards the present and repeats its last step
 * (receding horizon: call after mld_advance_batch).  Instances without an incumbent get no start. */
int mld_warm_start_from_previous(mld_problem_t *, int shift);

/* ---- sub-tree hand-off: open nodes of a stopped search become instances of the next batch -------------------------------------
 * The reference's backend runs one branch-and-bound per solve() call to its gap (controllers/controller_base.py:509; TimeLimit 20 s in
 * the example); here a batch gives every instance ONE workgroup, so a few instances with large trees hold a launch while the other CUs
 * idle.  With recording enabled, an instance that stops at a limit inside a complete search (plain depth-first search below its
 * incumbent) returns its stack: depth, and per level the variable (index into the decision vector), its current value and a flag
 * "sibling accounted for".  Everything still open is then: for each level k with flag 0 the node {levels < k at their values, level k
 * flipped}, plus the node {all levels at their values}.  The caller uploads those nodes as instances of a new batch (same x0 / omega,
 * the node's fixings in fixed_bin, everything else 255) with the parent's incumbent value as cutoff: every CU works on the tail.
 * mld_set_cutoffs: per instance of the resident batch the objective (constant term included) that must be beaten; +inf = none, NULL
 * clears.  Under a cutoff a search that finds nothing better ends INFEASIBLE with objective +inf ("nothing better exists in this node").
 * pyhybridcontrol_amd.gpu.GpuProblem.solve_handoff drives the rounds and merges the results. */
int mld_set_cutoffs(mld_problem_t *, const double *cutoff);
int mld_record_open_nodes(mld_problem_t *, int enable);
int mld_download_open_nodes(mld_problem_t *, int32_t *depth_out, int16_t *var_out, uint8_t *val_out, uint8_t *flag_out);

/* The same hand-off INSIDE one launch (round 4): what lets ONE solve() call of the reference (controllers/controller_base.py:491-540: one instance
 * per call, its backend free to use every core on that one tree) use more than one compute unit here, and a batch keep the device busy while its
 * largest trees finish.  With it enabled the solve kernel's work queue has room for ITEMS behind the instances: an instance (or item) whose complete
 * depth-first search stops at its node limit publishes the open nodes of its stack as items -- the instance's inputs, the node's fixings, the
 * search's incumbent as cutoff -- and whichever workgroup runs out of work solves them as instances of their own (max_nodes for an instance,
 * sub_nodes for an item; a search is split at most max_gen generations deep and only while it has at most max_children open nodes; a tree that
 * cannot be split further stays MLD_STATUS_NODE_LIMIT with its incumbent and bound).  The results of a tree are merged on the device -- smallest
 * objective, on ties the node that comes first in the tree -- so mld_download_results returns per instance what an unlimited search of that
 * instance would have returned; the set of items and every item's arithmetic do not depend on which workgroup ran what, so results are
 * reproducible.  room_factor: items the queue has room for, as a multiple of the batch (at least 4096; <= 0 keeps the current value, default 2).
 * Takes effect with the next mld_upload_batch.  Not with a quadratic cost and not on the LDS-resident LP path (those solves run as before).
 * A tree for which more than max_tree items (default 160) of ONE generation have been published keeps growing and is GIVEN UP: its remaining items are skipped and the instance
 * keeps what its own search returned (MLD_STATUS_NODE_LIMIT, incumbent, bound) -- whether that happens does not depend on the queue order.
 * mld_handoff_stats: out[0] items published by the last solve, out[1] trees given up for their size, out[2] instances that were split and are
 * still unfinished (the given-up ones included), out[3] trees given up because the queue was full (raise room_factor: the one order-dependent case). */
/* The reference's build(with_std_constraints=False) and set_constraints(std_evo_constaints=[...]) (controllers/mpc_controller.py:76-101,
 * controllers/controller_base.py:457-475): with enable = 0 the standard block -- the rows whose right-hand side comes from the batch's own
 * (x0, omega) -- is not part of the problem; only the blocks of mld_upload_constraint_blocks constrain, and a row none of them covers does not
 * exist for that solve.  Default 1. */
int mld_set_std_block(mld_problem_t *, int enable);
int mld_set_handoff(mld_problem_t *, int enable, int sub_nodes, int max_gen, int max_children, int max_tree, double room_factor);
int mld_handoff_stats(mld_problem_t *, int64_t out[4]);
/* How a search that reaches its node limit splits (default 0, 0 = stop and publish everything it leaves open): with donate > 0 it hands off only its
 * `donate` SHALLOWEST open nodes -- the largest open subtrees -- and goes on below them with another sub_nodes nodes, up to `rounds` times, before it
 * stops for good: the deep open nodes, which the warm dictionary closes in a few pivots each, stay where they are cheap. */
int mld_set_handoff_policy(mld_problem_t *, int donate, int rounds);
/* Scenario streaming with everything resident in HBM: the parameter update at the top of the reference's solve() (x_k and
 * omega_tilde set as cvx.Parameter values, controllers/controller_base.py:495-498; the example re-solves with new forecasts every
 * step, micro_grid_control_simulation.py:229-232) for a whole batch.  mld_stage_inputs uploads n_sets input sets of the uploaded
 * batch's size (x0_sets: n_sets x batch x nx, omega_sets: n_sets x batch x N_tilde*nomega; model_idx and fixed_bin stay those of
 * mld_upload_batch; n_sets = 0 frees them); mld_select_inputs makes set k the batch's current inputs by a device-to-device copy. */
int mld_stage_inputs(mld_problem_t *, int n_sets, const double *x0_sets, const double *omega_sets);
int mld_select_inputs(mld_problem_t *, int set);
int mld_download_inputs(mld_problem_t *, double *x0, double *omega);

/* Per-instance telemetry of the last solve: time spent inside the solve kernel (device wall clock, ns) and
 * the number of dictionary rows the rank-1 updates touched (x *row_bytes x 2 = bytes streamed by pivots). */
int mld_download_telemetry(mld_problem_t *, int64_t *latency_ns, int64_t *rows_updated, int64_t *row_bytes);

/* Constraint right-hand side only (kernel K3): h = H_x x_k + H_w w + H_5 per instance
 * (controllers/controller_base.py:446-450); `scenarios`>1 applies the row-min over scenario columns
 * of H_w Omega (:442-444) with omega laid out (batch, scenarios, N_tilde*nomega).  h_out (batch, N nc).
 * Uses the ORIGINAL (un-tightened) model so that it can populate gen_evo_constraints. */
int mld_rhs_batch(mld_problem_t *, int batch, int scenarios, const int32_t *model_idx, const double *x0,
                  const double *omega, double *h_out);

/* ---- multi-GPU result gather (RCCL over xGMI) ---------------------------------------------
 * One process per GPU.  Rank 0 calls mld_comm_unique_id, the bytes are broadcast by the launcher
 * (any side channel), every rank calls mld_comm_init.  mld_gather all-gathers `count` doubles per
 * rank from a HOST buffer (staged through device memory) into recv (n_ranks*count). */
#define MLD_COMM_ID_BYTES 128
int mld_comm_unique_id(uint8_t id[MLD_COMM_ID_BYTES]);
int mld_comm_init(int n_ranks, int rank, const uint8_t id[MLD_COMM_ID_BYTES]);
int mld_gather(const double *send, int count, double *recv);
/* The result gather of the path straight from the device buffers of the last solve (no host staging of the send side): per
 * instance 2 + nv doubles = (objective, status, step-0 slice [u0; delta0; z0; mu0] -- what `variables_k` hands the caller,
 * controllers/components/variables.py:75-85); every rank contributes its whole batch (equal on all ranks); recv (host) holds
 * n_ranks x batch x (2 + nv), rank-major. */
int mld_gather_results(mld_problem_t *, double *recv, int *width_out);
int mld_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* MLDGPU_H */
