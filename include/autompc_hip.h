/*
 * autompc_hip.h -- C ABI of libautompc_hip.so, the MI355X (gfx950) MPC inner-solve library.
 *
 * The reference (williamedwards/autompc) is pure Python and has no FFI; its hot path is reached
 * through two Python ABCs.  Each entry point below states which reference function(s) it
 * replaces (file:line relative to the reference tree).  The Python classes in autompc_amd/
 * bind these with ctypes and reproduce the reference's Controller / Model plugin surface.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; ampc_last_error() returns the message
 *     of the calling thread's last failure.  Nothing throws across this boundary.
 *   - all host arrays are caller-owned, contiguous, row-major, IEEE float64 ("double").  The
 *     library converts to its compute precision (AMPC_F64 or AMPC_F32) on the device.
 *   - a handle is bound to one HIP device and one stream; calls on one handle must be
 *     serialised by the caller (the reference is single-threaded).  Different handles may be
 *     used from different threads / processes (one per GPU).
 *   - "_dev" variants take DEVICE pointers (in the handle's compute precision) and only enqueue
 *     work on the handle's stream: no host synchronisation, no PCIe traffic.
 */
#ifndef AUTOMPC_HIP_H
#define AUTOMPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ampc_handle ampc_handle;
typedef struct ampc_mppi_plan ampc_mppi_plan;
typedef struct ampc_ilqr_plan ampc_ilqr_plan;

enum { AMPC_F64 = 0, AMPC_F32 = 1 };
enum { AMPC_ACT_RELU = 0, AMPC_ACT_TANH = 1, AMPC_ACT_SIGMOID = 2, AMPC_ACT_SELU = 3 };
/* MPPI terminal-cost mode: 0 = the reference's behaviour (terminal cost of the LAST particle
 * added to every particle, mppi.py:79-82,146-148); 1 = per-particle terminal cost. */
enum { AMPC_TERM_REFERENCE = 0, AMPC_TERM_PER_PARTICLE = 1 };

const char* ampc_last_error(void);
int ampc_version(void);   /* 100 * major + minor; 104: + ampc_mppi_run_legacy; 105: + ampc_set_affine_quad_costs;
                           * 106: + ampc_ilqr_solve_queue_var, ampc_ilqr_closed_loop_var, ampc_set_indicator_costs,
                           *      ampc_mppi_plan_set_models, ampc_ilqr_plan_set_models; 107: + ampc_set_mlp_dev, ampc_ilqr_plan_set_constants */
int ampc_device_count(void);

/* ---- handle ------------------------------------------------------------------------------ */
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL to let the library
 * create its own. */
int ampc_create(int device, int precision, void* stream, ampc_handle** out);
int ampc_destroy(ampc_handle* h);
int ampc_synchronize(ampc_handle* h);
int ampc_precision(const ampc_handle* h);

/* ---- model: MLP surrogate dynamics --------------------------------------------------------
 * Replaces the state held by autompc.sysid.MLP (mlp.py:137-165 net, :308-321 parameters).
 * weights[l] is torch.nn.Linear layout [out_l][in_l]; l = 0..n_hidden (last = output layer).
 * nx <= 64, hidden widths 16..256 (MPPI, prediction and Jacobians: tests/test_gpu_mppi.py
 * test_mppi_wide_states_with_wide_networks_vs_oracle runs 40 / 48 / 64 states against 192..256-unit layers);
 * iLQR plans on MLP models: nx + nu <= 63, the four- / twelve-row line search for nx <= 32 (else the sixteen-row kernel).
 * x' = x + dy_mean + dy_std * net(([x,u] - xu_mean) / xu_std)          (mlp.py:219-236)
 * activation: 0 relu, 1 tanh, 2 sigmoid, 3 selu (mlp.py:44-51); 4 identity (ampc_set_linear). */
int ampc_set_mlp(ampc_handle* h, int nx, int nu, int n_hidden, const int* hidden_sizes,
                 int activation, const double* const* weights, const double* const* biases,
                 const double* xu_mean, const double* xu_std, const double* dy_mean,
                 const double* dy_std);

/* The same model from DEVICE memory: every pointer (the two pointer arrays themselves are host arrays) is a
 * float64 array in the memory of the handle's device, laid out as above -- what a PyTorch-ROCm fit of the
 * network leaves behind (mlp.py:177-217: the reference trains on its torch device and predicts there,
 * mlp.py:229-236; its weights never visit numpy).  The normalisers are folded and the MFMA fragment
 * packings written by two kernels on the handle's stream (csrc/api_model.cpp), bit for bit what
 * ampc_set_mlp produces from the same numbers; no host copy of the parameters is made.  The arrays are
 * only read and may be released when the call returns; work that produces them must be complete
 * (synchronise the producing stream) before the call. */
int ampc_set_mlp_dev(ampc_handle* h, int nx, int nu, int n_hidden, const int* hidden_sizes,
                     int activation, const double* const* weights_dev, const double* const* biases_dev,
                     const double* xu_mean_dev, const double* xu_std_dev, const double* dy_mean_dev,
                     const double* dy_std_dev);

/* ---- kernels specialised for the staged model's shape -------------------------------------------
 * The reference's MLP configuration space is 1-4 hidden layers of 16-256 units (mlp.py:113-122).
 * Kernels specialised for a model shape (all dimensions, strides and LDS offsets immediates; 1.1-1.9x
 * faster than the run-time-shape kernels) are built in for the benchmark systems' default networks
 * (csrc/shapes.hpp).  For any other MLP shape the library compiles them itself: once a handle has
 * both a model and a cost (the observation dimension is part of the shape), an out-of-process hipcc
 * build of a "shape plugin" starts in the background -- cached on disk in $AMPC_JIT_CACHE (default
 * <package>/jit_cache; ~/.cache/autompc_amd when the package directory is read-only), keyed by precision, shape and a hash of the kernel sources -- and plans
 * created after it has finished use it; until then the run-time-shape kernels run.  Results are
 * bit-identical either way.  AMPC_JIT=0 in the environment disables it.
 *   ampc_jit_status  0 nothing to do (registered shape, SINDy / wide linear model, JIT disabled), 1 building,
 *                    2 ready, -1 failed; msg (optional) receives the plugin path or the build log's path
 *   ampc_jit_wait    block until the handle's shape is ready (returns 0) or failed (< 0)            */
int ampc_jit_status(ampc_handle* h, char* msg, int msg_len);
/* build = 0: this handle never STARTS (nor waits for) a run-time build -- for models that live for one evaluation
 * (a tuner fits a model per configuration, pipeline.py:138-145: a 5 s build for 40 ms of use); a plugin that is
 * already loaded or cached is still used.  Default 1.  Call before the model is staged. */
int ampc_handle_set_jit(ampc_handle* h, int build);
int ampc_jit_wait(ampc_handle* h);
/* Which kernels a plan launches (pass one plan, NULL for the other): 0 run-time-shape, 1 a shape
 * registered at build time, 2 a shape plugin compiled at run time, 3 the run-time-shape four-row rollout kernel
 * (small MPPI problems: f64, hidden width <= 64 -- <= 128 with at most two hidden layers --, at most
 * 32 states, and so few samples that sixteen-row tiles would leave most compute units idle). */
int ampc_plan_kernel_kind(const ampc_mppi_plan* mppi, const ampc_ilqr_plan* ilqr);

/* ---- model: linear dynamics (alternative to ampc_set_mlp) ----------------------------------
 * x' = A x + B u with A [nx][nx], B [nx][nu] row-major: the prediction of autompc.sysid.ARX
 * (arx.py:151-164, state = stacked observation/control history + constant 1) and
 * autompc.sysid.Koopman (koopman.py:170-184, state = lifted observation).  The model is staged
 * as a one-hidden-layer identity-activation network (activation code 4), so the MLP entry
 * points below (ampc_mlp_pred_batch / _pred_diff_batch, whose Jacobians are then A and B) and
 * every solver serve it unchanged.  Costs see the first obs_dim state entries
 * (mppi.py:73-82, ilqr.py:124-128).
 * Up to 64 states: above 32 (long-history ARX, large Koopman lifts; e.g. ARX with history 2 on
 * HalfCheetah: 41) the MFMA tile carries three or four output column tiles; iLQR plans need
 * nx + nu <= 63 (the augmented Quu system lives in one wave).
 * 65 .. 256 states (ARX history 3..10 on HalfCheetah: 66..235, arx.py:27,37-45): a dedicated
 * K-tiled MFMA kernel family (csrc/linear_kernels.hpp) serves ampc_mlp_pred_batch /
 * _pred_diff_batch, MPPI plans and the closed loop; iLQR plans are refused there. */
int ampc_set_linear(ampc_handle* h, int nx, int nu, const double* A, const double* B);

/* Model.pred_batch (model.py:109-130, mlp.py:229-236): out[n][nx]. */
int ampc_mlp_pred_batch(ampc_handle* h, const double* states, const double* ctrls, double* out,
                        int n);
/* Model.pred_diff_batch (model.py:155-184, mlp.py:281-305): out[n][nx], jx[n][nx][nx],
 * ju[n][nx][nu].  Analytic Jacobian chain instead of the reference's autograd. */
int ampc_mlp_pred_diff_batch(ampc_handle* h, const double* states, const double* ctrls,
                             double* out, double* jx, double* ju, int n);

/* ---- model: SINDy feature-library dynamics (alternative to ampc_set_mlp) -------------------
 * Replaces the inference half of autompc.sysid.SINDy (sindy.py:173-244; the fit stays on the
 * host).  Feature k is kind[k] applied to variables v = [x, u]:
 *   0 v_a   1 sin(p v_a)   2 cos(p v_a)   3 v_a sin(p v_b)   4 v_a cos(p v_b)   5 v_a ** p
 * with a = arg0[k], b = arg1[k], p = param[k]; xi [nx][n_feat] are the coefficients.
 *   6 monomial  prod_j v_{pair_var[j]} ** pair_exp[j]  over the arg1[k] (1..10) pairs starting at
 *     pair arg0[k] of the pair list (n_pairs, pair_var, pair_exp; NULL / 0 without monomials):
 *     the polynomial cross terms of basis_funcs.py:27-93 (sindy.py:143-145), gradient :74-84.
 *   discrete:   x' = Theta(v) xi'        continuous:  x' = x + dt Theta(v) xi'        (nx <= 64)
 * strict_reference != 0 reproduces the reference Jacobian's quirks (interaction terms counted
 * twice, polynomial gradient without the exponent factor; basis_funcs.py:24-25), as pinned by
 * tests/golden/sindy_*.npz (the reference's own pred_diff_batch outputs).  MPPI plans, iLQR
 * plans and the closed loop all work on a handle holding a SINDy model. */
int ampc_set_sindy(ampc_handle* h, int nx, int nu, int n_feat, const int* kind, const int* arg0,
                   const int* arg1, const double* param, const double* xi, int continuous,
                   double dt, int strict_reference, int n_pairs, const int* pair_var,
                   const int* pair_exp);
int ampc_sindy_pred_batch(ampc_handle* h, const double* states, const double* ctrls, double* out,
                          int n);
int ampc_sindy_pred_diff_batch(ampc_handle* h, const double* states, const double* ctrls,
                               double* out, double* jx, double* ju, int n);

/* ---- cost / bounds ------------------------------------------------------------------------
 * Replaces QuadCost (quad_cost.py:7-51) as read through Cost.eval_* (cost.py:66-213).
 * n_costs blocks (one per tuning candidate), each Q[no][no], R[nu][nu], F[no][no], goal[no]. */
int ampc_set_quad_costs(ampc_handle* h, int n_costs, int obs_dim, const double* Q,
                        const double* R, const double* F, const double* goal);
/* A sum of quadratic terms (SumCost._sum_results, sum_cost.py:49-54, over quad_cost.py:7-51) whose
 * goals differ -- what QuadCostFactory + GaussRegFactory produce (gauss_reg_factory.py:37-45: goal =
 * mean of the data; sum_cost_factory.py) and what the reference's MPPI / iLQR evaluate term by term
 * through Cost.eval_* (mppi.py:73-82, ilqr.py:124-129,159-174).  About the first term's goal g the
 * sum is an affine-quadratic form, per block:
 *   stage     (x-g)'Q(x-g) + lin'(x-g) + consts[0] + u'Ru          Q = sum Q_k, R = sum R_k,
 *   terminal  (x-g)'F(x-g) + lin_term'(x-g) + consts[1]            F = sum F_k,
 *   lin = sum (Q_k+Q_k')(g-g_k),  consts[0] = sum (g-g_k)'Q_k(g-g_k)   (lin_term, consts[1]: with F_k)
 * lin [n_costs][obs_dim], lin_term [n_costs][obs_dim], consts [n_costs][2]; any of the three may be
 * NULL (zeros: ampc_set_quad_costs).  iLQR's stage gradient is (Q+Q')(x-g) + lin; its terminal
 * gradient is (F+F')x as the reference computes it term by term (cost.py:195: no goal, hence no
 * lin_term), or (F+F')(x-g) + lin_term under ampc_ilqr_plan_set_terminal_goal(1). */
int ampc_set_affine_quad_costs(ampc_handle* h, int n_costs, int obs_dim, const double* Q,
                               const double* R, const double* F, const double* goal,
                               const double* lin, const double* lin_term, const double* consts);
/* Indicator terms of an MPPI controller's cost.  The reference's MPPI charges whatever Cost the task holds,
 * term by term (mppi.py:73-82 over sum_cost.py:49-54): e.g. QuadCost + ThresholdCost, or a bare
 * BoxThresholdCost -- 1 per time step whose observation violates the term, no control or terminal part
 * (thresh_cost.py:27-38, 73-83).  The n_terms (<= 8) terms are added to the stage cost x_0 .. x_{H-1} of
 * EVERY cost block of the handle; kinds / params as ampc_score_trajectories (1 threshold: goal[no]
 * obs_range_lo obs_range_hi threshold; 2 box: lower[no] upper[no]).  Call after ampc_set_quad_costs /
 * ampc_set_affine_quad_costs (the quadratic part; all zeros for a cost without one), which fix obs_dim;
 * n_terms = 0 removes the terms.  MPPI plans and the MPPI closed loop evaluate them (every model family);
 * ampc_ilqr_plan_create refuses a handle that has them (no gradient / Hessian). */
int ampc_set_indicator_costs(ampc_handle* h, int n_terms, const int* kinds, const double* params);
/* Task.get_ctrl_bounds (task.py:257-267): lo[nu], hi[nu] (MPPI requires finite bounds,
 * mppi.py:100-102; iLQR clips only if bounded, ilqr.py:62-64). */
int ampc_set_ctrl_bounds(ampc_handle* h, const double* lo, const double* hi);

/* ---- MPPI ---------------------------------------------------------------------------------
 * A plan owns device buffers for a batch of B independent MPPI problems (B = 1 for the plain
 * Controller.run() path; B = candidates-per-GPU for the tuning evaluator).
 * Problem p has num_path[p] samples, horizon[p] steps, noise variance sigma[p], temperature
 * lmda[p] and cost block cost_index[p]  (MPPI.__init__, mppi.py:87-105). */
int ampc_mppi_plan_create(ampc_handle* h, int B, const int* num_path, const int* horizon,
                          const double* sigma, const double* lmda, const int* cost_index,
                          int term_mode, ampc_mppi_plan** out);
int ampc_mppi_plan_destroy(ampc_mppi_plan* p);
/* Fix the launch geometry instead of letting the library derive it from the batch: tile_rows
 * (0 automatic, or 4 / 16 / 32 / 64 samples per rollout workgroup; a height that does not fit LDS
 * for the staged model / horizon, or 4 for a model the four-row kernel does not cover, is an error) and horizon_cap (the LDS / partial-sum layout is sized
 * for max(horizon_cap, longest horizon in the plan)).  Summation orders inside a solve depend on
 * the geometry; with both fixed, a problem's results are bit-identical whatever else shares the
 * plan -- the candidate evaluator uses this so that a candidate's surrogate score
 * (pipeline_tuner.py:213-258) is the same for any sharding of the batch.  Resets device buffers:
 * call before ampc_mppi_upload. */
int ampc_mppi_plan_set_geometry(ampc_mppi_plan* p, int tile_rows, int horizon_cap);
/* Host -> device.  Any pointer may be NULL (left unchanged on the device).
 *   x0      [B][nx]                       model state each solve starts from
 *   act_seq [sum_p H_p*nu]                warm-start sequences, problem-major, units of umax
 *   eps     [sum_p N_p*H_p*nu]            noise exactly as numpy draws it: per problem
 *                                         [N][H][nu] (mppi.py:126 before the transpose)       */
int ampc_mppi_upload(ampc_mppi_plan* p, const double* x0, const double* act_seq,
                     const double* eps);
/* Fill the plan's noise buffer on the device: eps ~ N(0, sigma_p) from Philox4x32-10 keyed by
 * (seed, stream, noise id of the problem, element).  `stream` (< 2^56) is the caller's step
 * counter.  Statistically equivalent to, not bit-identical with, numpy's draw (mppi.py:21-24).
 * Callers that draw stream, stream + 1, ... with one seed get the next draw formed by the previous solve's
 * update launch (the same values; this call then only swaps two buffers) -- AMPC_NOISE_AHEAD=0 disables. */
int ampc_mppi_generate_eps(ampc_mppi_plan* p, uint64_t seed, uint64_t stream);
/* The reference's noise draw itself, on the device: fills the plan's noise buffer with what
 *   np.random.normal(scale=sqrt(sigma_p), size=(N_p, H_p, nu))          (mppi.py:16-24, :126)
 * returns for every problem in turn when numpy's global LEGACY generator (MT19937 + polar method
 * with its cached second value) is in the state (key[624], pos, has_gauss, cached) -- the tuple
 * np.random.get_state() returns -- and hands back the state the generator is left in, to be
 * installed with np.random.set_state().  The raw MT19937 stream, the uniform doubles and every
 * accept / reject decision are bit-identical to numpy's; so are the normals when
 * ampc_legacy_log_mode() != 0 (see below).  Synchronises. */
/* Whether ampc_mppi_legacy_normal's normals are numpy's bit for bit.  numpy's legacy_gauss calls
 * the C library's log(); at first use the library locates glibc's log tables in the loaded libm
 * and checks its operation-by-operation restatement of glibc's algorithm (glibc >= 2.28,
 * sysdeps/ieee754/dbl-64/e_log.c) against log() itself:
 *   1  log() is glibc's FMA build and is reproduced exactly      } normals bit-identical
 *   2  log() is glibc's non-FMA build and is reproduced exactly  } to numpy's
 *   0  another log(): the device uses its own (normals within 3 ulp of numpy's; generator state,
 *      uniforms and accept / reject decisions still exact). */
int ampc_legacy_log_mode(void);
/* key_out may be key itself (the generator's own memory: the state is updated in place). */
int ampc_mppi_legacy_normal(ampc_mppi_plan* p, const uint32_t* key, int pos, int has_gauss,
                            double cached, uint32_t* key_out, int* pos_out, int* has_gauss_out,
                            double* cached_out);
/* How many draws of this plan were repeated because a wait inside the draw kernel expired (its workgroups look
 * back at their predecessors' pair counts with a bounded wait; under heavy contention from other streams or
 * processes a wait can run out).  A repeated draw starts from the same generator state and gives the same values;
 * the count is diagnostic.  AMPC_POLAR_SPIN_LIMIT in the environment shortens the first attempt's bound (tests). */
int ampc_mppi_plan_legacy_redraws(const ampc_mppi_plan* p, long long* count);
/* Jump polynomials of MT19937 (autompc_amd/data/mt19937_jump.npz, computed by tools/mt_jump.py):
 * polys[n_polys][624] = bits of t^(s * jump_blocks * 624) mod phi for s = 1 .. n_polys.  With a
 * table installed (process-wide) ampc_mppi_legacy_normal generates the raw stream block-parallel
 * (one workgroup per jump_blocks blocks) instead of sequentially; results are identical.  Up to 4
 * tables with different jump_blocks may be installed side by side: a stream is generated with the
 * shortest segments whose table covers it. */
int ampc_set_mt_jump_table(const uint32_t* polys, int n_polys, int jump_blocks);
/* ids[B]: the noise id of every problem (default: its index in the plan).  The candidate
 * evaluator passes each candidate's GLOBAL index, so that the noise -- and therefore the
 * surrogate score pipeline_tuner.py:213-258 returns for it -- does not depend on how a batch of
 * candidates is sharded over GPUs or on the candidate's position in its shard.
 * Noise already drawn with ampc_mppi_generate_eps is drawn again with the new ids (same seed and stream);
 * ampc_mppi_plan_set_geometry likewise re-draws it for the rebuilt plan. */
int ampc_mppi_plan_set_noise_ids(ampc_mppi_plan* p, const uint32_t* ids);
/* One MPPI solve per problem, enqueued on the handle's stream (MPPI.do_rollouts + update,
 * mppi.py:110-152): shift warm start, rollout all samples, softmin weights, update act_seq. */
int ampc_mppi_solve(ampc_mppi_plan* p);
/* Device -> host (synchronises).  Any pointer may be NULL.
 *   act_seq [sum_p H_p*nu]   updated sequences;  u [B][nu] = act_seq[p][0] * umax
 *   costs   [sum_p N_p]      per-sample costs (mppi.py:150-152)
 *   eps_out [sum_p H_p*N_p*nu] post-clip noise, per problem [H][N][nu] (as do_rollouts returns) */
int ampc_mppi_download(ampc_mppi_plan* p, double* act_seq, double* u, double* costs,
                       double* eps_out);
/* MPPI.run (mppi.py:154-168) for the plan's B controllers in ONE call with one host synchronisation:
 *   x0 [B][nx] in;  act_seq (optional, NULL = keep the device's warm start) in;
 *   noise: 0 = the plan's noise buffer as it is (ampc_mppi_upload / ampc_mppi_legacy_normal),
 *          1 = fresh Philox noise keyed by (seed, stream), as ampc_mppi_generate_eps;
 *   solve;  u [B][nu] = act_seq[p][0] * umax out.
 * Equivalent to ampc_mppi_upload + ampc_mppi_generate_eps + ampc_mppi_solve + ampc_mppi_download(u),
 * without their intermediate synchronisations (the drop-in classes' hot call). */
int ampc_mppi_run(ampc_mppi_plan* p, const double* x0, const double* act_seq, int noise, uint64_t seed,
                  uint64_t stream, double* u);
/* ampc_mppi_run with the reference's own noise: ampc_mppi_legacy_normal (numpy's legacy draw from
 * the generator state key / pos / has_gauss / cached, mppi.py:16-24, :126) + ampc_mppi_run(noise 0) in
 * one call with ONE synchronisation; the state after the draw comes back as from
 * ampc_mppi_legacy_normal (key_out may be key).  The default mode of the drop-in MPPI.run(). */
int ampc_mppi_run_legacy(ampc_mppi_plan* p, const double* x0, const double* act_seq, const uint32_t* key,
                         int pos, int has_gauss, double cached, uint32_t* key_out, int* pos_out,
                         int* has_gauss_out, double* cached_out, double* u);
/* Overwrite x0 of every problem from a device buffer [B][nx] in compute precision (closed-loop
 * evaluator: no host round trip). */
int ampc_mppi_set_x0_dev(ampc_mppi_plan* p, const void* x0_dev);
/* Kernel-level introspection for bench.py: grid size (workgroups) and samples per workgroup of
 * the rollout launch, algorithmic FLOPs and bytes of one solve. */
int ampc_mppi_plan_info(const ampc_mppi_plan* p, int* n_workgroups, int* samples_per_wg,
                        double* flops, double* bytes);

/* keep_eps_out = 0: do not materialise the clipped noise in HBM (it is only needed by
 * ampc_mppi_download(eps_out); the solve itself keeps it in LDS).  Default 1. */
int ampc_mppi_plan_set_outputs(ampc_mppi_plan* p, int keep_eps_out);

/* Per-kernel timing for the roofline report: when enabled, every ampc_mppi_solve brackets the
 * rollout and update launches with HIP events on the handle's stream.  ampc_mppi_plan_timing
 * synchronises, returns the AVERAGE duration (ms) of each kernel over the solves since the last
 * call, and resets the counters.  enable = n > 1: only every n-th solve is bracketed (the three event records
 * of a solve cost about 11 us on the stream: 3.7 % of a config-3 solve). */
int ampc_mppi_plan_set_timing(ampc_mppi_plan* p, int enable);
int ampc_mppi_plan_timing(ampc_mppi_plan* p, double* rollout_ms, double* update_ms, int* count);

/* ---- closed loop on the surrogate (device resident) ---------------------------------------
 * simulate() (utils/simulation.py:11-64) for the B controllers of a plan, as the tuner's
 * eval_cfg drives it (pipeline_tuner.py:222-231): n_steps x { MPPI solve from the current
 * observation; obs <- surrogate.pred(obs, u) } with no host round trip per step.
 * surrogate: handle holding the simulation model (NULL = the plan's own model); it shares the
 * plan's device, precision and dimensions; its work is enqueued on the plan's stream.
 * eps_all: NULL -> device Philox noise keyed by (seed, step, noise id); else host noise for every step,
 *          [n_steps][sum_p N_p*H_p*nu], each step laid out as ampc_mppi_upload expects.
 * traj_obs [B][n_steps+1][nx], traj_ctrls [B][n_steps+1][nu] (last control row zero, as
 * simulate() returns them).  Scoring (Cost.__call__, cost.py:27-41) is left to the caller. */
int ampc_mppi_closed_loop(ampc_mppi_plan* p, ampc_handle* surrogate, const double* init_obs,
                          int n_steps, uint64_t seed, const double* eps_all, double* traj_obs,
                          double* traj_ctrls);
/* Index of the first control step of the plan's NEXT closed loop (default 0): step s of that loop
 * draws its device noise from the Philox stream (seed, first_step + s, noise id).  An episode that
 * is run in several segments -- simulate() with a user termination condition
 * (utils/simulation.py:62-63, tasks/task.py:92-101) evaluated on the host between segments, each
 * segment starting from the state the previous one ended in -- thereby consumes exactly the noise
 * the unsegmented episode would.  One-shot: the closed loop that consumes the offset resets it to 0. */
int ampc_mppi_plan_set_step_offset(ampc_mppi_plan* p, uint64_t first_step);
/* Controller models whose state is REBUILT from every observation -- autompc.sysid.Koopman:
 * update_state(state, ctrl, obs) = lift(obs) (koopman.py:166-168), state = the basis functions applied
 * to the observation, basis-major [f_0(o_0..o_{no-1}), f_1(..), ...] (koopman.py:105-122) -- in the
 * device closed loop.  simulate() (utils/simulation.py:44-58) advances the SIMULATION model's state
 * with its own prediction and hands the controller only the observation (the first obs_dim entries);
 * with a lift set, ampc_mppi_closed_loop[_scored] does the same: init_obs is then the simulation
 * model's initial state [B][surrogate nx], every solve starts from x0 = lift(observation), and the
 * recorded rows traj_obs are [B][n_steps+1][surrogate nx].  The surrogate only has to share the
 * controls and start its state with the observation.
 *   kinds[k]: 0 identity, 1 o ** params[k] (integer power), 2 sin(params[k] o), 3 cos(params[k] o);
 *   n_basis * obs_dim must equal the controller model's state dimension;  n_basis = 0 removes the lift. */
int ampc_mppi_plan_set_state_lift(ampc_mppi_plan* p, int n_basis, const int* kinds, const double* params);

/* ---- trajectory scoring --------------------------------------------------------------------
 * Cost.__call__ (costs/cost.py:27-41) for n_traj finished trajectories of n_rows rows each:
 *   score = sum_t [eval_obs_cost(obs_t) + eval_ctrl_cost(ctrl_t)] + eval_term_obs_cost(obs_last)
 * with the task cost given as a sum of n_terms terms (SumCost._sum_results, sum_cost.py:49-54);
 * kinds[k] selects the term, its parameters are concatenated in `params` in term order:
 *   0 quadratic  (quad_cost.py:7-51)        Q[no*no] R[nu*nu] F[no*no] goal[no]
 *   1 threshold  (thresh_cost.py:8-38)      goal[no] obs_range_lo obs_range_hi threshold
 *                1 per row where max_{lo<=i<hi} |obs_i - goal_i| > threshold
 *   2 box        (thresh_cost.py:40-83)     lower[no] upper[no]  (+-inf allowed)
 *                1 per row where any obs_i < lower_i or obs_i > upper_i
 * obs [n_traj][n_rows][state_dim] (the observation is the first obs_dim entries of a row),
 * ctrls [n_traj][n_rows][ctrl_dim], scores [n_traj].  The handle only provides the device,
 * stream and precision; no model needs to be set. */
int ampc_score_trajectories(ampc_handle* h, int n_traj, int n_rows, int state_dim, int obs_dim,
                            int ctrl_dim, const double* obs, const double* ctrls, int n_terms,
                            const int* kinds, const double* params, double* scores);

/* ampc_mppi_closed_loop followed by ampc_score_trajectories on the device-resident trajectories:
 * the whole of eval_cfg's simulate + cost(traj) (pipeline_tuner.py:222-233) with only the B
 * scores coming back.  obs_dim is the one given to ampc_set_quad_costs.  traj_obs / traj_ctrls
 * may be NULL. */
int ampc_mppi_closed_loop_scored(ampc_mppi_plan* p, ampc_handle* surrogate, const double* init_obs,
                                 int n_steps, uint64_t seed, const double* eps_all, int n_terms,
                                 const int* kinds, const double* params, double* scores,
                                 double* traj_obs, double* traj_ctrls);

/* ---- iLQR ---------------------------------------------------------------------------------
 * B independent problems of horizon H (IterativeLQR.compute_ilqr_default, ilqr.py:100-265, with
 * its constants u_threshold 1e-3, ls_max_iter 10, ls_discount 0.2, ls_cost_threshold 0.3).
 * clip_to_bounds != 0 clips controls to the handle's bounds in the forward pass (ilqr.py:62-64,
 * 203-204).  Per problem:  x0 [B][nx], uguess [B][H][nu] in;  states [B][H+1][nx],
 * ctrls [B][H][nu], Ks [B][H][nu][nx], ks [B][H][nu], converged/iters/status [B], objective [B]
 * out (any output pointer may be NULL).
 * status: 0 ok; 1 singular Quu (the reference raises numpy.linalg.LinAlgError, ilqr.py:179);
 *         2 line search produced no candidate. */
int ampc_ilqr_plan_create(ampc_handle* h, int B, int horizon, double dt, const int* cost_index,
                          int clip_to_bounds, ampc_ilqr_plan** out);
int ampc_ilqr_plan_destroy(ampc_ilqr_plan* p);
/* Per-kernel timing for the roofline report: when enabled, every iteration of ampc_ilqr_solve
 * brackets its four launches with HIP events on the handle's stream.  ampc_ilqr_plan_timing
 * synchronises and returns the AVERAGE duration (ms) per iteration of
 *   kernel_ms[0] backward sweep   [1] line search + acceptance   [2] forward pass (activation
 *   derivatives of the accepted trajectory)   [3] Jacobian chain
 * over the iterations since the last call, and their number; then resets the counters.
 * enable = n > 1: a queue (ampc_ilqr_solve_queue) brackets only every n-th iteration. */
int ampc_ilqr_plan_set_timing(ampc_ilqr_plan* p, int enable);
int ampc_ilqr_plan_timing(ampc_ilqr_plan* p, double* kernel_ms, int* iterations);
/* Work of the last ampc_ilqr_solve: iterations performed (the largest per-problem count; the
 * polled loop may have queued a few no-op launches beyond it, which the timing averages skip), and
 * candidate rows the line searches rolled out, summed over the plan's problems.  The reference rolls out all ls_max_iter = 10 step
 * sizes in every iteration (ilqr.py:196-205) and then accepts the first that passes its test; the
 * f64 MLP path rolls them out four at a time and stops at the first accepted one (same decisions,
 * same results: a multiple of 4 rows per iteration) or -- many problems per launch, once some search
 * needs a third four-row pass -- all of them in one pass of a twelve-row tile (ls_max_iter rows per
 * iteration; same results again, csrc/ilqr_lsw.hpp).  Either pointer may be NULL. */
int ampc_ilqr_plan_stats(ampc_ilqr_plan* p, long long* iterations, long long* candidate_rows);
/* use_goal = 0 (default): the backward sweep is seeded with the terminal gradient exactly as the
 * reference computes it, (F + F') x_N -- Cost.eval_term_obs_cost_diff ignores the goal
 * (cost.py:195, 208-211).  use_goal != 0: (F + F') (x_N - goal), the derivative of the terminal
 * cost actually charged; what QuadCost(strict_reference=False) asks for. */
int ampc_ilqr_plan_set_terminal_goal(ampc_ilqr_plan* p, int use_goal);
/* The constants compute_ilqr_default takes as arguments (ilqr.py:100-101; defaults u_threshold 1e-3,
 * ls_max_iter 10, ls_discount 0.2, ls_cost_threshold 0.3 -- what a new plan holds): convergence / early-exit
 * norm, number of step sizes alpha_j = ls_discount^j (1..16: they are the rows of one MFMA tile), and the
 * acceptance ratio.  max_iter is an argument of every solve.  Applies to the solves that follow. */
int ampc_ilqr_plan_set_constants(ampc_ilqr_plan* p, double u_threshold, int ls_max_iter, double ls_discount,
                                 double ls_cost_threshold);
int ampc_ilqr_solve(ampc_ilqr_plan* p, const double* x0, const double* uguess, int max_iter,
                    double* states, double* ctrls, double* Ks, double* ks, int* converged,
                    int* iters, int* status, double* objective);

/* Continuous batching: n_problems independent problems streamed through the plan's B slots.
 * ampc_ilqr_solve runs a batch for as long as its slowest problem; here a slot whose problem has
 * converged, failed or used up max_iter iterations takes the next unsolved problem at the following
 * iteration boundary -- on the device, with no host round trip (the finished problem's results go to
 * its output row, the new problem's guess is rolled out by the same launch that line-searches the
 * other slots) -- so every slot stays busy until the queue is empty.  Each problem's arithmetic is
 * exactly that of a one-problem ampc_ilqr_solve (IterativeLQR.compute_ilqr_default, ilqr.py:100-265):
 * results are bit-identical to it, whatever shares the plan.  What the tuner's candidate evaluator
 * feeds (one problem per candidate and control step, ilqr.py:267-295) and what bench.py's c4 times.
 *   x0 [n][nx];  uguess [n][H][nu] or NULL (zeros: IterativeLQR.run's guess, ilqr.py:280-281);
 *   cost_index [n] or NULL (block 0);  outputs as ampc_ilqr_solve, [n] rows, any may be NULL.
 * The plan's horizon, dt, bounds mode and terminal-gradient mode apply to every problem. */
int ampc_ilqr_solve_queue(ampc_ilqr_plan* p, int n_problems, const double* x0, const double* uguess,
                          const int* cost_index, int max_iter, double* states, double* ctrls,
                          double* Ks, double* ks, int* converged, int* iters, int* status,
                          double* objective);

/* ampc_ilqr_solve_queue with a horizon per problem: horizon[n], each in [1, the plan's horizon H] (NULL: H for
 * every problem = ampc_ilqr_solve_queue).  The tuner's iLQR candidates differ in their horizon
 * (IterativeLQRFactory: 5..25, control/ilqr.py:31-41); with this they share ONE plan of H = the longest -- a
 * slot's sweep, line search and Jacobian refresh run over its own problem's horizon, the launches are shared.
 * All arrays keep H as their stride: uguess [n][H][nu] (rows past a problem's horizon are ignored), outputs
 * states [n][H+1][nx], ctrls [n][H][nu], Ks [n][H][nu][nx], ks [n][H][nu] (rows past it are zero).  A
 * problem's results are bit-identical to ampc_ilqr_solve on a one-problem plan of its own horizon.
 * model_index[n] (NULL: the plan's own model for every problem): the problem's controller model, an entry of
 * the table installed with ampc_ilqr_plan_set_models -- results are those of a plan built on that model. */
int ampc_ilqr_solve_queue_var(ampc_ilqr_plan* p, int n_problems, const double* x0, const double* uguess,
                              const int* cost_index, const int* horizon, const int* model_index, int max_iter,
                              double* states, double* ctrls, double* Ks, double* ks, int* converged, int* iters,
                              int* status, double* objective);
/* Controller models per problem.  The tuner's eval_cfg builds the controller with pipeline(cfg, task, trajs),
 * which instantiates and trains a model PER CONFIGURATION when the pipeline has a model factory
 * (pipeline.py:138-145; tuning/pipeline_tuner.py:213-215): candidates of one batch may carry different models.
 * models[n_models]: handles holding MLPs of the plan's shape (same dimensions, hidden layers, activation,
 * precision, device) with their own weights and normalisers; the plan keeps a device table of their
 * descriptors (and the handles alive).  Staging new weights into one of them afterwards needs another call.
 * n_models = 0 removes the table.  Cost blocks, bounds and the surrogate remain the plan handle's. */
int ampc_ilqr_plan_set_models(ampc_ilqr_plan* p, int n_models, ampc_handle* const* models);
/* ... for an MPPI plan: model_index[B] names the table entry of each of the plan's problems; every rollout of
 * problem b then runs on models[model_index[b]] (MLP plans: the sixteen-row and the four-row rollout). */
int ampc_mppi_plan_set_models(ampc_mppi_plan* p, int n_models, ampc_handle* const* models,
                              const int* model_index);

/* simulate() with IterativeLQR controllers, device resident (utils/simulation.py:44-63 as eval_cfg drives it,
 * pipeline_tuner.py:222-231; IterativeLQR.run, ilqr.py:267-295: every control step is a full solve from a zero
 * guess, u = ubar_0, then obs <- surrogate.pred(obs, u)).  n_chains episodes of n_steps control steps stream
 * through the plan's B slots: a slot keeps its episode -- solve, surrogate step, next solve, with no host round
 * trip -- until it ends, then takes the next episode; slots whose solve has converged do not wait for the
 * others (continuous batching as in ampc_ilqr_solve_queue).  The model state must be the observation.
 *   init_obs [n][nx];  cost_index [n] or NULL (block 0);  surrogate: NULL = the plan's own model;
 *   traj_obs [n][n_steps+1][nx], traj_ctrls [n][n_steps+1][nu] (last control row zero, as simulate() returns);
 *   failed [n]: 1 = a solve hit a singular Quu (the reference's LinAlgError, which eval_cfg scores inf,
 *   pipeline_tuner.py:236-239) -- the episode stops there, steps_done [n] tells after how many control steps;
 *   iterations [n]: iLQR iterations over the episode.  Any output may be NULL. */
int ampc_ilqr_closed_loop(ampc_ilqr_plan* p, ampc_handle* surrogate, int n_chains, const double* init_obs,
                          const int* cost_index, int n_steps, int max_iter, double* traj_obs,
                          double* traj_ctrls, int* failed, int* steps_done, long long* iterations);
/* ... with an iLQR horizon per episode: horizon[n] in [1, the plan's horizon] (NULL: the plan's), and a
 * controller model per episode: model_index[n] (NULL: the plan's own), as ampc_ilqr_solve_queue_var;
 * everything else as ampc_ilqr_closed_loop (the surrogate is shared). */
int ampc_ilqr_closed_loop_var(ampc_ilqr_plan* p, ampc_handle* surrogate, int n_chains, const double* init_obs,
                              const int* cost_index, const int* horizon, const int* model_index, int n_steps,
                              int max_iter, double* traj_obs, double* traj_ctrls, int* failed, int* steps_done,
                              long long* iterations);

#ifdef __cplusplus
}
#endif
#endif /* AUTOMPC_HIP_H */
