/*
 * bmpc.h -- C ABI of libbmpc.so: batched HECTOR force-and-moment MPC on MI355X (gfx950).
 *
 * Drop-in boundary for the hot path of junhengl/biped_mpc_py (REF = bipedalLocomotionMPC.py):
 * one call solves B independent instances of what REF:187-304 `solve_mpc` solves once.
 * The reference has no FFI of its own (it is one Python file); these entry points are what a
 * ctypes binding replacing the body of `solve_mpc` would call.
 *
 * Conventions: plain C, no torch / HIP types in signatures.  All arrays are row-major and
 * instance-major.  Every function returns 0 on success or a negative bmpc_status code;
 * bmpc_last_error() gives a thread-local message.  A handle owns its device memory and stream;
 * one handle is not thread-safe, distinct handles are.  Per-instance failures (iteration cap,
 * NaN) are reported in status[], never by failing the batch.
 */
#ifndef BMPC_H
#define BMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMPC_ABI_VERSION 11

/* `stream` arguments are hipStream_t values passed as void*.  NULL is HIP's null (legacy default)
 * stream -- what torch.cuda.current_stream().cuda_stream is when no stream context is active -- so a
 * launch is ordered after the caller's earlier work on that stream and before its later work, exactly
 * like a hip* call.  BMPC_STREAM_OWN selects the handle's private (non-blocking) stream: the one bmpc_debug_assemble and the
 * host-pointer low-level entries run on.  bmpc_solve_batch / bmpc_solve_batch_f64 / bmpc_solve_batch_io run their chunks on
 * three prioritised streams of their own, which wait for whatever the handle's own stream held when the call began. */
#define BMPC_STREAM_OWN ((void*)(intptr_t)-1)

enum bmpc_status {
  BMPC_OK = 0,
  BMPC_ERR_INVALID = -1,     /* bad argument (null pointer, B > max_batch, unsupported h ...) */
  BMPC_ERR_NO_DEVICE = -2,   /* no usable HIP device: the library never falls back to the CPU */
  BMPC_ERR_HIP = -3,         /* a HIP runtime call failed */
  BMPC_ERR_ALLOC = -4
};

/* bmpc_params.path: which kernel family solves (REF:203-216 keeps the stage structure that BMPC_PATH_STAGE exploits) */
enum bmpc_path {
  BMPC_PATH_AUTO = 0,
  BMPC_PATH_DENSE = 1,
  BMPC_PATH_STAGE = 2
};

enum bmpc_penalty_mode {
  BMPC_PENALTY_SCALED = 0,
  BMPC_PENALTY_ABSOLUTE = 1
};

/* bmpc_params.rescue */
enum bmpc_rescue_mode {
  BMPC_RESCUE_AUTO = -1,
  BMPC_RESCUE_OFF = 0,
  BMPC_RESCUE_ON = 1
};

/* per-instance status[] values written by the solver */
enum bmpc_instance_status {
  BMPC_SOLVED = 0,           /* stopping criteria met */
  BMPC_MAX_ITER = 1,         /* iteration cap reached (result is the last iterate).  The stopping criteria are three: the
                                primal and the step residual within eps_pri / eps_dua, and no inactive row still pulling
                                (penalty x |z~ - z| <= 1e-5 x 2 min R x max(1, |x|)); an instance that fails only the third
                                re-classifies at once, and one that can no longer re-classify (max_refactor spent, or
                                adapt_every = 0) is accepted on the first two alone -- it is never held to the cap by it */
  BMPC_NUMERICAL = 2         /* NaN/Inf encountered */
};

/*
 * Parameter block.  Field names follow the reference's attribute bags:
 *   MPC   (REF:22-32): h, dt, x_cmd, Q, R, kv
 *   Biped (REF:34-48): m, I, lt, lh, g, mu, f_max, f_min, tau_max, tau_min
 * plus the solver's own knobs.  `half` is the gait half period used by the reference-foot
 * generator (REF:101-105 hard-codes 5).  bmpc_default_params() fills the reference defaults.
 */
typedef struct bmpc_params {
  int32_t h;                 /* horizon length (REF:24); supported: see bmpc_supported_horizon */
  int32_t half;              /* gait half period in steps (REF:101: 5) */
  double dt;                 /* REF:25 */
  double kv;                 /* REF:29 */
  double x_cmd[12];          /* REF:26; used when the per-instance x_cmd argument is NULL */
  double Q[13];              /* REF:27 (13th weight acts on the constant state: inert) */
  double R[12];              /* REF:28 */
  double m;                  /* REF:36 */
  double I[9];               /* REF:37-39 body inertia, row-major */
  double lt, lh;             /* REF:40-41 (the 0.01 / 0.02 margins of REF:254-255 are applied inside) */
  double g;                  /* REF:42 */
  double mu;                 /* REF:44; used when the per-instance mu argument is NULL */
  double f_max[3], f_min[3];     /* REF:45-46 */
  double tau_max[3], tau_min[3]; /* REF:47-48 */
  /* solver (ADMM with active-set adaptive penalties) */
  double rho;                /* initial penalty on every row (default 0.03; 0.045 at h >= 20); see penalty_mode */
  double rho_eq_scale;       /* multiplier for rows with l == u (pinned variables) */
  double rho_lo;             /* floor of the per-row penalties */
  double rho_hi_f;           /* ceiling for force-like rows (force box, friction) */
  double rho_hi_m;           /* ceiling for moment-like rows (moment box, line-foot) */
  double kappa;              /* per re-classification a row's penalty moves by this factor: up
                                (towards its ceiling) if the row is active, down (towards rho_lo) if not;
                                damped to sqrt(kappa) after 10 factorisations of an instance, to its
                                fourth root after 16 (rare active-set cycles) */
  double alpha;              /* over-relaxation */
  double eps_pri, eps_dua;   /* relative stopping tolerances */
  int32_t max_iter;          /* iteration cap (default 1000 at h <= 12, else 1500; worst seen at the reference's weights: 240 / 315) */
  int32_t check_every;       /* stopping test period */
  int32_t adapt_start;       /* first penalty re-classification (default 5 at h <= 12, 10 at h = 14 .. 18 and h > 20, 20 at h = 20) */
  int32_t adapt_every;       /* re-classification period (0 = never; default 5 at h <= 12 -- see adapt_early --, 20 at h = 14 .. 20,
                                10 beyond: the period follows the cost of a factorisation relative to an iteration) */
  int32_t max_refactor;      /* cap on re-factorisations per instance (then plain ADMM with the penalties reached); default 60:
                                a decade or two away from the reference's weights 1 instance in ~300 keeps re-classifying
                                for up to 60 rounds (damped moves) and then converges; at the reference's weights <= 18 */
  int32_t warm_adapt_start;  /* first re-classification of a warm-started solve (bmpc_set_warm_start); 0 = adapt_start */
  int32_t path;              /* kernel family: BMPC_PATH_AUTO (default: the faster one for h), BMPC_PATH_DENSE (explicit
                                6h x 6h inverse in registers; h <= 20) or BMPC_PATH_STAGE (stage-structured Riccati solve,
                                O(h) work and state; every supported h).  Same optimum, same outer method. */
  int32_t penalty_mode;      /* BMPC_PENALTY_SCALED (default): rho, rho_lo, rho_hi_*, rho_eq are the values at the reference
                                problem -- REF:22-48 defaults at THIS problem's horizon for h <= 20 (so a fixed model shows
                                no horizon scaling between h = 8 and h = 20: the absolute values were tuned and soaked
                                there), at h = 10 for h > 20 (the stiff end then grows like sum k^2 ~ h^3) -- and are
                                scaled by the curvature of the problem at hand relative to it: stiff end (Q, dt, m, I, h)
                                for the ceilings and rho_eq, 2 R for the floor, their geometric mean for the start.  The
                                ceilings and rho_eq are capped at 1e6 (2 min R + rho_lo) -- 4e5 on the dense family --, what
                                the f32 factors hold.
                                Degenerate curvature scales (every Q of a state group zero) are BMPC_ERR_INVALID, not a
                                silent fallback.  BMPC_PENALTY_ABSOLUTE: the fields are taken as they are.
                                bmpc_effective_penalties() returns what a block resolves to. */
  int32_t rescue;            /* BMPC_RESCUE_AUTO (default), _OFF, _ON: after a solve on the dense family, the instances whose
                                status is not 0 are solved again by the stage family (one more launch on the same stream,
                                its workgroups leave at once where the status is 0; the outputs of a rescued instance,
                                iters / nfactor / residuals included, are those of the second solve).  AUTO: on unless the
                                model and weights are the reference's own (REF:22-48), where the dense family has not
                                lost an instance in 6 M and the launch would only cost ~1 %.  No effect on the stage path. */
  int32_t accel;             /* 1 (default) / 0: secant extrapolation of the iterate at the stopping tests (Anderson acceleration with
                                memory one: w <- T(w) - gamma (T(w) - w), gamma from the last two state changes; two sums in the
                                reduction the stopping test already pays for).  Both families (not the dense kernel of h = 12 nor
                                the stage kernel of h = 22 / 24: no LDS / registers for it): 5-7 % fewer iterations and
                                factorisations up to h = 20, 2-5 % beyond, 1-4 % less kernel time.  Same fixed point. */
  int32_t adapt_early;       /* two-rate re-classification schedule (ABI 10): the first `adapt_early` re-classifications -- the one at
                                adapt_start included -- are `adapt_every` iterations apart, the later ones `adapt_late`.  The active set
                                is found in the first ~20 iterations (45 of 240 rows change class between iterations 10 and 20,
                                < 1 after 40), so early re-classifications are worth a factorisation each and late ones mostly
                                walk a few rows along their ladder.  Default at h <= 12: 5 / 5 / 3 / 20 (iterations 5, 10, 15, 35,
                                55 ...); 0 (or adapt_late = 0): every re-classification adapt_every apart (the schedule of ABI <= 9) */
  int32_t adapt_late;
  int32_t adapt_busy;        /* ... but `adapt_busy` iterations after a (late) re-classification that still found more than `adapt_flips` of
                                the instance's rows in another class than the one before: the instances that keep turning are the
                                tail of a batch and are not made to wait.  0: always adapt_late.  Default at h <= 12: 10, 1.
                                (Ignored -- always adapt_late -- by the kernels that do not count class changes: the dense h = 12
                                kernel, whose reduction has no slot for it, and the five-steps-per-lane stage variant, h = 21 .. 24.) */
  int32_t adapt_flips;
  int32_t confirm_from;      /* confirmation: from re-classification number confirm_from + 1 on, a row found in the SAME class as at the
                                previous re-classification moves by kappa_confirm instead of kappa (>= the length of a ladder: straight
                                to its ceiling / floor) -- a row that turns late otherwise costs three more factorisations walking
                                there.  Default at h <= 12: 3, 400; kappa_confirm = 0: off.  (Ignored by the five-steps-per-lane
                                stage variant, h = 21 .. 24 -- no register for the previous classes; the dense h = 12 kernel DOES
                                confirm: it keeps the classes, it only does not count their changes.) */
  int32_t reserved0;
  double kappa_confirm;
  /* low-level control side of the loop (REF:29-32, 43): used by bmpc_low_level_control* / bmpc_foot_position_world* only */
  double kp[9], kd[9];       /* REF:30-31, row-major 3x3 */
  double swingHeight;        /* REF:32 */
  double hip_offset[3];      /* REF:43 */
} bmpc_params;

typedef struct bmpc_handle_s* bmpc_handle;

/* Library / ABI identification. */
int bmpc_abi_version(void);
const char* bmpc_last_error(void);
/* 1 if a kernel is built for this horizon, else 0: every h in [1, 40] (REF:24: the horizon is a plain field of MPC; odd and
 * short horizons run on the stage-structured family).
 * bmpc_supported_horizon_path(h, path) asks for one kernel family (dense: even h in [8, 20]). */
int bmpc_supported_horizon(int h);
int bmpc_supported_horizon_path(int h, int path);
/* The penalties the kernels will use for this parameter block after the scaling of `penalty_mode`:
 * out5 = {rho, rho_eq, rho_lo, rho_hi_f, rho_hi_m}.  Host arithmetic only (no device needed). */
int bmpc_effective_penalties(const bmpc_params* params, double* out5);
/* The kernel family a handle's solves run on (BMPC_PATH_DENSE or BMPC_PATH_STAGE; <0 on error). */
int bmpc_solver_path(bmpc_handle h);
/* 1 if this handle's solves are followed by the rescue pass (see bmpc_params.rescue), else 0 (<0 on error). */
int bmpc_rescue_enabled(bmpc_handle h);
/* Reference defaults (REF:22-48) and solver defaults for horizon h. */
int bmpc_default_params(bmpc_params* p, int h);

/* Create a solver bound to HIP device `device` for at most max_batch instances per call.
 * Replaces: constructing MPC() / Biped() (REF:475-476) -- the parameters are uploaded once. */
int bmpc_create(bmpc_handle* out, const bmpc_params* params, int device, int max_batch);
int bmpc_destroy(bmpc_handle h);
/* Replace the parameter block (same horizon as at creation). */
int bmpc_set_params(bmpc_handle h, const bmpc_params* params);
int bmpc_get_params(bmpc_handle h, bmpc_params* out);

/*
 * Solve B instances; HOST pointers.  Replaces REF:187-304 solve_mpc for a batch.
 *   x_fb     [B][12]   state feedback (REF:13 ordering: euler, pos, omega_w, v_w)
 *   foot     [B][6]    world foot positions [foot1 xyz, foot2 xyz] (REF:479)
 *   contact  [B][h][2] 0/1 contact schedule (REF:482-484)
 *   phase    [B]       k = int(t // dt) % h (REF:99-100), computed by the caller in fp64
 *   x_cmd    [B][12]   or NULL -> params.x_cmd
 *   mu       [B][h][2] or NULL -> params.mu
 * outputs
 *   controls [B][h][12] row k = [f1 f2 m1 m2] (REF:302)
 *   states   [B][h][13] or NULL; row k = predicted state at step k+1 (REF:301)
 *   iters    [B] or NULL, residuals [B][2] or NULL (primal, step), status [B] or NULL
 *   nfactor  [B] or NULL: factorisations used
 * Synchronous: returns after the results are in the output arrays.
 */
int bmpc_solve_batch(bmpc_handle h, int B,
                     const float* x_fb, const float* foot, const uint8_t* contact,
                     const int32_t* phase, const float* x_cmd, const float* mu,
                     float* controls, float* states,
                     int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor);

/*
 * Same with fp64 outputs -- the dtype REF:300-304 returns (`states`, `controls` are fp64 arrays there) -- so that a caller
 * who needs the reference's dtype does not widen 25 h values per instance in a second pass: the widening happens while the
 * results are unpacked from the pinned staging block, chunk by chunk, overlapped with the solve of the later chunks.  The
 * values are the fp32 results of bmpc_solve_batch, exactly (float -> double is exact).
 *
 * Both host-pointer entries stage through page-locked memory owned by the handle (one packed block in, one packed block per
 * chunk out) and split a batch of >= 1024 instances into up to 3 contiguous chunks (55 / 30 / 15 %) on streams of descending
 * priority: a chunk's device-to-host copy and its unpacking into the caller's pageable arrays overlap the later chunks' solves.  Results do not depend on the chunking (the kernels' arithmetic does not depend on the
 * position in a batch).  With warm start, a dispatch order or the profile buffer set the batch goes out as one chunk.
 * bmpc_last_kernel_ms afterwards: from the start of the first chunk's kernel to the end of the last one's.
 */
int bmpc_solve_batch_f64(bmpc_handle h, int B,
                         const float* x_fb, const float* foot, const uint8_t* contact,
                         const int32_t* phase, const float* x_cmd, const float* mu,
                         double* controls, double* states,
                         int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor);

/*
 * The handle's I/O block (ABI 10): host arrays the caller fills and reads IN PLACE -- what a control loop that keeps its buffers
 * wants, and the fastest way across PCIe.  The block is page-locked host memory owned by the handle and mapped into the device's
 * address space: the inputs cross in ONE copy, and the results arrive in the host arrays already widened to the fp64
 * REF:300-304 returns (the widening happens in the kernels' epilogues; the values are the fp32 results of bmpc_solve_batch,
 * exactly) with no unpacking pass: `controls` and the per-instance counters are stored by the kernels straight into the host
 * arrays, `states` follow by copy engine chunk by chunk (up to 3 chunks on prioritised streams, as in bmpc_solve_batch; the last
 * chunk's states go the way of the controls) -- on MI355X a kernel's own stores into host memory sustain ~8.6 GB/s, the copy
 * engine ~50 GB/s, and a 4096-instance batch returns 8.2 MB in 0.8 ms.
 *   bmpc_host_io(h, B, with_x_cmd, with_mu, with_states, &views)   lays the block out for batches of exactly B instances and
 *       returns the array pointers (layouts as in bmpc_solve_batch; x_cmd / mu / states NULL unless asked for).  The views hold
 *       this layout until the next bmpc_host_io of the handle (another layout re-uses or outgrows the block; the memory behind
 *       an old view is not freed before bmpc_destroy, so a stale view reads stale data, never unmapped memory).
 *   bmpc_solve_batch_io(h, B)   solves what the input views hold; synchronous: on return the output views hold the results.
 * Ordered after whatever the handle's own stream held when the call began (BMPC_STREAM_OWN launches).  Warm start, dispatch
 * order and the rescue pass apply as for bmpc_solve_batch_device (with warm start or a dispatch order: one chunk).
 */
typedef struct bmpc_host_views {
  float* x_fb;        /* [B][12] */
  float* foot;        /* [B][6] */
  uint8_t* contact;   /* [B][h][2] */
  int32_t* phase;     /* [B] */
  float* x_cmd;       /* [B][12] or NULL */
  float* mu;          /* [B][h][2] or NULL */
  double* controls;   /* [B][h][12] */
  double* states;     /* [B][h][13] or NULL */
  int32_t* iters;     /* [B] */
  float* residuals;   /* [B][2] */
  int32_t* status;    /* [B] */
  int32_t* nfactor;   /* [B] */
} bmpc_host_views;
int bmpc_host_io(bmpc_handle h, int B, int with_x_cmd, int with_mu, int with_states, bmpc_host_views* out);
int bmpc_solve_batch_io(bmpc_handle h, int B);
/* Layout generation of the handle's I/O block (ABI 11): a counter that moves with EVERY bmpc_host_io call of the handle, failed
 * ones included (a failed call leaves no layout: bmpc_solve_batch_io then refuses).  A caller that caches the views compares it
 * with the value it read after its own bmpc_host_io: any other layout call in between -- same B, other with_* flags, hence other
 * offsets -- shows, instead of the cached views being trusted.  Returns the counter (>= 0), or a negative error code. */
int bmpc_host_io_generation(bmpc_handle h);

/*
 * Same, DEVICE pointers (memory of the handle's device), asynchronous on `stream`
 * (see BMPC_STREAM_OWN above for NULL and the handle's own stream).  Nothing is copied.
 * This is the entry the bench times and the one a device-resident control loop uses.
 */
int bmpc_solve_batch_device(bmpc_handle h, int B,
                            const float* x_fb, const float* foot, const uint8_t* contact,
                            const int32_t* phase, const float* x_cmd, const float* mu,
                            float* controls, float* states,
                            int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor,
                            void* stream);

/* Block until everything queued on the handle's own stream is done, and the LAST bmpc_solve_batch_device /
 * bmpc_rollout_device solve launch of this handle whatever stream it was given (earlier launches on a caller's
 * stream are the caller's to wait for, with that stream).
 *
 * One stream in flight per handle: a handle owns ONE set of per-solve state -- the warm-start buffer (read at the
 * start of a solve, written at its end), the roll-out scratch, the event pair of bmpc_last_kernel_ms -- and nothing
 * orders two solves of the same handle that run on DIFFERENT streams.  Use one stream per handle at a time (or one
 * handle per stream); solves on one stream are ordered like any other work on it. */
int bmpc_synchronize(bmpc_handle h);

/*
 * Introspection for parity tests of the assembly stage (DEVICE->HOST copy inside).
 * Runs the assembly only and returns, per instance, in fp64:
 *   x_ref [B][h][12] (REF:61-70), foot_ref [B][h][6] (REF:72-109),
 *   Gt [B][6h][6h] wrench-space Hessian, qt [B][6h] wrench-space gradient,
 * any of which may be NULL.
 */
int bmpc_debug_assemble(bmpc_handle h, int B,
                        const float* x_fb, const float* foot, const uint8_t* contact,
                        const int32_t* phase, const float* x_cmd, const float* mu,
                        double* x_ref, double* foot_ref, double* Gt, double* qt);

/*
 * The step either side of the MPC solve, batched; HOST pointers, synchronous.
 *   bmpc_foot_position_world  replaces getFootPositionWorld (REF:406-424, with getFootPositionBody REF:367-404):
 *       x_fb [B][12], q [B][10] joint angles  ->  pf_w [B][6]
 *   bmpc_low_level_control    replaces lowLevelControl (REF:444-470, with getLegKinematics REF:306-365 and
 *       swingLegControl REF:426-442):
 *       x_fb [B][12], t [B] (seconds, fp64), pf_w [B][6], q [B][10], qd [B][10],
 *       contact0 [B][2] = contact[0, 0:2], u0 [B][12] = controls[0]  ->  tau [B][10]
 * The *_device variants take DEVICE pointers and a stream (NULL = the null stream, BMPC_STREAM_OWN = the
 * handle's) and do not synchronise.
 */
int bmpc_foot_position_world(bmpc_handle h, int B, const float* x_fb, const float* q, float* pf_w);
int bmpc_foot_position_world_device(bmpc_handle h, int B, const float* x_fb, const float* q, float* pf_w, void* stream);
int bmpc_low_level_control(bmpc_handle h, int B, const float* x_fb, const double* t, const float* pf_w,
                           const float* q, const float* qd, const uint8_t* contact0, const float* u0, float* tau);
int bmpc_low_level_control_device(bmpc_handle h, int B, const float* x_fb, const double* t, const float* pf_w,
                                  const float* q, const float* qd, const uint8_t* contact0, const float* u0,
                                  float* tau, void* stream);

/* Gait scheduler, batched (replaces get_contact_sequence REF:50-59 and the phase index of REF:99-100;
 * SURVEY 8(f) row 2).  phase[b] = int(t[b] // dt) % h in fp64 with Python's float floor division, and
 * contact[b][n][g] = 1 iff leg g is in stance at schedule step k + n:  ((k + n + offset[g]) mod period) < duty[g].
 * bmpc_gait_default(g, half) gives the reference's table generalised to a half period: period = 2 half,
 * offset = {0, half}, duty = {half, half} (half = 5: exactly REF:52-55).  Either output may be NULL.
 * h and dt are the handle's parameters; `contact` has h rows per instance (the reference slices ten rows
 * whatever mpc.h is -- identical for its h = 10). */
typedef struct bmpc_gait {
  int32_t period;
  int32_t offset[2];
  int32_t duty[2];
} bmpc_gait;
int bmpc_gait_default(bmpc_gait* g, int half);
int bmpc_contact_sequence(bmpc_handle h, int B, const double* t, const bmpc_gait* gait, int32_t* phase, uint8_t* contact);
int bmpc_contact_sequence_device(bmpc_handle h, int B, const double* t, const bmpc_gait* gait, int32_t* phase,
                                 uint8_t* contact, void* stream);

/*
 * Receding-horizon use (the reference solves one step, REF:13-17, and lists real-time use as
 * its TODO, README.md:6-7).
 *
 * bmpc_set_warm_start: with enable != 0 every later bmpc_solve_batch* call through this handle leaves the final
 * solver state of each instance (iterate, multipliers, per-row penalties: 48 B per lane) in a device buffer owned
 * by the handle, and starts from the state the previous call left for the same batch index (same B), advanced by
 * `shift` horizon steps (0: the schedule phase did not move; 1: one control period later) and with the penalties
 * pulled back towards their initial value, rho0 (rho / rho0)^theta (theta in [0, 1]; 1 keeps them).  The first
 * call after enabling, after bmpc_reset_warm_start, or with another B starts cold.  The optimum is the same; only
 * the iteration count changes.  enable == 0 switches it off.
 *
 * bmpc_rollout_device: `steps` closed-loop control periods of B instances on one stream, DEVICE pointers, no host
 * arithmetic and no synchronisation: per period  t -> (phase, contact) [bmpc_contact_sequence_device]  ->  solve
 * [bmpc_solve_batch_device]  ->  x_fb <- states[:, 0, 0:12] (the model's own prediction, REF:301), t += dt.
 *   x_fb [B][12] in/out, foot [B][6], t [B] fp64 in/out, gait NULL = the handle's default schedule,
 *   x_cmd [B][12] or NULL, mu [B][h][2] or NULL (constant over the roll-out);
 *   u0_traj [steps][B][12], x_traj [steps][B][12] (state after each period), iters_traj [steps][B],
 *   status_any [B] (OR of the per-period status values): each may be NULL.
 *
 * bmpc_set_dispatch_order: workgroups start in index order and an instance's duration varies (35-105 iterations at
 * h = 10), so the last instances of a batch decide when it ends.  Measured on MI355X (4096 instances): a roll-out
 * spends 3 % less per period with the longest instances first, a plain batch dispatched in its own longest-first order
 * 1-9 % less depending on the box (15 % only when the INPUTS are permuted, tools/order_probe.py: the indirection
 * costs part of it).  The duration cannot be predicted from the inputs, but in a
 * closed loop the previous period's iteration count predicts it: with longest_first_rollouts != 0 (default)
 * bmpc_rollout_device sorts every period's dispatch by the iteration counts of the period before (one small
 * kernel).  order_dev, if non-NULL, is a DEVICE permutation of 0 .. B-1 used by every later solve of this handle
 * (workgroup g solves instance order_dev[g]) and takes precedence; it must stay valid until replaced.  The results
 * never depend on the order.
 */
int bmpc_set_warm_start(bmpc_handle h, int enable, int shift, double theta);
int bmpc_reset_warm_start(bmpc_handle h);
int bmpc_rollout_device(bmpc_handle h, int B, int steps, float* x_fb, const float* foot, double* t,
                        const bmpc_gait* gait, const float* x_cmd, const float* mu,
                        float* u0_traj, float* x_traj, int32_t* iters_traj, int32_t* status_any, void* stream);
int bmpc_set_dispatch_order(bmpc_handle h, const int32_t* order_dev, int longest_first_rollouts);

/* Diagnostics: when device_buf (DEVICE pointer, [max_batch][16] int64) is non-NULL every later solve
 * writes per-instance shader-clock stamps {setup, block algebra, dense sweeps, total, iters,
 * factorisations, -, -, iteration phases P0..P5, stop test + adaptation, -}; NULL switches it off
 * (default).  Costs a few s_memtime per phase. */
int bmpc_debug_set_profile(bmpc_handle h, long long* device_buf);

/* Duration of the LAST bmpc_solve_batch* kernel launch made through this handle, measured with one pair of
 * HIP events recorded around the launch on the launch's stream (milliseconds); <0 if none.  Waits for
 * that launch.  A handle owns ONE event pair: with several launches in flight through the same handle
 * (back-to-back asynchronous calls, or calls on different streams) only the last one is reported, and it
 * is only meaningful if no other launch of this handle overlapped it. */
int bmpc_last_kernel_ms(bmpc_handle h, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* BMPC_H */
