// bmpc_capi.hip -- C ABI (include/bmpc.h) over the HIP kernels.  Built into libbmpc.so.
// There is deliberately no CPU path here: without a HIP device every entry point that would
// compute returns BMPC_ERR_NO_DEVICE.
#include "bmpc_kernels.hip"
#include "bmpc_stage.hip"
#include "bmpc_lowlevel.hip"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "bmpc.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(BMPC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

bool inv3(const double* a, double* o) {
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  if (!(std::fabs(det) > 0)) return false;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return true;
}

bool dense_horizon(int h) { return h >= 8 && h <= 20 && h % 2 == 0; }
bool stage_horizon(int h) { return h >= 1 && h <= 40; }       // (any parity, from ONE step: steps past the horizon are phantoms of the lane map)
// the kernel family that solves horizon h when the caller asks for `path`; 0 if there is none
int resolve_path(int h, int path) {
  if (path == BMPC_PATH_DENSE) return dense_horizon(h) ? BMPC_PATH_DENSE : 0;
  if (path == BMPC_PATH_STAGE) return stage_horizon(h) ? BMPC_PATH_STAGE : 0;
  if (path != BMPC_PATH_AUTO) return 0;
  if (dense_horizon(h)) return BMPC_PATH_DENSE;
  return stage_horizon(h) ? BMPC_PATH_STAGE : 0;
}

// Curvature scales of the condensed Hessian (closed forms at step 0 and zero attitude; S2 = sum_{k < h} k^2):
//   torque space  g_tau[a] = 2 (Q_e[a] (dt^2 Iinv_aa)^2 S2 + Q_w[a] (dt Iinv_aa)^2 h)        (REF:165-184, 278-286)
//   force space   g_F[a]   = 2 (Q_p[a] (dt^2 / m)^2 S2 + Q_v[a] (dt / m)^2 h)
// force-like rows see g_F plus the torque curvature through the lever arm of the nominal CoM height, moment-like rows
// g_tau; the soft end of the spectrum is 2 R.  The penalty fields of bmpc_params are the values AT THE REFERENCE PROBLEM
// (REF:22-48 defaults; horizon: the problem's own up to h = 20, else 10) and scale with these ratios, so that weights, step length, mass, inertia and horizon can
// change without re-tuning (at h = 40 the stiff scale is 70 times the one at h = 10, and with
// absolute ceilings the active rows of the early steps converge at 0.97 per iteration).
struct CurvScales { double force, moment, soft; };
CurvScales curvature_scales(const bmpc_params& p) {
  double Iinv[9];
  CurvScales c = {1.0, 1.0, 1.0};
  if (!inv3(p.I, Iinv)) return c;
  double s2 = 0;
  for (int k = 1; k < p.h; ++k) s2 += (double)k * k;
  const double dt = p.dt, h = p.h;
  double gt[3], gf = 0, rmin = p.R[0];
  for (int a = 0; a < 3; ++a) {
    const double ii = std::fabs(Iinv[4 * a]);
    gt[a] = 2 * (p.Q[a] * (dt * dt * ii) * (dt * dt * ii) * s2 + p.Q[6 + a] * (dt * ii) * (dt * ii) * h);
    gf = std::fmax(gf, 2 * (p.Q[3 + a] * (dt * dt / p.m) * (dt * dt / p.m) * s2 + p.Q[9 + a] * (dt / p.m) * (dt / p.m) * h));
  }
  for (int i = 1; i < 12; ++i) rmin = std::fmin(rmin, p.R[i]);
  const double z0 = p.x_cmd[5];
  c.force = gf + z0 * z0 * std::fmax(gt[0], gt[1]);
  c.moment = std::fmax(gt[0], std::fmax(gt[1], gt[2]));
  c.soft = rmin;
  return c;
}

// Is a dense solve followed by the rescue pass?  BMPC_RESCUE_AUTO: only away from the reference's model and weights
// (REF:22-48) -- there the dense family has converged on every one of 6 M soaked instances and the extra launch would
// cost the headline configuration ~1 % for nothing; at other weights 1 instance in 10^3..10^4 can stall.
bool resolve_rescue(const bmpc_params& p) {
  if (resolve_path(p.h, p.path) != BMPC_PATH_DENSE || !stage_horizon(p.h) || p.rescue == BMPC_RESCUE_OFF) return false;
  if (p.rescue == BMPC_RESCUE_ON) return true;
  bmpc_params ref;
  bmpc_default_params(&ref, p.h);
  bool same = p.dt == ref.dt && p.m == ref.m && p.g == ref.g && p.lt == ref.lt && p.lh == ref.lh;
  for (int i = 0; i < 12; ++i) same = same && p.Q[i] == ref.Q[i] && p.R[i] == ref.R[i];
  for (int i = 0; i < 9; ++i) same = same && p.I[i] == ref.I[i];
  for (int i = 0; i < 3; ++i)
    same = same && p.f_max[i] == ref.f_max[i] && p.f_min[i] == ref.f_min[i] && p.tau_max[i] == ref.tau_max[i] && p.tau_min[i] == ref.tau_min[i];
  return !same;
}

int make_dev_params(const bmpc_params& p, bmpc::DevParams* d) {
  if (!resolve_path(p.h, p.path)) return fail(BMPC_ERR_INVALID, "unsupported horizon h=%d for path %d", p.h, p.path);
  if (p.half < 1) return fail(BMPC_ERR_INVALID, "half must be >= 1");
  if (!(p.dt > 0) || !(p.m > 0)) return fail(BMPC_ERR_INVALID, "dt and m must be positive");
  if (!(p.rho > 0) || !(p.rho_lo > 0) || !(p.rho_hi_f > 0) || !(p.rho_hi_m > 0) || !(p.rho_eq_scale > 0))
    return fail(BMPC_ERR_INVALID, "penalties must be positive");
  if (!(p.kappa > 1)) return fail(BMPC_ERR_INVALID, "kappa must be > 1");
  if (p.kappa_confirm != 0 && !(p.kappa_confirm > 1)) return fail(BMPC_ERR_INVALID, "kappa_confirm must be 0 (off) or > 1");
  if (p.adapt_early < 0 || p.adapt_late < 0 || p.adapt_busy < 0 || p.adapt_flips < 0 || p.confirm_from < 0)
    return fail(BMPC_ERR_INVALID, "adapt_early, adapt_late, adapt_busy, adapt_flips, confirm_from must be >= 0");
  if (p.max_iter < 1 || p.check_every < 1) return fail(BMPC_ERR_INVALID, "max_iter, check_every must be >= 1");
  if (p.rescue < BMPC_RESCUE_AUTO || p.rescue > BMPC_RESCUE_ON) return fail(BMPC_ERR_INVALID, "unknown rescue mode %d", p.rescue);
  std::memset(d, 0, sizeof(*d));
  d->h = p.h; d->half = p.half; d->max_iter = p.max_iter; d->check_every = p.check_every;
  d->adapt_start = p.adapt_start; d->adapt_every = p.adapt_every; d->max_refactor = p.max_refactor;
  d->adapt_early = p.adapt_early; d->adapt_late = p.adapt_late;
  d->adapt_busy = p.adapt_busy; d->adapt_flips = p.adapt_flips;
  d->confirm_from = p.confirm_from; d->kappa_confirm = (float)p.kappa_confirm;
  d->dt = p.dt; d->kv = p.kv; d->m = p.m; d->g = p.g; d->mu = p.mu;
  d->lt = p.lt - 0.01;                       // REF:254
  d->lh = p.lh - 0.02;                       // REF:255
  d->alpha = p.alpha;
  d->accel = p.accel ? 1 : 0;
  for (int i = 0; i < 12; ++i) {
    d->x_cmd[i] = p.x_cmd[i];
    d->Q[i] = p.Q[i];
    d->R2[i] = 2.0 * p.R[i];
    if (!(p.R[i] > 0) || !(p.Q[i] >= 0)) return fail(BMPC_ERR_INVALID, "need R > 0, Q >= 0");
  }
  for (int k = 0; k < 3; ++k) { d->sq_e[k] = std::sqrt(2.0 * p.Q[k]); d->sq_w[k] = p.dt * std::sqrt(2.0 * p.Q[6 + k]); }
  d->kpm = p.dt * p.dt / p.m;
  d->kvm = p.dt / p.m;
  if (!inv3(p.I, d->Iinv)) return fail(BMPC_ERR_INVALID, "inertia matrix is singular");
  for (int i = 0; i < 3; ++i) {
    d->f_max[i] = p.f_max[i]; d->f_min[i] = p.f_min[i];
    d->tau_max[i] = p.tau_max[i]; d->tau_min[i] = p.tau_min[i];
    if (p.f_max[i] < p.f_min[i] || p.tau_max[i] < p.tau_min[i]) return fail(BMPC_ERR_INVALID, "upper bound below lower bound");
  }
  double pf = 1, pm = 1, pr = 1;               // curvature of this problem relative to the reference problem
  if (p.penalty_mode == BMPC_PENALTY_SCALED) {
    // Up to h = 20 the reference problem has the horizon of the problem at hand: the absolute values were tuned and soaked
    // at h = 10, 16 and 20 (10.5 M + 3.3 M instances), and the dense kernels' f32 sweep does not hold much larger ones
    // (ceilings 8x higher at h = 20: 2 % of a standing batch lose convergence, some to NaNs, where the stage-structured
    // kernels -- and the absolute values -- converge on every instance).  The long horizons (stage-structured kernels) follow
    // the stiff end, which grows like sum k^2 ~ h^3, relative to h = 10 (h = 40: 72x before the cap below; absolute values
    // there: 0.97 per iteration; relative to h = 20: 20 % more iterations and four times the non-converged instances).
    bmpc_params ref;
    bmpc_default_params(&ref, p.h <= 20 ? p.h : 10);
    const CurvScales c0 = curvature_scales(ref), c1 = curvature_scales(p);
    if (!(c1.force > 0 && c1.moment > 0 && c1.soft > 0))
      return fail(BMPC_ERR_INVALID, "penalty_mode SCALED needs positive curvature scales (force %g, moment %g, soft %g: some "
                  "tracking weight Q is zero on every state a control acts on); use BMPC_PENALTY_ABSOLUTE", c1.force, c1.moment, c1.soft);
    pf = c1.force / c0.force; pm = c1.moment / c0.moment; pr = c1.soft / c0.soft;
  } else if (p.penalty_mode != BMPC_PENALTY_ABSOLUTE) {
    return fail(BMPC_ERR_INVALID, "unknown penalty_mode %d", p.penalty_mode);
  }
  double rho0 = p.rho * std::sqrt(pr * pf);              // between the soft and the stiff end
  double rho_eq = p.rho * p.rho_eq_scale * pf, rho_lo = p.rho_lo * pr, hi_f = p.rho_hi_f * pf, hi_m = p.rho_hi_m * pm;
  if (p.penalty_mode == BMPC_PENALTY_SCALED) {
    // What f32 holds: the stored null-space factor Ka^-1 has entries up to 1 / (2R + rho_lo) with 6e-8 relative error,
    // and that error is multiplied by the stiffest penalty of the block when the step is taken: above
    // eps_f32 x rho_max / (2R + rho_lo) ~ 1 the iteration stops contracting (seen as active-set cycling at R / 100).
    // The ceilings stay six decades above the soft end (error factor 0.06).
    double rmin = p.R[0];
    for (int i = 1; i < 12; ++i) rmin = std::fmin(rmin, p.R[i]);
    // The dense family holds less: its f32 explicit inverse is the preconditioner of every iteration.  At Q x 10 the moment
    // ceiling resolved to 500 and 2 of 16384 standing h = 20 instances re-classified until the cap, where ceilings of 100 ..
    // 400 converge every instance in <= 490 iterations at the same mean (tools/soak.py options ... rho_hi_m=10..40): 4e5.  The
    // stage family (tuned and soaked at 1e6, h = 40: rho_eq = 500) keeps its cap.  Not binding at the reference's weights.
    const double top = (resolve_path(p.h, p.path) == BMPC_PATH_DENSE ? 4e5 : 1e6) * (2 * rmin + rho_lo);
    rho_eq = std::fmin(rho_eq, top); hi_f = std::fmin(hi_f, top); hi_m = std::fmin(hi_m, top);
    rho0 = std::fmin(rho0, hi_f);
  }
  {
    double rmin = p.R[0];
    for (int i = 1; i < 12; ++i) rmin = std::fmin(rmin, p.R[i]);
    d->r2min = (float)(2 * rmin);
  }
  d->rho = (float)rho0;
  d->rho_eq = (float)rho_eq;
  d->rho_lo = (float)rho_lo;
  d->rho_hi_f = (float)hi_f; d->rho_hi_m = (float)hi_m;
  d->eps_pri = (float)p.eps_pri; d->eps_dua = (float)p.eps_dua; d->kappa = (float)p.kappa;
  {                                           // (f32 products exactly as the kernels used to form them: SLOW_TOL = 1e-6, U0_TOL = 5)
    const float slow_tol = 1.0e-6f, u0_tol = 5.f;
    d->kappa_sqrt = std::sqrt(d->kappa);
    d->kappa_qrt = std::sqrt(std::sqrt(d->kappa));
    d->slow_tol_r2 = slow_tol * d->r2min;
    d->slow_tol_r2_u0 = u0_tol * slow_tol * d->r2min;
    d->eps_u0 = u0_tol * std::fmax(d->eps_pri, d->eps_dua);
  }
  return BMPC_OK;
}

// page-locked host memory (the staging of the host-pointer entry points: copies to and from it run at the full PCIe rate
// and asynchronously, which pageable memory does not allow)
struct PinnedBuf {
  char* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= n) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    // (mapped: the kernels of the host-pointer entry points store their results straight into this memory)
    // (coherent / non-coherent / write-combined make no difference to what a kernel's own stores into it sustain: ~8.6 GB/s, round 5)
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocMapped | hipHostMallocPortable);
    if (e == hipSuccess) n = bytes;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

// The same for memory the CALLER holds pointers into (the I/O block of bmpc_host_io): a block that has to grow is not freed but
// retired until the handle goes -- a view of the old layout that a caller still holds must not dangle --, and it grows by half at
// least, so that what is retired stays below twice what is in use.
struct RetiringPinnedBuf {
  char* p = nullptr;
  size_t n = 0;
  static constexpr int MAX_RETIRED = 64;
  char* retired[MAX_RETIRED] = {};
  int n_retired = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= n) return hipSuccess;
    size_t cap = n + n / 2;
    if (cap < bytes) cap = bytes;
    char* q = nullptr;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&q), cap, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return e;
    if (p) {
      if (n_retired < MAX_RETIRED) retired[n_retired++] = p;
      else (void)hipHostFree(p);              // (64 growths by half: a block 10^11 times the first one -- not reached)
    }
    p = q; n = cap;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    for (int i = 0; i < n_retired; ++i) (void)hipHostFree(retired[i]);
    p = nullptr; n = 0; n_retired = 0;
  }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t ensure(size_t count) {
    if (count <= n) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct bmpc_handle_s {
  int device = 0;
  int max_batch = 0;
  bmpc_params params;
  bmpc::DevParams dev;
  int path = BMPC_PATH_DENSE;      // resolved kernel family (resolve_path)
  bool rescue_on = false;          // dense solves are followed by the stage family's rescue pass (resolve_rescue)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  // staging for the host-pointer entry points.  bmpc_solve_batch*: ONE packed block of inputs and one of outputs on either side
  // of PCIe (pinned on the host), the batch solved in up to HOST_CHUNKS chunks on streams of descending priority so that a
  // chunk's results cross PCIe -- and are unpacked / widened by the calling thread -- while the later chunks still solve
  static constexpr int HOST_CHUNKS = 3;
  PinnedBuf pin_in, pin_out;
  DevBuf<char> dev_in, dev_out;
  hipStream_t cstream[HOST_CHUNKS] = {nullptr, nullptr, nullptr};
  hipEvent_t cev[HOST_CHUNKS] = {nullptr, nullptr, nullptr};
  hipEvent_t cev_own = nullptr;     // what the handle's own stream held when a host-pointer call began: the chunk streams wait for it
  // the chunking of the host-pointer entry points, fixed at creation (BMPC_HOST_CUTS = "a,b" with 0 < a <= b <= 1, or "1": one
  // chunk; anything else is ignored) and the diagnostics switch BMPC_HOST_TIMING
  int host_chunks = HOST_CHUNKS;
  double host_cut[HOST_CHUNKS + 1] = {0.0, 0.55, 0.85, 1.0};
  bool host_timing = false;
  // the handle's I/O block (bmpc_host_io / bmpc_solve_batch_io): page-locked host memory, mapped into the device's address
  // space -- the caller writes its inputs there, the kernels store the fp64 results there
  RetiringPinnedBuf io_in, io_out;
  DevBuf<char> io_dev;
  DevBuf<double> io_states;         // fp64 states of a batch in HBM, on their way to the I/O block by copy engine
  hipEvent_t cev_in = nullptr;      // the I/O block's inputs have arrived
  int io_gen = 0;                   // moves with every bmpc_host_io call (bmpc_host_io_generation)
  struct IoLayout {
    int B = 0;
    bool x_cmd = false, mu = false, states = false;
    size_t i_xfb = 0, i_foot = 0, i_phase = 0, i_xcmd = 0, i_mu = 0, i_con = 0, in_bytes = 0;
    size_t o_u = 0, o_s = 0, o_it = 0, o_st = 0, o_nf = 0, o_rs = 0, out_bytes = 0;
  } io;
  DevBuf<float> x_fb, foot, x_cmd, mu, controls, states, resid;
  DevBuf<uint8_t> contact;
  DevBuf<int32_t> phase, iters, status, nfactor;
  DevBuf<double> dbg;
  DevBuf<float> ll_q, ll_qd, ll_pf, ll_u0, ll_tau;
  DevBuf<double> ll_t;
  DevBuf<uint8_t> ll_c0;
  long long* prof_dev = nullptr;   // optional cycle-stamp buffer (bmpc_debug_set_profile)
  // receding-horizon warm start (bmpc_set_warm_start): solver state of the last batch, kept on the device
  DevBuf<double> warm;
  bool warm_on = false, warm_valid = false;
  int warm_batch = 0, warm_shift = 0;
  float warm_theta = 0.5f;
  // roll-out scratch (bmpc_rollout_device)
  DevBuf<float> ro_controls, ro_states;
  DevBuf<uint8_t> ro_contact;
  DevBuf<int32_t> ro_phase, ro_iters, ro_status, ro_order;
  const int32_t* order = nullptr;   // dispatch order of the next solves (bmpc_set_dispatch_order; roll-outs set their own)
  bool longest_first = true;        // roll-outs order each period's solve by the previous period's iteration counts
};

namespace {

// Horizons with a dense kernel (explicit 6h x 6h inverse held in registers).  h = 20 sits at 255 VGPRs; h = 22 / 24 were
// built and dropped: a 5-wave workgroup caps a lane at 256 registers whatever the launch bounds say, the row halves no
// longer fit (35 / 30 spilled registers) and 78 / 86 KB of LDS leave one workgroup per CU -- those horizons belong to the
// stage-structured kernels (bmpc_stage.hip).
#define BMPC_DENSE_HORIZONS(X) X(8) X(10) X(12) X(14) X(16) X(18) X(20)

template <int H>
int launch_h(bmpc_handle hd, int B, const float* x_fb, const float* foot, const uint8_t* contact,
             const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
             int32_t* iters, float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg,
             hipStream_t st, const int32_t* order, double* c64, double* s64) {
  constexpr int NT = bmpc::Dims<H>::NT;
  bmpc::WarmArgs warm = {nullptr, 0, 0, 0, 1.f, 0, dbg.assemble_only ? nullptr : order, nullptr, c64, s64};
  if (hd->warm_on && !dbg.assemble_only) {
    const size_t need = (size_t)B * NT * 6;
    if (need > hd->warm.n) hd->warm_valid = false;          // growing the buffer loses the stored state
    HIP_TRY(hd->warm.ensure(need));
    warm.buf = hd->warm.p;
    warm.load = (hd->warm_valid && hd->warm_batch == B) ? 1 : 0;
    warm.store = 1;
    warm.shift = hd->warm_shift;
    warm.theta = hd->warm_theta;
    warm.adapt_start = hd->params.warm_adapt_start;
  }
  if (dbg.prof)     // diagnostics build of the same body: in-kernel cycle stamps (bmpc_debug_set_profile)
    hipLaunchKernelGGL((bmpc::solve_kernel_prof<H>), dim3(B), dim3(NT), 0, st, hd->dev, B, x_fb, foot, contact,
                       phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
  else
    hipLaunchKernelGGL((bmpc::solve_kernel<H>), dim3(B), dim3(NT), 0, st, hd->dev, B, x_fb, foot, contact,
                       phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
  HIP_TRY(hipGetLastError());
  if (warm.buf) { hd->warm_valid = true; hd->warm_batch = B; }
  return BMPC_OK;
}

template <int NP, int NW>
int launch_stage(bmpc_handle hd, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                 const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
                 int32_t* iters, float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg,
                 hipStream_t st, const int32_t* order, const int32_t* rescue_status, double* c64, double* s64) {
  bmpc::WarmArgs warm = {nullptr, 0, 0, 0, 1.f, 0, dbg.assemble_only ? nullptr : order, rescue_status, c64, s64};
  if (hd->warm_on && !dbg.assemble_only && !rescue_status) {   // (a rescue pass starts cold: the stored state is the dense family's)
    const size_t need = (size_t)B * (5 * NP * NW) * 12 * 6;  // [B][5 NP NW][12][6] doubles
    if (need > hd->warm.n) hd->warm_valid = false;
    HIP_TRY(hd->warm.ensure(need));
    warm.buf = hd->warm.p;
    warm.load = (hd->warm_valid && hd->warm_batch == B) ? 1 : 0;
    warm.store = 1;
    warm.shift = hd->warm_shift;
    warm.theta = hd->warm_theta;
    warm.adapt_start = hd->params.warm_adapt_start;
  }
  if (dbg.prof)
    hipLaunchKernelGGL((bmpc::stage_kernel_prof<NP, NW>), dim3(B), dim3(64 * NW), 0, st, hd->dev, B, x_fb, foot, contact,
                       phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
  else
    hipLaunchKernelGGL((bmpc::stage_kernel<NP, NW>), dim3(B), dim3(64 * NW), 0, st, hd->dev, B, x_fb, foot, contact,
                       phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
  HIP_TRY(hipGetLastError());
  if (warm.buf) { hd->warm_valid = true; hd->warm_batch = B; }
  return BMPC_OK;
}

int launch_stage_any(bmpc_handle hd, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                     const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
                     int32_t* iters, float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg,
                     hipStream_t st, const int32_t* order, const int32_t* rescue_status, double* c64, double* s64) {
  // compiled per (steps a lane owns, waves per instance): bmpc::stage_steps_per_lane / stage_waves
  switch (10 * bmpc::stage_waves(hd->dev.h) + bmpc::stage_steps_per_lane(hd->dev.h)) {
#define BMPC_CASE(NN, WW) case 10 * WW + NN: return launch_stage<NN, WW>(hd, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, st, order, rescue_status, c64, s64);
    BMPC_CASE(2, 1) BMPC_CASE(3, 1) BMPC_CASE(4, 1) BMPC_CASE(5, 1) BMPC_CASE(3, 2) BMPC_CASE(4, 2)
#undef BMPC_CASE
    default: return fail(BMPC_ERR_INVALID, "unsupported horizon h=%d", hd->dev.h);
  }
}

int launch(bmpc_handle hd, int B, const float* x_fb, const float* foot, const uint8_t* contact,
           const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
           int32_t* iters, float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg,
           hipStream_t st, const int32_t* order, double* c64 = nullptr, double* s64 = nullptr) {
  const bool dense_views = dbg.assemble_only && (dbg.Gt || dbg.qt);      // Gt, qt only exist on the dense path
  if (hd->path == BMPC_PATH_STAGE && !(dense_views && dense_horizon(hd->dev.h))) {
    if (dense_views) return fail(BMPC_ERR_INVALID, "Gt / qt views exist for h <= 20 only (h=%d never forms them)", hd->dev.h);
    return launch_stage_any(hd, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, st,
                            order, nullptr, c64, s64);
  }
  // Dense family.  With the rescue pass on, the instances whose status is not 0 afterwards are solved again by the
  // stage-structured kernel of the same horizon (f32 Riccati recursion instead of the f32 explicit inverse: it does not
  // share the dense sweep's rare breakdowns; profiles/r03_soak.txt): one more launch whose workgroups leave at once
  // where the status is 0, no host round trip.
  const bool rescue = hd->rescue_on && !dbg.assemble_only;
  if (rescue && !status) {
    HIP_TRY(hd->status.ensure((size_t)B));
    status = hd->status.p;
  }
  int rc = BMPC_ERR_INVALID;
  switch (hd->dev.h) {
#define BMPC_CASE(HH) case HH: rc = launch_h<HH>(hd, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, st, order, c64, s64); break;
    BMPC_DENSE_HORIZONS(BMPC_CASE)
#undef BMPC_CASE
    default: return fail(BMPC_ERR_INVALID, "unsupported horizon h=%d", hd->dev.h);
  }
  if (rc != BMPC_OK || !rescue) return rc;
  bmpc::DebugOut quiet = dbg;
  quiet.prof = nullptr;                        // the cycle stamps stay those of the first solve
  return launch_stage_any(hd, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, quiet, st,
                          order, status, c64, s64);
}

// NULL is HIP's null (legacy default) stream, like every hip* call; BMPC_STREAM_OWN the handle's own stream
hipStream_t pick_stream(bmpc_handle h, void* stream) {
  if (stream == BMPC_STREAM_OWN) return h->stream;
  return static_cast<hipStream_t>(stream);
}

int check_common(bmpc_handle h, int B, const void* x_fb, const void* foot, const void* contact, const void* phase,
                 const void* controls) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B > 0 && (!x_fb || !foot || !contact || !phase || !controls))
    return fail(BMPC_ERR_INVALID, "x_fb, foot, contact, phase and controls must be non-null");
  return BMPC_OK;
}

}  // namespace

extern "C" {

int bmpc_abi_version(void) { return BMPC_ABI_VERSION; }

const char* bmpc_last_error(void) { return g_err; }

int bmpc_supported_horizon(int h) { return resolve_path(h, BMPC_PATH_AUTO) ? 1 : 0; }

int bmpc_supported_horizon_path(int h, int path) { return resolve_path(h, path) ? 1 : 0; }

int bmpc_effective_penalties(const bmpc_params* params, double* out5) {
  if (!params || !out5) return fail(BMPC_ERR_INVALID, "null argument");
  bmpc::DevParams d;
  int rc = make_dev_params(*params, &d);
  if (rc != BMPC_OK) return rc;
  out5[0] = d.rho; out5[1] = d.rho_eq; out5[2] = d.rho_lo; out5[3] = d.rho_hi_f; out5[4] = d.rho_hi_m;
  return BMPC_OK;
}

int bmpc_solver_path(bmpc_handle h) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  return h->path;
}

int bmpc_rescue_enabled(bmpc_handle h) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  return h->rescue_on ? 1 : 0;
}

int bmpc_default_params(bmpc_params* p, int h) {
  if (!p) return fail(BMPC_ERR_INVALID, "null params");
  std::memset(p, 0, sizeof(*p));
  p->h = h;
  p->half = (h == 10) ? 5 : (h > 1 ? h / 2 : 1);                      // REF:101 hard-codes 5 at h = 10
  p->dt = 0.04;                                                       // REF:25
  p->kv = 0.01;                                                       // REF:29
  const double xc[12] = {0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0};       // REF:26
  const double Q[13] = {500, 100, 100, 300, 300, 700, 1, 1, 1, 1, 1, 1, 1};   // REF:27
  for (int i = 0; i < 12; ++i) { p->x_cmd[i] = xc[i]; p->R[i] = 1e-4; }       // REF:28
  for (int i = 0; i < 13; ++i) p->Q[i] = Q[i];
  p->m = 12;                                                          // REF:36
  p->I[0] = 0.932; p->I[4] = 0.9420; p->I[8] = 0.0711;                // REF:37-39
  p->lt = 0.09; p->lh = 0.05; p->g = 9.81; p->mu = 0.5;               // REF:40-44
  for (int i = 0; i < 3; ++i) { p->f_max[i] = 500; p->f_min[i] = 0; } // REF:45-46
  p->tau_max[0] = 0; p->tau_max[1] = 67; p->tau_max[2] = 33.5;        // REF:47
  for (int i = 0; i < 3; ++i) p->tau_min[i] = -p->tau_max[i];         // REF:48
  p->rho = h < 20 ? 0.03 : 0.045;          // (h = 20: -2.5 % kernel time, fewer late re-classifications)
  p->rho_eq_scale = h < 20 ? 1e3 : 1e3 * 0.03 / 0.045;               // rho_eq = 30 for every horizon
  p->rho_lo = 3e-4; p->rho_hi_f = 1.0; p->rho_hi_m = 100.0; p->kappa = 20.0;
  p->alpha = 1.6; p->eps_pri = 1e-7; p->eps_dua = 1e-7;
  // (worst seen in the soaks at the reference's weights: 240 at h = 10, 315 at h = 16 / 20 with the periods below.  The caps
  //  are for the instances that keep re-classifying away from those weights: at R / 100 -- soft end two decades further down
  //  -- 55 of 16384 standing h = 20 instances needed up to 60 factorisations and 995 iterations; capped at 24 / 600 they were
  //  reported unsolved by both families.  The mean is untouched: 131.5 iterations either way.)
  p->max_iter = h <= 12 ? 1000 : 1500;
  p->check_every = 5;
  // (long horizons re-classify every 10 iterations: the 1-in-2000 instances that keep re-classifying need 25-35
  //  factorisations and converge by iteration ~350; capped at 24 they freeze their penalties at iteration 240 and run
  //  into max_iter)
  p->max_refactor = 60;
  // Re-classification period ~ (cost of a factorisation) / (cost of an iteration): 10.6 at h = 10, 15.6 at h = 16,
  // 21.5 at h = 20 (profiles/r02_cfg*_phase_cycles.txt).  Measured on MI355X (build_tmp-style A/B, round 2):
  // h = 16: period 20 from iteration 10 is 8 % faster than 10 / 10 (4.3 instead of 5.6 factorisations, 68 instead of
  // 53 iterations), h = 20: 20 / 20 is 12 % faster (4.8 instead of 6.8, 88 instead of 66); h = 10 is best at 10 / 10.
  // (h > 20 runs on the stage-structured kernels, where a factorisation costs 4-5 iterations instead of 15-20:
  //  period 10 again; measured with tools/stage_probe.py)
  p->adapt_every = (h <= 12 || h > 20) ? 10 : 20;
  p->adapt_start = (h < 20 || h > 20) ? 10 : 20;
  // Two rates at h <= 12 (round 5).  Traces of the model (oracle/ws_model.py, 512 oracle-solved instances): of 240 rows 45 change
  // class between iterations 10 and 20, 18 between 20 and 30, 4.5 between 30 and 40, < 1 after -- the active set is found
  // early, and from iteration 40 on a re-classification mostly walks the rows that flipped late along their ladder.  Early
  // re-classifications 5 apart and late ones 20 apart: 48.9 instead of 53.1 iterations AND 5.05 instead of 5.54 factorisations
  // on the standing set (43.8 / 4.76 instead of 47.7 / 5.02 on the mixed one); tools/schedule_explore.py has the grid.
  p->adapt_flips = 1;
  if (h <= 12) {
    p->adapt_start = 5; p->adapt_every = 5; p->adapt_early = 3; p->adapt_late = 20;
    p->adapt_busy = 10; p->confirm_from = 3; p->kappa_confirm = 400.0;
  } else if (h < 20) {
    // h = 14 .. 18 (a factorisation costs 15 iterations): two early re-classifications 10 apart, then 20, confirmation from the
    // third on.  Model (128 oracle-solved walking instances, h = 16): 57.1 / 4.27 instead of 64.7 / 4.08, worst cost 224 instead
    // of 270; MI355X, config 3: 2.07 instead of 2.13 ms per 4096.  h = 20 (21 iterations per factorisation) gains nothing from
    // any of this (6.09 +- 0.03 ms per 8192 over six schedules) and keeps its single rate.
    p->adapt_start = 10; p->adapt_every = 10; p->adapt_early = 2; p->adapt_late = 20;
    p->confirm_from = 2; p->kappa_confirm = 400.0;
  }
  p->rescue = BMPC_RESCUE_AUTO;
  p->accel = 1;
  p->warm_adapt_start = 5;                                            // (tools/warm_sweep.py)
  p->kp[0] = p->kp[4] = p->kp[8] = 500;                               // REF:30
  p->kd[0] = p->kd[4] = p->kd[8] = 10;                                // REF:31
  p->swingHeight = 0.1;                                               // REF:32
  p->hip_offset[0] = -0.005; p->hip_offset[1] = 0.047; p->hip_offset[2] = -0.126;   // REF:43
  return BMPC_OK;
}

int bmpc_create(bmpc_handle* out, const bmpc_params* params, int device, int max_batch) {
  if (!out || !params) return fail(BMPC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (max_batch < 1) return fail(BMPC_ERR_INVALID, "max_batch must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(BMPC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(BMPC_ERR_INVALID, "device %d outside [0, %d)", device, ndev);
  bmpc::DevParams dev;
  int rc = make_dev_params(*params, &dev);
  if (rc != BMPC_OK) return rc;
  bmpc_handle h = new (std::nothrow) bmpc_handle_s();
  if (!h) return fail(BMPC_ERR_ALLOC, "out of host memory");
  h->device = device; h->max_batch = max_batch; h->params = *params; h->dev = dev;
  h->path = resolve_path(params->h, params->path);
  h->rescue_on = resolve_rescue(*params);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  {
    // chunk streams of the host-pointer path: the first chunk on the highest priority the device offers, the last ones on the
    // lowest, so that the workgroup dispatcher serves the chunks in order and the first results leave early
    int least = 0, greatest = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    h->host_timing = std::getenv("BMPC_HOST_TIMING") != nullptr;
    if (h->host_timing) std::fprintf(stderr, "[bmpc host path] stream priorities: least %d greatest %d\n", least, greatest);
    if (const char* ev = std::getenv("BMPC_HOST_CUTS")) {          // (experiments: "0.5,0.8", "0.6" for two chunks, "1" for one)
      double a = 1.0, b = 1.0;
      const int k = std::sscanf(ev, "%lf,%lf", &a, &b);
      if (k == 1 && a >= 1.0) h->host_chunks = 1;
      else if (k == 1 && a > 0.0) { h->host_chunks = 2; h->host_cut[1] = a; h->host_cut[2] = 1.0; }
      else if (k == 2 && a > 0.0 && a <= b && b <= 1.0) { h->host_cut[1] = a; h->host_cut[2] = b; }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->cev_own, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->cev_in, hipEventDisableTiming);
    for (int c = 0; c < bmpc_handle_s::HOST_CHUNKS && e == hipSuccess; ++c) {
      int prio = greatest + c;                   // (numerically lower = more urgent)
      if (prio > least) prio = least;
      e = hipStreamCreateWithPriority(&h->cstream[c], hipStreamNonBlocking, prio);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&h->cev[c], hipEventDisableTiming);
    }
  }
  if (e != hipSuccess) {
    bmpc_destroy(h);
    return fail(BMPC_ERR_HIP, "bmpc_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return BMPC_OK;
}

int bmpc_destroy(bmpc_handle h) {
  if (!h) return BMPC_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->x_fb.release(); h->foot.release(); h->x_cmd.release(); h->mu.release(); h->controls.release();
  h->states.release(); h->resid.release(); h->contact.release(); h->phase.release(); h->iters.release();
  h->status.release(); h->nfactor.release(); h->dbg.release();
  h->ll_q.release(); h->ll_qd.release(); h->ll_pf.release(); h->ll_u0.release(); h->ll_tau.release();
  h->ll_t.release(); h->ll_c0.release();
  h->warm.release(); h->ro_controls.release(); h->ro_states.release(); h->ro_contact.release();
  h->ro_phase.release(); h->ro_iters.release(); h->ro_status.release(); h->ro_order.release();
  for (int c = 0; c < bmpc_handle_s::HOST_CHUNKS; ++c) {
    if (h->cstream[c]) { (void)hipStreamSynchronize(h->cstream[c]); (void)hipStreamDestroy(h->cstream[c]); }
    if (h->cev[c]) (void)hipEventDestroy(h->cev[c]);
  }
  if (h->cev_own) (void)hipEventDestroy(h->cev_own);
  if (h->cev_in) (void)hipEventDestroy(h->cev_in);
  h->io_states.release();
  h->pin_in.release(); h->pin_out.release(); h->dev_in.release(); h->dev_out.release();
  h->io_in.release(); h->io_out.release(); h->io_dev.release();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BMPC_OK;
}

int bmpc_set_params(bmpc_handle h, const bmpc_params* params) {
  if (!h || !params) return fail(BMPC_ERR_INVALID, "null argument");
  if (params->h != h->params.h) return fail(BMPC_ERR_INVALID, "horizon is fixed at creation (h=%d)", h->params.h);
  bmpc::DevParams dev;
  int rc = make_dev_params(*params, &dev);
  if (rc != BMPC_OK) return rc;
  if (resolve_path(params->h, params->path) != h->path) h->warm_valid = false;   // the two families keep different state
  h->params = *params; h->dev = dev;
  h->path = resolve_path(params->h, params->path);
  h->rescue_on = resolve_rescue(*params);
  return BMPC_OK;
}

int bmpc_get_params(bmpc_handle h, bmpc_params* out) {
  if (!h || !out) return fail(BMPC_ERR_INVALID, "null argument");
  *out = h->params;
  return BMPC_OK;
}

// one solve launch with the dispatch order given explicitly (roll-outs use their own, the handle's stays untouched)
// (ev: which of the handle's timing events this launch records -- bit 0: ev0 before it, bit 1: ev1 after it; a batch that goes
//  out in chunks records ev0 before its first kernel and ev1 after its last, so bmpc_last_kernel_ms spans them all.
//  c64 / s64: fp64 output arrays instead of controls / states, see bmpc::WarmArgs)
static int solve_device_ordered(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                                const int32_t* phase, const float* x_cmd, const float* mu, float* controls,
                                float* states, int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor,
                                void* stream, const int32_t* order, int ev = 3, double* c64 = nullptr, double* s64 = nullptr) {
  int rc = check_common(h, B, x_fb, foot, contact, phase, c64 ? static_cast<const void*>(c64) : static_cast<const void*>(controls));
  if (rc != BMPC_OK) return rc;
  if (B == 0) return BMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = pick_stream(h, stream);
  bmpc::DebugOut dbg = {nullptr, nullptr, nullptr, nullptr, h->prof_dev, 0};
  if (ev & 1) HIP_TRY(hipEventRecord(h->ev0, st));
  rc = launch(h, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, residuals, status, nfactor, dbg, st, order, c64, s64);
  if (rc != BMPC_OK) return rc;
  if (ev & 2) { HIP_TRY(hipEventRecord(h->ev1, st)); h->timed = true; }
  return BMPC_OK;
}

int bmpc_solve_batch_device(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                            const int32_t* phase, const float* x_cmd, const float* mu, float* controls,
                            float* states, int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor,
                            void* stream) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  return solve_device_ordered(h, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, residuals, status, nfactor,
                              stream, h->order);
}

// Host pointers in, host pointers out (T = float: bmpc_solve_batch; T = double: bmpc_solve_batch_f64, the dtype the reference
// returns).  What the reference's callers get (REF:487), so PCIe is part of the path:
//   * per chunk the inputs are packed into one pinned block and cross in ONE copy; outputs come back one packed block;
//   * the batch is split into up to HOST_CHUNKS contiguous chunks, each launched on its own stream (descending priority:
//     the dispatcher serves chunk 0's workgroups first), followed on that stream by the chunk's device-to-host copy: the
//     results of chunk c cross PCIe, and are unpacked (widened to fp64) into the caller's pageable arrays by the calling thread,
//     while chunks c + 1 .. still solve.  Only the last chunk's copy and unpacking are exposed.
//     (A caller that can take its results in the handle's own page-locked block has no unpacking at all: bmpc_solve_batch_io.)
// The kernels' arithmetic does not depend on the position in a batch, so the results are bit-identical to a single launch.
// Warm start, a dispatch order and the profile buffer index by the instance's position in the whole batch: with any of them
// set the batch goes out as one chunk.
// Ordering: the chunk streams wait for what was queued on the handle's own stream before the call (a device-pointer solve on
// BMPC_STREAM_OWN, its warm-start state), and the call returns with every chunk complete.
}  // extern "C"  (a template cannot have C linkage)

namespace {

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

template <typename T>
int solve_host(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact, const int32_t* phase,
               const float* x_cmd, const float* mu, T* controls, T* states, int32_t* iters, float* residuals, int32_t* status,
               int32_t* nfactor) {
  int rc = check_common(h, B, x_fb, foot, contact, phase, controls);
  if (rc != BMPC_OK) return rc;
  if (B == 0) return BMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, H = (size_t)h->dev.h;
  const bool timing = h->host_timing;                                 // (diagnostics: where a host-pointer call spends its time)
  auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = timing ? now() : 0.0;
  double t_wait = 0, t_unpack = 0, t_last_wait = 0, t_last_unpack = 0;
  // ---- chunks: contiguous, one per stream priority the device offers (MI355X: three; chunks that share a priority are
  // served round robin and finish together), the later ones smaller: the last chunk's unpacking is the exposed part,
  // and a chunk's unpacking (~1/3 of its solve time) has to fit before the next chunk arrives.
  const bool whole = h->warm_on || h->order || h->prof_dev;
  int nchunk = whole ? 1 : (int)(n / 512);
  nchunk = nchunk < 1 ? 1 : (nchunk > h->host_chunks ? h->host_chunks : nchunk);
  const double* kCut4 = h->host_cut;
  // per-instance bytes of a chunk's packed blocks.  in: x_fb, foot, phase [, x_cmd] [, mu], contact;
  // out: controls [, states], iters, status, nfactor, residuals (every sub-array starts 16-byte aligned)
  const size_t in_per = (12 + 6 + 1 + (x_cmd ? 12 : 0) + (mu ? 2 * H : 0)) * 4 + 2 * H;
  const size_t out_per = (H * 12 + (states ? H * 13 : 0)) * 4 + 3 * 4 + 2 * 4;
  const size_t in_bytes = n * in_per + (size_t)nchunk * 6 * 16, out_bytes = n * out_per + (size_t)nchunk * 6 * 16;
  HIP_TRY(h->pin_in.ensure(in_bytes));
  HIP_TRY(h->dev_in.ensure(in_bytes));
  HIP_TRY(h->pin_out.ensure(out_bytes));
  HIP_TRY(h->dev_out.ensure(out_bytes));
  // what the handle's own stream holds (a device-pointer solve on BMPC_STREAM_OWN, the warm-start state it writes) comes first;
  // an idle stream -- the usual case -- is not made to process a marker the chunk streams would then wait for (tens of us)
  const bool own_busy = hipStreamQuery(h->stream) != hipSuccess;
  if (own_busy) HIP_TRY(hipEventRecord(h->cev_own, h->stream));
  struct Chunk { size_t lo, nb, off, o_u, o_s, o_it, o_st, o_nf, o_rs, bytes; } ck[bmpc_handle_s::HOST_CHUNKS];
  size_t off = 0, ioff = 0;
  int issued = 0;                               // chunks whose work is queued (an error below waits for them before returning)
  auto bail = [&](int code) {
    for (int c = 0; c < issued; ++c) (void)hipStreamSynchronize(h->cstream[c]);   // their copies still read pin_in / write pin_out
    return code;
  };
#define HOST_TRY(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return bail(fail(BMPC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)));       \
  } while (0)
  for (int c = 0; c < nchunk; ++c) {
    Chunk& k = ck[c];
    k.lo = nchunk >= 2 ? (size_t)(n * kCut4[c]) : 0;
    k.nb = (nchunk >= 2 ? (size_t)(n * kCut4[c + 1]) : n) - k.lo;
    if (c == nchunk - 1) k.nb = n - k.lo;
    // inputs of this chunk: packed into the pinned block and sent in ONE copy (chunk 0 is on its way while the later
    // chunks are still being packed)
    const size_t i_xfb = 0, i_foot = align16(i_xfb + k.nb * 12 * 4), i_phase = align16(i_foot + k.nb * 6 * 4),
                 i_xcmd = align16(i_phase + k.nb * 4), i_mu = align16(i_xcmd + (x_cmd ? k.nb * 12 * 4 : 0)),
                 i_con = align16(i_mu + (mu ? k.nb * H * 2 * 4 : 0)), i_bytes = align16(i_con + k.nb * H * 2);
    char* pin = h->pin_in.p + ioff;
    char* din = h->dev_in.p + ioff;
    ioff += i_bytes;
    std::memcpy(pin + i_xfb, x_fb + k.lo * 12, k.nb * 12 * 4);
    std::memcpy(pin + i_foot, foot + k.lo * 6, k.nb * 6 * 4);
    std::memcpy(pin + i_phase, phase + k.lo, k.nb * 4);
    if (x_cmd) std::memcpy(pin + i_xcmd, x_cmd + k.lo * 12, k.nb * 12 * 4);
    if (mu) std::memcpy(pin + i_mu, mu + k.lo * H * 2, k.nb * H * 2 * 4);
    std::memcpy(pin + i_con, contact + k.lo * H * 2, k.nb * H * 2);
    k.off = off;
    k.o_u = 0;
    k.o_s = align16(k.o_u + k.nb * H * 12 * 4);
    k.o_it = align16(k.o_s + (states ? k.nb * H * 13 * 4 : 0));
    k.o_st = align16(k.o_it + k.nb * 4);
    k.o_nf = align16(k.o_st + k.nb * 4);
    k.o_rs = align16(k.o_nf + k.nb * 4);
    k.bytes = align16(k.o_rs + k.nb * 2 * 4);
    off += k.bytes;
    hipStream_t st = h->cstream[c];
    if (own_busy) HOST_TRY(hipStreamWaitEvent(st, h->cev_own, 0));
    issued = c + 1;
    HOST_TRY(hipMemcpyAsync(din, pin, i_bytes, hipMemcpyHostToDevice, st));
    char* dout = h->dev_out.p + k.off;
    rc = solve_device_ordered(h, (int)k.nb, reinterpret_cast<const float*>(din + i_xfb), reinterpret_cast<const float*>(din + i_foot),
                              reinterpret_cast<const uint8_t*>(din + i_con), reinterpret_cast<const int32_t*>(din + i_phase),
                              x_cmd ? reinterpret_cast<const float*>(din + i_xcmd) : nullptr,
                              mu ? reinterpret_cast<const float*>(din + i_mu) : nullptr,
                              reinterpret_cast<float*>(dout + k.o_u), states ? reinterpret_cast<float*>(dout + k.o_s) : nullptr,
                              reinterpret_cast<int32_t*>(dout + k.o_it), reinterpret_cast<float*>(dout + k.o_rs),
                              reinterpret_cast<int32_t*>(dout + k.o_st), reinterpret_cast<int32_t*>(dout + k.o_nf), st, h->order,
                              (c == 0 ? 1 : 0) | (c == nchunk - 1 ? 2 : 0));
    if (rc != BMPC_OK) return bail(rc);
    // (the copy engine moves a chunk at ~50 GB/s while the later chunks solve; stores of the kernels themselves into mapped host
    //  memory sustain ~8.6 GB/s on MI355X -- measured, round 5 -- which a 4096-instance batch's 4 MB would just fit under, with
    //  nothing to spare)
    HOST_TRY(hipMemcpyAsync(h->pin_out.p + k.off, dout, k.bytes, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipEventRecord(h->cev[c], st));
  }
  const double t_issued = timing ? now() : 0.0;
  // ---- unpack chunk by chunk, as each arrives
  for (int c = 0; c < nchunk; ++c) {
    const Chunk& k = ck[c];
    const double tw0 = timing ? now() : 0.0;
    HOST_TRY(hipEventSynchronize(h->cev[c]));
    const double tw1 = timing ? now() : 0.0;
    t_wait += tw1 - tw0; t_last_wait = tw1 - tw0;
    if (timing) std::fprintf(stderr, "[bmpc host path]   chunk %d ready %.0f us after the call began (waited %.0f)\n", c, tw1 - t_begin, tw1 - tw0);
    const char* src = h->pin_out.p + k.off;
    auto put = [&](T* dst, const float* from, size_t cnt) {
      if constexpr (sizeof(T) == sizeof(float)) std::memcpy(dst, from, cnt * sizeof(float));
      else for (size_t q = 0; q < cnt; ++q) dst[q] = (T)from[q];
    };
    put(controls + k.lo * H * 12, reinterpret_cast<const float*>(src + k.o_u), k.nb * H * 12);
    if (states) put(states + k.lo * H * 13, reinterpret_cast<const float*>(src + k.o_s), k.nb * H * 13);
    if (iters) std::memcpy(iters + k.lo, src + k.o_it, k.nb * 4);
    if (status) std::memcpy(status + k.lo, src + k.o_st, k.nb * 4);
    if (nfactor) std::memcpy(nfactor + k.lo, src + k.o_nf, k.nb * 4);
    if (residuals) std::memcpy(residuals + k.lo * 2, src + k.o_rs, k.nb * 2 * 4);
    if (timing) { const double tu = now() - tw1; t_unpack += tu; t_last_unpack = tu; }
  }
#undef HOST_TRY
  if (timing)
    std::fprintf(stderr, "[bmpc host path] B %d chunks %d: pack + issue %.0f us, waiting %.0f us (last chunk %.0f), unpack %.0f us (last chunk %.0f), total %.0f us\n",
                 B, nchunk, t_issued - t_begin, t_wait, t_last_wait, t_unpack, t_last_unpack, now() - t_begin);
  return BMPC_OK;
}

// layout of the handle's I/O block for a batch (bmpc_host_io): inputs | outputs, every array 64-byte aligned
inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

}  // namespace

extern "C" {

int bmpc_host_io(bmpc_handle h, int B, int with_x_cmd, int with_mu, int with_states, bmpc_host_views* out) {
  if (!h || !out) return fail(BMPC_ERR_INVALID, "null argument");
  if (B < 1 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [1, max_batch=%d]", B, h->max_batch);
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, H = (size_t)h->dev.h;
  bmpc_handle_s::IoLayout& L = h->io;
  L.B = 0;
  h->io_gen = h->io_gen == 0x7fffffff ? 1 : h->io_gen + 1;
  L.i_xfb = 0;
  L.i_foot = align64(L.i_xfb + n * 12 * 4);
  L.i_phase = align64(L.i_foot + n * 6 * 4);
  L.i_xcmd = align64(L.i_phase + n * 4);
  L.i_mu = align64(L.i_xcmd + (with_x_cmd ? n * 12 * 4 : 0));
  L.i_con = align64(L.i_mu + (with_mu ? n * H * 2 * 4 : 0));
  L.in_bytes = align64(L.i_con + n * H * 2);
  L.o_u = 0;
  L.o_s = align64(L.o_u + n * H * 12 * 8);
  L.o_it = align64(L.o_s + (with_states ? n * H * 13 * 8 : 0));
  L.o_st = align64(L.o_it + n * 4);
  L.o_nf = align64(L.o_st + n * 4);
  L.o_rs = align64(L.o_nf + n * 4);
  L.out_bytes = align64(L.o_rs + n * 2 * 4);
  // (the handle's own stream may still read the old block: a re-allocation waits for it)
  if (L.in_bytes > h->io_in.n || L.out_bytes > h->io_out.n) HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(h->io_in.ensure(L.in_bytes));
  HIP_TRY(h->io_out.ensure(L.out_bytes));
  HIP_TRY(h->io_dev.ensure(L.in_bytes));
  L.B = B; L.x_cmd = with_x_cmd != 0; L.mu = with_mu != 0; L.states = with_states != 0;
  char* in = h->io_in.p;
  char* o = h->io_out.p;
  out->x_fb = reinterpret_cast<float*>(in + L.i_xfb);
  out->foot = reinterpret_cast<float*>(in + L.i_foot);
  out->phase = reinterpret_cast<int32_t*>(in + L.i_phase);
  out->x_cmd = with_x_cmd ? reinterpret_cast<float*>(in + L.i_xcmd) : nullptr;
  out->mu = with_mu ? reinterpret_cast<float*>(in + L.i_mu) : nullptr;
  out->contact = reinterpret_cast<uint8_t*>(in + L.i_con);
  out->controls = reinterpret_cast<double*>(o + L.o_u);
  out->states = with_states ? reinterpret_cast<double*>(o + L.o_s) : nullptr;
  out->iters = reinterpret_cast<int32_t*>(o + L.o_it);
  out->status = reinterpret_cast<int32_t*>(o + L.o_st);
  out->nfactor = reinterpret_cast<int32_t*>(o + L.o_nf);
  out->residuals = reinterpret_cast<float*>(o + L.o_rs);
  return BMPC_OK;
}

int bmpc_host_io_generation(bmpc_handle h) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  return h->io_gen;
}

int bmpc_solve_batch_io(bmpc_handle h, int B) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  const bmpc_handle_s::IoLayout& L = h->io;
  if (B < 1 || B != L.B) return fail(BMPC_ERR_INVALID, "bmpc_solve_batch_io: batch %d, but the I/O block is laid out for %d (bmpc_host_io)", B, L.B);
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, H = (size_t)h->dev.h;
  char* din = h->io_dev.p;
  char* o = h->io_out.p;
  // How the results cross PCIe (MI355X, measured in round 5): stores of a kernel into mapped host memory sustain ~8.6 GB/s, the
  // copy engine ~50 GB/s.  A 4096-instance batch returns 3.9 MB of fp64 controls and 4.3 MB of fp64 states in 0.8 ms: the
  // controls (and the per-instance counters) go straight from the kernels' epilogues into the host arrays -- free below that
  // rate --, the states are stored in HBM and follow by copy engine, chunk by chunk on the chunk streams of the host-pointer
  // path, so that only the last chunk's copy (15 % of the states) is exposed.  Everything through the kernels: +0.12 ms.
  const bool whole = h->warm_on || h->order || h->prof_dev || !L.states;     // (one chunk: everything straight from the kernel)
  int nchunk = whole ? 1 : (int)(n / 512);
  nchunk = nchunk < 1 ? 1 : (nchunk > h->host_chunks ? h->host_chunks : nchunk);
  if (L.states) HIP_TRY(h->io_states.ensure(n * H * 13));
  const bool own_busy = hipStreamQuery(h->stream) != hipSuccess;      // what the handle's own stream holds comes first
  if (own_busy) HIP_TRY(hipEventRecord(h->cev_own, h->stream));
  int issued = 0;
  auto bail = [&](int code) {
    for (int c = 0; c < issued; ++c) (void)hipStreamSynchronize(h->cstream[c]);
    return code;
  };
#define IO_TRY(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return bail(fail(BMPC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)));       \
  } while (0)
  for (int c = 0; c < nchunk; ++c) {
    const size_t lo = nchunk >= 2 ? (size_t)(n * h->host_cut[c]) : 0;
    const size_t nb = (c == nchunk - 1 ? n : (size_t)(n * h->host_cut[c + 1])) - lo;
    hipStream_t st = h->cstream[c];
    if (own_busy) IO_TRY(hipStreamWaitEvent(st, h->cev_own, 0));
    issued = c + 1;
    if (c == 0) {                               // ONE copy in, for the whole batch
      IO_TRY(hipMemcpyAsync(din, h->io_in.p, L.in_bytes, hipMemcpyHostToDevice, st));
      IO_TRY(hipEventRecord(h->cev_in, st));
    } else {
      IO_TRY(hipStreamWaitEvent(st, h->cev_in, 0));
    }
    // (the last chunk's states go the way of the controls: nothing is left to copy when its kernel ends, and 15 % of the
    //  states on top of the controls stay well below what the kernels' own stores sustain)
    const bool by_copy = L.states && c < nchunk - 1;
    double* s64 = !L.states ? nullptr : (by_copy ? h->io_states.p + lo * H * 13 : reinterpret_cast<double*>(o + L.o_s) + lo * H * 13);
    int rc = solve_device_ordered(h, (int)nb, reinterpret_cast<const float*>(din + L.i_xfb) + lo * 12, reinterpret_cast<const float*>(din + L.i_foot) + lo * 6,
                                  reinterpret_cast<const uint8_t*>(din + L.i_con) + lo * H * 2, reinterpret_cast<const int32_t*>(din + L.i_phase) + lo,
                                  L.x_cmd ? reinterpret_cast<const float*>(din + L.i_xcmd) + lo * 12 : nullptr,
                                  L.mu ? reinterpret_cast<const float*>(din + L.i_mu) + lo * H * 2 : nullptr, nullptr, nullptr,
                                  reinterpret_cast<int32_t*>(o + L.o_it) + lo, reinterpret_cast<float*>(o + L.o_rs) + lo * 2,
                                  reinterpret_cast<int32_t*>(o + L.o_st) + lo, reinterpret_cast<int32_t*>(o + L.o_nf) + lo, st, h->order,
                                  (c == 0 ? 1 : 0) | (c == nchunk - 1 ? 2 : 0),
                                  reinterpret_cast<double*>(o + L.o_u) + lo * H * 12, s64);
    if (rc != BMPC_OK) return bail(rc);
    if (by_copy) IO_TRY(hipMemcpyAsync(reinterpret_cast<double*>(o + L.o_s) + lo * H * 13, s64, nb * H * 13 * sizeof(double), hipMemcpyDeviceToHost, st));
    IO_TRY(hipEventRecord(h->cev[c], st));
  }
  for (int c = 0; c < nchunk; ++c) IO_TRY(hipEventSynchronize(h->cev[c]));
#undef IO_TRY
  return BMPC_OK;
}

int bmpc_solve_batch(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                     const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
                     int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor) {
  return solve_host<float>(h, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, residuals, status, nfactor);
}

int bmpc_solve_batch_f64(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                         const int32_t* phase, const float* x_cmd, const float* mu, double* controls, double* states,
                         int32_t* iters, float* residuals, int32_t* status, int32_t* nfactor) {
  return solve_host<double>(h, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, residuals, status, nfactor);
}

int bmpc_synchronize(bmpc_handle h) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int c = 0; c < bmpc_handle_s::HOST_CHUNKS; ++c)
    if (h->cstream[c]) HIP_TRY(hipStreamSynchronize(h->cstream[c]));
  if (h->timed) HIP_TRY(hipEventSynchronize(h->ev1));     // the last solve launch, whatever stream it was given
  return BMPC_OK;
}

int bmpc_debug_assemble(bmpc_handle h, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                        const int32_t* phase, const float* x_cmd, const float* mu, double* x_ref, double* foot_ref,
                        double* Gt, double* qt) {
  float dummy = 0;
  int rc = check_common(h, B, x_fb, foot, contact, phase, &dummy);
  if (rc != BMPC_OK) return rc;
  if (B == 0) return BMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, H = (size_t)h->dev.h, NW = 6 * H;
  HIP_TRY(h->x_fb.ensure(n * 12)); HIP_TRY(h->foot.ensure(n * 6)); HIP_TRY(h->contact.ensure(n * H * 2));
  HIP_TRY(h->phase.ensure(n)); HIP_TRY(h->controls.ensure(n * H * 12));
  if (x_cmd) HIP_TRY(h->x_cmd.ensure(n * 12));
  if (mu) HIP_TRY(h->mu.ensure(n * H * 2));
  // only what the caller asked for is formed: Gt alone is n (6h)^2 doubles (1.9 GB at B = 4096, h = 40), and the Gt / qt
  // views exist on the dense family only (h <= 20) -- a caller that wants the references gets them at every horizon
  if ((Gt || qt) && !dense_horizon(h->dev.h))
    return fail(BMPC_ERR_INVALID, "Gt / qt views exist for h <= 20 only (h=%d never forms them); pass NULL for both", h->dev.h);
  const size_t o_xr = 0, o_fr = o_xr + n * H * 12, o_gt = o_fr + n * H * 6, o_qt = o_gt + (Gt ? n * NW * NW : 0),
               tot = o_qt + (qt ? n * NW : 0);
  HIP_TRY(h->dbg.ensure(tot));
  hipStream_t st = h->stream;
  HIP_TRY(hipMemsetAsync(h->dbg.p, 0, tot * sizeof(double), st));
  HIP_TRY(hipMemcpyAsync(h->x_fb.p, x_fb, n * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->foot.p, foot, n * 6 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->contact.p, contact, n * H * 2, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->phase.p, phase, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (x_cmd) HIP_TRY(hipMemcpyAsync(h->x_cmd.p, x_cmd, n * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  if (mu) HIP_TRY(hipMemcpyAsync(h->mu.p, mu, n * H * 2 * sizeof(float), hipMemcpyHostToDevice, st));
  bmpc::DebugOut dbg = {h->dbg.p + o_xr, h->dbg.p + o_fr, Gt ? h->dbg.p + o_gt : nullptr, qt ? h->dbg.p + o_qt : nullptr, nullptr, 1};
  rc = launch(h, B, h->x_fb.p, h->foot.p, h->contact.p, h->phase.p, x_cmd ? h->x_cmd.p : nullptr,
              mu ? h->mu.p : nullptr, h->controls.p, nullptr, nullptr, nullptr, nullptr, nullptr, dbg, st, nullptr);
  if (rc != BMPC_OK) return rc;
  if (x_ref) HIP_TRY(hipMemcpyAsync(x_ref, h->dbg.p + o_xr, n * H * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (foot_ref) HIP_TRY(hipMemcpyAsync(foot_ref, h->dbg.p + o_fr, n * H * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (Gt) HIP_TRY(hipMemcpyAsync(Gt, h->dbg.p + o_gt, n * NW * NW * sizeof(double), hipMemcpyDeviceToHost, st));
  if (qt) HIP_TRY(hipMemcpyAsync(qt, h->dbg.p + o_qt, n * NW * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return BMPC_OK;
}

static bmpc::LowLevelParams ll_params(const bmpc_params& p) {
  bmpc::LowLevelParams q;
  q.h = p.h; q.dt = p.dt; q.kv = p.kv; q.swing_height = p.swingHeight;
  for (int i = 0; i < 12; ++i) q.x_cmd[i] = p.x_cmd[i];
  for (int i = 0; i < 9; ++i) { q.kp[i] = p.kp[i]; q.kd[i] = p.kd[i]; }
  for (int i = 0; i < 3; ++i) q.hip_offset[i] = p.hip_offset[i];
  return q;
}

int bmpc_foot_position_world_device(bmpc_handle h, int B, const float* x_fb, const float* q, float* pf_w, void* stream) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!x_fb || !q || !pf_w) return fail(BMPC_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = pick_stream(h, stream);
  hipLaunchKernelGGL(bmpc::foot_world_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ll_params(h->params), B, x_fb, q, pf_w);
  HIP_TRY(hipGetLastError());
  return BMPC_OK;
}

int bmpc_foot_position_world(bmpc_handle h, int B, const float* x_fb, const float* q, float* pf_w) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!x_fb || !q || !pf_w) return fail(BMPC_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B;
  HIP_TRY(h->x_fb.ensure(n * 12)); HIP_TRY(h->ll_q.ensure(n * 10)); HIP_TRY(h->ll_pf.ensure(n * 6));
  hipStream_t st = h->stream;
  HIP_TRY(hipMemcpyAsync(h->x_fb.p, x_fb, n * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_q.p, q, n * 10 * sizeof(float), hipMemcpyHostToDevice, st));
  int rc = bmpc_foot_position_world_device(h, B, h->x_fb.p, h->ll_q.p, h->ll_pf.p, st);
  if (rc != BMPC_OK) return rc;
  HIP_TRY(hipMemcpyAsync(pf_w, h->ll_pf.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return BMPC_OK;
}

int bmpc_low_level_control_device(bmpc_handle h, int B, const float* x_fb, const double* t, const float* pf_w,
                                  const float* q, const float* qd, const uint8_t* contact0, const float* u0,
                                  float* tau, void* stream) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!x_fb || !t || !pf_w || !q || !qd || !contact0 || !u0 || !tau) return fail(BMPC_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = pick_stream(h, stream);
  hipLaunchKernelGGL(bmpc::lowlevel_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ll_params(h->params), B, x_fb, t,
                     pf_w, q, qd, contact0, u0, tau);
  HIP_TRY(hipGetLastError());
  return BMPC_OK;
}

int bmpc_low_level_control(bmpc_handle h, int B, const float* x_fb, const double* t, const float* pf_w,
                           const float* q, const float* qd, const uint8_t* contact0, const float* u0, float* tau) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!x_fb || !t || !pf_w || !q || !qd || !contact0 || !u0 || !tau) return fail(BMPC_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B;
  HIP_TRY(h->x_fb.ensure(n * 12)); HIP_TRY(h->ll_t.ensure(n)); HIP_TRY(h->ll_pf.ensure(n * 6));
  HIP_TRY(h->ll_q.ensure(n * 10)); HIP_TRY(h->ll_qd.ensure(n * 10)); HIP_TRY(h->ll_c0.ensure(n * 2));
  HIP_TRY(h->ll_u0.ensure(n * 12)); HIP_TRY(h->ll_tau.ensure(n * 10));
  hipStream_t st = h->stream;
  HIP_TRY(hipMemcpyAsync(h->x_fb.p, x_fb, n * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_t.p, t, n * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_pf.p, pf_w, n * 6 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_q.p, q, n * 10 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_qd.p, qd, n * 10 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_c0.p, contact0, n * 2, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->ll_u0.p, u0, n * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  int rc = bmpc_low_level_control_device(h, B, h->x_fb.p, h->ll_t.p, h->ll_pf.p, h->ll_q.p, h->ll_qd.p, h->ll_c0.p,
                                         h->ll_u0.p, h->ll_tau.p, st);
  if (rc != BMPC_OK) return rc;
  HIP_TRY(hipMemcpyAsync(tau, h->ll_tau.p, n * 10 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return BMPC_OK;
}

int bmpc_gait_default(bmpc_gait* g, int half) {
  if (!g) return fail(BMPC_ERR_INVALID, "null gait");
  if (half < 1) return fail(BMPC_ERR_INVALID, "half period must be >= 1");
  g->period = 2 * half;
  g->offset[0] = 0; g->offset[1] = half;      // REF:52-55: leg 1 stands first, leg 2 half a period later
  g->duty[0] = half; g->duty[1] = half;
  return BMPC_OK;
}

static int gait_params(bmpc_handle h, const bmpc_gait* gait, bmpc::GaitParams* G) {
  bmpc_gait g;
  if (gait) g = *gait;
  else bmpc_gait_default(&g, h->params.half);
  if (g.period < 1) return fail(BMPC_ERR_INVALID, "gait period must be >= 1");
  for (int k = 0; k < 2; ++k)
    if (g.duty[k] < 0 || g.duty[k] > g.period) return fail(BMPC_ERR_INVALID, "gait duty must be in [0, period]");
  G->h = h->params.h; G->period = g.period; G->dt = h->params.dt;
  for (int k = 0; k < 2; ++k) { G->offset[k] = g.offset[k]; G->duty[k] = g.duty[k]; }
  return BMPC_OK;
}

int bmpc_contact_sequence_device(bmpc_handle h, int B, const double* t, const bmpc_gait* gait, int32_t* phase,
                                 uint8_t* contact, void* stream) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!t) return fail(BMPC_ERR_INVALID, "null pointer");
  bmpc::GaitParams G;
  int rc = gait_params(h, gait, &G);
  if (rc != BMPC_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = pick_stream(h, stream);
  hipLaunchKernelGGL(bmpc::gait_kernel, dim3((B + 255) / 256), dim3(256), 0, st, G, B, t, phase, contact);
  HIP_TRY(hipGetLastError());
  return BMPC_OK;
}

int bmpc_contact_sequence(bmpc_handle h, int B, const double* t, const bmpc_gait* gait, int32_t* phase, uint8_t* contact) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (B == 0) return BMPC_OK;
  if (!t) return fail(BMPC_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, hh = (size_t)h->params.h;
  HIP_TRY(h->ll_t.ensure(n)); HIP_TRY(h->phase.ensure(n)); HIP_TRY(h->contact.ensure(n * hh * 2));
  hipStream_t st = h->stream;
  HIP_TRY(hipMemcpyAsync(h->ll_t.p, t, n * sizeof(double), hipMemcpyHostToDevice, st));
  int rc = bmpc_contact_sequence_device(h, B, h->ll_t.p, gait, h->phase.p, h->contact.p, st);
  if (rc != BMPC_OK) return rc;
  if (phase) HIP_TRY(hipMemcpyAsync(phase, h->phase.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (contact) HIP_TRY(hipMemcpyAsync(contact, h->contact.p, n * hh * 2, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return BMPC_OK;
}

int bmpc_set_warm_start(bmpc_handle h, int enable, int shift, double theta) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (shift < 0 || shift >= h->params.h) return fail(BMPC_ERR_INVALID, "shift must be in [0, h)");
  if (!(theta >= 0.0 && theta <= 1.0)) return fail(BMPC_ERR_INVALID, "theta must be in [0, 1]");
  h->warm_on = enable != 0;
  h->warm_shift = shift;
  h->warm_theta = (float)theta;
  if (!h->warm_on) h->warm_valid = false;
  return BMPC_OK;
}

int bmpc_reset_warm_start(bmpc_handle h) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  h->warm_valid = false;
  return BMPC_OK;
}

int bmpc_rollout_device(bmpc_handle h, int B, int steps, float* x_fb, const float* foot, double* t,
                        const bmpc_gait* gait, const float* x_cmd, const float* mu, float* u0_traj, float* x_traj,
                        int32_t* iters_traj, int32_t* status_any, void* stream) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  if (B < 0 || B > h->max_batch) return fail(BMPC_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, h->max_batch);
  if (steps < 0) return fail(BMPC_ERR_INVALID, "steps must be >= 0");
  if (B == 0 || steps == 0) return BMPC_OK;
  if (!x_fb || !foot || !t) return fail(BMPC_ERR_INVALID, "x_fb, foot and t must be non-null");
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)B, H = (size_t)h->dev.h;
  HIP_TRY(h->ro_controls.ensure(n * H * 12)); HIP_TRY(h->ro_states.ensure(n * H * 13));
  HIP_TRY(h->ro_contact.ensure(n * H * 2)); HIP_TRY(h->ro_phase.ensure(n));
  HIP_TRY(h->ro_iters.ensure(n)); HIP_TRY(h->ro_status.ensure(n));
  hipStream_t st = pick_stream(h, stream);
  if (status_any) HIP_TRY(hipMemsetAsync(status_any, 0, n * sizeof(int32_t), st));
  const int32_t* order = h->order;                 // this roll-out's dispatch order (the handle's is left alone)
  const bool own_order = h->longest_first && !order;
  if (own_order) HIP_TRY(h->ro_order.ensure(n));
  int rc = BMPC_OK;
  for (int s = 0; s < steps && rc == BMPC_OK; ++s) {
    // t -> (phase, contact) -> solve -> x_fb <- states[:, 0], t += dt: three launches on one stream, no host arithmetic
    // (plus, from the second period on, the dispatch order: the instances that iterated longest last period go first)
    rc = bmpc_contact_sequence_device(h, B, t, gait, h->ro_phase.p, h->ro_contact.p, st);
    if (rc != BMPC_OK) break;
    if (own_order && s > 0) {
      hipLaunchKernelGGL(bmpc::dispatch_order_kernel, dim3(1), dim3(1024), 0, st, B, h->ro_iters.p, h->ro_order.p);
      if (hipGetLastError() != hipSuccess) { rc = fail(BMPC_ERR_HIP, "dispatch-order launch failed"); break; }
      order = h->ro_order.p;
    }
    rc = solve_device_ordered(h, B, x_fb, foot, h->ro_contact.p, h->ro_phase.p, x_cmd, mu, h->ro_controls.p,
                              h->ro_states.p, h->ro_iters.p, nullptr, h->ro_status.p, nullptr, stream, order);
    if (rc != BMPC_OK) break;
    hipLaunchKernelGGL(bmpc::rollout_feedback_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, (int)H, h->params.dt,
                       h->ro_states.p, h->ro_controls.p, h->ro_iters.p, h->ro_status.p, x_fb, t,
                       u0_traj ? u0_traj + (size_t)s * n * 12 : nullptr, x_traj ? x_traj + (size_t)s * n * 12 : nullptr,
                       iters_traj ? iters_traj + (size_t)s * n : nullptr, status_any);
    if (hipGetLastError() != hipSuccess) rc = fail(BMPC_ERR_HIP, "roll-out launch failed");
  }
  return rc;
}

int bmpc_set_dispatch_order(bmpc_handle h, const int32_t* order_dev, int longest_first_rollouts) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  h->order = order_dev;
  h->longest_first = longest_first_rollouts != 0;
  return BMPC_OK;
}

int bmpc_debug_set_profile(bmpc_handle h, long long* device_buf) {
  if (!h) return fail(BMPC_ERR_INVALID, "null handle");
  h->prof_dev = device_buf;
  return BMPC_OK;
}

int bmpc_last_kernel_ms(bmpc_handle h, float* ms) {
  if (!h || !ms) return fail(BMPC_ERR_INVALID, "null argument");
  *ms = -1.f;
  if (!h->timed) return BMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->ev1));
  HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return BMPC_OK;
}

}  // extern "C"
