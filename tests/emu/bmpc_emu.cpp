// tests/emu/bmpc_emu.cpp -- TEST INFRASTRUCTURE: runs the HIP solve kernel's source on the CPU, one std::thread
// per lane of a workgroup, so that the kernel's logic (thread map, LDS exchanges, barriers, cross-lane swaps)
// can be checked against the oracle without a GPU.  Nothing of the product loads this; it is not a CPU path of
// the library (libbmpc.so has none) and it is orders of magnitude too slow to be one.
//
// How: the kernel file is included as plain C++.  __shared__ becomes a function-local static (one image shared
// by the lane threads; workgroups run one after another), threadIdx / blockIdx are thread-local, __syncthreads
// is a std::barrier over the workgroup, and the cross-lane operations (pair swap, wave maximum) go through a
// shared array between two barriers -- which demands what the GPU code must guarantee anyway: every lane of
// the workgroup reaches every barrier and every cross-lane operation.
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define BMPC_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_idx { int x; };
static thread_local emu_idx threadIdx, blockIdx;
static std::barrier<>* g_bar = nullptr;
static int g_or[2];
// byte the LDS image starts from (all-ones = NaNs; BMPC_EMU_POISON tries other leftovers: a result that changes with
// it reads LDS that nobody wrote)
static int g_poison = 0xFF;
static int g_swap[1024];
static std::atomic<unsigned> g_pair[512];
static std::barrier<>* g_wbar[16] = {};      // one per wave
static unsigned g_red[1024];

static inline void __syncthreads() { g_bar->arrive_and_wait(); }
static inline int __syncthreads_or(int v) {
  // two slots so that back-to-back calls cannot race on the reset
  static thread_local int phase = 0;
  int* slot = &g_or[phase & 1];
  __syncthreads();
  if (v) __atomic_store_n(slot, 1, __ATOMIC_RELAXED);
  __syncthreads();
  const int r = __atomic_load_n(slot, __ATOMIC_RELAXED);
  __syncthreads();
  if (threadIdx.x == 0) *slot = 0;
  ++phase;
  return r;
}
static inline long long clock64() { return 0; }
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __double2loint(double d) { long long i; std::memcpy(&i, &d, 8); return (int)(i & 0xffffffffLL); }
static inline int __double2hiint(double d) { long long i; std::memcpy(&i, &d, 8); return (int)(i >> 32); }
static inline double __hiloint2double(int hi, int lo) {
  const long long i = ((long long)hi << 32) | (unsigned)lo;
  double d; std::memcpy(&d, &i, 8); return d;
}
using std::fma; using std::fmin; using std::fmax; using std::fabs;

namespace bmpc {
static inline double rcp_approx(double x) { return 1.0 / x; }
static inline float rcp_approx(float x) { return 1.0f / x; }
static inline float rsq_approx(float x) { return 1.0f / std::sqrt(x); }
static inline void sync_workgroup() { __syncthreads(); }
static inline int sync_workgroup_or(int v) { return __syncthreads_or(v); }
// The cross-lane operations synchronise only the lanes that take part (the pair, the wave), as on the GPU, where a
// DPP exchange is no barrier: an LDS hand-over that relied on one would be a race there, and is one here (visible
// to ThreadSanitizer: tests/emu/tsan.sh).
static inline void pair_sync() {
  // two-party barrier of lanes (l, l ^ 1): the counter goes 2 k -> 2 k + 2 per rendezvous
  std::atomic<unsigned>& cnt = g_pair[threadIdx.x >> 1];
  const unsigned old = cnt.fetch_add(1, std::memory_order_acq_rel);
  const unsigned target = (old | 1u) + 1u;
  while (cnt.load(std::memory_order_acquire) < target) std::this_thread::yield();
}
static inline int pair_swap_i(int v) {
  g_swap[threadIdx.x] = v;
  pair_sync();
  const int r = g_swap[threadIdx.x ^ 1];
  pair_sync();
  return r;
}
static inline unsigned wave_umax(unsigned v) {          // maximum over the lane's wave (64 consecutive lanes)
  std::barrier<>& wb = *g_wbar[threadIdx.x >> 6];
  g_red[threadIdx.x] = v;
  wb.arrive_and_wait();
  unsigned m = 0;
  const int w0 = threadIdx.x & ~63;
  for (int i = 0; i < 64; ++i) m = g_red[w0 + i] > m ? g_red[w0 + i] : m;
  wb.arrive_and_wait();
  return m;
}
static inline unsigned row0_umax(unsigned v) {          // maximum over lanes 0 .. 15 of the lane's wave, to every lane of it
  std::barrier<>& wb = *g_wbar[threadIdx.x >> 6];
  g_red[threadIdx.x] = v;
  wb.arrive_and_wait();
  unsigned m = 0;
  const int w0 = threadIdx.x & ~63;
  for (int i = 0; i < 16; ++i) m = g_red[w0 + i] > m ? g_red[w0 + i] : m;
  wb.arrive_and_wait();
  return m;
}
// sum over the lane's wave in the order of the GPU's DPP tree (bmpc_kernels.hip wave_sum): inclusive scan inside rows of
// 16 by shifts 1, 2, 4, 8 (zero where the source lane is outside the row), then row 1 += lane 15, rows 2, 3 += lane 31
static float g_redf[1024];
static inline float wave_sum(float v) {
  std::barrier<>& wb = *g_wbar[threadIdx.x >> 6];
  const int w0 = threadIdx.x & ~63, ln = threadIdx.x & 63;
  for (int sh = 1; sh <= 8; sh <<= 1) {
    g_redf[threadIdx.x] = v;
    wb.arrive_and_wait();
    const float o = ((ln & 15) >= sh) ? g_redf[threadIdx.x - sh] : 0.f;
    wb.arrive_and_wait();
    v += o;
  }
  g_redf[threadIdx.x] = v;
  wb.arrive_and_wait();
  const float b15 = ln >= 16 ? g_redf[w0 + ((ln >> 4) - 1) * 16 + 15] : 0.f;     // row_bcast:15, every row enabled
  wb.arrive_and_wait();
  v += b15;
  g_redf[threadIdx.x] = v;
  wb.arrive_and_wait();
  const float b31 = ln >= 32 ? g_redf[w0 + 31] : 0.f;                            // row_bcast:31
  wb.arrive_and_wait();
  v += b31;
  g_redf[threadIdx.x] = v;
  wb.arrive_and_wait();
  const float r = g_redf[w0 + 63];
  wb.arrive_and_wait();
  return r;
}
}  // namespace bmpc
#define BMPC_WAVE_SYNC() g_wbar[threadIdx.x >> 6]->arrive_and_wait()
#define BMPC_DRAIN_LDS() do { } while (0)
#define BMPC_FENCE() do { } while (0)
#define BMPC_OPAQUE(x) do { } while (0)
#define BMPC_UNIFORM(x) (x)
#define BMPC_UNIFORM_INT(x) (x)
#define BMPC_SCHED_BARRIER() do { } while (0)

static float g_bc[1024];
namespace bmpc {
// value of lane N of the own row of 16 lanes (DPP row_newbcast on the GPU); all lanes of the wave call
template <int N>
static inline float row_bcast(float v) {
  std::barrier<>& wb = *g_wbar[threadIdx.x >> 6];
  g_bc[threadIdx.x] = v;
  wb.arrive_and_wait();
  const float r = g_bc[(threadIdx.x & ~15) + N];
  wb.arrive_and_wait();
  return r;
}
}  // namespace bmpc

namespace bmpc {
// v_mfma_f64_16x16x4_f64 on the CPU: every lane of the wave hands over its A and B entry (A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15]) and accumulates its 4 entries of D (row = (l >> 4) + 4 reg, col = l & 15)
typedef double emu_f64x4 __attribute__((ext_vector_type(4)));
static double g_mfa[1024], g_mfb[1024];
static inline emu_f64x4 mfma_f64_16x16x4(double a, double b, emu_f64x4 c) {
  std::barrier<>& wb = *g_wbar[threadIdx.x >> 6];
  const int w0 = threadIdx.x & ~63, ln = threadIdx.x & 63;
  g_mfa[threadIdx.x] = a;
  g_mfb[threadIdx.x] = b;
  wb.arrive_and_wait();
  for (int v = 0; v < 4; ++v) {
    const int row = (ln >> 4) + 4 * v, col = ln & 15;
    double acc = c[v];
    for (int k = 0; k < 4; ++k) acc = std::fma(g_mfa[w0 + 16 * k + row], g_mfb[w0 + 16 * k + col], acc);
    c[v] = acc;
  }
  wb.arrive_and_wait();
  return c;
}
}  // namespace bmpc

#include "../../biped_mpc_py_amd/csrc/bmpc_kernels.hip"
#include "../../biped_mpc_py_amd/csrc/bmpc_stage.hip"
#include "bmpc.h"

namespace {

template <int H>
void run_h(const bmpc::DevParams& P, int B, const float* x_fb, const float* foot, const uint8_t* contact,
           const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states, int32_t* iters,
           float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg, const bmpc::WarmArgs& warm) {
  constexpr int NT = bmpc::Dims<H>::NT;
  for (int b = 0; b < B; ++b) {
    std::barrier<> bar(NT);
    g_bar = &bar;
    std::vector<std::unique_ptr<std::barrier<>>> wb;
    for (int w = 0; w < NT / 64; ++w) { wb.emplace_back(new std::barrier<>(64)); g_wbar[w] = wb.back().get(); }
    for (int p = 0; p < NT / 2; ++p) g_pair[p].store(0);
    g_or[0] = g_or[1] = 0;
    std::vector<std::thread> th;
    th.reserve(NT);
    for (int t = 0; t < NT; ++t)
      th.emplace_back([&, t]() {
        threadIdx.x = t;
        blockIdx.x = b;
        bmpc::solve_kernel<H>(P, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
      });
    for (auto& x : th) x.join();
  }
}

template <int NP, int NW>
void run_stage(const bmpc::DevParams& P, int B, const float* x_fb, const float* foot, const uint8_t* contact,
               const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states, int32_t* iters,
               float* resid, int32_t* status, int32_t* nfactor, const bmpc::DebugOut& dbg, const bmpc::WarmArgs& warm) {
  constexpr int NT = 64 * NW;
  for (int b = 0; b < B; ++b) {
    std::barrier<> bar(NT);
    g_bar = &bar;
    std::vector<std::unique_ptr<std::barrier<>>> wb;
    for (int w = 0; w < NW; ++w) { wb.emplace_back(new std::barrier<>(64)); g_wbar[w] = wb.back().get(); }
    for (int p = 0; p < NT / 2; ++p) g_pair[p].store(0);
    g_or[0] = g_or[1] = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < NT; ++t)
      th.emplace_back([&, t]() {
        threadIdx.x = t;
        blockIdx.x = b;
        bmpc::stage_kernel<NP, NW>(P, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm);
      });
    for (auto& x : th) x.join();
  }
}

// the same mapping as make_dev_params in csrc/bmpc_capi.hip (kept in step by tests/test_emu.py: identical outputs)
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

}  // namespace

extern "C" int bmpc_emu_threads(int h) { return h == 10 ? bmpc::Dims<10>::NT : (h == 16 ? bmpc::Dims<16>::NT : (h == 20 ? bmpc::Dims<20>::NT : -1)); }
// doubles per instance of the warm-start buffer of the stage path
extern "C" int bmpc_emu_stage_warm(int h) { return 5 * bmpc::stage_steps_per_lane(h) * bmpc::stage_waves(h) * 12 * 6; }

extern "C" int bmpc_emu_solve(const bmpc_params* p, int B, const float* x_fb, const float* foot, const uint8_t* contact,
                              const int32_t* phase, const float* x_cmd, const float* mu, float* controls, float* states,
                              int32_t* iters, float* resid, int32_t* status, int32_t* nfactor,
                              double* dbg_x_ref, double* dbg_foot_ref, double* dbg_Gt, double* dbg_qt, int assemble_only,
                              double* warm_buf, int warm_load, int warm_store, int warm_shift, double warm_theta) {
  bmpc::DevParams d;
  std::memset(&d, 0, sizeof(d));
  d.h = p->h; d.half = p->half; d.max_iter = p->max_iter; d.check_every = p->check_every;
  d.adapt_start = p->adapt_start; d.adapt_every = p->adapt_every; d.max_refactor = p->max_refactor;
  d.adapt_early = p->adapt_early; d.adapt_late = p->adapt_late;
  d.adapt_busy = p->adapt_busy; d.adapt_flips = p->adapt_flips;
  d.confirm_from = p->confirm_from; d.kappa_confirm = (float)p->kappa_confirm;
  d.dt = p->dt; d.kv = p->kv; d.m = p->m; d.g = p->g; d.mu = p->mu;
  d.lt = p->lt - 0.01; d.lh = p->lh - 0.02; d.alpha = p->alpha;
  for (int i = 0; i < 12; ++i) { d.x_cmd[i] = p->x_cmd[i]; d.Q[i] = p->Q[i]; d.R2[i] = 2.0 * p->R[i]; }
  for (int k = 0; k < 3; ++k) { d.sq_e[k] = std::sqrt(2.0 * p->Q[k]); d.sq_w[k] = p->dt * std::sqrt(2.0 * p->Q[6 + k]); }
  d.kpm = p->dt * p->dt / p->m;
  d.kvm = p->dt / p->m;
  {
    double rmin = p->R[0];
    for (int i = 1; i < 12; ++i) rmin = std::fmin(rmin, p->R[i]);
    d.r2min = (float)(2 * rmin);
    d.accel = p->accel ? 1 : 0;
  }
  if (!inv3(p->I, d.Iinv)) return -1;
  for (int i = 0; i < 3; ++i) {
    d.f_max[i] = p->f_max[i]; d.f_min[i] = p->f_min[i]; d.tau_max[i] = p->tau_max[i]; d.tau_min[i] = p->tau_min[i];
  }
  d.rho = (float)p->rho; d.rho_eq = (float)(p->rho * p->rho_eq_scale); d.rho_lo = (float)p->rho_lo;
  d.rho_hi_f = (float)p->rho_hi_f; d.rho_hi_m = (float)p->rho_hi_m;
  d.eps_pri = (float)p->eps_pri; d.eps_dua = (float)p->eps_dua; d.kappa = (float)p->kappa;
  {                                           // (f32 products exactly as the kernels used to form them: SLOW_TOL = 1e-6, U0_TOL = 5)
    const float slow_tol = 1.0e-6f, u0_tol = 5.f;
    d.kappa_sqrt = std::sqrt(d.kappa);
    d.kappa_qrt = std::sqrt(std::sqrt(d.kappa));
    d.slow_tol_r2 = slow_tol * d.r2min;
    d.slow_tol_r2_u0 = u0_tol * slow_tol * d.r2min;
    d.eps_u0 = u0_tol * std::fmax(d.eps_pri, d.eps_dua);
  }
  bmpc::DebugOut dbg = {dbg_x_ref, dbg_foot_ref, dbg_Gt, dbg_qt, nullptr, assemble_only};
  bmpc::WarmArgs warm = {warm_buf, warm_load, warm_store, warm_shift, (float)warm_theta, p->warm_adapt_start};
  if (const char* e = std::getenv("BMPC_EMU_POISON")) g_poison = std::atoi(e);
  if (p->path == BMPC_PATH_STAGE) {
    switch (10 * bmpc::stage_waves(p->h) + bmpc::stage_steps_per_lane(p->h)) {
#define EMU_CASE(NN, WW) case 10 * WW + NN: run_stage<NN, WW>(d, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm); break;
      EMU_CASE(2, 1) EMU_CASE(3, 1) EMU_CASE(4, 1) EMU_CASE(5, 1) EMU_CASE(3, 2) EMU_CASE(4, 2)
#undef EMU_CASE
      default: return -1;
    }
    return 0;
  }
  switch (p->h) {
    case 10: run_h<10>(d, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm); break;
    case 16: run_h<16>(d, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm); break;
    case 20: run_h<20>(d, B, x_fb, foot, contact, phase, x_cmd, mu, controls, states, iters, resid, status, nfactor, dbg, warm); break;
    default: return -1;
  }
  return 0;
}
