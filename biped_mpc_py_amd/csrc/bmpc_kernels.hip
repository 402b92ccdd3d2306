// bmpc_kernels.hip -- HIP kernels (gfx950 / CDNA4) for the batched HECTOR force-and-moment MPC.
//
// One workgroup solves one MPC instance end to end; a launch covers a batch.  Nothing but the
// ~1.2 KB/instance of compulsory I/O touches HBM: the whole working set (wrench-space Hessian row,
// K^-1 row, per-step 6x6 blocks, iterates) lives in VGPRs and LDS.
//
// What is computed (citations: REF = /root/reference/bipedalLocomotionMPC.py, read-only spec):
//   references       x_ref, foot_ref                               REF:61-109
//   SRBM step data   Rot, I_w^-1, R_inv, r_f = foot - com          REF:148-185
//   condensing       X = s + Gam_t b, b_j = W_j u_j (net wrench)   REF:203-216 eliminated analytically
//   cost             Gt = 2 Gam_t' Q Gam_t, qt = 2 Gam_t' Q (s - x_ref),  + u'Ru     REF:278-286
//   constraints      box (REF:235-251), friction pyramid (REF:220-232), line foot (REF:254-271)
//   solve            the unique minimiser REF:297 asks cvxopt for, by ADMM with active-set adaptive
//                    penalties; unpack controls / states (REF:300-304)
//
// Thread map: TWO lanes per wrench row, (row, half hf = l % 2), row = (step j = row / 6, component c = row % 6); rows are
// dense over the lanes at h = 16 (row = l / 2), and at h = 10 / 20 five whole steps (30 rows, 60 lanes) fill a wave, so
// that a step never straddles waves and what its lanes exchange needs no s_barrier (Dims<H>::WL).  The two lanes of a
// row are neighbours, so they exchange through DPP (no LDS, no barrier).  Between them they split
//   * the row of V = (Gt + F)^-1 and of Gt by COLUMN halves: lane hf holds the columns of steps
//     [hf H/2, (hf + 1) H/2) -- half the sweep, half the mat-vecs, half the registers each,
//   * the per-step control-space work by FOOT: lane hf owns control variable c of foot hf at step j
//     (v = [f(3), m(3)] per foot), its box row and general row c (4 friction + 2 line-foot) of that foot.
// A workgroup is 2 waves at h = 10 (3 at h = 16, 4 at h = 20), every lane stays under 256 registers, and two
// waves share a SIMD: one wave's LDS round trips and barriers are covered by the other's arithmetic.
// Arithmetic: data and the application of the preconditioner K^-1 (V sweep, V mat-vec, stored 6x6
// factors) are f32, the sweep and mat-vecs on the packed-f32 pipe; the 6x6 block algebra, the
// iterates and the KKT residual the preconditioner is applied to are f64, which is what pins the
// fixed point to the fp64 optimum.
//
// The file also compiles as plain C++ for tests/emu (BMPC_EMU: one std::thread per lane, barriers for
// __syncthreads and for the cross-lane operations) so that the kernel's logic is testable without a GPU.

#ifndef BMPC_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace bmpc {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef double RT;                       // iterate / residual / block-algebra arithmetic

#ifndef BMPC_EMU
__device__ __forceinline__ double rcp_approx(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq_approx(float x) { return __builtin_amdgcn_rsqf(x); }
// value of the other lane of the pair (lane ^ 1): DPP quad_perm [1, 0, 3, 2].  Call with all lanes active.
__device__ __forceinline__ int pair_swap_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }
// Workgroup barrier.  __syncthreads() alone is not enough here: its release fence should make the compiler wait for
// the wave's outstanding LDS operations (s_waitcnt lgkmcnt(0)) before s_barrier, but ROCm 7.2's hipcc leaves that
// wait out at the top of the sweep loop, whose back edge carries a pending ds_write (the publication of the next
// pivot columns) -- the other wave then reads the pivot buffer before the store has landed.  Seen on MI355X as
// results that change from run to run once two waves share a SIMD (about 1 % of the instances of a 4096 batch went
// wrong after a change of the code around the barriers; the build before it happened to get away with it).  The
// wait is therefore written out; tests/test_kernel_resources.py checks the ISA for it at every s_barrier.
__device__ __forceinline__ void sync_workgroup() {
  __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0) (vmcnt, expcnt untouched)
  __syncthreads();
}
// Exchange between lanes of ONE wave through LDS: a wave's DS operations execute in order, so the hand-over needs
// no s_barrier and no wait; the fences only keep the compiler from reordering the accesses.
#define BMPC_WAVE_SYNC()                                        \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
  } while (0)
__device__ __forceinline__ int sync_workgroup_or(int v) {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  return __syncthreads_or(v);
}
// s_waitcnt lgkmcnt(0): used at the back edges of the two hot loops, so that the compiler's bookkeeping of LDS
// operations in flight never has to be right across a back edge (it was not, at the sweep loop: sync_workgroup above)
#define BMPC_DRAIN_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)
#define BMPC_FENCE() asm volatile("" ::: "memory")
#define BMPC_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)   // the instruction scheduler moves nothing across
// hides a loop-invariant f32 value from the optimiser at its point of use, so that its f64 conversion is
// redone there instead of being hoisted into a second, f64, register copy that lives across the loop
#define BMPC_OPAQUE(x) asm volatile("" : "+v"(x))
// a condition every lane of the wave evaluates alike, as a scalar (the branch on it is a scalar branch)
#define BMPC_UNIFORM(x) (__builtin_amdgcn_readfirstlane((int)(x)) != 0)
// an integer every lane of the wave computes alike, as a scalar (loop bounds on it make a scalar loop)
#define BMPC_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif
__device__ __forceinline__ double widen(float v) { BMPC_OPAQUE(v); return (double)v; }
__device__ __forceinline__ float pair_swap(float v) { return __int_as_float(pair_swap_i(__float_as_int(v))); }
__device__ __forceinline__ double pair_swap(double v) {
  const int lo = pair_swap_i(__double2loint(v)), hi = pair_swap_i(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ f2 pair_swap(f2 v) { return f2{pair_swap(v.x), pair_swap(v.y)}; }

typedef double f64x4 __attribute__((ext_vector_type(4)));
#ifndef BMPC_EMU
// D += A B on the matrix cores in f64 (v_mfma_f64_16x16x4_f64): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; it holds D[row = (l >> 4) + 4 reg][col = l & 15] in reg = 0 .. 3.  All 64 lanes of the wave call.
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
#endif

struct DevParams {
  int h, half, max_iter, check_every, adapt_start, adapt_every, max_refactor;
  int adapt_early, adapt_late;                 // two-rate schedule: the first adapt_early re-classifications adapt_every apart, then adapt_late (0: one rate)
  int adapt_busy, adapt_flips;                 // ... but adapt_busy after one that found more than adapt_flips rows in another class than the one before (0: off)
  int confirm_from;                            // from re-classification number confirm_from + 1 on a row found in the same class as at the previous
  float kappa_confirm;                         //   one moves by kappa_confirm instead of kappa (0: off)
  float r2min;                                 // 2 min R: the softest curvature of the problem
  double dt, kv, m, g, mu, lt, lh, alpha;      // lt, lh already carry the REF:254-255 margins
  double x_cmd[12], Q[12], R2[12], Iinv[9];    // R2 = 2 R;  Iinv = inverse body inertia
  double sq_e[3], sq_w[3];                     // sqrt(2 Q_euler), dt sqrt(2 Q_omega): the row scales of the Hessian-block GEMM (set-up)
  // uniform f32 values of the stopping test and the re-classification, formed once on the host (gfx950 has no scalar float unit:
  // formed in the kernel they are loop invariants in VECTOR registers -- the h = 20 kernel spilled six of them)
  float kappa_sqrt, kappa_qrt;                 // sqrt(kappa), sqrt(sqrt(kappa)): the damped moves after 10 / 16 factorisations
  float slow_tol_r2, slow_tol_r2_u0, eps_u0;   // SLOW_TOL 2 R_min;  U0_TOL times that;  U0_TOL max(eps_pri, eps_dua)
  double kpm, kvm;                             // dt^2 / m, dt / m (formed once on the host: two IEEE f64 divisions per lane otherwise, and
                                               //   four registers a lane kept -- or spilled -- across the iteration loop for the rebuilds)
  double f_max[3], f_min[3], tau_max[3], tau_min[3];
  float rho, rho_eq, rho_lo, rho_hi_f, rho_hi_m, eps_pri, eps_dua, kappa;
  int accel;                                   // secant extrapolation of the iterate at the stopping tests (dense family)
};

struct DebugOut {            // all nullable, fp64, device pointers
  double* x_ref;             // [B][H][12]
  double* foot_ref;          // [B][H][6]
  double* Gt;                // [B][6H][6H]
  double* qt;                // [B][6H]
  long long* prof;           // [B][16] cycle stamps (diagnostics / tools only)
  int assemble_only;
};

// Receding-horizon warm start: per lane the iterate (x, z, y of its box and general row)
// and its two penalties, kept in HBM between solves of the same batch slot.
struct WarmArgs {
  double* buf;               // [B][NT][6] or null: {x, z_box, z_gen, y_box, y_gen, (rho_box, rho_gen) as two floats}
  int load, store;           // start from buf / leave the final state in buf
  int shift;                 // the stored horizon is advanced by this many steps on load (0: same schedule phase)
  float theta;               // penalties restart at rho0 (rho_stored / rho0)^theta: 1 keeps them, 0 forgets them
  int adapt_start;           // first penalty re-classification of a warm-started solve (<= 0: as for a cold one)
  // dispatch order or null: workgroup g solves instance order[g] (a permutation of 0 .. B-1).  Workgroups start in
  // index order, so listing the instances expected to take longest first keeps the tail of a batch short.
  const int32_t* order;
  // rescue pass or null (stage-structured kernels only): a status array of an earlier solve of the same batch; the
  // instances it reports solved (0) are left alone, the others are solved again from a cold start
  const int32_t* rescue_status;
  // fp64 outputs or null: `controls` / `states` are then stored here, widened ((double)(float) value: exactly the fp32 result), and
  // the fp32 arrays are not touched.  What the host-pointer entry points use: the arrays are page-locked host memory mapped
  // into the device's address space, so the results cross PCIe as the instances finish, in the dtype REF:300-304 returns --
  // no device-to-host copy and no widening pass after the kernel.
  double* controls64;
  double* states64;
};

template <int H>
struct Dims {
  static_assert(H % 2 == 0, "the column halves are whole steps");
  static constexpr int NW = 6 * H;                       // wrench rows
  static constexpr int HN = NW / 2;                      // columns of V a lane holds (= 3 H)
  static constexpr int HH = H / 2;                       // steps per column half
  // Lane map.  Where five steps (30 rows, 60 lanes) fill a wave -- h = 10: 2 waves, h = 20: 4 -- the steps never straddle
  // waves (SW = 5 steps per wave, the last 4 lanes of every wave clone the wave's last row), and everything exchanged
  // between the lanes of ONE step (the 6x6 block algebra, the control-space residual, beta = L'r) stays inside a
  // wave: no s_barrier, only the wave's own LDS ordering.  h = 16 (16 steps: 3 waves of 32 rows) keeps rows dense.
  // (four steps per wave at h = 16 -- 4 waves instead of 3, a quarter of the lanes clones -- measured 8 % slower)
  static constexpr int SW = (H % 5 == 0) ? 5 : 0;
  static constexpr bool WL = SW > 0;                      // steps are wave-local
  static constexpr int RW = 6 * SW;                       // rows per wave
  static constexpr int NT = WL ? 64 * (H / SW) : ((2 * NW + 63) / 64) * 64;   // threads per workgroup (whole waves)
  static constexpr int NWV = NT / 64;
  __host__ __device__ static constexpr int lane_of(int row, int f) { return WL ? 64 * (row / RW) + 2 * (row % RW) + f : 2 * row + f; }
  // Waves per SIMD the register allocation aims at.  An instance is a latency-bound chain of LDS exchanges, so a
  // CU's throughput is (instances in flight) / (latency of one); two per SIMD = 4 instances per CU at h = 10.
  // Three per SIMD (<= 168 registers) was measured with the DPP-broadcast variant of this kernel (docs/
  // history_r02_r03.md): 5 instances per CU at h = 10 gained nothing at the 4096-instance batch (the vector pipe was
  // then ~80 % busy), h = 16 gained 11 %; this variant needs too many spills for it (155 at h = 16).
  static constexpr int WPE = 2;
  static constexpr int NPAIR = H * (H - 1) / 2;          // (i > j) step pairs
  // A vector over the wrench rows that both column halves read is stored in LDS as two 16-byte aligned
  // halves: entry i sits at slot(i).  The halves start 4 k dwords apart with 4 k mod 64 outside (-4, 4), so
  // the two addresses of a wave's ds_read_b128 (even lanes: half 0, odd lanes: half 1) never share a bank.
  static constexpr int HNP = ((HN + 3) / 4) * 4;
  static constexpr int VL = 2 * HNP;
  static constexpr int PVS = ((VL + NW + 3) / 4) * 4;    // one published pivot column: two-half vector + one dump slot per row
  static_assert(HNP % 64 >= 4 && HNP % 64 <= 60, "halves of a two-half vector would collide on LDS banks");
  // Gt row half: 3 component groups x HH steps, padded to whole float4s
  static constexpr int GH = ((3 * HH + 3) / 4) * 4;
  static constexpr int GS = (GH % 16 == 0 && GH % 64 != 16 && GH % 64 != 48) ? GH + 4 : GH;   // stride of the 4 gamma copies
};
template <int H>
__device__ __forceinline__ constexpr int slot(int i) { return i < Dims<H>::HN ? i : i - Dims<H>::HN + Dims<H>::HNP; }

// LDS image of one instance.  The factor scratch (f64 6x6 blocks) and the per-iteration exchange
// vectors (+ the set-up-only step data) are never live at the same time and share one region.
template <int H>
struct alignas(16) FacScratch {
  double M0[H][6][6];        // D0 -> Ka^-1 D0 W_0^-1
  double M1[H][6][6];        // D1 -> Ka^-1
  double M2[H][6][6];        // B = T' D1 T -> L_0
  double Ka[H][6][6];        // Ka = D0 + B, read whole by every lane of the step
};
template <int H>
struct IterScratch {
  static constexpr int NW = Dims<H>::NW;
  RT wg[H][2][6];            // y + rho (A x - z) on the general rows
  alignas(16) RT bwT[6][H];  // net wrench of x, component-major: a lane reads its inputs of Gt contiguously
  RT gb[NW];                 // wrench-space gradient Gt b + qt
  alignas(16) float r32[H][2][6];   // KKT residual, control space
  alignas(16) float beta[Dims<H>::VL];   // d .* L' r, two-half layout
  alignas(16) float gam[NW];
  // gamma again for the gradient increment: per (component group = torque / force, column half) the 3 HH
  // values a lane multiplies with its Gt row half, contiguous, zero padded to GH
  alignas(16) float gamT[4][Dims<H>::GS];
  // exact gradient (refresh): y_j = dt Iw_j tau_j, the weighted tracking error 2 Q (X - x_ref) of the 12 states
  // (component-major: a lane sums one component over the steps) and the adjoint's 3-vector m_j of the step
  RT xs[H][2][6];            // x (relaxed iterate): the exchange of the rebuild
  RT yw[3][H];
  RT qvT[12][H];
  RT mv[H][3];
  // set-up only
  RT Rv[H][9];               // R_inv (REF:160-164)
};
// Set-up: the torque block of Gt leaves the matrix cores as 16 x 16 accumulator tiles -- the NU = NTL (NTL + 1) / 2 tiles (I <= J)
// of the symmetric tiling of its 3 H rows -- and is handed to the lanes that keep its rows through LDS as a dense row-major
// matrix of the f32 values the rows are kept in: a tile is stored twice, as the block (I, J) and transposed as (J, I), so that a
// lane reads its row half at one base address + immediate offsets (one float of padding per row: the transposed stores walk
// down a column and hit distinct banks).  Rows / columns 3 H .. 16 NTL - 1 are the tiles' zero padding.
template <int H>
struct GramTiles {
  static constexpr int N3 = 3 * H;                       // torque rows
  static constexpr int NTL = (N3 + 15) / 16;
  static constexpr int NU = NTL * (NTL + 1) / 2;
  static constexpr int LD = 16 * NTL + 1;
  float g[16 * NTL][LD];
};
template <int H>
struct alignas(16) Smem {
  static constexpr int NW = Dims<H>::NW;
  union alignas(16) {
    FacScratch<H> fac;
    IterScratch<H> itv;
    GramTiles<H> gram;
  } u;
  // the two pivot columns of a sweep step, double buffered, two-half layout; behind each one dump slot per row:
  // the half-1 lanes, which hold no pivot-column entry, store there instead of branching around the store
  alignas(16) float piv[2][2 * Dims<H>::PVS];
  alignas(16) float dsc[Dims<H>::VL];      // Jacobi scaling of the current factorisation
  // block-diagonal part of K^-1.  Foot-major: a lane's row sits at 48 B x row + const.
  // Each entry is a pair {factor, G_f x factor}: the step d and its general-row image G_f d are the same dot
  // products against (t, gamma) and run as one packed FMA per term.
  // The two feet's blocks start 2 (mod 4) dwords apart: a wave's 8-byte reads of the rows (12 dwords apart) then
  // use banks 4 m, 4 m + 1 for foot 0 and 4 m + 2, 4 m + 3 for foot 1 instead of colliding pairwise.
  struct alignas(8) FootBlock { float d[H][6][6][2]; float pad[2]; };
  alignas(16) FootBlock LG[2];          // {L, G L}:   L_j = D^-1 W' F            [foot].d[step][var][wrench comp]
  alignas(16) FootBlock KG[2];          // {Kn, G Kn}: Kn = Ka^-1 (foot 0), T Ka^-1 (foot 1); N Ka^-1 N' r = N (Ka^-1 (N' r))
  // step data
  RT Iw[H][9];               // world inverse inertia
  RT rr[H][2][3];            // r_f = foot_ref - com_ref
  alignas(16) RT rx[H][2][6][2];  // per variable: {r_f[a+2], r_f[a+1]} (cyclic) for force variable a, zeros for moments
  RT err0[H][12];            // free response - reference: the tracking error at u = 0
  float xrf[H][12];          // the reference itself (f32 is enough: only the f32 `states` output adds it back)
  RT Q2[12];                 // 2 Q (REF:27, 278-284)
  // coefficient table of the exact gradient: rows 0..8 P_i[a][b] over the steps i (P_i = sum_{l <= i} R_inv,l, REF:160-171:
  // euler_i = ... + dt sum_{l < i} (P_i - P_l) y_l), row 9 the step index, row 10 zeros
  RT CT[11][H];
  float Me[Dims<H>::NPAIR > 0 ? Dims<H>::NPAIR : 1][9];   // dt^2 (P_i - P_j) Iw_j, i > j (data: f32)
  float rvg[H][2][6];
  float muf[H][2];           // friction coefficient per step and foot
  RT Gu[6][6];               // mu-free part of the general rows of a foot block, and its transpose
  RT GuT[6][6];
  float eyz[6];              // body y and z axes in the world frame (columns 1, 2 of eul2rotm(x_fb))
  float red[2][H == 12 ? 6 : 12][Dims<H>::NWV];
  float aag[H == 12 ? 0 : Dims<H>::NT][7];                   // secant extrapolation: a lane's state change (x, zb, zg, yb, yg, A x, gradient), kept from
                                               // the iteration before a stopping test, and over the test's reduction
                                               // (not at h = 12: the 3.8 KB would cost that kernel its fourth instance per CU)
};

__device__ __forceinline__ int pair_index(int i, int j) { return i * (i - 1) / 2 + j; }   // i > j

#ifndef BMPC_EMU
// max over a wave of a NON-NEGATIVE float (or NaN).  The order of such floats is the order of their bit
// patterns, with NaN above everything, so the reduction is an unsigned max: six DPP steps (shifts read
// 0 = the neutral element where a source lane does not exist).  NaNs propagate.
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#define BMPC_DPP_MAX(ctrl) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, true); v = v > o ? v : o; }
  BMPC_DPP_MAX(0x111)   // row_shr:1
  BMPC_DPP_MAX(0x112)   // row_shr:2
  BMPC_DPP_MAX(0x114)   // row_shr:4
  BMPC_DPP_MAX(0x118)   // row_shr:8   -> lane 15 of each row holds the row maximum
  BMPC_DPP_MAX(0x142)   // row_bcast:15 -> lane 31 / 63: rows 0-1 / 2-3
  BMPC_DPP_MAX(0x143)   // row_bcast:31 -> lane 63: whole wave
#undef BMPC_DPP_MAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#endif
#ifndef BMPC_EMU
// max of a NON-NEGATIVE float (or NaN) over the FIRST ROW of 16 lanes of the wave, returned to every lane of the wave: four
// DPP steps instead of six (the statistics of step 0, whose twelve lanes open wave 0 in every lane map of both families)
__device__ __forceinline__ unsigned row0_umax(unsigned v) {
#define BMPC_DPP_MAX(ctrl) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, true); v = v > o ? v : o; }
  BMPC_DPP_MAX(0x111)   // row_shr:1
  BMPC_DPP_MAX(0x112)   // row_shr:2
  BMPC_DPP_MAX(0x114)   // row_shr:4
  BMPC_DPP_MAX(0x118)   // row_shr:8   -> lane 15 holds the maximum of lanes 0 .. 15
#undef BMPC_DPP_MAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
#endif
#ifndef BMPC_EMU
// sum over a wave, in a FIXED order (the same DPP tree as wave_umax: within rows of 16 by shifts of 1, 2, 4, 8, then the
// rows): bitwise reproducible from run to run
__device__ __forceinline__ float wave_sum(float v) {
#define BMPC_DPP_ADD(ctrl) { v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true)); }
  BMPC_DPP_ADD(0x111)   // row_shr:1
  BMPC_DPP_ADD(0x112)   // row_shr:2
  BMPC_DPP_ADD(0x114)   // row_shr:4
  BMPC_DPP_ADD(0x118)   // row_shr:8   -> lane 15 of each row holds the row sum
  BMPC_DPP_ADD(0x142)   // row_bcast:15 -> lane 31 / 63: rows 0-1 / 2-3
  BMPC_DPP_ADD(0x143)   // row_bcast:31 -> lane 63: whole wave
#undef BMPC_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif
// max over the workgroup of NV such values at once; the waves combine through LDS.  All threads call.  ONE barrier:
// the caller alternates between two `red` buffers, so a buffer is rewritten only after another barrier.
template <int NT, int NV>
__device__ __forceinline__ void block_max(float (&v)[NV], float (*red)[NT / 64]) {
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = __uint_as_float(wave_umax(__float_as_uint(v[q])));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) red[q][w] = v[q];
  }
  sync_workgroup();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    unsigned m = __float_as_uint(red[q][0]);
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2) { const unsigned o = __float_as_uint(red[q][w2]); m = m > o ? m : o; }
    v[q] = __uint_as_float(m);
  }
}

// the same exchange with NS sums behind the NV maxima (v[NV .. NV + NS - 1]; waves added in index order); the last NR of
// the maxima are non-zero in the first 16 lanes of a wave only (step 0's statistics) and take the short reduction
template <int NT, int NV, int NS, int NR = 0>
__device__ __forceinline__ void block_max_sum(float (&v)[NV + NS], float (*red)[NT / 64]) {
#ifndef BMPC_EMU
  // the DPP steps of all values in lockstep (step by step over the values, not value by value): a DPP operation needs two
  // wait states after the write of its source, which the other values' steps fill instead of s_nops
  {
    unsigned u[NV];
    float f[NS > 0 ? NS : 1];
#pragma unroll
    for (int q = 0; q < NV; ++q) u[q] = __float_as_uint(v[q]);
#pragma unroll
    for (int q = 0; q < NS; ++q) f[q] = v[NV + q];
#define BMPC_STEP_ALL(ctrl, NMAX)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < NMAX; ++q) {                                                                         \
      const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[q], ctrl, 0xf, 0xf, true); u[q] = u[q] > o ? u[q] : o; } \
    _Pragma("unroll") for (int q = 0; q < NS; ++q)                                                                             \
      f[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f[q]), ctrl, 0xf, 0xf, true));
    BMPC_STEP_ALL(0x111, NV)
    BMPC_STEP_ALL(0x112, NV)
    BMPC_STEP_ALL(0x114, NV)
    BMPC_STEP_ALL(0x118, NV)
    BMPC_STEP_ALL(0x142, NV - NR)              // (the last NR values live in the first row of 16 lanes only: done)
    BMPC_STEP_ALL(0x143, NV - NR)
#undef BMPC_STEP_ALL
#pragma unroll
    for (int q = 0; q < NV - NR; ++q) v[q] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u[q], 63));
#pragma unroll
    for (int q = NV - NR; q < NV; ++q) v[q] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u[q], 15));
#pragma unroll
    for (int q = 0; q < NS; ++q) v[NV + q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f[q]), 63));
  }
#else
#pragma unroll
  for (int q = 0; q < NV - NR; ++q) v[q] = __uint_as_float(wave_umax(__float_as_uint(v[q])));
#pragma unroll
  for (int q = NV - NR; q < NV; ++q) v[q] = __uint_as_float(row0_umax(__float_as_uint(v[q])));
#pragma unroll
  for (int q = NV; q < NV + NS; ++q) v[q] = wave_sum(v[q]);
#endif
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NV + NS; ++q) red[q][w] = v[q];
  }
  sync_workgroup();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    unsigned m = __float_as_uint(red[q][0]);
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2) { const unsigned o = __float_as_uint(red[q][w2]); m = m > o ? m : o; }
    v[q] = __uint_as_float(m);
  }
#pragma unroll
  for (int q = NV; q < NV + NS; ++q) {
    float a = red[q][0];
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2) a += red[q][w2];
    v[q] = a;
  }
}

// out[b] += sum_q w[q] * M[q][b] for a 6x6 f64 matrix in LDS: whole rows are fetched (three 16-byte
// reads each) and the six outputs accumulate as independent chains.
__device__ __forceinline__ void row_times_mat6(const double (&w)[6], const double (*M)[6], double (&out)[6]) {
#pragma unroll
  for (int q0 = 0; q0 < 6; q0 += 3) {           // three rows (nine 16-byte reads) in flight, then their FMAs
    double rw[3][6];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int b = 0; b < 6; b += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&M[q0 + q][b]);
        rw[q][b] = v.x; rw[q][b + 1] = v.y;
      }
    BMPC_SCHED_BARRIER();
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int b = 0; b < 6; ++b) out[b] = fma(w[q0 + q], rw[q][b], out[b]);
  }
}

// Row c of the inverse of a 6x6 SPD matrix in LDS, computed by the calling lane alone (LDL' of the lower
// triangle, then L D L' x = e_c): ~130 f64 operations and no exchange, where a cooperative sweep over the six
// lanes of the step costs twelve barriers -- the factorisation is latency-bound, not flop-bound.  The six lanes
// of a step repeat the same LDL'.  Reciprocals by v_rcp_f64 + two Newton steps.
__device__ __forceinline__ void inv6_row(const double (*K)[6], int c, double (&x)[6]) {
  double a[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k <= i; ++k) a[i][k] = K[i][k];
  BMPC_SCHED_BARRIER();
  double dinv[6];
#pragma unroll
  for (int jj = 0; jj < 6; ++jj) {
    double d = a[jj][jj];
#pragma unroll
    for (int k = 0; k < jj; ++k) d = fma(-a[jj][k] * a[jj][k], a[k][k], d);      // a[k][k] holds d_k, a[i][k] holds l_ik
    double r = rcp_approx(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    dinv[jj] = r;
    a[jj][jj] = d;
#pragma unroll
    for (int i = jj + 1; i < 6; ++i) {
      double v = a[i][jj];
#pragma unroll
      for (int k = 0; k < jj; ++k) v = fma(-a[i][k] * a[jj][k], a[k][k], v);
      a[i][jj] = v * r;
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {                  // L y = e_c
    double v = (c == i) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) v = fma(-a[i][k], y[k], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {                 // L' x = D^-1 y
    double v = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v = fma(-a[k][i], x[k], v);
    x[i] = v;
  }
}

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// General (non-box) rows of one foot block over v = [f(3), m(3)]:
// rows 0..3 friction (+x, +y, -x, -y; REF:220-229), rows 4, 5 line foot (REF:259-262).
__device__ __forceinline__ void general_rows(float mu, const float* ey, const float* ez, float lh,
                                             float lt, float (&G)[6][6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int b = 0; b < 6; ++b) G[r][b] = 0.f;
  G[0][0] = 1.f;  G[0][2] = -mu;
  G[1][1] = 1.f;  G[1][2] = -mu;
  G[2][0] = -1.f; G[2][2] = -mu;
  G[3][1] = -1.f; G[3][2] = -mu;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    G[4][b] = -lh * ez[b];  G[4][3 + b] = ey[b];
    G[5][b] = -lt * ez[b];  G[5][3 + b] = -ey[b];
  }
}

// The solve, as the body of two kernels: solve_kernel<H> (PROF = false: no trace of the diagnostics in the code) and
// solve_kernel_prof<H> (in-kernel cycle stamps for tools/phase_cycles.py, launched while a profile buffer is set).
template <int H, bool PROF>
__device__ __forceinline__ void
solve_body(const DevParams& P, const int B,
             const float* __restrict__ x_fb, const float* __restrict__ foot,
             const uint8_t* __restrict__ contact, const int32_t* __restrict__ phase,
             const float* __restrict__ x_cmd, const float* __restrict__ mu_in,
             float* __restrict__ controls, float* __restrict__ states,
             int32_t* __restrict__ iters_out, float* __restrict__ resid_out,
             int32_t* __restrict__ status_out, int32_t* __restrict__ nfactor_out,
             const DebugOut& dbg, const WarmArgs& warm) {
  constexpr int NW = Dims<H>::NW;
  constexpr int HN = Dims<H>::HN;
  constexpr int HH = Dims<H>::HH;
  constexpr int HNP = Dims<H>::HNP;
  constexpr int GH = Dims<H>::GH;
  constexpr int NT = Dims<H>::NT;
  __shared__ Smem<H> sm;

  if ((int)blockIdx.x >= B) return;
  const int inst = warm.order ? warm.order[blockIdx.x] : (int)blockIdx.x;
#ifdef BMPC_EMU
  // the emulation poisons the LDS image (all-ones bytes: NaNs) so that a read of an entry nobody wrote shows
  if (threadIdx.x == 0) std::memset(&sm, 0xFF, sizeof(sm));
  sync_workgroup();
#endif
  long long t_start = 0, t_setup = 0, t_blocks = 0, t_sweep = 0, t_mark = 0;
  long long t_ph[7] = {0, 0, 0, 0, 0, 0, 0}, t_last = 0, t_red = 0, t_reb = 0;   // (t_red, t_reb: inside t_ph[6], the iteration's tail)
  int n_reb = 0;
  // (tools only: built with -DBMPC_PROF_BLOCKS the seven phase slots of the diagnostics kernel hold the stages of the 6x6 block
  //  algebra of factor() instead of the iteration's phases -- tools/phase_cycles.py --blocks)
#ifdef BMPC_PROF_BLOCKS
#define BMPC_STAMP(k)
#define BMPC_BSTAMP(k) if constexpr (PROF) { const long long t_ = clock64(); t_ph[k] += t_ - t_mark2; t_mark2 = t_; }
  long long t_mark2 = 0;
#else
#define BMPC_STAMP(k) if constexpr (PROF) { const long long t_ = clock64(); t_ph[k] += t_ - t_last; t_last = t_; }
#define BMPC_BSTAMP(k)
#endif
  if constexpr (PROF) t_start = clock64();
  const int l = threadIdx.x;
  const int hf = l & 1;                        // column half of V / Gt, and the foot this lane owns
  const int f = hf;
  // The lanes past the last row (of the wave: 4 per wave at h = 10, 20; of the workgroup at h = 16) CLONE the last row: same indices, same data, same
  // arithmetic, so their LDS writes repeat the real lane's values at the real lane's addresses and nothing has
  // to be predicated (every `if (lane is real)` would be an exec-mask branch, and the code sinking across such
  // branches is what blew up the sweep's register pressure); only their global stores are suppressed.
  const bool real = Dims<H>::WL ? ((l & 63) >> 1) < Dims<H>::RW : (l >> 1) < NW;
  const int row = Dims<H>::WL ? Dims<H>::RW * (l >> 6) + (real ? (l & 63) >> 1 : Dims<H>::RW - 1) : (real ? (l >> 1) : NW - 1);
  constexpr bool valid = true;
  // synchronisation of the lanes of one step (block algebra, control-space residual, beta): wave-local where the
  // lane map keeps a step inside a wave, a workgroup barrier otherwise
  auto sync_step = [&]() {
    if constexpr (Dims<H>::WL) { BMPC_WAVE_SYNC(); } else { sync_workgroup(); }
  };
  const int j = row / 6;
  const int c = row % 6;
  const int j_lane = j, c_lane = c, hf_lane = hf;
  const int jb = hf * HH;                      // first step of this lane's column half
  const RT dt = (RT)P.dt;

  // ------------------------------------------------------------------ A. references, step data
  RT xfb[12], xc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    xfb[i] = (RT)x_fb[(size_t)inst * 12 + i];
    xc[i] = x_cmd ? (RT)x_cmd[(size_t)inst * 12 + i] : (RT)P.x_cmd[i];
  }
  const int kph = phase[inst];
  RT xr[12];                                   // x_ref[:, j]  (REF:61-70)
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    if (j == 0) xr[i] = xfb[i];
    else if (i < 6) xr[i] = (xc[i + 6] != (RT)0) ? xfb[i] + xc[i + 6] * ((RT)j * dt) : xc[i];
    else xr[i] = xc[i];
  }
  RT fr[6];                                    // foot_ref[:, j]  (REF:72-109)
  {
    const int c0 = contact[(size_t)inst * H * 2 + 0], c1 = contact[(size_t)inst * H * 2 + 1];
    const bool single = (c0 + c1) == 1;        // REF:102
    const int kk = kph % P.half;               // REF:101
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = (RT)foot[(size_t)inst * 6 + i];
    if (single && j >= P.half - kk) {
      const bool second = j >= 2 * P.half - kk;
      const RT hor = second ? (RT)0.5 * (RT)H * dt : (RT)0.5 * (RT)H / (RT)2 * dt;   // REF:74, 78
      const RT fx = xfb[3] + xfb[9] * hor + (RT)P.kv * (xfb[3] - xc[3]);
      const RT fy = (second ? xfb[10] : xfb[4]) + xfb[10] * hor + (RT)P.kv * (xfb[4] - xc[4]);  // REF:87 quirk
      fr[0] = fx; fr[1] = fy; fr[2] = 0; fr[3] = fx; fr[4] = fy; fr[5] = 0;
    }
  }
  const bool lead = real && c == 0 && hf == 0;   // one lane per step
  if (lead) {                                  // debug views of the references (tests)
    if (dbg.x_ref) {
#pragma unroll
      for (int i = 0; i < 12; ++i) dbg.x_ref[((size_t)inst * H + j) * 12 + i] = (double)xr[i];
    }
    if (dbg.foot_ref) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dbg.foot_ref[((size_t)inst * H + j) * 6 + i] = (double)fr[i];
    }
  }

  RT Pj[9];                                    // prefix sum of R_inv up to this lane's step
  {
    RT sy, cy, sp, cp, sr, cr;                 // REF:151-153: yaw = x[0], pitch = x[1], roll = x[2]
    sincos(xr[0], &sy, &cy);
    sincos(xr[1], &sp, &cp);
    sincos(xr[2], &sr, &cr);
    // Rot = Rx(roll) Ry(pitch) Rz(yaw)   (scipy 'zyx' extrinsic, REF:154-156)
    const RT Rot[9] = {cp * cy, -cp * sy, sp,
                       cr * sy + sr * sp * cy, cr * cy - sr * sp * sy, -sr * cp,
                       sr * sy - cr * sp * cy, sr * cy + cr * sp * sy, cr * cp};
    RT T[9], Iw[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        T[3 * a + b] = (RT)P.Iinv[3 * a] * Rot[b] + (RT)P.Iinv[3 * a + 1] * Rot[3 + b] + (RT)P.Iinv[3 * a + 2] * Rot[6 + b];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        Iw[3 * a + b] = Rot[a] * T[b] + Rot[3 + a] * T[3 + b] + Rot[6 + a] * T[6 + b];   // Rot' Iinv Rot = (Rot' I Rot)^-1
    const RT tp = sp / cp;
    const RT Rv[9] = {cy / cp, sy / cp, 0, -sy, cy, 0, cy * tp, sy * tp, 1};             // REF:160-164 inverted
    if (lead) {
#pragma unroll
      for (int q = 0; q < 9; ++q) { sm.Iw[j][q] = Iw[q]; sm.u.itv.Rv[j][q] = Rv[q]; }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int a = 0; a < 3; ++a) sm.rr[j][ft][a] = fr[3 * ft + a] - xr[3 + a];         // REF:174-175
    }
    if (l == 0) {
      // Step 0's reference IS the feedback state (REF:63): its sines and cosines are the ones eul2rotm(x_fb[0:3]) needs for the
      // body axes of the line-foot rows (REF:124-138, 193: roll = x[0], pitch = x[1], yaw = x[2]) -- three f64 sincos every lane
      // used to repeat in front of the first factorisation.  Parked in the first row of Gu until phase B forms the rows.
      sm.Gu[0][0] = sy; sm.Gu[0][1] = cy; sm.Gu[0][2] = sp; sm.Gu[0][3] = cp; sm.Gu[0][4] = sr; sm.Gu[0][5] = cr;
    }
  }
  sync_workgroup();
  // P_i = sum_{s <= i} R_inv,s (REF:160-171), entry q by lane q < 9: the H loads carry immediate offsets and are in flight together,
  // the additions run in step order (every lane summing up to its own step cost one LDS round trip per step)
  if (l < 9) {
    RT rv[H], run = 0;
#pragma unroll
    for (int s = 0; s < H; ++s) rv[s] = sm.u.itv.Rv[s][l];
#pragma unroll
    for (int s = 0; s < H; ++s) { run += rv[s]; sm.CT[l][s] = run; }
  }
  sync_workgroup();
#pragma unroll
  for (int q = 0; q < 9; ++q) Pj[q] = sm.CT[q][j];
  if (valid) {
    // free response s_j - x_ref[:, j]   (X_j is the state after step j)
    const RT j1 = (RT)(j + 1);
    RT e12[12];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      e12[a] = xfb[a] + dt * (Pj[3 * a] * xfb[6] + Pj[3 * a + 1] * xfb[7] + Pj[3 * a + 2] * xfb[8]);
      e12[3 + a] = xfb[3 + a] + dt * j1 * xfb[9 + a];
      e12[6 + a] = xfb[6 + a];
      e12[9 + a] = xfb[9 + a];
    }
    e12[5] -= (RT)P.g * dt * dt * (RT)j * j1 / 2;
    e12[11] -= (RT)P.g * dt * j1;
    if (lead) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { sm.err0[j][i] = e12[i] - xr[i]; sm.xrf[j][i] = (float)xr[i]; }
    }
    if (l < 12) sm.Q2[l] = 2 * (RT)P.Q[l];
    if (l < H) { sm.CT[9][l] = (RT)l; sm.CT[10][l] = 0; }
  }
  sync_workgroup();
  // Me[i][j2] = dt^2 (P_i - P_j2) Iw_j2 for i > j2: one (i, j2) pair per lane and pass
  for (int idx = l; idx < Dims<H>::NPAIR; idx += NT) {
    int i = (int)((1.f + sqrtf(1.f + 8.f * (float)idx)) * 0.5f);     // invert idx = i (i - 1) / 2 + j2
    i -= (i * (i - 1) / 2 > idx) ? 1 : 0;
    i += ((i + 1) * i / 2 <= idx) ? 1 : 0;
    const int j2 = idx - i * (i - 1) / 2;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        RT s = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) s += (sm.CT[3 * a + q][i] - sm.CT[3 * a + q][j2]) * sm.Iw[j2][3 * q + b];
        sm.Me[idx][3 * a + b] = (float)(dt * dt * s);
      }
  }
  sync_workgroup();

  // Exact wrench-space gradient gb = Gam_t' 2Q (s - x_ref + Gam_t b) of this lane's row, formed in STATE space (REF:165-184
  // as prefix / suffix sums over the steps, all f64) -- the way the stage-structured family does it (bmpc_stage.hip).  The
  // f32 copy of the Gt row a lane holds is then used by the increments of the carried gradient (which vanish with the step)
  // and by the preconditioner only: it no longer enters the fixed point.  (Rebuilding the gradient from that f32 row left
  // the dense family 2e-5 .. 8e-5 from the fp64 optimum where the stage family ends at 1e-7: DESIGN.md section 4.)
  //   y_l = dt Iw_l tau_l;   omega_i = s + sum_{l <= i} y_l;   euler_i = s + dt sum_{l < i} (P_i - P_l) y_l
  //   p_i = s + dt^2/m sum_{l <= i} (i - l) F_l;   v_i = s + dt/m sum_{l <= i} F_l             (X_i: state after step i)
  // and the adjoint: g_F,j = sum_{i >= j} dt^2/m (i - j) q_p,i + dt/m q_v,i,   g_tau,j = dt Iw_j' m_j,
  //   m_j = sum_{i >= j} q_w,i + dt sum_{i > j} (P_i - P_j)' q_e,i,   q = 2 Q (X - x_ref).
  // The 12 lanes of a step take one state coordinate each: (c < 3, hf 0 / 1) euler / omega, (c >= 3, hf 0 / 1) p / v.
  // With have_b the caller has published the net wrench (bwT rows 3..5: F) and yw and passed a workgroup barrier; without,
  // b = 0.  All threads call (barriers inside).
  auto gradient_exact = [&](const bool have_b) -> RT {
    // (through opaque copies of the lane's indices: none of the address arithmetic below is hoisted out of the
    // iteration loop, where it would hold registers between two rebuilds)
    int rw_ = row, hf = hf_lane;               // (step and component formed again from the row: no copy of them kept for this)
    BMPC_OPAQUE(rw_);
    BMPC_OPAQUE(hf);
    const int j = rw_ / 6, c = rw_ - 6 * j;
    const int a = c < 3 ? c : c - 3;
    const RT kp = (RT)P.kpm, kvv = (RT)P.kvm;
    // Every sum has the form  sum_l (C - tab[l]) src[l]  with a row of the coefficient table CT:
    //   rows 0..8  P[a][b] over the steps   (euler: dt (P_j - P_l)[a][b] y_l,b;  adjoint: dt (P_i - P_j)[b][a] q_e,i,b)
    //   row 9      l                        (position: dt^2/m (j - l) F_l)
    //   row 10     0, with C = 1            (omega, velocity: plain sums)
    // so one loop serves all lane kinds (they sit in the same wave); coefficients are differences formed term by term,
    // never a difference of sums.  The euler sums have three such terms (b = 0, 1, 2): the two lanes of a torque row
    // share them -- lane 0 takes b = 0, 1, lane 1 its own plain sum and b = 2 -- so no lane runs more than two.
    // Both sums of a stage in ONE loop over the steps, two steps per trip (round 6): four table entries and four source entries
    // are in flight together, where a loop per sum with the lane's own bounds paid one LDS round trip per step and sum.  The
    // bounds are the wave's (the steps its lanes own: wj0 .. wj1), a lane masks the steps outside its own range -- exact zeros,
    // and the terms are added in the same order, so the sums are the same to the bit.  Force lanes have one sum: their second
    // one repeats the first and is dropped.  (Fully unrolled over the horizon -- every load of a stage in flight at once -- the
    // kernels spill: 47 registers at h = 10, 166 at h = 20.)  Up to h = 16; see FUSED below.
    constexpr int SU = H <= 12 ? 2 : 1;
    constexpr bool FUSED = H <= 12;
    int wj0 = 0, wj1 = 0;
    if constexpr (FUSED) {
      const int wv_ = l >> 6;
      wj0 = BMPC_UNIFORM_INT(Dims<H>::WL ? Dims<H>::SW * wv_ : (32 * wv_) / 6);
      wj1 = BMPC_UNIFORM_INT(Dims<H>::WL ? Dims<H>::SW * wv_ + Dims<H>::SW - 1 : ((32 * wv_ + 31 < NW - 1 ? 32 * wv_ + 31 : NW - 1) / 6));
    }
    // (up to h = 12, two steps per trip; from h = 14 on the fused loop costs spilled registers -- 1 at h = 14, 4 .. 26 at h = 16, more
    //  beyond -- and those horizons keep one loop per sum with the lane's own bounds)
    auto scan = [&](const int rb, const RT* src, const int lo, const int hi) -> RT {
      const RT* tb = &sm.CT[rb][0];
      const RT Cb = tb[j] + (rb == 10 ? (RT)1 : (RT)0);
      RT acc = 0;
#pragma unroll 1
      for (int l2 = lo; l2 < hi; ++l2) acc = fma(Cb - tb[l2], src[l2], acc);
      return acc;
    };
    auto scan2 = [&](const bool prefix, const int rb0, const RT* src0, const int rb1, const RT* src1, RT& acc0, RT& acc1) {
      const RT* tb0 = &sm.CT[rb0][0];
      const RT* tb1 = &sm.CT[rb1][0];
      const RT Cb0 = tb0[j] + (rb0 == 10 ? (RT)1 : (RT)0), Cb1 = tb1[j] + (rb1 == 10 ? (RT)1 : (RT)0);
      acc0 = 0; acc1 = 0;
      const int lo = prefix ? 0 : (wj0 & ~(SU - 1)), hi = prefix ? wj1 + 1 : H;       // (H is even: a pair of steps never leaves the horizon)
#pragma unroll 1
      for (int l2 = lo; l2 < hi; l2 += SU) {
        RT t0[SU], s0[SU], t1[SU], s1[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) { t0[q] = tb0[l2 + q]; s0[q] = src0[l2 + q]; t1[q] = tb1[l2 + q]; s1[q] = src1[l2 + q]; }
        BMPC_SCHED_BARRIER();
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const bool on = prefix ? (l2 + q <= j) : (l2 + q >= j);
          acc0 = fma(on ? Cb0 - t0[q] : (RT)0, s0[q], acc0);
          acc1 = fma(on ? Cb1 - t1[q] : (RT)0, s1[q], acc1);
        }
      }
    };
    const int sidx = c < 3 ? (hf == 0 ? a : 6 + a) : (hf == 0 ? 3 + a : 9 + a);
    RT ev = sm.err0[j][sidx];
    if (have_b) {                               // (uniform)
      const RT* src0 = c < 3 ? &sm.u.itv.yw[hf == 0 ? 0 : a][0] : &sm.u.itv.bwT[3 + a][0];
      const int rb0 = c < 3 ? (hf == 0 ? 3 * a : 10) : (hf == 0 ? 9 : 10);
      RT acc0, acc1 = 0;
      if constexpr (FUSED) {
        scan2(true, rb0, src0, c < 3 ? 3 * a + (hf == 0 ? 1 : 2) : rb0, c < 3 ? &sm.u.itv.yw[hf == 0 ? 1 : 2][0] : src0, acc0, acc1);
        acc1 = c < 3 ? acc1 : (RT)0;
      } else {
        acc0 = scan(rb0, src0, 0, j + 1);
        if (c < 3) acc1 = scan(3 * a + (hf == 0 ? 1 : 2), &sm.u.itv.yw[hf == 0 ? 1 : 2][0], 0, j + 1);
      }
      const RT got = pair_swap(acc1);           // (lane 0 of a torque row receives the b = 2 term)
      const RT scale = c < 3 ? (hf == 0 ? dt : (RT)1) : (hf == 0 ? kp : kvv);
      ev = fma(scale, c < 3 && hf == 0 ? acc0 + (acc1 + got) : acc0, ev);
    }
    sm.u.itv.qvT[sidx][j] = sm.Q2[sidx] * ev;
    sync_workgroup();
    RT part;
    {
      // (tab[i] - C_j = -(C_j - tab[i]): the sign goes into the scale of the table-driven kinds)
      const RT* src0 = &sm.u.itv.qvT[c < 3 ? (hf == 0 ? 0 : 6 + a) : (hf == 0 ? 3 + a : 9 + a)][0];
      const int rb0 = c < 3 ? (hf == 0 ? a : 10) : (hf == 0 ? 9 : 10);
      RT acc0, acc1 = 0;
      if constexpr (FUSED) {
        scan2(false, rb0, src0, c < 3 ? a + (hf == 0 ? 3 : 6) : rb0, c < 3 ? &sm.u.itv.qvT[hf == 0 ? 1 : 2][0] : src0, acc0, acc1);
        acc1 = c < 3 ? acc1 : (RT)0;
      } else {
        acc0 = scan(rb0, src0, j, H);
        if (c < 3) acc1 = scan(a + (hf == 0 ? 3 : 6), &sm.u.itv.qvT[hf == 0 ? 1 : 2][0], j, H);
      }
      part = c < 3 ? (hf == 0 ? -dt * (acc0 + acc1) : acc0 - dt * acc1) : (hf == 0 ? -kp * acc0 : kvv * acc0);
    }
    part += pair_swap(part);                   // both lanes of the pair: the same sum (a + b == b + a)
    if (c < 3) sm.u.itv.mv[j][a] = part;
    sync_step();
    RT g = part;
    if (c < 3) g = dt * (sm.Iw[j][a] * sm.u.itv.mv[j][0] + sm.Iw[j][3 + a] * sm.u.itv.mv[j][1] + sm.Iw[j][6 + a] * sm.u.itv.mv[j][2]);
    return g;
  };

  // ------------------------------------------------------------------ B. wrench-space Hessian row (column half)
  // Half a row of Gt against one component group of the wrench (torque lanes: tau, force lanes: F) over the
  // steps j2 = jb + jj of this lane's column half, laid out [b][jj]:
  // torque lane (j,a): Gt[(j,a)][(j2,b)] at b HH + jj ; force lane (j,3+a): Gt[(j,3+a)][(j2,3+a)] at a HH + jj, zeros elsewhere
  float Grow[GH];
#pragma unroll
  for (int q = 0; q < GH; ++q) Grow[q] = 0.f;
  // The torque block is a Gram matrix -- the dense horizon-block GEMM inside the condensed Hessian -- and is formed on the MATRIX
  // CORES (round 4): Gt_tt = M' M with M the (6 H - 3) x 3 H matrix whose rows are, per state step i and axis q,
  //   sqrt(2 Q_e[q]) Me[i][j][q][a]   (j < i;  the Euler angles of step i against the torque of step j: REF:165-171, 174-179)
  //   sqrt(2 Q_w[q]) dt Iw_j[q][a]    (j <= i; the angular velocity of step i)
  // in the columns (j, a).  v_mfma_f64_16x16x4_f64 takes four rows of M per instruction: a lane supplies ONE entry of them for
  // A = M' and one for B = M (the same entry on a diagonal tile); the 16 x 16 tiles (I <= J) of the symmetric tiling go round
  // the waves.  In f64: the row a lane keeps is f32, but it is also the operator of the carried gradient's increments, and
  // what a batch's worst instances end at follows its accuracy -- accumulated in f32 (vector loop or v_mfma_f32_32x32x2_f32 alike)
  // the at-scale maxima of config 5 rose from 2.5e-6 to 2.5e-5 (all controls) and from 6.5e-6 to 4.9e-5 (u0).
  {
    constexpr int N3 = GramTiles<H>::N3, NTL = GramTiles<H>::NTL, NU = GramTiles<H>::NU, NWV = Dims<H>::NWV;
    const int wv = l >> 6, ln = l & 63;
    const int kq = ln >> 4;                    // which of the four rows of M of a step this lane supplies
    const double sqe[3] = {P.sq_e[0], P.sq_e[1], P.sq_e[2]};      // (formed once on the host: six f64 square roots per lane otherwise)
    const double sqw[3] = {P.sq_w[0], P.sq_w[1], P.sq_w[2]};
    const float* MeF = &sm.Me[0][0];
    // The row k of M a lane supplies advances by four per instruction: with k = 3 i' + q, k + 4 = 3 (i' + 1) + (q + 1), so the step
    // and the axis are carried (q + 1, wrapped) instead of divided out, and what does not depend on the row -- the column's step
    // and axis, the three scaled entries of I_w a column can contribute -- is formed once per tile.  (The first version divided k
    // and the column by three for every entry: 550 cycles per instruction, 16.6 k / 38 k / 63 k cycles of the set-up at h = 10 / 16 / 20.)
    const int kq3 = kq == 3 ? 1 : 0, kqq = kq == 3 ? 0 : kq;                 // kq = 3 kq3 + kqq
    if (l == 64 * (NWV - 1)) {
      // General rows of a foot block: G = Gu - mu * [rows 0..3, column 2].  Gu (the mu-free part) is the same for every step
      // and foot of the instance and lives in LDS (plus its transpose); a lane keeps only the mu term it needs.  Formed here by
      // one lane of the LAST wave, which has the fewest tiles below and would wait for the others at the barrier.
      const RT s0 = sm.Gu[0][0], c0 = sm.Gu[0][1], s1 = sm.Gu[0][2], c1 = sm.Gu[0][3], s2 = sm.Gu[0][4], c2 = sm.Gu[0][5];
      const float ey[3] = {(float)(c2 * s1 * s0 - s2 * c0), (float)(s2 * s1 * s0 + c2 * c0), (float)(c1 * s0)};
      const float ez[3] = {(float)(c2 * s1 * c0 + s2 * s0), (float)(s2 * s1 * c0 - c2 * s0), (float)(c1 * c0)};
#pragma unroll
      for (int a = 0; a < 3; ++a) { sm.eyz[a] = ey[a]; sm.eyz[3 + a] = ez[a]; }
      float G[6][6];
      general_rows(0.f, ey, ez, (float)P.lh, (float)P.lt, G);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int b2 = 0; b2 < 6; ++b2) { sm.Gu[r][b2] = (RT)G[r][b2]; sm.GuT[b2][r] = (RT)G[r][b2]; }
    }
#pragma unroll 1
    for (int t = wv; t < NU; t += NWV) {       // (wave-uniform)
      int ti = 0, tj = t;                      // tile (ti <= tj) number t of the row-major list of the upper triangle
      while (tj >= NTL - ti) { tj -= NTL - ti; ++ti; }
      tj += ti;
      const int colA = 16 * ti + (ln & 15), colB = 16 * tj + (ln & 15);
      const bool diag = ti == tj;
      const int jA = colA / 3, aA = colA - 3 * jA, jB = colB / 3, aB = colB - 3 * jB;
      const bool cA = colA < N3, cB = colB < N3;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      {                                         // Euler rows k = 3 (i - 1) + q: entry sqrt(2 Q_e[q]) Me[i][j][q][a] for j < i < H
        int i = 1 + kq3, q = kqq, tri = kq3;     // tri = i (i - 1) / 2: the first (i, j) pair of step i in Me
        const int offA = 9 * jA + aA, offB = 9 * jB + aB;
#pragma unroll
        for (int k0 = 0; k0 < 3 * (H - 1); k0 += 4) {
          const double sc = q == 0 ? sqe[0] : (q == 1 ? sqe[1] : sqe[2]);
          const int base = 9 * tri + 3 * q;
          const bool okA = cA && i < H && jA < i, okB = cB && i < H && jB < i;
          const double va = (double)MeF[okA ? base + offA : 0], vb = (double)MeF[okB ? base + offB : 0];
          const double av = okA ? va * sc : 0.0;
          const double bv = diag ? av : (okB ? vb * sc : 0.0);
          acc = mfma_f64_16x16x4(av, bv, acc);
          ++q; tri += i; ++i;
          if (q == 3) { q = 0; tri += i; ++i; }
        }
      }
      {
        // Angular-velocity rows k = 3 i + q: entry sqrt(2 Q_w[q]) dt Iw_j[q][a] for j <= i < H -- the same three rows for every
        // state step i the column has reached.  Their Gram matrix is the rank-3 product of those rows times the number of steps
        // both columns have reached, H - max(j_A, j_B): ONE instruction (lane group q supplies row q, the fourth group zeros) and a
        // count per accumulator entry, instead of 3 H / 4 instructions of masked repeats.
        const int qw = kq == 3 ? 0 : kq;
        const double ua = sm.Iw[cA ? jA : 0][3 * qw + aA] * (qw == 0 ? sqw[0] : (qw == 1 ? sqw[1] : sqw[2]));
        const double ub = sm.Iw[cB ? jB : 0][3 * qw + aB] * (qw == 0 ? sqw[0] : (qw == 1 ? sqw[1] : sqw[2]));
        const double av = (cA && kq < 3) ? ua : 0.0;
        const double bv = diag ? av : ((cB && kq < 3) ? ub : 0.0);
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
        const f64x4 accw = mfma_f64_16x16x4(av, bv, zero);
#pragma unroll
        for (int v = 0; v < 4; ++v) {             // accumulator entry v of this lane: tile row kq + 4 v, tile column ln & 15
          const int jr = (16 * ti + kq + 4 * v) / 3;
          const int cnt = H - (jr > jB ? jr : jB);      // (rows and columns past 3 H hold zeros: whatever the count)
          acc[v] = fma((double)cnt, accw[v], acc[v]);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {              // accumulator entry v: tile row kq + 4 v, tile column ln & 15
        const float gv = (float)acc[v];
        sm.u.gram.g[16 * ti + kq + 4 * v][16 * tj + (ln & 15)] = gv;
        if (!diag) sm.u.gram.g[16 * tj + (ln & 15)][16 * ti + kq + 4 * v] = gv;      // (wave-uniform; a diagonal tile holds both triangles)
      }
    }
  }
  sync_workgroup();
  if (valid) {
    if (c < 3) {
      const int a = c;
      const int r = 3 * j + a;                  // this lane's torque row
      const float* grow = &sm.u.gram.g[r][3 * jb];         // its column half: 3 HH consecutive entries
#pragma unroll
      for (int jj = 0; jj < HH; ++jj) {
        const int j2 = jb + jj;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const float gv = grow[3 * jj + b];
          Grow[b * HH + jj] = gv;
          if (dbg.Gt && real) dbg.Gt[((size_t)inst * NW + row) * NW + 6 * j2 + b] = (double)gv;     // view of the row (tests)
        }
      }
    } else {
      const int a = c - 3;
      const RT kp = (RT)P.kpm, kvv = (RT)P.kvm;
#pragma unroll
      for (int jj = 0; jj < HH; ++jj) {
        const int j2 = jb + jj;
        // sum_{i = mx}^{H-1} (i - j)(i - j2), closed form: with n terms and offsets d1, d2 (one of them 0)
        const int mx = j > j2 ? j : j2;
        const int n = H - mx, d1 = mx - j, d2 = mx - j2;
        const int s2 = n * d1 * d2 + (d1 + d2) * (n * (n - 1) / 2) + (n - 1) * n * (2 * n - 1) / 6;
        const RT gval = 2 * ((RT)P.Q[3 + a] * kp * kp * (RT)s2 + (RT)P.Q[9 + a] * kvv * kvv * (RT)n);
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) Grow[a2 * HH + jj] = (a2 == a) ? (float)gval : 0.f;
        if (dbg.Gt && real) dbg.Gt[((size_t)inst * NW + row) * NW + 6 * j2 + 3 + a] = gval;
      }
    }
  }
  // (gdiag itself is not kept: factor() picks Gt[row][row] out of the row half again -- one register less across the loop)
  sync_workgroup();                            // (the accumulator tiles share their LDS with the exchange vectors of the gradient)
  // qt = 2 Gam_t' Q (s - x_ref): the exact gradient at u = 0
  const RT qt = gradient_exact(false);
  if (dbg.qt && hf == 0 && real) dbg.qt[(size_t)inst * NW + row] = (double)qt;
  if (dbg.assemble_only) return;
  if constexpr (PROF) t_setup = clock64() - t_start;

  // ------------------------------------------------------------------ C. constraint data (own foot)
  // (the mu-free general rows Gu / GuT and the body axes were stored by lane 0 in phase A)
  // (f32 values, widened at their points of use: as f64 they would hold twice the registers across the loop)
  float lb, ub, R2v;
  bool eqb;
  float cmu;                                  // -mu_f if this lane's variable is f_z (column 2 of the friction rows)
  float drf[3];                               // r_0 - r_1 of this step
  {
    {
      const float cont = (float)contact[((size_t)inst * H + j) * 2 + f];
      const float muf = mu_in ? mu_in[((size_t)inst * H + j) * 2 + f] : (float)P.mu;
      if (valid && c == 0) sm.muf[j][f] = muf;
      const int a = c < 3 ? c : c - 3;
      const float ubf = cont * (float)(c < 3 ? P.f_max[a] : P.tau_max[a]);      // REF:240-249
      const float lbf = cont * (float)(c < 3 ? P.f_min[a] : P.tau_min[a]);
      ub = ubf;
      lb = lbf;
      eqb = lbf == ubf;
      R2v = (float)(c < 3 ? P.R2[3 * f + a] : P.R2[6 + 3 * f + a]);
      cmu = c == 2 ? -muf : 0.f;
    }
#pragma unroll
    for (int a2 = 0; a2 < 3; ++a2) drf[a2] = (float)sm.rr[j][0][a2] - (float)sm.rr[j][1][a2];
    if (valid) {
      const int a3 = c < 3 ? c : c - 3;
      const int i1 = a3 == 2 ? 0 : a3 + 1, i2 = a3 == 0 ? 2 : a3 - 1;
      sm.rx[j][f][c][0] = c < 3 ? sm.rr[j][f][i2] : (RT)0;
      sm.rx[j][f][c][1] = c < 3 ? sm.rr[j][f][i1] : (RT)0;
    }
  }

  // ------------------------------------------------------------------ D. factor: L, Kn (and their G images), V for penalties rv
  float rvb, rvg;                             // penalties of this lane's box row / general row
  rvb = eqb ? P.rho_eq : P.rho; rvg = P.rho;
  RT irvb = (RT)1 / (RT)rvb, irvg = (RT)1 / (RT)rvg;  // reciprocals (refreshed with the penalties)
  // Half a row of -(S (Gt + F) S)^-1 after the sweep (S = Jacobi scaling to unit diagonal), as float pairs:
  // the sweep and the V mat-vec run on the packed-f32 pipe (v_pk_fma_f32: two f32 per lane and instruction)
  f2 Vr[HN / 2];
  float dsc = 1.f;                            // S[row]
#define VROW(q) Vr[(q) >> 1][(q) & 1]

  bool piv_bad = false;                       // the last sweep met a pivot below PIV_MIN (uniform: every lane reads the same pivots)
  float sweep_reg = 0.f;                      // diagonal regularisation of the next sweep (the repeat after a breakdown)
  constexpr float PIV_REG = 3.0e-5f;
  auto factor = [&]() {
    if constexpr (PROF) t_mark = clock64();
#ifdef BMPC_PROF_BLOCKS
    if constexpr (PROF) t_mark2 = t_mark;
#endif
    // 6x6 block algebra in f64 (blocks mix penalties over ~6 decades); results stored f32.
    if (valid) sm.rvg[j][f][c] = rvg;
    sync_workgroup();
    BMPC_BSTAMP(0)
    // factor-only data is rebuilt here from LDS and from an opaque copy of the component index, so that
    // none of it is hoisted out of the iteration loop (= holds registers during the iterations)
    int co = c, jo = j;
    BMPC_OPAQUE(co);
    BMPC_OPAQUE(jo);
    double mkd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mkd[k] = (co == k) ? 1.0 : 0.0;
    const float lh = (float)P.lh, lt = (float)P.lt;
    float ey[3], ez[3], rf[2][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ey[a] = sm.eyz[a]; ez[a] = sm.eyz[3 + a]; }
    const float muf = sm.muf[j][f];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int a = 0; a < 3; ++a) rf[ft][a] = (float)sm.rr[j][ft][a];
    // T = [[I, 0], [S, I]], S = [dr]x: (f2, m2) = -T (phi, nu) spans null(W).  What a lane needs of it (round 6: picked by the
    // component index, where the first version selected rows and columns of the 6x6 tables with 0 / 1 masks -- 144 f64 FMAs per
    // lane and factorisation that multiplied by zero):
    //   Scol = S[:, c] for c < 3, else 0      (column c of T below the diagonal: T[:, c] = e_c + [0; Scol])
    //   Srow = S[c - 3, :] for c >= 3, else 0 (row c of T left of the diagonal:  T[c, :] = e_c + [Srow, 0])
    //   R0row = [r_0]x[c - 3, :] for c >= 3, else 0  (row c of W_0^-T = [[0, I], [I, [r_0]x]] right of its unit entry)
    // (each formed where it is used, from the lever arms: held across the stages they would cost 36 registers)
    auto skew_pick = [&](const double (&v)[3], const bool rows, const int base, double (&o)[3]) {
      // rows: o = [v]x[co - base, :], else o = [v]x[:, co - base]; zeros if co - base is not 0, 1, 2
      const double M[3][3] = {{0.0, -v[2], v[1]}, {v[2], 0.0, -v[0]}, {-v[1], v[0], 0.0}};
      const int a = co - base;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double e0 = rows ? M[0][q] : M[q][0], e1 = rows ? M[1][q] : M[q][1], e2 = rows ? M[2][q] : M[q][2];
        o[q] = a == 0 ? e0 : (a == 1 ? e1 : (a == 2 ? e2 : 0.0));
      }
    };
    // D_f = 2R + A' diag(rv) A: row c of the own foot's block
    double m3[6];
    if (valid) {
      float G[6][6];
      general_rows(muf, ey, ez, lh, lt, G);
      double wc[6];                            // rho_r * G[r][c]
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        // column c of G_f from the mu-free table; the friction rows' f_z entry is -mu_f
        const double gc = (double)sm.GuT[co][r] - ((co == 2 && r < 4) ? (double)muf : 0.0);
        wc[r] = (double)sm.rvg[j][f][r] * gc;
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) s = fma(wc[r], (double)G[r][b], s);
        m3[b] = fma(mkd[b], (double)R2v + (double)rvb, s);
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) (f == 0 ? sm.u.fac.M0 : sm.u.fac.M1)[j][c][b] = m3[b];
    }
    sync_step();
    BMPC_BSTAMP(1)
    // One 6x6 inverse per step instead of four.  With Y = [W_0^-1; 0] (so W Y = I) and P the D-orthogonal
    // projector I - N Ka^-1 N' D:   L = D^-1 W' F = P Y,   F = (W D^-1 W')^-1 = Y' D L.  In blocks, with
    // B = T' D1 T (Ka = D0 + B) and I - Ka^-1 D0 = Ka^-1 B (no cancellation):
    //   L_0 = Ka^-1 B W_0^-1,   L_1 = T Ka^-1 D0 W_0^-1,   F = (W_0^-T D0) L_0,
    // W_0^-1 = [[0, I], [I, -[r_0]x]].
    // The two lanes of a row share every product by COLUMNS (round 6): lane hf forms the entries b = 3 hf .. 3 hf + 2 of row c
    // of each of them -- until round 5 lane 0 ran the Ka chain (B, L_0, F) and lane 1 the D0 chain (L_1, T Ka^-1) under
    // complementary exec masks, so a wave issued both chains with half its lanes idle.  The sums keep the order of the dense
    // 6-term form (rows of zeros skipped: fma(0, x, s) = s), so the factors are the same to the bit.  (the 6x6 inverse and the
    // two products Ka^-1 B / Ka^-1 D0, one per lane, stay whole rows: different operands, nothing to share)
    const int b0 = 3 * hf;                       // first column of this lane's half
    // entries b0 .. b0 + 2 of row q of a 6x6 block in LDS (8-byte aligned)
    auto row3 = [&](const double (*M)[6], const int q, double (&o)[3]) {
      o[0] = M[q][b0]; o[1] = M[q][b0 + 1]; o[2] = M[q][b0 + 2];
    };
    double urh[3];                              // row c of U = W_0^-T D0, own columns
    {
      double dc[3], d3[3], d4[3], d5[3], e0[3], e3[3], e4[3], e5[3], ep[3];
      const int pu = co < 3 ? co + 3 : co - 3;  // the unit entry of row c of W_0^-T
      row3(sm.u.fac.M1[j], co, dc); row3(sm.u.fac.M1[j], 3, d3); row3(sm.u.fac.M1[j], 4, d4); row3(sm.u.fac.M1[j], 5, d5);
      row3(sm.u.fac.M0[j], co, e0); row3(sm.u.fac.M0[j], pu, ep);
      row3(sm.u.fac.M0[j], 3, e3); row3(sm.u.fac.M0[j], 4, e4); row3(sm.u.fac.M0[j], 5, e5);
      BMPC_SCHED_BARRIER();
      const double dr[3] = {(double)rf[0][0] - rf[1][0], (double)rf[0][1] - rf[1][1], (double)rf[0][2] - rf[1][2]};
      const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
      double Scol[3], R0row[3];
      skew_pick(dr, false, 0, Scol);
      skew_pick(r0, true, 3, R0row);
      double yqh[3], oth[3];                    // row c of T' D1: own columns, the other lane's
#pragma unroll
      for (int k = 0; k < 3; ++k) yqh[k] = fma(Scol[2], d5[k], fma(Scol[1], d4[k], fma(Scol[0], d3[k], dc[k])));
#pragma unroll
      for (int k = 0; k < 3; ++k) oth[k] = pair_swap(yqh[k]);
      // row c of B = (T' D1) T: its columns 0..2 (lane 0) pick up the columns 3..5 of T' D1 through S = [dr]x, its columns 3..5
      // (lane 1: dz = 0) do not.  S[q][k] for k = 0, 1, 2 written out (the zero of the diagonal skipped).
      const double dz[3] = {hf == 0 ? dr[0] : 0.0, hf == 0 ? dr[1] : 0.0, hf == 0 ? dr[2] : 0.0};
      double br[3];
      br[0] = fma(oth[2], -dz[1], fma(oth[1], dz[2], yqh[0]));
      br[1] = fma(oth[2], dz[0], fma(oth[0], -dz[2], yqh[1]));
      br[2] = fma(oth[1], -dz[0], fma(oth[0], dz[1], yqh[2]));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sm.u.fac.Ka[j][c][b0 + k] = e0[k] + br[k];
        sm.u.fac.M2[j][c][b0 + k] = br[k];
        urh[k] = fma(R0row[2], e5[k], fma(R0row[1], e4[k], fma(R0row[0], e3[k], ep[k])));
      }
    }
    sync_step();                          // Ka, B published; D1 consumed
    BMPC_BSTAMP(2)
    double ka[6];                               // row c of Ka^-1
    inv6_row(sm.u.fac.Ka[j], co, ka);           // both lanes of the row, each for itself: no exchange
#pragma unroll
    for (int k = 0; k < 3; ++k) sm.u.fac.M1[j][c][b0 + k] = hf == 0 ? ka[k] : ka[3 + k];   // the whole Ka^-1 is needed for T Ka^-1 below
    double xk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // lane 0: row c of Ka^-1 B ; lane 1: row c of Ka^-1 D0
    row_times_mat6(ka, hf == 0 ? sm.u.fac.M2[j] : sm.u.fac.M0[j], xk);
    sync_step();                          // B, D0 consumed; Ka^-1 published
    BMPC_BSTAMP(3)
    if (valid) {
      const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
      // (v W_0^-1) for a row v = [p, q]: [q, p - q x r_0]
      double w0[6], cr[3];
      const double q1[3] = {xk[3], xk[4], xk[5]};
      cross3(q1, r0, cr);
      w0[0] = xk[3]; w0[1] = xk[4]; w0[2] = xk[5];
      w0[3] = xk[0] - cr[0]; w0[4] = xk[1] - cr[1]; w0[5] = xk[2] - cr[2];
      if (hf == 0) {
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          sm.LG[0].d[j][c][b][0] = (float)w0[b];
          sm.u.fac.M2[j][c][b] = w0[b];         // L_0 rows for F
        }
      } else {
#pragma unroll
        for (int b = 0; b < 6; ++b) sm.u.fac.M0[j][c][b] = w0[b];   // Ka^-1 D0 W_0^-1 rows for L_1
      }
    }
    sync_step();
    BMPC_BSTAMP(4)
    double fv64[6];                             // row c of F = U L_0 (both lanes)
    {
      double urow[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double o = pair_swap(urh[k]);
        urow[k] = hf == 0 ? urh[k] : o;
        urow[3 + k] = hf == 0 ? o : urh[k];
      }
      double fvh[3];
      {
        double l0[6][3];
#pragma unroll
        for (int q = 0; q < 6; ++q) row3(sm.u.fac.M2[j], q, l0[q]);
        BMPC_SCHED_BARRIER();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < 6; ++q) s = fma(urow[q], l0[q][k], s);
          fvh[k] = s;
        }
      }
      BMPC_SCHED_BARRIER();                     // (the two groups of loads one after the other: 36 + 48 registers in flight otherwise)
      {
        double m0[3][3], mc[3], k0[3][3], kc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { row3(sm.u.fac.M0[j], q, m0[q]); row3(sm.u.fac.M1[j], q, k0[q]); }
        row3(sm.u.fac.M0[j], co, mc); row3(sm.u.fac.M1[j], co, kc);
        BMPC_SCHED_BARRIER();
        const double dr[3] = {(double)rf[0][0] - rf[1][0], (double)rf[0][1] - rf[1][1], (double)rf[0][2] - rf[1][2]};
        double Srow[3];
        skew_pick(dr, true, 3, Srow);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          // L_1 = T (Ka^-1 D0 W_0^-1), T Ka^-1: row c of T is e_c + [Srow, 0]
          const double sl = fma(Srow[2], m0[2][k], fma(Srow[1], m0[1][k], Srow[0] * m0[0][k])) + mc[k];
          const double sk = fma(Srow[2], k0[2][k], fma(Srow[1], k0[1][k], Srow[0] * k0[0][k])) + kc[k];
          sm.LG[1].d[j][c][b0 + k][0] = (float)sl;
          // N Ka^-1 N' with N_0 = I, N_1 = -T is applied as N (Ka^-1 (N' r)): keep rows of Ka^-1 and T Ka^-1
          sm.KG[1].d[j][c][b0 + k][0] = (float)sk;
          sm.KG[0].d[j][c][b0 + k][0] = (float)(hf == 0 ? ka[k] : ka[3 + k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double o = pair_swap(fvh[k]);
        fv64[k] = hf == 0 ? fvh[k] : o;
        fv64[3 + k] = hf == 0 ? o : fvh[k];
      }
    }
    sync_step();
    BMPC_BSTAMP(5)
    if (valid) {                               // rows c of G_f Kn_f and G_f L_f (f32, from the stored f32 factors)
      float gr[6];                             // row c of G_f: mu-free table, -mu_f on the f_z entry of a friction row
#pragma unroll
      for (int b = 0; b < 6; ++b) gr[b] = (float)sm.Gu[co][b] - ((b == 2 && co < 4) ? muf : 0.f);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float gk = 0.f, gl = 0.f;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          gk = fmaf(gr[b], sm.KG[f].d[j][b][i][0], gk);
          gl = fmaf(gr[b], sm.LG[f].d[j][b][i][0], gl);
        }
        sm.KG[f].d[j][c][i][1] = gk;
        sm.LG[f].d[j][c][i][1] = gl;
      }
    }
    BMPC_BSTAMP(6)
    if constexpr (PROF) { const long long t = clock64(); t_blocks += t - t_mark; t_mark = t; }
    // K' row half = Gt row half + F row on the own step, scaled to unit diagonal (S K' S, S = diag(K')^-1/2:
    // every pivot of the sweep is then <= 1, which the pivot-row update below relies on).
    float fv[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) fv[b] = (float)fv64[b];     // (both lanes of the pair hold the whole row)
    {
      float fd = 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b) fd = fmaf((float)mkd[b], fv[b], fd);
      // (the regularised repeat after a pivot breakdown, see below: K' + reg diag(K'), i.e. S K' S + reg I up to the factor
      //  1 + reg; reg = 0 otherwise)
      // Gt[row][row]: in the half of the pair whose columns hold the own step (the other lane contributes 0)
      float gdiag = 0.f;
#pragma unroll
      for (int jj = 0; jj < HH; ++jj) {
        const bool own = (jb + jj == jo);
        const float g0 = Grow[jj], g1 = Grow[HH + jj], g2 = Grow[2 * HH + jj];
        const int a3 = co < 3 ? co : co - 3;
        gdiag = own ? (a3 == 0 ? g0 : (a3 == 1 ? g1 : g2)) : gdiag;
      }
      gdiag += pair_swap(gdiag);
      const float dg = gdiag + fd;
      if (sweep_reg > 0.f) {                     // (uniform)
#pragma unroll
        for (int b = 0; b < 6; ++b) fv[b] = fmaf((float)mkd[b], sweep_reg * dg, fv[b]);
      }
      dsc = rsq_approx(dg * (1.f + sweep_reg));
    }
    sm.dsc[slot<H>(row)] = dsc;                 // both lanes of the pair: same value
    // The sweep below runs in f32 on a matrix whose condition number reaches 1e6 at the reference's weights and 1e7 -- the
    // reciprocal of the f32 precision -- a decade away from them (Q x 10: 1 .. 2 instances in 16384 met a pivot that had
    // lost all its digits: NaNs, or a useless inverse and an instance that re-classifies for ever).  Every pivot is
    // therefore checked (PIV_MIN: a Schur complement of the unit-diagonal matrix below it is rounding noise), and a sweep
    // that met a bad one is repeated ONCE (the whole factorisation, by the caller: nothing is kept alive across the sweep for
    // it) on the matrix + PIV_REG I (added to the diagonal before the scaling): its pivots are then >= PIV_REG, and since the inverse only preconditions the residual
    // form (section 3) the regularisation costs that factorisation some convergence rate in the softest directions, never
    // the fixed point.  At the reference's weights no instance of the soaks takes the second pass.
    constexpr float PIV_MIN = 2.0e-6f;
#pragma unroll
    for (int q = 0; q < HN; ++q) {
      const int jj = q / 6, b = q % 6;
      // (through the opaque component index: the Gt part of the row is loop invariant, and the compiler would
      // otherwise keep a ready-made copy of it in HN registers across the iterations)
      float v;
      if (co < 3) v = (b < 3) ? Grow[(b < 3 ? b : 0) * HH + jj] : 0.f;
      else v = (b < 3) ? 0.f : Grow[(b < 3 ? 0 : b - 3) * HH + jj];   // zero unless b == c
      VROW(q) = v;
    }
#pragma unroll
    for (int jj = 0; jj < HH; ++jj) {
      const float mj = (jb + jj == jo) ? 1.f : 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b) VROW(6 * jj + b) = fmaf(mj, fv[b], VROW(6 * jj + b));
    }
    sync_workgroup();
    {
      const f2 d2 = {dsc, dsc};
#pragma unroll
      for (int q = 0; q < HN; q += 4) {
        if (q + 4 <= HN) {
          const float4 s4 = *reinterpret_cast<const float4*>(&sm.dsc[hf * HNP + q]);
          Vr[q / 2] = Vr[q / 2] * d2 * f2{s4.x, s4.y};
          Vr[q / 2 + 1] = Vr[q / 2 + 1] * d2 * f2{s4.z, s4.w};
        } else {
          const float2 s2 = *reinterpret_cast<const float2*>(&sm.dsc[hf * HNP + q]);
          Vr[q / 2] = Vr[q / 2] * d2 * f2{s2.x, s2.y};
        }
      }
    }
    float qmax = 0.f, qmin = 1.f;
    // Symmetric sweep, TWO pivots per step.  Step S = {k, k + 1}, P = V[S, S]: the lanes that hold the pivot columns
    // publish their two entries of them (= pivot rows, by symmetry), every lane fetches the 2 HN entries of its half, forms its
    // T[r, :] = V[r, S] P^-1 and updates with TWO packed FMAs per register pair:  row -= T[r, 0] row_k +
    // T[r, 1] row_k+1; the pivot rows themselves use T[r, :] = e_r - P^-1[r, :], which turns them into
    // P^-1 V[S, :] (exact up to rounding because the scaled pivots are <= 1; no second multiply, and a row is
    // never rebuilt from a column -- measured asymmetry 2e-7, same accuracy as the re-symmetrising form).
    // The entries in the columns S become T (V[r, S] P^-1) and, in the pivot block, -P^-1.
    // Two pivots per barrier and LDS round trip instead of one; the update of the NEXT pair of pivot columns
    // is done first and published at once (into the other buffer), so that its round trip overlaps with the
    // rest of this step's updates.
    constexpr int PVS = Dims<H>::PVS;            // floats per published column (two-half layout + dump slots)
    // Up to h = 18 the sweep is unrolled over the HN / 2 steps of a column half and the register file does NOT rotate (round 6):
    // column k sits in register k mod HN of the lanes of half k / HN for good, every register index below is static, and the
    // half whose lanes publish is the only thing that changes from the first pass to the second.  The rotating form (below)
    // keeps the code to three steps and pays for it per group of three: 6 DPP moves + ~24 register moves + the selects that
    // go with them, ~10 of a step's ~98 instructions at h = 10 (88.6 now).  Measured (tools/ab.py, one box, interleaved):
    // sweep 26.6 k -> 24.5 k cycles per factorisation at h = 10, kernel -2.3 % (configs 2, 4), -2.3 % at h = 16, -0.8 % at
    // h = 20 -- where the unrolled form spills 12 registers instead of 4 and the rotating one stays.  Same arithmetic in the
    // same order: iteration counts and results identical to the bit.
#ifndef BMPC_STATIC_HN
#define BMPC_STATIC_HN 54
#endif
    constexpr bool STATIC_SWEEP = HN <= BMPC_STATIC_HN;
    if constexpr (STATIC_SWEEP) {
      const int ps = slot<H>(row);              // the own row's entry in a published column
      {
        const int ws0 = hf == 0 ? ps : Dims<H>::VL + row;
        sm.piv[0][ws0] = Vr[0].x;
        sm.piv[0][PVS + ws0] = Vr[0].y;
      }
      int par = 0;
#pragma unroll 1
      for (int hh = 0; hh < 2; ++hh) {           // the column half that holds the pivots
        const int wsh = hf == hh ? ps : Dims<H>::VL + row;              // publication slot while this half publishes (else: dump)
        const int wso = hf == hh + 1 ? ps : Dims<H>::VL + row;          // ... and for the first pivots of the next half
        const int kb = hh * HN;                   // first pivot of the half
        const int pb0 = hh * HNP;                 // its slot
#pragma unroll
        for (int u = 0; u < HN; u += 2) {
          const float* bA = sm.piv[par];                   // column k = kb + u
          const float* bB = bA + PVS;                      // column k + 1
          float* nA = sm.piv[par ^ 1];
          par ^= 1;
          const int un = u + 2 < HN ? (u >> 1) + 1 : 0;  // register pair of the next pivot columns (last step of a half: the other half's first)
          sync_workgroup();
          const float2 pk = *reinterpret_cast<const float2*>(&bA[pb0 + u]);      // V[k][k], V[k + 1][k]
          const float p11 = bB[pb0 + u + 1];
          const float c0 = bA[ps], c1 = bB[ps];            // V[r][k], V[r][k + 1]
          BMPC_SCHED_BARRIER();                            // the step's scalar loads are in flight before anything is used
          const float id = rcp_approx(pk.x * p11 - pk.y * pk.y);
          const float q00 = p11 * id, q01 = -pk.y * id, q11 = pk.x * id;  // P^-1
          qmax = fmaxf(qmax, fmaxf(q00, q11));             // (pivot check: see the rotating form)
          qmin = fminf(qmin, fminf(q00, q11));
          const bool is0 = (row == kb + u), is1 = (row == kb + u + 1);
          float t0 = c0 * q00 + c1 * q01, t1 = c0 * q01 + c1 * q11;
          t0 = is0 ? 1.f - q00 : (is1 ? -q01 : t0);
          t1 = is0 ? -q01 : (is1 ? 1.f - q11 : t1);
          const f2 m0 = {-t0, -t0}, m1 = {-t1, -t1};
          // the pivot rows are fetched in chunks of at most CH entries each (registers), the chunk with the next pivot
          // columns first: they are updated first and published at once, so that their round trip overlaps with the rest
          constexpr int CH = HN <= 32 ? HN : 16;
          static_assert(CH % 4 == 0 || CH == HN, "chunk of whole float4s");
          constexpr int NCH = (HN + CH - 1) / CH;
          const int ch0 = (2 * un) / CH;                    // (static: the loops are unrolled)
#pragma unroll
          for (int cc = 0; cc < NCH; ++cc) {
            const int ci = cc == 0 ? ch0 : (cc <= ch0 ? cc - 1 : cc);       // chunk ch0 first, the others in order
            const int c0i = ci * CH;
            const int c1i = c0i + CH < HN ? c0i + CH : HN;
            f2 pa[CH / 2], pb[CH / 2];
#pragma unroll
            for (int q = c0i; q < c1i; q += 4) {
              if (q + 4 <= c1i) {
                const float4 a4 = *reinterpret_cast<const float4*>(&bA[hf * HNP + q]);
                const float4 b4 = *reinterpret_cast<const float4*>(&bB[hf * HNP + q]);
                pa[(q - c0i) / 2] = f2{a4.x, a4.y}; pa[(q - c0i) / 2 + 1] = f2{a4.z, a4.w};
                pb[(q - c0i) / 2] = f2{b4.x, b4.y}; pb[(q - c0i) / 2 + 1] = f2{b4.z, b4.w};
              } else {
                const float2 a2 = *reinterpret_cast<const float2*>(&bA[hf * HNP + q]);
                const float2 b2 = *reinterpret_cast<const float2*>(&bB[hf * HNP + q]);
                pa[(q - c0i) / 2] = f2{a2.x, a2.y};
                pb[(q - c0i) / 2] = f2{b2.x, b2.y};
              }
            }
            if (cc == 0) {
              Vr[un] = __builtin_elementwise_fma(m1, pb[un - c0i / 2], __builtin_elementwise_fma(m0, pa[un - c0i / 2], Vr[un]));
              const int wn = u + 2 < HN ? wsh : wso;         // (after the very last step: columns nobody reads)
              nA[wn] = Vr[un].x;
              nA[PVS + wn] = Vr[un].y;
            }
#pragma unroll
            for (int r = c0i / 2; r < c1i / 2; ++r)
              if (r != un) Vr[r] = __builtin_elementwise_fma(m1, pb[r - c0i / 2], __builtin_elementwise_fma(m0, pa[r - c0i / 2], Vr[r]));
            if (cc + 1 < NCH) BMPC_FENCE();
          }
          if (hf == hh) Vr[u >> 1] = is0 ? f2{-q00, -q01} : (is1 ? f2{-q01, -q11} : f2{t0, t1});
        }
        BMPC_DRAIN_LDS();                         // nothing in flight across the back edge (see sync_workgroup)
      }
    } else {
      // The rotating form (h = 20): groups of U = 6 pivots are unrolled; at group k0 register i of half hf holds column
      // (k0 + hf HN + i) mod NW, so the pivot columns are always the first registers of the half-0 lanes, and the register file
      // is rotated by U across the lane pair after every group.
      constexpr int U = 6;
      static_assert(NW % U == 0 && U % 2 == 0 && U + 2 <= HN, "sweep group must divide 6H and be even");
      int pos = row;                              // rotated index of the own row (group 0)
      int ws = hf == 0 ? slot<H>(pos) : Dims<H>::VL + row;
      sm.piv[0][ws] = Vr[0].x;
      sm.piv[0][PVS + ws] = Vr[0].y;
      int par = 0;                                // buffer of the current step (a group has an odd number of steps)
#pragma unroll 1
      for (int k0 = 0; k0 < NW; k0 += U) {
        const int ps = slot<H>(pos);
        int posn = pos - U;                       // ... and in the next group
        posn += (posn < 0) ? NW : 0;
        const int wsn = hf == 0 ? slot<H>(posn) : Dims<H>::VL + row;
#pragma unroll
        for (int u = 0; u < U; u += 2) {
          const float* bA = sm.piv[par];                   // column k
          const float* bB = bA + PVS;                      // column k + 1
          float* nA = sm.piv[par ^ 1];
          par ^= 1;
          const int un = (u >> 1) + 1;             // register pair of the next pivot columns (u + 2 == U: first of the next group)
          sync_workgroup();
          const float2 pk = *reinterpret_cast<const float2*>(&bA[u]);      // V[k][k], V[k + 1][k]
          const float p11 = bB[u + 1];
          const float c0 = bA[ps], c1 = bB[ps];            // V[r][k], V[r][k + 1]
          BMPC_SCHED_BARRIER();                            // the step's scalar loads are in flight before anything is used
          const float id = rcp_approx(pk.x * p11 - pk.y * pk.y);
          const float q00 = p11 * id, q01 = -pk.y * id, q11 = pk.x * id;  // P^-1
          // (pivot check, two instructions: the diagonal of P^-1 holds the reciprocals of the two pivots -- of the second one
          //  and of the first one's Schur complement against it; both must be positive -- rounding can turn a lost pivot
          //  negative -- and below 1 / PIV_MIN.  Every lane reads the same values.)
          qmax = fmaxf(qmax, fmaxf(q00, q11));
          qmin = fminf(qmin, fminf(q00, q11));
          const bool is0 = (row == k0 + u), is1 = (row == k0 + u + 1);
          float t0 = c0 * q00 + c1 * q01, t1 = c0 * q01 + c1 * q11;
          t0 = is0 ? 1.f - q00 : (is1 ? -q01 : t0);
          t1 = is0 ? -q01 : (is1 ? 1.f - q11 : t1);
          const f2 m0 = {-t0, -t0}, m1 = {-t1, -t1};
          // the pivot rows are fetched in chunks of at most CH entries each (registers), the chunk with the
          // next pivot columns first
          constexpr int CH = HN <= 32 ? HN : 16;
          static_assert(CH % 4 == 0 || CH == HN, "chunk of whole float4s");
#pragma unroll
          for (int c0i = 0; c0i < HN; c0i += CH) {
            const int c1i = c0i + CH < HN ? c0i + CH : HN;
            f2 pa[CH / 2], pb[CH / 2];
#pragma unroll
            for (int q = c0i; q < c1i; q += 4) {
              if (q + 4 <= c1i) {
                const float4 a4 = *reinterpret_cast<const float4*>(&bA[hf * HNP + q]);
                const float4 b4 = *reinterpret_cast<const float4*>(&bB[hf * HNP + q]);
                pa[(q - c0i) / 2] = f2{a4.x, a4.y}; pa[(q - c0i) / 2 + 1] = f2{a4.z, a4.w};
                pb[(q - c0i) / 2] = f2{b4.x, b4.y}; pb[(q - c0i) / 2 + 1] = f2{b4.z, b4.w};
              } else {
                const float2 a2 = *reinterpret_cast<const float2*>(&bA[hf * HNP + q]);
                const float2 b2 = *reinterpret_cast<const float2*>(&bB[hf * HNP + q]);
                pa[(q - c0i) / 2] = f2{a2.x, a2.y};
                pb[(q - c0i) / 2] = f2{b2.x, b2.y};
              }
            }
            if (c0i == 0) {
              static_assert(U / 2 + 1 <= CH / 2, "the next pivot pair lies in the first chunk");
              Vr[un] = __builtin_elementwise_fma(m1, pb[un], __builtin_elementwise_fma(m0, pa[un], Vr[un]));
              const int wn = u + 2 < U ? ws : wsn;         // (after the very last step: columns nobody reads)
              nA[wn] = Vr[un].x;
              nA[PVS + wn] = Vr[un].y;
            }
#pragma unroll
            for (int r = c0i / 2; r < c1i / 2; ++r)
              if (r != un) Vr[r] = __builtin_elementwise_fma(m1, pb[r - c0i / 2], __builtin_elementwise_fma(m0, pa[r - c0i / 2], Vr[r]));
            if (c1i < HN) BMPC_FENCE();
          }
          if (hf == 0) Vr[u >> 1] = is0 ? f2{-q00, -q01} : (is1 ? f2{-q01, -q11} : f2{t0, t1});
        }
        {                                        // rotate left by U across the pair
          f2 tmp[U / 2];
#pragma unroll
          for (int u = 0; u < U / 2; ++u) tmp[u] = pair_swap(Vr[u]);
#pragma unroll
          for (int r = 0; r + U / 2 < HN / 2; ++r) Vr[r] = Vr[r + U / 2];
#pragma unroll
          for (int u = 0; u < U / 2; ++u) Vr[HN / 2 - U / 2 + u] = tmp[u];
        }
        pos = posn;
        ws = wsn;
        BMPC_DRAIN_LDS();                         // nothing in flight across the back edge (see sync_workgroup)
      }
    }
    piv_bad = !(qmax < 1.f / PIV_MIN) || !(qmin > 0.f);     // (an infinite reciprocal -- a zero determinant -- fails too)
    if constexpr (PROF) t_sweep += clock64() - t_mark;
  };

  int nfac = 0;
  bool need_factor = true;

  // ------------------------------------------------------------------ E. ADMM iterations
  // A lane carries, next to its variable and rows, two products of the iterate that are linear in
  // it and so follow the relaxation x <- alpha x~ + (1 - alpha) x without an exchange:
  //   axg = (G_f x_f)[c]     : x~ = x - d with d_f = N_f Ka^-1 N' r + L_f gamma, so G_f x~ = axg - (GK tn + GL gamma)
  //   gbl = (Gt W x + qt)[row]: W N = 0 and W L = sum_f W_f D_f^-1 W_f' F = I, so W x~ = W x - gamma exactly and
  //                            gb~ = gb - Gt gamma (both lanes of the pair carry the full value)
  // The f32 rounding of these corrections vanishes with the step (r, gamma -> 0) and both carried values
  // are rebuilt exactly from x at every second stopping test, so the fixed point is unchanged.
  RT xo = 0;                                  // own variable
  RT zb = 0, zg = 0, yb = 0, yg = 0;
  RT axg = 0, gbl = qt;                       // x = 0: b = 0, gb = qt
  const RT alpha = (RT)P.alpha;
  int it = 0, status = 1;
  // counters instead of modulos; a cold start cannot pass the first test (check_every iterations in): it is skipped
  const int check_every = P.check_every > 0 ? P.check_every : 1;
  int next_check = 2 * check_every;
  int n_red = 0;
  // Secant extrapolation (Anderson acceleration with memory one) at the stopping tests: the ADMM step is a fixed-point map
  // w -> T(w) on w = (x, z, y); with g = T(w) - w of two consecutive iterations, w <- T(w) - gamma g,
  // gamma = <g - g', g> / |g - g'|^2, is the secant step on the residual.  It costs two sums in the reduction the stopping test
  // pays for anyway and takes 5 .. 7 % of the iterations AND of the factorisations off a solve (h = 10: 56.4 -> 53.4 / 5.86 ->
  // 5.57; h = 20: 87.9 -> 81.4 / 4.75 -> 4.48; the model oracle/ws_model.py agrees to 0.1); A/B on one box -2.5 % kernel time at
  // h = 10, -3.7 % at h = 20.  Memory two takes 9 % of the iterations but its 7 registers and 2.4 KB make the h = 10 kernel
  // 1.3 % SLOWER than memory one (built, A/B-timed, not kept).  The carried products A x and the gradient are linear in x and follow with the
  // same gamma.  g' is kept from the iteration before the test only; a factorisation in between drops it.
  bool aa_have = false;
  // exact rebuild of the carried products: see the stopping test
  // 10 since round 6 (20 before).  The carried gradient's increments use the f32 copy of the Gt row: a systematic error of
  // ~6e-8 |Gt| times the distance travelled since the last rebuild, which the soft directions (curvature 2R) amplify.  With
  // commanded angular rates (REF:64-69: Rot, I_w differ at every step, the torque block of Gt is full and carries the small yaw
  // inertia into every axis) a rebuild every 20 iterations left the dense family at 1.1e-5 / 1.9e-5 of the optimum on 4096
  // turning instances at h = 16 / 20 where the stage family -- exact gradient in every iteration -- ends at 8e-7 / 1.0e-6; every
  // 10: 1.1e-6 / 1.6e-6, and the BASELINE shapes improve too (config 5 u0: 6.1e-6 -> 1.6e-6).  Every 5 changes nothing more.
  // Costs one more rebuild per solve: +2.3 % at h = 10 before the rebuild's scans were fused (below: -1.6 %), +2.1 % at
  // h = 16, +1.4 % at h = 20 (profiles/r06_drift_*.txt).
#ifndef BMPC_REFRESH_ITERS
#define BMPC_REFRESH_ITERS 10
#endif
  constexpr int REFRESH_ITERS = BMPC_REFRESH_ITERS;
  int last_exact = 0;                          // iteration at which gbl, axg were last rebuilt exactly (the start: exact)
  // a stopping test that finds a residual more than FAR times its tolerance away cannot be followed by a
  // successful one check_every iterations later (the tail contracts by ~6 per 5 iterations): the next one is skipped
  constexpr float FAR = 1.0e3f;
  int next_adapt = P.adapt_every > 0 ? P.adapt_start : 0x7fffffff;
  while (next_adapt < 1) next_adapt += P.adapt_every;                          // the test runs after ++it
  int n_adapt = 0;                             // re-classifications taken (the two-rate schedule: DevParams::adapt_early)
  int prev_act = 0;                            // classes of this lane's rows at the previous re-classification (bit 0: box row, bit 1: general row)
  // (the residuals of the last stopping test go to resid_out AT the test -- eight bytes, 8.5 times per solve -- instead of living
  //  in two vector registers until the end of the kernel; an instance that never reaches a test reports zeros)
  if (l == 0 && resid_out) { resid_out[2 * inst] = 0.f; resid_out[2 * inst + 1] = 0.f; }

  // exact axg, bwl, gbl from x (exchange through LDS); all threads call
  auto refresh = [&](const bool with_gradient) {
    // (through opaque copies of the lane's indices, like gradient_exact: the addresses below are formed here, at a rebuild,
    //  instead of being kept -- or spilled -- across the iterations)
    int rw_ = row, f = hf_lane;
    BMPC_OPAQUE(rw_);
    BMPC_OPAQUE(f);
    const int j = rw_ / 6, c = rw_ - 6 * j;
    if (valid) sm.u.itv.xs[j][f][c] = xo;
    sync_step();
    if (valid) {
      RT xblk[2][6], gu[6];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int b = 0; b < 6; ++b) xblk[ft][b] = sm.u.itv.xs[j][ft][b];
#pragma unroll
      for (int b = 0; b < 6; ++b) gu[b] = sm.Gu[c][b];
      {
        RT a = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) a += gu[b] * (f == 0 ? xblk[0][b] : xblk[1][b]);
        const RT negmu = c < 4 ? -(RT)sm.muf[j][f] : (RT)0;
        axg = a + negmu * (f == 0 ? xblk[0][2] : xblk[1][2]);
      }
      {                                         // net wrench component of this row: (W x)[row] (both lanes: same value)
        RT v3[3];
        if (c < 3) {
          RT t0[3], t1[3];
          const RT r0[3] = {sm.rr[j][0][0], sm.rr[j][0][1], sm.rr[j][0][2]};
          const RT r1[3] = {sm.rr[j][1][0], sm.rr[j][1][1], sm.rr[j][1][2]};
          cross3(r0, &xblk[0][0], t0);
          cross3(r1, &xblk[1][0], t1);
#pragma unroll
          for (int k = 0; k < 3; ++k) v3[k] = t0[k] + t1[k] + xblk[0][3 + k] + xblk[1][3 + k];
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) v3[k] = xblk[0][k] + xblk[1][k];
        }
        const int c3 = c < 3 ? c : c - 3;
        sm.u.itv.bwT[c][j] = c3 == 0 ? v3[0] : (c3 == 1 ? v3[1] : v3[2]);
        // y_j = dt Iw_j tau_j (torque lanes hold the whole tau_j)
        if (c < 3) sm.u.itv.yw[c][j] = dt * (sm.Iw[j][3 * c] * v3[0] + sm.Iw[j][3 * c + 1] * v3[1] + sm.Iw[j][3 * c + 2] * v3[2]);
      }
    }
    sync_workgroup();
    if (with_gradient) gbl = gradient_exact(true);
  };

  if (warm.buf && warm.load) {                 // workgroup-uniform
    // Start from the previous solve of this batch slot: iterate and multipliers of the lane that owned the same
    // variable `shift` steps later (the last step repeats), projected onto the new bounds; penalties pulled back
    // towards rho0 so that a row whose activity changed needs one or two moves, not four.  A state that is not
    // finite (a failed solve) is ignored as a whole.
    int js = j + warm.shift;
    js = js > H - 1 ? H - 1 : js;
    const double* src = warm.buf + ((size_t)inst * NT + Dims<H>::lane_of(6 * js + c, f)) * 6;
    double wv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) wv[k] = src[k];
    const float pb0 = __int_as_float(__double2loint(wv[5])), pg0 = __int_as_float(__double2hiint(wv[5]));
    bool ok = pb0 > 0.f && pg0 > 0.f && pb0 < 3.0e38f && pg0 < 3.0e38f;
#pragma unroll
    for (int k = 0; k < 5; ++k) ok = ok && (fabs(wv[k]) < 1.0e300);
    ok = sync_workgroup_or(ok ? 0 : 1) == 0;    // all of the instance's state or none of it
    if (ok) {
      xo = wv[0];
      zb = fmin(fmax(wv[1], (RT)lb), (RT)ub);
      zg = fmin(wv[2], (RT)0);
      yb = wv[3];
      yg = wv[4];
      const float hib = c < 3 ? P.rho_hi_f : P.rho_hi_m, hig = c < 4 ? P.rho_hi_f : P.rho_hi_m;
      rvb = eqb ? P.rho_eq : fminf(fmaxf(P.rho * powf(pb0 / P.rho, warm.theta), P.rho_lo), hib);
      rvg = fminf(fmaxf(P.rho * powf(pg0 / P.rho, warm.theta), P.rho_lo), hig);
      irvb = (RT)1 / (RT)rvb; irvg = (RT)1 / (RT)rvg;
    }
    refresh(true);                             // axg, gbl of the loaded x (barriers inside: all lanes)
    if (ok && warm.adapt_start > 0 && P.adapt_every > 0) next_adapt = warm.adapt_start;
    if (ok) next_check = check_every;
  }

#pragma unroll 1
  for (it = 0; it < P.max_iter;) {
    if (need_factor) {                         // workgroup-uniform
      factor();
      if (BMPC_UNIFORM(piv_bad) && sweep_reg == 0.f) { sweep_reg = PIV_REG; continue; }   // once more, regularised (not counted)
      sweep_reg = 0.f;
      ++nfac;
      need_factor = false;
      aa_have = false;                         // another map: the stored state change belongs to the old one
    }
    if constexpr (PROF) t_last = clock64();
    // --- P0: row residuals w = y + rho (A x - z); publish them and the gradient
    RT wb = 0;
    if (valid) {
      wb = yb + widen(rvb) * (xo - zb);
      sm.u.itv.wg[j][f][c] = yg + widen(rvg) * (axg - zg);
      sm.u.itv.gb[row] = gbl;                 // both lanes carry the same value
    }
    // what P2 needs of the per-factorisation data is fetched before the barrier: its latency overlaps the wait
    float lcol[6];
    RT gut[6], rx0, rx1;
#pragma unroll
    for (int q = 0; q < 6; ++q) gut[q] = sm.GuT[c][q];
    rx0 = sm.rx[j][f][c][0]; rx1 = sm.rx[j][f][c][1];
#pragma unroll
    for (int i = 0; i < 6; ++i) lcol[i] = sm.LG[f].d[j][i][c][0];
    sync_step();
    BMPC_STAMP(0)
    // --- P2: KKT residual in control space r = W' gb + 2R x + A' w   (small at convergence)
    if (valid) {
      // W_f' g for this lane's variable: force variable a gets (g_tau x r_f)_a + g_F[a], moment variable a gets
      // g_tau[a].  One straight line for all lanes: the cyclic neighbours of a are picked by address, the lever
      // arm components come from a per-lane table that is zero for moment variables.
      const int a3 = c < 3 ? c : c - 3;
      const int i1 = a3 == 2 ? 0 : a3 + 1, i2 = a3 == 0 ? 2 : a3 - 1;
      const RT g1 = sm.u.itv.gb[6 * j + i1], g2 = sm.u.itv.gb[6 * j + i2];
      const RT gsel = sm.u.itv.gb[6 * j + (c < 3 ? c + 3 : c - 3)];
      RT wq[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) wq[q] = sm.u.itv.wg[j][f][q];
      // (every load of the phase is issued before the first dependent instruction: left to itself the
      // scheduler sinks each load next to its use and the phase pays one LDS round trip per load group)
      BMPC_SCHED_BARRIER();
      RT r = widen(R2v) * xo + wb;
#pragma unroll
      for (int q = 0; q < 6; ++q) r += gut[q] * wq[q];
      r += widen(cmu) * ((wq[0] + wq[1]) + (wq[2] + wq[3]));
      const RT wt = g1 * rx0 - g2 * rx1 + gsel;
      sm.u.itv.r32[j][f][c] = (float)(r + wt);
    }
    f2 kg[6], lg[6];                           // rows c of {Kn, G Kn} and {L, G L} of the own foot (for P5; fetched early)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      kg[i] = *reinterpret_cast<const f2*>(&sm.KG[f].d[j][c][i][0]);
      lg[i] = *reinterpret_cast<const f2*>(&sm.LG[f].d[j][c][i][0]);
    }
    sync_step();
    BMPC_STAMP(2)
    // --- P3: beta = L' r (own foot's part, summed over the pair), published scaled
    float rj[2][6];
    float bsum = 0.f;
    if (valid) {
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int i = 0; i < 6; i += 2) {
          const float2 v = *reinterpret_cast<const float2*>(&sm.u.itv.r32[j][ft][i]);
          rj[ft][i] = v.x; rj[ft][i + 1] = v.y;
        }
      BMPC_SCHED_BARRIER();
#pragma unroll
      for (int i = 0; i < 6; ++i) bsum = fmaf(lcol[i], f == 0 ? rj[0][i] : rj[1][i], bsum);
    }
    bsum += pair_swap(bsum);
    sm.u.itv.beta[slot<H>(row)] = bsum * dsc;   // both lanes of the pair: same value (a + b == b + a)
    // the part of the step that does not need gamma (off the critical path, before the barrier):
    // t = N' r = r_0 - T' r_1 ;  null-space part of d: foot 0 gets Ka^-1 t, foot 1 gets -(T Ka^-1) t
    f2 ddk = {0.f, 0.f};
    {
      float tn[6];
      const float d0 = drf[0], d1 = drf[1], d2 = drf[2];
      tn[0] = rj[0][0] - rj[1][0] + (d1 * rj[1][5] - d2 * rj[1][4]);
      tn[1] = rj[0][1] - rj[1][1] + (d2 * rj[1][3] - d0 * rj[1][5]);
      tn[2] = rj[0][2] - rj[1][2] + (d0 * rj[1][4] - d1 * rj[1][3]);
      tn[3] = rj[0][3] - rj[1][3];
      tn[4] = rj[0][4] - rj[1][4];
      tn[5] = rj[0][5] - rj[1][5];
#pragma unroll
      for (int i = 0; i < 6; ++i) ddk = __builtin_elementwise_fma(kg[i], f2{tn[i], tn[i]}, ddk);
      if (f == 1) ddk = -ddk;
    }
    sync_workgroup();
    BMPC_STAMP(3)
    // --- P4: gamma = V beta over the own column half, summed over the pair   (Vr holds -S V S)
    float gown;
    {
      f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < HN; q += 4) {
        if (q + 4 <= HN) {
          const float4 bq = *reinterpret_cast<const float4*>(&sm.u.itv.beta[hf * HNP + q]);
          a0 = __builtin_elementwise_fma(Vr[q / 2], f2{bq.x, bq.y}, a0);
          a1 = __builtin_elementwise_fma(Vr[q / 2 + 1], f2{bq.z, bq.w}, a1);
        } else {
          const float2 bq = *reinterpret_cast<const float2*>(&sm.u.itv.beta[hf * HNP + q]);
          a0 = __builtin_elementwise_fma(Vr[q / 2], f2{bq.x, bq.y}, a0);
        }
        if (HN > 32 && q % 16 == 12) BMPC_FENCE();
      }
      float part = (a0.x + a0.y) + (a1.x + a1.y);
      part += pair_swap(part);
      gown = -part * dsc;
    }
    {                                          // lane 0 of the pair: step-major copy, lane 1: the copy for the gradient increment
      float* gdst = hf == 0 ? &sm.u.itv.gam[row]
                            : &sm.u.itv.gamT[(c < 3 ? 0 : 2) + (j >= HH ? 1 : 0)][(c < 3 ? c : c - 3) * HH + (j >= HH ? j - HH : j)];
      *gdst = gown;
    }
    sync_workgroup();
    BMPC_STAMP(4)
    // --- P5: x~ = x - d, z~ = A x~ (carried), relaxation, projection, dual update
    float rp = 0.f, rs = 0.f, nz = 0.f, nx = 0.f, slw = 0.f;   // residual statistics: only where the stopping test runs
    float r0 = 0.f, nx0 = 0.f, slw0 = 0.f;                     // the same for the rows of step 0 alone (the applied control)
    const bool check_now = (it + 1 == next_check) || (it + 1 == P.max_iter);     // workgroup-uniform
    constexpr bool AA = (H != 12);
    // The applied control has a stopping test of its own.  REF:493 hands controls[0] -- and nothing else -- to the low-level
    // controller, and the tolerances below are relative to the largest force of the whole horizon (300-500 N): a first step
    // that carries 1-2 N (the body is to fall for a step: 1 instance in ~2000 of the walking configs) ended 2e-4 N off,
    // 1e-6 of the horizon's scale and 2e-4 of its own (SURVEY 8(d) measures u0 against max(1, |u0|)).  So the rows of step 0
    // are also held to U0_TOL x eps relative to max(1, |x_0|): not binding where step 0 carries its share of the load
    // (no iteration added on the standing and mixed batches), 15-150 x tighter where it does not.  (Not at h = 12: the
    // three reduction slots would cost that kernel its fourth instance per CU, like the extrapolation.)
    constexpr bool U0 = (H != 12);
    // (U0_TOL = 5: DevParams::eps_u0, slow_tol_r2_u0)
    constexpr int AA_MAX_FACTOR = 8;             // an instance still re-classifying after that is cycling between active sets: no extrapolation
#define AA_SLOT(x) (H == 12 ? 0 : (x))
    const bool aa_keep = AA && P.accel != 0 && (it + 2 == next_check);           // the iteration before a stopping test
    const bool aa_now = AA && P.accel != 0 && check_now && aa_have && (it + 1 < P.max_iter) && nfac <= AA_MAX_FACTOR;
    float aa1 = 0.f, aa2 = 0.f;
    float ginc = 0.f;
    if (valid) {
      float gm[6];
#pragma unroll
      for (int i = 0; i < 6; i += 2) {
        const float2 v = *reinterpret_cast<const float2*>(&sm.u.itv.gam[6 * j + i]);
        gm[i] = v.x; gm[i + 1] = v.y;
      }
      // gradient increment operands (4 loads) with the gamma loads, before anything is used
      const float* gsrc = &sm.u.itv.gamT[(c < 3 ? 0 : 2) + hf][0];
      constexpr int NG = 3 * HH;
      float gq[NG];
#pragma unroll
      for (int q = 0; q + 4 <= NG; q += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(&gsrc[q]);
        gq[q] = g4.x; gq[q + 1] = g4.y; gq[q + 2] = g4.z; gq[q + 3] = g4.w;
      }
      if constexpr (NG % 4 >= 2) {
        const float2 g2 = *reinterpret_cast<const float2*>(&gsrc[NG / 4 * 4]);
        gq[NG / 4 * 4] = g2.x; gq[NG / 4 * 4 + 1] = g2.y;
      }
      if constexpr (NG % 2 == 1) gq[NG - 1] = gsrc[NG - 1];
      BMPC_SCHED_BARRIER();
      RT st_pb, st_pg, st_x, st_g, st_dx;     // residual statistics inputs (used at stopping tests)
      RT d_zb, d_zg, d_yb, d_yg, d_ax;         // state change of this iteration (by-products of the update)
      {
        f2 dd = ddk;                            // {d_f[c], (G_f d_f)[c]}: the null-space part was formed in P3
#pragma unroll
        for (int i = 0; i < 6; ++i) dd = __builtin_elementwise_fma(lg[i], f2{gm[i], gm[i]}, dd);
        const float s = dd.x, sg = dd.y;
        const RT xto = xo - (RT)s;
        const RT ztg = axg - (RT)sg;
        const RT ztb = xto;
        // box row
        {
          const RT zr = alpha * ztb + (1 - alpha) * zb;
          const RT cand = zr + yb * irvb;
          const RT zn = fmin(fmax(cand, widen(lb)), widen(ub));
          d_yb = widen(rvb) * (zr - zn);
          yb += d_yb;
          d_zb = zn - zb;
          zb = zn;
          st_pb = ztb - zn;
        }
        // general row: l = -inf, u = 0
        {
          const RT zr = alpha * ztg + (1 - alpha) * zg;
          const RT cand = zr + yg * irvg;
          const RT zn = fmin(cand, (RT)0);
          d_yg = widen(rvg) * (zr - zn);
          yg += d_yg;
          d_zg = zn - zg;
          zg = zn;
          st_pg = ztg - zn;
        }
        st_x = xto; st_g = ztg; st_dx = xto - xo;
        d_ax = alpha * (ztg - axg);
        xo = alpha * xto + (1 - alpha) * xo;
        axg = alpha * ztg + (1 - alpha) * axg;
      }
      // gb follows b <- b - alpha gamma: gb -= alpha Gt gamma, the increment in f32 (it vanishes with the step)
      {
        // exactly the 3 HH entries that were written: the padding of gamT is never initialised (and the
        // factorisation scratch shares its LDS), 0 x garbage could be a NaN
        f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q + 4 <= NG; q += 4) {
          e0 = __builtin_elementwise_fma(f2{Grow[q], Grow[q + 1]}, f2{gq[q], gq[q + 1]}, e0);
          e1 = __builtin_elementwise_fma(f2{Grow[q + 2], Grow[q + 3]}, f2{gq[q + 2], gq[q + 3]}, e1);
        }
        if constexpr (NG % 4 >= 2)
          e0 = __builtin_elementwise_fma(f2{Grow[NG / 4 * 4], Grow[NG / 4 * 4 + 1]}, f2{gq[NG / 4 * 4], gq[NG / 4 * 4 + 1]}, e0);
        float etail = 0.f;
        if constexpr (NG % 2 == 1) etail = Grow[NG - 1] * gq[NG - 1];
        ginc = ((e0.x + e0.y) + (e1.x + e1.y)) + etail;
      }
      if (aa_keep || aa_now) {                  // (uniform)
        const float ag[6] = {(float)(alpha * st_dx), (float)d_zb, (float)d_zg, (float)d_yb, (float)d_yg, (float)d_ax};
        if (aa_now) {
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const float d = ag[q] - sm.aag[AA_SLOT(l)][q];
            aa1 = fmaf(d, ag[q], aa1);
            aa2 = fmaf(d, d, aa2);
          }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) sm.aag[AA_SLOT(l)][q] = ag[q];      // (own slot: read back by this lane after the reduction)
      }
      if (check_now) {
        // a real (uniform) branch: predicated, this costs ~25 instructions in every iteration
        BMPC_FENCE();
        rp = fmaxf(fabsf((float)st_pb), fabsf((float)st_pg));
        {                                       // pull of the inactive rows (see the stopping test)
          const bool actb = (zb <= widen(lb) || zb >= widen(ub)) && yb != (RT)0;     // (widen: no f64 copy of the bounds held across the loop)
          const bool actg = (zg >= (RT)0) && yg != (RT)0;
          slw = fmaxf((actb || eqb) ? 0.f : rvb * fabsf((float)st_pb), actg ? 0.f : rvg * fabsf((float)st_pg));
        }
        nz = fmaxf(fabsf((float)st_x), fabsf((float)st_g));
        rs = fabsf((float)st_dx);
        // a NaN iterate must reach the test (fmaxf drops NaNs): it is reported as an infinite norm
        nx = (st_x == st_x) ? fabsf((float)st_x) : __builtin_inff();
        if (U0 && j == 0) { r0 = fmaxf(rp, rs); nx0 = fabsf((float)st_x); slw0 = slw; }
      }
    }
    ginc += pair_swap(ginc);
    gbl -= alpha * (RT)ginc;
    if (aa_now) sm.aag[AA_SLOT(l)][6] = -(float)alpha * ginc;
    if (aa_keep) aa_have = true;
    ++it;
    BMPC_STAMP(5)
    // --- stopping test and penalty re-classification (workgroup-uniform decisions).  Both need a reduction over
    // the workgroup; a re-classification always falls on a stopping test with the default periods, and the two
    // share ONE exchange: the residual statistics and the "some penalty moves" flag are reduced together.
    const bool adapt_now = (it == next_adapt);
    const bool adapt_do = adapt_now && nfac <= P.max_refactor;
    // Damping: an instance that is still re-classifying after many rounds is cycling between active sets
    // (about one in a million at kappa = 20); smaller moves break the cycle (sqrt(kappa) after 10
    // factorisations, its square root after 16), where stopping the adaptation would leave hundreds of
    // plain-ADMM iterations.
    // Confirmation (round 5): the rows that turn late are what keeps an instance re-classifying -- a row that changed class at
    // iteration 60 walks its ladder at 70, 80, 90, three factorisations for a handful of rows, and those instances are the tail of a
    // batch.  From re-classification number confirm_from + 1 on, a row found in the SAME class as at the previous one is taken at
    // its word and moves by kappa_confirm (>= the length of a ladder: straight to its limit).  Not earlier: the classes of the first
    // iterations are wrong for a fifth of the rows, and a row sent to a limit on their word has to walk all the way back.
    auto reclassify = [&](float& nb, float& ng, int& act, const bool scheduled) {
      const float kap = nfac <= 10 ? P.kappa : (nfac <= 16 ? P.kappa_sqrt : P.kappa_qrt);
      const bool actb = (zb <= widen(lb) || zb >= widen(ub)) && yb != (RT)0;     // (widen: no f64 copy of the bounds held across the loop)
      const bool actg = (zg >= (RT)0) && yg != (RT)0;
      act = (actb ? 1 : 0) | (actg ? 2 : 0);
      const bool confirm = scheduled && P.kappa_confirm > 0.f && n_adapt >= P.confirm_from && n_adapt > 0 && nfac <= 10;   // (n_adapt: before this one)
      const int same = confirm ? ~(act ^ prev_act) : 0;
      const float kapb = (same & 1) ? P.kappa_confirm : kap, kapg = (same & 2) ? P.kappa_confirm : kap;
      // active rows move up by kappa towards their class ceiling, inactive ones down towards rho_lo
      int cq = row;                              // (opaque: the two class ceilings are picked here, six times per solve, instead of
      BMPC_OPAQUE(cq);                           //  living in registers -- or in scratch -- across the iterations)
      cq = cq % 6;
      const float hib = cq < 3 ? P.rho_hi_f : P.rho_hi_m, hig = cq < 4 ? P.rho_hi_f : P.rho_hi_m;
      nb = eqb ? P.rho_eq : (actb ? fminf(rvb * kapb, hib) : fmaxf(rvb / kapb, P.rho_lo));
      ng = actg ? fminf(rvg * kapg, hig) : fmaxf(rvg / kapg, P.rho_lo);
    };
    float chg = 0.f, nflip = 0.f;
    int act_now = 0;
    if (adapt_do) {
      float nb, ng;
      reclassify(nb, ng, act_now, true);
      chg = ((nb != rvb) | (ng != rvg)) ? 1.f : 0.f;
      // rows in another class than at the previous re-classification (the clones repeat a real lane: not counted)
      const int fl = act_now ^ prev_act;
      nflip = (real && n_adapt > 0) ? (float)((fl & 1) + ((fl >> 1) & 1)) : 0.f;
    }
    // An inactive row whose penalty is still far above the floor follows at 1 - alpha c / rho per iteration, c the curvature
    // it sees -- 0.9999 against the soft curvature 2R: the residuals are then small because the steps are, not because the
    // iterate has arrived (2 instances in 32768 stopped 1e-4 .. 2.5e-4 from the optimum that way, on either kernel family).
    // What such a row still pulls with, rho |z~ - z|, equals c |error|: bounded against the softest curvature 2 R_min it
    // bounds the error.  An instance that fails this third test does not stop; it re-classifies at once (the penalty of
    // the lagging row comes down).  `slw` is formed in P5, with the other statistics.
    // (1e-5 until round 4: the bound IS the accuracy this test enforces, and single instances of a batch sat right at it --
    //  1.0e-5 on 1 of 8192 standing instances, 3.5e-6 .. 5.6e-6 on the other shapes; at 1e-6 the batch maxima are 4e-7 ..
    //  2.6e-6 on both families for +0.0 .. 0.14 iterations: rows at the floor still pass it by two decades)
    // (SLOW_TOL = 1e-6: DevParams::slow_tol_r2 = SLOW_TOL 2 R_min, formed on the host)
    constexpr float AA_GAMMA_MAX = 100.f;       // a secant step beyond the one of a 0.99 contraction is not trusted
    bool force_adapt = false;
    float flips = 0.f;                          // rows of the instance that changed class since the previous re-classification
    if (check_now || adapt_do) {
      float v5[8] = {rp, rs, nz, nx, chg, slw, aa1, aa2};
      float u0v[3] = {0.f, 0.f, 0.f};           // step 0: residuals, norm, pull of its inactive rows
      long long t_r0 = 0;
      if constexpr (PROF) t_r0 = clock64();
      if constexpr (AA) {
        float v12[12] = {rp, rs, nz, nx, chg, slw, r0, nx0, slw0, aa1, aa2, nflip};
        block_max_sum<NT, 9, 3, 3>(v12, sm.red[n_red & 1]);
#pragma unroll
        for (int q = 0; q < 6; ++q) v5[q] = v12[q];
        u0v[0] = v12[6]; u0v[1] = v12[7]; u0v[2] = v12[8];
        v5[6] = v12[9]; v5[7] = v12[10];
        flips = v12[11];
      } else {
        float v6[6] = {rp, rs, nz, nx, chg, slw};
        block_max<NT, 6>(v6, sm.red[n_red & 1]);
#pragma unroll
        for (int q = 0; q < 6; ++q) v5[q] = v6[q];
      }
      ++n_red;
      if constexpr (PROF) t_red += clock64() - t_r0;
      if (check_now) {
        if (l == 0 && resid_out) { resid_out[2 * inst] = v5[0]; resid_out[2 * inst + 1] = v5[1]; }
        const float tol_p = P.eps_pri * fmaxf(1.f, v5[2]), tol_s = P.eps_dua * fmaxf(1.f, v5[3]);
        const bool bad = !(v5[0] == v5[0]) || !(v5[1] == v5[1]) || !(v5[3] < 3.0e38f);
        const float n0 = fmaxf(1.f, u0v[1]);
        const bool small = v5[0] <= tol_p && v5[1] <= tol_s && u0v[0] <= P.eps_u0 * n0;
        const bool far = v5[0] > FAR * tol_p || v5[1] > FAR * tol_s;
        // The third test can only be answered by a re-classification (the lagging row's penalty comes down); an instance
        // that may not re-classify any more -- budget of factorisations spent, or adaptation switched off -- is taken as
        // it is rather than held back until the iteration cap.
        const bool can_adapt = nfac <= P.max_refactor && P.adapt_every > 0;
        const bool slow_ok = !(v5[5] > P.slow_tol_r2 * fmaxf(1.f, v5[3]) || u0v[2] > P.slow_tol_r2_u0 * n0) || !can_adapt;
        // The exact rebuild of the carried products.  Their f32 increments drift by ~1e-7 of the distance travelled, and a
        // correction of that size moves the soft directions (curvature 2R against 1e2) by ~1e-5: it has to come while the
        // iteration is still further away than that, or the tail pays for it.  So the first stopping test that finds the
        // residuals within NEAR of their tolerances rebuilds -- one test before the tests turn dense (FAR) --, from there on
        // no carried value is older than REFRESH_ITERS iterations, and an instance stops only on values rebuilt at most
        // REFRESH_ITERS + 2 tests ago (a jump from far away to within the tolerances in one test -- the secant step can do
        // that -- rebuilds and goes on).  Measured (MI355X, configs 2 / 3 / 5): NEAR = 1e4 keeps the iteration counts of a
        // rebuild every 20 iterations (53.4 / 64.7 / 81.6) with 1.2 .. 1.4 rebuilds per solve instead of 2.7 .. 4;
        // NEAR = FAR = 1e3: +1 .. 2.4 % iterations; 1e5, 1e6: more rebuilds, nothing gained.
        constexpr float NEAR = 1.0e4f;
        const bool nearby = !(v5[0] > NEAR * tol_p || v5[1] > NEAR * tol_s);
        const int age = it - last_exact;
        // (and every FAR_REFRESH iterations wherever the instance is: one that re-classifies for hundreds of iterations --
        //  1 in 10^4 a decade away from the reference's weights -- otherwise never comes near, drifts, and then cycles for good:
        //  1 of 16384 standing instances at Q x 10 ran into the iteration cap that way, 505 iterations with this)
        constexpr int FAR_REFRESH = 100;
        const bool rebuild = (nearby && age >= REFRESH_ITERS) || age >= FAR_REFRESH;
        const bool done = small && slow_ok && age <= REFRESH_ITERS + 2 * check_every;
        force_adapt = small && !slow_ok && !bad && it < P.max_iter;
        next_check += far ? 2 * check_every : check_every;
        // (on the way out only the net wrench is needed, for the states: the iterate is what it is)
        const bool leaving = bad || done || it == P.max_iter;
        long long t_b0 = 0;
        if constexpr (PROF) t_b0 = clock64();
        if (leaving) refresh(false);
        else if (rebuild) { refresh(true); last_exact = it; ++n_reb; }
        if constexpr (PROF) { if (leaving || rebuild) t_reb += clock64() - t_b0; }
        if (bad) { status = 2; break; }
        if (done) { status = 0; break; }
      }
      if ((adapt_do && v5[4] > 0.f) || force_adapt) {   // (the new penalties are formed again rather than kept across the barrier)
        float nb, ng;
        int a2;
        reclassify(nb, ng, a2, adapt_do);
        rvb = nb; rvg = ng;
        irvb = (RT)1 / (RT)rvb; irvg = (RT)1 / (RT)rvg;
        need_factor = true;
      }
      if (aa_now) {                             // (uniform; the instance goes on)
        const float gam = v5[6] / v5[7];
        if (v5[7] > 0.f && fabsf(gam) < AA_GAMMA_MAX) {      // (NaN fails the comparison: a degenerate secant is skipped)
          const RT gr = (RT)gam;
          float ag[7];
#pragma unroll
          for (int q = 0; q < 7; ++q) ag[q] = sm.aag[AA_SLOT(l)][q];
          xo -= gr * (RT)ag[0];
          zb = fmin(fmax(zb - gr * (RT)ag[1], widen(lb)), widen(ub));
          zg = fmin(zg - gr * (RT)ag[2], (RT)0);
          yb -= gr * (RT)ag[3];
          yg -= gr * (RT)ag[4];
          axg -= gr * (RT)ag[5];
          gbl -= gr * (RT)ag[6];
        }
      }
      aa_have = false;
    }
    if (adapt_now) {
      // The schedule (round 5).  The classes are found early -- of the 240 rows of a standing h = 10 instance 45 change class
      // between iterations 10 and 20, 18 between 20 and 30, < 1 after 40 -- so the first adapt_early re-classifications come
      // adapt_every apart; later ones adapt_late apart, or adapt_busy if this one still found more than adapt_flips rows in
      // another class than the one before (the instances that keep turning are the tail of a batch: they are not made to wait).
      if (adapt_do) prev_act = act_now;
      ++n_adapt;
      int period = P.adapt_every;
      if (P.adapt_late > 0 && n_adapt >= P.adapt_early)
        period = (P.adapt_busy > 0 && (int)flips > P.adapt_flips) ? P.adapt_busy : P.adapt_late;     // (flips: a small whole number; compared as an integer,
                                                                                                       //  or (float)adapt_flips lives in a VECTOR register across the loop)
      next_adapt += period;
    }
    BMPC_STAMP(6)
    BMPC_DRAIN_LDS();                           // nothing in flight across the back edge (see sync_workgroup)
  }
  if (warm.buf && warm.store) {
    double* dst = warm.buf + ((size_t)inst * NT + l) * 6;
    dst[0] = xo; dst[1] = zb; dst[2] = zg; dst[3] = yb; dst[4] = yg;
    dst[5] = __hiloint2double(__float_as_int(rvg), __float_as_int(rvb));
  }

  // ------------------------------------------------------------------ F. outputs (REF:300-304)
  // Staged in LDS (the block-diagonal factors are dead: every wave is past the last barrier of the loop) and written out by
  // consecutive lanes, 8 / 16 bytes each: whole lines instead of the 4-byte pieces a lane's own variable makes.  For the
  // device arrays that is HBM traffic at the algorithmic bytes; for the host-mapped fp64 arrays of the host-pointer entry
  // points (WarmArgs::controls64) it is what lets the results cross PCIe while the batch runs -- written lane by lane they
  // cost a 4096-instance launch 0.10 ms (round 5, measured).
  float* ost = reinterpret_cast<float*>(&sm.LG[0]);              // [12 H] controls, then [13 H] states
  static_assert(sizeof(sm.LG) >= 25 * H * sizeof(float), "output staging does not fit the factor blocks");
  const bool want_states = states || warm.states64;
  if (real) {
    int fo = f;                                  // (opaque: 3 f is formed here, not kept from the set-up until the end of the kernel)
    BMPC_OPAQUE(fo);
    const int pos = c < 3 ? 3 * fo + c : 6 + 3 * fo + (c - 3);     // [f1 f2 m1 m2]
    ost[j * 12 + pos] = (float)xo;
  }
  if (want_states) {
    // the wrench of the final x is still in bwT (exact: every way out of the loop rebuilds it at its last
    // stopping test, and no factorisation -- which shares that LDS region -- follows); X_i = s_i + Gam_t b
    if (real) {
      float* so = ost + 12 * H + j * 13;
      const int i = j;
      if (c < 3) {
        const int a = c;
        // lane 0: euler = s + sum_{j2 < i} Me[i][j2] tau_j2 ; lane 1: omega = w_fb + dt sum_{j2 <= i} Iw_j2 tau_j2
        RT e = hf == 0 ? sm.err0[i][a] + (RT)sm.xrf[i][a] : sm.err0[i][6 + a] + (RT)sm.xrf[i][6 + a];
#pragma unroll 1
        for (int j2 = 0; j2 <= i; ++j2) {
          const RT t3[3] = {sm.u.itv.bwT[0][j2], sm.u.itv.bwT[1][j2], sm.u.itv.bwT[2][j2]};
          if (hf == 0) {
            if (j2 < i) {
              const float* m1 = sm.Me[pair_index(i, j2)];
              e += (RT)m1[3 * a] * t3[0] + (RT)m1[3 * a + 1] * t3[1] + (RT)m1[3 * a + 2] * t3[2];
            }
          } else {
            e += dt * (sm.Iw[j2][3 * a] * t3[0] + sm.Iw[j2][3 * a + 1] * t3[1] + sm.Iw[j2][3 * a + 2] * t3[2]);
          }
        }
        so[hf == 0 ? a : 6 + a] = (float)e;
      } else {
        const int a = c - 3;
        RT p = hf == 0 ? sm.err0[i][3 + a] + (RT)sm.xrf[i][3 + a] : sm.err0[i][9 + a] + (RT)sm.xrf[i][9 + a];
        const RT kp = (RT)P.kpm, kvv = (RT)P.kvm;
#pragma unroll 1
        for (int j2 = 0; j2 <= i; ++j2) {
          const RT fa = sm.u.itv.bwT[3 + a][j2];
          p += (hf == 0 ? kp * (RT)(i - j2) : kvv) * fa;
        }
        so[hf == 0 ? 3 + a : 9 + a] = (float)p;
      }
      if (c == 0 && hf == 0) so[12] = 1.0f;
    }
  }
  sync_workgroup();
  {
    // pairs of values per lane (H is even: 6 H pairs of controls, 13 H / 2 of states; an instance's arrays start 8-byte
    // aligned as fp32 and 16-byte aligned as fp64 when the caller's arrays do)
    const float2* o2 = reinterpret_cast<const float2*>(ost);
    constexpr int NC2 = 6 * H, NS2 = 13 * H / 2;
    if (warm.controls64) {
      double2* cu = reinterpret_cast<double2*>(warm.controls64 + (size_t)inst * H * 12);
      for (int q = l; q < NC2; q += NT) { const float2 v = o2[q]; cu[q] = double2{(double)v.x, (double)v.y}; }
    } else {
      float2* cu = reinterpret_cast<float2*>(controls + (size_t)inst * H * 12);
      for (int q = l; q < NC2; q += NT) cu[q] = o2[q];
    }
    if (warm.states64) {
      double2* su = reinterpret_cast<double2*>(warm.states64 + (size_t)inst * H * 13);
      for (int q = l; q < NS2; q += NT) { const float2 v = o2[NC2 + q]; su[q] = double2{(double)v.x, (double)v.y}; }
    } else if (states) {
      float2* su = reinterpret_cast<float2*>(states + (size_t)inst * H * 13);
      for (int q = l; q < NS2; q += NT) su[q] = o2[NC2 + q];
    }
  }
  if (PROF && dbg.prof && l == 0) {
    long long* pr = dbg.prof + (size_t)inst * 16;
    pr[0] = t_setup; pr[1] = t_blocks; pr[2] = t_sweep; pr[3] = clock64() - t_start; pr[4] = it; pr[5] = nfac;
#pragma unroll
    for (int k = 0; k < 7; ++k) pr[8 + k] = t_ph[k];
    pr[6] = t_red; pr[7] = t_reb; pr[15] = (long long)n_red * 1000 + n_reb;
  }
  if (l == 0) {
    if (iters_out) iters_out[inst] = it;
    if (status_out) status_out[inst] = status;
    if (nfactor_out) nfactor_out[inst] = nfac;
  }
}

#define BMPC_SOLVE_ARGS                                                                                            \
  const DevParams P, const int B, const float* __restrict__ x_fb, const float* __restrict__ foot,                  \
      const uint8_t* __restrict__ contact, const int32_t* __restrict__ phase, const float* __restrict__ x_cmd,     \
      const float* __restrict__ mu_in, float* __restrict__ controls, float* __restrict__ states,                   \
      int32_t* __restrict__ iters_out, float* __restrict__ resid_out, int32_t* __restrict__ status_out,            \
      int32_t* __restrict__ nfactor_out, const DebugOut dbg, const WarmArgs warm
template <int H>
__global__ void __launch_bounds__(Dims<H>::NT, Dims<H>::WPE) solve_kernel(BMPC_SOLVE_ARGS) {
  solve_body<H, false>(P, B, x_fb, foot, contact, phase, x_cmd, mu_in, controls, states, iters_out, resid_out, status_out,
                       nfactor_out, dbg, warm);
}
template <int H>
__global__ void __launch_bounds__(Dims<H>::NT, Dims<H>::WPE) solve_kernel_prof(BMPC_SOLVE_ARGS) {
  solve_body<H, true>(P, B, x_fb, foot, contact, phase, x_cmd, mu_in, controls, states, iters_out, resid_out, status_out,
                      nfactor_out, dbg, warm);
}
#undef BMPC_SOLVE_ARGS

}  // namespace bmpc
