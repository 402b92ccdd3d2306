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
// Thread map: lane l < 6H  <->  (step j = l / 6, component c = l % 6).  A lane owns
//   * wrench row c of step j (tau_xyz, F_xyz): one row of Gt and of V = (Gt + F)^-1 in VGPRs,
//   * control variable c of both feet at step j (v = [f(3), m(3)] per foot),
//   * the box row of those two variables and general row c (4 friction + 2 line-foot) of both feet.
// Arithmetic: data and the application of the preconditioner K^-1 (V sweep, V mat-vec, stored 6x6
// factors) are f32, the sweep and mat-vecs on the packed-f32 pipe; the 6x6 block algebra, the
// iterates and the KKT residual the preconditioner is applied to are RT (f64 by default), which is
// what pins the fixed point to the fp64 optimum.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmpc {

typedef float f2 __attribute__((ext_vector_type(2)));

// Explicit live-range splitting.  With one wave per SIMD the 256 accumulation registers are free;
// the values that merely have to survive the register-hungry factorisation are moved there by hand
// and back afterwards (2 moves per factorisation).  Left to the allocator they get an AGPR home for
// their whole life and pay a copy at every use in every iteration.
struct Parked64 { int lo, hi; };
__device__ __forceinline__ void park(float v, float& slot) {
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(slot) : "v"(v));
}
__device__ __forceinline__ void unpark(float& v, const float& slot) {
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(slot));
}
__device__ __forceinline__ void park(double v, Parked64& slot) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(slot.lo) : "v"(lo));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(slot.hi) : "v"(hi));
}
__device__ __forceinline__ void unpark(double& v, const Parked64& slot) {
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(slot.lo));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(slot.hi));
  v = __hiloint2double(hi, lo);
}
__device__ __forceinline__ void park(float v, Parked64& slot) {     // RT = float builds
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(slot.lo) : "v"(v));
}
__device__ __forceinline__ void unpark(float& v, const Parked64& slot) {
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(slot.lo));
}

struct DevParams {
  int h, half, max_iter, check_every, adapt_start, adapt_every, max_refactor, pad0;
  double dt, kv, m, g, mu, lt, lh, alpha;      // lt, lh already carry the REF:254-255 margins
  double x_cmd[12], Q[12], R2[12], Iinv[9];    // R2 = 2 R;  Iinv = inverse body inertia
  double f_max[3], f_min[3], tau_max[3], tau_min[3];
  float rho, rho_eq, rho_lo, rho_hi_f, rho_hi_m, eps_pri, eps_dua, kappa;
};

struct DebugOut {            // all nullable, fp64, device pointers
  double* x_ref;             // [B][H][12]
  double* foot_ref;          // [B][H][6]
  double* Gt;                // [B][6H][6H]
  double* qt;                // [B][6H]
  long long* prof;           // [B][16] cycle stamps (diagnostics / tools only)
  int assemble_only;
};

template <int H>
struct Dims {
  static constexpr int NW = 6 * H;                       // wrench rows = threads that work
  static constexpr int NT = ((NW + 63) / 64) * 64;       // threads per workgroup (whole waves)
  static constexpr int NPAIR = H * (H - 1) / 2;          // (i > j) step pairs
};

// LDS image of one instance.  The factor scratch (f64 6x6 blocks) and the per-iteration exchange
// vectors (+ the set-up-only step data) are never live at the same time and share one region, which
// brings h = 10 under 20 KB: eight workgroups per CU.
template <int H, typename RT>
struct alignas(16) FacScratch {
  double M0[H][6][6];        // D0 -> Ka^-1 D0 W_0^-1
  double M1[H][6][6];        // D1 -> Ka^-1
  double M2[H][6][6];        // B = T' D1 T -> L_0
  double ex[H][1][6];        // pivot-column exchange for the cooperative 6x6 sweep
};
template <int H, typename RT>
struct IterScratch {
  static constexpr int NW = Dims<H>::NW;
  RT wg[H][2][6];            // y + rho (A x - z) on the general rows
  alignas(16) RT bwT[6][H];  // net wrench of x, component-major: a lane reads its 3 H inputs of Gt contiguously
  RT gb[NW];                 // wrench-space gradient Gt b + qt
  alignas(16) float r32[H][2][6];   // KKT residual, control space
  alignas(16) float beta[NW];
  alignas(16) float gam[NW];
  // gamma again, component-major in two groups (torque, force) for the gradient increment; the second group
  // starts 4 banks after a multiple of 32 so that the two addresses of a read never share a bank
  alignas(16) float gamT[2][((3 * H + 35) / 32) * 32 + 4 > 3 * H ? ((3 * H + 31) / 32) * 32 + 4 : 3 * H];
  // set-up only
  RT Rv[H][9];               // R_inv (REF:160-164)
  RT Pre[H][9];              // prefix sums of R_inv
  RT err[H][12];             // free response - reference
};
template <int H, typename RT>
struct alignas(16) Smem {
  static constexpr int NW = Dims<H>::NW;
  union alignas(16) {
    FacScratch<H, RT> fac;
    IterScratch<H, RT> itv;
  } u;
  RT xs[H][2][6];            // x (relaxed iterate); re-read per iteration instead of living in VGPRs
  alignas(16) float piv[2][Dims<H>::NT];   // sweep pivot column, double buffered (slots >= NW: idle lanes)
  // block-diagonal part of K^-1.  Foot-major: a lane's row then sits at 24 B x lane + const, which the 32
  // LDS banks serve without conflicts (step-major interleaves the feet and collides every third step)
  // Each entry is a pair {factor, G_f x factor}: the step d and its general-row image G_f d are the same dot
  // products against (t, gamma) and run as one packed FMA per term.
  alignas(16) float LG[2][H][6][6][2];  // {L, G L}:   L_j = D^-1 W' F            [foot][step][var][wrench comp]
  alignas(16) float KG[2][H][6][6][2];  // {Kn, G Kn}: Kn = Ka^-1 (foot 0), T Ka^-1 (foot 1); N Ka^-1 N' r = N (Ka^-1 (N' r))
  // step data
  RT Iw[H][9];               // world inverse inertia
  RT rr[H][2][3];            // r_f = foot_ref - com_ref
  alignas(16) RT rx[H][2][6][2];  // per variable: {r_f[a+2], r_f[a+1]} (cyclic) for force variable a, zeros for moments
  RT s0[H][12];              // free response (X with u = 0)
  float Me[Dims<H>::NPAIR > 0 ? Dims<H>::NPAIR : 1][9];   // dt^2 (P_i - P_j) Iw_j, i > j (data: f32)
  float rvg[H][2][6];
  float muf[H][2];           // friction coefficient per step and foot
  RT Gu[6][6];               // mu-free part of the general rows of a foot block, and its transpose
  RT GuT[6][6];
  float eyz[6];              // body y and z axes in the world frame (columns 1, 2 of eul2rotm(x_fb))
  float red[4][Dims<H>::NT / 64];
};

__device__ __forceinline__ int pair_index(int i, int j) { return i * (i - 1) / 2 + j; }   // i > j

// Workgroup synchronisation point.  A one-wave workgroup needs no hardware barrier and no wait: the
// LDS executes a wave's instructions in issue order, so a ds_write is complete for all 64 lanes
// before the wave's next ds_read starts.  What remains is a compiler-level fence that keeps LDS
// accesses on their side of the point (the s_waitcnt lgkmcnt(0) that __syncthreads() would add costs
// a full LDS round trip per exchange, ~10 % of the kernel).  Larger workgroups take the real barrier.
template <int NT>
__device__ __forceinline__ void wg_sync() {
  if constexpr (NT == 64) asm volatile("" ::: "memory");
  else __syncthreads();
}

// max over the workgroup of 4 NON-NEGATIVE floats (or NaN) at once.  The order of such floats is
// the order of their bit patterns, with NaN above everything, so the reduction is an unsigned max:
// six DPP steps per value inside a wave (shifts read 0 = the neutral element where a source lane does
// not exist), then the waves of a larger workgroup combine through LDS.  NaNs propagate.
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
template <int NT>
__device__ __forceinline__ void block_max4(float (&v)[4], float (*red)[NT / 64]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = __uint_as_float(wave_umax(__float_as_uint(v[q])));
  if constexpr (NT > 64) {
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][w] = v[q];
    }
    wg_sync<NT>();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned m = __float_as_uint(red[q][0]);
#pragma unroll
      for (int w2 = 1; w2 < NT / 64; ++w2) { const unsigned o = __float_as_uint(red[q][w2]); m = m > o ? m : o; }
      v[q] = __uint_as_float(m);
    }
    wg_sync<NT>();
  }
}

// out[b] += sum_q w[q] * M[q][b] for a 6x6 f64 matrix in LDS: whole rows are fetched (three 16-byte
// reads each) and the six outputs accumulate as independent chains.
__device__ __forceinline__ void row_times_mat6(const double (&w)[6], const double (*M)[6], double (&out)[6]) {
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double row[6];
#pragma unroll
    for (int b = 0; b < 6; b += 2) {
      const double2 v = *reinterpret_cast<const double2*>(&M[q][b]);
      row[b] = v.x; row[b + 1] = v.y;
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) out[b] = fma(w[q], row[b], out[b]);
  }
}

// Cooperative symmetric sweep of NM 6x6 SPD matrices per step: lane (j, c) holds row c of each.
// On exit the rows hold the INVERSES.  All threads of the workgroup must call (barriers inside).
template <int H, typename RT, int NM>
__device__ __forceinline__ void sweep6(double (&m)[NM][6], Smem<H, RT>& sm, bool valid, int j, int c) {
  constexpr int NT = Dims<H>::NT;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (valid) {
#pragma unroll
      for (int q = 0; q < NM; ++q) sm.u.fac.ex[j][q][c] = m[q][k];
    }
    wg_sync<NT>();
    if (valid) {
#pragma unroll
      for (int q = 0; q < NM; ++q) {
        double col[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) col[b] = sm.u.fac.ex[j][q][b];
        // reciprocal by v_rcp_f64 + two Newton steps (a correctly rounded IEEE division is ~40 dependent instructions)
        double pinv = __builtin_amdgcn_rcp(col[k]);
        pinv = fma(fma(-col[k], pinv, 1.0), pinv, pinv);
        pinv = fma(fma(-col[k], pinv, 1.0), pinv, pinv);
        const bool isp = (c == k);
        const double t = isp ? -pinv : m[q][k] * pinv;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          if (b == k) continue;
          m[q][b] = isp ? col[b] * pinv : fma(-t, col[b], m[q][b]);
        }
        m[q][k] = t;            // pivot lane: -1/p ; others: a_ik / p
      }
    }
    wg_sync<NT>();
  }
  // swept matrix = -A^-1
#pragma unroll
  for (int q = 0; q < NM; ++q)
#pragma unroll
    for (int b = 0; b < 6; ++b) m[q][b] = -m[q][b];
}

__device__ __forceinline__ void sincos_rt(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void sincos_rt(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ double min_rt(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float min_rt(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double max_rt(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float max_rt(float a, float b) { return fmaxf(a, b); }

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

template <int H, typename RT>
__global__ void __launch_bounds__(Dims<H>::NT)
solve_kernel(const DevParams P, const int B,
             const float* __restrict__ x_fb, const float* __restrict__ foot,
             const uint8_t* __restrict__ contact, const int32_t* __restrict__ phase,
             const float* __restrict__ x_cmd, const float* __restrict__ mu_in,
             float* __restrict__ controls, float* __restrict__ states,
             int32_t* __restrict__ iters_out, float* __restrict__ resid_out,
             int32_t* __restrict__ status_out, int32_t* __restrict__ nfactor_out,
             const DebugOut dbg) {
  constexpr int NW = Dims<H>::NW;
  constexpr int NT = Dims<H>::NT;
  __shared__ Smem<H, RT> sm;

  const int inst = blockIdx.x;
  if (inst >= B) return;
  long long t_start = 0, t_setup = 0, t_blocks = 0, t_sweep = 0, t_mark = 0;
  long long t_ph[7] = {0, 0, 0, 0, 0, 0, 0}, t_last = 0;
#define BMPC_STAMP(k) if (dbg.prof) { const long long t_ = clock64(); t_ph[k] += t_ - t_last; t_last = t_; }
  if (dbg.prof) t_start = clock64();
  const int l = threadIdx.x;
  const bool valid = l < NW;
  const int j = valid ? l / 6 : 0;
  const int c = valid ? l % 6 : 0;
  const RT dt = (RT)P.dt;
  // 0/1 masks of this lane's component: runtime picks are done arithmetically (select chains over
  // register arrays get demoted to scratch by the compiler)
  // (the factor-only masks are rebuilt inside factor() so that they do not occupy registers during
  // the iterations)
  RT mk3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) mk3[k] = (c % 3 == k) ? (RT)1 : (RT)0;

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
  if (valid && c == 0) {                       // debug views of the references (tests)
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
    sincos_rt(xr[0], &sy, &cy);
    sincos_rt(xr[1], &sp, &cp);
    sincos_rt(xr[2], &sr, &cr);
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
    if (valid && c == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) { sm.Iw[j][q] = Iw[q]; sm.u.itv.Rv[j][q] = Rv[q]; }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int a = 0; a < 3; ++a) sm.rr[j][f][a] = fr[3 * f + a] - xr[3 + a];           // REF:174-175
    }
  }
  wg_sync<NT>();
#pragma unroll
  for (int q = 0; q < 9; ++q) Pj[q] = 0;
#pragma unroll 1
  for (int s = 0; s <= j; ++s)
#pragma unroll
    for (int q = 0; q < 9; ++q) Pj[q] += sm.u.itv.Rv[s][q];
  if (valid) {
    if (c == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) sm.u.itv.Pre[j][q] = Pj[q];
    }
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
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { sm.u.itv.err[j][i] = e12[i] - xr[i]; sm.s0[j][i] = e12[i]; }
    }
  }
  wg_sync<NT>();
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
        for (int q = 0; q < 3; ++q) s += (sm.u.itv.Pre[i][3 * a + q] - sm.u.itv.Pre[j2][3 * a + q]) * sm.Iw[j2][3 * q + b];
        sm.Me[idx][3 * a + b] = (float)(dt * dt * s);
      }
  }
  wg_sync<NT>();

  // ------------------------------------------------------------------ B. wrench-space Hessian row
  // Row of Gt against one component group of the wrench (torque lanes: tau, force lanes: F), laid out
  // [b][j2] like bwT so that P1 is the same straight-line code for every lane:
  // torque lane (j,a): Gt[(j,a)][(j2,b)] at b H + j2 ; force lane (j,3+a): Gt[(j,3+a)][(j2,3+a)] at a H + j2, zeros elsewhere
  float Grow[3 * H];
  RT qt = 0;
#pragma unroll
  for (int q = 0; q < 3 * H; ++q) Grow[q] = 0.f;
  if (valid) {
    // The step loops run over wave-uniform ranges with per-lane predicates, so the partner blocks
    // Me[i][j2] are broadcast reads and the (i, j2) work of one i is a batch of independent FMAs.
    if (c < 3) {
      const int a = c;
      RT nw[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) nw[q] = dt * sm.Iw[j][3 * q + a] * (RT)P.Q[6 + q];   // Q_w weighted column
      RT acc[3 * H];                            // sum_{i > max(j, j2)} (Me_i,j Q Me_i,j2)[a][b] at b H + j2 (then the row itself)
#pragma unroll
      for (int q = 0; q < 3 * H; ++q) acc[q] = 0;
      RT s = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) s += nw[q] * sm.u.itv.err[j][6 + q];             // i = j term of qt
#pragma unroll 1
      for (int i = 1; i < H; ++i) {             // uniform
        const bool act = i > j;
        const float* m1 = sm.Me[pair_index(i, act ? j : 0)];
        RT u[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) u[q] = act ? (RT)m1[3 * q + a] * (RT)P.Q[q] : (RT)0;
#pragma unroll
        for (int q = 0; q < 3; ++q) s += u[q] * sm.u.itv.err[i][q] + (act ? nw[q] : (RT)0) * sm.u.itv.err[i][6 + q];
#pragma unroll
        for (int j2 = 0; j2 < H - 1; ++j2) {
          if (j2 < i) {                         // uniform
            const float* m2 = sm.Me[pair_index(i, j2)];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
              for (int b = 0; b < 3; ++b) acc[b * H + j2] += u[q] * (RT)m2[3 * q + b];
          }
        }
      }
      qt = 2 * s;
#pragma unroll
      for (int j2 = 0; j2 < H; ++j2) {
        const RT cnt = (RT)(H - (j > j2 ? j : j2));
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          RT sw = 0;
#pragma unroll
          for (int q = 0; q < 3; ++q) sw += nw[q] * dt * sm.Iw[j2][3 * q + b];
          const RT gval = 2 * (acc[b * H + j2] + cnt * sw);
          Grow[b * H + j2] = (float)gval;
          acc[b * H + j2] = gval;
        }
      }
      if (dbg.Gt) {                             // fp64 view of the row (tests)
#pragma unroll
        for (int j2 = 0; j2 < H; ++j2)
#pragma unroll
          for (int b = 0; b < 3; ++b) dbg.Gt[((size_t)inst * NW + l) * NW + 6 * j2 + b] = (double)acc[b * H + j2];
      }
    } else {
      const int a = c - 3;
      const RT kp = dt * dt / (RT)P.m, kvv = dt / (RT)P.m;
      RT gdbg[H];
#pragma unroll
      for (int j2 = 0; j2 < H; ++j2) {
        // sum_{i = mx}^{H-1} (i - j)(i - j2), closed form: with n terms and offsets d1, d2 (one of them 0)
        const int mx = j > j2 ? j : j2;
        const int n = H - mx, d1 = mx - j, d2 = mx - j2;
        const int s2 = n * d1 * d2 + (d1 + d2) * (n * (n - 1) / 2) + (n - 1) * n * (2 * n - 1) / 6;
        const RT gval = 2 * ((RT)P.Q[3 + a] * kp * kp * (RT)s2 + (RT)P.Q[9 + a] * kvv * kvv * (RT)n);
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) Grow[a2 * H + j2] = (a2 == a) ? (float)gval : 0.f;
        gdbg[j2] = gval;
      }
      if (dbg.Gt) {
#pragma unroll
        for (int j2 = 0; j2 < H; ++j2) dbg.Gt[((size_t)inst * NW + l) * NW + 6 * j2 + 3 + a] = (double)gdbg[j2];
      }
      RT s = 0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const RT w1 = i >= j ? kp * (RT)(i - j) * (RT)P.Q[3 + a] : (RT)0, w2 = i >= j ? kvv * (RT)P.Q[9 + a] : (RT)0;
        s += w1 * sm.u.itv.err[i][3 + a] + w2 * sm.u.itv.err[i][9 + a];
      }
      qt = 2 * s;
    }
    if (dbg.qt) dbg.qt[(size_t)inst * NW + l] = (double)qt;
  }
  if (dbg.assemble_only) return;
  if (dbg.prof) t_setup = clock64() - t_start;

  // ------------------------------------------------------------------ C. constraint data
  // General rows of a foot block: G = Gu - mu * [rows 0..3, column 2].  Gu (the mu-free part) is the same
  // for every step and foot of the instance and lives in LDS (plus its transpose); a lane keeps only
  // the two mu terms it needs.  Coefficients used with the f64 iterates are kept as RT so that no
  // f32 copy + hoisted conversion doubles their register footprint.
  RT lb[2], ub[2], R2v[2];
  bool eqb[2];
  RT cmu[2];                                  // -mu_f if this lane's variable is f_z (column 2 of the friction rows)
  float drf[3];                               // r_0 - r_1 of this step
  {
    float ey[3], ez[3];                       // columns 1, 2 of R = eul2rotm(x_fb[0:3])  (REF:124-138, 193)
    {
      RT sr, cr, sp, cp, sy, cy;
      sincos_rt(xfb[0], &sr, &cr);
      sincos_rt(xfb[1], &sp, &cp);
      sincos_rt(xfb[2], &sy, &cy);
      ey[0] = (float)(cy * sp * sr - sy * cr); ey[1] = (float)(sy * sp * sr + cy * cr); ey[2] = (float)(cp * sr);
      ez[0] = (float)(cy * sp * cr + sy * sr); ez[1] = (float)(sy * sp * cr - cy * sr); ez[2] = (float)(cp * cr);
    }
    if (l == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { sm.eyz[a] = ey[a]; sm.eyz[3 + a] = ez[a]; }
      float G[6][6];
      general_rows(0.f, ey, ez, (float)P.lh, (float)P.lt, G);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int b2 = 0; b2 < 6; ++b2) { sm.Gu[r][b2] = (RT)G[r][b2]; sm.GuT[b2][r] = (RT)G[r][b2]; }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float cont = (float)contact[((size_t)inst * H + j) * 2 + f];
      const float muf = mu_in ? mu_in[((size_t)inst * H + j) * 2 + f] : (float)P.mu;
      if (valid && c == 0) sm.muf[j][f] = muf;
      const int a = c < 3 ? c : c - 3;
      const float ubf = cont * (float)(c < 3 ? P.f_max[a] : P.tau_max[a]);      // REF:240-249
      const float lbf = cont * (float)(c < 3 ? P.f_min[a] : P.tau_min[a]);
      ub[f] = (RT)ubf;
      lb[f] = (RT)lbf;
      eqb[f] = lbf == ubf;
      R2v[f] = (RT)(float)(c < 3 ? P.R2[3 * f + a] : P.R2[6 + 3 * f + a]);
      cmu[f] = c == 2 ? (RT)(-muf) : (RT)0;
    }
#pragma unroll
    for (int a2 = 0; a2 < 3; ++a2) drf[a2] = (float)sm.rr[j][0][a2] - (float)sm.rr[j][1][a2];
    if (valid) {
      const int a3 = c < 3 ? c : c - 3;
      const int i1 = a3 == 2 ? 0 : a3 + 1, i2 = a3 == 0 ? 2 : a3 - 1;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        sm.rx[j][f][c][0] = c < 3 ? sm.rr[j][f][i2] : (RT)0;
        sm.rx[j][f][c][1] = c < 3 ? sm.rr[j][f][i1] : (RT)0;
      }
    }
  }

  // ------------------------------------------------------------------ D. factor: L, Kn (and their G images), V for penalties rv
  RT rvb[2], rvg[2];                          // penalties of this lane's box rows / general rows (f32 values)
#pragma unroll
  for (int f = 0; f < 2; ++f) { rvb[f] = (RT)(eqb[f] ? P.rho_eq : P.rho); rvg[f] = (RT)P.rho; }
  RT irvb[2], irvg[2];                        // reciprocals (refreshed with the penalties)
#pragma unroll
  for (int f = 0; f < 2; ++f) { irvb[f] = (RT)1 / rvb[f]; irvg[f] = (RT)1 / rvg[f]; }
  // hides a loop-invariant f32 value from the optimiser at its point of use, so that its f64 conversion is
  // redone there instead of being hoisted into a second, f64, register copy that lives across the loop
#define BMPC_OPAQUE(x) asm volatile("" : "+v"(x))
  // row l of -(Gt + F)^-1 after the sweep, as float pairs: the sweep and the V mat-vec run on the
  // packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two lanes of f32 per instruction)
  f2 Vr[NW / 2];
#define VROW(q) Vr[(q) >> 1][(q) & 1]

  float gpark[3 * H];                         // Grow while the block algebra runs
  auto factor = [&]() {
    if (dbg.prof) t_mark = clock64();
    // 6x6 block algebra in f64 (blocks mix penalties over ~6 decades); results stored f32.
    // D_f = 2R + A' diag(rv) A, row c of both feet
    if (valid) {
      sm.rvg[j][0][c] = (float)rvg[0];
      sm.rvg[j][1][c] = (float)rvg[1];
    }
    wg_sync<NT>();
    // factor-only data is rebuilt here from LDS and from an opaque copy of the component index, so
    // that none of it is live (= holds registers) during the iterations
    int co = c;
    asm volatile("" : "+v"(co));
    double mkd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mkd[k] = (co == k) ? 1.0 : 0.0;
    const float lh = (float)P.lh, lt = (float)P.lt;
    float ey[3], ez[3], muf[2], rf[2][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ey[a] = sm.eyz[a]; ez[a] = sm.eyz[3 + a]; }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      muf[f] = sm.muf[j][f];
#pragma unroll
      for (int a = 0; a < 3; ++a) rf[f][a] = (float)sm.rr[j][f][a];
    }
    double m3[2][6];                           // rows of D0, D1
    double Tm[6][6];                           // T = [[I, 0], [[dr]x, I]]: (f2, m2) = -T (phi, nu) spans null(W)
    {
      const double dr[3] = {(double)rf[0][0] - rf[1][0], (double)rf[0][1] - rf[1][1], (double)rf[0][2] - rf[1][2]};
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = 0; q < 6; ++q) Tm[p][q] = (p == q) ? 1.0 : 0.0;
      Tm[3][1] = -dr[2]; Tm[3][2] = dr[1];
      Tm[4][0] = dr[2];  Tm[4][2] = -dr[0];
      Tm[5][0] = -dr[1]; Tm[5][1] = dr[0];
    }
    double Tcol[6], Trow[6];                   // T[:, c] and T[c, :]
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) { a1 = fma(mkd[cc], Tm[p][cc], a1); a2 = fma(mkd[cc], Tm[cc][p], a2); }
      Tcol[p] = a1; Trow[p] = a2;
    }
    if (valid) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float G[6][6];
        general_rows(muf[f], ey, ez, lh, lt, G);
        double wc[6];                          // rho_r * G[r][c]
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          // column c of G_f from the mu-free table; the friction rows' f_z entry is -mu_f
          const double gc = (double)sm.GuT[c][r] - ((co == 2 && r < 4) ? (double)muf[f] : 0.0);
          wc[r] = (double)sm.rvg[j][f][r] * gc;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < 6; ++r) s = fma(wc[r], (double)G[r][b], s);
          m3[f][b] = fma(mkd[b], (double)R2v[f] + (double)rvb[f], s);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) (f == 0 ? sm.u.fac.M0 : sm.u.fac.M1)[j][c][b] = m3[f][b];
      }
    }
    wg_sync<NT>();
    // One 6x6 inverse per step instead of four.  With Y = [W_0^-1; 0] (so W Y = I) and P the D-orthogonal
    // projector I - N Ka^-1 N' D:   L = D^-1 W' F = P Y,   F = (W D^-1 W')^-1 = Y' D L.  In blocks, with
    // B = T' D1 T (Ka = D0 + B) and I - Ka^-1 D0 = Ka^-1 B (no cancellation):
    //   L_0 = Ka^-1 B W_0^-1,   L_1 = T Ka^-1 D0 W_0^-1,   F = (W_0^-T D0) L_0,
    // W_0^-1 = [[0, I], [I, -[r_0]x]].
    double brow[6], urow[6];                    // rows c of B and of U = W_0^-T D0
    double ka[1][6];                            // row c of Ka -> Ka^-1
    if (valid) {
      double yq[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      row_times_mat6(Tcol, sm.u.fac.M1[j], yq);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s = fma(yq[q], Tm[q][b], s);
        brow[b] = s;
        ka[0][b] = m3[0][b] + s;
      }
      // U = W_0^-T D0 = [[0, I], [I, [r_0]x]] D0: rows 0..2 are rows 3..5 of D0, row 3+a is row a + ([r_0]x D0[3:6])_a
      double wti[6];                            // row c of W_0^-T
      {
        const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
        double Wt[6][6];
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int q = 0; q < 6; ++q) Wt[p][q] = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { Wt[a][3 + a] = 1.0; Wt[3 + a][a] = 1.0; }
        Wt[3][4] = -r0[2]; Wt[3][5] = r0[1];
        Wt[4][3] = r0[2];  Wt[4][5] = -r0[0];
        Wt[5][3] = -r0[1]; Wt[5][4] = r0[0];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double a1 = 0.0;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) a1 = fma(mkd[cc], Wt[cc][q], a1);
          wti[q] = a1;
        }
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) urow[b] = 0.0;
      row_times_mat6(wti, sm.u.fac.M0[j], urow);
#pragma unroll
      for (int b = 0; b < 6; ++b) sm.u.fac.M2[j][c][b] = brow[b];
    }
    sweep6<H, RT, 1>(ka, sm, valid, j, c);      // -> Ka^-1 (barriers inside: the M1 reads above are done)
    if (valid) {
#pragma unroll
      for (int b = 0; b < 6; ++b) sm.u.fac.M1[j][c][b] = ka[0][b];
    }
    double x1[6], x0[6];                        // rows c of Ka^-1 B and Ka^-1 D0
    if (valid) {
#pragma unroll
      for (int b = 0; b < 6; ++b) { x1[b] = 0.0; x0[b] = 0.0; }
      row_times_mat6(ka[0], sm.u.fac.M2[j], x1);
      row_times_mat6(ka[0], sm.u.fac.M0[j], x0);
    }
    wg_sync<NT>();                            // B, D0 consumed; Ka^-1 published
    double fv64[6];                             // row c of F
    if (valid) {
      const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
      // (v W_0^-1) for a row v = [p, q]: [q, p - q x r_0]
      double l0[6], y0[6];
      {
        double cr[3];
        const double q1[3] = {x1[3], x1[4], x1[5]};
        cross3(q1, r0, cr);
        l0[0] = x1[3]; l0[1] = x1[4]; l0[2] = x1[5];
        l0[3] = x1[0] - cr[0]; l0[4] = x1[1] - cr[1]; l0[5] = x1[2] - cr[2];
        const double q0[3] = {x0[3], x0[4], x0[5]};
        cross3(q0, r0, cr);
        y0[0] = x0[3]; y0[1] = x0[4]; y0[2] = x0[5];
        y0[3] = x0[0] - cr[0]; y0[4] = x0[1] - cr[1]; y0[5] = x0[2] - cr[2];
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        sm.LG[0][j][c][b][0] = (float)l0[b];
        sm.u.fac.M2[j][c][b] = l0[b];           // L_0 rows for F
        sm.u.fac.M0[j][c][b] = y0[b];           // Ka^-1 D0 W_0^-1 rows for L_1
      }
    }
    wg_sync<NT>();
    if (valid) {
      double sl[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < 6; ++b) fv64[b] = 0.0;
      row_times_mat6(Trow, sm.u.fac.M0[j], sl);     // L_1 = T (Ka^-1 D0 W_0^-1)
      row_times_mat6(urow, sm.u.fac.M2[j], fv64);   // F = U L_0
      row_times_mat6(Trow, sm.u.fac.M1[j], sk);     // T Ka^-1
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        sm.LG[1][j][c][b][0] = (float)sl[b];
        // N Ka^-1 N' with N_0 = I, N_1 = -T is applied as N (Ka^-1 (N' r)): keep rows of Ka^-1 and T Ka^-1
        sm.KG[0][j][c][b][0] = (float)ka[0][b];
        sm.KG[1][j][c][b][0] = (float)sk[b];
      }
    }
    wg_sync<NT>();
    if (valid) {                               // rows c of G_f Kn_f and G_f L_f (f32, from the stored f32 factors)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float gr[6];                           // row c of G_f: mu-free table, -mu_f on the f_z entry of a friction row
#pragma unroll
        for (int b = 0; b < 6; ++b) gr[b] = (float)sm.Gu[c][b] - ((b == 2 && co < 4) ? muf[f] : 0.f);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          float gk = 0.f, gl = 0.f;
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            gk = fmaf(gr[b], sm.KG[f][j][b][i][0], gk);
            gl = fmaf(gr[b], sm.LG[f][j][b][i][0], gl);
          }
          sm.KG[f][j][c][i][1] = gk;
          sm.LG[f][j][c][i][1] = gl;
        }
      }
    }
    if (dbg.prof) { const long long t = clock64(); t_blocks += t - t_mark; t_mark = t; }
#pragma unroll
    for (int q = 0; q < 3 * H; ++q) unpark(Grow[q], gpark[q]);
    // K' row = Gt row + F row on the own step; then symmetric sweep with a rotating register file:
    // at step k register r holds column (r + k) mod NW, so the pivot column is always register 0.
    if (valid) {
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const int j2 = q / 6, b = q % 6;
        float v;
        if (c < 3) v = (b < 3) ? Grow[(b < 3 ? b : 0) * H + j2] : 0.f;
        else v = (b < 3) ? 0.f : Grow[(b < 3 ? 0 : b - 3) * H + j2];   // zero unless b == c
        VROW(q) = v;
      }
      float fv[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) fv[b] = (float)fv64[b];
#pragma unroll
      for (int j2 = 0; j2 < H; ++j2) {
        const float mj = (j2 == j) ? 1.f : 0.f;
#pragma unroll
        for (int b = 0; b < 6; ++b) VROW(6 * j2 + b) = fmaf(mj, fv[b], VROW(6 * j2 + b));
      }
    }
    // Groups of U steps are unrolled so the pivot column sits in the static register u; the register
    // file is rotated by U once per group.  Pivot lane: row = column / p; other lanes: row -= (a_ik/p) * pivot row.
    constexpr int U = 6;
    static_assert(NW % U == 0 && U % 2 == 0, "sweep group must divide 6H and be even");
#pragma unroll 1
    // The idle lanes of the last wave run the sweep too (no exec-mask juggling per pivot): they publish
    // into slots nobody reads and update a row nobody uses.
    for (int k0 = 0; k0 < NW; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float* buf = sm.piv[u & 1];
        {
          int pos = l - k0;
          pos += (pos < 0) ? NW : 0;
          buf[valid ? pos : l] = VROW(u);
        }
        wg_sync<NT>();
        {
          // fetch the pivot vector first (back-to-back ds_read_b128, one wait), then compute: with one
          // wave per SIMD nothing else hides the LDS latency.  (Chunked fetch costs less registers.)
#ifdef BMPC_PIVOT_CHUNKS                     // for builds that target two waves per SIMD (256 registers)
          constexpr int PB = (NW % 24 == 0) ? 24 : 20;
#else
          constexpr int PB = NW;
#endif
          static_assert(NW % PB == 0 && PB % 4 == 0 && U <= PB, "pivot chunk");
          const float ci = VROW(u);
          const bool isp = (l == k0 + u);
          // the pivot lane rebuilds its row from the published COLUMN (buf), which re-symmetrises the
          // matrix at every pivot; scaling its own row instead lets f32 asymmetry grow and diverge
          const float sc = isp ? 0.f : 1.f;
          f2 t2 = {0.f, 0.f};
          const f2 sc2 = {sc, sc};
          float t = 0.f;
#pragma unroll
          for (int r0 = 0; r0 < NW; r0 += PB) {
            f2 pb[PB / 2];
#pragma unroll
            for (int r = 0; r < PB; r += 4) {
              const float4 q4 = *reinterpret_cast<const float4*>(&buf[r0 + r]);
              pb[r / 2] = f2{q4.x, q4.y};
              pb[r / 2 + 1] = f2{q4.z, q4.w};
            }
            if (r0 == 0) {
              const float pinv = __builtin_amdgcn_rcpf(pb[u >> 1][u & 1]);
              t = isp ? -pinv : ci * pinv;
              t2 = f2{t, t};
            }
#pragma unroll
            for (int r = 0; r < PB / 2; ++r)
              Vr[r0 / 2 + r] = __builtin_elementwise_fma(-t2, pb[r], sc2 * Vr[r0 / 2 + r]);
          }
          VROW(u) = t;
        }
      }
      {                                        // rotate left by U
        f2 tmp[U / 2];
#pragma unroll
        for (int u = 0; u < U / 2; ++u) tmp[u] = Vr[u];
#pragma unroll
        for (int r = 0; r + U / 2 < NW / 2; ++r) Vr[r] = Vr[r + U / 2];
#pragma unroll
        for (int u = 0; u < U / 2; ++u) Vr[NW / 2 - U / 2 + u] = tmp[u];
      }
    }
    if (dbg.prof) t_sweep += clock64() - t_mark;
  };

  int nfac = 0;
  bool need_factor = true;

  // ------------------------------------------------------------------ E. ADMM iterations
  // A lane carries, next to its variables and rows, two products of the iterate that are linear in
  // it and so follow the relaxation x <- alpha x~ + (1 - alpha) x without an exchange:
  //   axg_f = (G_f x_f)[c]   : x~ = x - d with d_f = N_f Ka^-1 N' r + L_f gamma, so G_f x~ = axg - (GK tn + GL gamma)
  //   bwl   = (W x)[l]       : W N = 0 and W L = sum_f W_f D_f^-1 W_f' F = I, so W x~ = bw - gamma exactly
  // The f32 rounding of these corrections vanishes with the step (r, gamma -> 0) and both carried values
  // are rebuilt exactly from x at every stopping test, so the fixed point is unchanged.
  RT xo[2] = {0, 0};                          // own variables
  RT zb[2] = {0, 0}, zg[2] = {0, 0}, yb[2] = {0, 0}, yg[2] = {0, 0};
  RT axg[2] = {0, 0}, bwl = 0, gbl = qt;     // x = 0: b = 0, gb = qt
  const RT alpha = (RT)P.alpha;
  int it = 0, status = 1;
  int next_check = P.check_every > 0 ? P.check_every : 1;                       // counters instead of modulos
  int n_check = 0;
  constexpr int REFRESH_CHECKS = 2;            // identical iterates and parity for 1, 2 and 4 on every test set
  int next_adapt = P.adapt_every > 0 ? P.adapt_start : 0x7fffffff;
  while (next_adapt < 1) next_adapt += P.adapt_every;                          // the test runs after ++it
  float res_p = 0.f, res_s = 0.f;

  // exact axg, bwl from x (exchange through LDS); all threads call
  auto refresh = [&]() {
    if (valid) { sm.xs[j][0][c] = xo[0]; sm.xs[j][1][c] = xo[1]; }
    wg_sync<NT>();
    if (valid) {
      RT xblk[2][6], gu[6];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int b = 0; b < 6; ++b) xblk[f][b] = sm.xs[j][f][b];
#pragma unroll
      for (int b = 0; b < 6; ++b) gu[b] = sm.Gu[c][b];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        RT a = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) a += gu[b] * xblk[f][b];
        const RT negmu = c < 4 ? -(RT)sm.muf[j][f] : (RT)0;
        axg[f] = a + negmu * xblk[f][2];
      }
      if (c < 3) {
        RT t0[3], t1[3];
        const RT r0[3] = {sm.rr[j][0][0], sm.rr[j][0][1], sm.rr[j][0][2]};
        const RT r1[3] = {sm.rr[j][1][0], sm.rr[j][1][1], sm.rr[j][1][2]};
        cross3(r0, &xblk[0][0], t0);
        cross3(r1, &xblk[1][0], t1);
        RT v3[3] = {t0[0] + t1[0] + xblk[0][3] + xblk[1][3], t0[1] + t1[1] + xblk[0][4] + xblk[1][4],
                    t0[2] + t1[2] + xblk[0][5] + xblk[1][5]};
        bwl = mk3[0] * v3[0] + mk3[1] * v3[1] + mk3[2] * v3[2];
      } else {
        RT v3[3] = {xblk[0][0] + xblk[1][0], xblk[0][1] + xblk[1][1], xblk[0][2] + xblk[1][2]};
        bwl = mk3[0] * v3[0] + mk3[1] * v3[1] + mk3[2] * v3[2];
      }
      sm.u.itv.bwT[c][j] = bwl;
    }
    wg_sync<NT>();
    if (valid) {
      // gbl = (Gt b + qt)[l] in f64: 3 H doubles of the lane's component group, 3 independent chains
      static_assert(H % 2 == 0, "the wrench is read in groups of 6");
      const RT* bsrc = &sm.u.itv.bwT[c < 3 ? 0 : 3][0];
      RT g0 = qt, g1 = 0, g2 = 0;
#pragma unroll
      for (int q = 0; q < 3 * H; q += 6) {
        RT v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = bsrc[q + k];
        g0 += (RT)Grow[q] * v[0];     g1 += (RT)Grow[q + 1] * v[1]; g2 += (RT)Grow[q + 2] * v[2];
        g0 += (RT)Grow[q + 3] * v[3]; g1 += (RT)Grow[q + 4] * v[4]; g2 += (RT)Grow[q + 5] * v[5];
      }
      gbl = g0 + (g1 + g2);
    }
  };

#pragma unroll 1
  for (it = 0; it < P.max_iter;) {
    if (need_factor) {                         // workgroup-uniform
      // park what the factorisation does not touch (see park() above)
      Parked64 pk[29];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        park(xo[f], pk[f]); park(zb[f], pk[2 + f]); park(zg[f], pk[4 + f]); park(yb[f], pk[6 + f]);
        park(yg[f], pk[8 + f]); park(axg[f], pk[10 + f]); park(irvb[f], pk[12 + f]); park(irvg[f], pk[14 + f]);
        park(lb[f], pk[16 + f]); park(ub[f], pk[18 + f]); park(cmu[f], pk[20 + f]);
      }
      park(gbl, pk[22]); park(qt, pk[23]);
#pragma unroll
      for (int k = 0; k < 3; ++k) park(mk3[k], pk[24 + k]);
      float pkd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) park(drf[k], pkd[k]);
#pragma unroll
      for (int q = 0; q < 3 * H; ++q) park(Grow[q], gpark[q]);
      factor();
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        unpark(xo[f], pk[f]); unpark(zb[f], pk[2 + f]); unpark(zg[f], pk[4 + f]); unpark(yb[f], pk[6 + f]);
        unpark(yg[f], pk[8 + f]); unpark(axg[f], pk[10 + f]); unpark(irvb[f], pk[12 + f]); unpark(irvg[f], pk[14 + f]);
        unpark(lb[f], pk[16 + f]); unpark(ub[f], pk[18 + f]); unpark(cmu[f], pk[20 + f]);
      }
      unpark(gbl, pk[22]); unpark(qt, pk[23]);
#pragma unroll
      for (int k = 0; k < 3; ++k) unpark(mk3[k], pk[24 + k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) unpark(drf[k], pkd[k]);
      ++nfac;
      need_factor = false;
    }
    if (dbg.prof) t_last = clock64();
    // --- P0: row residuals w = y + rho (A x - z); publish them and the net wrench
    RT wb[2];
    if (valid) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        wb[f] = yb[f] + rvb[f] * (xo[f] - zb[f]);
        sm.u.itv.wg[j][f][c] = yg[f] + rvg[f] * (axg[f] - zg[f]);
      }
      sm.u.itv.gb[l] = gbl;
    }
    wg_sync<NT>();
    BMPC_STAMP(0)
    // --- P2: KKT residual in control space r = W' gb + 2R x + A' w   (small at convergence)
    float lcol[2][6];
    if (valid) {
      RT gut[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) gut[q] = sm.GuT[c][q];
      // W_f' g for this lane's variable: force variable a gets (g_tau x r_f)_a + g_F[a], moment variable a gets
      // g_tau[a].  One straight line for all lanes: the cyclic neighbours of a are picked by address, the lever
      // arm components come from a per-lane table that is zero for moment variables.
      const int a3 = c < 3 ? c : c - 3;
      const int i1 = a3 == 2 ? 0 : a3 + 1, i2 = a3 == 0 ? 2 : a3 - 1;
      const RT g1 = sm.u.itv.gb[6 * j + i1], g2 = sm.u.itv.gb[6 * j + i2];
      const RT gsel = sm.u.itv.gb[6 * j + (c < 3 ? c + 3 : c - 3)];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        RT r = R2v[f] * xo[f] + wb[f];
        RT wq[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) wq[q] = sm.u.itv.wg[j][f][q];
#pragma unroll
        for (int q = 0; q < 6; ++q) r += gut[q] * wq[q];
        r += cmu[f] * ((wq[0] + wq[1]) + (wq[2] + wq[3]));
        const RT wt = g1 * sm.rx[j][f][c][0] - g2 * sm.rx[j][f][c][1] + gsel;
        sm.u.itv.r32[j][f][c] = (float)(r + wt);
      }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 6; ++i) lcol[f][i] = sm.LG[f][j][i][c][0];
    }
    wg_sync<NT>();
    BMPC_STAMP(2)
    // --- P3: beta = L' r
    float rj[2][6];
    if (valid) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          rj[f][i] = sm.u.itv.r32[j][f][i];
          s = fmaf(lcol[f][i], rj[f][i], s);
        }
      sm.u.itv.beta[l] = s;
    }
    wg_sync<NT>();
    BMPC_STAMP(3)
    // --- P4: gamma = V beta   (Vr holds -V)
    float gown = 0.f;
    f2 kg[2][6], lg[2][6];                             // rows c of {Kn, G Kn} and {L, G L} for P5
    if (valid) {
      f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NW; q += 4) {
        const float4 bq = *reinterpret_cast<const float4*>(&sm.u.itv.beta[q]);
        a0 = __builtin_elementwise_fma(Vr[q / 2], f2{bq.x, bq.y}, a0);
        a1 = __builtin_elementwise_fma(Vr[q / 2 + 1], f2{bq.z, bq.w}, a1);
      }
      gown = -((a0.x + a0.y) + (a1.x + a1.y));
      sm.u.itv.gam[l] = gown;
      sm.u.itv.gamT[c < 3 ? 0 : 1][(c < 3 ? c : c - 3) * H + j] = gown;
    }
    wg_sync<NT>();
    BMPC_STAMP(4)
    // --- P5: x~ = x - d, z~ = A x~ (carried), relaxation, projection, dual update
    float rp = 0.f, rs = 0.f, nz = 0.f, nx = 0.f;   // residual statistics: only where the stopping test runs
    const bool check_now = (it + 1 == next_check) || (it + 1 == P.max_iter);     // workgroup-uniform
    if (valid) {
      float gm[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) gm[i] = sm.u.itv.gam[6 * j + i];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          kg[f][i] = *reinterpret_cast<const f2*>(&sm.KG[f][j][c][i][0]);
          lg[f][i] = *reinterpret_cast<const f2*>(&sm.LG[f][j][c][i][0]);
        }
      // t = N' r = r_0 - T' r_1 ;  null-space part of d: foot 0 gets Ka^-1 t, foot 1 gets -(T Ka^-1) t
      float tn[6];
      {
        const float d0 = drf[0], d1 = drf[1], d2 = drf[2];
        tn[0] = rj[0][0] - rj[1][0] + (d1 * rj[1][5] - d2 * rj[1][4]);
        tn[1] = rj[0][1] - rj[1][1] + (d2 * rj[1][3] - d0 * rj[1][5]);
        tn[2] = rj[0][2] - rj[1][2] + (d0 * rj[1][4] - d1 * rj[1][3]);
        tn[3] = rj[0][3] - rj[1][3];
        tn[4] = rj[0][4] - rj[1][4];
        tn[5] = rj[0][5] - rj[1][5];
      }
      RT st_pb[2], st_pg[2], st_x[2], st_g[2], st_dx[2];     // residual statistics inputs (used at stopping tests)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        f2 dd = {0.f, 0.f};                     // {d_f[c], (G_f d_f)[c]}
#pragma unroll
        for (int i = 0; i < 6; ++i) dd = __builtin_elementwise_fma(kg[f][i], f2{tn[i], tn[i]}, dd);
        if (f == 1) dd = -dd;
#pragma unroll
        for (int i = 0; i < 6; ++i) dd = __builtin_elementwise_fma(lg[f][i], f2{gm[i], gm[i]}, dd);
        const float s = dd.x, sg = dd.y;
        const RT xto = xo[f] - (RT)s;
        const RT ztg = axg[f] - (RT)sg;
        const RT ztb = xto;
        // box row
        {
          const RT zr = alpha * ztb + (1 - alpha) * zb[f];
          const RT cand = zr + yb[f] * irvb[f];
          const RT zn = min_rt(max_rt(cand, lb[f]), ub[f]);
          yb[f] += rvb[f] * (zr - zn);
          zb[f] = zn;
          st_pb[f] = ztb - zn;
        }
        // general row: l = -inf, u = 0
        {
          const RT zr = alpha * ztg + (1 - alpha) * zg[f];
          const RT cand = zr + yg[f] * irvg[f];
          const RT zn = min_rt(cand, (RT)0);
          yg[f] += rvg[f] * (zr - zn);
          zg[f] = zn;
          st_pg[f] = ztg - zn;
        }
        st_x[f] = xto; st_g[f] = ztg; st_dx[f] = xto - xo[f];
        xo[f] = alpha * xto + (1 - alpha) * xo[f];
        axg[f] = alpha * ztg + (1 - alpha) * axg[f];
      }
      // gb follows b <- b - alpha gamma: gb -= alpha Gt gamma, the increment in f32 (it vanishes with the step)
      {
        const float* gsrc = &sm.u.itv.gamT[c < 3 ? 0 : 1][0];
        f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3 * H; q += 4) {
          if (q + 4 <= 3 * H) {
            const float4 g4 = *reinterpret_cast<const float4*>(&gsrc[q]);
            e0 = __builtin_elementwise_fma(f2{Grow[q], Grow[q + 1]}, f2{g4.x, g4.y}, e0);
            e1 = __builtin_elementwise_fma(f2{Grow[q + 2], Grow[q + 3]}, f2{g4.z, g4.w}, e1);
          } else {
            const float2 g2 = *reinterpret_cast<const float2*>(&gsrc[q]);
            e0 = __builtin_elementwise_fma(f2{Grow[q], Grow[q + 1]}, f2{g2.x, g2.y}, e0);
          }
        }
        gbl -= alpha * (RT)((e0.x + e0.y) + (e1.x + e1.y));
      }
      if (check_now) {
        // a real (uniform) branch: predicated, this costs ~25 instructions in every iteration
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          rp = fmaxf(rp, fmaxf(fabsf((float)st_pb[f]), fabsf((float)st_pg[f])));
          nz = fmaxf(nz, fmaxf(fabsf((float)st_x[f]), fabsf((float)st_g[f])));
          rs = fmaxf(rs, fabsf((float)st_dx[f]));
          // a NaN iterate must reach the test (fmaxf drops NaNs): it is reported as an infinite norm
          nx = (st_x[f] == st_x[f]) ? fmaxf(nx, fabsf((float)st_x[f])) : __builtin_inff();
        }
      }
    }
    ++it;
    BMPC_STAMP(5)
    // --- stopping test (workgroup-uniform); the carried products are rebuilt from x first
    if (check_now) {
      next_check += P.check_every;
      float v4[4] = {rp, rs, nz, nx};
      block_max4<NT>(v4, sm.red);
      res_p = v4[0];
      res_s = v4[1];
      const bool bad = !(v4[0] == v4[0]) || !(v4[1] == v4[1]) || !(v4[3] < 3.0e38f);
      const bool done = v4[0] <= P.eps_pri * fmaxf(1.f, v4[2]) && v4[1] <= P.eps_dua * fmaxf(1.f, v4[3]);
      // the exact rebuild of the carried products: before leaving (the outputs use it) and at every
      // REFRESH_CHECKS-th test otherwise
      ++n_check;
      if (bad || done || it == P.max_iter || n_check % REFRESH_CHECKS == 0) refresh();
      if (bad) { status = 2; break; }
      if (done) { status = 0; break; }
    }
    // --- penalty re-classification by the current active set
    if (it == next_adapt) {
      next_adapt += P.adapt_every;
      if (nfac <= P.max_refactor) {
      int changed = 0;
      float nb[2], ng[2];
      // Damping: an instance that is still re-classifying after many rounds is cycling between active sets
      // (about one in a million at kappa = 20); smaller moves break the cycle (sqrt(kappa) after 10
      // factorisations, its square root after 16), where stopping the adaptation would leave hundreds of
      // plain-ADMM iterations.
      const float kap = nfac <= 10 ? P.kappa : (nfac <= 16 ? sqrtf(P.kappa) : sqrtf(sqrtf(P.kappa)));
      if (valid) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const bool actb = (zb[f] <= lb[f] || zb[f] >= ub[f]) && yb[f] != (RT)0;
          const bool actg = (zg[f] >= (RT)0) && yg[f] != (RT)0;
          // active rows move up by kappa towards their class ceiling, inactive ones down towards rho_lo
          const float hib = c < 3 ? P.rho_hi_f : P.rho_hi_m, hig = c < 4 ? P.rho_hi_f : P.rho_hi_m;
          const float ob = (float)rvb[f], og = (float)rvg[f];
          nb[f] = eqb[f] ? P.rho_eq : (actb ? fminf(ob * kap, hib) : fmaxf(ob / kap, P.rho_lo));
          ng[f] = actg ? fminf(og * kap, hig) : fmaxf(og / kap, P.rho_lo);
          changed |= (nb[f] != ob) | (ng[f] != og);
        }
      }
      changed = __syncthreads_or(changed);
      if (changed) {
        if (valid) {
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            rvb[f] = (RT)nb[f]; rvg[f] = (RT)ng[f];
            irvb[f] = (RT)1 / rvb[f]; irvg[f] = (RT)1 / rvg[f];
          }
        }
        need_factor = true;
      }
      }
    }
    BMPC_STAMP(6)
  }
  if (valid) { sm.xs[j][0][c] = xo[0]; sm.xs[j][1][c] = xo[1]; }   // for the state roll-out below

  // ------------------------------------------------------------------ F. outputs (REF:300-304)
  if (valid) {
    float* uo = controls + ((size_t)inst * H + j) * 12;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int pos = c < 3 ? 3 * f + c : 6 + 3 * f + (c - 3);       // [f1 f2 m1 m2]
      uo[pos] = (float)xo[f];
    }
  }
  if (states) {
    // wrench of the final x (exact: rebuilt at the last stopping test), then X_i = s_i + Gam_t b
    wg_sync<NT>();
    if (valid) sm.u.itv.bwT[c][j] = bwl;
    wg_sync<NT>();
    if (valid) {
      float* so = states + ((size_t)inst * H + j) * 13;
      const int i = j;
      if (c < 3) {
        const int a = c;
        // euler: s + sum_{j2 < i} Me[i][j2] tau_j2 ; omega: w_fb + dt sum_{j2 <= i} Iw_j2 tau_j2
        RT e = sm.s0[i][a], w = sm.s0[i][6 + a];
#pragma unroll 1
        for (int j2 = 0; j2 <= i; ++j2) {
          const RT t3[3] = {sm.u.itv.bwT[0][j2], sm.u.itv.bwT[1][j2], sm.u.itv.bwT[2][j2]};
          if (j2 < i) {
            const float* m1 = sm.Me[pair_index(i, j2)];
            e += (RT)m1[3 * a] * t3[0] + (RT)m1[3 * a + 1] * t3[1] + (RT)m1[3 * a + 2] * t3[2];
          }
          w += dt * (sm.Iw[j2][3 * a] * t3[0] + sm.Iw[j2][3 * a + 1] * t3[1] + sm.Iw[j2][3 * a + 2] * t3[2]);
        }
        so[a] = (float)e;
        so[6 + a] = (float)w;
      } else {
        const int a = c - 3;
        RT p = sm.s0[i][3 + a], v = sm.s0[i][9 + a];
        const RT kp = dt * dt / (RT)P.m, kvv = dt / (RT)P.m;
#pragma unroll 1
        for (int j2 = 0; j2 <= i; ++j2) {
          const RT fa = sm.u.itv.bwT[3 + a][j2];
          p += kp * (RT)(i - j2) * fa;
          v += kvv * fa;
        }
        so[3 + a] = (float)p;
        so[9 + a] = (float)v;
      }
      if (c == 0) so[12] = 1.0f;
    }
  }
  if (dbg.prof && l == 0) {
    long long* pr = dbg.prof + (size_t)inst * 16;
    pr[0] = t_setup; pr[1] = t_blocks; pr[2] = t_sweep; pr[3] = clock64() - t_start; pr[4] = it; pr[5] = nfac;
#pragma unroll
    for (int k = 0; k < 7; ++k) pr[8 + k] = t_ph[k];
  }
  if (l == 0) {
    if (iters_out) iters_out[inst] = it;
    if (status_out) status_out[inst] = status;
    if (nfactor_out) nfactor_out[inst] = nfac;
    if (resid_out) { resid_out[2 * inst] = res_p; resid_out[2 * inst + 1] = res_s; }
  }
}

}  // namespace bmpc
