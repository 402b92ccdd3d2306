// bmpc_stage.hip -- stage-structured solve kernels (gfx950 / CDNA4): the long-horizon path of the batched HECTOR MPC.
//
// SURVEY 8(f) row 4.  The dense kernels (bmpc_kernels.hip) invert K' = Gt + F (6h x 6h) explicitly: O(h^3) work, a row
// half of the inverse per lane (255 VGPRs at h = 20) and an O(h^2) set-up table -- they end at h = 20.  This file keeps
// the SPARSE stage structure the reference's equality block has (REF:203-216: X_i = A_i X_{i-1} + B_i U_i) instead:
//
//   * (Gt + F) gamma = beta is the optimality system of an LQ problem in the 12 SRBM states,
//        min sum_i 1/2 a_i' Ft_i a_i - bt_i' a_i + 1/2 xi_i' 2Q xi_i,   xi_i = A_i xi_{i-1} + [0; a_i],   a = E gamma,
//     A_i = [[I, C_i], [0, I]], C_i = dt blkdiag(Rinv_i, I) (REF:165-171, 183), E_i = dt blkdiag(Iw_i^-1, I/m) (REF:174-184).
//     A backward Riccati recursion factorises it in h steps of 6x6 / 6x12 blocks (O(h) work and state), and a solve
//     is one backward and one forward pass over the steps (oracle/riccati_model.py is the executable specification).
//   * Gt itself is never formed: the tracking error err = s - x_ref + Gam_t W u is carried in the 12 h state
//     coordinates (f64, one per lane and step), and the wrench-space gradient is its adjoint (two suffix sums over the
//     steps).  Set-up is O(h) per lane.
//   * The outer method is the one of the dense kernels, unchanged (ADMM in residual form, active-set adaptive
//     penalties, f64 iterates and residual, f32 preconditioner: DESIGN.md sections 3, 4), so the iteration counts, the
//     convergence record of the soaks and the parity numbers carry over; only the application of K^-1 differs.
//
// Thread map: ONE WAVE per instance up to h = 24, TWO from h = 26 (NW).  Lane l of wave w = 12 q + 2 c + f: group q (5
// groups; lanes 60..63 clone 56..59), component c, foot f; the lane owns control variable c of foot f -- and state
// coordinate n = 2 c + f -- at the NP consecutive steps j = (q NW + w) NP + s (s = 0 .. NP-1).  The 12 lanes of a step
// always sit in one wave, so what they exchange stays wave-local; with two waves only the scans over the steps, the
// two sequential passes and the Riccati recursion (run by wave 0) and the reductions cross a workgroup barrier.
// (Two waves halve the steps a lane owns: at h = 40 one wave would need ~1000 registers and spill half of them.)
// Steps past the horizon are PHANTOMS: they have their own LDS slots (the step arrays hold 5 NP NW steps), read the
// inputs of step h - 1, do the same arithmetic as everybody else and are masked out of everything that crosses steps
// (scans, the sequential passes, reductions, outputs) -- nothing is predicated on them.  h is a launch parameter: the
// kernel is compiled per (NP, NW) = (steps a lane owns, waves): (2,1) h <= 10 ... (5,1) h <= 24, (3,2) h <= 30, (4,2) h <= 40.
// Within a wave no s_barrier: lanes of one wave exchange through LDS in program order (BMPC_WAVE_SYNC only fences the
// compiler) and through DPP.  The two sequential passes of a solve run on the 12 lanes of a DPP row, one state
// coordinate per lane, as 12x12 mat-vecs whose operands arrive by row broadcast (no LDS round trip in a chain) -- and
// BLOCK-wise: the steps are grouped into NB <= 4 NW blocks of S steps, one block per DPP row; every row forms the affine
// map of its block at the same time, a short chain over the blocks (their products A_b are part of the factorisation)
// gives each block its input, and the rows replay their steps from it: S + NB + S dependent mat-vecs per pass instead
// of h.  An instance needs no register-resident matrix, so many instances share a CU: the path is bound by the latency
// of its dependent chains, and throughput comes from the number of instances in flight.

#ifndef BMPC_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <type_traits>

namespace bmpc {

#ifndef BMPC_EMU
// value of lane N of the own DPP row (16 lanes): row_newbcast:N
template <int N>
__device__ __forceinline__ float row_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xf, 0xf, false));
}
#endif

// acc + sum_m cf[m] * (value of u in lane m of the own DPP row), m = 0 .. 11: the step of the two sequential passes of a
// stage solve.  A single wave issues one vector instruction per 4 cycles, so the latency of the chain IS its instruction
// count: the broadcasts ride on the FMAs (v_fmac_f32_dpp: one instruction per term; the compiler does not fold
// row_newbcast moves into the FMA by itself), three accumulators keep the FMAs independent of their neighbours.
// Inline asm: the hazard recogniser does not look inside, so the two wait states a DPP read needs after the VALU write
// of its source (u is produced right before) are written out.
__device__ __forceinline__ float row_matvec12(float acc, float u, const float (&cf)[12]) {
#ifdef BMPC_EMU
  float a0 = acc, a1 = 0.f, a2 = 0.f;
  a0 = fmaf(cf[0], row_bcast<0>(u), a0);  a1 = fmaf(cf[1], row_bcast<1>(u), a1);   a2 = fmaf(cf[2], row_bcast<2>(u), a2);
  a0 = fmaf(cf[3], row_bcast<3>(u), a0);  a1 = fmaf(cf[4], row_bcast<4>(u), a1);   a2 = fmaf(cf[5], row_bcast<5>(u), a2);
  a0 = fmaf(cf[6], row_bcast<6>(u), a0);  a1 = fmaf(cf[7], row_bcast<7>(u), a1);   a2 = fmaf(cf[8], row_bcast<8>(u), a2);
  a0 = fmaf(cf[9], row_bcast<9>(u), a0);  a1 = fmaf(cf[10], row_bcast<10>(u), a1); a2 = fmaf(cf[11], row_bcast<11>(u), a2);
  return a0 + (a1 + a2);
#else
  float a0 = acc, a1 = 0.f, a2 = 0.f;
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %3, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %3, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %3, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %3, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %3, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %3, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %3, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %3, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %3, %14 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %3, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf"
      : "+&v"(a0), "+&v"(a1), "+&v"(a2)      // early clobber: acc may hold the same value as u and must not share its register
      : "v"(u), "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]), "v"(cf[4]), "v"(cf[5]), "v"(cf[6]), "v"(cf[7]), "v"(cf[8]),
        "v"(cf[9]), "v"(cf[10]), "v"(cf[11]));
  return a0 + (a1 + a2);
#endif
}

// A per-step LDS array, stored with the step index PERMUTED: a lane group owns G = NP NW consecutive steps, so in one
// LDS instruction the five groups of a wave touch steps G apart -- and with row sizes of 6 .. 72 words G rows are a
// multiple of 32 banks for most arrays: 3- to 5-way conflicts (41 % of the LDS-array cycles at h = 40).  Step j lives
// in slot (j mod G) 5 + j / G: the groups' rows become neighbours, the rows of one lane 5 slots apart.  Call sites index
// by the step; only the storage order changes (no padding: the long-horizon variants have no LDS to spare).
// a step together with its slot: the steps a lane owns sit at slot0 + 5 s, constant offsets from one base address
struct Step {
  int j, slot;
  __device__ __forceinline__ operator int() const { return j; }
};

template <class Row, int HS, int G>
struct StepArr {
  Row v[HS];
  // (G = 0: plain order -- the five-steps-per-lane variant has no registers for a second family of base addresses)
  static __device__ __forceinline__ int slot_of(int j) { return G ? (int)(((unsigned)j % (unsigned)(G ? G : 1)) * 5u + (unsigned)j / (unsigned)(G ? G : 1)) : j; }
  __device__ __forceinline__ Row& operator[](Step st) { return v[G ? st.slot : st.j]; }
  __device__ __forceinline__ const Row& operator[](Step st) const { return v[G ? st.slot : st.j]; }
  __device__ __forceinline__ Row& operator[](int j) { return v[slot_of(j)]; }
  __device__ __forceinline__ const Row& operator[](int j) const { return v[slot_of(j)]; }
};

// the arrays the sequential parts walk step by step (Riccati recursion, the chains of a solve) keep the plain order:
// there one instruction touches one step per DPP row or one step at all, and the permutation would only cost its
// index arithmetic (measured: +5 % on the recursion, +8 % on the chains)
template <class Row, int HS>
struct PlainArr {
  Row v[HS];
  __device__ __forceinline__ Row& operator[](int j) { return v[j]; }
  __device__ __forceinline__ const Row& operator[](int j) const { return v[j]; }
};

template <int NP, int NW>
struct alignas(16) StageSmem {
  static constexpr int HS = 5 * NP * NW;       // step capacity
  static constexpr int G = NP <= 4 ? NP * NW : 0;   // consecutive steps of a lane group (0: per-step arrays in plain order)
  struct FootBlock { StepArr<float[6][6], HS, G> d; };
  // f64 6x6 scratch of the block algebra of ONE pass (5 steps), indexed by lane group
  struct Fac {
    double M0[5 * NW][6][6];   // D0 -> Ka^-1 D0 W_0^-1
    double M1[5 * NW][6][6];   // D1 -> Ka^-1
    double M2[5 * NW][6][6];   // B = T' D1 T -> L~_0
    double Ka[5 * NW][6][6];
  };
  // exchange vectors of an iteration; never live together with the factor scratch
  struct Itv {
    StepArr<RT[2][6], HS, G> wg;           // y + rho (A x - z) on the general rows; x itself for the exact rebuild and the outputs
    StepArr<RT[12], HS, G> lam;            // adjoint of the tracking error (acceleration space), and the scans' exchange
    alignas(16) StepArr<float[2][6], HS, G> r32;   // KKT residual, control space; after P3 the step d of x
    alignas(16) PlainArr<float[6], HS> bt;       // right-hand side of the stage solve (E^-T beta); after the backward pass w = Sinv g
    PlainArr<float[8], HS> gs;           // g = p2 - bt of the backward pass (slot 6: dump for the lanes that hold no g)
    PlainArr<float[8], HS> av;           // accelerations a = E gamma of the solve (slot 6: dump)
    alignas(16) PlainArr<float[12], HS> xi;      // state response Gam_t gamma
    alignas(16) PlainArr<float[12], HS> rbw;     // backward pass: r_i = M_i' [0; bt_i]
  };
  union alignas(16) { Fac fac; Itv itv; } u;
  RT tot[2][5 * NW][12];       // block totals of the scans over the steps (double buffered)
  float red[2][12][NW];        // reductions across the waves (nine maxima, three sums)
  // block-diagonal part of K^-1 (see bmpc_kernels.hip): L~ = L E^-1 (acceleration space), Kn = {Ka^-1, T Ka^-1}.
  // (no G images as in the dense kernels: an instance's LDS decides how many instances share a CU, and the step d
  //  is exchanged once per iteration instead)
  alignas(16) FootBlock LG[2];
  alignas(16) FootBlock KG[2];
  // stage solve.  Step i maps the state by F_i = I + M_i, M_i = [[0, C_i], [-K_i]] (12 x 12): the Riccati gain is kept as
  // Kn_i = -K_i (6 x 12; until the recursion reaches step i its first 36 floats hold the stage cost Ft_i of the block
  // algebra), C_i = dt blkdiag(Rinv_i, I) as its 3x3 block Cr_i.  The steps are grouped into NB <= 4 NW blocks of S
  // consecutive steps, one per DPP row of the instance, with the block products A_b = F_{last} .. F_{first}.
  static constexpr int NBM = 4 * NW;
  alignas(16) PlainArr<float[6][12], HS> Kn;
  alignas(16) PlainArr<float[12], HS> Cr;
  alignas(16) float Ab[NBM][12][12];
  alignas(16) float blkv[NBM][12];    // per block: the affine part of the block map
  alignas(16) float blkp[NBM][12];    // per block: the state entering it
  alignas(16) float zblk[72];         // zeros (coefficients of steps past the horizon)
  alignas(16) float dtrow[3][12];     // rows 3..5 of M_i (the same for every step): dt in column 9 + k
  alignas(16) PlainArr<float[6][6], HS> Sinv;
  alignas(16) float Pm[12][12];   // Riccati recursion: cost-to-go
  alignas(16) float Zm[12][12];   //                    Schur complement (blocks 12, 22; Z11 replaces Pi11 in Pm)
  alignas(16) float Tm[6][6];     //                    S^-1 Ft
  alignas(16) float Sm[6][6];     //                    S = Ft_i + Pi22 of the step the recursion reaches next (lower triangle)
  float q2[12];                   // 2 Q
  // step data
  StepArr<RT[9], HS, G> Iwi;               // world inverse inertia
  StepArr<RT[9], HS, G> Rv;                // R_inv (REF:160-164)
  StepArr<RT[2][3], HS, G> rr;             // r_f = foot_ref - com_ref
  StepArr<float[2], HS, G> muf;
  StepArr<float[2][6], HS, G> rvg;
  RT Gu[6][6];
  RT GuT[6][6];
  float eyz[6];
};

// x with S x = b for a 6x6 SPD S given by its lower triangle (LDL', f32; every lane for itself)
__device__ __forceinline__ void ldl6_solve(const float (&s)[6][6], const float (&b)[6], float (&x)[6]) {
  float a[6][6], dinv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k <= i; ++k) a[i][k] = s[i][k];
#pragma unroll
  for (int jj = 0; jj < 6; ++jj) {
    float d = a[jj][jj];
#pragma unroll
    for (int k = 0; k < jj; ++k) d = fmaf(-a[jj][k] * a[jj][k], a[k][k], d);
    float r = rcp_approx(d);
    r = fmaf(fmaf(-d, r, 1.f), r, r);
    dinv[jj] = r;
    a[jj][jj] = d;
#pragma unroll
    for (int i = jj + 1; i < 6; ++i) {
      float v = a[i][jj];
#pragma unroll
      for (int k = 0; k < jj; ++k) v = fmaf(-a[i][k] * a[jj][k], a[k][k], v);
      a[i][jj] = v * r;
    }
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v = fmaf(-a[i][k], y[k], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float v = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v = fmaf(-a[k][i], x[k], v);
    x[i] = v;
  }
}

template <int NP, int NW, bool PROF>
__device__ __forceinline__ void
stage_body(const DevParams& P, const int B,
           const float* __restrict__ x_fb, const float* __restrict__ foot,
           const uint8_t* __restrict__ contact, const int32_t* __restrict__ phase,
           const float* __restrict__ x_cmd, const float* __restrict__ mu_in,
           float* __restrict__ controls, float* __restrict__ states,
           int32_t* __restrict__ iters_out, float* __restrict__ resid_out,
           int32_t* __restrict__ status_out, int32_t* __restrict__ nfactor_out,
           const DebugOut& dbg, const WarmArgs& warm) {
  constexpr int HS = 5 * NP * NW;
  constexpr int NT = 64 * NW;
  __shared__ StageSmem<NP, NW> sm;

  if ((int)blockIdx.x >= B) return;
  const int inst = warm.order ? warm.order[blockIdx.x] : (int)blockIdx.x;
  if (warm.rescue_status && warm.rescue_status[inst] == 0) return;     // (workgroup uniform, before any barrier)
  const int H = P.h;                           // NP = ceil(H / 5) (checked on the host)
#ifdef BMPC_EMU
  if (threadIdx.x == 0) std::memset(&sm, g_poison, sizeof(sm));
  sync_workgroup();
#endif
  // For many passes per lane the scheduler would interleave all of them (every pass body is independent of the others)
  // and run out of registers; a fence after every second pass keeps two in flight.
#define BMPC_PASS_FENCE(s) do { if constexpr (NP > 4) { if (((s) & 1) == 1) BMPC_SCHED_BARRIER(); } } while (0)
  long long t_start = 0, t_setup = 0, t_blocks = 0, t_ric = 0, t_mark = 0;
  long long t_ph[7] = {0, 0, 0, 0, 0, 0, 0}, t_last = 0;
#define BMPC_SSTAMP(k) if constexpr (PROF) { const long long t_ = clock64(); t_ph[k] += t_ - t_last; t_last = t_; }
  if constexpr (PROF) t_start = clock64();
  const int lt = threadIdx.x;                  // thread of the workgroup
  const int wv = NW > 1 ? lt >> 6 : 0;         // wave
  const int l = lt & 63;                       // lane of the wave
  const int lc = l < 60 ? l : l - 4;           // lanes 60..63 clone lanes 56..59 (same indices, same data, same stores)
  const bool lane_real = l < 60;
  // exchange between ALL lanes of the instance: the wave's own LDS ordering for one wave, a workgroup barrier for two
  auto sync_all = [&]() __attribute__((always_inline)) {
    if constexpr (NW > 1) { sync_workgroup(); } else { BMPC_WAVE_SYNC(); }
  };
  const int q = lc / 12;                       // lane group: steps q NP .. q NP + NP - 1
  const int n = lc % 12;                       // state coordinate [e(3), p(3), w(3), v(3)] of the scans
  const int c = n >> 1;                        // component of the control variable v = [f(3), m(3)]
  const int f = n & 1;                         // foot
  const int hf = f;
  const int rn_lane = (l & 15) < 12 ? (l & 15) : 11;
  const int rn = rn_lane;   // state coordinate of the lane in the chains (DPP row; lanes 12..15 clone 11)
  const RT dt = (RT)P.dt;
  const int qb = q * NW + wv;                   // block of NP consecutive steps this lane owns
  // steps of this lane (= LDS slots); past the horizon: phantoms with the inputs of the last step
  int js[NP], jg[NP];
  bool sreal[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    js[s] = (q * NW + wv) * NP + s;
    sreal[s] = js[s] < H;
    jg[s] = js[s] < H ? js[s] : H - 1;          // index into the per-step inputs in HBM
  }
  // step s of this lane with its LDS slot: StepArr's permutation of js[s] spelled out, one base + 5 s
  const int slot0 = wv * NP * 5 + q;
#define BMPC_STEP(s) (Step{js[s], slot0 + 5 * (s)})

  // (every lambda below is forced inline: one that is called from two places -- the scans, the exact rebuild -- is
  //  otherwise a real function inside a large module, and whatever it captures by reference then lives in scratch)
  // scans over the steps of one value per (lane, step), f64: the lane's NP steps are consecutive, so a scan is a local
  // pass in registers plus one exchange of the 5 NW block totals.  Steps past the horizon contribute nothing.
  int n_scan = 0, n_red = 0;
  auto group_add = [&](RT run, bool suffix) __attribute__((always_inline)) -> RT {
    RT (*tb)[12] = sm.tot[n_scan & 1];
    ++n_scan;
    tb[qb][n] = run;
    sync_all();
    RT add = 0;
#pragma unroll
    for (int q2 = 0; q2 < 5 * NW; ++q2) {
      const RT t = tb[q2][n];
      add += (suffix ? q2 > qb : q2 < qb) ? t : (RT)0;
    }
    return add;
  };
  auto suffix_incl = [&](RT (&v)[NP]) {        // v[s] <- sum over steps j' >= j(s)
    RT run = 0;
#pragma unroll
    for (int s = NP - 1; s >= 0; --s) { run += sreal[s] ? v[s] : (RT)0; v[s] = run; }
    const RT add = group_add(run, true);
#pragma unroll
    for (int s = 0; s < NP; ++s) v[s] += add;
  };
  auto suffix_excl = [&](RT (&v)[NP]) {        // v[s] <- sum over steps j' > j(s)
    RT run = 0;
#pragma unroll
    for (int s = NP - 1; s >= 0; --s) { const RT o = sreal[s] ? v[s] : (RT)0; v[s] = run; run += o; }
    const RT add = group_add(run, true);
#pragma unroll
    for (int s = 0; s < NP; ++s) v[s] += add;
  };
  auto prefix_incl = [&](RT (&v)[NP]) {        // v[s] <- sum over steps j' <= j(s)
    RT run = 0;
#pragma unroll
    for (int s = 0; s < NP; ++s) { run += sreal[s] ? v[s] : (RT)0; v[s] = run; }
    const RT add = group_add(run, false);
#pragma unroll
    for (int s = 0; s < NP; ++s) v[s] += add;
  };

  // ------------------------------------------------------------------ A. references, step data, free response
  RT xfb[12], xc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    xfb[i] = (RT)x_fb[(size_t)inst * 12 + i];
    xc[i] = x_cmd ? (RT)x_cmd[(size_t)inst * 12 + i] : (RT)P.x_cmd[i];
  }
  const int kph = phase[inst];
  const RT xfb_n = (RT)x_fb[(size_t)inst * 12 + n];
  const RT xc_n = x_cmd ? (RT)x_cmd[(size_t)inst * 12 + n] : (RT)P.x_cmd[n];
  const RT xc_n6 = n < 6 ? (x_cmd ? (RT)x_cmd[(size_t)inst * 12 + n + 6] : (RT)P.x_cmd[n + 6]) : (RT)0;
  const bool lead = lane_real && n == 0;       // one lane per (group, step)
  RT e0[NP];                                   // free response - reference, coordinate n of the lane's steps
  RT err[NP];                                  // tracking error s - x_ref + Gam_t W x (carried)
  {
    const int c0 = contact[(size_t)inst * H * 2 + 0], c1 = contact[(size_t)inst * H * 2 + 1];
    const bool single = (c0 + c1) == 1;        // REF:102
    const int kk = kph % P.half;               // REF:101
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const int j = jg[s];
      const Step jst = BMPC_STEP(s);
      RT xr[12];                               // x_ref[:, j]  (REF:61-70)
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        if (j == 0) xr[i] = xfb[i];
        else if (i < 6) xr[i] = (xc[i + 6] != (RT)0) ? xfb[i] + xc[i + 6] * ((RT)j * dt) : xc[i];
        else xr[i] = xc[i];
      }
      RT fr[6];                                // foot_ref[:, j]  (REF:72-109)
#pragma unroll
      for (int i = 0; i < 6; ++i) fr[i] = (RT)foot[(size_t)inst * 6 + i];
      if (single && j >= P.half - kk) {
        const bool second = j >= 2 * P.half - kk;
        const RT hor = second ? (RT)0.5 * (RT)H * dt : (RT)0.5 * (RT)H / (RT)2 * dt;   // REF:74, 78
        const RT fx = xfb[3] + xfb[9] * hor + (RT)P.kv * (xfb[3] - xc[3]);
        const RT fy = (second ? xfb[10] : xfb[4]) + xfb[10] * hor + (RT)P.kv * (xfb[4] - xc[4]);  // REF:87 quirk
        fr[0] = fx; fr[1] = fy; fr[2] = 0; fr[3] = fx; fr[4] = fy; fr[5] = 0;
      }
      if (lead && sreal[s]) {                  // debug views of the references (tests)
        if (dbg.x_ref) {
#pragma unroll
          for (int i = 0; i < 12; ++i) dbg.x_ref[((size_t)inst * H + j) * 12 + i] = (double)xr[i];
        }
        if (dbg.foot_ref) {
#pragma unroll
          for (int i = 0; i < 6; ++i) dbg.foot_ref[((size_t)inst * H + j) * 6 + i] = (double)fr[i];
        }
      }
      RT sy, cy, sp, cp, sr, cr;               // REF:151-153: yaw = x[0], pitch = x[1], roll = x[2]
      sincos(xr[0], &sy, &cy);
      sincos(xr[1], &sp, &cp);
      sincos(xr[2], &sr, &cr);
      // Rot = Rx(roll) Ry(pitch) Rz(yaw)   (scipy 'zyx' extrinsic, REF:154-156)
      const RT Rot[9] = {cp * cy, -cp * sy, sp,
                         cr * sy + sr * sp * cy, cr * cy - sr * sp * sy, -sr * cp,
                         sr * sy - cr * sp * cy, sr * cy + cr * sp * sy, cr * cp};
      RT T[9], Iwi[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          T[3 * a + b] = (RT)P.Iinv[3 * a] * Rot[b] + (RT)P.Iinv[3 * a + 1] * Rot[3 + b] + (RT)P.Iinv[3 * a + 2] * Rot[6 + b];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          Iwi[3 * a + b] = Rot[a] * T[b] + Rot[3 + a] * T[3 + b] + Rot[6 + a] * T[6 + b];   // Rot' Iinv Rot = (Rot' I Rot)^-1
      const RT tp = sp / cp;
      const RT Rv[9] = {cy / cp, sy / cp, 0, -sy, cy, 0, cy * tp, sy * tp, 1};             // REF:160-164 inverted
      if (lead) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { sm.Iwi[jst][k] = Iwi[k]; sm.Rv[jst][k] = Rv[k]; }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int a = 0; a < 3; ++a) sm.rr[jst][ft][a] = fr[3 * ft + a] - xr[3 + a];       // REF:174-175
      }
      // coordinate n of x_ref[:, j], and the term of the free response that is a sum over the steps:
      // euler_j = euler_fb + dt sum_{i <= j} Rinv_i omega_fb  (the other coordinates are closed forms)
      const RT xrn = (j == 0) ? xfb_n : ((n < 6 && xc_n6 != (RT)0) ? xfb_n + xc_n6 * ((RT)j * dt) : xc_n);
      RT rw = 0;
      if (n < 3) rw = dt * ((n == 0 ? Rv[0] : (n == 1 ? Rv[3] : Rv[6])) * xfb[6] + (n == 0 ? Rv[1] : (n == 1 ? Rv[4] : Rv[7])) * xfb[7] +
                            (n == 0 ? Rv[2] : (n == 1 ? Rv[5] : Rv[8])) * xfb[8]);
      err[s] = rw;
      e0[s] = xrn;
    }
  }
  BMPC_WAVE_SYNC();
  prefix_incl(err);
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    const Step j = BMPC_STEP(s);
    const RT j1 = (RT)(jg[s] + 1);
    RT e = xfb_n + err[s];                     // (err is zero for n >= 3)
    if (n >= 3 && n < 6) e += dt * j1 * (n == 3 ? xfb[9] : (n == 4 ? xfb[10] : xfb[11]));
    if (n == 5) e -= (RT)P.g * dt * dt * (RT)jg[s] * j1 / 2;
    if (n == 11) e -= (RT)P.g * dt * j1;
    e0[s] = e - e0[s];
    err[s] = e0[s];                            // x = 0
  }
  if (dbg.assemble_only) return;
  // the rotational block of C_i = dt blkdiag(Rinv_i, I) (constant over the factorisations), zeros, and 2 Q
  for (int e = lt; e < H * 12; e += NT) {
    const int i = e / 12, k = e % 12;
    sm.Cr[i][k] = k < 9 ? (float)(dt * sm.Rv[i][k]) : 0.f;
  }
  for (int e = lt; e < 72; e += NT) sm.zblk[e] = 0.f;
  for (int e = lt; e < 36; e += NT) sm.dtrow[e / 12][e % 12] = (e % 12 == 9 + e / 12) ? (float)P.dt : 0.f;
  if (lt < 12) sm.q2[lt] = 2.f * (float)P.Q[lt];
  if constexpr (PROF) t_setup = clock64() - t_start;

  // ------------------------------------------------------------------ C. constraint data (own variable, per step)
  float lb[NP], ub[NP], cmu[NP];
  bool eqb[NP];
  const float R2v = (float)(c < 3 ? P.R2[3 * f + c] : P.R2[6 + 3 * f + (c - 3)]);
  {
    float ey[3], ez[3];                        // columns 1, 2 of R = eul2rotm(x_fb[0:3])  (REF:124-138, 193)
    {
      RT sr, cr, sp, cp, sy, cy;
      sincos(xfb[0], &sr, &cr);
      sincos(xfb[1], &sp, &cp);
      sincos(xfb[2], &sy, &cy);
      ey[0] = (float)(cy * sp * sr - sy * cr); ey[1] = (float)(sy * sp * sr + cy * cr); ey[2] = (float)(cp * sr);
      ez[0] = (float)(cy * sp * cr + sy * sr); ez[1] = (float)(sy * sp * cr - cy * sr); ez[2] = (float)(cp * cr);
    }
    if (lt == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { sm.eyz[a] = ey[a]; sm.eyz[3 + a] = ez[a]; }
      float G[6][6];
      general_rows(0.f, ey, ez, (float)P.lh, (float)P.lt, G);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int b2 = 0; b2 < 6; ++b2) { sm.Gu[r][b2] = (RT)G[r][b2]; sm.GuT[b2][r] = (RT)G[r][b2]; }
    }
    const int a = c < 3 ? c : c - 3;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      const float cont = (float)contact[((size_t)inst * H + jg[s]) * 2 + f];
      const float muf = mu_in ? mu_in[((size_t)inst * H + jg[s]) * 2 + f] : (float)P.mu;
      if (c == 0) sm.muf[j][f] = muf;
      ub[s] = cont * (float)(c < 3 ? P.f_max[a] : P.tau_max[a]);      // REF:240-249
      lb[s] = cont * (float)(c < 3 ? P.f_min[a] : P.tau_min[a]);
      eqb[s] = lb[s] == ub[s];
      cmu[s] = c == 2 ? -muf : 0.f;
    }
  }

  // ------------------------------------------------------------------ D. factor: block-diagonal factors and the Riccati recursion
  float rvb[NP], rvg[NP];                     // penalties of this lane's box row / general row
  RT irvb[NP], irvg[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    rvb[s] = eqb[s] ? P.rho_eq : P.rho; rvg[s] = P.rho;
    irvb[s] = (RT)1 / (RT)rvb[s]; irvg[s] = (RT)1 / (RT)rvg[s];
  }

  // Blocks of the stage solve: S consecutive steps per block, NB <= 4 NW blocks, block `sb` on DPP row `sb` of the instance.
#ifdef BMPC_EMU
#define BMPC_OPAQUE(x) do { } while (0)
#else
#define BMPC_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#define BMPC_BLOCK_GEOMETRY                                                                                   \
  const int BS = (H + 4 * NW - 1) / (4 * NW); /* steps per block */                                           \
  const int NB = (H + BS - 1) / BS;           /* blocks */                                                    \
  const int sb = wv * 4 + (l >> 4);           /* block of this lane's DPP row */
  // row rn / column rn of M_i = F_i - I = [[0, C_i], [Kn_i]] for the lane's coordinate, zero for a step past the horizon
  // (rows 6..11 and their part of a column are Kn_i; the rest is C_i: its 3x3 block Cr_i and dt on the diagonal)
  // Loads only, no arithmetic on what was read (a product or a select on a loaded value would make the prefetch of the
  // next step wait for LDS inside this one): the zero parts come from `zblk`, the constant rows from `dtrow`, by address.
  auto m_row = [&](int rn, int i, bool ok, float (&cf)[12]) __attribute__((always_inline)) {
    const float* kp = rn >= 6 ? &sm.Kn[i][rn - 6][0] : rn >= 3 ? &sm.dtrow[rn - 3][0] : &sm.zblk[0];
    kp = ok ? kp : &sm.zblk[0];
    const float* cp = (ok && rn < 3) ? &sm.Cr[i][3 * rn] : kp + 6;
    const float4 v0 = *reinterpret_cast<const float4*>(kp);
    const float2 v1 = *reinterpret_cast<const float2*>(kp + 4);
    const float2 v2 = *reinterpret_cast<const float2*>(kp + 10);
    cf[0] = v0.x; cf[1] = v0.y; cf[2] = v0.z; cf[3] = v0.w; cf[4] = v1.x; cf[5] = v1.y;
    cf[6] = cp[0]; cf[7] = cp[1]; cf[8] = cp[2];
    cf[9] = kp[9]; cf[10] = v2.x; cf[11] = v2.y;
  };
  auto m_col = [&](int rn, int i, bool ok, float (&cf)[12]) __attribute__((always_inline)) {
    const float dtf_c = (float)P.dt;
    const float* kp = ok ? &sm.Kn[i][0][rn] : &sm.zblk[rn];
    const float* cp = (ok && rn >= 6 && rn < 9) ? &sm.Cr[i][rn - 6] : &sm.zblk[0];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      cf[a] = cp[3 * a];
      cf[3 + a] = (ok && rn == 9 + a) ? dtf_c : 0.f;
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) cf[6 + m] = kp[12 * m];
  };

  auto factor = [&]() __attribute__((always_inline)) {
    if constexpr (PROF) t_mark = clock64();
#pragma unroll
    for (int s = 0; s < NP; ++s) sm.rvg[BMPC_STEP(s)][f][c] = rvg[s];
    sync_all();                                 // (also: the shared tables of the set-up, the iteration's exchange vectors are dead)
    int co = c;
    BMPC_OPAQUE(co);
    double mkd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mkd[k] = (co == k) ? 1.0 : 0.0;
    const float lh = (float)P.lh, lt = (float)P.lt;
    float ey[3], ez[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ey[a] = sm.eyz[a]; ez[a] = sm.eyz[3 + a]; }
    const double idt = 1.0 / (double)P.dt, mdt = (double)P.m / (double)P.dt;
    const int qs = wv * 5 + q;                  // scratch slot of this lane group
    // ---- 6x6 block algebra in f64, one pass (5 steps) at a time through the scratch; bmpc_kernels.hip has the
    // derivation.  Here the wrench space is the ACCELERATION space a = E gamma: L~ = L E^-1, Ft = E^-T F E^-1.
#pragma unroll 1
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      const float muf = sm.muf[j][f];
      float rf[2][3];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int a = 0; a < 3; ++a) rf[ft][a] = (float)sm.rr[j][ft][a];
      double Ei[9];                              // E^-1 torque block: Iw / dt = (dt Iw^-1)^-1 (3x3 cofactors)
      {
        double w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = sm.Iwi[j][k];
        const double c00 = w[4] * w[8] - w[5] * w[7], c01 = w[5] * w[6] - w[3] * w[8], c02 = w[3] * w[7] - w[4] * w[6];
        const double id = idt / (w[0] * c00 + w[1] * c01 + w[2] * c02);
        Ei[0] = c00 * id; Ei[1] = (w[2] * w[7] - w[1] * w[8]) * id; Ei[2] = (w[1] * w[5] - w[2] * w[4]) * id;
        Ei[3] = c01 * id; Ei[4] = (w[0] * w[8] - w[2] * w[6]) * id; Ei[5] = (w[2] * w[3] - w[0] * w[5]) * id;
        Ei[6] = c02 * id; Ei[7] = (w[1] * w[6] - w[0] * w[7]) * id; Ei[8] = (w[0] * w[4] - w[1] * w[3]) * id;
      }
      double Tm[6][6];                           // T = [[I, 0], [[dr]x, I]]: (f2, m2) = -T (phi, nu) spans null(W)
      {
        const double dr[3] = {(double)rf[0][0] - rf[1][0], (double)rf[0][1] - rf[1][1], (double)rf[0][2] - rf[1][2]};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int r = 0; r < 6; ++r) Tm[p][r] = (p == r) ? 1.0 : 0.0;
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
      // D_f = 2R + A' diag(rv) A: row c of the own foot's block
      double m3[6];
      {
        float G[6][6];
        general_rows(muf, ey, ez, lh, lt, G);
        double wc[6];                            // rho_r * G[r][c]
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const double gc = (double)sm.GuT[co][r] - ((co == 2 && r < 4) ? (double)muf : 0.0);
          wc[r] = (double)sm.rvg[j][f][r] * gc;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          double sacc = 0.0;
#pragma unroll
          for (int r = 0; r < 6; ++r) sacc = fma(wc[r], (double)G[r][b], sacc);
          m3[b] = fma(mkd[b], (double)R2v + (double)rvb[s], sacc);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) (f == 0 ? sm.u.fac.M0 : sm.u.fac.M1)[qs][c][b] = m3[b];
      }
      BMPC_WAVE_SYNC();
      const bool on0 = hf == 0, on1 = hf == 1;
      double urow[6];                             // row c of U~ = E^-T W_0^-T D0
      double ka[6];                               // row c of Ka^-1
      if (on0) {
        double yq[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        row_times_mat6(Tcol, sm.u.fac.M1[qs], yq);
        double brow[6];                           // row c of B
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          double sacc = 0.0;
#pragma unroll
          for (int r = 0; r < 6; ++r) sacc = fma(yq[r], Tm[r][b], sacc);
          brow[b] = sacc;
          sm.u.fac.Ka[qs][c][b] = m3[b] + sacc;
        }
        // U~ = E^-T W_0^-T D0,  W_0^-T = [[0, I], [I, [r_0]x]]: row c of E^-T W_0^-T is a combination of rows of W_0^-T
        double wti[6];
        {
          const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
          double Wt[6][6];
#pragma unroll
          for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int r = 0; r < 6; ++r) Wt[p][r] = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a) { Wt[a][3 + a] = 1.0; Wt[3 + a][a] = 1.0; }
          Wt[3][4] = -r0[2]; Wt[3][5] = r0[1];
          Wt[4][3] = r0[2];  Wt[4][5] = -r0[0];
          Wt[5][3] = -r0[1]; Wt[5][4] = r0[0];
          // E^-T = blkdiag(Iw / dt, (m / dt) I) (Iw symmetric): row c < 3 mixes rows 0..2 of W_0^-T
          double ecol[6];                         // column c of E^-1 = row c of E^-T
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            double v = 0.0;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) v = fma(mkd[cc], p < 3 ? Ei[3 * p + cc] : 0.0, v);
#pragma unroll
            for (int cc = 3; cc < 6; ++cc) v = fma(mkd[cc], p == cc ? mdt : 0.0, v);
            ecol[p] = v;
          }
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            double a1 = 0.0;
#pragma unroll
            for (int p = 0; p < 6; ++p) a1 = fma(ecol[p], Wt[p][r], a1);
            wti[r] = a1;
          }
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) urow[b] = 0.0;
        row_times_mat6(wti, sm.u.fac.M0[qs], urow);
#pragma unroll
        for (int b = 0; b < 6; ++b) sm.u.fac.M2[qs][c][b] = brow[b];
      }
      BMPC_WAVE_SYNC();                         // Ka, B published; D1 consumed
      inv6_row(sm.u.fac.Ka[qs], co, ka);           // both lanes of the row, each for itself: no exchange
      if (on0) {
#pragma unroll
        for (int b = 0; b < 6; ++b) sm.u.fac.M1[qs][c][b] = ka[b];     // the whole Ka^-1 is needed for T Ka^-1 below
      }
      double xk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // lane 0: row c of Ka^-1 B ; lane 1: row c of Ka^-1 D0
      if (on0) row_times_mat6(ka, sm.u.fac.M2[qs], xk);
      if (on1) row_times_mat6(ka, sm.u.fac.M0[qs], xk);
      BMPC_WAVE_SYNC();                         // B, D0 consumed; Ka^-1 published
      {
        const double r0[3] = {(double)rf[0][0], (double)rf[0][1], (double)rf[0][2]};
        // (v W_0^-1) for a row v = [p, q]: [q, p - q x r_0];  then (.) E^-1
        double w0[6], cr[3], wE[6];
        const double q1[3] = {xk[3], xk[4], xk[5]};
        cross3(q1, r0, cr);
        w0[0] = xk[3]; w0[1] = xk[4]; w0[2] = xk[5];
        w0[3] = xk[0] - cr[0]; w0[4] = xk[1] - cr[1]; w0[5] = xk[2] - cr[2];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          wE[b] = w0[0] * Ei[b] + w0[1] * Ei[3 + b] + w0[2] * Ei[6 + b];
          wE[3 + b] = w0[3 + b] * mdt;
        }
        if (hf == 0) {
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            sm.LG[0].d[j][c][b] = (float)wE[b];
            sm.u.fac.M2[qs][c][b] = wE[b];         // L~_0 rows for Ft
          }
        } else {
#pragma unroll
          for (int b = 0; b < 6; ++b) sm.u.fac.M0[qs][c][b] = wE[b];   // Ka^-1 D0 W_0^-1 E^-1 rows for L~_1
        }
      }
      BMPC_WAVE_SYNC();
      if (on0) {
        double fv64[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        row_times_mat6(urow, sm.u.fac.M2[qs], fv64);     // Ft = U~ L~_0
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          sm.KG[0].d[j][c][b] = (float)ka[b];
          sm.Kn[j][c / 2][6 * (c % 2) + b] = (float)fv64[b];          // Ft[c][b], parked in the first 36 floats of Kn_j
        }
      }
      if (on1) {
        double sl[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        row_times_mat6(Trow, sm.u.fac.M0[qs], sl);       // L~_1 = T (Ka^-1 D0 W_0^-1 E^-1)
        row_times_mat6(Trow, sm.u.fac.M1[qs], sk);       // T Ka^-1
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          sm.LG[1].d[j][c][b] = (float)sl[b];
          sm.KG[1].d[j][c][b] = (float)sk[b];
        }
      }
      BMPC_WAVE_SYNC();                         // the scratch is reused by the next pass
    }
    if constexpr (PROF) { const long long t = clock64(); t_blocks += t - t_mark; t_mark = t; }

    // ---- backward Riccati recursion over the steps (f32; oracle/riccati_model.py::factor).  Every lane works, three
    // exchanges per step:
    //   R2  column jn of [M | Ft | I] solved against S = Ft + Pi22 (each lane its own LDL'): gain K, T = S^-1 Ft, S^-1
    //   R3  Schur complement Z = Pi - Pi[:,2] S^-1 Pi[2,:], cancellation-free in the (., 2) blocks: Z12 = Pi12 T, Z22 = Pi22 T
    //   R4  P = A' Z A:  P11 = Z11, P12 = Z11 C + Z12, P22 = C' P12 + Z12' C + Z22
    sync_all();                                 // every step's Ft is in place
    if (wv == 0) {                              // (wave-uniform)
      for (int e = l; e < 144; e += 64) sm.Pm[e / 12][e % 12] = 0.f;
      if (l < 36) {                             // S of the last step: Ft + Pi22 with P = 0 (same sum order as below)
        const int a = l / 6, b2 = l % 6;
        const float ft = sm.Kn[H - 1][(6 * a + b2) / 12][(6 * a + b2) % 12];
        sm.Sm[a][b2] = ft + 0.f + (a == b2 ? sm.q2[6 + a] : 0.f);
      }
      BMPC_WAVE_SYNC();
      const float dtf = (float)P.dt;
      const int jn = l < 24 ? l : 23;           // column of [M | Ft | I] this lane solves (lanes 24.. repeat column 23)
      // R3: entry e = l and l + 64 of the 108 entries [Z11 | Z12 | Z22]: base + sum_m Lrow[m] Rcol[m]
      int z_a[2], z_b[2], z_blk[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e = (l + 64 * t) < 108 ? l + 64 * t : 107;
        z_blk[t] = e / 36; z_a[t] = (e % 36) / 6; z_b[t] = e % 6;
      }
      // R4: lanes 0..35 own P22[k][k2], lanes 28..63 own P12[a2][k2b].  C_i = dt blkdiag(Rinv_i, I) enters through (row,
      // coefficient) triples: column k < 3 of C has rows 0..2, column k >= 3 the single entry dt in row k.
      const int pk = (l < 36 ? l : 35) / 6, pk2 = (l < 36 ? l : 35) % 6;
      const int qa = (l >= 28 ? l - 28 : 0) / 6, qk = (l >= 28 ? l - 28 : 0) % 6;
      auto ctriple = [&](int i, int k, int (&row)[3], float (&cf)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          row[t] = k < 3 ? t : k;
          cf[t] = k < 3 ? sm.Cr[i][3 * t + k] : (t == 0 ? dtf : 0.f);
        }
      };
      // Ft_i[a][b] sits in the first 36 floats of Kn_i until the gain overwrites them
      // loop-invariant pieces of 2Q, read once (the compiler does not hoist LDS loads over the recursion's stores)
      float q_za[2], q_zb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ro = z_blk[t] == 2 ? 6 : 0;
        q_za[t] = sm.q2[ro + z_a[t]]; q_zb[t] = sm.q2[ro + z_b[t]];
      }
      const float q_pk = sm.q2[6 + pk];
      const float q_col = (jn >= 6 && jn < 12) ? sm.q2[jn] : (jn >= 18 ? 1.f : 0.f);   // unit part of the right-hand side column
#define BMPC_FT(a, b) sm.Kn[i][(6 * (a) + (b)) / 12][(6 * (a) + (b)) % 12]
#pragma unroll 1
      for (int i = H - 1; i >= 0; --i) {
        // (every stage issues all its LDS loads first -- the scheduling barrier keeps the compiler from sinking each load
        //  next to its use, which costs one exposed LDS round trip per group: a single wave has nobody to hide it)
        // R2
        {
          float S[6][6], rhs[6], x[6];
          {
            // S = Ft_i + Pi22 was summed by stage R4 of the step before (Sm).  Right-hand side: column jn of
            // [Pi21 | Pi21 C + Pi22 | Ft | I] = base + (Pi21 C) + unit part, every lane the same instructions: the base column
            // by address (Pi21, Pi22, Ft or zeros), the C product with zero coefficients where the column has none
            const int k = jn < 12 ? (jn < 6 ? jn : jn - 6) : 0;
            int row[3];
            float cf[3];
            ctriple(i, k, row, cf);
            const bool m2 = jn >= 6 && jn < 12;
#pragma unroll
            for (int t = 0; t < 3; ++t) cf[t] = m2 ? cf[t] : 0.f;
            const int kf = (jn >= 12 && jn < 18) ? jn - 12 : 0, ki = jn >= 18 ? jn - 18 : -1;
            const float* bp = jn < 6 ? &sm.Pm[6][jn] : (jn < 12 ? &sm.Pm[6][jn] : (jn < 18 ? &sm.Kn[i][0][kf] : &sm.zblk[0]));
            const int bs = jn < 12 ? 12 : (jn < 18 ? 6 : 0);     // (Ft_i[m][kf] sits at flat offset 6 m + kf of Kn_i)
            const float dq = q_col;
            const int dm = m2 ? k : ki;
            float base[6], pc0[6], pc1[6], pc2[6];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int b2 = 0; b2 <= a; ++b2) S[a][b2] = sm.Sm[a][b2];
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              base[m] = bp[m * bs];
              pc0[m] = sm.Pm[6 + m][row[0]]; pc1[m] = sm.Pm[6 + m][row[1]]; pc2[m] = sm.Pm[6 + m][row[2]];
            }
            BMPC_SCHED_BARRIER();
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              const float pc = pc0[m] * cf[0] + pc1[m] * cf[1] + pc2[m] * cf[2];
              rhs[m] = (pc + base[m]) + (m == dm ? dq : 0.f);
            }
          }
          BMPC_WAVE_SYNC();                       // Ft_i consumed by every lane before K_i lands on it
          ldl6_solve(S, rhs, x);
          // K -> Kn_i (negated), T -> Tm, S^-1 -> Sinv_i: one store address per lane
          float* dst = jn < 12 ? &sm.Kn[i][0][jn] : (jn < 18 ? &sm.Tm[0][jn - 12] : &sm.Sinv[i][0][jn - 18]);
          const int stride = jn < 12 ? 12 : 6;
          const float sg = jn < 12 ? -1.f : 1.f;
#pragma unroll
          for (int m = 0; m < 6; ++m) dst[m * stride] = sg * x[m];
        }
        BMPC_WAVE_SYNC();
        // R3 (Z11 and Z22 are symmetrised: both lanes of a mirrored pair of entries form the same two sums, so the
        // cost-to-go stays exactly symmetric over the h steps -- without it the f32 asymmetry grows with the horizon
        // and a few h = 40 instances in 4096 lose their convergence)
        {
          float zv[2];
          float la[2][6], lb[2][6], ra_[2][6], rb_[2][6], pab[2], pba[2], qa2[2], qb2[2], rda[2], rdb[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int blk = z_blk[t], a = z_a[t], b = z_b[t];
            const int rs = blk == 0 ? 12 : 6;
            const float* rbase = blk == 0 ? &sm.Kn[i][0][0] : &sm.Tm[0][0];
            const int ro = blk == 2 ? 6 : 0;
            qa2[t] = q_za[t]; qb2[t] = q_zb[t];
            pab[t] = sm.Pm[a][b]; pba[t] = sm.Pm[b][a];
            rda[t] = rbase[a * rs + b]; rdb[t] = rbase[b * rs + a];      // (the 2Q diagonal of Pi22 meets these)
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              la[t][m] = sm.Pm[ro + a][6 + m]; lb[t][m] = sm.Pm[ro + b][6 + m];
              ra_[t][m] = rbase[m * rs + b]; rb_[t][m] = rbase[m * rs + a];
            }
          }
          BMPC_SCHED_BARRIER();
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int blk = z_blk[t], a = z_a[t], b = z_b[t];
            const float dq = (blk == 0 && a == b) ? qa2[t] : 0.f;
            float v1 = blk == 0 ? pab[t] + dq : 0.f, v2 = blk == 0 ? pba[t] + dq : 0.f;
#pragma unroll
            for (int m = 0; m < 6; ++m) { v1 = fmaf(la[t][m], ra_[t][m], v1); v2 = fmaf(lb[t][m], rb_[t][m], v2); }
            if (blk == 2) { v1 = fmaf(qa2[t], rda[t], v1); v2 = fmaf(qb2[t], rdb[t], v2); }
            zv[t] = blk == 1 ? v1 : 0.5f * (v1 + v2);
          }
          BMPC_WAVE_SYNC();                       // (emulation: Z11 replaces Pi11 in place, all reads first)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int blk = z_blk[t], a = z_a[t], b = z_b[t];
            float* d1 = blk == 0 ? &sm.Pm[a][b] : (blk == 1 ? &sm.Zm[a][6 + b] : &sm.Zm[6 + a][6 + b]);
            if (t == 0 || l + 64 < 108) *d1 = zv[t];
          }
        }
        BMPC_WAVE_SYNC();
        // R4
        {
          float p22 = 0.f, p12 = 0.f;
          int ra[3], rb[3], rq[3];
          float ca[3], cb[3], cq[3];
          ctriple(i, pk, ra, ca);
          ctriple(i, pk2, rb, cb);
          ctriple(i, qk, rq, cq);
          float z_v = sm.Zm[6 + pk][6 + pk2], z_w = sm.Zm[6 + pk2][6 + pk], z_q = sm.Zm[qa][6 + qk];
          const int inx = i > 0 ? i - 1 : 0;
          const float ft_next = sm.Kn[inx][(6 * pk + pk2) / 12][(6 * pk + pk2) % 12];      // Ft of the step reached next
          float zq12[3], zr12[3], zbk[3], zak[3], pab[3][3], pba[3][3], pq[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            zq12[t] = sm.Zm[ra[t]][6 + pk2];                            // Z12 part of P12[ra[t]][pk2]
            zr12[t] = sm.Zm[rb[t]][6 + pk];                             //            P12[rb[t]][pk]
            zbk[t] = zr12[t];                                           // (Z12' C)[pk][pk2] reads the same entries
            zak[t] = zq12[t];
            pq[t] = sm.Pm[qa][rq[t]];
#pragma unroll
            for (int u = 0; u < 3; ++u) { pab[t][u] = sm.Pm[ra[t]][rb[u]]; pba[t][u] = sm.Pm[rb[t]][ra[u]]; }
          }
          BMPC_SCHED_BARRIER();
          {
            // entry (pk, pk2) and its mirror, averaged
            float v = z_v, w = z_w;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              float q12 = zq12[t], r12 = zr12[t];
#pragma unroll
              for (int u = 0; u < 3; ++u) { q12 = fmaf(pab[t][u], cb[u], q12); r12 = fmaf(pba[t][u], ca[u], r12); }
              v = fmaf(ca[t], q12, v);
              v = fmaf(zbk[t], cb[t], v);
              w = fmaf(cb[t], r12, w);
              w = fmaf(zak[t], ca[t], w);
            }
            p22 = 0.5f * (v + w);
            float vq = z_q;
#pragma unroll
            for (int u = 0; u < 3; ++u) vq = fmaf(pq[u], cq[u], vq);
            p12 = vq;
          }
          BMPC_WAVE_SYNC();                       // (emulation: all reads of Pm / Zm before the stores)
          if (l < 36) {
            sm.Pm[6 + pk][6 + pk2] = p22;
            if (i > 0)                              // S of step i - 1: Ft + Pi22 (Pi22 = 2Q + P22), same sum order as the first step's
              sm.Sm[pk][pk2] = ft_next + p22 + (pk == pk2 ? q_pk : 0.f);
          }
          if (l >= 28) { sm.Pm[qa][6 + qk] = p12; sm.Pm[6 + qk][qa] = p12; }
        }
        BMPC_WAVE_SYNC();
      }
#undef BMPC_FT
    }
    sync_all();
    // ---- block products A_b = F_last .. F_first, one block per DPP row (row rn of A_b in 12 registers; a step multiplies
    // from the left: column c of the new A is (I + M_i) times column c of the old one, a row-broadcast mat-vec per column)
    {
      BMPC_BLOCK_GEOMETRY
      float A[12];
#pragma unroll
      for (int cidx = 0; cidx < 12; ++cidx) A[cidx] = (rn == cidx) ? 1.f : 0.f;
#pragma unroll 1
      for (int t = 0; t < BS; ++t) {
        const int i = sb * BS + t;
        const bool ok = sb < NB && i < H;
        float cf[12];
        m_row(rn, i < H ? i : H - 1, ok, cf);
#pragma unroll
        for (int cidx = 0; cidx < 12; ++cidx) A[cidx] = row_matvec12(A[cidx], A[cidx], cf);
      }
#pragma unroll
      for (int cidx = 0; cidx < 12; cidx += 4)
        *reinterpret_cast<float4*>(&sm.Ab[sb][rn][cidx]) = float4{A[cidx], A[cidx + 1], A[cidx + 2], A[cidx + 3]};
    }
    sync_all();
    if constexpr (PROF) t_ric += clock64() - t_mark;
  };

  int nfac = 0;
  bool need_factor = true;
  // Secant extrapolation at the stopping tests (bmpc_kernels.hip), in the form that needs no memory
  // but NP registers: the secant coefficient is taken from the x part of the state alone (the model: 52.9 iterations against
  // 53.0 with the full (x, z, y) metric), so the history is the x change of the iteration before a test; and at a test the
  // update runs in two passes -- statistics first, from temporaries, the commit after the reduction, when gamma is known and
  // the OLD state is still in its registers: nothing is parked.  Not with five steps per lane (no registers).
  constexpr bool AA = NP <= 4;
  const int AA_MAX_FACTOR = P.adapt_every <= 10 && H > 20 ? 16 : 8;   // (the long horizons re-classify twice as often: 8 factorisations are their mean)
  constexpr float AA_GAMMA_MAX = 100.f;
  bool aa_have = false;
  float aa_gx[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) aa_gx[s] = 0.f;

  // ------------------------------------------------------------------ E. ADMM iterations
  RT xo[NP], zb[NP], zg[NP], yb[NP], yg[NP], axg[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) { xo[s] = 0; zb[s] = 0; zg[s] = 0; yb[s] = 0; yg[s] = 0; axg[s] = 0; }
  const RT alpha = (RT)P.alpha;
  int it = 0, status = 1;
  const int check_every = P.check_every > 0 ? P.check_every : 1;
  int next_check = 2 * check_every;
  constexpr int REFRESH_ITERS = 20;
  int next_refresh = REFRESH_ITERS;
  constexpr float FAR = 1.0e3f;
  int next_adapt = P.adapt_every > 0 ? P.adapt_start : 0x7fffffff;
  while (next_adapt < 1) next_adapt += P.adapt_every;
  int n_adapt = 0;                             // re-classifications taken (the two-rate schedule: DevParams::adapt_early)
  int prev_act = 0;                            // classes of this lane's rows at the previous re-classification (bits 2 s: box row, 2 s + 1: general row of step slot s)
  // (like the secant step, not with five steps per lane -- h = 21 .. 24: no registers; those horizons re-classify adapt_late apart
  //  after the early ones whatever the rows did, and confirm nothing.  Their defaults use neither.)
  constexpr bool FLIPS = NP <= 4;
  float res_p = 0.f, res_s = 0.f;
  const RT idt_r = (RT)1 / dt, dtm = dt / (RT)P.m;

  // exact axg and err from x (f64): err = e0 + Gam_t W x by two prefix sums over the steps
  auto refresh = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NP; ++s) sm.u.itv.wg[BMPC_STEP(s)][f][c] = xo[s];
    BMPC_WAVE_SYNC();
    RT v2[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      RT xblk[2][6], gu[6];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int b = 0; b < 6; ++b) xblk[ft][b] = sm.u.itv.wg[j][ft][b];
#pragma unroll
      for (int b = 0; b < 6; ++b) gu[b] = sm.Gu[c][b];
      {
        RT a = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) a += gu[b] * (f == 0 ? xblk[0][b] : xblk[1][b]);
        const RT negmu = c < 4 ? -(RT)sm.muf[j][f] : (RT)0;
        axg[s] = a + negmu * (f == 0 ? xblk[0][2] : xblk[1][2]);
      }
      // acceleration of state coordinate n >= 6 at step j: (E W x)[n - 6]
      RT acc = 0;
      if (n >= 9) {
        const int k = n - 9;
        acc = dtm * ((k == 0 ? xblk[0][0] : (k == 1 ? xblk[0][1] : xblk[0][2])) + (k == 0 ? xblk[1][0] : (k == 1 ? xblk[1][1] : xblk[1][2])));
      } else if (n >= 6) {
        RT t0[3], t1[3], tau[3];
        const RT r0[3] = {sm.rr[j][0][0], sm.rr[j][0][1], sm.rr[j][0][2]};
        const RT r1[3] = {sm.rr[j][1][0], sm.rr[j][1][1], sm.rr[j][1][2]};
        cross3(r0, &xblk[0][0], t0);
        cross3(r1, &xblk[1][0], t1);
#pragma unroll
        for (int k = 0; k < 3; ++k) tau[k] = t0[k] + t1[k] + xblk[0][3 + k] + xblk[1][3 + k];
        const int k = n - 6;
        acc = dt * (sm.Iwi[j][3 * k] * tau[0] + sm.Iwi[j][3 * k + 1] * tau[1] + sm.Iwi[j][3 * k + 2] * tau[2]);
      }
      v2[s] = acc;
      BMPC_PASS_FENCE(s);
    }
    prefix_incl(v2);                            // (w, v) part of the state response (zero for n < 6)
    // (e, p) part: sum_{i <= j} C_i xi2_{i-1}: the lanes n < 6 need xi2 of the step before, other coordinates
#pragma unroll
    for (int s = 0; s < NP; ++s) sm.u.itv.lam[BMPC_STEP(s)][n] = v2[s];
    sync_all();                                 // (the step before may belong to another wave)
    RT v1[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      RT d = 0;
      if (n < 6 && j > 0) {
        if (n < 3) d = dt * (sm.Rv[j][3 * n] * sm.u.itv.lam[j - 1][6] + sm.Rv[j][3 * n + 1] * sm.u.itv.lam[j - 1][7] + sm.Rv[j][3 * n + 2] * sm.u.itv.lam[j - 1][8]);
        else d = dt * sm.u.itv.lam[j - 1][6 + n];
      }
      v1[s] = d;
    }
    prefix_incl(v1);
#pragma unroll
    for (int s = 0; s < NP; ++s) err[s] = e0[s] + (n < 6 ? v1[s] : v2[s]);
  };

  if (warm.buf && warm.load) {                 // wave-uniform
    // Start from the previous solve of this batch slot (layout [B][HS][12][6]); see bmpc_kernels.hip.
    bool ok = true;
    double wv[NP][6];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      int jw = js[s] + warm.shift;
      jw = jw > H - 1 ? H - 1 : jw;
      const double* src = warm.buf + (((size_t)inst * HS + jw) * 12 + n) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) wv[s][k] = src[k];
      const float pb0 = __int_as_float(__double2loint(wv[s][5])), pg0 = __int_as_float(__double2hiint(wv[s][5]));
      ok = ok && pb0 > 0.f && pg0 > 0.f && pb0 < 3.0e38f && pg0 < 3.0e38f;
#pragma unroll
      for (int k = 0; k < 5; ++k) ok = ok && (fabs(wv[s][k]) < 1.0e300);
    }
    ok = wave_umax(ok ? 0u : 1u) == 0u;        // all of the instance's state or none of it
    if constexpr (NW > 1) ok = sync_workgroup_or(ok ? 0 : 1) == 0;
    if (ok) {
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const float pb0 = __int_as_float(__double2loint(wv[s][5])), pg0 = __int_as_float(__double2hiint(wv[s][5]));
        xo[s] = wv[s][0];
        zb[s] = fmin(fmax(wv[s][1], (RT)lb[s]), (RT)ub[s]);
        zg[s] = fmin(wv[s][2], (RT)0);
        yb[s] = wv[s][3];
        yg[s] = wv[s][4];
        const float hib = c < 3 ? P.rho_hi_f : P.rho_hi_m, hig = c < 4 ? P.rho_hi_f : P.rho_hi_m;
        rvb[s] = eqb[s] ? P.rho_eq : fminf(fmaxf(P.rho * powf(pb0 / P.rho, warm.theta), P.rho_lo), hib);
        rvg[s] = fminf(fmaxf(P.rho * powf(pg0 / P.rho, warm.theta), P.rho_lo), hig);
        irvb[s] = (RT)1 / (RT)rvb[s]; irvg[s] = (RT)1 / (RT)rvg[s];
      }
    }
    refresh();
    if (ok && warm.adapt_start > 0 && P.adapt_every > 0) next_adapt = warm.adapt_start;
    if (ok) next_check = check_every;
  }

#pragma unroll 1
  for (it = 0; it < P.max_iter;) {
    if (need_factor) {                         // wave-uniform
      factor();
      ++nfac;
      need_factor = false;
      aa_have = false;                         // another map: the stored change belongs to the old one
    }
    if constexpr (PROF) t_last = clock64();
    // --- P0: adjoint of the tracking error: lam = sum_{i >= j} (A_{i+1} .. A_{j+1})' 2Q err_i (acceleration space);
    //     the wrench-space gradient is gb = E' lam2.  Two suffix sums over the steps.
    {
      RT v[NP];
#pragma unroll
      for (int s = 0; s < NP; ++s) v[s] = (RT)2 * (RT)P.Q[n] * err[s];
      suffix_incl(v);                           // n < 6: lam1;  n >= 6: the own part of lam2
#pragma unroll
      for (int s = 0; s < NP; ++s) sm.u.itv.lam[BMPC_STEP(s)][n] = v[s];
      BMPC_WAVE_SYNC();
      RT cpl[NP];                               // C_{i}' lam1_{i} of the steps i > j enters lam2_j
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const Step j = BMPC_STEP(s);
        RT d = 0;
        if (n >= 9) d = dt * sm.u.itv.lam[j][n - 6];
        else if (n >= 6) {
          const int k = n - 6;
          d = dt * (sm.Rv[j][k] * sm.u.itv.lam[j][0] + sm.Rv[j][3 + k] * sm.u.itv.lam[j][1] + sm.Rv[j][6 + k] * sm.u.itv.lam[j][2]);
        }
        cpl[s] = d;
        BMPC_PASS_FENCE(s);
      }
      suffix_excl(cpl);
#pragma unroll
      for (int s = 0; s < NP; ++s)
        if (n >= 6) sm.u.itv.lam[BMPC_STEP(s)][n] = v[s] + cpl[s];
    }
    // --- P1: row residuals w = y + rho (A x - z)
    RT wb[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      wb[s] = yb[s] + widen(rvb[s]) * (xo[s] - zb[s]);
      sm.u.itv.wg[BMPC_STEP(s)][f][c] = yg[s] + widen(rvg[s]) * (axg[s] - zg[s]);
    }
    BMPC_WAVE_SYNC();
    BMPC_SSTAMP(0)
    // --- P2: KKT residual in control space r = W' gb + 2R x + A' w
    {
      RT gut[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) gut[r] = sm.GuT[c][r];
      const int a3 = c < 3 ? c : c - 3;
      const int i1 = a3 == 2 ? 0 : a3 + 1, i2 = a3 == 0 ? 2 : a3 - 1;
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const Step j = BMPC_STEP(s);
        RT l2[6], wq[6], iw[9];
#pragma unroll
        for (int k = 0; k < 6; ++k) { l2[k] = sm.u.itv.lam[j][6 + k]; wq[k] = sm.u.itv.wg[j][f][k]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) iw[k] = sm.Iwi[j][k];
        const RT rx0 = c < 3 ? sm.rr[j][f][i2] : (RT)0, rx1 = c < 3 ? sm.rr[j][f][i1] : (RT)0;
        BMPC_SCHED_BARRIER();
        RT gt[3];                                // torque part of gb = E' lam2
#pragma unroll
        for (int k = 0; k < 3; ++k) gt[k] = dt * (iw[k] * l2[0] + iw[3 + k] * l2[1] + iw[6 + k] * l2[2]);
        const RT g1 = i1 == 0 ? gt[0] : (i1 == 1 ? gt[1] : gt[2]), g2 = i2 == 0 ? gt[0] : (i2 == 1 ? gt[1] : gt[2]);
        const RT gta = a3 == 0 ? gt[0] : (a3 == 1 ? gt[1] : gt[2]);
        const RT gfa = dtm * (a3 == 0 ? l2[3] : (a3 == 1 ? l2[4] : l2[5]));
        const RT gsel = c < 3 ? gfa : gta;
        RT r = widen(R2v) * xo[s] + wb[s];
#pragma unroll
        for (int k = 0; k < 6; ++k) r += gut[k] * wq[k];
        r += widen(cmu[s]) * ((wq[0] + wq[1]) + (wq[2] + wq[3]));
        const RT wt = g1 * rx0 - g2 * rx1 + gsel;
        sm.u.itv.r32[j][f][c] = (float)(r + wt);
        BMPC_PASS_FENCE(s);
      }
    }
    BMPC_WAVE_SYNC();
    BMPC_SSTAMP(1)
    // --- P3: right-hand side of the stage solve bt = L~' r (own foot's part, summed over the pair), and the
    //     null-space part of the step: t = N' r = r_0 - T' r_1;  foot 0 gets Ka^-1 t, foot 1 -(T Ka^-1) t
    float ddk[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      float rj[2][6], lcol[6], kg[6];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int i = 0; i < 6; i += 2) {
          const float2 v = *reinterpret_cast<const float2*>(&sm.u.itv.r32[j][ft][i]);
          rj[ft][i] = v.x; rj[ft][i + 1] = v.y;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        lcol[i] = sm.LG[f].d[j][i][c];
        kg[i] = sm.KG[f].d[j][c][i];
      }
      const float d0 = (float)sm.rr[j][0][0] - (float)sm.rr[j][1][0], d1 = (float)sm.rr[j][0][1] - (float)sm.rr[j][1][1],
                  d2 = (float)sm.rr[j][0][2] - (float)sm.rr[j][1][2];
      BMPC_SCHED_BARRIER();
      float bsum = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) bsum = fmaf(lcol[i], f == 0 ? rj[0][i] : rj[1][i], bsum);
      bsum += pair_swap(bsum);
      sm.u.itv.bt[j][c] = bsum;                 // both lanes of the pair: same value
      float tn[6];
      tn[0] = rj[0][0] - rj[1][0] + (d1 * rj[1][5] - d2 * rj[1][4]);
      tn[1] = rj[0][1] - rj[1][1] + (d2 * rj[1][3] - d0 * rj[1][5]);
      tn[2] = rj[0][2] - rj[1][2] + (d0 * rj[1][4] - d1 * rj[1][3]);
      tn[3] = rj[0][3] - rj[1][3];
      tn[4] = rj[0][4] - rj[1][4];
      tn[5] = rj[0][5] - rj[1][5];
      float dd = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) dd = fmaf(kg[i], tn[i], dd);
      ddk[s] = f == 1 ? -dd : dd;
      BMPC_PASS_FENCE(s);
    }
    sync_all();
    BMPC_SSTAMP(2)
    // --- P4: stage solve.  A step maps the costate by p_i = F_i' p_{i+1} - r_i (r_i = M_i' [0; bt_i], g_i = p2_{i+1} - bt_i) and
    //     the state by xi_i = F_i xi_{i-1} - [0; w_i] (w = Sinv g; a_i = xi2_i - xi2_{i-1}): 2 h dependent 12x12 mat-vecs if taken
    //     step by step.  They are taken BLOCK-wise instead: per block b of S steps the affine map out = A_b in - c_b, whose
    //     c_b is a recurrence over the block's own steps; all blocks form theirs at the same time, one per DPP row; a short
    //     chain over the NB blocks gives every block's input; the blocks then replay their steps from it, again in parallel.
    //     Dependent depth S + NB + S instead of h (h = 40: 18).  Every mat-vec is `row_matvec12`: one state coordinate per
    //     lane of a DPP row, operands by row broadcast, no LDS round trip inside a chain.
    {
      // (five steps per lane: the lane's coordinates pass through an opaque asm, or the address arithmetic of the six loops
      //  below is hoisted out of the ITERATION loop, stays in registers across every other phase and spills; the smaller
      //  variants have the registers and are 1.5 % faster with the hoisting)
      int rn = rn_lane, wvq = wv, lq = l;
      if constexpr (NP > 4) { BMPC_OPAQUE(rn); BMPC_OPAQUE(wvq); BMPC_OPAQUE(lq); }
      const int BS = (H + 4 * NW - 1) / (4 * NW), NB = (H + BS - 1) / BS, sb = wvq * 4 + (lq >> 4);
      const int r6 = rn >= 6 ? rn - 6 : 0;      // component of bt / w the lane reads (0 for the lanes that read none)
      const bool bok = sb < NB;
      // r_i = M_i' [0; bt_i] = Kn_i' bt_i: per step, by the step's own lanes
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const Step j = BMPC_STEP(s);
        float acc = 0.f;
        if (sreal[s]) {
#pragma unroll
          for (int m = 0; m < 6; ++m) acc = fmaf(sm.Kn[j][m][n], sm.u.itv.bt[j][m], acc);
        }
        sm.u.itv.rbw[j][n] = acc;
        BMPC_PASS_FENCE(s);
      }
      sync_all();
      // two steps per trip; the coefficients AND the vector operands of the next step are requested before this step's
      // mat-vec (LDS returns in order: a read issued inside a step would wait for the whole prefetch in front of it)
      float ca[12], cb[12], va[2], vb[2];
      // B2: c_b of the backward map: y <- F_i' y + r_i over the block's steps, last step first
      auto loadB = [&](int t, float (&cf)[12], float (&v)[2]) __attribute__((always_inline)) {
        const int i = sb * BS + t;               // (t < 0: the prefetch past the last step; loaded and dropped)
        const bool ok = bok && i < H;
        const int ic = i < 0 ? 0 : (i < H ? i : H - 1);
        m_col(rn, ic, ok, cf);
        const float* rp = ok ? &sm.u.itv.rbw[ic][rn] : &sm.zblk[0];
        v[0] = *rp;
        v[1] = sm.u.itv.bt[ic][r6];
      };
      {
        float y = 0.f;
        auto stepB = [&](const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          y = row_matvec12(y + v[0], y, cf);
        };
        int t = BS - 1;
        loadB(t, ca, va);
#pragma unroll 1
        for (; t >= 1; t -= 2) {
          loadB(t - 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepB(ca, va);
          loadB(t - 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepB(cb, vb);
        }
        if (t == 0) stepB(ca, va);
        sm.blkv[sb][rn] = y;
      }
      BMPC_SSTAMP(6)
      sync_all();
      // chain over the blocks, last block first: p_in(b) is stored, p <- A_b' p - c_b
      if (wv == 0) {
        float p = 0.f;
        auto loadC = [&](int b, float (&cf)[12], float (&v)[2]) __attribute__((always_inline)) {
          const int bc = b >= 0 ? b : 0;
#pragma unroll
          for (int m = 0; m < 12; ++m) cf[m] = sm.Ab[bc][m][rn];
          v[0] = sm.blkv[bc][rn];
        };
        auto stepC = [&](int b, const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          sm.blkp[b][rn] = p;
          p = row_matvec12(-v[0], p, cf);     // (A_b is the full block map: the accumulator starts from -c_b alone)
        };
        int b = NB - 1;
        loadC(b, ca, va);
#pragma unroll 1
        for (; b >= 1; b -= 2) {
          loadC(b - 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepC(b, ca, va);
          loadC(b - 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepC(b - 1, cb, vb);
        }
        if (b == 0) stepC(0, ca, va);
      }
      sync_all();
      // B3: replay of the blocks from their inputs: g_i = p2 - bt_i, then p <- F_i' p - r_i
      {
        float y = sm.blkp[bok ? sb : 0][rn];
        y = bok ? y : 0.f;
        auto stepR = [&](int t, const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          const int i = sb * BS + t;
          const bool ok = bok && i < H;
          sm.u.itv.gs[ok ? i : 0][(ok && rn >= 6) ? rn - 6 : 6] = y - v[1];      // g_i (slot 6: dump)
          y = row_matvec12(y - v[0], y, cf);
        };
        int t = BS - 1;
        loadB(t, ca, va);
#pragma unroll 1
        for (; t >= 1; t -= 2) {
          loadB(t - 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepR(t, ca, va);
          loadB(t - 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepR(t - 1, cb, vb);
        }
        if (t == 0) stepR(0, ca, va);
      }
      sync_all();
      // w = Sinv g, one component per lane pair (both lanes of a pair compute the same value); it replaces bt
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const Step j = BMPC_STEP(s);
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc = fmaf(sm.Sinv[j][c][m], sm.u.itv.gs[j][m], acc);
        sm.u.itv.bt[j][c] = acc;
        BMPC_PASS_FENCE(s);
      }
      sync_all();
      // F2: c_b of the forward map: y <- F_i y + [0; w_i] over the block's steps, first step first
      auto loadF = [&](int t, float (&cf)[12], float (&v)[2]) __attribute__((always_inline)) {
        const int i = sb * BS + t;               // (t >= BS: the prefetch past the last step; loaded and dropped)
        const bool ok = bok && i < H;
        const int ic = i < H ? i : H - 1;
        m_row(rn, ic, ok, cf);
        const float* wp = (ok && rn >= 6) ? &sm.u.itv.bt[ic][rn - 6] : &sm.zblk[0];
        v[0] = *wp;
      };
      {
        float y = 0.f;
        auto stepF = [&](const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          y = row_matvec12(y + v[0], y, cf);
        };
        int t = 0;
        loadF(0, ca, va);
#pragma unroll 1
        for (; t + 2 <= BS; t += 2) {
          loadF(t + 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepF(ca, va);
          loadF(t + 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepF(cb, vb);
        }
        if (t < BS) stepF(ca, va);
        sm.blkv[sb][rn] = y;
      }
      sync_all();
      // chain over the blocks, first block first: xi_in(b) is stored, xi <- A_b xi - c_b
      if (wv == 0) {
        float x = 0.f;
        auto loadC = [&](int b, float (&cf)[12], float (&v)[2]) __attribute__((always_inline)) {
          const int bc = b < NB ? b : 0;
#pragma unroll
          for (int m = 0; m < 12; m += 4) {
            const float4 q = *reinterpret_cast<const float4*>(&sm.Ab[bc][rn][m]);
            cf[m] = q.x; cf[m + 1] = q.y; cf[m + 2] = q.z; cf[m + 3] = q.w;
          }
          v[0] = sm.blkv[bc][rn];
        };
        auto stepC = [&](int b, const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          sm.blkp[b][rn] = x;
          x = row_matvec12(-v[0], x, cf);
        };
        int b = 0;
        loadC(0, ca, va);
#pragma unroll 1
        for (; b + 2 <= NB; b += 2) {
          loadC(b + 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepC(b, ca, va);
          loadC(b + 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepC(b + 1, cb, vb);
        }
        if (b < NB) stepC(b, ca, va);
      }
      sync_all();
      // F3: replay: xi_i = F_i xi_{i-1} - [0; w_i]; the acceleration is the increment of the (w, v) half
      {
        float y = sm.blkp[bok ? sb : 0][rn];
        y = bok ? y : 0.f;
        auto stepP = [&](int t, const float (&cf)[12], const float (&v)[2]) __attribute__((always_inline)) {
          const int i = sb * BS + t;
          const bool ok = bok && i < H;
          const int ic = ok ? i : 0;
          const float yn = row_matvec12(y - v[0], y, cf);
          sm.u.itv.av[ic][(ok && rn >= 6) ? rn - 6 : 6] = yn - y;
          if (ok) sm.u.itv.xi[ic][rn] = yn;
          y = yn;
        };
        int t = 0;
        loadF(0, ca, va);
#pragma unroll 1
        for (; t + 2 <= BS; t += 2) {
          loadF(t + 1, cb, vb);
          BMPC_SCHED_BARRIER();
          stepP(t, ca, va);
          loadF(t + 2, ca, va);
          BMPC_SCHED_BARRIER();
          stepP(t + 1, cb, vb);
        }
        if (t < BS) stepP(t, ca, va);
      }
    }
    sync_all();
    BMPC_SSTAMP(3)
    // (penalty re-classification of step s from given (z, y) values: active rows move up by kappa towards their class
    //  ceiling, inactive ones down towards rho_lo; damped after many factorisations -- bmpc_kernels.hip)
    //  `scheduled`: a re-classification of the schedule (not the one the third stopping test forces): from number confirm_from + 1
    //  on a row found in the same class as at the previous one moves by kappa_confirm -- bmpc_kernels.hip.  act2: the two classes.)
    auto reclassify_v = [&](int s, RT zbv, RT ybv, RT zgv, RT ygv, float& nb, float& ng, int& act2, const bool scheduled) __attribute__((always_inline)) {
      const float kap = nfac <= 10 ? P.kappa : (nfac <= 16 ? P.kappa_sqrt : P.kappa_qrt);
      const bool actb = (zbv <= (RT)lb[s] || zbv >= (RT)ub[s]) && ybv != (RT)0;
      const bool actg = (zgv >= (RT)0) && ygv != (RT)0;
      act2 = (actb ? 1 : 0) | (actg ? 2 : 0);
      const bool confirm = FLIPS && scheduled && P.kappa_confirm > 0.f && n_adapt >= P.confirm_from && n_adapt > 0 && nfac <= 10;   // (n_adapt: before this one)
      const int same = confirm ? ~(act2 ^ (prev_act >> (2 * s))) : 0;
      const float kapb = (same & 1) ? P.kappa_confirm : kap, kapg = (same & 2) ? P.kappa_confirm : kap;
      const float hib = c < 3 ? P.rho_hi_f : P.rho_hi_m, hig = c < 4 ? P.rho_hi_f : P.rho_hi_m;
      nb = eqb[s] ? P.rho_eq : (actb ? fminf(rvb[s] * kapb, hib) : fmaxf(rvb[s] / kapb, P.rho_lo));
      ng = actg ? fminf(rvg[s] * kapg, hig) : fmaxf(rvg[s] / kapg, P.rho_lo);
    };
    float nflip = 0.f;                        // this lane's rows in another class than at the previous re-classification
    // --- P5: x~ = x - d, z~ = A x~ (carried), relaxation, projection, dual update, tracking error
    float rp = 0.f, rs = 0.f, nz = 0.f, nx = 0.f, slw = 0.f;
    float r0 = 0.f, nx0 = 0.f, slw0 = 0.f;       // the rows of step 0 alone: the applied control has its own stopping test (bmpc_kernels.hip)
    const bool check_now = (it + 1 == next_check) || (it + 1 == P.max_iter);     // wave-uniform
    float dstep[NP];                          // d_f[c] = (null-space part) + L~ a
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const Step j = BMPC_STEP(s);
      float gm[6], lg[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        gm[i] = sm.u.itv.av[j][i];
        lg[i] = sm.LG[f].d[j][c][i];
      }
      BMPC_SCHED_BARRIER();
      float d = ddk[s];
#pragma unroll
      for (int i = 0; i < 6; ++i) d = fmaf(lg[i], gm[i], d);
      dstep[s] = d;
      sm.u.itv.r32[j][f][c] = d;                // (the residual is consumed: its slots carry the step to the foot's lanes)
      BMPC_PASS_FENCE(s);
    }
    BMPC_WAVE_SYNC();
    float gub[6];                             // row c of the mu-free general rows
#pragma unroll
    for (int b = 0; b < 6; ++b) gub[b] = (float)sm.Gu[c][b];
    const bool aa_keep = AA && P.accel != 0 && (it + 2 == next_check);           // the iteration before a stopping test
    const bool aa_now = AA && P.accel != 0 && check_now && aa_have && (it + 1 < P.max_iter) && nfac <= AA_MAX_FACTOR;
    const bool adapt_next = (it + 1 == next_adapt) && nfac <= P.max_refactor;    // this iteration ends on a re-classification
    float aa1 = 0.f, aa2 = 0.f, chg_t = 0.f;
    // one step of the update.  MODE 0: commit (an ordinary iteration); 1: the statistics of a stopping test from temporaries,
    // nothing committed; 2: commit after the test's reduction, moved along the step by -gam (gam = 0: the plain update).
    auto p5_step = [&](int s, auto mode_c, RT gam) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode_c)::value;
      const Step j = BMPC_STEP(s);
      float db[6];
#pragma unroll
      for (int i = 0; i < 6; i += 2) {
        const float2 v = *reinterpret_cast<const float2*>(&sm.u.itv.r32[j][f][i]);
        db[i] = v.x; db[i + 1] = v.y;
      }
      const float xin = sm.u.itv.xi[j][n];
      const float mus = c < 4 ? sm.muf[j][f] : 0.f;
      BMPC_SCHED_BARRIER();
      f2 dd;                                    // {d_f[c], (G_f d_f)[c]}
      dd.x = dstep[s];
      {
        float sg = -mus * db[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) sg = fmaf(gub[i], db[i], sg);
        dd.y = sg;
      }
      const RT xto = xo[s] - (RT)dd.x;
      const RT ztg = axg[s] - (RT)dd.y;
      const RT ztb = xto;
      RT st_pb, st_pg, znb, zng, ybn, ygn;
      {
        const RT zr = alpha * ztb + (1 - alpha) * zb[s];
        const RT cand = zr + yb[s] * irvb[s];
        znb = fmin(fmax(cand, widen(lb[s])), widen(ub[s]));
        ybn = yb[s] + widen(rvb[s]) * (zr - znb);
        st_pb = ztb - znb;
      }
      {
        const RT zr = alpha * ztg + (1 - alpha) * zg[s];
        const RT cand = zr + yg[s] * irvg[s];
        zng = fmin(cand, (RT)0);
        ygn = yg[s] + widen(rvg[s]) * (zr - zng);
        st_pg = ztg - zng;
      }
      if constexpr (MODE != 2) {
        if (check_now && sreal[s]) {
          rp = fmaxf(rp, fmaxf(fabsf((float)st_pb), fabsf((float)st_pg)));
          {
            const bool actb = (znb <= (RT)lb[s] || znb >= (RT)ub[s]) && ybn != (RT)0;
            const bool actg = (zng >= (RT)0) && ygn != (RT)0;
            slw = fmaxf(slw, fmaxf((actb || eqb[s]) ? 0.f : rvb[s] * fabsf((float)st_pb), actg ? 0.f : rvg[s] * fabsf((float)st_pg)));
          }
          nz = fmaxf(nz, fmaxf(fabsf((float)xto), fabsf((float)ztg)));
          rs = fmaxf(rs, fabsf((float)(xto - xo[s])));
          // a NaN iterate must reach the test (fmaxf drops NaNs): it is reported as an infinite norm
          nx = fmaxf(nx, (xto == xto) ? fabsf((float)xto) : __builtin_inff());
          if (s == 0 && js[0] == 0) {             // (s is a constant of the unrolled pass: only the first slot can be step 0)
            r0 = fmaxf(fmaxf(fabsf((float)st_pb), fabsf((float)st_pg)), fabsf((float)(xto - xo[s])));
            nx0 = fabsf((float)xto);
            const bool actb = (znb <= (RT)lb[s] || znb >= (RT)ub[s]) && ybn != (RT)0;
            const bool actg = (zng >= (RT)0) && ygn != (RT)0;
            slw0 = fmaxf((actb || eqb[s]) ? 0.f : rvb[s] * fabsf((float)st_pb), actg ? 0.f : rvg[s] * fabsf((float)st_pg));
          }
        }
      }
      if constexpr (MODE == 1) {
        // the secant sums (x part of the state) and the "some penalty moves" flag, from the temporaries
        if (sreal[s]) {
          const float gx = -(float)alpha * dd.x;
          const float d = gx - aa_gx[s];
          aa1 = fmaf(d, gx, aa1);
          aa2 = fmaf(d, d, aa2);
          if (adapt_next) {
            float nb, ng;
            int a2;
            reclassify_v(s, znb, ybn, zng, ygn, nb, ng, a2, true);
            chg_t = ((nb != rvb[s]) | (ng != rvg[s])) ? 1.f : chg_t;
            if constexpr (FLIPS) {
              const int fl = a2 ^ ((prev_act >> (2 * s)) & 3);
              nflip += (lane_real && n_adapt > 0) ? (float)((fl & 1) + (fl >> 1)) : 0.f;
            }
          }
        }
      } else {
        RT xn = alpha * xto + (1 - alpha) * xo[s];
        RT an = alpha * ztg + (1 - alpha) * axg[s];
        // W x~ = W x - gamma exactly, so the tracking error follows the state response of the solve (f32 increment: it
        // vanishes with the step; rebuilt exactly in f64 every REFRESH_ITERS iterations and before leaving)
        RT en = err[s] - alpha * (RT)xin;
        if constexpr (MODE == 2) {
          xn -= gam * (xn - xo[s]);
          an -= gam * (an - axg[s]);
          en -= gam * (en - err[s]);
          znb = fmin(fmax(znb - gam * (znb - zb[s]), widen(lb[s])), widen(ub[s]));
          zng = fmin(zng - gam * (zng - zg[s]), (RT)0);
          ybn -= gam * (ybn - yb[s]);
          ygn -= gam * (ygn - yg[s]);
        }
        xo[s] = xn; axg[s] = an; err[s] = en;
        zb[s] = znb; yb[s] = ybn; zg[s] = zng; yg[s] = ygn;
      }
      BMPC_PASS_FENCE(s);
    };
    const bool two_pass = AA && P.accel != 0 && check_now;      // (uniform) a stopping test with the extrapolation enabled
    if (two_pass) {
#pragma unroll
      for (int s = 0; s < NP; ++s) p5_step(s, std::integral_constant<int, 1>{}, (RT)0);
    } else {
#pragma unroll
      for (int s = 0; s < NP; ++s) p5_step(s, std::integral_constant<int, 0>{}, (RT)0);
    }
    if (aa_keep) {
#pragma unroll
      for (int s = 0; s < NP; ++s) aa_gx[s] = -(float)alpha * dstep[s];
      aa_have = true;
    }
    ++it;
    BMPC_SSTAMP(4)
    // --- stopping test and penalty re-classification (wave-uniform decisions; one wave: reductions by DPP alone)
    const bool adapt_now = (it == next_adapt);
    const bool adapt_do = adapt_now && nfac <= P.max_refactor;
    auto reclassify = [&](int s, float& nb, float& ng, int& a2, const bool scheduled) __attribute__((always_inline)) {
      reclassify_v(s, zb[s], yb[s], zg[s], yg[s], nb, ng, a2, scheduled);
    };
    bool force_adapt = false;
    float flips = 0.f;                        // rows of the instance that changed class since the previous re-classification
    if (check_now || adapt_do) {
      float chg = 0.f;
      if (two_pass) {
        chg = adapt_do ? chg_t : 0.f;           // (formed in the statistics pass: the state is not committed yet)
      } else if (adapt_do) {
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          float nb, ng;
          int a2;
          reclassify(s, nb, ng, a2, true);
          chg = (sreal[s] && ((nb != rvb[s]) | (ng != rvg[s]))) ? 1.f : chg;
          if constexpr (FLIPS) {
            const int fl = a2 ^ ((prev_act >> (2 * s)) & 3);
            nflip += (sreal[s] && lane_real && n_adapt > 0) ? (float)((fl & 1) + (fl >> 1)) : 0.f;
          }
        }
      }
      // (see bmpc_kernels.hip: the third stopping test -- the pull rho |z~ - z| of the inactive rows against the softest
      //  curvature; an instance that fails it re-classifies at once instead of stopping)
      constexpr float SLOW_TOL = 1.0e-6f;
      float v5[12] = {rp, rs, nz, nx, chg, slw, aa1, aa2, r0, nx0, slw0, adapt_do ? nflip : 0.f};
#pragma unroll
      for (int k = 0; k < 6; ++k) v5[k] = __uint_as_float(wave_umax(__float_as_uint(v5[k])));
#pragma unroll
      for (int k = 8; k < 11; ++k) v5[k] = __uint_as_float(row0_umax(__float_as_uint(v5[k])));   // (step 0: lanes 0 .. 11 of wave 0)
      if (two_pass) {                           // (uniform) the two secant sums ride in the same exchange
        v5[6] = wave_sum(v5[6]);
        v5[7] = wave_sum(v5[7]);
      }
      if (adapt_do) v5[11] = wave_sum(v5[11]);  // (uniform) ... and the count of rows that changed class
      if constexpr (NW > 1) {                   // combine the waves (two buffers: a buffer is rewritten after another barrier)
        float (*red)[NW] = sm.red[n_red & 1];
        ++n_red;
        if (l == 0) {
#pragma unroll
          for (int k = 0; k < 12; ++k) red[k][wv] = v5[k];
        }
        sync_workgroup();
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          if (k == 6 || k == 7) continue;
          unsigned m = __float_as_uint(red[k][0]);
#pragma unroll
          for (int w2 = 1; w2 < NW; ++w2) { const unsigned o = __float_as_uint(red[k][w2]); m = m > o ? m : o; }
          v5[k] = __uint_as_float(m);
        }
#pragma unroll
        for (int k = 6; k < 12; ++k) {
          if (k >= 8 && k < 11) continue;
          float a2 = red[k][0];
#pragma unroll
          for (int w2 = 1; w2 < NW; ++w2) a2 += red[k][w2];
          v5[k] = a2;
        }
      }
      flips = v5[11];
      if (check_now) {
        res_p = v5[0];
        res_s = v5[1];
        const float tol_p = P.eps_pri * fmaxf(1.f, v5[2]), tol_s = P.eps_dua * fmaxf(1.f, v5[3]);
        const bool bad = !(v5[0] == v5[0]) || !(v5[1] == v5[1]) || !(v5[3] < 3.0e38f);
        // (the rows of step 0 -- the applied control, REF:493 -- are also held to U0_TOL x eps relative to their own norm:
        //  bmpc_kernels.hip)
        constexpr float U0_TOL = 5.f;
        const float n0 = fmaxf(1.f, v5[9]);
        const bool small = v5[0] <= tol_p && v5[1] <= tol_s && v5[8] <= P.eps_u0 * n0;
        // (an instance that may not re-classify any more -- budget of factorisations spent, adaptation switched off -- is
        //  taken as it is: the third test can only be answered by a re-classification)
        const bool can_adapt = nfac <= P.max_refactor && P.adapt_every > 0;
        const bool done = small && (!(v5[5] > P.slow_tol_r2 * fmaxf(1.f, v5[3]) || v5[10] > P.slow_tol_r2_u0 * n0) || !can_adapt);
        force_adapt = small && !done && !bad && it < P.max_iter;
        const bool far = v5[0] > FAR * tol_p || v5[1] > FAR * tol_s;
        next_check += far ? 2 * check_every : check_every;
        if (two_pass) {                         // the commit of this iteration's update, along the secant where the instance goes on
          float gam = v5[6] / v5[7];
          const bool ext = aa_now && !done && !bad && v5[7] > 0.f && fabsf(gam) < AA_GAMMA_MAX;   // (NaN fails the comparison)
          gam = ext ? gam : 0.f;
#pragma unroll
          for (int s = 0; s < NP; ++s) p5_step(s, std::integral_constant<int, 2>{}, (RT)gam);
          aa_have = false;
        }
        const bool rebuild = it >= next_refresh;
        if (rebuild) next_refresh = it + REFRESH_ITERS;
        if (bad || done || it == P.max_iter || rebuild) refresh();
        if (bad) { status = 2; break; }
        if (done) { status = 0; break; }
      }
      if ((adapt_do && v5[4] > 0.f) || force_adapt) {
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          float nb, ng;
          int a2;
          reclassify(s, nb, ng, a2, adapt_do);
          rvb[s] = nb; rvg[s] = ng;
          irvb[s] = (RT)1 / (RT)nb; irvg[s] = (RT)1 / (RT)ng;
        }
        need_factor = true;
      }
    }
    if (adapt_now) {                          // the schedule: bmpc_kernels.hip
      if constexpr (FLIPS) {
        if (adapt_do) {                         // the classes as the instance leaves this re-classification (from the committed state:
          int an = 0;                           //  nothing is carried across the reduction for it)
#pragma unroll
          for (int s = 0; s < NP; ++s) {
            const bool actb = (zb[s] <= (RT)lb[s] || zb[s] >= (RT)ub[s]) && yb[s] != (RT)0;
            const bool actg = (zg[s] >= (RT)0) && yg[s] != (RT)0;
            an |= ((actb ? 1 : 0) | (actg ? 2 : 0)) << (2 * s);
          }
          prev_act = an;
        }
      }
      ++n_adapt;
      int period = P.adapt_every;
      if (P.adapt_late > 0 && n_adapt >= P.adapt_early)
        period = (P.adapt_busy > 0 && (int)flips > P.adapt_flips) ? P.adapt_busy : P.adapt_late;
      next_adapt += period;
    }
    BMPC_SSTAMP(5)
    BMPC_DRAIN_LDS();
  }
  if (warm.buf && warm.store) {
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      if (!(sreal[s] && lane_real)) continue;
      double* dst = warm.buf + (((size_t)inst * HS + js[s]) * 12 + n) * 6;
      dst[0] = xo[s]; dst[1] = zb[s]; dst[2] = zg[s]; dst[3] = yb[s]; dst[4] = yg[s];
      dst[5] = __hiloint2double(__float_as_int(rvg[s]), __float_as_int(rvb[s]));
    }
  }

  // ------------------------------------------------------------------ F. outputs (REF:300-304)
  // every way out of the loop rebuilt err exactly at its last stopping test: X = x_ref + err
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    if (!(sreal[s] && lane_real)) continue;
    const Step j = BMPC_STEP(s);
    const int pos = c < 3 ? 3 * f + c : 6 + 3 * f + (c - 3);       // [f1 f2 m1 m2]
    // (fp64 output arrays: the fp32 result widened, bmpc_kernels.hip WarmArgs::controls64)
    if (warm.controls64) warm.controls64[((size_t)inst * H + j) * 12 + pos] = (double)(float)xo[s];
    else controls[((size_t)inst * H + j) * 12 + pos] = (float)xo[s];
    if (states || warm.states64) {
      const size_t sbase = ((size_t)inst * H + j) * 13;
      const RT xrn = (j == 0) ? xfb_n : ((n < 6 && xc_n6 != (RT)0) ? xfb_n + xc_n6 * ((RT)j * dt) : xc_n);      // x_ref[n, j]
      const float sv = (float)(xrn + err[s]);
      if (warm.states64) { warm.states64[sbase + n] = (double)sv; if (n == 0) warm.states64[sbase + 12] = 1.0; }
      else { states[sbase + n] = sv; if (n == 0) states[sbase + 12] = 1.0f; }
    }
  }
  if (PROF && dbg.prof && lt == 0) {
    long long* pr = dbg.prof + (size_t)inst * 16;
    pr[0] = t_setup; pr[1] = t_blocks; pr[2] = t_ric; pr[3] = clock64() - t_start; pr[4] = it; pr[5] = nfac;
#pragma unroll
    for (int k = 0; k < 7; ++k) pr[8 + k] = t_ph[k];
  }
  if (lt == 0) {
    if (iters_out) iters_out[inst] = it;
    if (status_out) status_out[inst] = status;
    if (nfactor_out) nfactor_out[inst] = nfac;
    if (resid_out) { resid_out[2 * inst] = res_p; resid_out[2 * inst + 1] = res_s; }
  }
#undef BMPC_SSTAMP
#undef BMPC_PASS_FENCE
#undef BMPC_STEP
}

#define BMPC_STAGE_ARGS                                                                                            \
  const DevParams P, const int B, const float* __restrict__ x_fb, const float* __restrict__ foot,                  \
      const uint8_t* __restrict__ contact, const int32_t* __restrict__ phase, const float* __restrict__ x_cmd,     \
      const float* __restrict__ mu_in, float* __restrict__ controls, float* __restrict__ states,                   \
      int32_t* __restrict__ iters_out, float* __restrict__ resid_out, int32_t* __restrict__ status_out,            \
      int32_t* __restrict__ nfactor_out, const DebugOut dbg, const WarmArgs warm
template <int NP, int NW>
__global__ void __launch_bounds__(64 * NW) stage_kernel(BMPC_STAGE_ARGS) {
  stage_body<NP, NW, false>(P, B, x_fb, foot, contact, phase, x_cmd, mu_in, controls, states, iters_out, resid_out, status_out,
                            nfactor_out, dbg, warm);
}
template <int NP, int NW>
__global__ void __launch_bounds__(64 * NW) stage_kernel_prof(BMPC_STAGE_ARGS) {
  stage_body<NP, NW, true>(P, B, x_fb, foot, contact, phase, x_cmd, mu_in, controls, states, iters_out, resid_out, status_out,
                           nfactor_out, dbg, warm);
}
// steps per lane and waves per instance for horizon h: one wave up to h = 24 (NP = ceil(h / 5)), two from h = 26
__host__ __device__ constexpr int stage_waves(int h) { return h <= 24 ? 1 : 2; }
__host__ __device__ constexpr int stage_steps_per_lane(int h) {
  const int np = (h + 5 * stage_waves(h) - 1) / (5 * stage_waves(h));
  return np < 2 ? 2 : np;                      // (h <= 5: the smallest variant, most of its lane map phantoms)
}
#undef BMPC_STAGE_ARGS

}  // namespace bmpc
