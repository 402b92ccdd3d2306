// bmpc_lowlevel.hip -- batched consumer/producer either side of the MPC path:
//   foot_world_kernel   getFootPositionBody / getFootPositionWorld        REF:367-424
//   lowlevel_kernel     getLegKinematics, swingLegControl, lowLevelControl  REF:306-365, 426-470
// Closed-form trigonometry, one thread per instance, f64 arithmetic (a few hundred flops: the kernels are
// bound by their ~200 B/instance of I/O), f32 I/O like the rest of the ABI.  Quirks kept: R' is used for
// body->world and world->body alike (REF:423, 461, 465).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmpc {

struct LowLevelParams {
  double h, dt, kv, swing_height;
  double x_cmd[12], kp[9], kd[9], hip_offset[3];
};

__device__ __forceinline__ void eul2rotm_d(const double* e, double* R) {   // REF:111-138: Rz(e2) Ry(e1) Rx(e0)
  double sr, cr, sp, cp, sy, cy;
  sincos(e[0], &sr, &cr); sincos(e[1], &sp, &cp); sincos(e[2], &sy, &cy);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

// REF:367-404
__device__ __forceinline__ void foot_body(const double* q, double side, double* pf) {
  double s0, c0, s1, c1, s2, c2, s3, c3, s4, c4;
  sincos(q[0], &s0, &c0); sincos(q[1], &s1, &c1); sincos(q[2], &s2, &c2); sincos(q[3], &s3, &c3); sincos(q[4], &s4, &c4);
  const double u = c0 * s2 + c2 * s0 * s1, v = c0 * c2 - s0 * s1 * s2;
  const double w = s0 * s2 - c0 * c2 * s1, y = c2 * s0 + c0 * s1 * s2;
  pf[0] = -3 * c0 / 200 - 9 * s4 * (c3 * v - s3 * u) / 250 - 11 * c0 * s2 / 50 - side * s0 / 50 - 11 * c3 * u / 50 -
          11 * s3 * v / 50 - 9 * c4 * (c3 * u + s3 * v) / 250 - 23 * c1 * side * s0 / 1000 - 11 * c2 * s0 * s1 / 50;
  pf[1] = c0 * side / 50 - 9 * s4 * (c3 * y - s3 * w) / 250 - 3 * s0 / 200 - 11 * s0 * s2 / 50 - 11 * c3 * w / 50 -
          11 * s3 * y / 50 - 9 * c4 * (c3 * w + s3 * y) / 250 + 23 * c0 * c1 * side / 1000 + 11 * c0 * c2 * s1 / 50;
  pf[2] = 23 * side * s1 / 1000 - 11 * c1 * c2 / 50 - 9 * c4 * (c1 * c2 * c3 - c1 * s2 * s3) / 250 +
          9 * s4 * (c1 * c2 * s3 + c1 * c3 * s2) / 250 - 11 * c1 * c2 * c3 / 50 + 11 * c1 * s2 * s3 / 50 - 3.0 / 50.0;
}

// REF:306-365: Jm (6x5, row-major)
__device__ __forceinline__ void leg_jacobian(const double* q, double side, double* Jm) {
  double s0, c0, s1, c1, s2, c2, s23, c23, s234, c234;
  sincos(q[0], &s0, &c0); sincos(q[1], &s1, &c1); sincos(q[2], &s2, &c2);
  sincos(q[2] + q[3], &s23, &c23); sincos(q[2] + q[3] + q[4], &s234, &c234);
  const double a[3] = {0.04 * s234 + 0.22 * s23 + 0.22 * s2, 0.04 * s234 + 0.22 * s23, 0.04 * s234};
  const double b[3] = {0.04 * c234 + 0.22 * c23 + 0.22 * c2, 0.04 * c234 + 0.22 * c23, 0.04 * c234};
  const double e = 0.018 * side + 0.0025;
#pragma unroll
  for (int i = 0; i < 30; ++i) Jm[i] = 0.0;
  Jm[0 * 5 + 0] = s0 * (a[0] + 0.0135) + c0 * (0.015 * side + c1 * e - s1 * b[0]);
  Jm[1 * 5 + 0] = s0 * (0.015 * side + c1 * e - s1 * b[0]) - c0 * (a[0] + 0.0135);
  Jm[5 * 5 + 0] = 1.0;
  Jm[0 * 5 + 1] = -s0 * (s1 * e + c1 * b[0]);
  Jm[1 * 5 + 1] = c0 * (s1 * e + c1 * b[0]);
  Jm[2 * 5 + 1] = s1 * b[0] - c1 * e;
  Jm[3 * 5 + 1] = c0;
  Jm[4 * 5 + 1] = s0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int col = 2 + k;
    Jm[0 * 5 + col] = s0 * s1 * a[k] - c0 * b[k];
    Jm[1 * 5 + col] = -s0 * b[k] - c0 * s1 * a[k];
    Jm[2 * 5 + col] = c1 * a[k];
    Jm[3 * 5 + col] = -c1 * s0;
    Jm[4 * 5 + col] = c0 * c1;
    Jm[5 * 5 + col] = s1;
  }
}

// pf_w[B][6] = p_c + R' (pf_b + hip)   (REF:406-424)
__global__ void __launch_bounds__(256)
foot_world_kernel(const LowLevelParams P, const int B, const float* __restrict__ x_fb, const float* __restrict__ q,
                  float* __restrict__ pf_w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double e[3], R[9];
#pragma unroll
  for (int a = 0; a < 3; ++a) e[a] = x_fb[(size_t)i * 12 + a];
  eul2rotm_d(e, R);
#pragma unroll
  for (int leg = 0; leg < 2; ++leg) {
    const double side = leg == 0 ? 1.0 : -1.0;
    double ql[5], pf[3];
#pragma unroll
    for (int a = 0; a < 5; ++a) ql[a] = q[(size_t)i * 10 + 5 * leg + a];
    foot_body(ql, side, pf);
    const double v[3] = {pf[0] + P.hip_offset[0], pf[1] + side * P.hip_offset[1], pf[2] + P.hip_offset[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a)
      pf_w[(size_t)i * 6 + 3 * leg + a] =
          (float)((double)x_fb[(size_t)i * 12 + 3 + a] + R[0 * 3 + a] * v[0] + R[1 * 3 + a] * v[1] + R[2 * 3 + a] * v[2]);
  }
}

// tau[B][10]  (REF:444-470);  contact0[B][2] = contact[0, 0:2],  u0[B][12] = controls[0]
__global__ void __launch_bounds__(256)
lowlevel_kernel(const LowLevelParams P, const int B, const float* __restrict__ x_fb, const double* __restrict__ t,
                const float* __restrict__ pf_w, const float* __restrict__ q, const float* __restrict__ qd,
                const uint8_t* __restrict__ contact0, const float* __restrict__ u0, float* __restrict__ tau) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[12], R[9];
#pragma unroll
  for (int a = 0; a < 12; ++a) x[a] = x_fb[(size_t)i * 12 + a];
  eul2rotm_d(x, R);
  const double Ts = P.dt * P.h / 2;                                    // REF:436-437
  const double ts = fmod(t[i], Ts) < 0 ? fmod(t[i], Ts) + Ts : fmod(t[i], Ts);
  const double dx = x[3] + x[9] * 0.5 * P.h / 2 * P.dt + P.kv * (x[3] - P.x_cmd[3]);   // REF:428-431
  const double dy0 = x[4] + x[10] * 0.5 * P.h / 2 * P.dt + P.kv * (x[4] - P.x_cmd[4]);
  const double dz = P.swing_height * sin(3.14159265358979323846 * ts / Ts);
#pragma unroll
  for (int leg = 0; leg < 2; ++leg) {
    const double side = leg == 0 ? 1.0 : -1.0;
    double ql[5], qdl[5], Jm[30];
#pragma unroll
    for (int a = 0; a < 5; ++a) { ql[a] = q[(size_t)i * 10 + 5 * leg + a]; qdl[a] = qd[(size_t)i * 10 + 5 * leg + a]; }
    leg_jacobian(ql, side, Jm);
    double jq[3], vf[3];                                               // vf_w = R' Jf qd   (REF:461)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) s += Jm[a * 5 + k] * qdl[k];
      jq[a] = s;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) vf[a] = R[0 * 3 + a] * jq[0] + R[1 * 3 + a] * jq[1] + R[2 * 3 + a] * jq[2];
    const double des[3] = {dx, dy0 + 0.04 * side, dz};                 // REF:432-439
    double err[3], fs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) err[a] = des[a] - (double)pf_w[(size_t)i * 6 + 3 * leg + a];
#pragma unroll
    for (int a = 0; a < 3; ++a)                                        // REF:441
      fs[a] = P.kp[3 * a] * err[0] + P.kp[3 * a + 1] * err[1] + P.kp[3 * a + 2] * err[2] -
              (P.kd[3 * a] * vf[0] + P.kd[3 * a + 1] * vf[1] + P.kd[3 * a + 2] * vf[2]);
    double uw[6];                                                      // -[R' f; R' m]   (REF:465)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double sf = 0, sm2 = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sf += R[k * 3 + a] * (double)u0[(size_t)i * 12 + 3 * leg + k];
        sm2 += R[k * 3 + a] * (double)u0[(size_t)i * 12 + 3 * leg + 6 + k];
      }
      uw[a] = -sf; uw[3 + a] = -sm2;
    }
    const double c = (double)contact0[(size_t)i * 2 + leg];
#pragma unroll
    for (int k = 0; k < 5; ++k) {                                      // REF:466-468
      double st = 0, sw = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) st += Jm[a * 5 + k] * uw[a];
#pragma unroll
      for (int a = 0; a < 3; ++a) sw += Jm[a * 5 + k] * fs[a];
      tau[(size_t)i * 10 + 5 * leg + k] = (float)(st * c + sw * (1.0 - c));
    }
  }
}

// ---- gait scheduler (REF:50-59, generalised) ---------------------------------------------------------
// phase[b] = int(t[b] // dt) % h with CPython's float floor division (so that times that are multiples of dt
// up to rounding fall on the side the reference puts them), and rows k .. k+h-1 of a periodic contact
// schedule: leg g is in stance at step n iff ((n + offset[g]) mod period) < duty[g].  The reference's
// table (REF:52-55) is period 10, offset {0, 5}, duty {5, 5}.
struct GaitParams { int h, period, offset[2], duty[2]; double dt; };

__device__ __forceinline__ double py_floordiv(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0 && ((wx < 0) != (mod < 0))) div -= 1.0;
  if (div != 0.0) {
    double fl = floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return fl;
  }
  return copysign(0.0, vx / wx);
}

__global__ void __launch_bounds__(256)
gait_kernel(const GaitParams G, const int B, const double* __restrict__ t, int32_t* __restrict__ phase,
            uint8_t* __restrict__ contact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const double ph = py_floordiv(t[i], G.dt);                           // REF:56
  // int(ph) % h with Python's sign convention (non-negative for h > 0); |ph| < 2^53 keeps it exact
  double kd = fmod(ph, (double)G.h);
  if (kd < 0) kd += (double)G.h;
  const int k = (int)kd;                                                // REF:57
  if (phase) phase[i] = k;
  if (contact) {
    for (int n = 0; n < G.h; ++n)                                       // REF:58: rows k .. k+h-1
      for (int g = 0; g < 2; ++g) {
        int m = (k + n + G.offset[g]) % G.period;
        if (m < 0) m += G.period;
        contact[((size_t)i * G.h + n) * 2 + g] = m < G.duty[g] ? 1 : 0;
      }
  }
}

// One closed-loop step of the batched roll-out: the MPC's own prediction of the next
// state (row 0 of `states`, REF:301) becomes the state feedback, time advances by dt, and the applied control /
// the new state / the solver's iteration count are recorded.  One thread per instance.
__global__ void rollout_feedback_kernel(int B, int h, double dt, const float* __restrict__ states,
                                        const float* __restrict__ controls, const int32_t* __restrict__ iters,
                                        const int32_t* __restrict__ status, float* __restrict__ x_fb,
                                        double* __restrict__ t, float* __restrict__ u0_out, float* __restrict__ x_out,
                                        int32_t* __restrict__ iters_out, int32_t* __restrict__ status_any) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* s0 = states + (size_t)b * h * 13;
  const float* u0 = controls + (size_t)b * h * 12;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float v = s0[i];
    x_fb[(size_t)b * 12 + i] = v;
    if (x_out) x_out[(size_t)b * 12 + i] = v;
    if (u0_out) u0_out[(size_t)b * 12 + i] = u0[i];
  }
  t[b] += dt;
  if (iters_out) iters_out[b] = iters[b];
  if (status_any) status_any[b] |= status[b];
}

// Dispatch order for the next solve of a batch whose previous solve took iters[b] iterations: instances sorted by
// descending iteration count (counting sort over iteration buckets, ONE workgroup; the order inside a bucket is
// whatever the atomics give -- the results of a solve do not depend on the order, only its duration does).
constexpr int ORDER_BINS = 512;
__global__ void __launch_bounds__(1024) dispatch_order_kernel(int B, const int32_t* __restrict__ iters, int32_t* __restrict__ order) {
  __shared__ int bin[ORDER_BINS];
  for (int k = threadIdx.x; k < ORDER_BINS; k += blockDim.x) bin[k] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int key = min(max(iters[b], 0), ORDER_BINS - 1);
    atomicAdd(&bin[key], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {                       // start of every bucket, longest first (512 additions: microseconds)
    int acc = 0;
    for (int k = ORDER_BINS - 1; k >= 0; --k) { const int n = bin[k]; bin[k] = acc; acc += n; }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int key = min(max(iters[b], 0), ORDER_BINS - 1);
    order[atomicAdd(&bin[key], 1)] = b;
  }
}

}  // namespace bmpc
