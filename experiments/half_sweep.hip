// Standalone experiment (round 6): the symmetric HALF-ROW sweep VERDICT r3 / r4 / r5 asked for, in the form that can pay --
// ONE wave per matrix, one lane per row, each lane holding only the forward half of its row.
//
// The shipped sweep (bmpc_kernels.hip, factor()): two lanes per row, each holding a column half (3 H floats), two waves at
// h = 10; per two-pivot step and WAVE 30 v_pk_fma_f32 + 14 ds_read_b128 + ~45 other instructions, i.e. ~190 wave-instructions
// per step and matrix.  It is bound by the instructions issued (docs/history_r05.md), and a symmetric matrix holds every
// off-diagonal entry twice.  Keeping the two-lanes-per-row map and halving each lane's window saves 12 of the 30 FMAs per
// wave and nothing of the per-step overhead both waves repeat.  This variant halves the WAVES instead:
//
//   * the matrix is padded to 64 x 64 (rows / columns n .. 63: identity); lane r of ONE wave owns row r and stores the 36
//     columns [r4, r4 + 35] (mod 64), r4 = 4 floor(r / 4): 32 registers w[i], i = column mod 32, and 4 registers x[] for the
//     antipodal quad r4 + 32 .. r4 + 35.  Every unordered pair {r, c} lies in the window of r or of c (in both for the own
//     and the antipodal quad): the whole symmetric matrix, 36 instead of 60 floats per row.
//   * column k sits in register k mod 32 of EVERY row that holds it (or in x[k mod 4] of the antipodal quad's rows), so with
//     the sweep unrolled over 32 consecutive pivots all register indices are static -- no rotation of the register file.
//   * per step (two pivots k, k + 1) the full pivot rows V[k, :], V[k + 1, :] are assembled in LDS: the rows that hold column
//     k write their entry V[c][k] to R[c] (one ds_write_b32 per pivot, a dump slot for the others), the pivot lane writes its
//     own window (9 ds_write_b128 under an exec mask, to the very addresses it READS pivot rows from: register quad q of lane
//     r always faces columns 4 q + 32 s_q(r)).  Then every lane reads the two rows at its 9 quads (18 ds_read_b128 at
//     lane-dependent addresses: two distinct addresses per instruction), forms the 2 x 2 pivot inverse and its multipliers
//     exactly as the shipped sweep does, and updates 36 entries with 36 v_pk_fma_f32.
//   * one wave: no s_barrier; a wave's LDS operations execute in order, the hand-over needs a wait and a compiler fence.
//
// Measured on MI355X: see docs/history_r06.md ("the half-row sweep").  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w experiments/half_sweep.hip -o /tmp/hs && /tmp/hs
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_s_waitcnt(0xc07f);                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

constexpr int NP = 64;              // padded size
constexpr int RS = NP + 64;         // one published row: 64 entries + 64 dump slots

// VAR 0: the sweep; VAR 1: timing variant without the pivot lanes' row publication (wrong results: what the 18 wide stores cost);
// VAR 2: the sweep with the pivot lanes storing only the seven quads nobody else holds (own and antipodal quad come from the
// column entries of the other rows): 14 wide stores per step instead of 18
template <int VAR>
__global__ void __launch_bounds__(128, 2) half_sweep(const float* __restrict__ Ain, float* __restrict__ Vout, long long* cyc,
                                                      int n, int reps, int pad_words) {
  __shared__ __attribute__((aligned(16))) float R[2][2][RS];          // [buffer][pivot of the step][entry]
  extern __shared__ float pad_lds[];                                  // (occupancy: the solver's instance holds 37.6 KB)
  if (pad_words < 0) pad_lds[threadIdx.x] = 0.f;
  const int wave = threadIdx.x >> 6;
  const float* A = Ain + (size_t)blockIdx.x * n * n;
  long long t_total = 0;
  if (wave == 0) {
    const int r = threadIdx.x & 63;
    const int r4 = r & ~3;
    const int b = r4 & 31, hi = r4 >> 5;
    // byte address of the entries register quad q faces in a published row: columns 4 q + 32 s_q
    int aq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) aq[q] = 4 * (4 * q + 32 * (hi ^ (4 * q < b ? 1 : 0)));
    const int ax = 4 * ((r4 + 32) & 63);
    for (int rep = 0; rep < reps; ++rep) {
      float w[32], x[4];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int c = i + 32 * (hi ^ (i < b ? 1 : 0));
        w[i] = (r < n && c < n) ? A[r * n + c] : (r == c ? 1.f : 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (r4 + 32 + q) & 63;
        x[q] = (r < n && c < n) ? A[r * n + c] : (r == c ? 1.f : 0.f);
      }
      const long long t0c = clock64();
      char* Rb = reinterpret_cast<char*>(&R[0][0][0]);
      // publication of the pivots (k, k + 1): column entries by the rows that hold column k, whole windows by the pivot lanes
      auto publish = [&](const int k, const int kk, const int buf) {
        const int dq = ((k & ~3) - r4) & 63;               // distance of the pivots' quad from the own window's start
        const bool inW = dq < 32, inX = dq == 32;
        char* RA = Rb + buf * (2 * RS * 4);
        char* RBp = RA + RS * 4;
        const float va = inX ? x[kk & 3] : w[kk], vb = inX ? x[(kk + 1) & 3] : w[kk + 1];
        const int ad = (inW || inX) ? 4 * r : 4 * (NP + r);
        *reinterpret_cast<float*>(RA + ad) = va;
        *reinterpret_cast<float*>(RBp + ad) = vb;
        if (VAR != 1) {
          if (r == k) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (VAR == 0 || q != (kk >> 2)) *reinterpret_cast<float4*>(RA + aq[q]) = float4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            if (VAR == 0) *reinterpret_cast<float4*>(RA + ax) = float4{x[0], x[1], x[2], x[3]};
          }
          if (r == k + 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (VAR == 0 || q != (kk >> 2)) *reinterpret_cast<float4*>(RBp + aq[q]) = float4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            if (VAR == 0) *reinterpret_cast<float4*>(RBp + ax) = float4{x[0], x[1], x[2], x[3]};
          }
        }
      };
      publish(0, 0, 0);
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
          const int k = 32 * pass + kk;
          if (k < n) {                                      // (uniform)
            const int par = (kk >> 1) & 1;                    // (static: sixteen steps per pass, the buffers alternate)
            const char* RA = Rb + par * (2 * RS * 4);
            const char* RBp = RA + RS * 4;
            WAVE_SYNC();
            const float2 pk = *reinterpret_cast<const float2*>(RA + 4 * k);       // V[k][k], V[k][k + 1]
            const float p11 = *reinterpret_cast<const float*>(RBp + 4 * (k + 1));
            const float c0 = *reinterpret_cast<const float*>(RA + 4 * r), c1 = *reinterpret_cast<const float*>(RBp + 4 * r);
            f2 pa[18], pb[18];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float4 a4 = *reinterpret_cast<const float4*>(RA + aq[q]);
              const float4 b4 = *reinterpret_cast<const float4*>(RBp + aq[q]);
              pa[2 * q] = f2{a4.x, a4.y}; pa[2 * q + 1] = f2{a4.z, a4.w};
              pb[2 * q] = f2{b4.x, b4.y}; pb[2 * q + 1] = f2{b4.z, b4.w};
            }
            {
              const float4 a4 = *reinterpret_cast<const float4*>(RA + ax);
              const float4 b4 = *reinterpret_cast<const float4*>(RBp + ax);
              pa[16] = f2{a4.x, a4.y}; pa[17] = f2{a4.z, a4.w};
              pb[16] = f2{b4.x, b4.y}; pb[17] = f2{b4.z, b4.w};
            }
            __builtin_amdgcn_sched_barrier(0);
            const float id = rcp_approx(pk.x * p11 - pk.y * pk.y);
            const float q00 = p11 * id, q01 = -pk.y * id, q11 = pk.x * id;      // P^-1
            const bool is0 = (r == k), is1 = (r == k + 1);
            float t0 = c0 * q00 + c1 * q01, t1 = c0 * q01 + c1 * q11;
            t0 = is0 ? 1.f - q00 : (is1 ? -q01 : t0);
            t1 = is0 ? -q01 : (is1 ? 1.f - q11 : t1);
            const f2 m0 = {-t0, -t0}, m1 = {-t1, -t1};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              f2 v = {w[2 * i], w[2 * i + 1]};
              v = __builtin_elementwise_fma(m1, pb[i], __builtin_elementwise_fma(m0, pa[i], v));
              w[2 * i] = v.x; w[2 * i + 1] = v.y;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              f2 v = {x[2 * i], x[2 * i + 1]};
              v = __builtin_elementwise_fma(m1, pb[16 + i], __builtin_elementwise_fma(m0, pa[16 + i], v));
              x[2 * i] = v.x; x[2 * i + 1] = v.y;
            }
            // the entries in the pivot columns: T = V[r, S] P^-1, and -P^-1 in the pivot block
            {
              const int dq = ((k & ~3) - r4) & 63;
              const bool inW = dq < 32, inX = dq == 32;
              const float s0 = is0 ? -q00 : (is1 ? -q01 : t0), s1 = is0 ? -q01 : (is1 ? -q11 : t1);
              w[kk] = inW ? s0 : w[kk];
              w[kk + 1] = inW ? s1 : w[kk + 1];
              x[kk & 3] = inX ? s0 : x[kk & 3];
              x[(kk + 1) & 3] = inX ? s1 : x[(kk + 1) & 3];
            }
            if (k + 2 < n) {
              // (the next pivots' register index: (kk + 2) mod 32 -- static; at kk = 30 the next pass starts at register 0)
              if (kk + 2 < 32) publish(k + 2, kk + 2, par ^ 1); else publish(k + 2, 0, par ^ 1);
            }
          }
        }
      }
      t_total += clock64() - t0c;
      if (rep == reps - 1 && Vout) {
        float* V = Vout + (size_t)blockIdx.x * n * n;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int c = i + 32 * (hi ^ (i < b ? 1 : 0));
          if (r < n && c < n) { V[r * n + c] = w[i]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = (r4 + 32 + q) & 63;
          if (r < n && c < n) V[r * n + c] = x[q];
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0 && (threadIdx.x & 63) == 0 && cyc) cyc[blockIdx.x] = t_total / reps;
}

static void invert_ref(const std::vector<double>& A, int n, std::vector<double>& Vi) {
  std::vector<double> M(A);
  Vi.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Vi[(size_t)i * n + i] = 1.0;
  for (int k = 0; k < n; ++k) {
    const double p = M[(size_t)k * n + k];
    for (int c = 0; c < n; ++c) { M[(size_t)k * n + c] /= p; Vi[(size_t)k * n + c] /= p; }
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const double f = M[(size_t)r * n + k];
      for (int c = 0; c < n; ++c) { M[(size_t)r * n + c] -= f * M[(size_t)k * n + c]; Vi[(size_t)r * n + c] -= f * Vi[(size_t)k * n + c]; }
    }
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 60;
  const int nmat = argc > 2 ? atoi(argv[2]) : 4096;
  const int reps = argc > 3 ? atoi(argv[3]) : 6;
  // SPD test matrices with unit diagonal (Jacobi-scaled like the solver's K'): G G' / m + ridge, scaled
  std::vector<float> hA((size_t)nmat * n * n);
  std::vector<double> A0((size_t)n * n);
  srand(7);
  for (int m = 0; m < nmat; ++m) {
    std::vector<double> G((size_t)n * n), K((size_t)n * n, 0.0);
    for (auto& g : G) g = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int q = 0; q < n; ++q) s += G[(size_t)i * n + q] * G[(size_t)j * n + q];
        K[(size_t)i * n + j] = s / n + (i == j ? 0.05 : 0.0);
      }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        const double v = K[(size_t)i * n + j] / std::sqrt(K[(size_t)i * n + i] * K[(size_t)j * n + j]);
        hA[((size_t)m * n + i) * n + j] = (float)v;
        if (m == 0) A0[(size_t)i * n + j] = (double)(float)v;
      }
  }
  float *dA, *dV;
  long long* dC;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dV, hA.size() * 4); hipMalloc(&dC, nmat * 8);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  // 36 KB + 2 KB static = the solver's 37.6 KB per instance: 4 matrices per CU, the second wave of the workgroup idle (as the
  // solver's second wave would be); 17 KB and one wave per workgroup: 8 matrices per CU, two sweeping waves per SIMD
  for (int run = 0; run < 6; ++run) {
    const int var = run % 3 == 0 ? 0 : (run % 3 == 1 ? 2 : 1);
    const int per_cu = run < 3 ? 4 : 8;
    const int pad_bytes = per_cu == 4 ? 36 * 1024 : 17 * 1024, nthr = per_cu == 4 ? 128 : 64;
    auto kern = var == 0 ? half_sweep<0> : (var == 1 ? half_sweep<1> : half_sweep<2>);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pad_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nmat), dim3(nthr), pad_bytes, 0, dA, dV, dC, n, 1, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nmat), dim3(nthr), pad_bytes, 0, dA, dV, dC, n, reps, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(nmat);
    hipMemcpy(hc.data(), dC, nmat * 8, hipMemcpyDeviceToHost);
    double cm = 0;
    for (auto c : hc) cm += (double)c;
    cm /= nmat;
    printf("n = %d, %d matrices, %d per CU, variant %d (%s): %.0f cycles per sweep (all matrices of a CU sweeping at once), %.1f us per %d sweeps\n",
           n, nmat, per_cu, var, var == 0 ? "half-row sweep, one wave per matrix, 18 window stores per step" : (var == 2 ? "the same, 14 window stores per step" : "timing only: without the pivot lanes' window stores"), cm, 1e3 * ms / reps, nmat);
    if (var != 1) {
      std::vector<float> hV((size_t)n * n);
      hipMemcpy(hV.data(), dV, (size_t)n * n * 4, hipMemcpyDeviceToHost);
      std::vector<double> Vi;
      invert_ref(A0, n, Vi);
      // the sweep leaves -A^-1 in the stored half: compare what each row stores
      double emax = 0, vmax = 0;
      int cnt = 0;
      for (int r = 0; r < n; ++r) {
        const int r4 = r & ~3;
        for (int d = 0; d < 36; ++d) {
          const int c = (r4 + d) & 63;
          if (c >= n) continue;
          const double e = std::fabs((double)hV[(size_t)r * n + c] + Vi[(size_t)r * n + c]);
          emax = e > emax ? e : emax;
          vmax = std::fabs(Vi[(size_t)r * n + c]) > vmax ? std::fabs(Vi[(size_t)r * n + c]) : vmax;
          ++cnt;
        }
      }
      printf("  matrix 0: max |V + A^-1| over the %d stored entries %.3e (max |A^-1| %.3e)\n", cnt, emax, vmax);
    }
  }
  return 0;
}
