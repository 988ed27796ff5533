// micro-benchmark: Cholesky factor + triangular inverse of ONE 32 x 32 SPD block by one workgroup -- the serial core of
// every k_chol_panel launch.  Variants: A = one wave, rows in registers, v_readlane broadcasts; B = one wave, rows in
// registers, column through LDS; C = 256 threads, everything in LDS with workgroup barriers; D = one wave, everything in
// LDS, runtime loops.   hipcc --offload-arch=gfx950 -O3 diag32.hip -o diag32 && ./diag32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ void inverse_lds(double* Dm, double* Li, double* dinv, int lane) {   // one wave; Dm = L (lower), Li out
  const int row = lane & 31;
  if (lane < 32) {
    for (int r = 0; r < 32; ++r) Li[r * 33 + row] = (r == row) ? dinv[r] : 0.0;
    for (int r = 1; r < 32; ++r) {
      double s0 = 0.0, s1 = 0.0;
      int k = 0;
      for (; k + 1 < r; k += 2) {
        s0 += Dm[r * 33 + k] * Li[k * 33 + row];
        s1 += Dm[r * 33 + k + 1] * Li[(k + 1) * 33 + row];
      }
      if (k < r) s0 += Dm[r * 33 + k] * Li[k * 33 + row];
      if (r > row) Li[r * 33 + row] = -(s0 + s1) * dinv[r];
    }
  }
}

__device__ __forceinline__ double fast_rcp(double d) {   // v_rcp_f64 + two Newton steps: the short form of 1.0 / d (normal range)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// variant 3: one wave; L D L' with rows in registers (no sqrt / divide in the 32-step chain: one reciprocal per step),
// pivots and columns broadcast by v_readlane; then L = Lt sqrt(D) and N = sqrt(D)^-1 Lt^-1 with the unit-triangular inverse
// accumulated from broadcast reads issued ahead of the dependent sums
__device__ void diag_v3(double* Dm, double* Li, int lane) {
  const int row = lane & 31;
  double a[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) a[c] = Dm[row * 33 + c];
  double dsel = 0.0;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double d = readlane_f64(a[k], k);      // pivot D_k
    const double rd = fast_rcp(d);
    dsel = (row == k) ? d : dsel;
    const double ak = a[k];                       // a_rk (unscaled) = Lt_rk D_k
    a[k] = ak * rd;                               // Lt_rk
#pragma unroll
    for (int c = k + 1; c < 32; ++c) a[c] -= a[k] * readlane_f64(ak, c);   // a_rc -= Lt_rk (Lt_ck D_k)
  }
  const double sd = sqrt(dsel), isd = 1.0 / sd;
  // unit lower triangle -> LDS for the broadcast reads of the inverse
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) Dm[row * 33 + c] = (c < row) ? a[c] : 0.0;
  }
  wave_lds_sync();
  // column `row` of Lt^-1 in registers: n[r] = -(sum_{k=row}^{r-1} Lt[r][k] n[k]),  n[row] = 1
  double n[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) {
      const double t = Dm[r * 33 + k] * n[k];   // (n[k] = 0 for k < row)
      if (k & 1) s1 += t;
      else s0 += t;
    }
    n[r] = (r == row) ? 1.0 : ((r > row) ? -(s0 + s1) : 0.0);
  }
  wave_lds_sync();
  // L = Lt sqrt(D): column c scaled by sqrt(D_c); N = sqrt(D)^-1 Lt^-1: row r scaled by 1 / sqrt(D_r)
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const double sc = readlane_f64(sd, c), ic = readlane_f64(isd, c);
      Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      Li[c * 33 + row] = n[c] * ic;
    }
  }
}

// variant 4: as variant 3, but the unit-triangular inverse keeps its column in LDS (runtime loops, 8 products in flight)
template <int MODE>
__device__ void diag_v4(double* Dm, double* Li, int lane) {
  const int row = lane & 31;
  double a[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) a[c] = Dm[row * 33 + c];
  double dsel = 1.0;
  if (MODE != 2)
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double d = readlane_f64(a[k], k);
    const double rd = fast_rcp(d);
    dsel = (row == k) ? d : dsel;
    const double ak = a[k];
    a[k] = ak * rd;
#pragma unroll
    for (int c = k + 1; c < 32; ++c) a[c] -= a[k] * readlane_f64(ak, c);
  }
  const double sd = sqrt(dsel), isd = 1.0 / sd;
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      Dm[row * 33 + c] = (c < row) ? a[c] : 0.0;       // unit lower triangle (strict part)
      Li[c * 33 + row] = (c == row) ? 1.0 : 0.0;       // column `row` of the inverse, in this lane's LDS column
    }
  }
  wave_lds_sync();
  if (MODE != 1 && lane < 32) {
    for (int r = 1; r < 32; ++r) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
#pragma unroll 2
      for (; k + 3 < r; k += 4) {
        s0 += Dm[r * 33 + k] * Li[k * 33 + row];
        s1 += Dm[r * 33 + k + 1] * Li[(k + 1) * 33 + row];
        s2 += Dm[r * 33 + k + 2] * Li[(k + 2) * 33 + row];
        s3 += Dm[r * 33 + k + 3] * Li[(k + 3) * 33 + row];
      }
      for (; k < r; ++k) s0 += Dm[r * 33 + k] * Li[k * 33 + row];
      if (r > row) Li[r * 33 + row] = -((s0 + s1) + (s2 + s3));
    }
  }
  wave_lds_sync();
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const double sc = readlane_f64(sd, c), ic = readlane_f64(isd, c);
      Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      Li[c * 33 + row] *= ic;
    }
  }
}

// variant 7: L D L' and the inverse of the unit triangle in ONE 32-step loop, rows of both in registers (lane r: row r of
// A and row r of N), v_readlane broadcasts only: step k scales column k, updates the trailing row entries
// a_rc -= Lt_rk (Lt_ck D_k) for c > k, and eliminates column k from the inverse N_r. -= Lt_rk N_k. (row k of N is final
// by then); no LDS access and no sqrt / divide in the chain
__device__ void diag_v7(double* Dm, double* Li, int lane) {
  const int row = lane & 31;
  double a[32], n[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    a[c] = Dm[row * 33 + c];
    n[c] = (c == row) ? 1.0 : 0.0;
  }
  double dsel = 1.0;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double d = readlane_f64(a[k], k);
    const double rd = fast_rcp(d);
    dsel = (row == k) ? d : dsel;
    const double ak = a[k];
    const double l = (row > k) ? ak * rd : 0.0;   // Lt_rk (0 on and above the diagonal: those rows are finished)
    a[k] = l;
#pragma unroll
    for (int c = k + 1; c < 32; ++c) a[c] -= l * readlane_f64(ak, c);
#pragma unroll
    for (int c = 0; c <= k; ++c) n[c] -= l * readlane_f64(n[c], k);
  }
  const double sd = sqrt(dsel), isd = 1.0 / sd;
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const double sc = readlane_f64(sd, c);
      Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      Li[row * 33 + c] = n[c] * isd;
    }
  }
}

__device__ __forceinline__ double bperm_f64(double v, int src_lane) {   // LDS-crossbar broadcast (no SGPR round trip)
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// variant 8: variant 7 with ds_bpermute broadcasts
__device__ void diag_v8(double* Dm, double* Li, int lane) {
  const int row = lane & 31;
  double a[32], n[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    a[c] = Dm[row * 33 + c];
    n[c] = (c == row) ? 1.0 : 0.0;
  }
  double dsel = 1.0;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double d = bperm_f64(a[k], k);
    const double rd = fast_rcp(d);
    dsel = (row == k) ? d : dsel;
    const double ak = a[k];
    const double l = (row > k) ? ak * rd : 0.0;
    a[k] = l;
#pragma unroll
    for (int c = k + 1; c < 32; ++c) a[c] -= l * bperm_f64(ak, c);
#pragma unroll
    for (int c = 0; c <= k; ++c) n[c] -= l * bperm_f64(n[c], k);
  }
  const double sd = sqrt(dsel), isd = 1.0 / sd;
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const double sc = bperm_f64(sd, c);
      Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      Li[row * 33 + c] = n[c] * isd;
    }
  }
}

// variant 9: TWO wavefronts.  Wave 0 factorises (L D L', rows in registers, v_readlane) and hands every finished column of
// Lt to wave 1 through LDS; wave 1 eliminates that column from the inverse one step behind.  The two 32-step loops of
// variant 7 run side by side instead of one after the other inside each step.
__device__ void diag_v9(double* Dm, double* Li, double* colL, volatile int* flag, int tid) {
  const int w = tid >> 6, lane = tid & 63, row = lane & 31;
  if (w == 0) {
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = Dm[row * 33 + c];
    double dsel = 1.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const double d = readlane_f64(a[k], k);
      const double rd = fast_rcp(d);
      dsel = (row == k) ? d : dsel;
      const double ak = a[k];
      const double l = (row > k) ? ak * rd : 0.0;
      a[k] = l;
      if (lane < 32) colL[k * 32 + row] = l;
      wave_lds_sync();
      if (lane == 0) __hip_atomic_store((int*)flag, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of one wavefront complete in order)
#pragma unroll
      for (int c = k + 1; c < 32; ++c) a[c] -= l * readlane_f64(ak, c);
    }
    const double sd = sqrt(dsel), isd = 1.0 / sd;
    if (lane < 32) {
      colL[32 * 32 + row] = isd;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double sc = readlane_f64(sd, c);
        Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      }
    }
    wave_lds_sync();
    if (lane == 0) __hip_atomic_store((int*)flag, 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if (w == 1) {
    double n[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) n[c] = (c == row) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      int spins = 0;
      while (__hip_atomic_load((int*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k && ++spins < (1 << 20)) {}
      wave_lds_sync();
      const double l = colL[k * 32 + row];
#pragma unroll
      for (int c = 0; c <= k; ++c) n[c] -= l * readlane_f64(n[c], k);
    }
    int spins = 0;
    while (__hip_atomic_load((int*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 33 && ++spins < (1 << 20)) {}
    wave_lds_sync();
    const double isd = colL[32 * 32 + row];
    if (lane < 32) {
#pragma unroll
      for (int c = 0; c < 32; ++c) Li[row * 33 + c] = n[c] * isd;
    }
  }
}
template <int V>
__global__ __launch_bounds__(V == 2 ? 256 : (V == 9 ? 128 : 64)) void k_diag(const double* in, double* outL, double* outI, int reps) {
  __shared__ double Dm[32 * 33], Li[32 * 33], dinv[32], colb[64], colL[33 * 32];
  __shared__ int flag9;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int rep = 0; rep < reps; ++rep) {
    for (int e = tid; e < 1024; e += blockDim.x) Dm[(e >> 5) * 33 + (e & 31)] = in[e];
    if (tid == 0) flag9 = 0;
    __syncthreads();
    if (V == 0) {
      const int row = lane & 31;
      double a[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) a[c] = Dm[row * 33 + c];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const double d = sqrt(readlane_f64(a[k], k));
        const double rs = 1.0 / d;
        a[k] = (row == k) ? d : a[k] * rs;
#pragma unroll
        for (int c = k + 1; c < 32; ++c) a[c] -= a[k] * readlane_f64(a[k], c);
      }
      if (lane < 32) {
        double dsel = 0.0;
#pragma unroll
        for (int c = 0; c < 32; ++c) { Dm[row * 33 + c] = (c <= row) ? a[c] : 0.0; dsel = (row == c) ? a[c] : dsel; }
        dinv[row] = 1.0 / dsel;
      }
      wave_lds_sync();
      inverse_lds(Dm, Li, dinv, lane);
    } else if (V == 3) {
      diag_v3(Dm, Li, lane);
    } else if (V == 4) {
      diag_v4<0>(Dm, Li, lane);
    } else if (V == 9) {
      diag_v9(Dm, Li, colL, &flag9, tid);
    } else if (V == 7) {
      diag_v7(Dm, Li, lane);
    } else if (V == 8) {
      diag_v8(Dm, Li, lane);
    } else if (V == 5) {
      diag_v4<1>(Dm, Li, lane);
    } else if (V == 6) {
      diag_v4<2>(Dm, Li, lane);
    } else if (V == 1) {   // everything in LDS, one wave, runtime loops: lane r owns row r
      const int row = lane & 31;
      for (int k = 0; k < 32; ++k) {
        const double d = sqrt(Dm[k * 33 + k]);
        const double rs = 1.0 / d;
        wave_lds_sync();
        if (lane < 32) {
          if (row == k) Dm[k * 33 + k] = d;
          else if (row > k) Dm[row * 33 + k] *= rs;
        }
        wave_lds_sync();
        if (lane < 32 && row > k) {
          const double lr = Dm[row * 33 + k];
          for (int c = k + 1; c <= row; ++c) Dm[row * 33 + c] -= lr * Dm[c * 33 + k];
        }
        wave_lds_sync();
      }
      if (lane < 32) dinv[row] = 1.0 / Dm[row * 33 + row];
      wave_lds_sync();
      inverse_lds(Dm, Li, dinv, lane);
    } else {   // 256 threads, barriers
      const int tx = tid & 31, tyb = tid >> 5;
      for (int k = 0; k < 32; ++k) {
        if (tid < 32 && tid > k) Dm[tid * 33 + k] = Dm[tid * 33 + k] / sqrt(Dm[k * 33 + k]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ty = tyb + 8 * u;
          if (tx > k && ty >= tx) Dm[ty * 33 + tx] -= Dm[ty * 33 + k] * Dm[tx * 33 + k];
        }
        __syncthreads();
      }
      if (tid < 32) { const double d = sqrt(Dm[tid * 33 + tid]); Dm[tid * 33 + tid] = d; dinv[tid] = 1.0 / d; }
      __syncthreads();
      const int c = tid >> 3, p = tid & 7;
      for (int e = tid; e < 1024; e += 256) Li[(e >> 5) * 33 + (e & 31)] = 0.0;
      __syncthreads();
      if (p == 0) Li[c * 33 + c] = dinv[c];
      wave_lds_sync();
      for (int r = 1; r < 32; ++r) {
        double sum = 0.0;
        for (int k = c + p; k < r; k += 8) sum += Dm[r * 33 + k] * Li[k * 33 + c];
        sum += __shfl_xor(sum, 1, 8); sum += __shfl_xor(sum, 2, 8); sum += __shfl_xor(sum, 4, 8);
        if (p == 0 && r > c) Li[r * 33 + c] = -sum * dinv[r];
        wave_lds_sync();
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < 1024; e += blockDim.x) {
    const int r = e >> 5, c = e & 31;
    outL[e] = c <= r ? Dm[r * 33 + c] : 0.0;
    outI[e] = Li[r * 33 + c];
  }
}
int main() {
  std::vector<double> A(1024), L(1024), I(1024);
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) A[r * 32 + c] = (r == c ? 40.0 : 0.0) + std::sin(0.37 * (r + 1) * (c + 1)) + std::sin(0.37 * (c + 1) * (r + 1));
  double *dA, *dL, *dI;
  hipMalloc(&dA, 8192); hipMalloc(&dL, 8192); hipMalloc(&dI, 8192);
  hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  for (int v = 7; v < 10; v += 2) {
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k_diag<0>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 1) hipLaunchKernelGGL(k_diag<1>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 2) hipLaunchKernelGGL(k_diag<2>, dim3(1), dim3(256), 0, 0, dA, dL, dI, reps);
      if (v == 3) hipLaunchKernelGGL(k_diag<3>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 4) hipLaunchKernelGGL(k_diag<4>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 5) hipLaunchKernelGGL(k_diag<5>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 6) hipLaunchKernelGGL(k_diag<6>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 7) hipLaunchKernelGGL(k_diag<7>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 8) hipLaunchKernelGGL(k_diag<8>, dim3(1), dim3(64), 0, 0, dA, dL, dI, reps);
      if (v == 9) hipLaunchKernelGGL(k_diag<9>, dim3(1), dim3(128), 0, 0, dA, dL, dI, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(L.data(), dL, 8192, hipMemcpyDeviceToHost); hipMemcpy(I.data(), dI, 8192, hipMemcpyDeviceToHost);
    double e_f = 0, e_i = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
      double s = 0, t = 0;
      for (int k = 0; k < 32; ++k) { s += L[r * 32 + k] * L[c * 32 + k]; t += L[r * 32 + k] * I[k * 32 + c]; }
      e_f = std::fmax(e_f, std::fabs(s - A[r * 32 + c])); e_i = std::fmax(e_i, std::fabs(t - (r == c)));
    }
    printf("variant %d: %.2f us per factor+inverse   |LL'-A| %.1e  |L Linv - I| %.1e\n", v, 1e3 * ms / reps, e_f, e_i);
  }
  return 0;
}
