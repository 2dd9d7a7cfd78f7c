// Leaf of the blocked Cholesky: factor one 128x128 diagonal tile in LDS and invert
// the triangular factor (the inverse turns every panel solve into an MFMA GEMM).
// Replaces the unblocked interior of LAPACK potrf behind np.linalg.cholesky
// (GPr.py:62, GPpref.py:128).
//
// One workgroup of 1024 threads, the whole tile resident in LDS (128 x 129 doubles =
// 129 KiB of the CU's 160 KiB; the odd row stride makes column walks conflict-free).
//  1. square-root-free right-looking elimination, ONE barrier per column:
//        S[i][k] -= S[i][j] * S[k][j] / S[j][j]      (j < k <= i)
//     column j and the pivot are never written again after step j, so the scaling
//     L[i][j] = S[i][j]/sqrt(S[j][j]) is applied once at the end.
//  2. in-place triangular inverse by recursive doubling (block sizes 1,2,..,64):
//        X21 = -X22 * (L21 * X11), intermediate product held in registers.
#include "gpk_internal.h"

#define PLD 129

__global__ __launch_bounds__(1024) void potf2_trtri_128_kernel(const double* __restrict__ A, long lda,
                                                               double* __restrict__ L, long ldl,
                                                               double* __restrict__ Dinv, int* info, int col0) {
  __shared__ double S[128 * PLD];
  const int tid = threadIdx.x;
  for (int e = tid; e < 128 * 128; e += 1024) {
    const int i = e >> 7, j = e & 127;
    S[i * PLD + j] = A[(long)i * lda + j];
  }
  __syncthreads();

  const int i = tid & 127, kg = tid >> 7;
  int fail = 0;
  for (int j = 0; j < 128; ++j) {
    const double ajj = S[j * PLD + j];
    if (!(ajj > 0.0) || !(ajj < 1.0e300)) {  // also catches NaN / Inf; wave- and block-uniform
      fail = j + 1;
      break;
    }
    const double r = 1.0 / ajj;
    if (i > j) {
      const double lij = S[i * PLD + j];
      for (int k = j + 1 + kg; k <= i; k += 8) S[i * PLD + k] -= lij * (S[k * PLD + j] * r);
    }
    __syncthreads();
  }
  __syncthreads();
  if (fail) {
    if (tid == 0) atomicCAS(info, 0, col0 + fail);
    // keep downstream kernels on finite data: emit identity factors for this tile
    for (int e = tid; e < 128 * 128; e += 1024) {
      const int ii = e >> 7, jj = e & 127;
      const double v = (ii == jj) ? 1.0 : 0.0;
      L[(long)ii * ldl + jj] = v;
      Dinv[ii * 128 + jj] = v;
    }
    return;
  }

  // ---- scale columns: L = S * diag(1/sqrt(pivot))
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = tid + 1024 * q;
    const int ii = e >> 7, jj = e & 127;
    double x = 0.0;
    if (jj <= ii) {
      const double dj = sqrt(S[jj * PLD + jj]);
      x = (ii == jj) ? dj : S[ii * PLD + jj] / dj;
    }
    v[q] = x;
    L[(long)ii * ldl + jj] = x;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = tid + 1024 * q;
    S[(e >> 7) * PLD + (e & 127)] = v[q];
  }
  __syncthreads();

  // ---- in-place inverse of the lower-triangular factor
  if (tid < 128) S[tid * PLD + tid] = 1.0 / S[tid * PLD + tid];
  __syncthreads();
  for (int lb = 0; lb < 7; ++lb) {  // b = 1 << lb
    const int b = 1 << lb;
    const int nel = 64 * b;  // (64/b pairs) * b*b elements
    double t[4];
    int rr[4], cc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 1024 * q;
      t[q] = 0.0;
      rr[q] = -1;
      cc[q] = 0;
      if (e < nel) {
        const int pair = e >> (2 * lb);
        const int rem = e & (b * b - 1);
        const int ii = rem >> lb, c = rem & (b - 1);
        const int r0 = 2 * b * pair;
        const int row = r0 + b + ii;
        double sum = 0.0;
        for (int k = c; k < b; ++k) sum += S[row * PLD + r0 + k] * S[(r0 + k) * PLD + r0 + c];
        t[q] = sum;
        rr[q] = row;
        cc[q] = r0 + c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rr[q] >= 0) S[rr[q] * PLD + cc[q]] = t[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rr[q] >= 0) {
        const int row = rr[q];
        const int rb = row & ~(2 * b - 1);  // r0
        const int ii = row - rb - b;
        double sum = 0.0;
        for (int k = 0; k <= ii; ++k) sum += S[row * PLD + rb + b + k] * S[(rb + b + k) * PLD + cc[q]];
        t[q] = -sum;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rr[q] >= 0) S[rr[q] * PLD + cc[q]] = t[q];
    __syncthreads();
  }
  for (int e = tid; e < 128 * 128; e += 1024) {
    const int ii = e >> 7, jj = e & 127;
    Dinv[e] = (jj <= ii) ? S[ii * PLD + jj] : 0.0;
  }
}

int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0,
                           int /*n_valid*/, hipStream_t s) {
  hipLaunchKernelGGL(potf2_trtri_128_kernel, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0);
  GPK_HIP(hipGetLastError());
  return 0;
}
