// Round-1 hardware/library probe for MI355X (gfx950): FP64 GEMM/TRSM rate (rocBLAS), batched 3-D
// real FFT throughput (hipFFT), HBM copy rate, and the f64 MFMA lane map / k-order.
// Build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe -lrocblas -lhipfft
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void mfma_f64_probe(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
__global__ void copy_k(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) b[i] = a[i];
}
static float elapsed(hipEvent_t a, hipEvent_t b){ float ms; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms,a,b)); return ms; }

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d mem %.1f GB clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.totalGlobalMem/1e9, p.clockRate);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  { // MFMA f64 map
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + 0.37 * i + 1e-9 * i * i; B[i] = 0.5 - 0.11 * i + 3e-10 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s = fma(A[i*4+k], B[k*16+j], s); R[i*16+j] = s; }
    double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    mfma_f64_probe<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
    double md = 0; int nbit = 0; for (int i = 0; i < 256; ++i) { md = fmax(md, fabs(C[i]-R[i])); nbit += (C[i] == R[i]); }
    printf("mfma_f64_16x16x4 vs k-ordered fma chain: maxdiff %.3e bit-identical %d/256\n", md, nbit);
  }
  { // HBM copy
    size_t n = (size_t)1 << 28; // 4 GiB of double2
    double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
    copy_k<<<2048, 256>>>(a, b, n); CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) copy_k<<<2048, 256>>>(a, b, n);
    CK(hipEventRecord(e1)); float ms = elapsed(e0, e1) / 5;
    printf("copy 4GiB->4GiB: %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * n * 16 / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b));
  }
  rocblas_handle h; rocblas_create_handle(&h);
  { // dgemm shapes
    struct S { int m, n, k; rocblas_operation ta, tb; const char* nm; } shapes[] = {
      {4096, 4096, 4096, rocblas_operation_none, rocblas_operation_none, "square4k NN"},
      {8192, 8192, 8192, rocblas_operation_none, rocblas_operation_none, "square8k NN"},
      {2080, 512, 512000, rocblas_operation_transpose, rocblas_operation_none, "W cfg2: (P x nb) K=G  TN"},
      {1664, 1664, 216000, rocblas_operation_transpose, rocblas_operation_none, "J cfg3 chunk: N x N, K=G/8 TN"},
      {216000, 1664, 1664, rocblas_operation_none, rocblas_operation_none, "D*phi chunk: G/8 x N x N NN"},
      {16640, 16640, 1664, rocblas_operation_none, rocblas_operation_transpose, "K: P x P x N NT"},
      {130, 512, 13500, rocblas_operation_transpose, rocblas_operation_none, "W local block TN"},
    };
    for (auto& s : shapes) {
      size_t sa = (size_t)s.m * s.k, sb = (size_t)s.k * s.n, sc = (size_t)s.m * s.n;
      double *A, *B, *C; CK(hipMalloc(&A, sa * 8)); CK(hipMalloc(&B, sb * 8)); CK(hipMalloc(&C, sc * 8));
      std::vector<double> hA(1 << 20); for (auto& x : hA) x = rand() / (double)RAND_MAX - 0.5;
      for (size_t o = 0; o < sa; o += hA.size()) CK(hipMemcpy(A + o, hA.data(), std::min(hA.size(), sa - o) * 8, hipMemcpyHostToDevice));
      for (size_t o = 0; o < sb; o += hA.size()) CK(hipMemcpy(B + o, hA.data(), std::min(hA.size(), sb - o) * 8, hipMemcpyHostToDevice));
      double one = 1, zero = 0;
      int lda = s.ta == rocblas_operation_none ? s.m : s.k, ldb = s.tb == rocblas_operation_none ? s.k : s.n;
      rocblas_dgemm(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, lda, B, ldb, &zero, C, s.m);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      int reps = 3; for (int i = 0; i < reps; ++i) rocblas_dgemm(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, lda, B, ldb, &zero, C, s.m);
      CK(hipEventRecord(e1)); float ms = elapsed(e0, e1) / reps;
      printf("dgemm %-34s m=%d n=%d k=%d: %.3f ms %.1f TF/s\n", s.nm, s.m, s.n, s.k, ms, 2.0 * s.m * s.n * s.k / ms / 1e9);
      CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
  }
  { // dtrsm right/lower: X T' = B, B is (m x k) col-major, T' k x k
    int m = 512000, k = 2080; double *T, *B; CK(hipMalloc(&T, (size_t)k * k * 8)); CK(hipMalloc(&B, (size_t)m * k * 8));
    std::vector<double> hT((size_t)k * k, 0.0); for (int j = 0; j < k; ++j) for (int i = j; i < k; ++i) hT[(size_t)j * k + i] = (i == j) ? 2.0 + 0.001 * i : 0.3 / (1 + i - j);
    CK(hipMemcpy(T, hT.data(), hT.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(B, 0, (size_t)m * k * 8));
    double one = 1;
    rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, m, k, &one, T, k, B, m);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, m, k, &one, T, k, B, m);
    CK(hipEventRecord(e1)); float ms = elapsed(e0, e1);
    printf("dtrsm right-lower m=%d k=%d: %.3f ms %.1f TF/s\n", m, k, ms, 1.0 * m * k * k / ms / 1e9);
    CK(hipFree(T)); CK(hipFree(B));
  }
  for (int n : {80, 120, 160}) { // batched 3-D real FFT
    int nb = n == 160 ? 64 : (n == 120 ? 128 : 256);
    int dims[3] = {n, n, n}; size_t G = (size_t)n * n * n, Gc = (size_t)n * n * (n / 2 + 1);
    double* R; hipfftDoubleComplex* Cx; CK(hipMalloc(&R, nb * G * 8)); CK(hipMalloc(&Cx, nb * Gc * 16)); CK(hipMemset(R, 0, nb * G * 8));
    hipfftHandle pf, pb; size_t ws;
    if (hipfftPlanMany(&pf, 3, dims, nullptr, 1, (int)G, nullptr, 1, (int)Gc, HIPFFT_D2Z, nb) != HIPFFT_SUCCESS) { printf("plan fail\n"); return 1; }
    if (hipfftPlanMany(&pb, 3, dims, nullptr, 1, (int)Gc, nullptr, 1, (int)G, HIPFFT_Z2D, nb) != HIPFFT_SUCCESS) { printf("plan fail\n"); return 1; }
    hipfftExecD2Z(pf, R, Cx); hipfftExecZ2D(pb, Cx, R); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); int reps = 3;
    for (int i = 0; i < reps; ++i) { hipfftExecD2Z(pf, R, Cx); hipfftExecZ2D(pb, Cx, R); }
    CK(hipEventRecord(e1)); float ms = elapsed(e0, e1) / reps;
    printf("hipfft D2Z+Z2D %d^3 batch %d: %.3f ms per pair-batch, %.3f ms per field pair, algorithmic %.1f GB/s (32 B/pt)\n", n, nb, ms, ms / nb, 32.0 * G * nb / ms / 1e6);
    hipfftDestroy(pf); hipfftDestroy(pb); CK(hipFree(R)); CK(hipFree(Cx));
  }
  rocblas_destroy_handle(h);
  printf("probe done\n");
  return 0;
}
