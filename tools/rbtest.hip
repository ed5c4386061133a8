// probe: rocBLAS gemm_strided_batched_ex with f16 operands and f32 output (C[n][k] = sum_v A[v][n] B[v][k], row-major operands)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cmath>
int main() {
    const int H = 512, F = 2048, Vb = 32768, nb = 8;
    const long long V = (long long) Vb * nb;
    std::vector<_Float16> hA((size_t) V * H), hB((size_t) V * F);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = (_Float16) (((i * 2654435761u) >> 20) % 17 / 17.0f - 0.5f);
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = (_Float16) (((i * 40503u) >> 7) % 13 / 13.0f - 0.5f);
    _Float16 *A, *B; float *C;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&B, hB.size() * 2); hipMalloc(&C, (size_t) nb * H * F * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    rocblas_handle h; rocblas_create_handle(&h);
    const float alpha = 1.f, beta = 0.f;
    // column-major view: Bcm [F x V] (ld F), Acm [H x V] (ld H); Ccm[f, n] = sum_v Bcm[f,v] Acm[n,v]  -> row-major C[n][f]
    auto run = [&]() {
        return rocblas_gemm_strided_batched_ex(h, rocblas_operation_none, rocblas_operation_transpose, F, H, Vb, &alpha,
                                               B, rocblas_datatype_f16_r, F, (rocblas_stride) Vb * F,
                                               A, rocblas_datatype_f16_r, H, (rocblas_stride) Vb * H, &beta,
                                               C, rocblas_datatype_f32_r, F, (rocblas_stride) H * F,
                                               C, rocblas_datatype_f32_r, F, (rocblas_stride) H * F, nb,
                                               rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    };
    rocblas_status st = run();
    hipDeviceSynchronize();
    printf("status %d\n", (int) st);
    if (st != rocblas_status_success) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%.3f ms  %.1f TFLOP/s\n", ms, 2.0 * H * F * (double) V / ms / 1e9);
    std::vector<float> hC((size_t) H * F);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int n : {0, 5, 511}) for (int f : {0, 77, 2047}) {
        double s = 0;
        for (int v = 0; v < Vb; ++v) s += (double) (float) hA[(size_t) v * H + n] * (double) (float) hB[(size_t) v * F + f];
        err = fmax(err, fabs(s - hC[(size_t) n * F + f]));
    }
    printf("max err %.3e\n", err);
    return 0;
}
