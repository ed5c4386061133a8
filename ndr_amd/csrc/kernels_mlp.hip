// Fourier-feature MLP density field, fused forward pass on the gfx950 matrix cores.
//
// Reference: networks.MLP.forward (networks.py:181-185):
//     x in [0,1]^3  ->  gamma = [sin(2 pi x B^T), cos(2 pi x B^T)]   (B fixed, [es,3])
//                   ->  Linear(2es,nn)+ReLU -> (nl-2) x [Linear(nn,nn)+ReLU] -> Linear(nn,1) [-> Sigmoid]
// evaluated for every voxel of the grid each design iteration (train_xdg.py:282-287).  The reference
// materialises the [nVox, 2es] feature matrix (275 GB at 512x256x256); here nothing wider than the final
// scalar per voxel ever reaches HBM.
//
// One 512-thread block owns 128 voxels.  Activations live in LDS as [voxel][k] half precision with a 16-byte
// row pad (conflict-free ds_read_b128); every layer is computed transposed,  D[n][v] = sum_k W[n][k] X[v][k],
// with v_mfma_f32_32x32x16_f16: the A operand is a weight fragment (8 consecutive k of one output row =
// one 16-byte global load from the row-major [N][K] weight matrix, L2-resident), the B operand an activation
// fragment from LDS, and the 32x32 fp32 result holds, per lane, 4 consecutive n of one voxel per register
// quad, so the epilogue (bias, ReLU, fp32->fp16) writes 8-byte runs straight back into the [voxel][k] image.
// The Fourier features of the first layer are produced per 64-wide K chunk directly into LDS with
// v_fract/v_sin/v_cos (inputs in revolutions), double-buffered against the MFMAs of the previous chunk.
// Each wave owns 2 of the 16 output row tiles x all 4 voxel tiles = 8 accumulators (128 registers).
// fp16 operands / fp32 accumulation (the reference is fp32 end to end; measured error in tests/test_gpu_mlp.py).
#include "vfem_internal.h"

#include <hip/hip_fp16.h>

#include "mlp_args.h"

namespace vfem {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));

constexpr int MLP_TM = 128;                 // voxels per block
constexpr int MLP_MAXN = 512;               // max hidden width
constexpr int MLP_HSTRIDE = MLP_MAXN + 8;   // halves per activation row (16-byte pad)
constexpr int MLP_KC = 64;                  // feature chunk
constexpr int MLP_FSTRIDE = MLP_KC + 8;     // halves per feature row (16-byte pad)

__device__ __forceinline__ void voxel_coord(const MlpArgs &a, long long v, float x[3]) {
    if (a.coords) { x[0] = a.coords[3 * v]; x[1] = a.coords[3 * v + 1]; x[2] = a.coords[3 * v + 2]; return; }
    const long long k = v % a.gn[2], j = (v / a.gn[2]) % a.gn[1], i = v / ((long long) a.gn[2] * a.gn[1]);
    x[0] = a.glo[0] + a.gstep[0] * (float) i;
    x[1] = a.glo[1] + a.gstep[1] * (float) j;
    x[2] = a.glo[2] + a.gstep[2] * (float) k;
}

// one GEMM layer: acc[t][c] (row tile t of this wave, voxel tile c) += W[n][k] X[v][k] over k in [0, K)
// X image: xs[v * xstride + k]; row tiles of this wave: rt0 + 8 t (t < ntile)
template <int XSTRIDE>
__device__ __forceinline__ void gemm_chunk(f16_t acc[2][4], const _Float16 *__restrict__ W, int ldw, int k_base,
                                           const _Float16 *xs, int k_local0, int ksteps, int wave, int ntiles, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll 2
    for (int ks = 0; ks < ksteps; ++ks) {
        h8_t bfrag[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            bfrag[c] = *reinterpret_cast<const h8_t *>(xs + (c * 32 + r) * XSTRIDE + k_local0 + ks * 16 + 8 * h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wave + 8 * t;
            if (tile < ntiles) {
                const h8_t afrag = *reinterpret_cast<const h8_t *>(W + (long long) (tile * 32 + r) * ldw + k_base + ks * 16 + 8 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag[c], acc[t][c], 0, 0, 0);
            }
        }
    }
}

__global__ void __launch_bounds__(512) k_mlp_forward(MlpArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    _Float16 *H = reinterpret_cast<_Float16 *>(smem);                            // [128][HSTRIDE]
    _Float16 *F = reinterpret_cast<_Float16 *>(smem);                            // layer 1: 2 x [128][FSTRIDE] (aliases H)
    float *xc = reinterpret_cast<float *>(smem + (size_t) MLP_TM * MLP_HSTRIDE * 2);   // [128][3] coordinates

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long v0 = (long long) blockIdx.x * MLP_TM;
    const int ntiles = a.nn / 32;

    if (tid < MLP_TM) {
        float x[3] = {0.f, 0.f, 0.f};
        if (v0 + tid < a.nvox) voxel_coord(a, v0 + tid, x);
        xc[3 * tid] = x[0]; xc[3 * tid + 1] = x[1]; xc[3 * tid + 2] = x[2];
    }
    __syncthreads();

    f16_t acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

    // ---- layer 1: K = 2 es, features generated chunk by chunk --------------------------------
    const int K1 = 2 * a.es;
    const int nchunks = K1 / MLP_KC;
    auto make_features = [&](int chunk, int buf) {
        _Float16 *Fb = F + buf * (MLP_TM * MLP_FSTRIDE);
        // 128 voxels x 64 features = 8192 values, 16 per thread: thread -> (voxel = tid & 127, 16 features)
        const int v = tid & 127, fq = tid >> 7;                  // fq in 0..3 -> features fq*16 .. +15
        const float x0 = xc[3 * v], x1 = xc[3 * v + 1], x2 = xc[3 * v + 2];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            h4_t o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int f = chunk * MLP_KC + fq * 16 + j + jj;     // feature index in [0, 2 es)
                const bool is_cos = f >= a.es;
                const int fi = is_cos ? f - a.es : f;
                const float t = fmaf(x0, a.B[3 * fi], fmaf(x1, a.B[3 * fi + 1], x2 * a.B[3 * fi + 2]));   // revolutions
                const float fr = t - floorf(t);
                const float s = is_cos ? __builtin_amdgcn_cosf(fr) : __builtin_amdgcn_sinf(fr);
                o[jj] = (_Float16) s;
            }
            *reinterpret_cast<h4_t *>(Fb + v * MLP_FSTRIDE + fq * 16 + j) = o;
        }
    };
    make_features(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) make_features(ch + 1, (ch + 1) & 1);
        gemm_chunk<MLP_FSTRIDE>(acc, (const _Float16 *) a.W1, K1, ch * MLP_KC, F + (ch & 1) * (MLP_TM * MLP_FSTRIDE), 0, MLP_KC / 16, wave, ntiles, lane);
        __syncthreads();
    }

    // ---- epilogue of a layer: bias + ReLU -> fp16 activations in H ([voxel][k]) ----------------
    auto store_layer = [&](const float *bias) {
        const int col = lane & 31, h = lane >> 5;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wave + 8 * t;
            if (tile >= ntiles) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tile * 32 + 8 * g + 4 * h;     // rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                    h4_t o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float vv = acc[t][c][4 * g + q] + bias[n + q];
                        o[q] = (_Float16) (vv > 0.f ? vv : 0.f);
                        acc[t][c][4 * g + q] = 0.f;
                    }
                    *reinterpret_cast<h4_t *>(H + (c * 32 + col) * MLP_HSTRIDE + n) = o;
                }
        }
    };
    store_layer(a.bias);          // all waves passed the last barrier of the chunk loop: F is dead, H may be written
    __syncthreads();

    // ---- hidden layers ----------------------------------------------------------------------
    for (int l = 0; l < a.n_hidden; ++l) {
        gemm_chunk<MLP_HSTRIDE>(acc, (const _Float16 *) a.Wh + (long long) l * a.nn * a.nn, a.nn, 0, H, 0, a.nn / 16, wave, ntiles, lane);
        __syncthreads();          // every wave finished reading H
        store_layer(a.bias + (l + 1) * a.nn);
        __syncthreads();
    }

    // ---- output layer: one scalar per voxel, 4 threads per voxel ------------------------------
    {
        const int v = tid >> 2, part = tid & 3;
        const int kper = a.nn / 4;
        float s = 0.f;
        for (int k = part * kper; k < (part + 1) * kper; ++k) s = fmaf((float) H[v * MLP_HSTRIDE + k], a.wout[k], s);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0 && v0 + v < a.nvox) {
            float o = s + a.bout;
            if (a.sigmoid) o = 1.f / (1.f + __expf(-o));
            if (a.out32) a.out32[v0 + v] = o;
            if (a.out64) a.out64[v0 + v] = (double) o;
        }
    }
}

void launch_mlp_forward(const MlpArgs &a, hipStream_t s) {
    const size_t lds = (size_t) MLP_TM * MLP_HSTRIDE * 2 + MLP_TM * 3 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        attr_set = true;
    }
    const long long blocks = (a.nvox + MLP_TM - 1) / MLP_TM;
    k_mlp_forward<<<dim3((unsigned) blocks), dim3(512), lds, s>>>(a);
    VFEM_HIP(hipGetLastError());
}

__global__ void k_f32_to_f16(long long n, const float *__restrict__ in, _Float16 *__restrict__ out) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
        out[i] = (_Float16) in[i];
}
void launch_f32_to_f16(long long n, const float *in, void *out, hipStream_t s) {
    long long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    k_f32_to_f16<<<dim3((unsigned) g), dim3(256), 0, s>>>(n, in, (_Float16 *) out);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
