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
    v += a.v_offset;
    const long long k = v % a.gn[2], j = (v / a.gn[2]) % a.gn[1], i = v / ((long long) a.gn[2] * a.gn[1]);
    x[0] = a.glo[0] + a.gstep[0] * (float) i;
    x[1] = a.glo[1] + a.gstep[1] * (float) j;
    x[2] = a.glo[2] + a.gstep[2] * (float) k;
}

// Weight (A operand) fragments are prefetched MLP_PD k-steps ahead in a register ring: a fragment is one 16-byte load from
// the L2-resident weight matrix, whose latency (~600 cycles) is several times the 256 MFMA cycles of a k-step, so without
// the ring the matrix pipe idles on every step.
constexpr int MLP_PD = 4;

template <bool FULL>                            // FULL: 16 row tiles (both tiles of every wave live), k-steps a multiple of MLP_PD
struct APipe {
    const _Float16 *row[2];
    bool on[2];
    h8_t ring[MLP_PD][2];
    int total;                                  // k-steps of the whole layer

    __device__ __forceinline__ void load(int slot, int ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (FULL || on[t]) ring[slot][t] = *reinterpret_cast<const h8_t *>(row[t] + ks * 512);
    }
    __device__ __forceinline__ void init(const _Float16 *W, int ldw, int wave, int ntiles, int lane, int total_ksteps) {
        total = total_ksteps;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wave + 8 * t;
            on[t] = FULL || tile < ntiles;
            row[t] = W + ((long long) (on[t] ? tile : 0) * (ldw >> 4) * 64 + lane) * 8;       // fragment order [row tile][k-step][lane][8] (k_f32_to_f16_frag)
        }
#pragma unroll
        for (int p = 0; p < MLP_PD; ++p)
            if (FULL || p < total) load(p, p);
    }
};

// MLP_PD consecutive k-steps (one revolution of the ring) starting at k-step ks0 of the layer:
// acc[t][c] (row tile t of this wave, voxel tile c) += W[n][k] X[v][k];  X image: xs[v * XSTRIDE + k_local], the first
// k-step of this call sits at k_local0.
// Operand (B) fragments alternate between two register sets at compile time -- no copies, so no MFMA-source hazards -- and
// the fragments of the first k-step are handed in by the caller (`bpre`), which receives those of the next revolution
// (image `xs_next`, offset `k_next`; null = none) in return: the LDS latency is never exposed between revolutions.
template <int XSTRIDE>
__device__ __forceinline__ void load_bfrags(h8_t (&b)[4], const _Float16 *xs, int k_local, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const h8_t *>(xs + (c * 32 + r) * XSTRIDE + k_local + 8 * h);
}

template <int XSTRIDE, bool FULL>
__device__ __forceinline__ void gemm_ring(f16_t acc[2][4], APipe<FULL> &ap, int ks0, const _Float16 *xs, int k_local0, int lane,
                                          h8_t (&bpre)[4], const _Float16 *xs_next, int k_next) {
    static_assert(MLP_PD % 2 == 0, "the two operand register sets alternate with the k-step parity");
    h8_t balt[4];
#pragma unroll
    for (int p = 0; p < MLP_PD; ++p) {
        const int ks = ks0 + p;
        if (FULL || ks < ap.total) {
            h8_t (&bc)[4] = (p & 1) ? balt : bpre;          // fragments of this k-step
            h8_t (&bn)[4] = (p & 1) ? bpre : balt;          // filled for the next one
            if (p + 1 < MLP_PD) { if (FULL || ks + 1 < ap.total) load_bfrags<XSTRIDE>(bn, xs, k_local0 + (p + 1) * 16, lane); }
            else if (xs_next) load_bfrags<XSTRIDE>(bn, xs_next, k_next, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (FULL || ap.on[t]) {
                    const h8_t afrag = ap.ring[p][t];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bc[c], acc[t][c], 0, 0, 0);
                }
            }
            if (ks + MLP_PD < ap.total) ap.load(p, ks + MLP_PD);
        }
    }
}

// a whole layer whose operand image is resident in LDS (hidden layers, backward data path)
template <int XSTRIDE, bool FULL>
__device__ __forceinline__ void gemm_layer(f16_t acc[2][4], const _Float16 *W, int ldw, int K, const _Float16 *xs, int wave, int ntiles,
                                           int lane) {
    APipe<FULL> ap;
    ap.init(W, ldw, wave, ntiles, lane, K / 16);
    h8_t bpre[4];
    load_bfrags<XSTRIDE>(bpre, xs, 0, lane);
    for (int ks0 = 0; ks0 < ap.total; ks0 += MLP_PD) {
        const bool more = ks0 + MLP_PD < ap.total;
        gemm_ring<XSTRIDE, FULL>(acc, ap, ks0, xs, ks0 * 16, lane, bpre, more ? xs : nullptr, (ks0 + MLP_PD) * 16);
    }
}

template <bool FULL>
__global__ void __launch_bounds__(512) k_mlp_forward(MlpArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    _Float16 *H = reinterpret_cast<_Float16 *>(smem);                            // [128][HSTRIDE]
    _Float16 *F = reinterpret_cast<_Float16 *>(smem);                            // layer 1: 3 x [128][FSTRIDE] (aliases H)
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
    // 128 voxels x 64 features = 8192 values, 16 per thread in four parts of 4: thread -> (voxel = tid & 127, features fq*16 ..)
    auto make_features_part = [&](int chunk, int buf, int part) {
        _Float16 *Fb = F + buf * (MLP_TM * MLP_FSTRIDE);
        // fq in 0..3 -> features fq*16 .. +15; constant inside a wave, so the rows of B are fetched with scalar loads
        const int v = tid & 127, fq = __builtin_amdgcn_readfirstlane(tid >> 7);
        const float x0 = xc[3 * v], x1 = xc[3 * v + 1], x2 = xc[3 * v + 2];
        {
            const int j = 4 * part;
            // the 16 features of (chunk, fq) are all sines or all cosines (es is a multiple of 32) and their rows of B are
            // contiguous: one wave-uniform base pointer, constant offsets -> a few wide scalar loads instead of one address
            // computation per feature (the scalar unit was executing ~6 instructions per MFMA)
            const int f0 = chunk * MLP_KC + fq * 16;                 // first feature index in [0, 2 es)
            const bool is_cos = f0 >= a.es;
            const float *Bp = a.B + 3 * (is_cos ? f0 - a.es : f0) + 3 * j;
            const float ph = is_cos ? 0.25f : 0.f;                   // cos x = sin(x + 1/4 turn): one transcendental per feature
            h4_t o;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float t = fmaf(x0, Bp[3 * jj], fmaf(x1, Bp[3 * jj + 1], fmaf(x2, Bp[3 * jj + 2], ph)));   // phase in revolutions
                const float fr = t - floorf(t);
                o[jj] = (_Float16) __builtin_amdgcn_sinf(fr);
            }
            *reinterpret_cast<h4_t *>(Fb + v * MLP_FSTRIDE + fq * 16 + j) = o;
        }
    };
    auto make_features = [&](int chunk, int buf) {
#pragma unroll
        for (int part = 0; part < 4; ++part) make_features_part(chunk, buf, part);
    };
    APipe<FULL> ap1;
    ap1.init((const _Float16 *) a.W1, K1, wave, ntiles, lane, K1 / 16);      // weight prefetch runs across the feature chunks
    // Three feature buffers: while chunk ch is multiplied, chunk ch+2 is generated.  The two waves that share a SIMD (w and
    // w+4) do the two halves of an iteration in opposite order, so that one is on the vector pipe (features) while the other
    // is on the matrix pipe; with both in the same order the barrier per chunk keeps them in lockstep and the pipes alternate.
    make_features(0, 0);
    if (nchunks > 1) make_features(1, 1);
    __syncthreads();
    static_assert(MLP_KC / 16 == MLP_PD, "one feature chunk = one revolution of the weight ring");
    h8_t bpre1[4];
    load_bfrags<MLP_FSTRIDE>(bpre1, F, 0, lane);
    for (int ch = 0, cur = 0, nxt = 2; ch < nchunks; ++ch) {
        const bool more = ch + 2 < nchunks && a.ablate != 1;
        // the buffer of chunk ch+1 is complete (written during iteration ch-1, a barrier ago): its first fragments are fetched
        // at the end of this chunk's MFMAs
        const int after = cur == 2 ? 0 : cur + 1;
        const _Float16 *Fn = ch + 1 < nchunks ? F + after * (MLP_TM * MLP_FSTRIDE) : nullptr;
        if (wave < 4) {
            if (more) make_features(ch + 2, nxt);
            if (a.ablate != 3) gemm_ring<MLP_FSTRIDE, FULL>(acc, ap1, ch * MLP_PD, F + cur * (MLP_TM * MLP_FSTRIDE), 0, lane, bpre1, Fn, 0);
        } else {
            if (a.ablate != 3) gemm_ring<MLP_FSTRIDE, FULL>(acc, ap1, ch * MLP_PD, F + cur * (MLP_TM * MLP_FSTRIDE), 0, lane, bpre1, Fn, 0);
            if (more) make_features(ch + 2, nxt);
        }
        __syncthreads();
        cur = after;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }

    // ---- epilogue of a layer: bias + ReLU -> fp16 activations in H ([voxel][k]) ----------------
    auto store_layer = [&](const float *bias) {
        const int col = lane & 31, h = lane >> 5;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wave + 8 * t;
            if (!FULL && tile >= ntiles) continue;
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
    // training: copy the [voxel][k] fp16 image of a finished layer to HBM in 16-byte pieces (coalesced)
    auto save_layer = [&](int layer) {
        if (!a.save_act) return;
        _Float16 *dst = reinterpret_cast<_Float16 *>(a.save_act) + ((long long) layer * a.act_rows + v0) * a.nn;
        const int ppr = a.nn / 8;
        for (int q = tid; q < MLP_TM * ppr; q += 512) {
            const int v = q / ppr, c = q - v * ppr;
            if (v0 + v < a.nvox)
                *reinterpret_cast<h8_t *>(dst + (long long) v * a.nn + 8 * c) = *reinterpret_cast<const h8_t *>(H + v * MLP_HSTRIDE + 8 * c);
        }
    };
    store_layer(a.bias);          // all waves passed the last barrier of the chunk loop: F is dead, H may be written
    __syncthreads();
    save_layer(0);

    // ---- hidden layers ----------------------------------------------------------------------
    for (int l = 0; l < (a.ablate == 2 ? 0 : a.n_hidden); ++l) {
        gemm_layer<MLP_HSTRIDE, FULL>(acc, (const _Float16 *) a.Wh + (long long) l * a.nn * a.nn, a.nn, a.nn, H, wave, ntiles, lane);
        __syncthreads();          // every wave finished reading H
        store_layer(a.bias + (l + 1) * a.nn);
        __syncthreads();
        save_layer(l + 1);
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
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        attr_set = true;
    }
    const long long blocks = (a.nvox + MLP_TM - 1) / MLP_TM;
    if (a.nn == MLP_MAXN) k_mlp_forward<true><<<dim3((unsigned) blocks), dim3(512), lds, s>>>(a);      // every branch on tile / k-step validity folds away
    else                  k_mlp_forward<false><<<dim3((unsigned) blocks), dim3(512), lds, s>>>(a);
    VFEM_HIP(hipGetLastError());
}


// out[i] = beta * out[i] + alpha * sum_b partial[b][i]; four consecutive outputs per thread (16-byte loads), the loop over
// the partials unrolled so that many loads are in flight
__global__ void __launch_bounds__(256) k_reduce_partials(int nb, long long n, const float *__restrict__ partial, float alpha,
                                                         float beta, float *__restrict__ out) {
    const long long i4 = ((long long) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n && (n & 3) == 0) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int b = 0; b < nb; ++b) {
            const float4 v = *reinterpret_cast<const float4 *>(partial + (long long) b * n + i4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (beta != 0.f) { o = *reinterpret_cast<const float4 *>(out + i4); o.x *= beta; o.y *= beta; o.z *= beta; o.w *= beta; }
        o.x += alpha * acc.x; o.y += alpha * acc.y; o.z += alpha * acc.z; o.w += alpha * acc.w;
        *reinterpret_cast<float4 *>(out + i4) = o;
        return;
    }
    for (long long i = i4; i < n && i < i4 + 4; ++i) {
        float acc = 0.f;
        for (int b = 0; b < nb; ++b) acc += partial[(long long) b * n + i];
        out[i] = (beta == 0.f ? 0.f : beta * out[i]) + alpha * acc;
    }
}
void launch_reduce_partials(int nb, long long n, const float *partial, float alpha, float beta, float *out, hipStream_t s) {
    k_reduce_partials<<<dim3((unsigned) (((n + 3) / 4 + 255) / 256)), dim3(256), 0, s>>>(nb, n, partial, alpha, beta, out);
    VFEM_HIP(hipGetLastError());
}

// sum of a float vector in two passes: 4096 elements per block into partial[], then one small block over the partials
__global__ void __launch_bounds__(256) k_sum_f32_partial(long long n, const float *__restrict__ x, float *__restrict__ partial) {
    __shared__ float sm[256];
    const long long base = (long long) blockIdx.x * 4096;
    float acc = 0.f;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const long long i = base + threadIdx.x + 256 * k;
        if (i < n) acc += x[i];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int) threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void __launch_bounds__(256) k_sum_f32(long long n, const float *__restrict__ x, float alpha, float beta, float *__restrict__ out) {
    __shared__ float sm[256];
    float acc = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) acc += x[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int) threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (beta == 0.f ? 0.f : beta * out[0]) + alpha * sm[0];
}
void launch_sum_f32(long long n, const float *x, float alpha, float beta, float *out, float *scratch, hipStream_t s) {
    const long long nblk = (n + 4095) / 4096;                      // scratch holds >= nblk floats
    k_sum_f32_partial<<<dim3((unsigned) nblk), dim3(256), 0, s>>>(n, x, scratch);
    k_sum_f32<<<dim3(1), dim3(256), 0, s>>>(nblk, scratch, alpha, beta, out);
    VFEM_HIP(hipGetLastError());
}

// torch.optim.Adam step (amsgrad off, weight_decay 0): one fused pass over a parameter tensor
__global__ void __launch_bounds__(256) k_adam(long long n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                              float *__restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
void launch_adam(long long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps, int step, hipStream_t s) {
    const float bc1 = 1.f - powf(b1, (float) step), bc2 = 1.f - powf(b2, (float) step);
    k_adam<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s>>>(n, p, g, m, v, lr, b1, b2, eps, bc1, bc2);
    VFEM_HIP(hipGetLastError());
}

// fp32 [N][K] (or, transposed = 1, the transpose of an fp32 [K][N]) -> fp16 in MFMA-fragment order [N / 32][K / 16][lane][8]: the A
// operand of a (row tile, k-step) is one contiguous KB instead of 32 pieces of 32 bytes (kernels_mlp_x3.hip: k_split_f32_frag)
__global__ void k_f32_to_f16_frag(int N, int K, int transposed, const float *__restrict__ in, _Float16 *__restrict__ out) {
    const long long n = (long long) N * K;
    const int nks = K / 16;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const int row = (int) (i / K), k = (int) (i - (long long) row * K);
        const long long o = ((((long long) (row >> 5) * nks + (k >> 4)) * 64) + (row & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
        out[o] = (_Float16) (transposed ? in[(long long) k * N + row] : in[i]);
    }
}
void launch_f32_to_f16_frag(int N, int K, int transposed, const float *in, void *out, hipStream_t s) {
    if (N % 32 || K % 16) throw Error("fragment-order weights need N % 32 == 0 and K % 16 == 0");
    long long g = ((long long) N * K + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    k_f32_to_f16_frag<<<dim3((unsigned) g), dim3(256), 0, s>>>(N, K, transposed, in, (_Float16 *) out);
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
