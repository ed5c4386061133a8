// Fourier-feature MLP forward at the reference's precision (networks.MLP.forward is fp32 end to end, networks.py:178-185), fused,
// on the f16 matrix pipe with SPLIT operands.
//
// Every operand x (a weight, a Fourier feature, a hidden activation) is carried as two halves
//     hi = fp16(x),   lo = fp16((x - hi) * 2^11)        (x = hi + lo 2^-11 to 2^-22 relative: 22 significant bits)
// and a product of two operands is three MFMA products with fp32 accumulation,
//     w x  ~=  wh xh  +  2^-11 (wh xl + wl xh)          (the dropped wl xl term is 2^-22 relative),
// the first into one accumulator, the two cross terms into a second one that is folded in at 2^-11 in the epilogue.  Scaling
// only the low halves keeps THEM in fp16's normal range whatever the magnitude of x; the high half is fp16(x) itself, so |x| must stay
// below fp16's largest number (65 504): weights are checked when they are loaded, hidden activations by a flag the kernel raises and the
// next entry point reports (vfem_mlp_*: "activation outside fp16's range") -- the fp32 reference has no such limit.  An f16
// product is exact in fp32 (11 x 11 bits), so the result differs from an fp32 FMA chain only by the dropped term and by the
// summation order.  v_mfma_f32_32x32x2_f32 (exact fp32, 1/16 of the f16 rate) would spend 16 x the matrix cycles, this
// spends 3 x; the rocBLAS SGEMM chain of rounds 1-2 ran at 33 Mvoxel/s and passed every activation through HBM.
//
// Tiling as the fp16 kernel (kernels_mlp.hip), with half the voxels per block so that both halves of the activation image fit
// in LDS: 512 threads own 64 voxels; activations [voxel][k] fp16 x 2 (hi, lo) with a 16-byte row pad; every layer transposed,
// D[n][v] = sum_k W[n][k] X[v][k] with v_mfma_f32_32x32x16_f16, a wave owns 2 of the 16 row tiles x both voxel tiles
// (4 + 4 accumulators).  Fourier features: arg = (2 pi x) . B_j in fp32 exactly as the reference forms it, accurate sinf / cosf,
// generated per 64-wide K chunk into LDS, double-buffered against the previous chunk's MFMAs.
#include "vfem_internal.h"

#include <hip/hip_fp16.h>

#include "mlp_args.h"

namespace vfem {

namespace x3 {
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
constexpr int TM = 64;                    // voxels per block
constexpr int MAXN = 512;                 // hidden width limit
constexpr int HS = MAXN + 8;              // halves per activation row (16-byte pad: conflict-free ds_read_b128)
// feature chunk of the first layer: KC columns = the sines of KC / 2 rows of B followed by their cosines (one barrier per chunk: 128
// where the embedding size allows it -- a multiple of 64 --, else 64); FS = KC + 8 halves per staged feature row
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void split(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16) x;
    lo = (_Float16) ((x - (float) hi) * LO_SCALE);
}
// sin and cos of an fp32 argument together, to fp32 rounding (max error 9.2e-8 for |t| <= 1000, tools/ numpy check in DESIGN 3.5; numpy's own
// fp32 sin: 7e-8): one Cody-Waite reduction by pi/2 in three fma steps (pi/2 = c1 + c2 + c3), the cephes single-precision minimax
// polynomials on [-pi/4, pi/4], quadrant by the low bits of n.  ~25 instructions for the pair; two library calls (sinf, cosf) were ~90
// and made feature generation 27 % of the SIMD time of the forward kernel (profiles/r04_mlp_x3_pmc.json: 1686 vector instructions per voxel).
__device__ __forceinline__ void sincos_f32(float t, float &sn, float &cs) {
    const float n = __builtin_rintf(t * 0.636619772367581343f);
    float y = fmaf(-n, 1.5707963705062866f, t);
    y = fmaf(-n, -4.371138828673793e-08f, y);
    y = fmaf(-n, -1.7763568394002505e-15f, y);
    const float z = y * y;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s = fmaf(ps * z, y, y);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const int q = (int) n;
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void voxel_xyz(const MlpArgs &a, long long v, float x[3]) {
    if (a.coords) { x[0] = a.coords[3 * v]; x[1] = a.coords[3 * v + 1]; x[2] = a.coords[3 * v + 2]; return; }
    v += a.v_offset;
    const long long k = v % a.gn[2], j = (v / a.gn[2]) % a.gn[1], i = v / ((long long) a.gn[2] * a.gn[1]);
    x[0] = a.glo[0] + a.gstep[0] * (float) i;
    x[1] = a.glo[1] + a.gstep[1] * (float) j;
    x[2] = a.glo[2] + a.gstep[2] * (float) k;
}
// the low half without its 2^11 scale (what the backward pass's products take).  A nonzero half never becomes zero: an activation
// of 1e-9 has hi = 0 and lives in its low half alone, whose unscaled value underflows fp16 -- and "h > 0" is the ReLU mask of the
// backward pass (the smallest subnormal, 6e-8, stands in: its value is immaterial, its sign is not)
__device__ __forceinline__ _Float16 unscale_lo(_Float16 ls) {
    const float f = (float) ls;
    _Float16 u = (_Float16) (f * LO_INV);
    if ((float) u == 0.f && f != 0.f) u = (_Float16) (f > 0.f ? 5.9604645e-8f : -5.9604645e-8f);
    return u;
}
}  // namespace x3

struct MlpX3Weights { const _Float16 *W1h, *W1l, *Whh, *Whl; };

// weights: hi / lo halves of an fp32 array
__global__ void __launch_bounds__(256) k_split_f32(long long n, const float *__restrict__ in, _Float16 *__restrict__ hi, _Float16 *__restrict__ lo) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) x3::split(in[i], hi[i], lo[i]);
}
// The same halves in MFMA-FRAGMENT order: the A operand of v_mfma_f32_32x32x16_f16 for the row tile T (32 rows) and k-step S is, per
// lane (r = lane & 31, h = lane >> 5), the eight halves W[32 T + r][16 S + 8 h .. + 7].  Row-major, the 64 lanes of that load touch 32
// rows = 32 separate 32-byte pieces: the forward kernel then sits on the L2's REQUEST rate (profiles/r04_mlp_x3_pmc.json: 99 % L2 hits,
// 0.74 requests per clock and channel, waves 63 % of their time in s_waitcnt).  Stored as [T][S][lane][8] the load is one contiguous KB
// = eight whole 128-byte lines, and successive k-steps of a tile follow each other in memory.
// pair_es > 0 (first layer): the K order is permuted so that a pair_kc-wide chunk holds the sines of pair_kc / 2 rows of B followed by the
// cosines of the same rows (the forward kernel forms both from one argument)
// transposed: `in` is the fp32 [K][N] matrix whose transpose is packed (hidden weights for the backward data pass)
__global__ void __launch_bounds__(256) k_split_f32_frag(int N, int K, int pair_es, int pair_kc, int transposed, const float *__restrict__ in, _Float16 *__restrict__ hi, _Float16 *__restrict__ lo) {
    const long long n = (long long) N * K;
    const int nks = K / 16;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const int row = (int) (i / K);
        int k = (int) (i - (long long) row * K);
        if (pair_es > 0) { const int f = k < pair_es ? k : k - pair_es; k = pair_kc * (f / (pair_kc / 2)) + f % (pair_kc / 2) + (k < pair_es ? 0 : pair_kc / 2); }
        const long long o = ((((long long) (row >> 5) * nks + (k >> 4)) * 64) + (row & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
        x3::split(transposed ? in[(long long) (int) (i - (long long) row * K) * N + row] : in[i], hi[o], lo[o]);
    }
}
void launch_split_f32_frag(int N, int K, const float *in, void *hi, void *lo, hipStream_t s, int pair_es, int transposed, int pair_kc) {
    if (N % 32 || K % 16) throw Error("fragment-order weights need N % 32 == 0 and K % 16 == 0");
    if (pair_es > 0 && (K != 2 * pair_es || pair_es % (pair_kc / 2) || (pair_kc != 64 && pair_kc != 128))) throw Error("sine / cosine pairing needs K = 2 es and es a multiple of half the chunk width (64 or 128)");
    long long g = ((long long) N * K + 255) / 256;
    if (g > 4096) g = 4096;
    k_split_f32_frag<<<dim3((unsigned) (g < 1 ? 1 : g)), 256, 0, s>>>(N, K, pair_es, pair_kc, transposed, in, (_Float16 *) hi, (_Float16 *) lo);
    VFEM_HIP(hipGetLastError());
}
// 1 in *flag when any |in[i]| is not below `limit` (weights of the split-operand kernels: fp16(w) must be finite)
__global__ void __launch_bounds__(256) k_range_check_f32(long long n, const float *__restrict__ in, float limit, int *__restrict__ flag) {
    bool bad = false;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) bad |= !(fabsf(in[i]) < limit);
    if (bad) *flag = 1;
}
void launch_range_check_f32(long long n, const float *in, float limit, int *flag, hipStream_t s) {
    if (n <= 0) return;
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    k_range_check_f32<<<dim3((unsigned) g), 256, 0, s>>>(n, in, limit, flag);
    VFEM_HIP(hipGetLastError());
}
void launch_split_f32(long long n, const float *in, void *hi, void *lo, hipStream_t s) {
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    k_split_f32<<<dim3((unsigned) (g < 1 ? 1 : g)), 256, 0, s>>>(n, in, (_Float16 *) hi, (_Float16 *) lo);
    VFEM_HIP(hipGetLastError());
}

template <bool FULL, int KC>        // FULL: hidden width 512 = 16 row tiles, both tiles of every wave live: the branches on tile validity fold away
__global__ void __launch_bounds__(512) k_mlp_forward_x3(MlpArgs a, MlpX3Weights w) {
    using namespace x3;
    constexpr int FS = KC + 8;
    extern __shared__ __align__(16) unsigned char smem[];
    _Float16 *Hh = reinterpret_cast<_Float16 *>(smem);                       // [64][HS] high halves of the activations
    _Float16 *Hl = Hh + TM * HS;                                             // low halves
    _Float16 *F = Hh;                                                        // layer 1: 2 buffers x (hi, lo) x [64][FS], aliases the activation images
    float *xc = reinterpret_cast<float *>(smem + (size_t) 2 * TM * HS * 2);  // [64][3] coordinates, pre-multiplied by 2 pi

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long v0 = (long long) blockIdx.x * TM;
    const int ntiles = a.nn / 32;
    const bool on[2] = {FULL || wave < ntiles, FULL || wave + 8 < ntiles};

    if (tid < TM) {
        float x[3] = {0.f, 0.f, 0.f};
        if (v0 + tid < a.nvox) voxel_xyz(a, v0 + tid, x);
        const float twopi = 6.283185307179586f;                              // (2. * math.pi * coords) in fp32, networks.py:179
        xc[3 * tid] = twopi * x[0]; xc[3 * tid + 1] = twopi * x[1]; xc[3 * tid + 2] = twopi * x[2];
    }
    __syncthreads();

    // the younger half of the workgroup (waves 4-7) runs at priority 1 throughout: the two waves of a SIMD then do not contend
    // symmetrically for the matrix pipe (+1.5 %; raising the priority around every MFMA cluster instead: -1 %).  The guard is a scalar
    // compare: s_setprio ignores EXEC
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    f16_t acch[2][2], accx[2][2];         // [row tile of this wave][voxel tile]: hi x hi products / cross products (scaled by 2^11)
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) { acch[t][c][q] = 0.f; accx[t][c][q] = 0.f; }
    };
    zero_acc();

    // one k-step (16 of K) of this wave's tiles: A = weight fragments (registers), B = activation fragments (LDS images xh / xl)
    auto kstep = [&](const h8_t (&ah)[2], const h8_t (&al)[2], const _Float16 *xh, const _Float16 *xl, int stride, int klocal) {
        h8_t bh[2], bl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bh[c] = *reinterpret_cast<const h8_t *>(xh + (c * 32 + r) * stride + klocal + 8 * h);
            bl[c] = *reinterpret_cast<const h8_t *>(xl + (c * 32 + r) * stride + klocal + 8 * h);
        }
        // term by term over the four tiles: two products into the same accumulator are then four MFMAs apart, not back to back
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!FULL && !on[t]) continue;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (term == 0) acch[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[c], acch[t][c], 0, 0, 0);
                    if (term == 1) accx[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[c], accx[t][c], 0, 0, 0);
                    if (term == 2) accx[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[c], accx[t][c], 0, 0, 0);
                }
            }
    };
    // weight fragments of k-step ks of a layer (fragment order [row tile][k-step][lane][8 halves], k_split_f32_frag): one 16-byte load
    // per lane, one contiguous KB per wave
    auto load_a = [&](const _Float16 *Wh_, const _Float16 *Wl_, int ldw, int ks, h8_t (&ah)[2], h8_t (&al)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long off = (((long long) ((FULL || on[t]) ? wave + 8 * t : 0) * (ldw >> 4) + ks) * 64 + lane) * 8;
            ah[t] = *reinterpret_cast<const h8_t *>(Wh_ + off);
            al[t] = *reinterpret_cast<const h8_t *>(Wl_ + off);
        }
    };

    constexpr int PD = 4;
    h8_t ah[PD][2], al[PD][2];
    // ---- layer 1: K = 2 es, features generated chunk by chunk ------------------------------------------------------------
    const int K1 = 2 * a.es, nchunks = K1 / KC;
    if (a.h0_hi == nullptr) {
    // a chunk = the sines and the cosines of KC / 2 rows of B (the K order of W1 is permuted to match, k_split_f32_frag): 64 voxels x
    // KC / 2 arguments, thread -> voxel tid & 63, rows FR (tid >> 6) .. + FR - 1, one sincos per argument
    constexpr int HR = KC / 2, FR = HR / 8;
    auto make_features = [&](int chunk, int buf) {
        _Float16 *Fh = F + (2 * buf) * (TM * FS), *Fl = Fh + TM * FS;
        const int v = tid & 63, fq = __builtin_amdgcn_readfirstlane(tid >> 6);
        const float c0 = xc[3 * v], c1 = xc[3 * v + 1], c2 = xc[3 * v + 2];
        const float *Bp = a.B + 3 * (chunk * HR + fq * FR);
        _Float16 sh[FR], sl[FR], ch[FR], cl[FR];
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const float arg = fmaf(c2, Bp[3 * j + 2], fmaf(c1, Bp[3 * j + 1], c0 * Bp[3 * j]));
            float sn, cs;
            sincos_f32(arg, sn, cs);
            split(sn, sh[j], sl[j]);
            split(cs, ch[j], cl[j]);
        }
        if constexpr (FR == 8) {
            h8_t p0, p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 8; ++j) { p0[j] = sh[j]; p1[j] = sl[j]; p2[j] = ch[j]; p3[j] = cl[j]; }
            *reinterpret_cast<h8_t *>(Fh + v * FS + fq * 8) = p0;
            *reinterpret_cast<h8_t *>(Fl + v * FS + fq * 8) = p1;
            *reinterpret_cast<h8_t *>(Fh + v * FS + HR + fq * 8) = p2;
            *reinterpret_cast<h8_t *>(Fl + v * FS + HR + fq * 8) = p3;
        } else {
            h4_t p0, p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) { p0[j] = sh[j]; p1[j] = sl[j]; p2[j] = ch[j]; p3[j] = cl[j]; }
            *reinterpret_cast<h4_t *>(Fh + v * FS + fq * 4) = p0;
            *reinterpret_cast<h4_t *>(Fl + v * FS + fq * 4) = p1;
            *reinterpret_cast<h4_t *>(Fh + v * FS + HR + fq * 4) = p2;
            *reinterpret_cast<h4_t *>(Fl + v * FS + HR + fq * 4) = p3;
        }
    };
    make_features(0, 0);
    __syncthreads();
    // weight fragments run PD k-steps ahead of their use in a register ring (an L2 hit is ~700 cycles away, a k-step of this wave 384
    // MFMA cycles, twice that with the other wave of the SIMD in between: one k-step of lookahead left the matrix pipe waiting)
    {
        const int nks1 = K1 / 16;
#pragma unroll
        for (int p = 0; p < PD; ++p)
            if (p < nks1) load_a(w.W1h, w.W1l, K1, p, ah[p], al[p]);
        for (int ch = 0; ch < nchunks; ++ch) {
            const int cur = ch & 1;
            const _Float16 *Fh = F + (2 * cur) * (TM * FS), *Fl = Fh + TM * FS;
            // waves 0-3 generate the next chunk before their MFMAs, waves 4-7 after: the two waves of a SIMD use the vector and
            // the matrix pipe at different times
            if (wave < 4 && ch + 1 < nchunks) make_features(ch + 1, 1 - cur);
#pragma unroll
            for (int q = 0; q < KC / 16; ++q) {
                const int ks = ch * (KC / 16) + q;
                kstep(ah[q % PD], al[q % PD], Fh, Fl, FS, q * 16);
                if (ks + PD < nks1) load_a(w.W1h, w.W1l, K1, ks + PD, ah[q % PD], al[q % PD]);
            }
            if (wave >= 4 && ch + 1 < nchunks) make_features(ch + 1, 1 - cur);
            __syncthreads();
        }
    }
    }   // (first layer)

    // ---- epilogue of a layer: bias + ReLU, split into the two activation images --------------------------------------------
    auto store_layer = [&](const float *bias) {
        float big = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!FULL && !on[t]) continue;
            const int tile = wave + 8 * t;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tile * 32 + 8 * g + 4 * h;     // rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                    h4_t oh, ol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float vv = fmaf(accx[t][c][4 * g + q], LO_INV, acch[t][c][4 * g + q]) + bias[n + q];
                        const float y = vv > 0.f ? vv : 0.f;
                        big = fmaxf(big, y);
                        _Float16 yh, yl;
                        split(y, yh, yl);
                        oh[q] = yh; ol[q] = yl;
                    }
                    *reinterpret_cast<h4_t *>(Hh + (c * 32 + r) * HS + n) = oh;
                    *reinterpret_cast<h4_t *>(Hl + (c * 32 + r) * HS + n) = ol;
                }
        }
        zero_acc();
        if (!(big < 65000.f) && a.range_flag) *a.range_flag = 1;      // (also catches NaN; plain store: every writer writes 1)
    };
    // training: the finished layer's two [voxel][k] images to HBM in 16-byte pieces, the low halves unscaled (h = hi + lo)
    auto save_layer = [&](int layer) {
        if (!a.save_act || (a.save_first_only && layer > 0)) return;
        _Float16 *dh = reinterpret_cast<_Float16 *>(a.save_act) + ((long long) layer * a.act_rows + v0) * a.nn;
        _Float16 *dl = reinterpret_cast<_Float16 *>(a.save_act_lo) + ((long long) layer * a.act_rows + v0) * a.nn;
        const int ppr = a.nn / 8;
        for (int q = tid; q < TM * ppr; q += 512) {
            const int v = q / ppr, c = q - v * ppr;
            if (v0 + v < a.nvox) {
                *reinterpret_cast<h8_t *>(dh + (long long) v * a.nn + 8 * c) = *reinterpret_cast<const h8_t *>(Hh + v * HS + 8 * c);
                h8_t l = *reinterpret_cast<const h8_t *>(Hl + v * HS + 8 * c);
                // (the kept first layer stays SCALED, the exact image: the unscaled half can be subnormal, and a forward pass that resumed
                // from it would differ from the recomputed one in the last bits -- enough to flip ReLU masks)
                if (!a.save_first_only) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) l[j] = unscale_lo(l[j]);
                }
                *reinterpret_cast<h8_t *>(dl + (long long) v * a.nn + 8 * c) = l;
            }
        }
    };
    if (a.h0_hi == nullptr) {
        store_layer(a.bias);      // all waves passed the last barrier of the chunk loop: the feature buffers are dead
        __syncthreads();
        save_layer(0);
    } else {
        // training, first-layer activations kept by the forward pass of this step (vfem_mlp: h0): two thirds of the forward's products
        // are not recomputed -- the images are loaded exactly as that forward left them
        const _Float16 *sh = reinterpret_cast<const _Float16 *>(a.h0_hi), *sl = reinterpret_cast<const _Float16 *>(a.h0_lo);
        const int ppr = a.nn / 8;
        for (int q = tid; q < TM * ppr; q += 512) {
            const int v = q / ppr, c = q - v * ppr;
            h8_t xh, xl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { xh[j] = (_Float16) 0.f; xl[j] = (_Float16) 0.f; }
            if (v0 + v < a.nvox) {
                xh = *reinterpret_cast<const h8_t *>(sh + (v0 + v) * a.nn + 8 * c);
                xl = *reinterpret_cast<const h8_t *>(sl + (v0 + v) * a.nn + 8 * c);
            }
            *reinterpret_cast<h8_t *>(Hh + v * HS + 8 * c) = xh;
            *reinterpret_cast<h8_t *>(Hl + v * HS + 8 * c) = xl;
        }
        __syncthreads();
    }

    // ---- hidden layers ------------------------------------------------------------------------------------------------------
    for (int l = 0; l < a.n_hidden; ++l) {
        const _Float16 *Wlh = w.Whh + (long long) l * a.nn * a.nn, *Wll = w.Whl + (long long) l * a.nn * a.nn;
        const int nks = a.nn / 16;
#pragma unroll
        for (int p = 0; p < PD; ++p)
            if (p < nks) load_a(Wlh, Wll, a.nn, p, ah[p], al[p]);
        for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ks = ks0 + q;
                if (ks < nks) {
                    kstep(ah[q], al[q], Hh, Hl, HS, ks * 16);
                    if (ks + PD < nks) load_a(Wlh, Wll, a.nn, ks + PD, ah[q], al[q]);
                }
            }
        }
        __syncthreads();          // every wave finished reading the images
        store_layer(a.bias + (l + 1) * a.nn);
        __syncthreads();
        save_layer(l + 1);
    }

    // ---- output layer: one scalar per voxel, 8 threads per voxel -----------------------------------------------------------
    {
        const int v = tid >> 3, part = tid & 7;
        const int kper = a.nn / 8;
        float s = 0.f;
        for (int k = part * kper; k < (part + 1) * kper; ++k)
            s = fmaf(fmaf((float) Hl[v * HS + k], LO_INV, (float) Hh[v * HS + k]), a.wout[k], s);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (part == 0 && v0 + v < a.nvox) {
            float o = s + a.bout;
            if (a.sigmoid) o = 1.f / (1.f + expf(-o));
            if (a.out32) a.out32[v0 + v] = o;
            if (a.out64) a.out64[v0 + v] = (double) o;
        }
    }
}

void launch_mlp_forward_x3(const MlpArgs &a, const void *W1h, const void *W1l, const void *Whh, const void *Whl, hipStream_t s, int kc) {
    using namespace x3;
    if (a.nn % 32 || a.nn > MAXN || a.es % 32) throw Error("fused MLP kernel: hidden width must be a multiple of 32 up to 512, embedding size a multiple of 32");
    if ((kc != 64 && kc != 128) || a.es % (kc / 2)) throw Error("fused MLP kernel: feature chunk 64 or 128, embedding size a multiple of half of it");
    const size_t lds = (size_t) 2 * TM * HS * 2 + TM * 3 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward_x3<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward_x3<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward_x3<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_forward_x3<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        attr_set = true;
    }
    MlpX3Weights w{(const _Float16 *) W1h, (const _Float16 *) W1l, (const _Float16 *) Whh, (const _Float16 *) Whl};
    const long long blocks = (a.nvox + TM - 1) / TM;
    const dim3 grd((unsigned) blocks), blk(512);
    if (kc == 128) {
        if (a.nn == MAXN) k_mlp_forward_x3<true, 128><<<grd, blk, lds, s>>>(a, w);
        else              k_mlp_forward_x3<false, 128><<<grd, blk, lds, s>>>(a, w);
    } else {
        if (a.nn == MAXN) k_mlp_forward_x3<true, 64><<<grd, blk, lds, s>>>(a, w);
        else              k_mlp_forward_x3<false, 64><<<grd, blk, lds, s>>>(a, w);
    }
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
