// Backward pass of the Fourier-feature MLP at the reference's precision (the reference differentiates networks.MLP in fp32 through
// torch.autograd, train_xdg.py:282-329, fem.py:125), on the f16 matrix pipe with split operands, no library GEMM.
//
// Per voxel chunk the host (capi.hip: mlp_backward_impl) runs
//   1. the reference-precision forward kernel (kernels_mlp_x3.hip) with `save_act`: post-ReLU activations of every hidden layer as
//      two fp16 arrays (hi, lo), h = hi + lo, row-major [voxel][k];
//   2. k_mlp_backward_x3: dL/d(out) -> gradients wrt the pre-activations of every layer, dz (hi, lo), loss-scaled; the products
//      dh[v][k] = sum_n Wh[n][k] dz[v][n] are the forward kernel's hidden-layer GEMM with the transposed weights (three MFMA
//      products per product, two accumulators), ReLU masks from the saved activations;
//   3. k_mlp_dw: the weight gradients dW[n][k] = sum_v dz[v][n] h[v][k], a GEMM whose reduction runs over the VOXELS: both operands
//      are stored voxel-major, i.e. with the reduction index as the slow one.  Tiles of 32 voxels are staged in LDS as they lie in
//      memory and read back column-wise by gfx950's transposing LDS read (ds_read_b64_tr_b16): a lane receives four consecutive
//      voxels of one column, which is the MFMA operand layout -- no shuffles, no second copy.  For the first layer the second
//      operand is never read: the block regenerates the Fourier features of its 32 voxels x 256 columns into the same LDS image
//      (the reference materialises them, 275 GB at 512 x 256 x 256; rounds 1-3 of this build materialised them per chunk).
//      Products: hi hi + hi lo + lo hi into ONE fp32 accumulator (the low halves are stored unscaled here: every operand is O(1)
//      or loss-scaled to it, so the low halves stay above fp16's subnormal step 6e-8 by ten bits where it matters).
//   4. column sums for the biases, the output layer's weight and bias.
#include "vfem_internal.h"

#include <hip/hip_fp16.h>

#include "mlp_args.h"

namespace vfem {

namespace bw {
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef __fp16 hf4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int TM = 64, MAXN = 512, HS = MAXN + 8;
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;
__device__ __forceinline__ void split_scaled(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16) x;
    lo = (_Float16) ((x - (float) hi) * LO_SCALE);
}
// (the pair of x3::unscale_lo in kernels_mlp_x3.hip: same value for the same half)
__device__ __forceinline__ _Float16 unscale_lo(_Float16 ls) {
    const float f = (float) ls;
    _Float16 u = (_Float16) (f * LO_INV);
    if ((float) u == 0.f && f != 0.f) u = (_Float16) (f > 0.f ? 5.9604645e-8f : -5.9604645e-8f);
    return u;
}
__device__ __forceinline__ void split_plain(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16) x;
    lo = (_Float16) (x - (float) hi);
}
// the pair of sincos_f32 in kernels_mlp_x3.hip (same constants, same order of operations: the features the weight gradient sees are
// bit for bit the ones the forward pass multiplied)
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
}  // namespace bw

// ---------------------------------------------------------------------------------------------------------------------------
// data path: 64 voxels per 512-thread block, the tiling and the k-step of k_mlp_forward_x3 (A = transposed-weight fragments in
// fragment order, B = the two images of dz in LDS)
// ---------------------------------------------------------------------------------------------------------------------------
template <bool FULL>                // FULL: hidden width 512 (see k_mlp_forward_x3)
__global__ void __launch_bounds__(512) k_mlp_backward_x3(MlpBwdArgs a) {
    using namespace bw;
    extern __shared__ __align__(16) unsigned char smem[];
    _Float16 *Hh = reinterpret_cast<_Float16 *>(smem);
    _Float16 *Hl = Hh + TM * HS;
    float *gsl = reinterpret_cast<float *>(smem + (size_t) 2 * TM * HS * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long v0 = (long long) blockIdx.x * TM;
    const int ntiles = a.nn / 32, ppr = a.nn / 8, top = a.n_hidden;
    const bool on[2] = {FULL || wave < ntiles, FULL || wave + 8 < ntiles};
    const _Float16 *acth = reinterpret_cast<const _Float16 *>(a.act_hi), *actl = reinterpret_cast<const _Float16 *>(a.act_lo);
    _Float16 *dzh = reinterpret_cast<_Float16 *>(a.dz_hi), *dzl = reinterpret_cast<_Float16 *>(a.dz_lo);

    if (tid < TM) {
        float g = 0.f;
        if (v0 + tid < a.nvox) {
            g = a.g[v0 + tid] * a.scale;
            if (a.sigmoid) { const float o = a.out32[v0 + tid]; g *= o * (1.f - o); }
        }
        gsl[tid] = g;
        a.gs[v0 + tid] = g;
    }
    __syncthreads();
    // the gradient of a layer's pre-activations from the raw gradient of its outputs in the LDS images (top layer: from gs wout):
    // ReLU mask from the saved activations, coalesced 16-byte pieces; result to the images (operand of the next product) and to HBM
    auto mask_pass = [&](int layer, bool is_top) {
        for (int q = tid; q < TM * ppr; q += 512) {
            const int v = q / ppr, c = q - v * ppr;
            h8_t oh, ol, ou;
#pragma unroll
            for (int j = 0; j < 8; ++j) { oh[j] = (_Float16) 0.f; ol[j] = (_Float16) 0.f; ou[j] = (_Float16) 0.f; }
            if (v0 + v < a.nvox) {
                const bool kept = layer == 0 && a.act0_hi != nullptr;      // layer 0 from the activations the forward pass kept
                const long long at = kept ? (v0 + v) * a.nn + 8 * c : ((long long) layer * a.act_rows + v0 + v) * a.nn + 8 * c;
                const _Float16 *ph = kept ? reinterpret_cast<const _Float16 *>(a.act0_hi) : acth, *pl = kept ? reinterpret_cast<const _Float16 *>(a.act0_lo) : actl;
                const h8_t hv = *reinterpret_cast<const h8_t *>(ph + at), lv = *reinterpret_cast<const h8_t *>(pl + at);
                h8_t rh, rl;
                if (!is_top) { rh = *reinterpret_cast<const h8_t *>(Hh + v * HS + 8 * c); rl = *reinterpret_cast<const h8_t *>(Hl + v * HS + 8 * c); }
                const float g = gsl[v];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool pos = (float) hv[j] > 0.f || (float) lv[j] > 0.f;
                    if (!pos) continue;
                    if (is_top) {
                        _Float16 xh, xl;
                        split_scaled(g * a.wout[8 * c + j], xh, xl);
                        oh[j] = xh; ol[j] = xl;
                    } else { oh[j] = rh[j]; ol[j] = rl[j]; }
                    ou[j] = (_Float16) ((float) ol[j] * LO_INV);
                }
            }
            *reinterpret_cast<h8_t *>(Hh + v * HS + 8 * c) = oh;
            *reinterpret_cast<h8_t *>(Hl + v * HS + 8 * c) = ol;
            const long long at = ((long long) layer * a.act_rows + v0 + v) * a.nn + 8 * c;
            *reinterpret_cast<h8_t *>(dzh + at) = oh;
            *reinterpret_cast<h8_t *>(dzl + at) = ou;
        }
    };
    mask_pass(top, true);
    __syncthreads();

    f16_t acch[2][2], accx[2][2];
    constexpr int PD = 4;
    h8_t ah[PD][2], al[PD][2];
    for (int l = a.n_hidden - 1; l >= 0; --l) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) { acch[t][c][q] = 0.f; accx[t][c][q] = 0.f; }
        const _Float16 *Wh_ = reinterpret_cast<const _Float16 *>(a.WhTh) + (long long) l * a.nn * a.nn;
        const _Float16 *Wl_ = reinterpret_cast<const _Float16 *>(a.WhTl) + (long long) l * a.nn * a.nn;
        const int nks = a.nn / 16;
        auto load_a = [&](int ks, h8_t (&fh)[2], h8_t (&fl)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const long long off = (((long long) ((FULL || on[t]) ? wave + 8 * t : 0) * nks + ks) * 64 + lane) * 8;
                fh[t] = *reinterpret_cast<const h8_t *>(Wh_ + off);
                fl[t] = *reinterpret_cast<const h8_t *>(Wl_ + off);
            }
        };
#pragma unroll
        for (int p = 0; p < PD; ++p)
            if (p < nks) load_a(p, ah[p], al[p]);
        for (int ks0 = 0; ks0 < nks; ks0 += PD) {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ks = ks0 + q;
                if (ks < nks) {
                    h8_t bh[2], bl[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bh[c] = *reinterpret_cast<const h8_t *>(Hh + (c * 32 + r) * HS + ks * 16 + 8 * h);
                        bl[c] = *reinterpret_cast<const h8_t *>(Hl + (c * 32 + r) * HS + ks * 16 + 8 * h);
                    }
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if (!FULL && !on[t]) continue;
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                if (term == 0) acch[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][t], bh[c], acch[t][c], 0, 0, 0);
                                if (term == 1) accx[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][t], bl[c], accx[t][c], 0, 0, 0);
                                if (term == 2) accx[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][t], bh[c], accx[t][c], 0, 0, 0);
                            }
                        }
                    if (ks + PD < nks) load_a(ks + PD, ah[q], al[q]);
                }
            }
        }
        __syncthreads();          // every wave finished reading the images
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!FULL && !on[t]) continue;
            const int tile = wave + 8 * t;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tile * 32 + 8 * g + 4 * h;
                    h4_t oh, ol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        _Float16 yh, yl;
                        split_scaled(fmaf(accx[t][c][4 * g + q], LO_INV, acch[t][c][4 * g + q]), yh, yl);
                        oh[q] = yh; ol[q] = yl;
                    }
                    *reinterpret_cast<h4_t *>(Hh + (c * 32 + r) * HS + n) = oh;
                    *reinterpret_cast<h4_t *>(Hl + (c * 32 + r) * HS + n) = ol;
                }
        }
        __syncthreads();
        mask_pass(l, false);
        __syncthreads();
    }
}

void launch_mlp_backward_x3(const MlpBwdArgs &a, long long rows, hipStream_t s) {
    using namespace bw;
    if (a.nn % 32 || a.nn > MAXN) throw Error("MLP backward: hidden width must be a multiple of 32 up to 512");
    if (rows % TM) throw Error("MLP backward: the padded chunk must be a multiple of 64 voxels");
    const size_t lds = (size_t) 2 * TM * HS * 2 + TM * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_backward_x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_backward_x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        attr_set = true;
    }
    if (a.nn == MAXN) k_mlp_backward_x3<true><<<dim3((unsigned) (rows / TM)), dim3(512), lds, s>>>(a);
    else              k_mlp_backward_x3<false><<<dim3((unsigned) (rows / TM)), dim3(512), lds, s>>>(a);
    VFEM_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradients: one block = (voxel slice, BN x BK output tile), 8 waves, a wave owns 4 x 2 tiles of 32 x 32 = 128 accumulator
// registers.  The tile shape is a template parameter; both layers' kinds run 256 x 256 with stages of 32 voxels (waves 2 x 4).  A
// 512 x 128 shape with stages of 16 voxels (waves 4 x 2: all rows in one block, so the first layer's Fourier features are regenerated
// once per column tile and not once per row tile as well -- its kernel is as much vector as matrix work, 1 810 vector instructions per
// voxel, profiles/r04_mlp_pipeline_pmc.json) was built and measured: correct, and 8 % SLOWER per backward pass (0.68 against 0.63 s
// from the kept first layer; twice the barriers per voxel and twice the dz bytes through L2 outweigh the halved feature work).
// Block ids are dealt to the XCDs round-robin, so the numbering puts all tiles of a
// slice on ONE XCD: the slice's operands leave HBM once and the other tiles find them in that XCD's L2.
// ---------------------------------------------------------------------------------------------------------------------------
namespace dw {
template <int BN_, int BK_, int SV_>
struct Shape {
    static constexpr int BN = BN_, BK = BK_, SV = SV_;       // output tile rows / columns, voxels per stage
    static constexpr int WN = BN / 128, WK = 8 / WN;           // waves along the rows / columns (a wave: 128 rows x 64 columns)
    static_assert(WN * WK == 8 && BK == 64 * WK, "eight waves of 4 x 2 tiles");
    static_assert(SV % 16 == 0 && SV * BN / 8 == 1024 && ((SV * BK / 8) % 512 == 0 || SV * BK / 8 == 256), "staging: two A pieces per thread");
    static constexpr int PA = BN * 2 + 64, PB = BK * 2 + 64;  // bytes per staged row: + 64 puts the four rows of a transposing read on distinct banks
    static constexpr int IA = SV * PA, IB = SV * PB;          // one image (one operand half, one stage)
    static constexpr int STAGE = 2 * IA + 2 * IB;             // A hi, A lo, B hi, B lo
    static constexpr size_t LDS_BYTES = (size_t) 2 * STAGE;
    static_assert(LDS_BYTES <= 160 * 1024, "two stages must fit the LDS of a CU");
};
using Hidden = Shape<256, 256, 32>;
using First = Shape<256, 256, 32>;      // (Shape<512, 128, 16>: measured slower, see above)
}  // namespace dw

template <int TERMS, bool FEATURES, class C>
__global__ void __launch_bounds__(512) k_mlp_dw(MlpDwArgs a) {
    using namespace bw;
    constexpr int BN = C::BN, BK = C::BK, SV = C::SV, PA = C::PA, PB = C::PB, IA = C::IA, IB = C::IB, STAGE = C::STAGE;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % C::WN, wk = wave / C::WN;
    const int tiles_n = (a.nn + BN - 1) / BN, tiles_k = (a.K + BK - 1) / BK, ntile = tiles_n * tiles_k;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slice = (idx / ntile) * 8 + xcd, tile = idx % ntile;
    if (slice >= a.slices) return;
    const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
    const long long per_slice = a.rows / a.slices;
    const long long vbase = (long long) slice * per_slice;
    const int nstages = (int) (per_slice / SV);
    const _Float16 *Ah = reinterpret_cast<const _Float16 *>(a.dz_hi), *Al = reinterpret_cast<const _Float16 *>(a.dz_lo);
    const _Float16 *Bh = reinterpret_cast<const _Float16 *>(a.h_hi), *Bl = reinterpret_cast<const _Float16 *>(a.h_lo);

    // staging: an A image is SV rows x BN / 8 pieces of 16 bytes = 1024 pieces, thread -> pieces tid and tid + 512; a B image likewise
    constexpr int APR = BN / 8, BPR = BK / 8;                 // pieces per row
    constexpr int NB = (SV * BPR + 511) / 512;                // B pieces per thread and image (hidden layers: 2)
    int arow[2], acol[2], brow[NB], bcol[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) { arow[i] = (tid + 512 * i) / APR; acol[i] = (tid + 512 * i) % APR; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { brow[i] = (tid + 512 * i) / BPR; bcol[i] = (tid + 512 * i) % BPR; }
    h8_t ra[2][2], rb[2][NB];                                 // [hi / lo][piece]
    auto zero8 = [](h8_t &x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (_Float16) 0.f;
    };
    auto fetch = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long v = vbase + (long long) st * SV + arow[i];
            if (n0 + 8 * acol[i] < a.nn) {                    // (widths are multiples of 8)
                ra[0][i] = *reinterpret_cast<const h8_t *>(Ah + v * a.nn + n0 + 8 * acol[i]);
                if (TERMS == 3) ra[1][i] = *reinterpret_cast<const h8_t *>(Al + v * a.nn + n0 + 8 * acol[i]);
            } else { zero8(ra[0][i]); zero8(ra[1][i]); }
        }
        if (!FEATURES) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const long long v = vbase + (long long) st * SV + brow[i];
                if (brow[i] < SV && k0 + 8 * bcol[i] < a.K) {
                    rb[0][i] = *reinterpret_cast<const h8_t *>(Bh + v * a.K + k0 + 8 * bcol[i]);
                    if (TERMS == 3) {
                        rb[1][i] = *reinterpret_cast<const h8_t *>(Bl + v * a.K + k0 + 8 * bcol[i]);
                        if (a.h_lo_scaled) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) rb[1][i][j] = unscale_lo(rb[1][i][j]);
                        }
                    }
                } else { zero8(rb[0][i]); zero8(rb[1][i]); }
            }
        }
    };
    auto commit = [&](int buf) {
        unsigned char *base = smem + (size_t) buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = arow[i] * PA + 16 * acol[i];
            *reinterpret_cast<h8_t *>(base + off) = ra[0][i];
            if (TERMS == 3) *reinterpret_cast<h8_t *>(base + IA + off) = ra[1][i];
        }
        if (!FEATURES) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (brow[i] >= SV) continue;
                const int off = brow[i] * PB + 16 * bcol[i];
                *reinterpret_cast<h8_t *>(base + 2 * IA + off) = rb[0][i];
                if (TERMS == 3) *reinterpret_cast<h8_t *>(base + 2 * IA + IB + off) = rb[1][i];
            }
        }
    };
    // first layer: the block's BK columns are BK / 64 chunks of the forward kernel's K order (the sines of 32 rows of B, then their
    // cosines): SV voxels x BK / 2 arguments per stage, thread -> voxel tid % SV, FR consecutive rows of B
    constexpr int FG = 512 / SV, FR = (BK / 2) / FG;          // row groups, rows of B per thread (hidden shape: 16 x 8; first: 32 x 2)
    static_assert(FR >= 1 && 32 % FR == 0, "a thread's rows of B lie in one chunk");
    auto features = [&](int st, int buf) {
        unsigned char *base = smem + (size_t) buf * STAGE + 2 * IA;
        const int v = tid % SV, rg = tid / SV;
        const int chunk = (rg * FR) >> 5, ro = (rg * FR) & 31;
        const int brow0 = ((k0 >> 6) + chunk) * 32 + ro;                             // first row of B
        float x[3] = {0.f, 0.f, 0.f};
        const long long vv = vbase + (long long) st * SV + v;
        if (vv < a.grid.nvox) voxel_xyz(a.grid, vv, x);           // (padded rows carry dz = 0; their features only have to be finite)
        const float twopi = 6.283185307179586f;
        const float c0 = twopi * x[0], c1 = twopi * x[1], c2 = twopi * x[2];
        _Float16 sh[FR], sl[FR], ch[FR], cl[FR];
        const bool in = brow0 < a.grid.es;
        const float *Bp = a.grid.B + 3 * (in ? brow0 : 0);
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const float arg = fmaf(c2, Bp[3 * j + 2], fmaf(c1, Bp[3 * j + 1], c0 * Bp[3 * j]));
            float sn, cs;
            sincos_f32(arg, sn, cs);
            if (!in) { sn = 0.f; cs = 0.f; }
            split_plain(sn, sh[j], sl[j]);
            split_plain(cs, ch[j], cl[j]);
        }
        const int off = v * PB + 2 * (64 * chunk + ro);
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            *reinterpret_cast<_Float16 *>(base + off + 2 * j) = sh[j];
            *reinterpret_cast<_Float16 *>(base + off + 64 + 2 * j) = ch[j];
            if (TERMS == 3) {
                *reinterpret_cast<_Float16 *>(base + IB + off + 2 * j) = sl[j];
                *reinterpret_cast<_Float16 *>(base + IB + off + 64 + 2 * j) = cl[j];
            }
        }
    };

    // bias gradient of the layer = column sums of dz, the A operand: the blocks of the first column tile add up the pieces they stage
    // anyway (eight columns x two rows per thread and stage), no pass of its own over dz
    const bool colsum = a.colsum_partial != nullptr && k0 == 0;
    float cs[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[i][j] = 0.f;
    auto add_colsum = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[i][j] += (float) ra[0][i][j] + (TERMS == 3 ? (float) ra[1][i][j] : 0.f);
    };

    f16_t acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

    // transposing read of one operand fragment: tile column block `col0` (32 columns), k-step ks of the stage.  Lane (group g = lane >> 4,
    // i = lane & 15): the group's block is rows 16 ks + 8 (g >> 1) + 4 t .. + 3, columns col0 + 16 (g & 1) .. + 15; the lane SUPPLIES the
    // address of row (i >> 2), columns 4 (i & 3) .. + 3 of the block and RECEIVES column i of its four rows -- column col0 + (lane & 31),
    // voxels 8 (lane >> 5) + 4 t + 0..3 of the k-step: the MFMA operand layout.
    const int g4 = lane >> 4, i16 = lane & 15;
    auto frag = [&](const unsigned char *img, int pitch, int col0, int ks) {
        union { hf4_t q[2]; h8_t v; } u;
        const unsigned char *p = img + (16 * ks + 8 * (g4 >> 1) + (i16 >> 2)) * pitch + 2 * (col0 + 16 * (g4 & 1) + 4 * (i16 & 3));
        u.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hf4_t *) (p));
        u.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hf4_t *) (p + 4 * pitch));
        return u.v;
    };

    if (nstages > 0) {
        fetch(0);
        commit(0);
        if (colsum) add_colsum();
        if (FEATURES) features(0, 0);
    }
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const unsigned char *base = smem + (size_t) buf * STAGE;
        if (st + 1 < nstages) fetch(st + 1);
        // waves 0-3 generate the next stage's features before their products, waves 4-7 after: the two waves of a SIMD are on the
        // vector and on the matrix pipe at different times
        if (FEATURES && wave < 4 && st + 1 < nstages) features(st + 1, 1 - buf);
#pragma unroll
        for (int ks = 0; ks < SV / 16; ++ks) {
            h8_t fa[4], fal[4], fb[2], fbl[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = frag(base, PA, wn * 128 + 32 * t, ks);
                if (TERMS == 3) fal[t] = frag(base + IA, PA, wn * 128 + 32 * t, ks);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                fb[c] = frag(base + 2 * IA, PB, wk * 64 + 32 * c, ks);
                if (TERMS == 3) fbl[c] = frag(base + 2 * IA + IB, PB, wk * 64 + 32 * c, ks);
            }
            // term by term over the eight tiles: products into the same accumulator are eight MFMAs apart, not back to back
#pragma unroll
            for (int term = 0; term < TERMS; ++term)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? fal[t] : fa[t], term == 1 ? fbl[c] : fb[c], acc[t][c], 0, 0, 0);
        }
        if (FEATURES && wave >= 4 && st + 1 < nstages) features(st + 1, 1 - buf);
        if (st + 1 < nstages) {
            commit(1 - buf);
            if (colsum) add_colsum();
        }
        __syncthreads();
    }
    if (colsum) {
        // the threads that share a column piece (rows arow) meet in LDS (the staging buffers are dead), fixed order
        float *red = reinterpret_cast<float *>(smem);          // [SV rows][BN columns]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[arow[i] * BN + 8 * acol[i] + j] = cs[i][j];
        __syncthreads();
        for (int c = tid; c < BN; c += 512) {
            float sum = 0.f;
            for (int q = 0; q < SV; ++q) sum += red[q * BN + c];
            if (n0 + c < a.nn) a.colsum_partial[(long long) slice * a.nn + n0 + c] = sum;
        }
    }

    // partial[slice][n][k]: column (lane & 31) of a tile, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); first layer: back from the
    // forward kernel's K order to the feature index (sines 0 .. es-1, cosines es .. 2 es-1)
    float *out = a.partial + (long long) slice * a.nn * a.K;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int k = k0 + wk * 64 + 32 * c + r;
        if (k >= a.K) continue;
        if (FEATURES) { const int kk = k & 63; k = (kk < 32 ? 0 : a.grid.es) + 32 * (k >> 6) + (kk & 31); }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = n0 + wn * 128 + 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (n < a.nn) out[(long long) n * a.K + k] = acc[t][c][q];
            }
    }
}

void launch_mlp_dw(const MlpDwArgs &a, hipStream_t s) {
    using namespace dw;
    if (a.slices % 8 || a.slices <= 0) throw Error("MLP weight gradient: the number of voxel slices must be a positive multiple of 8");
    if (a.rows % ((long long) a.slices * 32)) throw Error("MLP weight gradient: chunk rows must be a multiple of 32 x slices");
    if (a.nn % 8 || a.K % 8) throw Error("MLP weight gradient: widths must be multiples of 8");
    const bool feat = a.h_hi == nullptr;
    if (feat && (a.K != 2 * a.grid.es || a.grid.es % 32)) throw Error("MLP weight gradient: first layer needs K = 2 es, es % 32 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_dw<3, false, Hidden>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Hidden::LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_dw<1, false, Hidden>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Hidden::LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_dw<3, true, First>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) First::LDS_BYTES));
        VFEM_HIP(hipFuncSetAttribute((const void *) k_mlp_dw<1, true, First>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) First::LDS_BYTES));
        attr_set = true;
    }
    auto blocks_of = [&](int BN, int BK) { return (unsigned) ((a.slices / 8) * ((a.nn + BN - 1) / BN) * ((a.K + BK - 1) / BK) * 8); };
    if (feat) {
        const unsigned blocks = blocks_of(First::BN, First::BK);
        if (a.terms == 3) k_mlp_dw<3, true, First><<<dim3(blocks), dim3(512), First::LDS_BYTES, s>>>(a);
        else              k_mlp_dw<1, true, First><<<dim3(blocks), dim3(512), First::LDS_BYTES, s>>>(a);
    } else {
        const unsigned blocks = blocks_of(Hidden::BN, Hidden::BK);
        if (a.terms == 3) k_mlp_dw<3, false, Hidden><<<dim3(blocks), dim3(512), Hidden::LDS_BYTES, s>>>(a);
        else              k_mlp_dw<1, false, Hidden><<<dim3(blocks), dim3(512), Hidden::LDS_BYTES, s>>>(a);
    }
    VFEM_HIP(hipGetLastError());
}

// column sums of a split fp16 [rows][ncols] matrix (hi + lo), optionally weighted per row: partial[blk][col], 512 rows per block.
// 256 threads: a thread owns eight consecutive columns (16-byte loads) and every (256 / (ncols / 8))-th row of the block's rows
__global__ void __launch_bounds__(256) k_colsum_split(long long rows, int ncols, const _Float16 *__restrict__ Xh, const _Float16 *__restrict__ Xl,
                                                      const float *__restrict__ w, float *__restrict__ partial) {
    using namespace bw;
    __shared__ float red[256 * 8];
    const int groups = ncols / 8;                       // column groups (<= 64)
    const int lanes = 256 / groups;                     // row lanes per group
    const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
    const long long r0 = (long long) blockIdx.x * 512, r1 = r0 + 512 < rows ? r0 + 512 : rows;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < lanes)
        for (long long r = r0 + rl; r < r1; r += lanes) {
            const h8_t h = *reinterpret_cast<const h8_t *>(Xh + r * ncols + 8 * cg), l = *reinterpret_cast<const h8_t *>(Xl + r * ncols + 8 * cg);
            const float wr = w ? w[r] : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(wr, (float) h[j] + (float) l[j], acc[j]);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < ncols; c += 256) {
        float sum = 0.f;
        for (int q = 0; q < lanes; ++q) sum += red[(q * groups + (c >> 3)) * 8 + (c & 7)];
        partial[(long long) blockIdx.x * ncols + c] = sum;
    }
}
void launch_colsum_split(long long rows, int ncols, const void *Xh, const void *Xl, const float *w, float *partial, hipStream_t s) {
    if (ncols % 8 || ncols > 512) throw Error("column sums: width must be a multiple of 8, at most 512");
    k_colsum_split<<<dim3((unsigned) ((rows + 511) / 512)), dim3(256), 0, s>>>(rows, ncols, (const _Float16 *) Xh, (const _Float16 *) Xl, w, partial);
    VFEM_HIP(hipGetLastError());
}

}  // namespace vfem
