// argument block of the fused MLP kernel (shared by capi.hip and kernels_mlp.hip)
#pragma once
namespace vfem {
struct MlpArgs {
    int es, nn, n_hidden;                   // embedding size, hidden width, number of nn x nn layers
    int sigmoid;
    const float *B;                         // [es][3]
    const void *W1;                         // [nn][2 es]
    const void *Wh;                         // [n_hidden][nn][nn]
    const float *bias;                      // [(1 + n_hidden)][nn]
    const float *wout;                      // [nn]
    float bout;
    // input: either an explicit coordinate list or a regular grid (utils.get_mgrid: linspace incl. both ends)
    const float *coords;                    // [nvox][3] or null
    int gn[3];
    float glo[3], gstep[3];
    long long nvox;
    float *out32;
    double *out64;
    // training: first voxel of this chunk (grid mode: coordinates are generated for v_offset + local index) and the
    // post-ReLU activations of every hidden layer, fp16 [1 + n_hidden][act_rows][nn], for the backward pass
    long long v_offset;
    void *save_act;
    void *save_act_lo;                      // reference-precision forward only: the low halves, UNSCALED (h = hi + lo), same layout
    long long act_rows;
    int ablate;                             // timing ablations (wrong results): 1 no feature generation, 2 no hidden layers, 3 no layer-1 MFMAs
    const void *h0_hi, *h0_lo;              // reference-precision forward, training: first-layer activations kept by an earlier forward
                                            // ([nvox][nn] fp16 pairs, h = hi + lo 2^-11, indexed from the chunk's first voxel): the first layer is skipped
    int save_first_only;                    // save_act / save_act_lo receive the first layer's activations only (the forward that keeps them)
    int *range_flag;                        // reference-precision forward: set to 1 when a hidden activation leaves fp16's range (its high half would be inf)
};

// backward pass of one voxel chunk (networks.MLP under torch.autograd in the reference, train_xdg.py:282-329), split operands:
// every saved quantity q is a pair of fp16 arrays (hi, lo) with q = hi + lo (lo unscaled), [layer][act_rows][nn]
struct MlpBwdArgs {
    int nn, n_hidden, sigmoid;
    const void *WhTh, *WhTl;                // transposed hidden weights, split, MFMA-fragment order (lo scaled by 2^11 as the forward's)
    const float *wout;                      // [nn]
    const float *g;                         // [nvox] dL/d(out)
    const float *out32;                     // [nvox] forward outputs (sigmoid derivative)
    float scale;                            // loss scale applied to g
    const void *act_hi, *act_lo;            // [1 + n_hidden][act_rows][nn]: post-ReLU activations saved by the forward kernel
    const void *act0_hi, *act0_lo;          // non-null: layer 0 lives here instead ([act_rows][nn], the kept first-layer activations)
    void *dz_hi, *dz_lo;                    // [1 + n_hidden][act_rows][nn]: scaled gradients wrt the pre-activations
    float *gs;                              // [act_rows] scaled dL/d(pre-sigmoid out), zero beyond nvox
    long long act_rows, nvox;
};

// weight gradient of one layer over one voxel chunk: dW[n][k] = sum_v dz[v][n] h[v][k]   (h: saved activations, or -- first layer -- the
// Fourier features, regenerated from the coordinates)
struct MlpDwArgs {
    int nn, K;                              // output rows (hidden width), columns (nn, or 2 es for the first layer)
    const void *dz_hi, *dz_lo;              // [rows][nn]
    const void *h_hi, *h_lo;                // [rows][K], null for the first layer
    int h_lo_scaled;                        // the low halves of h carry the forward kernel's 2^11 scale (the kept first-layer activations)
    MlpArgs grid;                           // first layer: coordinates / grid of the chunk and B (es, B, coords, gn, glo, gstep, v_offset, nvox)
    long long rows;                         // voxels of the chunk, padded to a multiple of 32 x slices (rows beyond nvox hold dz = 0)
    int slices;                             // voxel slices: one block per (slice, output tile), partial sums [slice][nn][K]
    int terms;                              // 3: hi hi + hi lo + lo hi (reference precision);  1: hi hi only
    float *partial;
    float *colsum_partial;                  // [slice][nn] column sums of dz (the layer's bias gradient), or null
};

}  // namespace vfem
