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
    long long act_rows;
    int ablate;                             // timing ablations (wrong results): 1 no feature generation, 2 no hidden layers, 3 no layer-1 MFMAs
};

// backward pass of one voxel chunk (networks.MLP under torch.autograd in the reference, train_xdg.py:282-329)
struct MlpBwdArgs {
    int nn, n_hidden, sigmoid;
    const void *WhT;                        // [n_hidden][nn (k)][nn (n)] fp16: transposed hidden weights
    const float *wout;                      // [nn]
    const float *g;                         // [nvox] dL/d(out)
    const float *out32;                     // [nvox] forward outputs (sigmoid derivative)
    float scale;                            // loss scale applied to g before it enters fp16
    const void *act;                        // [1 + n_hidden][act_rows][nn] fp16
    void *dz;                               // [1 + n_hidden][act_rows][nn] fp16: scaled gradients wrt the pre-activations
    float *gs;                              // [act_rows] scaled dL/d(pre-sigmoid out), zero beyond nvox
    long long act_rows, nvox;
};

}  // namespace vfem
