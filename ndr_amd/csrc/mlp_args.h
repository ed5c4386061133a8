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
};

}  // namespace vfem
