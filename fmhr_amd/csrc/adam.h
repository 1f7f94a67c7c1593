// fmhr_amd/csrc/adam.h — torch.optim.Adam's update rule (amsgrad=False, weight_decay=0, maximize=False: mesh_sfs_optim.py:193,244,310)
// as one device statement sequence shared by every kernel that applies it: k_adam / k_adam2 (mesh.hip), the fused forms in
// k_mesh_tail_b / k_ham_combine (mesh.hip) and in the stage-1 backward (stage1.hip) — so all of them agree bit for bit.
#pragma once
#include "common.h"

namespace fmhr {

struct AdamTensor {
    float* p; const float* g; float* m; float* v; long long n;
    float beta1, beta2, eps, step_size, bc2_sqrt;
};

__device__ __forceinline__ void adam_element(const AdamTensor& T, long long i, float gi)
{
    float mi = T.m[i];
    mi = mi + (gi - mi) * (1.0f - T.beta1);                      // exp_avg.lerp_(grad, 1 - beta1)
    float vi = T.v[i] * T.beta2 + (1.0f - T.beta2) * gi * gi;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    T.m[i] = mi; T.v[i] = vi;
    float denom = sqrtf(vi) / T.bc2_sqrt + T.eps;
    T.p[i] = T.p[i] - T.step_size * (mi / denom);
}

// device-side form of an optional fused optimiser update (a == null: switched off, T.p == null); mesh.hip
int adam_tensor_from(AdamTensor& T, const fmhr_adam_param* a, long long n);

}  // namespace fmhr
