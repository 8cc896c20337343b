// The three small elementwise entry points of the reference backend that lie outside the per-iteration hot path of the
// garden configuration but are exported by the package (FasterGSCudaBackend/__init__.py:14-18):
//   update_3d_filter      filter3d/src/filter3d.cu:9-38            Mip-Splatting 3D filter update for one view
//   relocation_adjustment densification/include/kernels_mcmc.cuh:28-59   3DGS-MCMC relocation, Eq. (9)
//   add_noise             densification/include/kernels_mcmc.cuh:69-127  3DGS-MCMC SGLD noise on the means
// One lane per point, grid-stride-free (N threads); all are streaming kernels bounded by HBM.
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

__global__ void __launch_bounds__(256) update_3d_filter_kernel(const float* __restrict__ positions, const float* __restrict__ w2c,
                                                               float* __restrict__ filter_3d, uint8_t* __restrict__ visibility_mask,
                                                               const int n, const float left, const float right, const float top,
                                                               const float bottom, const float near_plane, const float distance2filter) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = positions[3 * (size_t)i], py = positions[3 * (size_t)i + 1], pz = positions[3 * (size_t)i + 2];
    const float z = (w2c[8] * px + w2c[9] * py + w2c[10] * pz) + w2c[11];
    if (z < near_plane) return;
    const float xc = (w2c[0] * px + w2c[1] * py + w2c[2] * pz) + w2c[3];
    if (xc < left * z || xc > right * z) return;
    const float yc = (w2c[4] * px + w2c[5] * py + w2c[6] * pz) + w2c[7];
    if (yc < top * z || yc > bottom * z) return;
    const float f_new = distance2filter * z;
    if (filter_3d[i] < f_new) return;
    filter_3d[i] = f_new;
    visibility_mask[i] = 1;
}

constexpr int kMcmcMaxSamples = 50;                       // densification_config.h:10
struct RelocationTable { float c[kMcmcMaxSamples * kMcmcMaxSamples]; };   // 10 KB: lives in global memory, L1/K$-resident

__global__ void __launch_bounds__(256) relocation_kernel(const float* __restrict__ old_opacities, const float* __restrict__ old_scales,
                                                         const int64_t* __restrict__ n_samples_per_primitive, const float* __restrict__ table,
                                                         float* __restrict__ new_opacities, float* __restrict__ new_scales, const unsigned n) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float old_opacity = old_opacities[i];
    const int n_samples = min(max(static_cast<int>(n_samples_per_primitive[i]), 1), kMcmcMaxSamples);
    const float new_opacity = 1.0f - powf(1.0f - old_opacity, 1.0f / static_cast<float>(n_samples));
    new_opacities[i] = new_opacity;
    float denominator = 0.0f;
    for (int s = 0; s < n_samples; ++s) {
        float power = new_opacity;
        for (int k = 0; k <= s; ++k, power *= new_opacity) denominator += table[s * kMcmcMaxSamples + k] * power;
    }
    const float factor = old_opacity / denominator;
#pragma unroll
    for (int c = 0; c < 3; ++c) new_scales[3 * (size_t)i + c] = factor * old_scales[3 * (size_t)i + c];
}

__global__ void __launch_bounds__(256) add_noise_kernel(const float* __restrict__ raw_scales, const float* __restrict__ raw_rotations,
                                                        const float* __restrict__ raw_opacities, const float* __restrict__ random_samples,
                                                        float* __restrict__ means, const unsigned n, const float current_lr) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float var[3], R[9], norm_sq;
#pragma unroll
    for (int c = 0; c < 3; ++c) var[c] = expf(2.0f * raw_scales[3 * (size_t)i + c]);
    quat_to_rotation(raw_rotations[4 * (size_t)i], raw_rotations[4 * (size_t)i + 1], raw_rotations[4 * (size_t)i + 2],
                     raw_rotations[4 * (size_t)i + 3], R, norm_sq);
    if (norm_sq < 1e-8f) return;
    float G[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) G[3 * r + c] = R[3 * r + c] * var[c];
    const float c11 = G[0] * R[0] + G[1] * R[1] + G[2] * R[2], c12 = G[0] * R[3] + G[1] * R[4] + G[2] * R[5];
    const float c13 = G[0] * R[6] + G[1] * R[7] + G[2] * R[8], c22 = G[3] * R[3] + G[4] * R[4] + G[5] * R[5];
    const float c23 = G[3] * R[6] + G[4] * R[7] + G[5] * R[8], c33 = G[6] * R[6] + G[7] * R[7] + G[8] * R[8];
    const float nx = random_samples[3 * (size_t)i], ny = random_samples[3 * (size_t)i + 1], nz = random_samples[3 * (size_t)i + 2];
    const float tx = c11 * nx + c12 * ny + c13 * nz, ty = c12 * nx + c22 * ny + c23 * nz, tz = c13 * nx + c23 * ny + c33 * nz;
    const float opacity = 1.0f / (1.0f + expf(-raw_opacities[i]));
    const float factor = current_lr * (1.0f / (1.0f + expf(100.0f * opacity - 0.5f)));
    means[3 * (size_t)i] += factor * tx; means[3 * (size_t)i + 1] += factor * ty; means[3 * (size_t)i + 2] += factor * tz;
}

hipError_t launch_update_3d_filter(const float* positions, const float* w2c, float* filter_3d, uint8_t* visibility_mask, int n,
                                   float left, float right, float top, float bottom, float near_plane, float distance2filter, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(update_3d_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, positions, w2c, filter_3d, visibility_mask, n, left,
                       right, top, bottom, near_plane, distance2filter);
    return hipGetLastError();
}

void relocation_coefficients(float* out /*[50*50]*/) {   // kernels_mcmc.cuh:13-26 (Eq. 9 of the 3DGS-MCMC paper)
    for (int i = 0; i < kMcmcMaxSamples * kMcmcMaxSamples; ++i) out[i] = 0.0f;
    for (int n = 0; n < kMcmcMaxSamples; ++n) {
        double binom = 1.0, sign = 1.0;
        for (int k = 0; k <= n; ++k, sign = -sign) {
            out[n * kMcmcMaxSamples + k] = static_cast<float>(binom * sign * (1.0 / std::sqrt(static_cast<double>(k + 1))));
            binom *= static_cast<double>(n - k) / static_cast<double>(k + 1);
        }
    }
}

hipError_t launch_relocation(const float* old_opacities, const float* old_scales, const int64_t* n_samples, const float* table_device,
                             float* new_opacities, float* new_scales, unsigned n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(relocation_kernel, dim3((n + 255) / 256), dim3(256), 0, s, old_opacities, old_scales, n_samples, table_device,
                       new_opacities, new_scales, n);
    return hipGetLastError();
}

hipError_t launch_add_noise(const float* raw_scales, const float* raw_rotations, const float* raw_opacities, const float* random_samples,
                            float* means, unsigned n, float current_lr, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(add_noise_kernel, dim3((n + 255) / 256), dim3(256), 0, s, raw_scales, raw_rotations, raw_opacities, random_samples,
                       means, n, current_lr);
    return hipGetLastError();
}

}  // namespace fgs
