// Source coordinate of bilinear up-sampling, shared by every kernel that interpolates so that all of
// them produce the same bits: torch's upsample_bilinear2d rule (scale * dst with aligned corners,
// otherwise scale * (dst + 0.5) - 0.5 clamped at 0), in fp32 with the same fused multiply-add.
#pragma once

namespace spice {

__device__ __forceinline__ float src_index(float scale, int dst, int align_corners) {
    if (align_corners) return scale * static_cast<float>(dst);
    const float s = __builtin_fmaf(scale, static_cast<float>(dst) + 0.5f, -0.5f);
    return s < 0.0f ? 0.0f : s;
}

}  // namespace spice
