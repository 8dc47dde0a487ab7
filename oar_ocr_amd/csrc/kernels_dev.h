// kernels_dev.h -- device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace oar {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float apply_act(float v, int kind, float alpha, float beta) {
    switch (kind) {
        case ACT_NONE: return v;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_HSWISH: {
            float t = fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.f), 1.f);
            return v * t;
        }
        case ACT_HSIGMOID: return fminf(fmaxf(v * alpha + beta, 0.f), 1.f);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_SWISH: return v * (1.0f / (1.0f + expf(-v)));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_CLIP: return fminf(fmaxf(v, alpha), beta);
        case ACT_TANH: return tanhf(v);
        case ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

// stand-alone element-wise kernel only: the fused-epilogue set plus the math ops of decomposed GELU / LayerNorm exports
__device__ __forceinline__ float apply_unary(float v, int kind, float alpha, float beta) {
    switch (kind) {
        case ACT_ERF: return erff(v);
        case ACT_SQRT: return sqrtf(v);
        case ACT_EXP: return expf(v);
        case ACT_ABS: return fabsf(v);
        case ACT_NEG: return -v;
        case ACT_RECIP: return 1.0f / v;
        case ACT_LOG: return logf(v);
        case ACT_GELU_TANH: return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
        case ACT_SOFTPLUS: return v > 20.0f ? v : log1pf(expf(v));
        case ACT_FLOOR: return floorf(v);
        case ACT_CEIL: return ceilf(v);
        case ACT_ROUND: return rintf(v);   // ONNX Round: half to even
        case ACT_NOT: return v != 0.0f ? 0.0f : 1.0f;
        default: return apply_act(v, kind, alpha, beta);
    }
}

}  // namespace k
}  // namespace oar
