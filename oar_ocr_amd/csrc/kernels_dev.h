// kernels_dev.h -- device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace oar {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ activations
// erf for the fused GELU epilogues (SVTRv2: 2.7 G activated elements per 256-crop batch behind the MLPs' first Linear; libm's erff is ~50 instructions with a
// divergent branch at |x| = 1): the rational form x P(x^2) / Q(x^2) on [-4, 4] (degree 6 / 4 in x^2, the coefficients XLA and Eigen use for f32), ~20 instruction
// slots.  Maximum absolute error 4.5e-7 against erf in double over [-6, 6] (tests/test_gpu_config3.py pins it through GELU against torch): a few ulps of a
// result near 1, i.e. GELU within 1e-6 absolute.
__device__ __forceinline__ float erf_rational(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return x * p * __builtin_amdgcn_rcpf(q);
}

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
        case ACT_GELU_ERF: return 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
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
