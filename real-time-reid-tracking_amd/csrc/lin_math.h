// Arithmetic shared by the f16-operand linear kernels (gemm_f16.hip linear builds, two_linear_f16.hip): one definition, so the
// fused and the unfused forms of a layer round the same way.
#pragma once

// nn.GELU() (erf form), x * 0.5 * (1 + erf(x / sqrt 2)), for the fp16-storage mode.  erf by Abramowitz-Stegun 7.1.28,
// 1 - (1 + a1 z + ... + a6 z^6)^-16, |error| <= 3e-7 - three orders below the f16 rounding of the value it produces.  Every
// multiply-add is an explicit fma: ocml's erff, inlined into the 64- and the 128-wide instantiation, was contracted differently
// in the two and an image's embedding depended on the batch it came in.  20 issue slots instead of ~35.
__device__ __forceinline__ float gelu_f16_storage(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    float q = fmaf(0.0000430638f, z, 0.0002765672f);
    q = fmaf(q, z, 0.0001520143f);
    q = fmaf(q, z, 0.0092705272f);
    q = fmaf(q, z, 0.0422820123f);
    q = fmaf(q, z, 0.0705230784f);
    q = fmaf(q, z, 1.0f);
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;
    const float erfz = 1.0f - __builtin_amdgcn_rcpf(q);
    const float h = 0.5f * v;
    return fmaf(h, copysignf(erfz, v), h);
}

// The same function on a PAIR of values: every multiply-add is the packed form of the scalar one (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32 round each half like their scalar counterparts), so the two results are bit-identical to gelu_f16_storage() of the halves -
// at ~10 issue slots per value instead of ~17 (conv3x3_x3.hip, lin_x3_kernel: a 256 x 128 tile of fc1 is 128 values per lane, and
// beside the other block's MFMAs every vector instruction waits 11-45 cycles for its slot).
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu2_f16_storage(gelu_f32x2 v) {
    const gelu_f32x2 z = gelu_f32x2{fabsf(v.x), fabsf(v.y)} * 0.70710678118654752440f;
    gelu_f32x2 q = __builtin_elementwise_fma(gelu_f32x2{0.0000430638f, 0.0000430638f}, z, gelu_f32x2{0.0002765672f, 0.0002765672f});
    q = __builtin_elementwise_fma(q, z, gelu_f32x2{0.0001520143f, 0.0001520143f});
    q = __builtin_elementwise_fma(q, z, gelu_f32x2{0.0092705272f, 0.0092705272f});
    q = __builtin_elementwise_fma(q, z, gelu_f32x2{0.0422820123f, 0.0422820123f});
    q = __builtin_elementwise_fma(q, z, gelu_f32x2{0.0705230784f, 0.0705230784f});
    q = __builtin_elementwise_fma(q, z, gelu_f32x2{1.0f, 1.0f});
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;
    const gelu_f32x2 erfz = gelu_f32x2{1.0f - __builtin_amdgcn_rcpf(q.x), 1.0f - __builtin_amdgcn_rcpf(q.y)};
    const gelu_f32x2 h = v * 0.5f;
    return __builtin_elementwise_fma(h, gelu_f32x2{copysignf(erfz.x, v.x), copysignf(erfz.y, v.y)}, h);
}

// fp32 -> f16 with the fp32 value MATERIALISED first.  Without the (empty) asm the compiler folds the conversion into the
// instruction that produced the value - v_fma_mixlo_f16 computes fma(a, b, c) and rounds the exact result ONCE to f16 - and it
// does so per unrolled instance: in the round-3 build 63 of the 64 GELU instances of the 128-wide linear kernel ended in
// v_fma_mixlo_f16 and one (the element whose conversion was sunk below the loop's back edge) in v_fmac_f32 + v_cvt_f16_f32, which
// rounds twice (fp32, then f16).  The two differ by one f16 ulp whenever the fp32 rounding lands on an f16 tie (2^-13 of the
// values), so 1/64 of a tile's rows - the rows of that accumulator register - disagreed with the others on ~2e-6 of their
// elements: an image's Swin embedding depended on the tile row its tokens landed on (tests/test_gpu_parity.py,
// test_full_size_config2_swin_properties; tools/linear_row_check.py, tools/swin_position_probe.py).  Every f16 result of the
// linear epilogues goes through here: fp32 arithmetic, then one v_cvt_f16_f32, in every instance of every instantiation.
__device__ __forceinline__ _Float16 cvt_f16_rn(float v) {
    asm("" : "+v"(v));
    return (_Float16)v;
}
