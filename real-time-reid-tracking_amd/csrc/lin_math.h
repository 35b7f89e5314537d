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
