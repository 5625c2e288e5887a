// Argument block of the VAE convolution kernels (conv3d_f32.hip: gather kernel, every geometry; conv3d_halo.hip / conv3d_halo16.hip: LDS-halo
// kernels for the 3x3(x3) stride-1 convolutions).
#pragma once
#include "common.h"

struct ConvArgs {
    const float* in;     // [Tin, Hin, Win, ld_in] channels-last
    const float* w;      // [Cout, taps * Cin]
    const float* bias;   // [Cout] or nullptr
    const float* resid;  // [M, ldr] or nullptr
    float* out;
    const float* zeros;
    long ld_in, ldo, ldr;
    int Tout, Hout, Wout, Tin, Hin, Win;
    int Cin, Cout, kt, kh, kw, st, sh, sw, t_off, ph, pw, up, interleave;
    int M, tiles_m, tiles_n;
    int ophase;          // -1: plain; 0..3 = (a, b) = (ophase >> 1, ophase & 1): output pixel (t, y, x) of this launch is stored at
                         // (t, 2y + a, 2x + b) of a [Tout, 2 Hout, 2 Wout] tensor (one phase of a 2x-upsampling convolution, see uv_conv3d_f32)
    const float* act_scale;   // f16x3: nullptr, or a device scalar the result is multiplied by as well (uv_vae_split_f16's 1 / s)
    float out_scale;     // f16x3 (PREC 4): the accumulators carry the power-of-two scale of the split weights; out = acc * out_scale + bias
};

// conv3d_halo.hip
bool uv_conv3d_halo_eligible(const ConvArgs& a, int prec);
int uv_launch_conv3d_halo(ConvArgs& a, int prec, hipStream_t stream);
// conv3d_halo16.hip (f16x3)
bool uv_conv3d_halo16_eligible(const ConvArgs& a);
int uv_launch_conv3d_halo16(ConvArgs& a, hipStream_t stream);
