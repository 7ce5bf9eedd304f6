// fft_x2.h -- geometry of the two-register-stage x passes (fft.hip: k_fft_x_fwd2, k_fft_x_inv2, k_fft_x_inv2c; kick_fused.hip)
#pragma once
#define BXC 16
// half lengths h = n/2 = R1*R2 with two-register-stage x kernels (k_fft_x_fwd2, k_fft_x_inv2)
#define P3M_X2_SIZES(X) X(32, 8, 4) X(40, 8, 5) X(48, 8, 6) X(56, 8, 7) X(64, 8, 8) X(80, 10, 8) X(88, 11, 8) X(96, 12, 8) X(104, 13, 8) X(112, 14, 8) \
  X(128, 16, 8) X(152, 19, 8) X(160, 16, 10) X(176, 16, 11) X(192, 16, 12) X(224, 16, 14) X(256, 16, 16) X(280, 20, 14) X(304, 19, 16) \
  X(320, 20, 16) X(352, 22, 16) X(384, 24, 16) X(416, 26, 16) X(448, 28, 16) X(512, 16, 32)

// row geometry of the two-register-stage x kernels (see k_fft_x_inv2)
template <int R1, int R2> struct X2Cfg {
  static constexpr int h = R1 * R2, Q = R1 > R2 ? R1 : R2, RPW = 64 / Q, TB = 256, RB = RPW * (TB / 64), R2P = R2 | 1, P = h + 1;
  static constexpr int NCH = h / BXC + 1, NLD = (RB * NCH * 8 + TB - 1) / TB;   // chunks holding columns 0..h; 16-byte loads per lane
  static constexpr size_t lds = sizeof(float2) * ((size_t)RB * P + (size_t)RB * R1 * R2P + h);
};

