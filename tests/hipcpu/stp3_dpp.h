// TEST INFRASTRUCTURE: CPU model of st-p3_amd/csrc/stp3_dpp.h (found first on the stand-in's include path).
#pragma once
template <int J>
inline void fmac_row_bcast(float& acc, float v, float f) {
    const float b = hipcpu::exchange<float>(v, (hipcpu::cur->lane & ~15) | J);     // row_newbcast:J
    acc = fmaf(b, f, acc);
}
