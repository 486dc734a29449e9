// stp3_dpp.h -- the one CDNA instruction the kernels issue by hand.
//
// fmac_row_bcast<J>(acc, v, f):  acc += v[(lane & ~15) | J] * f   in ONE VALU instruction: v_fmac_f32 with the DPP
// row_newbcast control (gfx90a+: lane J of every 16-lane row is broadcast to the row).  The compiler emits the
// broadcast as a separate v_mov_b32_dpp when it is written with __builtin_amdgcn_update_dpp (row_newbcast is not
// folded into the multiply-add), which doubles the instruction count of the lift kernel's inner loop.
// (tests/hipcpu provides a header of the same name that models the instruction for the CPU stand-in.)
#pragma once

template <int J>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float v, float f) {
    static_assert(J >= 0 && J < 16, "row_newbcast lane");
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(f), "n"(J));
}
