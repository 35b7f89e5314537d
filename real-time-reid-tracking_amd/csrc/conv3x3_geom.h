// Row maps of the LDS-halo 3x3 convolutions (conv3x3_f16.hip, conv3x3_x3.hip): which pixel of the block's 256 output pixels an
// MFMA row is, chosen per geometry so that the 16 lanes of every ds_read_b128 group address 16 halo pixels with distinct
// (halo index mod 16) at EVERY tap (a tap only shifts the index):
//   W = 32: a 32-row MFMA tile = one image row;  W = 16: one lane group = one image row;
//   W = 8 : one lane group = rows y and y+4 (halo pitch 10: 40 = 8 mod 16).
#pragma once

namespace {

// ds_read_b128 lane groups within a 32-lane half: {0-3,12-15,20-27} and {4-11,16-19,28-31}
__device__ __forceinline__ int lane_group(int i) { return ((i >= 4 && i < 12) || (i >= 16 && i < 20) || i >= 28) ? 1 : 0; }
__device__ __forceinline__ int lane_rank(int i) {
    // position of lane i inside its group (0..15)
    if (i < 4) return i;            // g0: 0-3   -> 0-3
    if (i < 12) return i - 4;       // g1: 4-11  -> 0-7
    if (i < 16) return i - 8;       // g0: 12-15 -> 4-7
    if (i < 20) return i - 8;       // g1: 16-19 -> 8-11
    if (i < 28) return i - 12;      // g0: 20-27 -> 8-15
    return i - 16;                  // g1: 28-31 -> 12-15
}

// MFMA row (wave-row wm 0..3, 32-row tile a 0..1, row i 0..31) -> (image in block, y in tile, x)
template <int TW, int IMGS>
__device__ __forceinline__ void row_to_pixel(int wm, int a, int i, int& img, int& y, int& x) {
    if constexpr (TW == 32) {          // block = 8 rows x 32: wave = 2 rows, tile = 1 row
        img = 0; y = wm * 2 + a; x = i;
    } else if constexpr (TW == 16) {   // block = 16 rows x 16: wave = 4 rows, tile = 2 rows, group = row
        img = 0; y = wm * 4 + a * 2 + lane_group(i); x = lane_rank(i);
    } else {                           // TW == 8, two 16x8 images: wave = 8 rows of one image, group = rows (y, y+4)
        const int k = lane_rank(i);
        img = wm >> 1; y = (wm & 1) * 8 + a * 2 + lane_group(i) + 4 * (k >> 3); x = k & 7;
    }
}

// Natural row (inside the block's 256 rows) of accumulator register e of MFMA tile (wm, a) in lane half lh.
// C row i = (e&3) + 8*(e>>2) + 4*lh; for that i the ds_read_b128 lane group is lh ^ (q == 1 || q == 2) with q = e >> 2 and the
// rank inside the group is simply e (check: q=0 -> lanes 0-3 / 4-7, q=1 -> 8-11 / 12-15, q=2 -> 16-19 / 20-23, q=3 -> 24-27 / 28-31).
template <int TW, int IMGS>
__device__ __forceinline__ int c_row_natural(int wm, int a, int e, int lh) {
    const int q = e >> 2;
    const int g = lh ^ ((q == 1 || q == 2) ? 1 : 0);
    if constexpr (TW == 32) return (wm * 2 + a) * 32 + (e & 3) + 8 * q + 4 * lh;
    else if constexpr (TW == 16) return (wm * 4 + a * 2 + g) * 16 + e;
    else return ((wm >> 1) * 16 + (wm & 1) * 8 + a * 2 + g + 4 * (e >> 3)) * 8 + (e & 7);
}

// PAIR (SPLIT builds with loader waves, 128-wide tiles): the three products of the fp32-class arithmetic in an order that shares
// operands - see the PAIR branch of the main loop.
// The same row split into the part that is uniform over the wave (per MFMA tile a and register e) and the part that depends on
// the lane half (and on x = "the register's quarter is 1 or 2"): c_row_natural = c_row_uniform + c_row_lane.  Buffer instructions
// take the first in the scalar offset and the second in the vector offset.
template <int TW, int IMGS>
__device__ __forceinline__ int c_row_uniform(int wm, int a, int e) {
    if constexpr (TW == 32) return (wm * 2 + a) * 32 + (e & 3) + 8 * (e >> 2);
    else if constexpr (TW == 16) return (wm * 4 + a * 2) * 16 + e;
    else return ((wm >> 1) * 16 + (wm & 1) * 8 + a * 2 + 4 * (e >> 3)) * 8 + (e & 7);
}
template <int TW, int IMGS>
__device__ __forceinline__ int c_row_lane(int lh, int x) {
    if constexpr (TW == 32) return 4 * lh;
    else if constexpr (TW == 16) return 16 * (lh ^ x);
    else return 8 * (lh ^ x);
}

}  // namespace
