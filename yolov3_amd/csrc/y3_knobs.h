// Run-time tuning knobs of libyolov3_hip.so: ids (this enum) and names / defaults (api.cpp::kKnobs) in the same order.
#pragma once
enum y3_knob_id {
    Y3K_CONV = 0,       // "conv":        0 per-shape dispatch; 2 register-staged v2; 4 / 5 / 6 the v3 tiles 128x256 / 128x128 BK32 / 128x128 BK64; 15 v6
    Y3K_CONV_AHEAD,     // "conv_ahead":  K-steps the LDS-DMA requests of v6 / wgrad_big run ahead of the MFMAs (3; 2 = the round-2 schedule)
    Y3K_BN_NT_BYTES,    // "bn_nt_bytes": tensors at least this large take the non-temporal forms of the elementwise BatchNorm passes
    Y3K_WGRAD,          // "wgrad":       0 per-shape; 2 the 128x128 kernel; 3 the 256x256 kernel wherever the shape allows; 4 the direct fp32 kernel
    Y3K_WGRAD_XCD,      // "wgrad_xcd":   0 dispatch order; 1 a slice's tiles on one XCD for the 128-tile kernel; 2 + the 256-tile kernel; 3 all
    Y3K_DGRAD_QUAD,     // "dgrad_quad":  1 the four parity classes of a stride-2 data gradient in one launch; 0 four launches
    Y3K_SPP_DIRECT,     // "spp_direct":  1 SPP pools without the LDS pyramid
    Y3K_WGRAD_STRIP,    // "wgrad_strip": 1 the strip-walking filter-gradient kernel for the 3x3 layers with 32 -> 64 / 64 -> 128 channels (wgrad_strip.h); 0 never; 2 also small launches; N > 2: N K-steps per block (tests)
    Y3K_CONV_STRIP,     // "conv_strip":  1 the strip-walking 3x3 kernel with register-resident filters for 32 -> 64 / 64 -> 32 / 64 -> 128 channels (conv_strip.h); 0 never; 2 also small launches; N > 2: N rows per block (tests)
    Y3K_CONV_V10,       // "conv_v10":    1 the persistent one-wave-per-SIMD 3x3 kernel with register-resident filter fragments (conv_v10.h) for Cin >= 128 at a quarter round of tiles or more; 0 never; 2 every eligible shape
    Y3K_V10_MP,         // "v10_mp":      0 the widest body whose halo patch fits; 6 / 7 / 8 cap the wave-tile width (32-pixel column blocks) of conv_v10.h (tests)
    Y3K_V10_BLOCKS,     // "v10_blocks":  0 one block per CU; N > 0 blocks per filter tile (tests: blocks that walk many tiles, single-column-block tiles)
    Y3K_V10_HALF,       // "v10_half":    2 the measured choice (Cin <= 256); 0 one block per CU (bodies of 6 / 7 / 8 column blocks); 1 two half-size blocks per CU (bodies of 3 / 4) wherever the form fits
    Y3K_V10_KSPLIT,     // "v10_ksplit":  1 small launches (below a quarter round of tiles, workspace given) run conv_v10.h's K-split form; 0 never; 2 every eligible launch (tests)
    Y3K_V10_SLICES,     // "v10_slices":  0 as many slices of the channel blocks as fill the chip; N > 0 force N (tests: uneven splits, one channel block per slice)
    Y3K_V10_GROUP,      // "v10_group":   1 the blocks of a filter tile on one XCD take the tiles of their common pixel range round-robin (neighbouring tiles in flight together: halo rows meet in that L2); 0 every block walks a contiguous run
    Y3K_TILE_XCD,       // "tile_xcd":    1 the persistent tile loops of stem_pair / bneck_pair walk XCD-grouped tile ids (64 neighbouring tiles per XCD and round: halo rows and
                        //                shared cache lines meet in one L2); 0 dispatch order (A/B)
    Y3K_CONV_1X1S,      // "conv_1x1s":   1 the persistent 1x1 kernel with register-resident filters (conv_1x1s.h) on the HBM-bound 1x1 layers (Cin <= 384, Cout <= 256, >= 32768 pixels); 0 never; 2 also small launches (tests)
    Y3K_WGRAD_PATCH,    // "wgrad_patch": 1 the padded-position filter-gradient kernel (wgrad_patch.h) on the 3x3 / stride-1 layers with Cin % 64 == 0, Cout % 128 == 0 at >= 16384 pixels; 0 never (wgrad_big / wgrad_dma); 2 every eligible shape (tests)
    Y3K_WGRAD_BLOCKS,   // "wgrad_blocks": blocks (tiles x pixel slices) the 128 x 128-tile filter-gradient kernel aims at: every slice writes a 64 KiB partial tile per column tile (512 = one round of resident blocks; 1024 until round 6: +0.6 ms per step of slab traffic)
    Y3K_NMS_SORT,       // "nms_sort":    1 the two orderings of y3_nms (per-image score order, per-(image, class) segments) by ONE block-per-image launch of stable counting passes (detect_nms.hip::nms_sort_kernel); 0 two rocPRIM device radix sorts (rounds 1-5; A/B and cross-check in the tests)
    Y3K_V10_DEFER,      // "v10_defer":   1 inference launches of conv_v10.h's half-form shapes (SiLU, no statistics) run conv_v10d.h: one block per CU, the previous tile's activation inside the next tile's K loop; 0 never; 2 every eligible shape incl. the full-form ones (tests / A/B)
    Y3K_COUNT
};
long long y3_knob(int id);
