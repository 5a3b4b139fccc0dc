// Shared between the exact-fp32 (tdr_wgrad_mfma.hip) and split-bf16 (tdr_wgrad_bx3.hip) weight-gradient kernels.
#pragma once
#include "tdr_common.h"

struct TdrWgradDesc;

struct WgArgs {
    const float* in; long in_ns; int Cin, H, W; long gate_off;
    const float* dout; long dout_ns; int Cout, OH, OW;
    int pad, tw_log2, tiles_x, tiles_y, tpi /*tiles per image*/, tps /*tiles per split*/, spi /*splits per image*/;
    float* part;
    float* dbpart;      // optional [nsplit][Cout]: per-split sums of dout rows (bias gradient), ci-tile 0 only
    int scheme;         // split-kernel operand scheme: 0 = 3-way bf16, 1 = 2-way fp16 (TdrWgradDesc.math == 2)
    const void* grp_tab; // grouped launch (tdr_wgrad1x1_group): TdrWg1GroupEntry[] in device memory, in / dout / part / dbpart per problem
    int grp_bpp;        // ... (image, split) pairs per problem (= N * spi); 0: a single problem, pointers above
    int grp_pairs;      // ... nprob * grp_bpp; the grid is 1-D: 64-block chunks = 8 pairs (one per XCD) x the output tiles of a pair
};

struct WgPlan { int tw_log2, tiles_x, tiles_y, tpi, tps, spi, cfg, WKw, BMc, BNc; };

// tdr_wgrad_bx3.hip
bool tdr_wgrad_bx3_supported(const TdrWgradDesc* d);
WgPlan tdr_wgrad_bx3_plan(const TdrWgradDesc* d);
int tdr_wgrad_bx3_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st);

// tdr_wgrad_s2.hip (3x3 stride 2 on the 2-way fp16 split)
bool tdr_wgrad_s2_supported(const TdrWgradDesc* d);
WgPlan tdr_wgrad_s2_plan(const TdrWgradDesc* d);
int tdr_wgrad_s2_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st);

// tdr_wgrad_1x1.hip (1x1 on the split schemes: LDS-DMA ring of raw rows, operands split in registers)
bool tdr_wgrad_1x1_supported(const TdrWgradDesc* d);
WgPlan tdr_wgrad_1x1_plan(const TdrWgradDesc* d);
int tdr_wgrad_1x1_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st);
