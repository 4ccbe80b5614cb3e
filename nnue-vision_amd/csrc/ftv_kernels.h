// Internal interface of ftv_kernels.hip (big-map value gradient with D planes by LDS-DMA); called by nnue_ftm_backward_values_ws.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool ftv_supported(int B, int F, int P, int L1);
int64_t ftv_scratch_bytes(int B, int L1);
int ftv_launch(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1, float* d_conv_out, void* scratch,
               hipStream_t st);
// d_out [rows][cols] f32 -> K-tile-major planes [cols / 32][3][rows][32] bf16 (exact hi / mid / lo truncation split); cols % 32 == 0
void ftv_split_planes(const float* src, int rows, int cols, void* planes, hipStream_t st);
