// Internal interface of ftv_kernels.hip (big-map value gradient with D planes by LDS-DMA); called by nnue_ftm_backward_values_ws.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool ftv_supported(int B, int F, int P, int L1);
int64_t ftv_scratch_bytes(int B, int L1);
int ftv_launch(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1, float* d_conv_out, void* scratch,
               hipStream_t st);
