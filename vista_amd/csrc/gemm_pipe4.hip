// Four-wave build of the pipelined 256x320 GEMM (round-6 experiment): gemm_pipe.hip compiled a second time with PIPE_W4 -- four waves of 128 x 160,
// one per SIMD, 512 registers per lane. Entry points vk_gemm_pipe4_{fit,gnstat_ok,launch}; routed to by gemm.hip when VISTA_GEMM_PIPE4 asks for it.
#define PIPE_W4 1
#include "gemm_pipe.hip"
