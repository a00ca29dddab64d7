// How many cycles does one SIMD need per wave64 vector-ALU instruction, with 1, 2, 4 and 8
// wavefronts resident per SIMD?  Every wavefront runs a loop of independent v_fma_f32 (eight
// accumulators); the shader clock is read around the loop.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_issue_probe tools/valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096
#define UNROLL 8

__global__ void __launch_bounds__(256) k_fma(float* out, long long* cycles, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_fmaf(x0, a, b);
        x1 = __builtin_fmaf(x1, a, b);
        x2 = __builtin_fmaf(x2, a, b);
        x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b);
        x5 = __builtin_fmaf(x5, a, b);
        x6 = __builtin_fmaf(x6, a, b);
        x7 = __builtin_fmaf(x7, a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int max_blocks = cus * 8;
    float* out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * max_blocks);
    hipMalloc(&cyc, sizeof(long long) * 4 * max_blocks);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {  // 256-thread workgroups per CU = wavefronts per SIMD
        const int blocks = cus * per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int reps = 10;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(4 * blocks);
        hipMemcpy(h.data(), cyc, sizeof(long long) * 4 * blocks, hipMemcpyDeviceToHost);
        double sum = 0;
        for (long long v : h) sum += (double)v;
        const double wave_cycles = sum / h.size();                         // shader clocks one wavefront spent in its loop
        const double instr = (double)ITERS * UNROLL;                       // vector instructions per wavefront
        // per SIMD: per_cu wavefronts share it, so it retires per_cu * instr instructions in wave_cycles
        printf("%d wavefront(s) per SIMD: %.0f clocks per wavefront loop, %.2f clocks per instruction per wavefront, "
               "%.2f clocks per instruction per SIMD; %.3f ms per launch -> %.2f G wave-instr/s per SIMD\n",
               per_cu, wave_cycles, wave_cycles / instr, wave_cycles / (instr * per_cu), ms / reps,
               instr * per_cu / (ms / reps * 1e-3) / 1e9);
    }
    return 0;
}
