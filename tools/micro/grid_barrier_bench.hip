// Micro-benchmark for DESIGN.md §8: what does a dependency between two phases of the step cost
//   (a) as a kernel boundary (two launches on one stream), and
//   (b) as a grid-wide barrier inside one persistent launch (one 1 024-thread workgroup per CU, all resident)?
// Build + run (through gpurun):  hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier_bench.hip -o gpurun_out/gbb && gpurun_out/gbb
// The spin loops give up after a bounded number of polls, so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void phase_kernel(unsigned int *buf, int round)
{
    // one small dependent memory operation per thread, like the head of every stage of the step
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    buf[i] = buf[i] + (unsigned int)round;
}

// variant 1: relaxed polls, one acquire fence after the last; variant 2: no data fences at all (lower bound: what the
// arrival counter and the polling alone cost)
template <int VARIANT>
__global__ __launch_bounds__(1024) void persistent_kernel_v(unsigned int *buf, unsigned int *counter, int rounds, unsigned int *gave_up)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        buf[i] = buf[i] + (unsigned int)r;
        if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = (unsigned int)(r + 1) * gridDim.x;
            unsigned int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
                if (++polls > 4000000u) { atomicAdd(gave_up, 1u); break; }
            if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void persistent_kernel(unsigned int *buf, unsigned int *counter, int rounds, unsigned int *gave_up)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        buf[i] = buf[i] + (unsigned int)r;
        // grid barrier: writes visible, one arrival per workgroup, poll until all have arrived
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned int want = (unsigned int)(r + 1) * gridDim.x;
            unsigned int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > 4000000u) { atomicAdd(gave_up, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount, rounds = 400;
    unsigned int *buf, *counter, *gave_up;
    CK(hipMalloc(&buf, (size_t)grid * 1024 * 4));
    CK(hipMalloc(&counter, 8));
    gave_up = counter + 1;
    CK(hipMemset(buf, 0, (size_t)grid * 1024 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(1024), 0, 0, buf, r);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        printf("%d CUs: %d dependent launches of a 1024-thread-per-CU kernel: %.2f us per launch\n", grid, rounds, 1e3 * ms / rounds);
        CK(hipMemset(counter, 0, 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(1024), 0, 0, buf, counter, rounds, gave_up);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        unsigned int h[2];
        CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost));
        printf("%d CUs: one persistent launch, %d phases separated by grid barriers (threadfence + acquire polls): %.2f us per phase (gave up: %u)\n", grid, rounds, 1e3 * ms / rounds, h[1]);
        CK(hipMemset(counter, 0, 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(persistent_kernel_v<1>, dim3(grid), dim3(1024), 0, 0, buf, counter, rounds, gave_up);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost));
        printf("%d CUs: ... release fence, relaxed polls, one acquire fence: %.2f us per phase (gave up: %u)\n", grid, 1e3 * ms / rounds, h[1]);
        CK(hipMemset(counter, 0, 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(persistent_kernel_v<2>, dim3(grid), dim3(1024), 0, 0, buf, counter, rounds, gave_up);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost));
        printf("%d CUs: ... no data fences (counter + polling only; NOT a correct barrier for data): %.2f us per phase (gave up: %u)\n", grid, 1e3 * ms / rounds, h[1]);
    }
    return 0;
}
