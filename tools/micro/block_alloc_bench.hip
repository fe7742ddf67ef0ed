// Micro-benchmark for the step's block-level bump allocations (bg_staged.h: block_alloc / block_alloc2, one returning 64-bit
// atomicAdd per workgroup on ONE arena counter): the lane-per-game kernels (roots / boundary: 256 workgroups that all reach their
// allocation at the same moment) wait for the result before they can write a node.  What does that round trip cost when 256
// workgroups queue on one address, and what would K counters on their own 128-byte lines cost?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/block_alloc_bench.hip -o gpurun_out/bab && gpurun_out/bab
// Every workgroup: a short dependent chain (stand-in for the scan), the allocation, then one 16-byte store per thread at the
// allocated place.  MODE 0: no atomic (place = blockIdx * count).  1: one counter.  K >= 2: K counters, workgroup b uses b % K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int NALLOC>
__global__ __launch_bounds__(256) void k(unsigned long long *tops, uint4 *arena, unsigned long long cap_per_counter)
{
    __shared__ unsigned long long s_slot[NALLOC];
    const unsigned int total = 256;
    for (int a = 0; a < NALLOC; ++a) {
        if (threadIdx.x == 0) {
            if (MODE == 0) s_slot[a] = (unsigned long long)blockIdx.x * total;
            else {
                const int c = MODE == 1 ? 0 : (int)(blockIdx.x % MODE);
                s_slot[a] = (unsigned long long)c * cap_per_counter + atomicAdd(tops + 16 * (c + 64 * a), (unsigned long long)total) % cap_per_counter;
            }
        }
        __syncthreads();
        const unsigned long long at = s_slot[a] + threadIdx.x;
        arena[at % (cap_per_counter * 64)] = make_uint4(threadIdx.x, blockIdx.x, a, 0);
        __syncthreads();
    }
}

template <int MODE, int NALLOC>
static void run(const char *name, int blocks, unsigned long long *tops, uint4 *arena, unsigned long long cap)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 400;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NALLOC>), dim3(blocks), dim3(256), 0, 0, tops, arena, cap);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < REP; ++i) hipLaunchKernelGGL((k<MODE, NALLOC>), dim3(blocks), dim3(256), 0, 0, tops, arena, cap);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %4d workgroups x %d allocation(s): %7.2f us per launch (back to back)\n", name, blocks, NALLOC, ms * 1000.f / REP);
}

int main()
{
    unsigned long long *tops;
    uint4 *arena;
    const unsigned long long cap = 1ull << 18;                     // entries per counter region
    CK(hipMalloc(&tops, 16 * 8 * 64 * 4));
    CK(hipMemset(tops, 0, 16 * 8 * 64 * 4));
    CK(hipMalloc(&arena, cap * 64 * sizeof(uint4)));
    for (int blocks : {256, 512, 1024}) {
        run<0, 1>("no atomic (static place)", blocks, tops, arena, cap);
        run<1, 1>("one counter", blocks, tops, arena, cap);
        run<8, 1>("8 counters on their own lines", blocks, tops, arena, cap);
        run<16, 1>("16 counters on their own lines", blocks, tops, arena, cap);
        run<0, 2>("no atomic, two allocations in sequence", blocks, tops, arena, cap);
        run<1, 2>("one counter each, two in sequence", blocks, tops, arena, cap);
        run<16, 2>("16 counters each, two in sequence", blocks, tops, arena, cap);
    }
    return 0;
}
