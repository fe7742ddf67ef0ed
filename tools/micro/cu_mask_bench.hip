// Micro-benchmark for the CU-partitioned pipeline: does hipExtStreamCreateWithCUMask confine a stream's
// workgroups to the CUs of its mask on MI355X (8 XCDs x 32 CUs), how do mask bits map to (XCD, CU), and do two streams
// with disjoint masks really run side by side?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_mask_bench.hip -o /tmp/cmb && /tmp/cmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// where: [block] = xcc_id << 16 | hw_id bits (cu_id 11:8, sh_id 12, se_id 15:13 on gfx9)
__global__ __launch_bounds__(1024) void where_kernel(unsigned *where, int spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) where[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
    // busy work so that all blocks of a launch are resident together
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) where[0] = 0;
}

__global__ __launch_bounds__(1024) void spin_kernel(float *out, int spin)
{
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = a;
}

static void census(const char *what, hipStream_t s, int blocks, unsigned *d_where)
{
    std::vector<unsigned> h(blocks);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(1024), 0, s, d_where, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_where, blocks * 4, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    bool seen[16][512];
    memset(seen, 0, sizeof seen);
    int distinct = 0;
    for (unsigned w : h) {
        const int x = (w >> 16) & 15, id = ((w >> 8) & 0xF) | (((w >> 12) & 1) << 4) | (((w >> 13) & 7) << 5);
        per_xcc[x]++;
        if (!seen[x][id]) { seen[x][id] = true; distinct++; }
    }
    printf("%-44s %4d blocks on %3d distinct CUs; blocks per XCD:", what, blocks, distinct);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
}

int main()
{
    unsigned *d_where; float *d_out;
    CK(hipMalloc(&d_where, 4096 * 4)); CK(hipMalloc(&d_out, 4096));
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    census("no mask", plain, 256, d_where);
    // masks: 8 x 32-bit words = 256 CUs
    struct { const char *name; uint32_t m[8]; } masks[] = {
        {"first 64 bits (0-63)", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}},
        {"last 64 bits (192-255)", {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu}},
        {"bits with (i % 8) < 2", {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u}},
        {"bits with (i % 4) == 0", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}},
    };
    hipStream_t ms[4];
    for (int k = 0; k < 4; ++k) {
        hipError_t e = hipExtStreamCreateWithCUMask(&ms[k], 8, masks[k].m);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s): %s\n", masks[k].name, hipGetErrorString(e)); return 0; }
        census(masks[k].name, ms[k], 64, d_where);
        census(masks[k].name, ms[k], 256, d_where);
    }
    // concurrency: a 192-CU job and a 64-CU job, alone and together
    hipStream_t a, b;
    uint32_t ma[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0};
    uint32_t mb[8] = {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu};
    CK(hipExtStreamCreateWithCUMask(&a, 8, ma));
    CK(hipExtStreamCreateWithCUMask(&b, 8, mb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *what, bool ra, bool rb) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, plain));
        CK(hipStreamWaitEvent(a, e0, 0)); CK(hipStreamWaitEvent(b, e0, 0));
        if (ra) for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin_kernel, dim3(192), dim3(1024), 0, a, d_out, 40000);
        if (rb) for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(1024), 0, b, d_out, 40000);
        hipEvent_t ea, eb;
        CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        CK(hipEventRecord(ea, a)); CK(hipEventRecord(eb, b));
        CK(hipStreamWaitEvent(plain, ea, 0)); CK(hipStreamWaitEvent(plain, eb, 0));
        CK(hipEventRecord(e1, plain));
        CK(hipEventSynchronize(e1));
        float ms_ = 0;
        CK(hipEventElapsedTime(&ms_, e0, e1));
        printf("%-60s %.3f ms\n", what, ms_);
    };
    timed("warm-up", true, true);
    timed("20 x 192-block kernels on the 192-CU stream alone", true, false);
    timed("20 x 64-block kernels on the 64-CU stream alone", false, true);
    timed("both together (disjoint masks)", true, true);
    return 0;
}
