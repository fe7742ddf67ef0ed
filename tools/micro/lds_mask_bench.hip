// Micro-benchmark for the incremental value net's gather phase: what does a ds_read_b128 wave-instruction
// cost the LDS when only some of its lanes are active?  The gather loop runs as many trips as the longest (feature, delta)
// list of the wave; lanes that have run out read a dummy row.  If the LDS skips fully inactive 16-lane groups, sorting a
// tile's rows by list length and masking the exhausted lanes off would cut LDS time by the waste (7.2 trips vs 5.0 mean).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_mask_bench.hip -o gpurun_out/lmb && gpurun_out/lmb
// 16 waves per CU (1 024 threads, 117 KB of LDS like the real kernel), every CU busy; per wave ITER x 16 ds_read_b128 of
// 528-byte rows chosen per lane, consumed by packed FMAs (so nothing is optimised away).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 222, STRIDE = 132, ITER = 2000;

// MODE 0: all 64 lanes active.  1: lanes 0..15 (one lane group).  2: lanes 0..31.  3: lanes 0..47.  4: every 4th lane (16 lanes,
// spread over all four groups).  5: all lanes, but the inactive ones of mode 1 read ONE shared dummy address (what the kernel does today)
template <int MODE, int FMAS>
__global__ __launch_bounds__(1024) void k(const int *rows, float *out)
{
    extern __shared__ float4 sW[];
    for (int i = threadIdx.x; i < ROWS * STRIDE / 4; i += 1024) sW[i] = make_float4(1.f + i, 2.f, 3.f, 4.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bool act = true;
    if (MODE == 1) act = lane < 16;
    if (MODE == 2) act = lane < 32;
    if (MODE == 3) act = lane < 48;
    if (MODE == 4) act = (lane & 3) == 0;
    int r = rows[(blockIdx.x * 1024 + threadIdx.x) % 4096];
    if (MODE == 5 && lane >= 16) r = 0;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (act) {
        for (int it = 0; it < ITER; ++it) {
            const f32x4 *p = reinterpret_cast<const f32x4 *>(sW) + r * (STRIDE / 4);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 w = p[c];
                if (FMAS) acc[c & 3] = __builtin_elementwise_fma(w, (f32x4){1.0001f, 1.0001f, 1.0001f, 1.0001f}, acc[c & 3]);
                else acc[c & 3] += w;
            }
            r = (r * 9 + 4 + (int)acc[0].x % 2) % ROWS;        // next row depends on the data: reads cannot be hoisted
            if (r < 0) r = -r;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}

template <int MODE>
void run(const char *what, const int *d_rows, float *d_out)
{
    auto fn = k<MODE, 1>;
    CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * STRIDE * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(256), dim3(1024), ROWS * STRIDE * 4, 0, d_rows, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(256), dim3(1024), ROWS * STRIDE * 4, 0, d_rows, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double reads = 16.0 * ITER * 16;                       // ds_read_b128 wave-instructions per CU
    printf("%-58s %8.3f ms  %6.2f ns per ds_read_b128 wave-instruction per CU  (%.1f LDS clk at 2.4 GHz)\n", what, ms,
           ms * 1e6 / reads, ms * 1e6 / reads * 2.4);
}

int main()
{
    std::vector<int> h(4096);
    srand(7);
    for (auto &x : h) x = rand() % ROWS;
    int *d_rows; float *d_out;
    CK(hipMalloc(&d_rows, 4096 * 4)); CK(hipMalloc(&d_out, 256 * 1024 * 4));
    CK(hipMemcpy(d_rows, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("all 64 lanes, random rows", d_rows, d_out);
    run<1>("lanes 0-15 active (EXEC prefix), others masked off", d_rows, d_out);
    run<2>("lanes 0-31 active", d_rows, d_out);
    run<3>("lanes 0-47 active", d_rows, d_out);
    run<4>("every 4th lane active (16 lanes in 4 groups)", d_rows, d_out);
    run<5>("all 64 active, lanes 16-63 read one shared dummy row", d_rows, d_out);
    return 0;
}
