// How does the issue rate of vector instructions on gfx950 depend on the number of enabled lanes?
// Times a long chain of independent FMAs with different sets of lanes enabled, in all waves or in classes of waves
// (by wave, by block, by XCD, alternating in time), and reports the kernel's time and the mean time of a wave of
// either class.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/exec_rows_probe.hip -o tools/_bin/exec_rows_probe_scalar
//   (without -fno-slp-vectorize the compiler pairs the FMAs into v_pk_fma_f32)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

enum { kByWave = 0, kByBlock = 1, kByXcd = 2, kInTime = 3 };

__global__ __launch_bounds__(256) void k_chain(float *out, uint32_t *ticks, uint64_t mask_a, uint64_t mask_b, int mode,
                                               int period, int iters)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool b = false;
    if (mode == kByWave) b = wave & 1u;
    if (mode == kByBlock) b = (blockIdx.x >> 3) & 1u;
    if (mode == kByXcd) b = (blockIdx.x & 7u) != 0u;
    float a0 = lane, a1 = lane + 1.f, a2 = lane + 2.f, a3 = lane + 3.f, a4 = lane + 4.f, a5 = lane + 5.f, a6 = lane + 6.f,
          a7 = lane + 7.f;
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (mode == kInTime) b = (i / period) & 1;
        const uint64_t mask = b ? mask_b : mask_a;
        if ((mask >> lane) & 1ull) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fmaf(a0, 1.0000001f, 0.5f);
                a1 = __builtin_fmaf(a1, 1.0000001f, 0.5f);
                a2 = __builtin_fmaf(a2, 1.0000001f, 0.5f);
                a3 = __builtin_fmaf(a3, 1.0000001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0000001f, 0.5f);
                a5 = __builtin_fmaf(a5, 1.0000001f, 0.5f);
                a6 = __builtin_fmaf(a6, 1.0000001f, 0.5f);
                a7 = __builtin_fmaf(a7, 1.0000001f, 0.5f);
            }
        }
    }
    const uint64_t t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (lane == 0u) ticks[blockIdx.x * 4u + wave] = uint32_t(t1 - t0) | (b ? 0x80000000u : 0u);
}

int main()
{
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    float *out;
    uint32_t *ticks;
    if (hipMalloc(&out, sizeof(float) * blocks * threads) != hipSuccess) return 1;
    if (hipMalloc(&ticks, sizeof(uint32_t) * blocks * 4) != hipSuccess) return 1;
    const uint64_t all = ~0ull, l16 = 0xFFFFull, l17 = 0x1FFFFull, l8 = 0xFFull, l32 = 0xFFFFFFFFull;
    struct { const char *name; uint64_t a, b; int mode, period; } cases[] = {
        {"all waves 64 lanes", all, all, kByWave, 1},
        {"all waves 32 lanes", l32, l32, kByWave, 1},
        {"all waves 17 lanes", l17, l17, kByWave, 1},
        {"all waves 16 lanes", l16, l16, kByWave, 1},
        {"all waves 16 lanes, every 4th", 0x1111111111111111ull, 0x1111111111111111ull, kByWave, 1},
        {"all waves 8 lanes", l8, l8, kByWave, 1},
        {"waves: even 64, odd 16", all, l16, kByWave, 1},
        {"waves: even 17, odd 16", l17, l16, kByWave, 1},
        {"waves: even 17, odd 8", l17, l8, kByWave, 1},
        {"blocks: even 64, odd 16", all, l16, kByBlock, 1},
        {"blocks: even 17, odd 8", l17, l8, kByBlock, 1},
        {"XCD 0 64 lanes, others 8", all, l8, kByXcd, 1},
        {"XCD 0 17 lanes, others 8", l17, l8, kByXcd, 1},
        {"in time: 64 / 8, period 1", all, l8, kInTime, 1},
        {"in time: 64 / 8, period 16", all, l8, kInTime, 16},
        {"in time: 64 / 8, period 256", all, l8, kInTime, 256},
        {"in time: 17 / 8, period 1", l17, l8, kInTime, 1},
        {"in time: 17 / 8, period 16", l17, l8, kInTime, 16},
        {"in time: 17 / 8, period 256", l17, l8, kInTime, 256},
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<uint32_t> h(blocks * 4);
    for (auto &c : cases) {
        k_chain<<<blocks, threads>>>(out, ticks, c.a, c.b, c.mode, c.period, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_chain<<<blocks, threads>>>(out, ticks, c.a, c.b, c.mode, c.period, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), ticks, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        double sum[2] = {0, 0};
        long n[2] = {0, 0};
        for (uint32_t v : h) sum[v >> 31] += v & 0x7FFFFFFFu, ++n[v >> 31];
        std::printf("%-32s kernel %7.3f ms   wave of class a %7.3f ms   of class b %7.3f ms\n", c.name, ms,
                    n[0] ? sum[0] / n[0] * 1e-5 : 0.0, n[1] ? sum[1] / n[1] * 1e-5 : 0.0);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
