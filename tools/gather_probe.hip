// Random-sector roofline probe (stand-alone, no library): how many random 64-byte sectors per second the chip delivers to vector gathers, in the two
// regimes that matter for the sequential LSD core (structure-slam-pointline_amd/csrc/lsd_regions.h):
//   throughput   every lane keeps K independent 4-byte loads in flight at random 64-B-aligned addresses of a `bytes`-sized buffer (default 16 GiB, above the 14 GB of the
//                planes of the 6 144 resident frames of the bench step), 8 waves per SIMD: the fabric's request rate
//   dependent    one load per lane and round whose address depends on the value loaded before (the buffer holds a random permutation step), W waves
//                per SIMD (default 6 = the core's residency): what a chain of dependent stagings can reach, and the round-trip time behind it
//   unloaded     the same chain with one wave per compute unit: the round trip of a dependent gather when nothing queues
// Prints one JSON object.  Built by tools/build_c_harnesses.sh (hipcc --offload-arch=gfx950); bench.py runs it for roofline.random_sector.
//     tools/gather_probe [GiB=16] [waves_dependent=6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_fill(unsigned* __restrict__ buf, size_t nSect, unsigned long long seed) {
    // one dword per 64-B sector is read by the dependent chain: it holds the next sector's index (a pseudo-random function of this one)
    for (size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x; s < nSect; s += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (s + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        buf[s * 16] = (unsigned)(x % nSect);
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_gather_indep(const unsigned* __restrict__ buf, size_t sectMask, int iters, unsigned* __restrict__ sink) {
    unsigned long long x = 0x9E3779B97F4A7C15ull * (1 + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v[k] = buf[((size_t)(x >> 11) & sectMask) * 16];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
    }
    if (acc == 0x12345u) *sink = acc;
}

__global__ __launch_bounds__(64) void k_gather_dep(const unsigned* __restrict__ buf, size_t nSect, int iters, unsigned* __restrict__ sink) {
    size_t s = (0x9E3779B97F4A7C15ull * (1 + blockIdx.x * 64ull + threadIdx.x)) % nSect;
    for (int it = 0; it < iters; ++it) s = buf[s * 16];
    if (s == 0xFFFFFFFFu) *sink = (unsigned)s;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int wavesDep = argc > 2 ? atoi(argv[2]) : 6;
    CHK(hipSetDevice(0));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    size_t nSect = 1; while ((nSect * 2) * 64 <= (size_t)(gib * (1ull << 30))) nSect *= 2;      // power of two: the independent probe masks
    unsigned *buf, *sink;
    CHK(hipMalloc((void**)&buf, nSect * 64)); CHK(hipMalloc((void**)&sink, 4));
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, buf, nSect, 0x5EEDull);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    // throughput regime: K = 8 loads in flight per lane, 8 waves per SIMD
    const int blocksI = cus * 8, itersI = 512;
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather_indep<8>, dim3(blocksI), dim3(256), 0, 0, buf, nSect - 1, itersI, sink);
        CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
    }
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double sectI = (double)blocksI * 256 * itersI * 8 / (ms * 1e-3);
    // dependent regime: W single-wave workgroups per SIMD, one load per lane and round, the next address is the loaded value
    const int blocksD = cus * 4 * wavesDep, itersD = 2048;
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather_dep, dim3(blocksD), dim3(64), 0, 0, buf, nSect, itersD, sink);
        CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
    }
    float msD = 0; CHK(hipEventElapsedTime(&msD, e0, e1));
    const double sectD = (double)blocksD * 64 * itersD / (msD * 1e-3);
    // unloaded round trip: one wave per compute unit (16 K requests in flight on the whole chip: far below the fabric's rate), the same dependent chain
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather_dep, dim3(cus), dim3(64), 0, 0, buf, nSect, itersD, sink);
        CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
    }
    float msU = 0; CHK(hipEventElapsedTime(&msU, e0, e1));
    printf("{\"buffer_gib\": %.2f, \"compute_units\": %d, \"independent\": {\"loads_in_flight_per_lane\": 8, \"waves_per_simd\": 8, \"gsectors_per_s\": %.2f, \"gb_per_s_at_64B\": %.1f, \"ms\": %.3f}, "
           "\"dependent\": {\"waves_per_simd\": %d, \"gsectors_per_s\": %.2f, \"gb_per_s_at_64B\": %.1f, \"round_trip_us\": %.3f, \"ms\": %.3f}, "
           "\"unloaded\": {\"waves\": %d, \"round_trip_us\": %.3f, \"gsectors_per_s\": %.2f}}\n",
           nSect * 64.0 / (1ull << 30), cus, sectI / 1e9, sectI * 64 / 1e9, ms, wavesDep, sectD / 1e9, sectD * 64 / 1e9, msD * 1e3 / itersD, msD,
           cus, msU * 1e3 / itersD, (double)cus * 64 * itersD / (msU * 1e-3) / 1e9);
    return 0;
}
