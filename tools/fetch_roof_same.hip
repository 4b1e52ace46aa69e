// fetch_roof_same.hip -- a second look at what a scattered wave-level 16-byte load costs the vector memory pipe of an MI355X CU
// (tools/fetch_roof.hip prices SHAPES of contiguous runs): what do lanes cost that read the SAME 16 bytes (rays of a wavefront on
// the same BVH node), adjacent or spread over the wavefront, and what do lanes cost that do not take part at all (the walk
// steps with ~45 of 64 lanes)?
//
//   hipcc --offload-arch=gfx950 -O2 -o build/fetch_roof_same tools/fetch_roof_same.hip && build/fetch_roof_same > fetch_roof_same.json
//
// MODE 0: lane L reads 16 bytes at a random 128-byte-aligned node chosen by L / GROUP (GROUP adjacent lanes read the same address)
// MODE 1: ... chosen by L % (64 / GROUP) (the GROUP lanes that share an address are spread over the wavefront)
// ACTIVE: only lanes < ACTIVE load.  16 loads in flight per lane, 5 wavefronts per SIMD on every CU, table of 16 KiB (L1 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1; };

template <int GROUP, int MODE, int ACTIVE>
__global__ void __launch_bounds__(256) k_fetch(Stamp *out, int iters, const char *table, unsigned node_mask, float *sink)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned who = wave * 64u + (MODE == 0 ? lane / GROUP : lane % (64 / GROUP));
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        float4 r4[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            unsigned h = (who * 2654435761u) ^ ((unsigned)(i * 16 + k) * 2246822519u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const char *p = table + (size_t)(h & node_mask) * 128;
            r4[k] = make_float4(0, 0, 0, 0);
            if (lane < ACTIVE) r4[k] = *(const float4 *)p;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc += r4[k].x + r4[k].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { Stamp s; s.t0 = t0; s.t1 = t1; out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int GROUP, int MODE, int ACTIVE>
static void run(bool &first, int n_cu, Stamp *d_out, const char *d_table, unsigned mask, float *d_sink)
{
    const int W = 5, blocks = n_cu * W, iters = 512;
    const size_t lds = 30 * 1024;                       // five workgroups per CU, like the walk
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_fetch<GROUP, MODE, ACTIVE>), dim3(blocks), dim3(256), lds, 0, d_out, iters, d_table, mask, d_sink);
        CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
    }
    std::vector<Stamp> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    for (const Stamp &s : h) cyc.push_back((double)(s.t1 - s.t0));
    std::sort(cyc.begin(), cyc.end());
    const double cyc_per_load_cu = cyc[cyc.size() / 2] / ((double)iters * 16.0) / (4.0 * W);
    printf("%s\n  {\"lanes_sharing_an_address\": %d, \"sharing_lanes\": \"%s\", \"active_lanes\": %d, \"distinct_addresses_per_wave_load\": %d, \"cycles_per_wave_load_per_cu\": %.2f, \"cycles_per_active_lane\": %.3f}",
           first ? "" : ",", GROUP, MODE == 0 ? "adjacent" : "spread", ACTIVE, (ACTIVE + GROUP - 1) / GROUP < 64 / GROUP ? (MODE == 0 ? (ACTIVE + GROUP - 1) / GROUP : std::min(ACTIVE, 64 / GROUP)) : 64 / GROUP,
           cyc_per_load_cu, cyc_per_load_cu / ACTIVE);
    first = false; fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    Stamp *d_out = nullptr; char *d_table = nullptr; float *d_sink = nullptr;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMalloc(&d_table, (size_t)128 * 128 + 4096));
    CHECK(hipMemset(d_table, 0, (size_t)128 * 128 + 4096));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"waves_per_simd\": 5, \"table\": \"16 KiB (L1)\", \"bytes_per_lane\": 16, \"results\": [", prop.gcnArchName, n_cu);
    bool first = true;
    const unsigned mask = (1u << 7) - 1u;
    run<1, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<2, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<4, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<8, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<16, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<64, 0, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<2, 1, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<4, 1, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<8, 1, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<16, 1, 64>(first, n_cu, d_out, d_table, mask, d_sink);
    run<1, 0, 48>(first, n_cu, d_out, d_table, mask, d_sink);
    run<1, 0, 32>(first, n_cu, d_out, d_table, mask, d_sink);
    run<1, 0, 16>(first, n_cu, d_out, d_table, mask, d_sink);
    run<4, 1, 48>(first, n_cu, d_out, d_table, mask, d_sink);
    printf("\n]}\n");
    return 0;
}
