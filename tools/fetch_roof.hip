// fetch_roof.hip -- what a wave-level global load costs the vector memory pipeline (TA/TCP) of an MI355X compute unit,
// by access SHAPE: how many lanes share a contiguous run, and how wide each lane's load is.  The BVH walk's loads are
// scattered (rays of a wavefront sit on different nodes); this tool prices the candidate node-fetch layouts.
//
//   hipcc --offload-arch=gfx950 -O2 -o build/fetch_roof tools/fetch_roof.hip && build/fetch_roof > fetch_roof.json
//
// Every wavefront issues ITER x 16 independent loads (16 in flight), 5 wavefronts per SIMD on every CU.  For a shape
// (GROUP lanes x BYTES each): lane L reads BYTES at  node(hash(L / GROUP, step)) * 128 + (L % GROUP) * BYTES  (+ run offset),
// i.e. GROUP consecutive lanes read one contiguous run of GROUP*BYTES bytes starting at a random 128-byte-aligned node.
// Reported: shader cycles per wave-level load per CU (= 20 waves' loads serialised through the CU's one TCP), bytes per cycle
// per CU, and the aggregate rate.  Tables of 16 KiB (L1), 2 MiB (L2), 64 MiB (Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int GROUP, int BYTES>
__global__ void __launch_bounds__(256) k_fetch(Stamp *out, int iters, const char *table, unsigned node_mask, float *sink)
{
    extern __shared__ char lds_[];
    const unsigned lane = threadIdx.x & 63, who = (blockIdx.x * 256 + threadIdx.x) / GROUP, off = (threadIdx.x % GROUP) * BYTES;
    float acc = 0.0f;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        float4 r4[16]; float2 r2[16]; float r1[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            unsigned h = (who * 2654435761u) ^ ((unsigned)(i * 16 + k) * 2246822519u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const char *p = table + (size_t)(h & node_mask) * 128 + off;
            if (BYTES == 16) r4[k] = *(const float4 *)p;
            else if (BYTES == 8) r2[k] = *(const float2 *)p;
            else r1[k] = *(const float *)p;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc += (BYTES == 16) ? r4[k].x + r4[k].w : (BYTES == 8) ? r2[k].x + r2[k].y : r1[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { Stamp s; s.t0 = t0; s.t1 = t1; s.r0 = r0; s.r1 = r1_; out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; }
    if (acc == 12345.678f) { sink[0] = acc; lds_[0] = 1; }
}

template <int GROUP, int BYTES>
static void run(bool &first, int n_cu, Stamp *d_out, const char *d_table, unsigned mask, const char *where, float *d_sink)
{
    const int W = 5, blocks = n_cu * W, iters = 512;
    size_t lds = (size_t)(160 * 1024) / W; lds -= lds / (2 * (W + 1)); lds &= ~(size_t)1023;
    CHECK(hipFuncSetAttribute((const void *)k_fetch<GROUP, BYTES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_fetch<GROUP, BYTES>), dim3(blocks), dim3(256), lds, 0, d_out, iters, d_table, mask, d_sink);
        CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
    }
    std::vector<Stamp> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    unsigned long long rmin = ~0ull, rmax = 0; std::vector<double> cyc, clk;
    for (const Stamp &s : h) { rmin = std::min(rmin, s.r0); rmax = std::max(rmax, s.r1); cyc.push_back((double)(s.t1 - s.t0)); clk.push_back((double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double wave_cycles = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];
    const double loads_per_wave = (double)iters * 16.0;
    const double cyc_per_load_cu = wave_cycles / loads_per_wave / (4.0 * W);          // 4*W waves share the CU's TCP
    const double bytes_per_load = 64.0 * BYTES;
    const double span_s = (double)(rmax - rmin) * 1e-8;
    printf("%s\n  {\"lanes_per_run\": %d, \"bytes_per_lane\": %d, \"table\": \"%s\", \"cycles_per_wave_load_per_cu\": %.2f, \"bytes_per_cycle_per_cu\": %.2f, "
           "\"aggregate_TBps\": %.3f, \"clock_ghz\": %.3f, \"distinct_lines_per_wave_load\": %d}",
           first ? "" : ",", GROUP, BYTES, where, cyc_per_load_cu, bytes_per_load / cyc_per_load_cu, (double)blocks * 4 * loads_per_wave * bytes_per_load / span_s / 1e12, ghz,
           (GROUP * BYTES >= 128) ? 64 * BYTES / 128 : 64 / GROUP);
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
    const size_t max_nodes = (size_t)1 << 19;
    CHECK(hipMalloc(&d_table, max_nodes * 128 + 4096));
    CHECK(hipMemset(d_table, 0, max_nodes * 128 + 4096));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"waves_per_simd\": 5, \"results\": [", prop.gcnArchName, n_cu);
    bool first = true;
    const unsigned masks[3] = { (1u << 7) - 1u, (1u << 14) - 1u, (1u << 19) - 1u };
    const char *names[3] = { "16 KiB (L1)", "2 MiB (L2)", "64 MiB (Infinity Cache)" };
    for (int t = 0; t < 3; t++) {
        run<1, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);     // every lane its own line (one lane per ray, SoA node)
        run<2, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);
        run<4, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);     // a quad reads 64 contiguous bytes
        run<8, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);     // eight lanes read one whole 128-byte line
        run<16, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);
        run<64, 16>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);    // the wavefront reads 1 KiB contiguous
        run<1, 8>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);
        run<1, 4>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);
        run<16, 8>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);     // sixteen lanes x 8 B = one line
        run<32, 4>(first, n_cu, d_out, d_table, masks[t], names[t], d_sink);     // thirty-two lanes x 4 B = one line
    }
    printf("\n]}\n");
    return 0;
}
