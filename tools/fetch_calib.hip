// fetch_calib.hip -- what FETCH_SIZE reports for the WALK's access shapes on gfx950 (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count
// in your own access pattern before trusting an absolute").  Every kernel reads a KNOWN number of bytes from a 4 GiB table (far past the
// 256 MiB Infinity Cache), each address at most once per launch:
//   k_stream   coalesced, 16 bytes per lane, consecutive lanes consecutive addresses            (the guide's case: expect 1/2)
//   k_node64   every lane reads ONE random 64-byte node as 4 x 16 bytes                          (k_trace_lane's node fetch)
//   k_tri96    every lane reads ONE random 96-byte triangle record (96-byte stride) as 6 x 16    (its leaf fetch)
//   k_gather8  every lane reads ONE random 8-byte texture cell                                   (k_march's gather)
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- build/fetch_calib ; rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ... ; the tool prints the bytes it asked for
//   hipcc --offload-arch=gfx950 -O2 -o build/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned long long mix(unsigned long long i, unsigned long long n)     // a bijection on [0, n), n a power of two: no address twice
{
    i = (i * 0x9E3779B97F4A7C15ull) & (n - 1ull);            // odd multiplier: a permutation of the low bits
    i ^= i >> 17; i &= (n - 1ull);                          // (xor-shift keeps it a bijection on n = 2^k for shifts < k)
    i = (i * 0xD1B54A32D192ED03ull) & (n - 1ull);
    return i;
}
__global__ void k_stream(const float4 *t, size_t n16, float *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0.f;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_node64(const float4 *t, unsigned long long n_nodes_table, size_t n_reads, float *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0.f;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) {
        const float4 *N = t + 4 * mix(i, n_nodes_table);
        const float4 a = N[0], b = N[1], c = N[2], d = N[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_tri96(const float4 *t, unsigned long long n_recs_table, size_t n_reads, float *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0.f;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) {
        const float4 *T = t + 6 * mix(i, n_recs_table);
        const float4 a = T[0], b = T[1], c = T[2], d = T[3], e = T[4], f = T[5];
        acc += a.x + b.y + c.z + d.w + e.x + f.y;
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_gather8(const float2 *t, unsigned long long n_cells_table, size_t n_reads, float *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0.f;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) { const float2 v = t[mix(i, n_cells_table)]; acc += v.x + v.y; }
    if (acc == 12345.678f) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;
    char *tab = nullptr; float *sink = nullptr;
    CHECK(hipMalloc(&tab, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(tab, 0, bytes)); CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), blk(256);
    const size_t n_stream16 = ((size_t)1 << 30) / 16;                 // 1 GiB streamed
    const size_t n_reads = (size_t)1 << 24;                           // 16 M random nodes / records / cells
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream, grid, blk, 0, 0, (const float4 *)tab + ((size_t)rep << 26), n_stream16, sink);
        hipLaunchKernelGGL(k_node64, grid, blk, 0, 0, (const float4 *)tab, (unsigned long long)(bytes / 64), n_reads, sink);
        hipLaunchKernelGGL(k_tri96, grid, blk, 0, 0, (const float4 *)tab, (unsigned long long)1 << 25, n_reads, sink);       // 2^25 records x 96 B = 3 GiB
        hipLaunchKernelGGL(k_gather8, grid, blk, 0, 0, (const float2 *)tab, (unsigned long long)(bytes / 8), n_reads, sink);
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"asked_bytes\": {\"k_stream\": %zu, \"k_node64\": %zu, \"k_tri96\": %zu, \"k_gather8\": %zu}, \"reads\": %zu, \"note\": \"per launch; every kernel launched twice\"}\n",
           n_stream16 * 16, n_reads * 64, n_reads * 96, n_reads * 8, n_reads);
    return 0;
}
