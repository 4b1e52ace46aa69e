// quad_line.hip -- what a wavefront's node fetch costs the vector memory pipe of an MI355X CU, by WHO fetches WHAT (round 4).
// tools/fetch_roof_same.hip found that four ADJACENT lanes on one address cost the pipe a third of four lanes on four addresses: the pipe
// works through a 16-byte-per-lane load a quad of lanes at a time.  The walk's node fetch is the worst case of that: every lane reads the
// four 16-byte pieces of ITS OWN 64-byte node, so each of the four load instructions finds four different lines in every quad.
// Three ways to bring 64 nodes (one per lane, random, 64-byte aligned) to 64 lanes, as a DEPENDENT chain (the next node comes out of the
// loaded one, as in the walk), 4 wavefronts per SIMD on every CU:
//   MODE 0  lane-per-node: 4 x global_load_dwordx4 at node + 0/16/32/48 (the walk today)
//   MODE 1  quad-transposed: in load k, lane 4q+j reads piece j of the node of lane 4q+k -- a quad reads ONE line per load; the lane then
//           holds piece j of four nodes (no transpose back: the pipe's cost only)
//   MODE 3  lane-per-node through LDS: the walk's four loads as global_load_lds_dwordx4 (piece p of every lane's node lands in image p), then
//           4 x ds_read_b128 -- no quad logic at all: separates WHERE the data lands from WHO shares a line
//   MODE 2  quad-transposed through LDS: the same four loads as global_load_lds_dwordx4 (destination = base_k + lane x 16: the quad's line lands
//           as one contiguous node), then lane 4q+k reads its node with 4 x ds_read_b128
//
//   hipcc --offload-arch=gfx950 -O2 -o build/quad_line tools/quad_line.hip && build/quad_line > quad_line.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1; };
#define PITCH 1040          // bytes between the LDS images of the four loads of a wavefront (1 KiB + 16: lanes 4q+k of a quad then read different banks)

__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h; }

template <int MODE>
__global__ void __launch_bounds__(256) k_fetch(Stamp *out, int iters, const char *table, unsigned node_mask, unsigned *sink)
{
    __shared__ __attribute__((aligned(16))) char stage[4 * 4 * PITCH + 32 * 1024 - 4 * 4 * PITCH];      // 32 KiB per workgroup: four per CU, like the walk
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned who = wave * 64u + lane;
    unsigned node = mix(who * 2654435761u) & node_mask, acc = 0;
    const unsigned j = lane & 3u;
    char *my_stage = stage + wv * 4 * PITCH;
    const uint4 *my_node = (const uint4 *)(my_stage + (lane & 3u) * PITCH + (lane >> 2) * 64);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        uint4 Q0, Q1, Q2, Q3;
        if (MODE == 0) {
            const uint4 *N = (const uint4 *)(table + ((size_t)node << 6));
            Q0 = N[0]; Q1 = N[1]; Q2 = N[2]; Q3 = N[3];
        } else if (MODE == 3) {
            const char *N = table + ((size_t)node << 6);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(N), (__attribute__((address_space(3))) void *)(my_stage + 0 * PITCH), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(N + 16), (__attribute__((address_space(3))) void *)(my_stage + 1 * PITCH), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(N + 32), (__attribute__((address_space(3))) void *)(my_stage + 2 * PITCH), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(N + 48), (__attribute__((address_space(3))) void *)(my_stage + 3 * PITCH), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char *mine = my_stage + lane * 16;
            Q0 = *(const uint4 *)(mine); Q1 = *(const uint4 *)(mine + PITCH); Q2 = *(const uint4 *)(mine + 2 * PITCH); Q3 = *(const uint4 *)(mine + 3 * PITCH);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // the nodes of the quad's four lanes (DPP quad broadcast)
            const unsigned n0 = (unsigned)__builtin_amdgcn_mov_dpp((int)node, 0x00, 0xf, 0xf, true), n1 = (unsigned)__builtin_amdgcn_mov_dpp((int)node, 0x55, 0xf, 0xf, true);
            const unsigned n2 = (unsigned)__builtin_amdgcn_mov_dpp((int)node, 0xaa, 0xf, 0xf, true), n3 = (unsigned)__builtin_amdgcn_mov_dpp((int)node, 0xff, 0xf, 0xf, true);
            const char *p0 = table + ((size_t)n0 << 6) + j * 16, *p1 = table + ((size_t)n1 << 6) + j * 16, *p2 = table + ((size_t)n2 << 6) + j * 16, *p3 = table + ((size_t)n3 << 6) + j * 16;
            if (MODE == 1) {
                Q0 = *(const uint4 *)p0; Q1 = *(const uint4 *)p1; Q2 = *(const uint4 *)p2; Q3 = *(const uint4 *)p3;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p0, (__attribute__((address_space(3))) void *)(my_stage + 0 * PITCH), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p1, (__attribute__((address_space(3))) void *)(my_stage + 1 * PITCH), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p2, (__attribute__((address_space(3))) void *)(my_stage + 2 * PITCH), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p3, (__attribute__((address_space(3))) void *)(my_stage + 3 * PITCH), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                Q0 = my_node[0]; Q1 = my_node[1]; Q2 = my_node[2]; Q3 = my_node[3];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        const unsigned got = Q0.x ^ Q1.y ^ Q2.z ^ Q3.w;         // (the table holds zeros: the chain is real, the walk stays random)
        acc += got;
        node = (mix(who * 2654435761u ^ (unsigned)(i + 1) * 2246822519u) ^ got) & node_mask;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { Stamp s; s.t0 = t0; s.t1 = t1; out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; }
    if (acc == 12345u) sink[0] = acc + (unsigned)stage[threadIdx.x];
}

template <int MODE>
static void run(bool &first, int n_cu, Stamp *d_out, const char *d_table, unsigned mask, const char *where, unsigned *d_sink)
{
    const int W = 4, blocks = n_cu * W, iters = 2048;
    double ms = 0;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_fetch<MODE>), dim3(blocks), dim3(256), 0, 0, d_out, iters, d_table, mask, d_sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms = t;
    }
    std::vector<Stamp> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    for (const Stamp &s : h) cyc.push_back((double)(s.t1 - s.t0));
    std::sort(cyc.begin(), cyc.end());
    // every wavefront of the launch is resident at once (4 workgroups per CU): launch time / iterations = one dependent fetch of a wavefront,
    // with 15 other wavefronts of its CU doing the same
    const double ns_per_fetch_wave = ms * 1e6 / iters;
    static const char *names[4] = { "lane-per-node (the walk today)", "quad-transposed, to registers", "quad-transposed, global_load_lds + 4 ds_read_b128", "lane-per-node, global_load_lds + 4 ds_read_b128" };
    printf("%s\n  {\"mode\": \"%s\", \"table\": \"%s\", \"ns_per_dependent_fetch_of_a_wavefront\": %.1f, \"ns_per_wavefront_fetch_per_cu\": %.2f, \"launch_ms\": %.3f}",
           first ? "" : ",", names[MODE], where, ns_per_fetch_wave, ns_per_fetch_wave / (4.0 * W), ms);
    first = false; fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    Stamp *d_out = nullptr; char *d_table = nullptr; unsigned *d_sink = nullptr;
    const size_t table_bytes = (size_t)1 << 30;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMalloc(&d_table, table_bytes));
    CHECK(hipMemset(d_table, 0, table_bytes));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"waves_per_simd\": 4, \"node_bytes\": 64, \"results\": [", prop.gcnArchName, n_cu);
    bool first = true;
    struct { unsigned mask; const char *where; } T[] = { { (1u << 8) - 1u, "16 KiB (L1)" }, { (1u << 15) - 1u, "2 MiB (L2)" }, { (1u << 19) - 1u, "32 MiB (the walk's tree: L2 + Infinity Cache)" }, { (1u << 24) - 1u, "1 GiB (HBM)" } };
    for (auto &t : T) {
        run<0>(first, n_cu, d_out, d_table, t.mask, t.where, d_sink);
        run<1>(first, n_cu, d_out, d_table, t.mask, t.where, d_sink);
        run<2>(first, n_cu, d_out, d_table, t.mask, t.where, d_sink);
        run<3>(first, n_cu, d_out, d_table, t.mask, t.where, d_sink);
    }
    printf("\n]}\n");
    return 0;
}
