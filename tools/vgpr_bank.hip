// vgpr_bank.hip -- round 6 micro-benchmark: does the VALU issue rate of gfx950 depend on WHICH vector registers an instruction reads?
// tools/valu_roof.hip (round 2/4) found 0.24-0.29 wave-instructions per cycle and SIMD for independent fp32 streams however many wavefronts share the
// SIMD, but 0.44 for a dependent v_fma_f32 chain of four wavefronts -- the same instruction, other registers.  Here every pattern names its registers:
//   hipcc --offload-arch=gfx950 -O2 -o build/vgpr_bank tools/vgpr_bank.hip && build/vgpr_bank      -> profiles/round6/vgpr_bank.json
// Every wavefront runs ITERS x 64 instructions of one pattern (explicit VGPR numbers, inline asm), stamped with s_memtime; W workgroups of four
// wavefronts per CU (LDS-sized), so W wavefronts per SIMD; reported: wave-instructions per cycle and SIMD (median over the SIMDs).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Stamp { unsigned long long t0, t1; unsigned hw_id, xcc_id; };

#define CLOB "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"
// eight instructions of pattern P with destinations D0..D7 (register NUMBERS as strings)
#define I8(OP, S, D0, D1, D2, D3, D4, D5, D6, D7) OP " v" D0 S(D0) "\n" OP " v" D1 S(D1) "\n" OP " v" D2 S(D2) "\n" OP " v" D3 S(D3) "\n" OP " v" D4 S(D4) "\n" OP " v" D5 S(D5) "\n" OP " v" D6 S(D6) "\n" OP " v" D7 S(D7) "\n"
#define B64(X) X X X X X X X X
// source lists (D = the destination's number, for accumulate forms)
#define S_ACC_1_2(D) ", v" D ", v1, v2"
#define S_ACC_1_5(D) ", v" D ", v1, v5"
#define S_ACC_S_1(D) ", v" D ", s2, v1"
#define S_ACC_C_1(D) ", v" D ", 2.0, v1"
#define S_1_2_3(D) ", v1, v2, v3"
#define S_1_2_6(D) ", v1, v2, v6"
#define S_1_5_5(D) ", v1, v5, v5"
#define S_1_1_1(D) ", v1, v1, v1"
#define S2_1_2(D) ", v1, v2"
#define S2_1_5(D) ", v1, v5"
#define S2_ACC_1(D) ", v" D ", v1"
#define S2_ACC_C(D) ", 1.0, v" D

enum { P_FMA_ACC_CYC, P_FMA_ACC_BANK0, P_FMA_ACC_BANK1, P_FMA_ACC_SAMEBANK_SRC, P_FMA_ACC_SGPR, P_FMA_ACC_CONST, P_FMA_123, P_FMA_126, P_FMA_155, P_FMA_111,
       P_ADD_12, P_ADD_15, P_ADD_ACC, P_ADD_ACC_CONST, P_MIN3_123, P_MIN3_126, P_MAX_12, P_MUL64, P_FMAMIX_123, N_PAT };
static const char *pat_name[N_PAT] = {
    "v_fma_f32 vA, vA, v1, v2   A = 8..15 (accumulators in all four banks, if bank = number mod 4)",
    "v_fma_f32 vA, vA, v1, v2   A = 8,12,..,36 (all in bank 0)",
    "v_fma_f32 vA, vA, v1, v2   A = 9,13,..,37 (all in v1's bank)",
    "v_fma_f32 vA, vA, v1, v5   A = 8,12,..,36 (the two fixed sources share a bank)",
    "v_fma_f32 vA, vA, s2, v1   A = 8,12,..,36 (one source scalar)",
    "v_fma_f32 vA, vA, 2.0, v1  A = 8,12,..,36 (one source an inline constant)",
    "v_fma_f32 vA, v1, v2, v3   A = 8..15 (three sources, three banks, independent of the destination)",
    "v_fma_f32 vA, v1, v2, v6   A = 8..15 (two of three sources share a bank)",
    "v_fma_f32 vA, v1, v5, v5   A = 8..15 (one register twice + one of the same bank)",
    "v_fma_f32 vA, v1, v1, v1   A = 8..15 (one register three times)",
    "v_add_f32 vA, v1, v2       A = 8..15",
    "v_add_f32 vA, v1, v5       A = 8..15 (sources share a bank)",
    "v_add_f32 vA, vA, v1       A = 8,12,..,36",
    "v_add_f32 vA, 1.0, vA      A = 8,12,..,36 (one register source)",
    "v_min3_f32 vA, v1, v2, v3  A = 8..15",
    "v_min3_f32 vA, v1, v2, v6  A = 8..15",
    "v_max_f32 vA, v1, v2       A = 8..15",
    "v_mul_f64 v[A:A+1], v[2:3], v[4:5]  A = 8,10,..,22",
    "v_fma_mix_f32 vA, v1, v2, v3 op_sel_hi:[1,0,0]  A = 8..15 (the walk's plane distance)" };

template <int P>
__global__ void __launch_bounds__(256) k_bank(Stamp *out, int iters)
{
    extern __shared__ char lds_[];
    asm volatile("v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 2.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 2.0\n v_mov_b32 v7, 1.0\n s_mov_b32 s2, 1.0\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"
                 "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n" ::: CLOB, "s2");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (P == P_FMA_ACC_CYC) asm volatile(B64(I8("v_fma_f32", S_ACC_1_2, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_FMA_ACC_BANK0) asm volatile(B64(I8("v_fma_f32", S_ACC_1_2, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_FMA_ACC_BANK1) asm volatile(B64(I8("v_fma_f32", S_ACC_1_2, "9", "13", "17", "21", "25", "29", "33", "37")) ::: CLOB);
        if (P == P_FMA_ACC_SAMEBANK_SRC) asm volatile(B64(I8("v_fma_f32", S_ACC_1_5, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_FMA_ACC_SGPR) asm volatile(B64(I8("v_fma_f32", S_ACC_S_1, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_FMA_ACC_CONST) asm volatile(B64(I8("v_fma_f32", S_ACC_C_1, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_FMA_123) asm volatile(B64(I8("v_fma_f32", S_1_2_3, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_FMA_126) asm volatile(B64(I8("v_fma_f32", S_1_2_6, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_FMA_155) asm volatile(B64(I8("v_fma_f32", S_1_5_5, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_FMA_111) asm volatile(B64(I8("v_fma_f32", S_1_1_1, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_ADD_12) asm volatile(B64(I8("v_add_f32", S2_1_2, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_ADD_15) asm volatile(B64(I8("v_add_f32", S2_1_5, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_ADD_ACC) asm volatile(B64(I8("v_add_f32", S2_ACC_1, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_ADD_ACC_CONST) asm volatile(B64(I8("v_add_f32", S2_ACC_C, "8", "12", "16", "20", "24", "28", "32", "36")) ::: CLOB);
        if (P == P_MIN3_123) asm volatile(B64(I8("v_min3_f32", S_1_2_3, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_MIN3_126) asm volatile(B64(I8("v_min3_f32", S_1_2_6, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_MAX_12) asm volatile(B64(I8("v_max_f32", S2_1_2, "8", "9", "10", "11", "12", "13", "14", "15")) ::: CLOB);
        if (P == P_MUL64) asm volatile(B64("v_mul_f64 v[8:9], v[2:3], v[4:5]\n v_mul_f64 v[10:11], v[2:3], v[4:5]\n v_mul_f64 v[12:13], v[2:3], v[4:5]\n v_mul_f64 v[14:15], v[2:3], v[4:5]\n"
                                           "v_mul_f64 v[16:17], v[2:3], v[4:5]\n v_mul_f64 v[18:19], v[2:3], v[4:5]\n v_mul_f64 v[20:21], v[2:3], v[4:5]\n v_mul_f64 v[22:23], v[2:3], v[4:5]\n") ::: CLOB);
        if (P == P_FMAMIX_123) asm volatile(B64("v_fma_mix_f32 v8, v1, v2, v3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v9, v1, v2, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v10, v1, v2, v3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v11, v1, v2, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                                                "v_fma_mix_f32 v12, v1, v2, v3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v13, v1, v2, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v14, v1, v2, v3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v15, v1, v2, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { Stamp s; s.t0 = t0; s.t1 = t1; s.hw_id = hw; s.xcc_id = xcc; out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; }
    if (iters < 0) lds_[threadIdx.x] = 1;
}

template <int P>
static void run(bool &first, int n_cu, Stamp *d_out, int iters)
{
    for (int W : { 1, 2, 4 }) {
        const int blocks = n_cu * W;
        size_t lds = (size_t)(160 * 1024) / (size_t)W;
        lds -= lds / (size_t)(2 * (W + 1));
        lds &= ~(size_t)1023;
        CHECK(hipFuncSetAttribute((const void *)k_bank<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_bank<P>, dim3(blocks), dim3(256), lds, 0, d_out, iters); CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize()); }
        std::vector<Stamp> h((size_t)blocks * 4);
        CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        struct Agg { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; };
        std::map<unsigned long long, Agg> simd;
        for (const Stamp &s : h) {
            const unsigned long long key = ((unsigned long long)s.xcc_id << 32) | (s.hw_id & 0xfffffff0u & ~0x000000c0u & 0x00ffffffu & ~0x00ff0000u);
            Agg &a = simd[key]; a.t0 = std::min(a.t0, s.t0); a.t1 = std::max(a.t1, s.t1); a.n++;
        }
        std::vector<double> ipc; int wmin = 1 << 30, wmax = 0;
        for (auto &kv : simd) { ipc.push_back(kv.second.n * (double)iters * 64.0 / (double)(kv.second.t1 - kv.second.t0)); wmin = std::min(wmin, kv.second.n); wmax = std::max(wmax, kv.second.n); }
        std::sort(ipc.begin(), ipc.end());
        printf("%s\n  {\"pattern\": \"%s\", \"waves_per_simd\": %d, \"simd_ipc\": %.4f, \"cycles_per_instruction\": %.2f, \"waves_per_simd_seen\": [%d, %d]}", first ? "" : ",", pat_name[P], W, ipc[ipc.size() / 2], 1.0 / ipc[ipc.size() / 2], wmin, wmax);
        first = false; fflush(stdout);
    }
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, iters = 4096;
    Stamp *d_out = nullptr; CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)n_cu * 8 * 4));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"instructions_per_wave\": %d, \"results\": [", prop.gcnArchName, n_cu, iters * 64);
    bool first = true;
    run<P_FMA_ACC_CYC>(first, n_cu, d_out, iters); run<P_FMA_ACC_BANK0>(first, n_cu, d_out, iters); run<P_FMA_ACC_BANK1>(first, n_cu, d_out, iters); run<P_FMA_ACC_SAMEBANK_SRC>(first, n_cu, d_out, iters);
    run<P_FMA_ACC_SGPR>(first, n_cu, d_out, iters); run<P_FMA_ACC_CONST>(first, n_cu, d_out, iters); run<P_FMA_123>(first, n_cu, d_out, iters); run<P_FMA_126>(first, n_cu, d_out, iters);
    run<P_FMA_155>(first, n_cu, d_out, iters); run<P_FMA_111>(first, n_cu, d_out, iters); run<P_ADD_12>(first, n_cu, d_out, iters); run<P_ADD_15>(first, n_cu, d_out, iters);
    run<P_ADD_ACC>(first, n_cu, d_out, iters); run<P_ADD_ACC_CONST>(first, n_cu, d_out, iters); run<P_MIN3_123>(first, n_cu, d_out, iters); run<P_MIN3_126>(first, n_cu, d_out, iters);
    run<P_MAX_12>(first, n_cu, d_out, iters); run<P_MUL64>(first, n_cu, d_out, iters); run<P_FMAMIX_123>(first, n_cu, d_out, iters);
    printf("\n]}\n");
    return 0;
}
