// issue_pairs.hip -- round 6 micro-benchmark (same harness as valu_classes.hip): do the four-cycle classes cost four cycles between two-cycle instructions?: issue cycles per wave64 instruction by instruction CLASS and operand KIND
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

#define CLOB "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "vcc", "s2", "s4", "s5"
#define B64(X) X X X X X X X X
static const char *pat_name[] = {
    "v_cndmask_b32_e64 vA, v1, v2, s[4:5] ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_mul_f32 vA, s2, v1 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_min_f32 vA, v1, v2 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_fma_mix_f32 vA, v1, v2, v3 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_cndmask_b32 vA, v1, v2, vcc ; v_fma_mix_f32 vB, v1, v2, v3 (per pair)",
    "v_cndmask_b32 vA, v1, v2, vcc ; v_min_f32 vB, v1, v2 (per pair)",
    "v_cndmask_b32 vA, v1, v2, vcc ; v_fma_mix_f32 vB, vA, v2, v3 (dependent; per pair)",
    "s_mov_b64 vcc, s[4:5] ; v_cndmask_b32 vA, v1, v2, vcc ; v_mul_f32 vB, v1, v2 (per triple)",
    "v_min_f32 vA, v1, v2 ; v_max_f32 vB, v1, v2 (per pair)",
    "v_cmp_lt_f32 vcc, v1, v2 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_lshlrev_b32 vA, 3, v1 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_add_f64 v[A:A+1], v[2:3], v[4:5] ; v_mul_f32 vB, v1, v2 (per pair; 4 pairs per block)",
    "v_cndmask_b32 vA, v1, v2, vcc ; v_add_f64 (per pair; 4 pairs per block)",
    "v_bitop3_b32 vA, v1, v2, v3 bitop3:0x48",
    "v_bitop3_b32 vA, v1, v2, v3 bitop3:0x48 ; v_mul_f32 vB, v1, v2 (per pair)",
    "v_xor_b32 vA, v1, v2 ; v_and_b32 vB, v1, v2 (per pair)",
    "v_min_f32 vA, v1, v2 ; v_xor_b32 vB, v1, v2 ; v_and_b32 vC, v1, v2 (per triple)" };
constexpr int N_PAT = 17;
template <int P>
__global__ void __launch_bounds__(256) k_cls(Stamp *out, int iters)
{
    extern __shared__ char lds_[];
    asm volatile("v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 2.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 2.0\n v_mov_b32 v7, 1.0\n s_mov_b32 s2, 1.0\n s_mov_b64 vcc, exec\n s_mov_b64 s[4:5], exec\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" ::: CLOB);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (P == 0) { asm volatile(B64("v_cndmask_b32_e64 v8, v1, v2, s[4:5]\nv_mul_f32 v16, v1, v2\nv_cndmask_b32_e64 v9, v1, v2, s[4:5]\nv_mul_f32 v17, v1, v2\nv_cndmask_b32_e64 v10, v1, v2, s[4:5]\nv_mul_f32 v18, v1, v2\nv_cndmask_b32_e64 v11, v1, v2, s[4:5]\nv_mul_f32 v19, v1, v2\nv_cndmask_b32_e64 v12, v1, v2, s[4:5]\nv_mul_f32 v20, v1, v2\nv_cndmask_b32_e64 v13, v1, v2, s[4:5]\nv_mul_f32 v21, v1, v2\nv_cndmask_b32_e64 v14, v1, v2, s[4:5]\nv_mul_f32 v22, v1, v2\nv_cndmask_b32_e64 v15, v1, v2, s[4:5]\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 1) { asm volatile(B64("v_mul_f32 v8, s2, v1\nv_mul_f32 v16, v1, v2\nv_mul_f32 v9, s2, v1\nv_mul_f32 v17, v1, v2\nv_mul_f32 v10, s2, v1\nv_mul_f32 v18, v1, v2\nv_mul_f32 v11, s2, v1\nv_mul_f32 v19, v1, v2\nv_mul_f32 v12, s2, v1\nv_mul_f32 v20, v1, v2\nv_mul_f32 v13, s2, v1\nv_mul_f32 v21, v1, v2\nv_mul_f32 v14, s2, v1\nv_mul_f32 v22, v1, v2\nv_mul_f32 v15, s2, v1\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 2) { asm volatile(B64("v_min_f32 v8, v1, v2\nv_mul_f32 v16, v1, v2\nv_min_f32 v9, v1, v2\nv_mul_f32 v17, v1, v2\nv_min_f32 v10, v1, v2\nv_mul_f32 v18, v1, v2\nv_min_f32 v11, v1, v2\nv_mul_f32 v19, v1, v2\nv_min_f32 v12, v1, v2\nv_mul_f32 v20, v1, v2\nv_min_f32 v13, v1, v2\nv_mul_f32 v21, v1, v2\nv_min_f32 v14, v1, v2\nv_mul_f32 v22, v1, v2\nv_min_f32 v15, v1, v2\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 3) { asm volatile(B64("v_fma_mix_f32 v8, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v16, v1, v2\nv_fma_mix_f32 v9, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v17, v1, v2\nv_fma_mix_f32 v10, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v18, v1, v2\nv_fma_mix_f32 v11, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v19, v1, v2\nv_fma_mix_f32 v12, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v20, v1, v2\nv_fma_mix_f32 v13, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v21, v1, v2\nv_fma_mix_f32 v14, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v22, v1, v2\nv_fma_mix_f32 v15, v1, v2, v3 op_sel_hi:[1,0,0]\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 4) { asm volatile(B64("v_cndmask_b32 v8, v1, v2, vcc\nv_fma_mix_f32 v16, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v9, v1, v2, vcc\nv_fma_mix_f32 v17, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v10, v1, v2, vcc\nv_fma_mix_f32 v18, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v11, v1, v2, vcc\nv_fma_mix_f32 v19, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v12, v1, v2, vcc\nv_fma_mix_f32 v20, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v13, v1, v2, vcc\nv_fma_mix_f32 v21, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v14, v1, v2, vcc\nv_fma_mix_f32 v22, v1, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v15, v1, v2, vcc\nv_fma_mix_f32 v23, v1, v2, v3 op_sel_hi:[1,0,0]\n") ::: CLOB); }
        if (P == 5) { asm volatile(B64("v_cndmask_b32 v8, v1, v2, vcc\nv_min_f32 v16, v1, v2\nv_cndmask_b32 v9, v1, v2, vcc\nv_min_f32 v17, v1, v2\nv_cndmask_b32 v10, v1, v2, vcc\nv_min_f32 v18, v1, v2\nv_cndmask_b32 v11, v1, v2, vcc\nv_min_f32 v19, v1, v2\nv_cndmask_b32 v12, v1, v2, vcc\nv_min_f32 v20, v1, v2\nv_cndmask_b32 v13, v1, v2, vcc\nv_min_f32 v21, v1, v2\nv_cndmask_b32 v14, v1, v2, vcc\nv_min_f32 v22, v1, v2\nv_cndmask_b32 v15, v1, v2, vcc\nv_min_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 6) { asm volatile(B64("v_cndmask_b32 v8, v1, v2, vcc\nv_fma_mix_f32 v16, v8, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v9, v1, v2, vcc\nv_fma_mix_f32 v17, v9, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v10, v1, v2, vcc\nv_fma_mix_f32 v18, v10, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v11, v1, v2, vcc\nv_fma_mix_f32 v19, v11, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v12, v1, v2, vcc\nv_fma_mix_f32 v20, v12, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v13, v1, v2, vcc\nv_fma_mix_f32 v21, v13, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v14, v1, v2, vcc\nv_fma_mix_f32 v22, v14, v2, v3 op_sel_hi:[1,0,0]\nv_cndmask_b32 v15, v1, v2, vcc\nv_fma_mix_f32 v23, v15, v2, v3 op_sel_hi:[1,0,0]\n") ::: CLOB); }
        if (P == 7) { asm volatile(B64("s_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v8, v1, v2, vcc\nv_mul_f32 v16, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v9, v1, v2, vcc\nv_mul_f32 v17, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v10, v1, v2, vcc\nv_mul_f32 v18, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v11, v1, v2, vcc\nv_mul_f32 v19, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v12, v1, v2, vcc\nv_mul_f32 v20, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v13, v1, v2, vcc\nv_mul_f32 v21, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v14, v1, v2, vcc\nv_mul_f32 v22, v1, v2\ns_mov_b64 vcc, s[4:5]\nv_cndmask_b32 v15, v1, v2, vcc\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 8) { asm volatile(B64("v_min_f32 v8, v1, v2\nv_max_f32 v16, v1, v2\nv_min_f32 v9, v1, v2\nv_max_f32 v17, v1, v2\nv_min_f32 v10, v1, v2\nv_max_f32 v18, v1, v2\nv_min_f32 v11, v1, v2\nv_max_f32 v19, v1, v2\nv_min_f32 v12, v1, v2\nv_max_f32 v20, v1, v2\nv_min_f32 v13, v1, v2\nv_max_f32 v21, v1, v2\nv_min_f32 v14, v1, v2\nv_max_f32 v22, v1, v2\nv_min_f32 v15, v1, v2\nv_max_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 9) { asm volatile(B64("v_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v16, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v17, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v18, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v19, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v20, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v21, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v22, v1, v2\nv_cmp_lt_f32 vcc, v1, v2\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 10) { asm volatile(B64("v_lshlrev_b32 v8, 3, v1\nv_mul_f32 v16, v1, v2\nv_lshlrev_b32 v9, 3, v1\nv_mul_f32 v17, v1, v2\nv_lshlrev_b32 v10, 3, v1\nv_mul_f32 v18, v1, v2\nv_lshlrev_b32 v11, 3, v1\nv_mul_f32 v19, v1, v2\nv_lshlrev_b32 v12, 3, v1\nv_mul_f32 v20, v1, v2\nv_lshlrev_b32 v13, 3, v1\nv_mul_f32 v21, v1, v2\nv_lshlrev_b32 v14, 3, v1\nv_mul_f32 v22, v1, v2\nv_lshlrev_b32 v15, 3, v1\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 11) { asm volatile(B64("v_add_f64 v[16:17], v[2:3], v[4:5]\nv_mul_f32 v8, v1, v2\nv_add_f64 v[18:19], v[2:3], v[4:5]\nv_mul_f32 v9, v1, v2\nv_add_f64 v[20:21], v[2:3], v[4:5]\nv_mul_f32 v10, v1, v2\nv_add_f64 v[22:23], v[2:3], v[4:5]\nv_mul_f32 v11, v1, v2\nv_add_f64 v[16:17], v[2:3], v[4:5]\nv_mul_f32 v12, v1, v2\nv_add_f64 v[18:19], v[2:3], v[4:5]\nv_mul_f32 v13, v1, v2\nv_add_f64 v[20:21], v[2:3], v[4:5]\nv_mul_f32 v14, v1, v2\nv_add_f64 v[22:23], v[2:3], v[4:5]\nv_mul_f32 v15, v1, v2\n") ::: CLOB); }
        if (P == 12) { asm volatile(B64("v_cndmask_b32 v8, v1, v2, vcc\nv_add_f64 v[16:17], v[2:3], v[4:5]\nv_cndmask_b32 v9, v1, v2, vcc\nv_add_f64 v[18:19], v[2:3], v[4:5]\nv_cndmask_b32 v10, v1, v2, vcc\nv_add_f64 v[20:21], v[2:3], v[4:5]\nv_cndmask_b32 v11, v1, v2, vcc\nv_add_f64 v[22:23], v[2:3], v[4:5]\nv_cndmask_b32 v12, v1, v2, vcc\nv_add_f64 v[16:17], v[2:3], v[4:5]\nv_cndmask_b32 v13, v1, v2, vcc\nv_add_f64 v[18:19], v[2:3], v[4:5]\nv_cndmask_b32 v14, v1, v2, vcc\nv_add_f64 v[20:21], v[2:3], v[4:5]\nv_cndmask_b32 v15, v1, v2, vcc\nv_add_f64 v[22:23], v[2:3], v[4:5]\n") ::: CLOB); }
        if (P == 13) { asm volatile(B64("v_bitop3_b32 v8, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v9, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v10, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v11, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v12, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v13, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v14, v1, v2, v3 bitop3:0x48\nv_bitop3_b32 v15, v1, v2, v3 bitop3:0x48\n") ::: CLOB); }
        if (P == 14) { asm volatile(B64("v_bitop3_b32 v8, v1, v2, v3 bitop3:0x48\nv_mul_f32 v16, v1, v2\nv_bitop3_b32 v9, v1, v2, v3 bitop3:0x48\nv_mul_f32 v17, v1, v2\nv_bitop3_b32 v10, v1, v2, v3 bitop3:0x48\nv_mul_f32 v18, v1, v2\nv_bitop3_b32 v11, v1, v2, v3 bitop3:0x48\nv_mul_f32 v19, v1, v2\nv_bitop3_b32 v12, v1, v2, v3 bitop3:0x48\nv_mul_f32 v20, v1, v2\nv_bitop3_b32 v13, v1, v2, v3 bitop3:0x48\nv_mul_f32 v21, v1, v2\nv_bitop3_b32 v14, v1, v2, v3 bitop3:0x48\nv_mul_f32 v22, v1, v2\nv_bitop3_b32 v15, v1, v2, v3 bitop3:0x48\nv_mul_f32 v23, v1, v2\n") ::: CLOB); }
        if (P == 15) { asm volatile(B64("v_xor_b32 v8, v1, v2\nv_and_b32 v16, v1, v2\nv_xor_b32 v9, v1, v2\nv_and_b32 v17, v1, v2\nv_xor_b32 v10, v1, v2\nv_and_b32 v18, v1, v2\nv_xor_b32 v11, v1, v2\nv_and_b32 v19, v1, v2\nv_xor_b32 v12, v1, v2\nv_and_b32 v20, v1, v2\nv_xor_b32 v13, v1, v2\nv_and_b32 v21, v1, v2\nv_xor_b32 v14, v1, v2\nv_and_b32 v22, v1, v2\nv_xor_b32 v15, v1, v2\nv_and_b32 v23, v1, v2\n") ::: CLOB); }
        if (P == 16) { asm volatile(B64("v_min_f32 v8, v1, v2\nv_xor_b32 v16, v1, v2\nv_and_b32 v19, v1, v2\nv_min_f32 v9, v1, v2\nv_xor_b32 v17, v1, v2\nv_and_b32 v20, v1, v2\nv_min_f32 v10, v1, v2\nv_xor_b32 v18, v1, v2\nv_and_b32 v21, v1, v2\nv_min_f32 v11, v1, v2\nv_xor_b32 v19, v1, v2\nv_and_b32 v22, v1, v2\nv_min_f32 v12, v1, v2\nv_xor_b32 v20, v1, v2\nv_and_b32 v23, v1, v2\nv_min_f32 v13, v1, v2\nv_xor_b32 v21, v1, v2\nv_and_b32 v16, v1, v2\nv_min_f32 v14, v1, v2\nv_xor_b32 v22, v1, v2\nv_and_b32 v17, v1, v2\nv_min_f32 v15, v1, v2\nv_xor_b32 v23, v1, v2\nv_and_b32 v18, v1, v2\n") ::: CLOB); }
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
        CHECK(hipFuncSetAttribute((const void *)k_cls<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_cls<P>, dim3(blocks), dim3(256), lds, 0, d_out, iters); CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize()); }
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

template <int P> struct All { static void go(bool &first, int n_cu, Stamp *d_out, int iters) { All<P - 1>::go(first, n_cu, d_out, iters); run<P>(first, n_cu, d_out, iters); } };
template <> struct All<-1> { static void go(bool &, int, Stamp *, int) {} };
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, iters = 2048;
    Stamp *d_out = nullptr; CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)n_cu * 8 * 4));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"instructions_per_wave\": %d, \"results\": [", prop.gcnArchName, n_cu, iters * 64);
    bool first = true;
    All<N_PAT - 1>::go(first, n_cu, d_out, iters);
    printf("\n]}\n");
    return 0;
}
