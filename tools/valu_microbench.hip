// valu_microbench.hip - issue rate of the 32-bit integer VALU instructions the Myers DP kernels are made of, on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_microbench tools/valu_microbench.hip && ./valu_microbench
//
// For every instruction kind: a wave runs kIter x 64 instructions on 8 independent register chains ("ind": issue
// bound) or on one chain ("dep": latency bound).  Two numbers per case:
//   cyc/inst (1 wave/SIMD)   s_memtime ticks of one wave / its instructions - what ONE wave can issue
//   cyc/inst (8 waves/SIMD)  SIMD-cycles per wave-instruction with the chip full: elapsed ticks x 1024 SIMDs / all
//                            wave-instructions - the throughput the roofline of DESIGN.md 5.3 must be priced with
// (s_memtime counts at a fixed 100 MHz on this part, so ticks are converted with the measured shader clock: a
//  v_xor_b32 chain of known length, and cross-checked against hipEvent wall time x the clock rate HIP reports.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIter = 2048;       // loop trips; 64 instructions per trip

enum Kind { XOR = 0, BITOP3, ADDU32, ADDCO_ADDC, ALIGNBIT, LSHL_OR, BFE_I32, DPP_MOV, CNDMASK, LSHL_B64, ADD_U64, AND_OR, KINDS };
static const char* kNames[KINDS] = {"v_xor_b32", "v_bitop3_b32", "v_add_u32", "v_add_co_u32+v_addc_co_u32 (pair)", "v_alignbit_b32", "v_lshl_or_b32",
                                    "v_bfe_i32", "v_mov_b32 dpp row_shr:1", "v_cndmask_b32", "v_lshlrev_b64", "v_lshl_add_u64", "v_and_or_b32"};

// one instruction on register r (and the loop-invariant s / t)
#define I_XOR(r)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(s));
#define I_BITOP3(r)   asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(s), "v"(t));
#define I_ADD(r)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(s));
#define I_ADDC(r)     asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(r) : "v"(s), "v"(t) : "vcc");
#define I_ALIGN(r)    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r) : "v"(s));
#define I_LSHLOR(r)   asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(s));
#define I_BFE(r)      asm volatile("v_bfe_i32 %0, %0, %1, 1" : "+v"(r) : "v"(s));
#define I_DPP(r)      asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
#define I_CND(r)      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(s) : "vcc");
#define I_SHL64(r)    asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r));
#define I_ADD64(r)    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(s64));
#define I_ANDOR(r)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(s), "v"(t));

#define IND8(I, T)  { I(T##0) I(T##1) I(T##2) I(T##3) I(T##4) I(T##5) I(T##6) I(T##7) }
#define DEP8(I, T)  { I(T##0) I(T##0) I(T##0) I(T##0) I(T##0) I(T##0) I(T##0) I(T##0) }
#define BODY64(M, I, T) { M(I, T) M(I, T) M(I, T) M(I, T) M(I, T) M(I, T) M(I, T) M(I, T) }

template <int KIND, bool DEP>
__global__ void __launch_bounds__(64) k_bench(unsigned* out, unsigned long long* ticks, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    const unsigned s = seed * 2654435761u + threadIdx.x, t = seed ^ 0x9e3779b9u;
    const unsigned long long s64 = ((unsigned long long)s << 32) | t;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(s), "v"(t) : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();     // s_memtime
    for (int it = 0; it < kIter; ++it) {
        if (KIND == XOR)        { if (DEP) BODY64(DEP8, I_XOR, a) else BODY64(IND8, I_XOR, a) }
        if (KIND == BITOP3)     { if (DEP) BODY64(DEP8, I_BITOP3, a) else BODY64(IND8, I_BITOP3, a) }
        if (KIND == ADDU32)     { if (DEP) BODY64(DEP8, I_ADD, a) else BODY64(IND8, I_ADD, a) }
        if (KIND == ADDCO_ADDC) { if (DEP) BODY64(DEP8, I_ADDC, a) else BODY64(IND8, I_ADDC, a) }
        if (KIND == ALIGNBIT)   { if (DEP) BODY64(DEP8, I_ALIGN, a) else BODY64(IND8, I_ALIGN, a) }
        if (KIND == LSHL_OR)    { if (DEP) BODY64(DEP8, I_LSHLOR, a) else BODY64(IND8, I_LSHLOR, a) }
        if (KIND == BFE_I32)    { if (DEP) BODY64(DEP8, I_BFE, a) else BODY64(IND8, I_BFE, a) }
        if (KIND == DPP_MOV)    { if (DEP) BODY64(DEP8, I_DPP, a) else BODY64(IND8, I_DPP, a) }
        if (KIND == CNDMASK)    { if (DEP) BODY64(DEP8, I_CND, a) else BODY64(IND8, I_CND, a) }
        if (KIND == LSHL_B64)   { if (DEP) BODY64(DEP8, I_SHL64, b) else BODY64(IND8, I_SHL64, b) }
        if (KIND == ADD_U64)    { if (DEP) BODY64(DEP8, I_ADD64, b) else BODY64(IND8, I_ADD64, b) }
        if (KIND == AND_OR)     { if (DEP) BODY64(DEP8, I_ANDOR, a) else BODY64(IND8, I_ANDOR, a) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
    if (r == 0x12345678u) out[threadIdx.x] = r;               // keeps the chains alive
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

struct Res { double tick_per_inst_1, simd_ns_per_inst_8; };

template <int KIND, bool DEP>
static Res run(unsigned* d_out, unsigned long long* d_ticks, int n_simd)
{
    Res r;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int per_inst = KIND == ADDCO_ADDC ? 2 : 1;
    const double insts = (double)kIter * 64 * per_inst;
    for (int pass = 0; pass < 2; ++pass) {
        const int waves = pass == 0 ? n_simd : n_simd * 8;
        k_bench<KIND, DEP><<<waves, 64>>>(d_out, d_ticks, 1u);          // warm-up (clocks, code)
        CHECK(hipEventRecord(e0));
        k_bench<KIND, DEP><<<waves, 64>>>(d_out, d_ticks, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(waves);
        CHECK(hipMemcpy(h.data(), d_ticks, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        if (pass == 0) r.tick_per_inst_1 = sum / waves / insts;
        else r.simd_ns_per_inst_8 = (double)ms * 1e6 * n_simd / (insts * waves);
    }
    return r;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_simd = p.multiProcessorCount * 4;
    printf("device: %s (%s), %d CUs, clockRate %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    unsigned* d_out; unsigned long long* d_ticks;
    CHECK(hipMalloc(&d_out, 4096)); CHECK(hipMalloc(&d_ticks, (size_t)n_simd * 8 * 8));
    Res res[KINDS][2];
#define RUN(K) res[K][0] = run<K, false>(d_out, d_ticks, n_simd); res[K][1] = run<K, true>(d_out, d_ticks, n_simd);
    RUN(XOR) RUN(BITOP3) RUN(ADDU32) RUN(ADDCO_ADDC) RUN(ALIGNBIT) RUN(LSHL_OR) RUN(BFE_I32) RUN(DPP_MOV) RUN(CNDMASK) RUN(LSHL_B64) RUN(ADD_U64) RUN(AND_OR)
    // the tick of s_memtime in shader cycles: assume nothing, report both raw ticks and ns
    printf("\n| instruction (wave64) | ticks/inst, 1 wave/SIMD, 8 chains | ticks/inst, 1 wave/SIMD, 1 chain | SIMD-ns/inst, 8 waves/SIMD, 8 chains | SIMD-ns/inst, 8 waves/SIMD, 1 chain |\n|---|---|---|---|---|\n");
    for (int k = 0; k < KINDS; ++k)
        printf("| `%s` | %.3f | %.3f | %.4f | %.4f |\n", kNames[k], res[k][0].tick_per_inst_1, res[k][1].tick_per_inst_1, res[k][0].simd_ns_per_inst_8, res[k][1].simd_ns_per_inst_8);
    const double ns_xor = res[XOR][0].simd_ns_per_inst_8;
    printf("\nIf v_xor_b32 issues in 2 cycles per wave64 (SIMD-32), the shader clock during these runs was %.2f GHz; if in 4 cycles, %.2f GHz.\n",
           2.0 / ns_xor, 4.0 / ns_xor);
    printf("relative cost (v_xor_b32 = 1): ");
    for (int k = 0; k < KINDS; ++k) printf("%s %.2f; ", kNames[k], res[k][0].simd_ns_per_inst_8 / ns_xor);
    printf("\n");
    return 0;
}
