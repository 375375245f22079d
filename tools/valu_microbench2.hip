// valu_microbench2.hip - what makes a 32-bit VALU instruction cost 2 or 4 SIMD cycles on gfx950: encoding (VOP2 vs
// VOP3 / DPP), the number of VGPR sources, the VGPR banks of the sources (register index mod 4), SGPR-pair carries.
// Every case is one hand-written block of 64 instructions on FIXED registers (v40..v55), run kIter times by every wave,
// 8 waves per SIMD on the whole chip; reported: SIMD-ns per instruction and cycles at the clock the plain VOP2 case
// implies (v_add_u32 e32, 2 cycles).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_microbench2 tools/valu_microbench2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 4096;

#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","vcc","s20","s21","s22","s23"
// 8 instructions with destination chains v40..v47 (independent), sources given per case
#define R8(fmt) fmt(40) fmt(41) fmt(42) fmt(43) fmt(44) fmt(45) fmt(46) fmt(47)
#define R64(fmt) R8(fmt) R8(fmt) R8(fmt) R8(fmt) R8(fmt) R8(fmt) R8(fmt) R8(fmt)
#define S(x) #x

struct Case { const char* name; int insts_per_rep; };

template <int C> __device__ __forceinline__ void body();
#define CASE(id, text) template <> __device__ __forceinline__ void body<id>() { asm volatile(text ::: CLOB); }

// ---- 2-source ops
#define F0(d) "v_add_u32_e32 v" S(d) ", v" S(d) ", v48\n"
CASE(0, R64(F0))
#define F1(d) "v_xor_b32_e32 v" S(d) ", v" S(d) ", v48\n"
CASE(1, R64(F1))
#define F2(d) "v_xor_b32_e64 v" S(d) ", v" S(d) ", v48\n"
CASE(2, R64(F2))
#define F3(d) "v_lshlrev_b32_e32 v" S(d) ", 1, v" S(d) "\n"
CASE(3, R64(F3))
#define F4(d) "v_not_b32_e32 v" S(d) ", v" S(d) "\n"
CASE(4, R64(F4))
// ---- 3-source VOP3, sources in different banks (d, 49, 50 -> banks d%4, 1, 2) / all the same register class
#define F5(d) "v_bitop3_b32 v" S(d) ", v" S(d) ", v49, v50 bitop3:0x96\n"
CASE(5, R64(F5))
#define F6(d) "v_bitop3_b32 v" S(d) ", v" S(d) ", v48, v52 bitop3:0x96\n"
CASE(6, R64(F6))
#define F7(d) "v_and_or_b32 v" S(d) ", v" S(d) ", v49, v50\n"
CASE(7, R64(F7))
#define F8(d) "v_or3_b32 v" S(d) ", v" S(d) ", v49, v50\n"
CASE(8, R64(F8))
#define F9(d) "v_bfi_b32 v" S(d) ", v" S(d) ", v49, v50\n"
CASE(9, R64(F9))
// ---- 2 VGPR sources + inline constant (VOP3)
#define F10(d) "v_alignbit_b32 v" S(d) ", v" S(d) ", v49, 31\n"
CASE(10, R64(F10))
#define F11(d) "v_lshl_or_b32 v" S(d) ", v" S(d) ", 1, v49\n"
CASE(11, R64(F11))
#define F12(d) "v_bfe_i32 v" S(d) ", v" S(d) ", v49, 1\n"
CASE(12, R64(F12))
#define F13(d) "v_bfe_u32 v" S(d) ", v" S(d) ", 3, 5\n"
CASE(13, R64(F13))
// ---- DPP
#define F14(d) "v_mov_b32_dpp v" S(d) ", v" S(d) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
CASE(14, R64(F14))
#define F15(d) "v_add_u32_dpp v" S(d) ", v" S(d) ", v49 row_shr:1 row_mask:0xf bank_mask:0xf\n"
CASE(15, R64(F15))
// ---- carries: vcc (VOP2) vs an SGPR pair (VOP3 + the s_nop the compiler has to put between them)
#define F16(d) "v_add_co_u32_e32 v" S(d) ", vcc, v" S(d) ", v48\n v_addc_co_u32_e32 v" S(d) ", vcc, v" S(d) ", v49, vcc\n"
CASE(16, R64(F16))
#define F17(d) "v_add_co_u32_e64 v" S(d) ", s[20:21], v" S(d) ", v48\n s_nop 1\n v_addc_co_u32_e64 v" S(d) ", s[20:21], v" S(d) ", v49, s[20:21]\n"
CASE(17, R64(F17))
// ---- compares and selects
#define F18(d) "v_cmp_lt_u32_e32 vcc, v" S(d) ", v48\n"
CASE(18, R64(F18))
#define F19(d) "v_cmp_lt_u32_e64 s[20:21], v" S(d) ", v48\n"
CASE(19, R64(F19))
#define F20(d) "v_cndmask_b32_e32 v" S(d) ", v" S(d) ", v48, vcc\n"
CASE(20, R64(F20))
#define F21(d) "v_cndmask_b32_e64 v" S(d) ", v" S(d) ", v48, s[22:23]\n"
CASE(21, R64(F21))
// ---- 64-bit
#define F22(d) "v_lshlrev_b64 v[50:51], 1, v[50:51]\n"
CASE(22, R64(F22))
#define F23(d) "v_lshl_add_u64 v[50:51], v[50:51], 0, v[52:53]\n"
CASE(23, R64(F23))
// ---- scalar ALU next to nothing (how many SALU ops a wave can interleave)
#define F24(d) "v_xor_b32_e32 v" S(d) ", v" S(d) ", v48\n"      // (was: s_add_u32 alone - 27 us per instruction on this box, minutes per run)
CASE(24, R64(F24))
// ---- a VOP2 and a SALU op alternating (do they share an issue slot within ONE wave?)
#define F25(d) "v_xor_b32_e32 v" S(d) ", v" S(d) ", v48\n"      // (was: VOP2 + s_add_u32 alternating, see F24)
CASE(25, R64(F25))

// ---- mixes: what the DP kernel's step looks like (25 full-rate + 14 half-rate instructions), and dependent chains
#define F26(d) "v_xor_b32_e32 v" S(d) ", v" S(d) ", v48\n v_alignbit_b32 v" S(d) ", v" S(d) ", v49, 31\n"
CASE(26, R64(F26))
#define F27(d) "v_xor_b32_e32 v" S(d) ", v" S(d) ", v48\n v_bitop3_b32 v" S(d) ", v" S(d) ", v49, v50 bitop3:0x96\n v_alignbit_b32 v" S(d) ", v" S(d) ", v49, 31\n"
CASE(27, R64(F27))
// one dependent chain per wave (every instruction reads the previous one's result)
#define F28(d) "v_xor_b32_e32 v40, v40, v48\n"
CASE(28, R64(F28))
#define F29(d) "v_alignbit_b32 v40, v40, v49, 31\n"
CASE(29, R64(F29))
#define F30(d) "v_xor_b32_e32 v40, v40, v48\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_alignbit_b32 v40, v40, v49, 31\n"
CASE(30, R64(F30))
// two chains per wave
#define F31(d) "v_xor_b32_e32 v40, v40, v48\n v_xor_b32_e32 v41, v41, v48\n"
CASE(31, R64(F31))
// full-rate ops whose sources collide in a VGPR bank (v48, v52: both bank 0): the VOP2 case
#define F32(d) "v_xor_b32_e32 v" S(d) ", v48, v52\n"
CASE(32, R64(F32))

static const Case kCases[] = {
    {"v_add_u32_e32 (VOP2)", 64}, {"v_xor_b32_e32 (VOP2)", 64}, {"v_xor_b32_e64 (same op, VOP3 encoding)", 64}, {"v_lshlrev_b32_e32 v, 1, v", 64}, {"v_not_b32_e32 (VOP1)", 64},
    {"v_bitop3_b32 d, d, v49, v50 (sources in 3 banks for 3 of 4 d)", 64}, {"v_bitop3_b32 d, d, v48, v52 (v48, v52 same bank)", 64}, {"v_and_or_b32 d, d, v49, v50", 64},
    {"v_or3_b32 d, d, v49, v50", 64}, {"v_bfi_b32 d, d, v49, v50", 64}, {"v_alignbit_b32 d, d, v49, 31", 64}, {"v_lshl_or_b32 d, d, 1, v49", 64}, {"v_bfe_i32 d, d, v49, 1", 64},
    {"v_bfe_u32 d, d, 3, 5", 64}, {"v_mov_b32_dpp row_shr:1", 64}, {"v_add_u32_dpp row_shr:1", 64}, {"v_add_co_u32_e32 + v_addc_co_u32_e32 (vcc)", 128},
    {"v_add_co_u32_e64 + s_nop 1 + v_addc_co_u32_e64 (SGPR pair; per VALU inst)", 128}, {"v_cmp_lt_u32_e32 vcc", 64}, {"v_cmp_lt_u32_e64 s[20:21]", 64},
    {"v_cndmask_b32_e32 (vcc)", 64}, {"v_cndmask_b32_e64 (SGPR pair)", 64}, {"v_lshlrev_b64", 64}, {"v_lshl_add_u64", 64}, {"(placeholder) v_xor_b32_e32", 64},
    {"(placeholder) v_xor_b32_e32", 64},
    {"mix 1:1 v_xor_e32 / v_alignbit (8 chains; per instruction)", 128}, {"mix 2:1 v_xor_e32, v_bitop3, v_alignbit (8 chains; per instruction)", 192},
    {"v_xor_b32_e32 ONE dependent chain per wave", 64}, {"v_alignbit_b32 ONE dependent chain per wave", 64},
    {"mix 2:1 ONE dependent chain per wave (per instruction)", 192}, {"v_xor_b32_e32 TWO dependent chains per wave (per instruction)", 128},
    {"v_xor_b32_e32 d, v48, v52 (both sources in one VGPR bank)", 64}};
constexpr int kNumCases = sizeof(kCases) / sizeof(kCases[0]);

template <int C> __global__ void __launch_bounds__(64) k(unsigned* out, unsigned long long* ticks)
{
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, 3\n v_mov_b32 v49, 5\n v_mov_b32 v50, 7\n v_mov_b32 v51, 9\n v_mov_b32 v52, 11\n v_mov_b32 v53, 13\n s_mov_b64 s[22:23], 0x5555\n s_mov_b64 vcc, 0x3333\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n"
                 :: "v"(threadIdx.x) : CLOB);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIter; ++it) body<C>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned r; asm volatile("v_xor_b32 %0, v40, v47" : "=v"(r) :: CLOB);
    if (r == 0x12345678u) out[threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int C> static void run(unsigned* d_out, unsigned long long* d_ticks, int n_simd, double* ns8, double* tick1)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double insts = (double)kIter * kCases[C].insts_per_rep;
    for (int pass = 0; pass < 2; ++pass) {
        const int waves = pass == 0 ? n_simd : n_simd * 8;
        k<C><<<waves, 64>>>(d_out, d_ticks);
        CHECK(hipEventRecord(e0));
        k<C><<<waves, 64>>>(d_out, d_ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(waves);
        CHECK(hipMemcpy(h.data(), d_ticks, waves * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        if (pass == 0) tick1[C] = sum / waves / insts; else ns8[C] = (double)ms * 1e6 * n_simd / (insts * waves);
    }
}

template <int C> struct Runner { static void go(unsigned* o, unsigned long long* t, int n, double* a, double* b) { run<C>(o, t, n, a, b); Runner<C + 1>::go(o, t, n, a, b); } };
template <> struct Runner<kNumCases> { static void go(unsigned*, unsigned long long*, int, double*, double*) {} };

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_simd = p.multiProcessorCount * 4;
    unsigned* d_out; unsigned long long* d_ticks;
    CHECK(hipMalloc(&d_out, 4096)); CHECK(hipMalloc(&d_ticks, (size_t)n_simd * 64));
    double ns8[kNumCases], tick1[kNumCases];
    Runner<0>::go(d_out, d_ticks, n_simd, ns8, tick1);
    const double ghz = 2.0 / ns8[0];
    printf("device: %s, %d CUs; shader clock implied by v_add_u32_e32 = 2 cycles: %.2f GHz\n\n", p.gcnArchName, p.multiProcessorCount, ghz);
    printf("| instruction (wave64, per instruction) | SIMD-ns, 8 waves/SIMD | SIMD cycles | s_memtime ticks, 1 wave/SIMD |\n|---|---|---|---|\n");
    for (int c = 0; c < kNumCases; ++c) printf("| `%s` | %.3f | %.2f | %.2f |\n", kCases[c].name, ns8[c], ns8[c] * ghz, tick1[c]);
    return 0;
}
