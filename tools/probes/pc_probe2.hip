// pc_probe2.hip -- round 5: the consumer's fragment statement of dsblock_pc.inc in isolation (4 waves per workgroup, one per SIMD): clocks per
// MFMA for (a) 12 MFMAs whose accumulators repeat every 4 instructions (the kernel's order), (b) the same with the two weight reads + wait,
// (c) accumulators repeating every 12.    hipcc --offload-arch=gfx950 -O3 -o /tmp/pc_probe2 tools/probes/pc_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(C, A, B) "v_mfma_f32_16x16x32_bf16 %[" #C "], " A ", " B ", %[" #C "]\n\t"
#define MFMAS4(ALH, AHM) \
    MF(c0, ALH, "v[212:215]") MF(c1, ALH, "v[220:223]") MF(c2, ALH, "v[228:231]") MF(c3, ALH, "v[236:239]") \
    MF(c0, AHM, "v[210:213]") MF(c1, AHM, "v[218:221]") MF(c2, AHM, "v[226:229]") MF(c3, AHM, "v[234:237]") \
    MF(c0, AHM, "v[208:211]") MF(c1, AHM, "v[216:219]") MF(c2, AHM, "v[224:227]") MF(c3, AHM, "v[232:235]")
#define MFMAS12(ALH, AHM) \
    MF(c0, ALH, "v[212:215]") MF(c1, ALH, "v[220:223]") MF(c2, ALH, "v[228:231]") MF(c3, ALH, "v[236:239]") \
    MF(c4, AHM, "v[210:213]") MF(c5, AHM, "v[218:221]") MF(c6, AHM, "v[226:229]") MF(c7, AHM, "v[234:237]") \
    MF(c8, AHM, "v[208:211]") MF(c9, AHM, "v[216:219]") MF(c10, AHM, "v[224:227]") MF(c11, AHM, "v[232:235]")
#define CLOB "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251"
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(long long* out, int iters, float seed) {
    __shared__ float4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(seed, 0.f, 0.f, 0.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    const unsigned pa = lane * 16, pm = lane * 8;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0)
            asm volatile(MFMAS4("v[240:243]", "v[242:245]") : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]) :: CLOB);
        else if constexpr (MODE == 1)
            asm volatile("ds_read_b128 v[246:249], %[pa] offset:1536\n\tds_read_b64 v[250:251], %[pm] offset:1536\n\t" MFMAS4("v[240:243]", "v[242:245]") "s_waitcnt lgkmcnt(0)"
                         : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]) : [pa] "v"(pa), [pm] "v"(pm) : CLOB, "memory");
        else if constexpr (MODE == 2)
            asm volatile(MFMAS12("v[240:243]", "v[242:245]") : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]), [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]), [c10] "+v"(acc[10]), [c11] "+v"(acc[11]) :: CLOB);
        else
            asm volatile("ds_read_b128 v[246:249], %[pa] offset:1536\n\tds_read_b64 v[250:251], %[pm] offset:1536\n\t" MFMAS12("v[240:243]", "v[242:245]") "s_waitcnt lgkmcnt(0)"
                         : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]), [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]), [c10] "+v"(acc[10]), [c11] "+v"(acc[11]) : [pa] "v"(pa), [pm] "v"(pm) : CLOB, "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0];
    if (s == 12345.f) out[4000] = 1;
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    long long* d; (void)hipMalloc(&d, 65536);
    const int iters = 4000;
    probe<MODE><<<256, 256>>>(d, iters, 0.f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(1024);
    (void)hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v;
    printf("%-60s clk/mfma = %.2f\n", name, m / 1024 / iters / 12);
    (void)hipFree(d);
}
int main() {
    run<0>("12 MFMAs, accumulator reused every 4");
    run<1>("12 MFMAs (every 4) + ds_read_b128 + ds_read_b64 + wait");
    run<2>("12 MFMAs, 12 accumulators");
    run<3>("12 MFMAs (12 acc) + ds_read_b128 + ds_read_b64 + wait");
    return 0;
}
