// gather_bench.hip -- what does one per-lane record gather cost on gfx950?
// Measures the CU-level throughput of the vector-memory path for the access
// shapes the SPH pair kernel can use to fetch neighbour records.  The table is
// small enough to stay in every XCD's L2 (the pair kernel's L2 hit rate is
// 97 %), so the numbers are L1/TA/L2->L1 costs, not HBM.
//
// LAYOUT 0: AoS, `strideP` 16-B pieces per record, NP pieces read
//        1: SoA of pieces: piece q of record j at rec[q*ntab + j]
//        2/3: AoSoA: blocks of 8 / 16 records, piece q of record j at rec[(j/B)*B*NP + q*B + j%B]
// MODE   0: every lane loads its own record's pieces (global_load_dwordx4)
//        1: quad cooperative: 4 lanes load the 4 pieces of ONE record (4 instr serve 4 records), rest per lane
//        2: as 0 through raw buffer loads with cache-policy bits AUX (1 sc0, 2 nt, 16 sc1)
//        4: records in LDS, random per-lane ds_read_b128
//        9: quad cooperative through LDS: 4 x global_load_lds_dwordx4 (the 4 lanes of a quad fetch the 4 first
//           pieces of ONE record into LDS, lane-linear), 4 x ds_read_b128 transposed, remaining pieces per lane
//        6/7/8: as 0, but the 4 lanes of a quad / 8 / 16 adjacent lanes gather the SAME record
// PATTERN 0: record index uniformly random in the wave's window of W records
//         1: sliding front: lane + 4*step + rnd(36), a new "row" every 10 steps
//            (neighbouring lanes touch neighbouring records, like the pair kernel)
// Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

struct Args {
    const f4 *rec;
    int strideP;
    uint32_t ntab, W;
    int iters, pattern;
    float *out;
    int verify;
};

template <int MODE, int NP, int LAYOUT, int AUX>
__global__ __launch_bounds__(256, 4) void k_gather(Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t s = (blockIdx.x * 256 + t) * 2654435761u + 12345u;
    const uint32_t wbase = (uint32_t)(((uint64_t)(blockIdx.x * 4 + wv) * 977u) % (a.ntab - a.W));
    f4 acc = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.rec, 0, 0x7fffffff, 0x00020000);
    auto idx = [&](uint32_t j, int q) -> size_t {
        if (LAYOUT == 0) return (size_t)j * a.strideP + q;
        if (LAYOUT == 1) return (size_t)q * a.ntab + j;
        constexpr uint32_t B = LAYOUT == 2 ? 8u : 16u;
        return (size_t)(j / B) * (B * NP) + q * B + (j % B);
    };
    for (int it = 0; it < a.iters; it++) {
        uint32_t j;
        if (a.pattern == 0) j = wbase + (uint32_t)(((uint64_t)lcg(s) * a.W) >> 24);
        else {
            const uint32_t row = (uint32_t)it / 10u;
            j = (wbase + row * 1531u + lane + 4u * ((uint32_t)it % 10u) + (uint32_t)(((uint64_t)lcg(s) * 36u) >> 24)) % a.ntab;
        }
        if (MODE == 6) j = __shfl(j, lane & ~3, 64);
        if (MODE == 7) j = __shfl(j, lane & ~7, 64);
        if (MODE == 8) j = __shfl(j, lane & ~15, 64);
        if (MODE == 0 || (MODE >= 6 && MODE <= 8)) {
#pragma unroll
            for (int q = 0; q < NP; q++) acc += a.rec[idx(j, q)];
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx(j, q) * 16), 0, AUX);
                acc += __builtin_bit_cast(f4, v);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t jk = __shfl(j, (lane & ~3) | k, 64);
                acc += a.rec[(size_t)jk * a.strideP + (lane & 3)];
            }
#pragma unroll
            for (int q = 4; q < NP; q++) acc += a.rec[(size_t)j * a.strideP + q];
        } else if (MODE == 9) {
            // region i (1 KiB + 16 B skew) <- instruction i: lane l holds piece (l & 3) of the record of lane (l & ~3) | i
            char *wbase_lds = smem + wv * (4 * 1040);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t jk = __shfl(j, (lane & ~3) | k, 64);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.rec + (size_t)jk * a.strideP + (lane & 3)),
                                                 (__attribute__((address_space(3))) void *)(wbase_lds + k * 1040), 16, 0, 0);
            }
#pragma unroll
            for (int q = 4; q < NP; q++) acc += a.rec[(size_t)j * a.strideP + q];
            __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
            const f4 *mine = reinterpret_cast<const f4 *>(wbase_lds + (lane & 3) * 1040) + (lane & ~3);
#pragma unroll
            for (int q = 0; q < 4; q++) acc += mine[q];
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the reads are done before the next DMA overwrites
        } else if (MODE == 10) {
            // as 9, staged through registers: 4 cooperative global loads, ds_write_b128 lane-linear, ds_read_b128 transposed
            char *wbase_lds = smem + wv * (4 * 1040);
            f4 r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t jk = __shfl(j, (lane & ~3) | k, 64);
                r[k] = a.rec[(size_t)jk * a.strideP + (lane & 3)];
            }
#pragma unroll
            for (int q = 4; q < NP; q++) acc += a.rec[(size_t)j * a.strideP + q];
#pragma unroll
            for (int k = 0; k < 4; k++) reinterpret_cast<f4 *>(wbase_lds + k * 1040)[lane] = r[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f4 *mine = reinterpret_cast<const f4 *>(wbase_lds + (lane & 3) * 1040) + (lane & ~3);
#pragma unroll
            for (int q = 0; q < 4; q++) acc += mine[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 11) {
            // as 10 with the record SPLIT: pieces 0..3 in a 64-B-aligned array (stride 64 B), the rest in a second array
            char *wbase_lds = smem + wv * (4 * 1040);
            const f4 *rec2 = a.rec + (size_t)a.ntab * 4;
            f4 r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t jk = __shfl(j, (lane & ~3) | k, 64);
                r[k] = a.rec[(size_t)jk * 4 + (lane & 3)];
            }
#pragma unroll
            for (int q = 4; q < NP; q++) acc += rec2[(size_t)j * (NP - 4) + (q - 4)];
#pragma unroll
            for (int k = 0; k < 4; k++) reinterpret_cast<f4 *>(wbase_lds + k * 1040)[lane] = r[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f4 *mine = reinterpret_cast<const f4 *>(wbase_lds + (lane & 3) * 1040) + (lane & ~3);
#pragma unroll
            for (int q = 0; q < 4; q++) acc += mine[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 12) {
            // per-lane gathers from the same split layout (no cooperation)
            const f4 *rec2 = a.rec + (size_t)a.ntab * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += a.rec[(size_t)j * 4 + q];
#pragma unroll
            for (int q = 4; q < NP; q++) acc += rec2[(size_t)j * (NP - 4) + (q - 4)];
        } else if (MODE == 4) {
            const uint32_t jl = (uint32_t)(((uint64_t)lcg(s) * a.W) >> 24);
            const f4 *p = reinterpret_cast<const f4 *>(smem) + (size_t)jl * a.strideP;
#pragma unroll
            for (int q = 0; q < NP; q++) acc += p[q];
        }
    }
    if (a.verify || acc.x + acc.y + acc.z + acc.w == 123.456f) a.out[blockIdx.x * 256 + t] = acc.x + acc.y + acc.z + acc.w;
}

static f4 *d_rec; static float *d_out;

template <int MODE, int NP, int LAYOUT, int AUX>
static void run(const char *name, int strideP, uint32_t ntab, uint32_t W, int pattern, size_t lds = 0)
{
    const int blocks = 256 * 4 * 4, iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (lds) CK(hipFuncSetAttribute((const void *)k_gather<MODE, NP, LAYOUT, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Args a{d_rec, strideP, ntab, W, 50, pattern, d_out, 0};
    hipLaunchKernelGGL((k_gather<MODE, NP, LAYOUT, AUX>), dim3(blocks), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    a.iters = iters;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<MODE, NP, LAYOUT, AUX>), dim3(blocks), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_iters = (double)blocks * 4 * iters;
    const double cu_ns = ms * 1e6 * 256.0 / wave_iters;
    printf("%-14s NP=%d stride=%3dB tab=%5.0fKB W=%6u pat=%d: %7.3f ms  %6.1f cyc/wave-iter  %5.2f cyc/lane-hit  (%5.1f cyc per piece-instr)\n",
           name, NP, strideP * 16, (double)ntab * strideP * 16 / 1024.0, W, pattern, ms, cu_ns * 2.4, cu_ns * 2.4 / 64, cu_ns * 2.4 / NP);
}

template <int MODE, int NP>
static std::vector<float> sums(int strideP, uint32_t ntab, size_t lds)
{
    if (lds) CK(hipFuncSetAttribute((const void *)k_gather<MODE, NP, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Args a{d_rec, strideP, ntab, 1024, 7, 1, d_out, 1};
    hipLaunchKernelGGL((k_gather<MODE, NP, 0, 0>), dim3(64), dim3(256), lds, 0, a);
    std::vector<float> h(64 * 256);
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    return h;
}

int main()
{
    const size_t bytes = 512u << 20;
    std::vector<float> h(bytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.01f;
    CK(hipMalloc(&d_rec, bytes));
    CK(hipMemcpy(d_rec, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 256 * 16 * 256 * 4));
    for (int pat = 0; pat < 2; pat++) {
        const uint32_t nt = 24576; // records in the table (80 B -> 1.9 MB)
        for (uint32_t W : {64u, 1024u}) {
            if (pat == 1 && W != 1024u) continue;
            printf("---- pattern %d window %u\n", pat, W);
            run<0, 5, 0, 0>("aos80", 5, nt, W, pat);
            run<6, 5, 0, 0>("aos80 quad-same", 5, nt, W, pat);
            run<7, 5, 0, 0>("aos80 oct-same", 5, nt, W, pat);
            run<8, 5, 0, 0>("aos80 16-same", 5, nt, W, pat);
            run<6, 3, 0, 0>("aos48 quad-same", 3, nt, W, pat);
            run<0, 5, 0, 0>("aos128", 8, nt, W, pat);
            run<0, 4, 0, 0>("aos64", 4, nt, W, pat);
            run<0, 3, 0, 0>("aos48", 3, nt, W, pat);
            run<0, 1, 0, 0>("aos16", 1, nt, W, pat);
            run<0, 5, 1, 0>("soa5", 5, nt, W, pat);
            run<0, 3, 1, 0>("soa3", 3, nt, W, pat);
            run<1, 5, 0, 0>("quad64+1(128)", 8, nt, W, pat);
            run<1, 4, 0, 0>("quad64", 4, nt, W, pat);
            run<2, 5, 0, 0>("buf aos80", 5, nt, W, pat);
            run<2, 5, 0, 1>("buf aos80 sc0", 5, nt, W, pat);
            run<2, 5, 0, 2>("buf aos80 nt", 5, nt, W, pat);
            run<2, 5, 0, 16>("buf aos80 sc1", 5, nt, W, pat);
            run<2, 5, 0, 17>("buf aos80 sc0sc1", 5, nt, W, pat);
            run<2, 5, 0, 16>("buf aos128 sc1", 8, nt, W, pat);
            run<2, 5, 1, 16>("buf soa5 sc1", 5, nt, W, pat);
        }
    }
    {
        auto r0 = sums<0, 5>(5, 24576, 0), r9 = sums<9, 5>(5, 24576, 4 * 4 * 1040);
        size_t bad = 0;
        for (size_t i = 0; i < r0.size(); i++) bad += fabsf(r0[i] - r9[i]) > 1e-3f * fabsf(r0[i]);
        printf("quadlds vs per-lane gather: %zu of %zu lane sums differ\n", bad, r0.size());
        auto r10 = sums<10, 5>(5, 24576, 4 * 4 * 1040);
        bad = 0;
        for (size_t i = 0; i < r0.size(); i++) bad += fabsf(r0[i] - r10[i]) > 1e-3f * fabsf(r0[i]);
        printf("quadreg vs per-lane gather: %zu of %zu lane sums differ\n", bad, r0.size());
    }
    printf("---- quad cooperative through LDS-DMA\n");
    for (uint32_t nt : {24576u, 4000000u}) {
        run<0, 5, 0, 0>("aos80", 5, nt, 1024, 1);
        run<9, 5, 0, 0>("quadlds 80", 5, nt, 1024, 1, 4 * 4 * 1040);
        run<10, 5, 0, 0>("quadreg 80", 5, nt, 1024, 1, 4 * 4 * 1040);
        run<10, 7, 0, 0>("quadreg 112", 7, nt, 1024, 1, 4 * 4 * 1040);
        run<1, 4, 0, 0>("quad64 (no transpose)", 4, nt, 1024, 1);
        run<11, 4, 0, 0>("quadreg 64", 4, nt, 1024, 1, 4 * 4 * 1040);
        run<0, 4, 0, 0>("aos64", 4, nt, 1024, 1);
        run<0, 1, 0, 0>("aos16", 1, nt, 1024, 1);
        run<11, 5, 0, 0>("quadreg 64+16", 5, nt, 1024, 1, 4 * 4 * 1040);
        run<12, 5, 0, 0>("split 64+16", 5, nt, 1024, 1);
        run<11, 7, 0, 0>("quadreg 64+48", 7, nt, 1024, 1, 4 * 4 * 1040);
        run<12, 7, 0, 0>("split 64+48", 7, nt, 1024, 1);
        run<9, 4, 0, 0>("quadlds 64", 4, nt, 1024, 1, 4 * 4 * 1040);
        run<9, 7, 0, 0>("quadlds 112", 7, nt, 1024, 1, 4 * 4 * 1040);
        run<0, 7, 0, 0>("aos112", 7, nt, 1024, 1);
    }
    printf("---- table-size sweep, pattern 1\n");
    for (uint32_t nt : {24576u, 400000u, 4000000u}) {
        run<0, 5, 0, 0>("aos80", 5, nt, 1024, 1);
        run<0, 5, 1, 0>("soa5", 5, nt, 1024, 1);
        run<0, 5, 2, 0>("aosoa8", 5, nt, 1024, 1);
        run<0, 5, 3, 0>("aosoa16", 5, nt, 1024, 1);
        run<0, 3, 0, 0>("aos48", 3, nt, 1024, 1);
        run<0, 3, 2, 0>("aosoa8 x3", 3, nt, 1024, 1);
    }
    printf("---- LDS-resident records, random ds_read_b128\n");
    run<4, 5, 0, 0>("lds128", 5, 24576, 400, 0, 400 * 80);
    run<4, 4, 0, 0>("lds128", 4, 24576, 500, 0, 500 * 64);
    return 0;
}
