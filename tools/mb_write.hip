// Development microbenchmark: variants of the label-write kernel (who eats the time: stores, bit loads, label loads, index math?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define WR_K 8
// MODE bit0: load bits, bit1: load labels, bit2: nontemporal stores
template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void k_write(const u64* __restrict__ bits, int w, int ww, int h, const u32* __restrict__ seglabel, int32_t* __restrict__ labels, u32 total_rows, u32 gpr, u32 gpr_magic)
{
    const u32 row0 = blockIdx.x * ROWS;
    const u32 nrows = min((u32)ROWS, total_rows - row0);
    const u32 ngroups = nrows * gpr;
    const u64* brow0 = bits + (size_t)row0 * ww;
    int32_t* lrow0 = labels + (size_t)row0 * w;
    for (u32 qb = 0; qb < ngroups; qb += 256 * WR_K) {
        u64 wv[WR_K]; u32 rl[WR_K], g[WR_K]; bool live[WR_K];
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            const u32 q = qb + (u32)k * 256 + threadIdx.x;
            live[k] = q < ngroups; const u32 qq = live[k] ? q : 0;
            rl[k] = __umulhi(qq, gpr_magic); g[k] = qq - rl[k] * gpr;
            wv[k] = (MODE & 1) ? brow0[rl[k] * (u32)ww + (g[k] >> 4)] : 0ull;
        }
        u32 la[WR_K];
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            const u32 nib = (u32)(wv[k] >> ((g[k] * 4) & 63)) & 0xfu;
            la[k] = (MODE & 2) ? seglabel[nib ? (rl[k] * 64 + g[k]) & 1023 : 0] : nib;
        }
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            if (!live[k]) continue;
            const u32 nib = (u32)(wv[k] >> ((g[k] * 4) & 63)) & 0xfu;
            int4 v = make_int4((nib & 1) ? la[k] : 0, (nib & 2) ? la[k] : 0, (nib & 4) ? la[k] : 0, (nib & 8) ? la[k] : 0);
            int4* d = reinterpret_cast<int4*>(lrow0 + (size_t)rl[k] * w + g[k] * 4);
            if (MODE & 4) { typedef int v4i __attribute__((ext_vector_type(4))); v4i vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(d)); } else *d = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_store_blk(int4* __restrict__ dst, size_t n16, int per)
{
    size_t base = (size_t)blockIdx.x * 256 * per;
#pragma unroll 4
    for (int k = 0; k < per; k++) { size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n16) dst[i] = make_int4(0, 0, 0, 0); }
}
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ src, size_t n16, uint4* dst)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
template <typename F, typename P> float timeit(F f, P pre, int reps = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float tot = 0;
    for (int i = 0; i < reps + 2; i++) { pre(); CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i >= 2) tot += ms; }
    return tot / reps * 1000.f;
}
int main()
{
    const int N = 64, W = 1920, H = 1080, WW = 30;
    const size_t npx = (size_t)N * W * H;
    std::vector<u64> hb((size_t)N * H * WW, 0);
    for (size_t i = 0; i < hb.size(); i++) { size_t y = (i / WW) % H, j = i % WW; if ((y / 100) % 3 == 0 && (j % 7) < 2) hb[i] = ~0ull; if ((y / 100) % 3 == 0 && (j % 7) == 2) hb[i] = 0xffffull; }
    u64* d_bits; u32* d_sl; int32_t* d_lab; uint4 *d_a, *d_b;
    CK(hipMalloc(&d_bits, hb.size() * 8)); CK(hipMalloc(&d_sl, 4096)); CK(hipMalloc(&d_lab, npx * 4));
    const size_t other = 400u << 20; CK(hipMalloc(&d_a, other)); CK(hipMalloc(&d_b, other));
    CK(hipMemcpy(d_bits, hb.data(), hb.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(d_sl, 1, 4096)); CK(hipMemset(d_a, 1, other));
    const u32 total_rows = N * H, gpr = W / 4, magic = (u32)((0x100000000ull + gpr - 1) / gpr);
    auto nop = [&] {};
    auto dirty = [&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, d_a, other / 16, d_b); };   // leaves 400 MB of dirty lines around
    const double bytes = (double)npx * 4;
#define RUN(MODE, ROWS, name) { float t = timeit([&] { hipLaunchKernelGGL((k_write<MODE, ROWS>), dim3((total_rows + ROWS - 1) / ROWS), dim3(256), 0, 0, d_bits, W, WW, H, d_sl, d_lab, total_rows, gpr, magic); }, nop); \
    float t2 = timeit([&] { hipLaunchKernelGGL((k_write<MODE, ROWS>), dim3((total_rows + ROWS - 1) / ROWS), dim3(256), 0, 0, d_bits, W, WW, H, d_sl, d_lab, total_rows, gpr, magic); }, dirty); \
    printf("%-46s %7.1f us (%.2f TB/s) | after 400MB copy: %7.1f us\n", name, t, bytes / t / 1e6, t2); }
    RUN(0, 8, "rows=8 stores only");
    RUN(1, 8, "rows=8 +bits loads");
    RUN(3, 8, "rows=8 +bits +label loads");
    RUN(7, 8, "rows=8 +bits +label, nontemporal stores");
    RUN(4, 8, "rows=8 stores only, nontemporal");
    RUN(3, 2, "rows=2 +bits +label loads");
    RUN(3, 4, "rows=4 +bits +label loads");
    RUN(3, 16, "rows=16 +bits +label loads");
    const size_t n16 = npx * 4 / 16;
    for (int per : {4, 16}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_store_blk, dim3((unsigned)((n16 + 256 * per - 1) / (256 * per))), dim3(256), 0, 0, (int4*)d_lab, n16, per); }, nop);
        float t2 = timeit([&] { hipLaunchKernelGGL(k_store_blk, dim3((unsigned)((n16 + 256 * per - 1) / (256 * per))), dim3(256), 0, 0, (int4*)d_lab, n16, per); }, dirty);
        printf("plain store blocked per=%2d %27s %7.1f us (%.2f TB/s) | after 400MB copy: %7.1f us\n", per, "", t, bytes / t / 1e6, t2);
    }
    return 0;
}
