// Microbenchmark (development tool, not product): variants of the colour+inRange kernel and of a pure
// streaming store, to find the memory floor and the cost of the LDS LUT gathers on MI355X.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mb_color.hip cuauv-vision-pipeline_amd/csrc/vp_tables.cpp -o tools/mb_color
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>
typedef unsigned long long u64;
typedef unsigned int u32;
void vp_host_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180, int32_t* labC);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define BYTE_OF(arr, i) (((arr)[(i) >> 2] >> (8 * ((i)&3))) & 0xffu)
__device__ __forceinline__ u32 expand4(u32 nib) { return (((nib & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// VARIANT: 0 = plain LDS tables, 1 = no LUT (memory floor), 2 = gamma bank-replicated, 3 = 2 + cbrt x8, 4 = gamma replicated as packed u32 per bank
template <int VARIANT>
__global__ __launch_bounds__(256) void k_color(const uint8_t* __restrict__ src, size_t ngroups, const uint16_t* __restrict__ g_gamma,
                                               const uint16_t* __restrict__ g_cbrt, int lo, int hi, uint8_t* __restrict__ mask,
                                               u64* __restrict__ bits)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // layouts
    uint16_t* gam = (uint16_t*)smem;            // V0: [256]
    u32* gam32 = (u32*)smem;                    // V2/3: [128 rows][32 banks] each u32 = entries (2k, 2k+1)
    uint16_t* cb;                               // cbrt
    const int lane32 = threadIdx.x & 31;
    if (VARIANT == 0 || VARIANT == 1) {
        cb = (uint16_t*)(smem + 512);
        for (int i = threadIdx.x; i < 256; i += 256) gam[i] = g_gamma[i];
        for (int i = threadIdx.x; i < 2048; i += 256) cb[i] = g_cbrt[i];
    } else {
        cb = (uint16_t*)(smem + 16384);
        for (int i = threadIdx.x; i < 128 * 32; i += 256) { int k = i >> 5; gam32[i] = (u32)g_gamma[2 * k] | ((u32)g_gamma[2 * k + 1] << 16); }
        if (VARIANT == 2) { for (int i = threadIdx.x; i < 2048; i += 256) cb[i] = g_cbrt[i]; }
        else {
            // 8 copies, copy c in banks [4c, 4c+4): dword d of the table (entries 2d, 2d+1) at row d/4, bank 4c + d%4
            u32* cb32 = (u32*)cb;
            for (int i = threadIdx.x; i < 1024 * 8; i += 256) { int c = i & 7, d = i >> 3; cb32[(d >> 2) * 32 + c * 4 + (d & 3)] = (u32)g_cbrt[2 * d] | ((u32)g_cbrt[2 * d + 1] << 16); }
        }
    }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += stride) {
        const uint4* p = reinterpret_cast<const uint4*>(src + g * 48);
        const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
        const u32 in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
        u32 m = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int b = BYTE_OF(in, 3 * k), gg = BYTE_OF(in, 3 * k + 1), r = BYTE_OF(in, 3 * k + 2);
            bool ok;
            if (VARIANT == 1) { ok = (b + gg * 2 + r * 3) > lo * 4; }
            else {
                int R, G, B;
                if (VARIANT == 0) { R = gam[r]; G = gam[gg]; B = gam[b]; }
                else {
                    const u32 tr = gam32[(r >> 1) * 32 + lane32], tg = gam32[(gg >> 1) * 32 + lane32], tb = gam32[(b >> 1) * 32 + lane32];
                    R = (r & 1) ? (tr >> 16) : (tr & 0xffff); G = (gg & 1) ? (tg >> 16) : (tg & 0xffff); B = (b & 1) ? (tb >> 16) : (tb & 0xffff);
                }
                const int iy = (R * 871 + G * 2929 + B * 296 + 2048) >> 12;
                const int ix = (R * 1777 + G * 1541 + B * 778 + 2048) >> 12;
                int fY, fX;
                if (VARIANT <= 2) { fY = cb[iy]; fX = cb[ix]; }
                else {
                    const u32* cb32 = (const u32*)cb; const int c = lane32 & 7;
                    const u32 ty = cb32[((iy >> 1) >> 2) * 32 + c * 4 + ((iy >> 1) & 3)], tx = cb32[((ix >> 1) >> 2) * 32 + c * 4 + ((ix >> 1) & 3)];
                    fY = (iy & 1) ? (ty >> 16) : (ty & 0xffff); fX = (ix & 1) ? (tx >> 16) : (tx & 0xffff);
                }
                const int A = clamp255((500 * (fX - fY) + (128 << 15) + 16384) >> 15);
                ok = (A >= lo) & (A <= hi);
            }
            m |= (u32)ok << k;
        }
        uint4 o; o.x = expand4(m); o.y = expand4(m >> 4); o.z = expand4(m >> 8); o.w = expand4(m >> 12);
        reinterpret_cast<uint4*>(mask)[g] = o;
        u64 wv = (u64)m << (16 * (threadIdx.x & 3));
        wv |= __shfl_xor(wv, 1); wv |= __shfl_xor(wv, 2);
        if ((threadIdx.x & 3) == 0) bits[g >> 2] = wv;
    }
}

__global__ __launch_bounds__(256) void k_store(int4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = make_int4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_store_blk(int4* __restrict__ dst, size_t n16, int per)
{
    size_t base = (size_t)blockIdx.x * 256 * per;
#pragma unroll 4
    for (int k = 0; k < per; k++) { size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n16) dst[i] = make_int4(0, 0, 0, 0); }
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, size_t n16, u32* out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345) out[0] = acc;
}

template <typename F> float timeit(F f, int reps = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1000.f;
}

int main()
{
    const int N = 64, W = 1920, H = 1080;
    const size_t npx = (size_t)N * W * H;
    std::vector<uint8_t> h(npx * 3);
    u32 s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffff; };
    for (size_t i = 0; i < npx; i++) {
        int base[3] = {150, 110, 40};
        size_t x = i % W, y = (i / W) % H;
        bool disc = ((x / 200 + y / 200) % 5) == 0 && ((x % 200 - 100) * (x % 200 - 100) + (y % 200 - 100) * (y % 200 - 100) < 6400);
        if (disc) { base[0] = 40; base[1] = 45; base[2] = 210; }
        for (int c = 0; c < 3; c++) { int nz = (int)((rnd() % 13) + (rnd() % 13) + (rnd() % 13)) - 18; int v = base[c] + nz; h[i * 3 + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    }
    std::vector<uint16_t> gamma(256), cbrt(3072);
    vp_host_tables(gamma.data(), cbrt.data(), nullptr, nullptr, nullptr);
    uint8_t *d_src, *d_mask; u64* d_bits; uint16_t *d_g, *d_c; int4* d_lab; u32* d_out;
    CK(hipMalloc(&d_src, npx * 3)); CK(hipMalloc(&d_mask, npx)); CK(hipMalloc(&d_bits, npx / 8)); CK(hipMalloc(&d_g, 512)); CK(hipMalloc(&d_c, 6144));
    CK(hipMalloc(&d_lab, npx * 4)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_src, h.data(), npx * 3, hipMemcpyHostToDevice)); CK(hipMemcpy(d_g, gamma.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(d_c, cbrt.data(), 6144, hipMemcpyHostToDevice));
    const size_t ngroups = npx / 16;
    std::vector<uint8_t> ref(npx), got(npx);
    const double cbytes = (double)npx * 4.125;
    for (int blocks_per_cu : {4, 8, 12}) {
        dim3 grid(256 * blocks_per_cu);
        float t0 = timeit([&] { hipLaunchKernelGGL((k_color<0>), grid, dim3(256), 512 + 4096, 0, d_src, ngroups, d_g, d_c, 150, 255, d_mask, d_bits); });
        CK(hipMemcpy(ref.data(), d_mask, npx, hipMemcpyDeviceToHost));
        float t1 = timeit([&] { hipLaunchKernelGGL((k_color<1>), grid, dim3(256), 512 + 4096, 0, d_src, ngroups, d_g, d_c, 150, 255, d_mask, d_bits); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_color<2>), grid, dim3(256), 16384 + 4096, 0, d_src, ngroups, d_g, d_c, 150, 255, d_mask, d_bits); });
        CK(hipMemcpy(got.data(), d_mask, npx, hipMemcpyDeviceToHost)); bool ok2 = memcmp(ref.data(), got.data(), npx) == 0;
        float t3 = timeit([&] { hipLaunchKernelGGL((k_color<3>), grid, dim3(256), 16384 + 32768, 0, d_src, ngroups, d_g, d_c, 150, 255, d_mask, d_bits); });
        CK(hipMemcpy(got.data(), d_mask, npx, hipMemcpyDeviceToHost)); bool ok3 = memcmp(ref.data(), got.data(), npx) == 0;
        printf("blocks/CU %2d: V0 plain %.1f us (%.2f TB/s) | V1 noLUT %.1f us (%.2f TB/s) | V2 gammaRep %.1f us ok=%d | V3 +cbrtx8 %.1f us ok=%d\n", blocks_per_cu, t0,
               cbytes / t0 / 1e6, t1, cbytes / t1 / 1e6, t2, ok2, t3, ok3);
    }
    const size_t n16 = npx * 4 / 16;
    for (int bpc : {4, 8, 16, 32}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_store, dim3(256 * bpc), dim3(256), 0, 0, d_lab, n16); });
        printf("store grid-stride %2d blocks/CU: %.1f us (%.2f TB/s)\n", bpc, t, (double)npx * 4 / t / 1e6);
    }
    for (int per : {4, 8, 16}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_store_blk, dim3((unsigned)((n16 + 256 * per - 1) / (256 * per))), dim3(256), 0, 0, d_lab, n16, per); });
        printf("store blocked per=%2d: %.1f us (%.2f TB/s)\n", per, t, (double)npx * 4 / t / 1e6);
    }
    for (int bpc : {8, 16}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(256 * bpc), dim3(256), 0, 0, (const uint4*)d_src, npx * 3 / 16, d_out); });
        printf("read 16B/lane coalesced %2d blocks/CU: %.1f us (%.2f TB/s)\n", bpc, t, (double)npx * 3 / t / 1e6);
    }
    return 0;
}
