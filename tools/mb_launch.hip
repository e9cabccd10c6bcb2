// Microbenchmark: what does a launch cost on the chain's stream when its blocks leave at once?
// (Decides whether the crowded-frame fallback of the labelling can be launched unconditionally.)
//   hipcc -O3 --offload-arch=gfx950 tools/mb_launch.hip -o tools/mb_launch && tools/mb_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_exit(const int* __restrict__ flag, int* __restrict__ out)
{
    if (flag[blockIdx.y] == 0) return;
    out[blockIdx.y * gridDim.x + blockIdx.x] = 1;
}
// a chain of `hops` dependent global loads by one wave per block (what a bookkeeping kernel's floor looks like)
__global__ void k_chase(const int* __restrict__ next, int* __restrict__ out, int hops)
{
    int p = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < hops; i++) p = next[p];
    if (p == -1) out[0] = p;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    int *flag, *out, *next;
    CK(hipMalloc(&flag, 4096 * 4));
    CK(hipMemset(flag, 0, 4096 * 4));
    CK(hipMalloc(&out, 1 << 24));
    const int NN = 1 << 22;
    CK(hipMalloc(&next, NN * 4));
    std::vector<int> h(NN);
    for (int i = 0; i < NN; i++) h[i] = (int)(((long long)i * 40503 + 12345) % NN);
    CK(hipMemcpy(next, h.data(), NN * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 2000;
    struct { int gx, gy, bs; } cfg[] = {{1, 128, 64}, {1, 128, 256}, {16, 128, 256}, {34, 128, 256}, {270, 128, 256}};
    for (auto c : cfg) {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_exit, dim3(c.gx, c.gy), dim3(c.bs), 0, s, flag, out);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_exit, dim3(c.gx, c.gy), dim3(c.bs), 0, s, flag, out);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("exit-at-once kernel grid (%d,%d) x %d threads: %.2f us per launch\n", c.gx, c.gy, c.bs, 1e3 * ms / reps);
    }
    for (int hops : {1, 2, 4, 8}) {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_chase, dim3(128), dim3(64), 0, s, next, out, hops);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_chase, dim3(128), dim3(64), 0, s, next, out, hops);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dependent-load chain, 128 blocks x 1 wave, %d hops: %.2f us per launch\n", hops, 1e3 * ms / reps);
    }
    return 0;
}
