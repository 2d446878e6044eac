// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA) accept source addresses that are only 8-byte aligned, and what does it cost?
//   hipcc --offload-arch=gfx950 -O3 -o dma_align_probe dma_align_probe.hip && ./dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double *src, double *out, int shift_doubles, int iters)
{
    __shared__ double s[64 * 2 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(src + shift_doubles), 0, 0xffffffff, 0x00020000);
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(s + wave * 128), 16, lane * 16,
                                                 (it * 4 + wave) * 1024 + blockIdx.x * 65536, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += s[wave * 128 + lane * 2] + s[wave * 128 + lane * 2 + 1];
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const size_t n = 1 << 26;
    std::vector<double> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (double)(i % 1000003) * 0.5;
    double *d, *o;
    hipMalloc(&d, n * 8);
    hipMalloc(&o, 1024 * 256 * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<double> ho(1024 * 256);
    for (int shift = 0; shift < 4; shift++) {
        const int iters = 16, nblk = 512;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<nblk, 256>>>(d, o, shift, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; rep++) probe<<<nblk, 256>>>(d, o, shift, iters);
        hipEventRecord(e1);
        hipError_t err = hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(ho.data(), o, nblk * 256 * 8, hipMemcpyDeviceToHost);
        // expected: thread (b, t): sum over it of src[shift + ((it*4+wave)*1024 + b*65536)/8 + lane*2 (+1)]
        double maxerr = 0;
        for (int b = 0; b < nblk; b += 37)
            for (int t = 0; t < 256; t += 5) {
                const int lane = t & 63, wave = t >> 6;
                double want = 0;
                for (int it = 0; it < iters; it++) {
                    size_t base = shift + ((size_t)(it * 4 + wave) * 1024 + (size_t)b * 65536) / 8 + lane * 2;
                    want += h[base] + h[base + 1];
                }
                double e = ho[b * 256 + t] - want;
                if (e < 0) e = -e;
                if (e > maxerr) maxerr = e;
            }
        printf("shift %d doubles (%s): err=%s maxerr=%g  %.3f ms per launch\n", shift, (shift & 1) ? "8-byte aligned only" : "16-byte aligned",
               hipGetErrorString(err), maxerr, ms / 20);
    }
    return 0;
}
