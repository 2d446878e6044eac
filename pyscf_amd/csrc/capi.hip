// Library-level entry points of libpyscf_amd.so.
#include "common.h"

namespace pamd {
thread_local char g_errmsg[512] = "";
}

extern "C" {

const char *PAMD_last_error(void) { return pamd::g_errmsg; }

int PAMD_version(void) { return 100; }

// Number of visible HIP devices (0 when no GPU/driver is present); never fails.
int PAMD_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int PAMD_set_device(int dev)
{
    PAMD_CHECK_HIP(hipSetDevice(dev));
    return 0;
}

int PAMD_stream_synchronize(void *stream)
{
    PAMD_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"

// ---- micro-benchmark: register-only v_mfma_f64_16x16x4_f64 issue rate (measures the practical
// FP64 matrix ceiling of the chip under its power budget; used by bench/tools, not by the product path)
namespace {
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double *out, int iters, double scale)
{
    pamd::double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = pamd::double4_t{0, 0, 0, 0};
    // scale = 0 -> all-zero operands (lowest switching power); scale = 1 -> pseudo-random operands
    double a = scale * (0.37 + 0.61 * sin(1.0 + threadIdx.x)), b = scale * (0.53 - 0.45 * cos(2.0 + threadIdx.x));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++)   // inline asm: keeps the accumulators in place (no AGPR shuffles)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;     // keep the result alive
}
// The k-loop of the J/K kernels WITHOUT its memory system: per k-group of 4 a wave reads 5 + 4 fresh operand fragments from LDS
// (pseudo-random doubles, a different set every group - live data toggling, as the real operands) and issues 20 MFMAs on 20
// accumulators; no global loads, no barriers, no DMA.  What this sustains is the practical FP64 matrix ceiling of the chip under
// its power limit for THIS instruction mix - the number the measured kernels should be compared with next to the 78.6 TF/s
// data-sheet figure (which a register-only stream reaches only on all-zero operands).
__global__ __launch_bounds__(256, 2) void mfma_f64_live_kernel(double *out, int iters, unsigned seed)
{
    __shared__ double lds[2048 + 9 * 64];
    for (int i = threadIdx.x; i < 2048 + 9 * 64; i += 256) {
        unsigned h = (i * 2654435761u) ^ (seed + blockIdx.x * 40503u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        lds[i] = ((h & 0xffffff) / 16777216.0 - 0.5) * 2.0;
    }
    __syncthreads();
    pamd::double4_t acc[5][4];
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = pamd::double4_t{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const int base = (it * 37) & 2047;
        double af[5], bf[4];
#pragma unroll
        for (int a = 0; a < 5; a++) af[a] = lds[base + a * 64 + lane];
#pragma unroll
        for (int b = 0; b < 4; b++) bf[b] = lds[base + (5 + b) * 64 + lane];
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = pamd::mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (s == 12345.678) out[0] = s;
}
}  // namespace

extern "C" {
// nblocks x 256 threads (2 workgroups per CU resident), each wave issuing 20 * iters MFMAs (2048 flop each) on live LDS operands
int PAMD_mfma_f64_live(double *d_out, int nblocks, int iters, void *stream)
{
    mfma_f64_live_kernel<<<nblocks, 256, 0, (hipStream_t)stream>>>(d_out, iters, 12345u);
    PAMD_CHECK_LAUNCH();
    return 0;
}
// launches nblocks x 256 threads, each wave issuing nacc*iters MFMAs (2048 flop each); nacc = 8 or 20
int PAMD_mfma_f64_peak(double *d_out, int nblocks, int iters, int nacc, double scale, void *stream)
{
    if (nacc == 20) mfma_f64_peak_kernel<20><<<nblocks, 256, 0, (hipStream_t)stream>>>(d_out, iters, scale);
    else mfma_f64_peak_kernel<8><<<nblocks, 256, 0, (hipStream_t)stream>>>(d_out, iters, scale);
    PAMD_CHECK_LAUNCH();
    return 0;
}
}
