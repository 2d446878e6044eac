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
}  // namespace

extern "C" {
// launches nblocks x 256 threads, each wave issuing nacc*iters MFMAs (2048 flop each); nacc = 8 or 20
int PAMD_mfma_f64_peak(double *d_out, int nblocks, int iters, int nacc, double scale, void *stream)
{
    if (nacc == 20) mfma_f64_peak_kernel<20><<<nblocks, 256, 0, (hipStream_t)stream>>>(d_out, iters, scale);
    else mfma_f64_peak_kernel<8><<<nblocks, 256, 0, (hipStream_t)stream>>>(d_out, iters, scale);
    PAMD_CHECK_LAUNCH();
    return 0;
}
}
