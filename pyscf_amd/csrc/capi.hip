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
