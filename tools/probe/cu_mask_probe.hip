// Which CUs does a CU-masked stream run on?  (tools/probe: one-off hardware probes, not product code.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
// Launches 4096 one-wave workgroups on streams created with hipExtStreamCreateWithCUMask and histograms (XCC_ID, SE, SH, CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    for (volatile int i = 0; i < 2000; i++) {}
}

static void run(const char *tag, const std::vector<unsigned> &mask)
{
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", tag, hipGetErrorString(e)); return; }
    const int n = 4096;
    unsigned *d, h[2 * n];
    hipMalloc(&d, sizeof(h));
    where<<<n, 64, 0, st>>>(d);
    hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    std::map<unsigned, std::set<unsigned>> cus;
    for (int i = 0; i < n; i++) cus[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xff);      // cu_id[11:8] sh[12] se[15:13]
    printf("%s:", tag);
    int tot = 0;
    for (auto &kv : cus) { printf(" xcc%u=%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total CUs %d\n", tot);
    hipFree(d);
    hipStreamDestroy(st);
}

int main()
{
    run("no mask", {});
    run("bits 0-15", {0xffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("every 16th bit", std::vector<unsigned>(8, 0x00010001u));
    run("all but every 16th", std::vector<unsigned>(8, ~0x00010001u));
    run("all 256", std::vector<unsigned>(8, 0xffffffffu));
    return 0;
}
