// rsx_api.hip -- library plumbing of librsx: version, error text, device info.
#include <stdarg.h>
#include <string.h>

#include "rsx_common.h"

static thread_local char g_err[512] = "";

void rsx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rsx_num_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// LDS per CU of the current device in bytes (cached; 0 if the runtime does not say)
int rsx_lds_per_cu()
{
    static int lds[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (lds[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || n <= 0) n = -1;
        lds[dev] = n;
    }
    return lds[dev] > 0 ? lds[dev] : 0;
}

int g_rsx_score_lanes = 2;
int g_rsx_sort_cap = 0;
int g_rsx_apply_stream = 0;
int g_rsx_step_waves = 0;
int g_rsx_mesh_blocks = 0;
int g_rsx_touched_apply = 1;

RSX_API int rsx_set_option(const char *name, int64_t value)
{
    RSX_CHECK_ARG(name != nullptr, "null option name");
    if (strcmp(name, "score_lanes") == 0) {
        RSX_CHECK_ARG(value >= 1 && value <= 4, "score_lanes must be in [1, 4]");
        g_rsx_score_lanes = (int)value;
        return RSX_OK;
    }
    if (strcmp(name, "sample_sort_cap") == 0) {
        RSX_CHECK_ARG(value >= 0 && value <= 2048, "sample_sort_cap must be in [0, 2048] (0 = default)");
        g_rsx_sort_cap = (int)value;
        return RSX_OK;
    }
    if (strcmp(name, "step_waves") == 0) {
        RSX_CHECK_ARG(value == 0 || (value >= 2 && value <= 8), "step_waves must be 0 (default) or in [2, 8]");
        g_rsx_step_waves = (int)value;
        return RSX_OK;
    }
    if (strcmp(name, "apply_stream") == 0) {
        RSX_CHECK_ARG(value == 0 || value == 1, "apply_stream must be 0 or 1");
        g_rsx_apply_stream = (int)value;
        return RSX_OK;
    }
    if (strcmp(name, "touched_apply") == 0) {
        RSX_CHECK_ARG(value >= 0 && value <= 2, "touched_apply must be 0, 1 or 2");
        g_rsx_touched_apply = (int)value;
        return RSX_OK;
    }
    if (strcmp(name, "mesh_blocks") == 0) {
        RSX_CHECK_ARG(value >= 0 && value <= 4096, "mesh_blocks must be in [0, 4096] (0 = one workgroup per CU)");
        g_rsx_mesh_blocks = (int)value;
        return RSX_OK;
    }
    rsx_set_error("rsx_set_option: unknown option '%s'", name);
    return RSX_E_INVALID;
}

RSX_API int rsx_version(void) { return RSX_ABI_VERSION; }

RSX_API const char *rsx_last_error(void) { return g_err; }

RSX_API int rsx_device_info_get(int device, rsx_device_info *out)
{
    RSX_CHECK_ARG(out != nullptr, "null output");
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        rsx_set_error("rsx_device_info_get: %s", hipGetErrorString(e));
        return RSX_E_HIP;
    }
    memset(out, 0, sizeof(*out));
    out->device = device;
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->total_mem_bytes = (int64_t)p.totalGlobalMem;
    out->lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    out->clock_khz = p.clockRate;
    strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
    return RSX_OK;
}
