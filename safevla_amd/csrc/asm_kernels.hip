// Loader of the embedded assembly code object (see asm_kernels.h).
#include "asm_kernels.h"
#include "common.h"
#include <map>
#include <mutex>
#include <string>

extern "C" const unsigned char svla_asm_hsaco[];       // _obj/asm_blob.hip (generated)
extern "C" const unsigned long svla_asm_hsaco_len;
extern "C" const char* const svla_asm_kernel_names[];  // nullptr-terminated

namespace {
std::mutex g_mu;
hipModule_t g_mod = nullptr;
std::map<std::string, hipFunction_t> g_fn;
int load_locked() {
    if (g_mod) return 0;
    HIP_CHECK_RET(hipModuleLoadData(&g_mod, (const void*)svla_asm_hsaco));
    for (int i = 0; svla_asm_kernel_names[i]; ++i) {
        hipFunction_t f = nullptr;
        HIP_CHECK_RET(hipModuleGetFunction(&f, g_mod, svla_asm_kernel_names[i]));
        g_fn[svla_asm_kernel_names[i]] = f;
    }
    return 0;
}
}  // namespace

int svla_asm_preload() {
    std::lock_guard<std::mutex> lk(g_mu);
    return load_locked();
}

int svla_asm_has(const char* name) {
    for (int i = 0; svla_asm_kernel_names[i]; ++i)
        if (std::string(svla_asm_kernel_names[i]) == name) return 1;
    return 0;
}

int svla_asm_launch(const char* name, const void* kernarg, size_t kernarg_bytes, int grid, int block, hipStream_t stream) {
    return svla_asm_launch2(name, kernarg, kernarg_bytes, grid, 1, block, stream);
}

int svla_asm_launch2(const char* name, const void* kernarg, size_t kernarg_bytes, int grid_x, int grid_y, int block, hipStream_t stream) {
    hipFunction_t f = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int rc = load_locked();
        if (rc) return rc;
        auto it = g_fn.find(name);
        if (it == g_fn.end()) return SVLA_EINVAL;
        f = it->second;
    }
    size_t sz = kernarg_bytes;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)kernarg, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HIP_CHECK_RET(hipModuleLaunchKernel(f, grid_x, grid_y, 1, block, 1, 1, 0, stream, nullptr, extra));
    return svla_launch_status();
}

extern "C" int svla_asm_launch_raw(const char* name, const void* kernarg, int kernarg_bytes, int grid, int block, void* stream) {
    if (!name || !kernarg || kernarg_bytes <= 0 || grid <= 0 || block <= 0) return SVLA_EINVAL;
    return svla_asm_launch(name, kernarg, (size_t)kernarg_bytes, grid, block, (hipStream_t)stream);
}
