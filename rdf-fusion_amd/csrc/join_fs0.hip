// join_fs0.hip — the lds_join_kernel instantiations with join-filter shape FS = 0 (one translation unit per shape so
// that the build parallelises; the code is join_device.hpp).
#include "join_device.hpp"

namespace rdfgpu {
RDFGPU_DEFINE_JOIN_FS(0)
// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_join_fs0() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>((lds_join_kernel<0, 0, 1, kJoinTableLds, false, 1>)))); }
}  // namespace rdfgpu
