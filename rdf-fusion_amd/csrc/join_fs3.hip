// join_fs3.hip — the lds_join_kernel instantiations with join-filter shape FS = 3 (one translation unit per shape so
// that the build parallelises; the code is join_device.hpp).
#include "join_device.hpp"

namespace rdfgpu {
RDFGPU_DEFINE_JOIN_FS(3)
}  // namespace rdfgpu
