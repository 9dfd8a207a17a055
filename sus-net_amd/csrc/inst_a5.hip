// one compiled-in configuration per translation unit (parallel build): see susnet_kernels.h
#include "susnet_kernels.h"
namespace susnet {
SUSNET_INSTANTIATE(SpecA<5>)
}
