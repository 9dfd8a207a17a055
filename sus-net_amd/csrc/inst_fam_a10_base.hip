// the byte-parallel family: 10 agents, BASE -- one translation unit of the parallel build (tools/gen_family.py, susnet_family.h)
#include "susnet_family.h"
namespace susnet {
SUSNET_FAMILY_INSTANTIATE(10, SUSNET_VARIANT_BASE, 1, 1)
SUSNET_FAMILY_INSTANTIATE(10, SUSNET_VARIANT_BASE, 0, 1)
}
