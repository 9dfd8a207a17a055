// the byte-parallel family: 12 agents, ITG -- one translation unit of the parallel build (tools/gen_family.py, susnet_family.h)
#include "susnet_family.h"
namespace susnet {
SUSNET_FAMILY_INSTANTIATE(12, SUSNET_VARIANT_ITG, 0, 1)
}
