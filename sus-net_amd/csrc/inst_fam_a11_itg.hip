// the byte-parallel family: 11 agents, ITG -- one translation unit of the parallel build (tools/gen_family.py, susnet_family.h)
#include "susnet_family.h"
namespace susnet {
SUSNET_FAMILY_INSTANTIATE(11, SUSNET_VARIANT_ITG, 0, 1)
}
