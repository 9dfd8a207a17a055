// the policy loop's Q-network kernels (susnet_qnet.h) for one feature layout -- a translation unit of its own: the library's largest kernels
#include "susnet_qnet.h"
namespace susnet {
SUSNET_QNET_FOR(, QRowC, QSpec2)
}
