// brt_trace_prod.hip -- the production instantiations of k_trace_persistent: TUNABLE = false.
#include "brt_trace.h"

namespace brt {
hipError_t launch_trace_persistent_prod(const TraceLaunch& tl) { return launch_persistent_all<false>(tl); }
}  // namespace brt
