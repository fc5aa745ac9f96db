// brt_trace_tune.hip -- k_trace_persistent with the tuning knobs live (TUNABLE = true): chosen by
// brt_api.cpp when any BRT_* tuning variable differs from its default (experiments, knob tests).
#include "brt_trace.h"

namespace brt {
hipError_t launch_trace_persistent_tune(const TraceLaunch& tl) { return launch_persistent_all<true>(tl); }
}  // namespace brt
