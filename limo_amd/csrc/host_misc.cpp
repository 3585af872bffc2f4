// host_misc.cpp — C-ABI entry points that run on the host: the quantile trimmer exposed for callers / tests (the
// batched solve trims on the device, kba_kernels.hip) and, ONLY inside the emulated ABI of the CPU test tier, the host
// loop of the landmark initialisation (the product's limo_landmark_init is the device kernel of landmark_init.hip).
//
//   limo_landmark_init   BundleAdjusterKeyframes::calculateLandmark (both overloads),
//                        keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:332-382,
//                        internal/triangulator.hpp:51-75, convertMeasurementToRay definitions.cpp:98-102
//   limo_trim_quantile   TrimmerQuantile::getOutliers,
//                        robust_optimization/include/robust_optimization/internal/trimmer_quantile.hpp:40-63
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/limo_hip.h"
#include "landmark_init.hpp"

extern "C" {

#ifdef KBA_EMU_EXPORT_ABI
// Host loop over the SAME per-landmark statements the device kernel runs (landmark_init.hpp) - only inside the
// emulated C-ABI of the CPU test tier; the product's limo_landmark_init is landmark_init.hip.
int limo_landmark_init(limo_ctx* /*ctx*/, int32_t n, const int32_t* ray_off, const limo_ray* rays,
                       const uint8_t* use_depth, double* pos_out, uint8_t* ok) {
    if (n < 0 || (n > 0 && (!ray_off || !rays || !use_depth || !pos_out || !ok))) return LIMO_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (ray_off[i + 1] < ray_off[i]) return LIMO_ERR_INVALID;
        double p[3] = {0.0, 0.0, 0.0};
        ok[i] = kba::lminit_one(ray_off, rays, use_depth, i, p) ? 1 : 0;
        for (int k = 0; k < 3; ++k) pos_out[3 * (size_t)i + k] = p[k];
    }
    return LIMO_OK;
}
#endif

int limo_trim_quantile(int32_t n, const int64_t* ids, const double* values, double quantile, int64_t* outliers_out) {
    if (n < 0 || (n > 0 && (!ids || !values || !outliers_out))) return LIMO_ERR_INVALID;
    // the reference copies a std::map (unique ids, ascending) into a vector before nth_element
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        return values[a] < values[b] || (values[a] == values[b] && ids[a] < ids[b]);
    });
    const int num = static_cast<int>(static_cast<double>(n) * quantile);
    int k = 0;
    for (int i = std::max(num, 0); i < n; ++i) outliers_out[k++] = ids[order[i]];
    return k;
}

}  // extern "C"
