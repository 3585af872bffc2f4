// depth.hip — LiDAR depth assignment (filled in below; placeholder keeps the ABI complete while the BA path is
// brought up on the GPU).
#include <hip/hip_runtime.h>

#include "../../include/limo_hip.h"

extern "C" {

void limo_depth_default_params(limo_depth_params* p) {
    if (!p) return;
    p->pixelarea_search_width = 6;
    p->pixelarea_search_height = 9;
    p->pixelarea_search_offset_x = 0;
    p->pixelarea_search_offset_y = 0;
    p->neighbors_count_min = 3;
    p->do_use_histogram_segmentation = 1;
    p->histogram_segmentation_bin_width = 0.3;
    p->histogram_segmentation_min_pointcount = 1;
    p->treshold_depth_enabled = 1;
    p->treshold_depth_max = 100.0;
    p->treshold_depth_min = 0.0;
    p->treshold_depth_local_enabled = 1;
    p->treshold_depth_local_valuetype = 1;
    p->treshold_depth_local_value = 0.5;
    p->do_use_cut_behind_camera = 1;
    p->do_use_triangle_size_maximation = 1;
    p->do_check_triangleplanar_condition = 1;
    p->triangleplanar_crossnorm_treshold = 0.1;
    p->viewray_plane_orthoganality_treshold = 0.1;
    p->do_use_ransac_plane = 1;
    p->ransac_plane_distance_treshold = 0.2;
    p->ransac_plane_min_z = -3.5;
    p->ransac_plane_max_z = -1.0;
    p->ransac_plane_max_iterations = 600;
    p->ransac_plane_probability = 0.99;
    p->ransac_plane_use_refinement = 1;
    p->ransac_plane_refinement_treshold = 10.2;
    p->ransac_plane_point_distance_treshold = 0.2;
    p->plane_estimator_use_mestimator = 1;
    p->ransac_seed = 1;
}

int limo_depth_estimate(limo_ctx*, const float*, size_t, const double*, double, double, double, int32_t, int32_t,
                        const float*, size_t, const uint8_t*, const limo_depth_params*, float*) {
    return LIMO_ERR_RUNTIME;
}
}
