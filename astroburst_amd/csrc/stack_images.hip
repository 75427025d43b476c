// stack_images driver (core/stacking/combine.rs:94-193): crop to the minimum dims, register
// frames 1..n-1 on frame 0, then per-pixel kappa-sigma combine.
#include "ab_common.hpp"

int ab_stack_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                    const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev,
                    uint64_t *out_rejected, bool median_only);

extern "C" int ab_stack_images(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                               ab_plane_mut *out, int32_t *offsets_dy_dx, uint64_t *out_rejected) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");  // combine.rs:98-100
    AB_CHECK(ctx, cfg && out, "null config or output");
    int64_t min_rows = planes[0].rows, min_cols = planes[0].cols;  // combine.rs:104-105
    for (size_t i = 1; i < n; ++i) {
        min_rows = planes[i].rows < min_rows ? planes[i].rows : min_rows;
        min_cols = planes[i].cols < min_cols ? planes[i].cols : min_cols;
    }
    AB_CHECK(ctx, out->rows == min_rows && out->cols == min_cols, "output must be %lldx%lld (minimum frame dims)",
             (long long)min_rows, (long long)min_cols);
    if (cfg->align && n > 1)
        return ab_set_error(ctx, AB_ERR_UNSUPPORTED,
                            "stack_images(align=true): phase-correlation registration is not in this build yet; "
                            "register with ab_shift_image_subpixel / ab_warp_image and stack with align=false");
    if (offsets_dy_dx)
        for (size_t i = 0; i < 2 * n; ++i) offsets_dy_dx[i] = 0;  // combine.rs:121,140
    // the top-left crop (combine.rs:107-113) is the row stride handed to the kernel
    return ab_stack_sigma_clip(ctx, planes, n, cfg, out, out_rejected);
}
