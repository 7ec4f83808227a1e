/*
 * gip_pose.h — C-ABI of the OpenPose control-map drawer.
 *
 *   gip_openpose_draw  <->  Skeleton.openpose_draw's canvas part (threestudio/utils/poser.py:832-904): 18 filled
 *       key-point discs of radius 4 (cv2.circle, thickness -1) then 17 limbs, each an ellipse polygon
 *       (cv2.ellipse2Poly + cv2.fillConvexPoly on a copy) blended 0.4 / 0.6 into the canvas (cv2.addWeighted), uint8
 *       arithmetic, output float32 / 255.  The reference draws on the CPU per view and ships each map to the GPU
 *       (GaussianIP.py:175-196: one D2H of mvp + one H2D per view); here all V maps of a step are one launch and the
 *       projected key points never leave the device.
 *
 * Every pixel replays the draw order on its own byte triple: disc i if visible[i] (midpoint-circle footprint of
 * radius 4 around (int(x_i), int(y_i))), then for limb l with both ends visible: inside the ellipse with centre
 * (int(mean x), int(mean y)), half-axes (int(len / 2), 4), angle int(degrees(atan2(y0 - y1, x0 - x1))) the blend source
 * is the limb colour, outside it is the canvas itself; canvas = round_half_even(0.4 canvas + 0.6 source).
 * OpenCV is not available in the build environment: the ellipse footprint is the analytic ellipse inflated by half a
 * pixel rather than cv2's 1-degree polygon scan conversion, so boundary pixels may differ from cv2 — parity against
 * OpenCV itself is UNPINNED; tests pin the kernel bit-exactly against oracle/pose_oracle.py, which states this spec.
 *
 * points_px [V,18,2] int32 = (int(x), int(y)) of the projected key points; visible [V,18] uint8; limbs [V,17,6] float =
 * (centre x, centre y, half-axis a, drawn?, cos(angle), sin(angle)) per limb, prepared by the caller with tensor ops on the
 * device (gaussianip_amd/poser.py) so that the kernel holds only exactly reproducible arithmetic; out [V,H,W,3] float.
 * Status 0 ok, 1 bad argument, 3 HIP error.
 */
#ifndef GIP_POSE_H
#define GIP_POSE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define GIP_POSE_POINTS 18
#define GIP_POSE_LIMBS 17
int gip_openpose_draw(const int32_t* points_px, const uint8_t* visible, const float* limbs, float* out, int32_t V,
                      int32_t H, int32_t W, void* stream);
#ifdef __cplusplus
}
#endif
#endif
