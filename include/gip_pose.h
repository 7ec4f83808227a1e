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
 * Every pixel replays the draw order on its own byte triple: disc i if visible[i] (cv::Circle's filled midpoint circle of
 * radius 4 around (int(x_i), int(y_i))), then for limb l with both ends visible the footprint OpenCV paints for
 *   cv2.fillConvexPoly(canvas, cv2.ellipse2Poly((int(mean x), int(mean y)), (int(len / 2), 4), int(degrees(atan2(y0 - y1, x0 - x1))), 0, 360, 1), colour)
 * is the blend source, elsewhere the canvas itself; canvas = round_half_even(0.4 canvas + 0.6 source) (cv2.addWeighted).
 * The footprint is computed as OpenCV 4.x computes it (modules/imgproc/src/drawing.cpp: SinTable / ellipse2Poly in double
 * arithmetic + cvRound + duplicate removal; FillConvexPoly = outline by clipLine + 8-connected LineIterator, then the
 * XY_SHIFT = 16 fixed-point scanline fill), one wave per (view, limb), and handed to the pixel kernel as one [lo, hi]
 * span per image row through `workspace` (gip_openpose_workspace_bytes(V, H) bytes).  opencv-python is a dependency of
 * the reference that is absent from its tree and not installed in the build environment: the restatement follows the
 * published source; parity against the OpenCV binary is unpinned; tests pin the kernel bit-exactly against
 * oracle/pose_oracle.py, which states the same routines in numpy.
 *
 * points_px [V,18,2] int32 = (int(x), int(y)) of the projected key points; visible [V,18] uint8; limbs [V,17,6] float =
 * (int centre x, int centre y, int half-axis a, drawn?, int angle in degrees, 0) per limb, prepared by the caller
 * (gaussianip_amd/poser.py) so that the kernels hold only exactly reproducible arithmetic; out [V,H,W,3] float.
 * H <= 2048, W <= 32766.
 * Status 0 ok, 1 bad argument, 2 workspace too small, 3 HIP error.
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
size_t gip_openpose_workspace_bytes(int32_t V, int32_t H);
int gip_openpose_draw(const int32_t* points_px, const uint8_t* visible, const float* limbs, float* out, int32_t V,
                      int32_t H, int32_t W, void* workspace, size_t workspace_bytes, void* stream);
#ifdef __cplusplus
}
#endif
#endif
