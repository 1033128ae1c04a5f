// amvs_prep.hip -- image preparation on the device (reference: PatchMatchMVS._prepare_images
// mvs_patchmatch.py:167-191, DenseStereoReconstructor._prepare_images dense_stereo.py:156-176):
//     img_scaled = cv.resize(img, (new_w, new_h))            INTER_LINEAR, 8-bit, 3 channels
//     gray = cv.cvtColor(img_scaled, cv.COLOR_BGR2GRAY).astype(np.float32) / 255.0
// OpenCV is a third-party dependency of the reference (requirements.txt: opencv-python>=4.5.0) that is
// absent from the build container, so this restates its published algorithm (imgproc/resize.cpp
// resizeGeneric_ + HResizeLinear + VResizeLinear<uchar>, imgproc/color_rgb.simd.hpp RGB2Gray<uchar>)
// and is checked bit for bit against the NumPy restatement in core/imageprep.py; parity with cv2 itself
// is UNPINNED.
//   resize:  per destination column dx:  fx = (float)((dx + 0.5) * (src_w / dst_w) - 0.5);
//            sx = floor(fx); fx -= sx; clamped at both ends (fx = 0); weights a0 = cvRound((1 - fx) *
//            2048), a1 = cvRound(fx * 2048) as shorts -- built on the host in float32 exactly as
//            OpenCV does and passed as tables; the same for rows (rows clamped, weights kept).
//            horizontal: D = S[sx] * a0 + S[sx+1] * a1                                (int32)
//            vertical:   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
//   gray:    (B * 3735 + G * 19235 + R * 9798 + 16384) >> 15        (the 15-bit form of OpenCV >= 4.x)
#include "amvs_kernels.h"

namespace amvs {

__global__ __launch_bounds__(256) void prep_resize_bgr8_kernel(const unsigned char *__restrict__ src, int sh, int sw,
                                                               int dh, int dw, const int *__restrict__ xofs,
                                                               const short *__restrict__ ialpha,
                                                               const int *__restrict__ yofs,
                                                               const short *__restrict__ ibeta,
                                                               unsigned char *__restrict__ dst)
{
    const long long n = (long long)dh * dw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int dy = (int)(i / dw), dx = (int)(i - (long long)dy * dw);
        const int sx0 = xofs[dx], sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
        const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 > sh - 1 ? sh - 1 : sy0);
        sy1 = sy1 < 0 ? 0 : (sy1 > sh - 1 ? sh - 1 : sy1);
        const unsigned char *r0 = src + (long long)sy0 * sw * 3, *r1 = src + (long long)sy1 * sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int D0 = r0[3 * sx0 + c] * a0 + r0[3 * sx1 + c] * a1;
            const int D1 = r1[3 * sx0 + c] * a0 + r1[3 * sx1 + c] * a1;
            dst[3 * i + c] = (unsigned char)((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

__global__ __launch_bounds__(256) void prep_gray_kernel(const unsigned char *__restrict__ bgr, long long n,
                                                        float *__restrict__ gray)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const unsigned v = (bgr[3 * i] * 3735u + bgr[3 * i + 1] * 19235u + bgr[3 * i + 2] * 9798u + 16384u) >> 15;
        gray[i] = (float)v / 255.0f;                     // .astype(np.float32) / 255.0
    }
}

hipError_t launch_prep_bgr8(const unsigned char *src, int sh, int sw, int dh, int dw, const int *xofs,
                            const short *ialpha, const int *yofs, const short *ibeta, unsigned char *scaled,
                            float *gray, hipStream_t st)
{
    const long long n = (long long)dh * dw;
    const dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), blk(256);
    if (sh == dh && sw == dw) {
        hipError_t e = hipMemcpyAsync(scaled, src, 3 * n, hipMemcpyDeviceToDevice, st);     // cv.resize to the same size copies
        if (e != hipSuccess) return e;
    } else {
        hipLaunchKernelGGL(prep_resize_bgr8_kernel, grid, blk, 0, st, src, sh, sw, dh, dw, xofs, ialpha, yofs, ibeta, scaled);
    }
    hipLaunchKernelGGL(prep_gray_kernel, grid, blk, 0, st, scaled, n, gray);
    return hipGetLastError();
}

}  // namespace amvs
