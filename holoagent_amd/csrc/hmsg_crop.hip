// Row N4: the encoder-side crop / resize batching -- every mask's bounding-box crop AND its background-blocked crop,
// resized to S x S (512), in ONE launch per frame instead of 2 M numpy slices + cv2.resize calls
// (reference: memory/hmsg/utils/sam_utils.py increase_bbox_by_margin :58-81, crop_all_bounding_boxs :119-147,
//  crop_image :150-164, crop_bbox :167-183; call sites perception/models/sam_clip_feats_extractor.py:148-151;
//  restated in oracle/crop_oracle.py).
//
// cv2.resize(crop, (512, 512)) = INTER_LINEAR on 8-bit data, OpenCV's fixed-point path (imgproc/resize.cpp): source
// index and weight per destination index from float((d + 0.5) * scale - 0.5), weights rounded to 1/2048, horizontal
// pass in int32, vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  The kernel evaluates
// exactly that per output pixel (the four taps come through L1 / L2; a crop's source window is a few hundred KB).
// The launch is bound by the HBM WRITE of the crops: 2 M S^2 3 bytes per frame (50 MB at M = 32, S = 512).
#include "hmsg_common.h"

#include <cmath>

namespace {

struct CropRect {
    int x0, y0, w, h;      // source window (already clipped to the image the way numpy clips a slice)
    int mask;              // -1: plain crop; m: multiply by segmentation m (crop_image)
    int out;               // index of the output crop in the destination array
    int variant;           // 0 plain, 1 masked
    int pad;
};

__device__ __forceinline__ void lin_coef(int d, double scale, int& s, float& f) {
    f = (float)__dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
    s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
}
__device__ __forceinline__ int coef_short(float w) {   // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)
    const float v = rintf(__fmul_rn(w, 2048.0f));
    return (int)fminf(fmaxf(v, -32768.0f), 32767.0f);
}

#define CROP_ROWS 16      // output rows per workgroup: the column coefficients of a thread's 4 pixels are reused for all of them

__global__ void __launch_bounds__(256) k_crop_resize(const unsigned char* __restrict__ image, const unsigned char* __restrict__ segs,
                                                     int H, int W, const CropRect* __restrict__ rects, int S,
                                                     unsigned char* __restrict__ out_plain, unsigned char* __restrict__ out_masked) {
    const CropRect r = rects[blockIdx.y];
    const int quads = S / 4, row0 = blockIdx.x * CROP_ROWS;
    __shared__ int s_y0[CROP_ROWS], s_y1[CROP_ROWS], s_b0[CROP_ROWS], s_b1[CROP_ROWS];
    if (threadIdx.x < CROP_ROWS && row0 + (int)threadIdx.x < S) {
        const double scale_y = __ddiv_rn(1.0, __ddiv_rn((double)S, (double)r.h));
        int sy;
        float fy;
        lin_coef(row0 + threadIdx.x, scale_y, sy, fy);
        s_b0[threadIdx.x] = coef_short(__fsub_rn(1.0f, fy));
        s_b1[threadIdx.x] = coef_short(fy);
        s_y0[threadIdx.x] = r.y0 + min(max(sy, 0), r.h - 1);       // rows are clamped one by one, weights untouched
        s_y1[threadIdx.x] = r.y0 + min(max(sy + 1, 0), r.h - 1);
    }
    __syncthreads();
    const double scale_x = __ddiv_rn(1.0, __ddiv_rn((double)S, (double)r.w));
    const unsigned char* seg = r.mask >= 0 ? segs + (size_t)r.mask * H * W : nullptr;
    unsigned char* out = (r.variant ? out_masked : out_plain) + (size_t)r.out * S * S * 3;
    const int rows = min(CROP_ROWS, S - row0);
    const size_t img_bytes = (size_t)H * W * 3;
    int last_q = -1, xa[4], xb[4], a0[4], a1[4];
    for (int item = threadIdx.x; item < rows * quads; item += 256) {
        const int rl = item / quads, q = item - rl * quads;
        if (q != last_q) {
            last_q = q;
            for (int j = 0; j < 4; ++j) {
                int sx;
                float fx;
                lin_coef(q * 4 + j, scale_x, sx, fx);
                if (sx < 0) {
                    sx = 0;
                    fx = 0.0f;
                }
                if (sx >= r.w - 1) {
                    sx = r.w - 1;
                    fx = 0.0f;
                }
                a0[j] = coef_short(__fsub_rn(1.0f, fx));
                a1[j] = coef_short(fx);
                xa[j] = r.x0 + sx;
                xb[j] = r.x0 + min(sx + 1, r.w - 1);
            }
        }
        const int b0 = s_b0[rl], b1 = s_b1[rl];
        const size_t o0 = (size_t)s_y0[rl] * W, o1 = (size_t)s_y1[rl] * W;
        unsigned px[3] = {0u, 0u, 0u};          // 12 output bytes
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t p0 = o0 + xa[j], p1 = o1 + xa[j];
            // the two taps of a row are neighbouring pixels (6 bytes): ONE unaligned 8-byte load per row instead of six
            // byte loads (the L1 request rate, not HBM, bounds this kernel); the last pixels of the image, and the
            // clamped right edge where both taps are the same pixel, take the byte path
            unsigned long long w0, w1;
            const bool pair = xb[j] == xa[j] + 1;
            if (pair && p0 * 3 + 8 <= img_bytes && p1 * 3 + 8 <= img_bytes) {
                __builtin_memcpy(&w0, image + p0 * 3, 8);
                __builtin_memcpy(&w1, image + p1 * 3, 8);
            } else {
                const size_t q0 = o0 + xb[j], q1 = o1 + xb[j];
                w0 = w1 = 0ull;
                for (int c = 0; c < 3; ++c) {
                    w0 |= (unsigned long long)image[p0 * 3 + c] << (8 * c) | (unsigned long long)image[q0 * 3 + c] << (8 * (c + 3));
                    w1 |= (unsigned long long)image[p1 * 3 + c] << (8 * c) | (unsigned long long)image[q1 * 3 + c] << (8 * (c + 3));
                }
            }
            if (seg) {
                unsigned short m0, m1;                       // the two taps' mask bytes (neighbours when `pair`)
                if (pair) {
                    __builtin_memcpy(&m0, seg + p0, 2);
                    __builtin_memcpy(&m1, seg + p1, 2);
                } else {
                    m0 = (unsigned short)(seg[p0] | (seg[o0 + xb[j]] << 8));
                    m1 = (unsigned short)(seg[p1] | (seg[o1 + xb[j]] << 8));
                }
                if (!(m0 & 0xff)) w0 &= ~0xffffffull;
                if (!(m0 >> 8)) w0 &= ~0xffffff000000ull;
                if (!(m1 & 0xff)) w1 &= ~0xffffffull;
                if (!(m1 >> 8)) w1 &= ~0xffffff000000ull;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = (int)((w0 >> (8 * c)) & 0xff) * a0[j] + (int)((w0 >> (8 * (c + 3))) & 0xff) * a1[j];
                const int r1 = (int)((w1 >> (8 * c)) & 0xff) * a0[j] + (int)((w1 >> (8 * (c + 3))) & 0xff) * a1[j];
                int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                v = min(max(v, 0), 255);
                const int byte = j * 3 + c;
                px[byte >> 2] |= (unsigned)v << ((byte & 3) * 8);
            }
        }
        unsigned* d32 = (unsigned*)(out + ((size_t)(row0 + rl) * S + (size_t)q * 4) * 3);   // (12-byte groups: 4-byte aligned)
        d32[0] = px[0];
        d32[1] = px[1];
        d32[2] = px[2];
    }
}

bool crop_is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// numpy slice [a : a + n) of an axis of length L, a >= 0, n > 0
inline void clip_slice(long long a, long long n, int L, int& start, int& len) {
    const long long s = std::min<long long>(a, L), e = std::min<long long>(a + n, L);
    start = (int)s;
    len = (int)std::max<long long>(e - s, 0);
}

}  // namespace

extern "C" int hmsg_crop_resize_batch(int32_t device_id, int32_t H, int32_t W, const uint8_t* image, int32_t M, const uint8_t* segs,
                                      const double* bbox, double bbox_margin, int32_t out_size, uint8_t* out_plain,
                                      uint8_t* out_masked, double* device_ms) {
    if (H <= 0 || W <= 0 || !image || M < 0 || (M > 0 && !bbox) || out_size <= 0 || (out_size & 3)) return HMSG_ERR_INVALID;
    if (out_masked && !segs) return HMSG_ERR_INVALID;
    if (M == 0 || (!out_plain && !out_masked)) return HMSG_OK;
    std::vector<CropRect> rects;
    for (int v = 0; v < 2; ++v) {
        if (!(v ? out_masked : out_plain)) continue;
        for (int m = 0; m < M; ++m) {
            double x = bbox[m * 4], y = bbox[m * 4 + 1], w = bbox[m * 4 + 2], h = bbox[m * 4 + 3];
            if (v == 0) {                       // crop_bbox: increase_bbox_by_margin (sam_utils.py:67-81)
                x -= bbox_margin;
                y -= bbox_margin;
                w += bbox_margin * 2;
                h += bbox_margin * 2;
                if (x < 0) {
                    w += x;
                    x = 0;
                }
                if (y < 0) {
                    h += y;
                    y = 0;
                }
            }
            CropRect r{0, 0, 0, 0, v ? m : -1, m, v, 0};
            const bool finite = std::fabs(x) < 1e15 && std::fabs(y) < 1e15 && std::fabs(w) < 1e15 && std::fabs(h) < 1e15;   // (NaN fails)
            const long long xi = finite ? (long long)x : -1, yi = finite ? (long long)y : -1, wi = finite ? (long long)w : 0,
                            hi = finite ? (long long)h : 0;                                                // int(): toward zero
            if (xi >= 0 && yi >= 0 && wi > 0 && hi > 0) {
                clip_slice(xi, wi, W, r.x0, r.w);
                clip_slice(yi, hi, H, r.y0, r.h);
            }
            if (r.w <= 0 || r.h <= 0) {
                fprintf(stderr, "hmsg_crop_resize_batch: mask %d has an empty %s crop (cv2.resize raises on it)\n", m,
                        v ? "masked" : "bounding-box");
                return HMSG_ERR_INVALID;
            }
            rects.push_back(r);
        }
    }
    hipStream_t s = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int rc = HMSG_OK;
    try {
        HIP_TRY(hipSetDevice(device_id));
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        {
            const size_t S = (size_t)out_size, crop_bytes = S * S * 3, img_bytes = (size_t)H * W * 3, seg_bytes = (size_t)M * H * W;
            DevBuf<unsigned char> d_img, d_seg, d_plain, d_masked;
            DevBuf<CropRect> d_rects;
            const unsigned char* p_img = image;
            const unsigned char* p_seg = segs;
            if (!crop_is_device_ptr(image)) {
                d_img.alloc(img_bytes);
                HIP_TRY(hipMemcpyAsync(d_img.p, image, img_bytes, hipMemcpyHostToDevice, s));
                p_img = d_img.p;
            }
            if (out_masked && !crop_is_device_ptr(segs)) {
                d_seg.alloc(seg_bytes);
                HIP_TRY(hipMemcpyAsync(d_seg.p, segs, seg_bytes, hipMemcpyHostToDevice, s));
                p_seg = d_seg.p;
            }
            unsigned char* p_plain = out_plain;
            unsigned char* p_masked = out_masked;
            const bool plain_host = out_plain && !crop_is_device_ptr(out_plain), masked_host = out_masked && !crop_is_device_ptr(out_masked);
            if (plain_host) {
                d_plain.alloc(crop_bytes * M);
                p_plain = d_plain.p;
            }
            if (masked_host) {
                d_masked.alloc(crop_bytes * M);
                p_masked = d_masked.p;
            }
            d_rects.alloc(rects.size());
            HIP_TRY(hipMemcpyAsync(d_rects.p, rects.data(), rects.size() * sizeof(CropRect), hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(ev0, s));
            hipLaunchKernelGGL(k_crop_resize, dim3(cdiv(S, CROP_ROWS), (unsigned)rects.size()), dim3(256), 0, s, p_img, p_seg, H, W,
                               (const CropRect*)d_rects.p, out_size, p_plain, p_masked);
            HMSG_CHECK_LAUNCH();
            HIP_TRY(hipEventRecord(ev1, s));
            if (plain_host) HIP_TRY(hipMemcpyAsync(out_plain, p_plain, crop_bytes * M, hipMemcpyDeviceToHost, s));
            if (masked_host) HIP_TRY(hipMemcpyAsync(out_masked, p_masked, crop_bytes * M, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
            if (device_ms) *device_ms = ms;
        }
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_crop_resize_batch: %s\n", e.msg.c_str());
        rc = e.code;
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (s) (void)hipStreamDestroy(s);
    return rc;
}
