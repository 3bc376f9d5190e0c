// kernels_post.hip — the sliding-window post-processing chain of sw_processing.py on device
// (SURVEY §8-f "next" rows 1-2): per-window head mean + min-max (:245, :253-254), the 8x bilinear
// upsample (:255-257), the overlap-blended stitcher concat_crops (:113-149) and the Otsu mask of the
// stitched heat map (threshold(), :37-81: min_max_normalize -> *255 -> uint8 -> cv2 THRESH_OTSU).
// All of it is HBM-bound elementwise / reduction work: coalesced loads, one pass per stage.
#include "launch.h"

// The reference evaluates these stages with separate IEEE multiplies and adds (numpy / cv2): forbid the
// compiler's default fused-multiply-add contraction so that roundings happen at the same points
// (HIP's __dmul_rn / __dadd_rn are inline functions compiled under the headers' contraction mode and
// still fuse; plain operators under this pragma do not).
#pragma clang fp contract(off)

__device__ __forceinline__ float block_reduce(float v, float *scratch, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, t) : fminf(v, t);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();  // scratch reuse
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, scratch[i]) : fminf(r, scratch[i]);
    return r;
}

// One workgroup per window. rows: (T, H, n_rows, P) CLS-row maps (row 0 is used); maps: (T, P).
//   avg = np.mean over heads (sequential fp32 sum, then / H)                       sw_processing.py:245
//   maps = (avg - avg.min()) / (avg.max() - avg.min()) * 255                       :253-254
// The reference takes the mean / min / max of the nearest-upsampled (x p) map: same values.
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void tile_post_kernel(const float *__restrict__ rows, float *__restrict__ maps,
                                                        int H, int n_rows, int P) {
    __shared__ float scratch[4];
    const int t = blockIdx.x;
    const float *src = rows + (size_t)t * H * n_rows * P;
    float *dst = maps + (size_t)t * P;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < P; i += 256) {
        float s = src[i];
        for (int h = 1; h < H; ++h) s = s + src[(size_t)h * n_rows * P + i];
        s = s / (float)H;
        dst[i] = s;
        mn = fminf(mn, s);
        mx = fmaxf(mx, s);
    }
    if (!NORMALIZE) return;  // eval.py:142 keeps the plain head mean
    mn = block_reduce(mn, scratch, false);
    mx = block_reduce(mx, scratch, true);
    const float range = mx - mn;
    for (int i = threadIdx.x; i < P; i += 256)  // each thread re-reads only what it wrote
        dst[i] = ((dst[i] - mn) / range) * 255.0f;
}

hipError_t launch_tile_postprocess(const float *rows, float *maps, int tiles, int heads, int n_rows, int pixels,
                                   hipStream_t s) {
    if (tiles <= 0) return hipSuccess;
    tile_post_kernel<true><<<dim3(tiles), dim3(256), 0, s>>>(rows, maps, heads, n_rows, pixels);
    return hipGetLastError();
}

hipError_t launch_head_mean(const float *rows, float *maps, int tiles, int heads, int n_rows, int pixels, hipStream_t s) {
    if (tiles <= 0) return hipSuccess;
    tile_post_kernel<false><<<dim3(tiles), dim3(256), 0, s>>>(rows, maps, heads, n_rows, pixels);
    return hipGetLastError();
}

// cv2.resize(src, (w*scale, h*scale), interpolation=INTER_LINEAR) for float32 (:257; the preceding
// cv2.resize down by 8 of the nearest-upsampled map returns the hf x wf map itself): half-pixel
// centres, border replicate, horizontal pass then vertical pass in fp32.
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          int h, int w, int scale) {
    const int W = w * scale, Hh = h * scale;
    const int t = blockIdx.y;
    const float *sp = src + (size_t)t * h * w;
    float *dp = dst + (size_t)t * Hh * W;
    const float inv = 1.0f / (float)scale;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hh * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const float fx = (x + 0.5f) * inv - 0.5f, fy = (y + 0.5f) * inv - 0.5f;
        int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const float ax = fx - x0, ay = fy - y0;
        const int x1 = min(max(x0 + 1, 0), w - 1), y1 = min(max(y0 + 1, 0), h - 1);
        x0 = min(max(x0, 0), w - 1);
        y0 = min(max(y0, 0), h - 1);
        const float r0 = sp[y0 * w + x0] * (1.0f - ax) + sp[y0 * w + x1] * ax;
        const float r1 = sp[y1 * w + x0] * (1.0f - ax) + sp[y1 * w + x1] * ax;
        dp[i] = r0 * (1.0f - ay) + r1 * ay;
    }
}

hipError_t launch_bilinear_up(const float *src, float *dst, int tiles, int h, int w, int scale, hipStream_t s) {
    if (tiles <= 0) return hipSuccess;
    int bx = (h * scale * w * scale + 255) / 256;
    if (bx > 64) bx = 64;
    bilinear_up_kernel<<<dim3(bx, tiles), dim3(256), 0, s>>>(src, dst, h, w, scale);
    return hipGetLastError();
}

// concat_crops (sw_processing.py:113-134) in closed form per output pixel. The reference stitches
// sequentially: along a row of windows every new window j >= 1 is blended over its first
// step = window - stride columns, new = old * w[c] + crop_j[c] * (1 - w[c]) with
// w = np.linspace(1, 0, step) in float64 and the result rounded to float32 (:136-149), and its last
// `stride` columns are appended; the row strips are then stitched top to bottom the same way. A pixel
// is therefore a left fold over the <= 3 windows (rows: <= 3 strips) that cover it, which this kernel
// evaluates with the same float64 multiplies / add and the same float32 rounding points: bit-exact.
__device__ __forceinline__ float blend64(float a, float b, double w) {
    const double l = (double)a * w, r = (double)b * (1.0 - w);  // plain operators: contraction is off in this file
    return (float)(l + r);
}

__device__ __forceinline__ float strip_value(const float *__restrict__ crops, const double *__restrict__ ramp, int n,
                                             int window, int stride, int row_i, int r, int X) {
    // horizontal fold of window row `row_i` at local row r, global column X
    const int step = window - stride;
    int j0 = (X - window + stride) / stride;  // first window with X - stride*j < window
    j0 = max(j0, 0);
    while (X - stride * j0 >= window) ++j0;
    const int j1 = min(X / stride, n - 1);
    const float *base = crops + ((size_t)(row_i * n) * window + r) * window;
    float v = base[(size_t)j0 * window * window + (X - stride * j0)];
    for (int j = j0 + 1; j <= j1; ++j) {
        const int c = X - stride * j;  // < step by construction
        v = blend64(v, base[(size_t)j * window * window + c], ramp[c]);
    }
    (void)step;
    return v;
}

__global__ __launch_bounds__(256) void stitch_kernel(const float *__restrict__ crops, float *__restrict__ out,
                                                     const double *__restrict__ ramp, int n, int window, int stride) {
    const int S = window + (n - 1) * stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
        const int Y = i / S, X = i - Y * S;
        int i0 = max((Y - window + stride) / stride, 0);
        while (Y - stride * i0 >= window) ++i0;
        const int i1 = min(Y / stride, n - 1);
        float v = strip_value(crops, ramp, n, window, stride, i0, Y - stride * i0, X);
        for (int ii = i0 + 1; ii <= i1; ++ii) {
            const int r = Y - stride * ii;
            v = blend64(v, strip_value(crops, ramp, n, window, stride, ii, r, X), ramp[r]);
        }
        out[i] = v;
    }
}

hipError_t launch_stitch(const float *crops, float *out, const double *ramp, int n, int window, int stride,
                         hipStream_t s) {
    const long S = window + (long)(n - 1) * stride;
    long blocks = (S * S + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    stitch_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(crops, out, ramp, n, window, stride);
    return hipGetLastError();
}

// ---- heat-map mask: min_max_normalize (:30-35) -> * 255 -> astype(uint8) -> Otsu (:48-49, :62)
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float *__restrict__ img, size_t count,
                                                             float *__restrict__ part) {
    __shared__ float scratch[4];
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float v = img[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = block_reduce(mn, scratch, false);
    mx = block_reduce(mx, scratch, true);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = mn;
        part[2 * blockIdx.x + 1] = mx;
    }
}

__global__ __launch_bounds__(256) void normalize_u8_kernel(const float *__restrict__ img, size_t count,
                                                           const float *__restrict__ part, int nparts,
                                                           uint8_t *__restrict__ out, unsigned long long *hist) {
    __shared__ unsigned int lh[256];
    __shared__ float mm[2];
    lh[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        float mn = INFINITY, mx = -INFINITY;
        for (int i = 0; i < nparts; ++i) {
            mn = fminf(mn, part[2 * i]);
            mx = fmaxf(mx, part[2 * i + 1]);
        }
        mm[0] = mn;
        mm[1] = mx;
    }
    __syncthreads();
    const float mn = mm[0], mx = mm[1], range = mx - mn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float v = img[i];
        if (mx != mn) v = (v - mn) / range;  // min_max_normalize returns the image if flat
        v = v * 255.0f;
        const int q = (int)v;  // astype(np.uint8) of a value in [0, 255]: truncation
        const uint8_t u = (uint8_t)q;
        out[i] = u;
        atomicAdd(&lh[u], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_normalize_u8(const float *img, size_t count, float *part /*[2*256]*/, uint8_t *out,
                               unsigned long long *hist256, hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    minmax_partial_kernel<<<dim3(256), dim3(256), 0, s>>>(img, count, part);
    normalize_u8_kernel<<<dim3(1024), dim3(256), 0, s>>>(img, count, part, 256, out, hist256);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void threshold_u8_kernel(const uint8_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                           size_t count, int thresh) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        mask[i] = img[i] > thresh ? 255 : 0;  // cv2.THRESH_BINARY
}

hipError_t launch_threshold_u8(const uint8_t *img, uint8_t *mask, size_t count, int thresh, hipStream_t s) {
    threshold_u8_kernel<<<dim3(1024), dim3(256), 0, s>>>(img, mask, count, thresh);
    return hipGetLastError();
}

// ---- eval.py's per-image mask chain (eval.py:126-171 -> utils.py:61-115 threshold()) ----
// transform(img).convert("L") (eval.py:122,166): torchvision ToPILImage of a float CHW tensor is
// pic.mul(255).byte() (truncation), PIL's RGB -> L is (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
__global__ __launch_bounds__(256) void image_to_gray_u8_kernel(const float *__restrict__ img, int64_t stride_c, int chans,
                                                               size_t count, uint8_t *__restrict__ out,
                                                               unsigned long long *hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        unsigned int u;
        if (chans == 1) {
            u = (uint8_t)(int)(img[i] * 255.0f);
        } else {
            const unsigned int r = (uint8_t)(int)(img[i] * 255.0f), g = (uint8_t)(int)(img[stride_c + i] * 255.0f),
                               b = (uint8_t)(int)(img[2 * stride_c + i] * 255.0f);
            u = (19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16;
        }
        out[i] = (uint8_t)u;
        if (hist) atomicAdd(&lh[u], 1u);
    }
    __syncthreads();
    if (hist && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_image_to_gray_u8(const float *img, int64_t stride_c, int chans, size_t count, uint8_t *out,
                                   unsigned long long *hist256, hipStream_t s) {
    if (hist256) {
        hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned long long), s);
        if (e != hipSuccess) return e;
    }
    image_to_gray_u8_kernel<<<dim3(1024), dim3(256), 0, s>>>(img, stride_c, chans, count, out, hist256);
    return hipGetLastError();
}

// utils.py:79-80: result = (img / 2) * (1 - alpha) + (attention / 2) * alpha in float64 (numpy promotes
// uint8 / int to double), then astype(np.uint8) (truncation). one_minus_alpha is passed in as the host
// computed it (Python's 1 - 0.4), so the device multiplies by the very same doubles.
__global__ __launch_bounds__(256) void blend_u8_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ att,
                                                       size_t count, double alpha, double one_minus_alpha,
                                                       uint8_t *__restrict__ out, unsigned long long *hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const double a = ((double)img[i] / 2.0) * one_minus_alpha;
        const double b = ((double)att[i] / 2.0) * alpha;
        const uint8_t u = (uint8_t)(int)(a + b);
        out[i] = u;
        if (hist) atomicAdd(&lh[u], 1u);
    }
    __syncthreads();
    if (hist && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_blend_u8(const uint8_t *img, const uint8_t *att, size_t count, double alpha, double one_minus_alpha,
                           uint8_t *out, unsigned long long *hist256, hipStream_t s) {
    if (hist256) {
        hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned long long), s);
        if (e != hipSuccess) return e;
    }
    blend_u8_kernel<<<dim3(1024), dim3(256), 0, s>>>(img, att, count, alpha, one_minus_alpha, out, hist256);
    return hipGetLastError();
}
