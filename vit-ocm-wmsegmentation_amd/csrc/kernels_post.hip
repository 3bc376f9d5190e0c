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

// Nearest-neighbour x`rep` of (T,h,w) maps: dst[t][y][x] = src[t][y / rep][x / rep] — pure index replication (what
// np.repeat / F.interpolate(mode="nearest") with an integer factor and the //8 *8 resize chain of sw_processing.py:255-257 do)
__global__ __launch_bounds__(256) void nearest_up_kernel(const float *__restrict__ src, float *__restrict__ dst, int h, int w,
                                                         int rep) {
    const int W = w * rep, Hh = h * rep;
    const float *sp = src + (size_t)blockIdx.y * h * w;
    float *dp = dst + (size_t)blockIdx.y * Hh * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hh * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        dp[i] = sp[(y / rep) * w + x / rep];
    }
}

hipError_t launch_nearest_up(const float *src, float *dst, int tiles, int h, int w, int rep, hipStream_t s) {
    if (tiles <= 0) return hipSuccess;
    int bx = (h * rep * w * rep + 255) / 256;
    if (bx > 64) bx = 64;
    nearest_up_kernel<<<dim3(bx, tiles), dim3(256), 0, s>>>(src, dst, h, w, rep);
    return hipGetLastError();
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

// ---- sw_processing.py:224-227: the stitched image the masks th / th2 are computed on ------------------------------
// output_image = concat_crops(cropped_images, ...) on the uint8 RGB windows, then Image.fromarray(...).convert("L").
// Same left fold as stitch_kernel, but numpy evaluates `uint8 * float64 + uint8 * float64` in float64 and the
// assignment into the uint8 overlap array TRUNCATES (:136-149) — a blended pixel can therefore come out one below the
// two (identical) values it blends. The windows are read in place from the float slab (what ToTensor made of the uint8
// image: x = u / 255, so u = rint(x * 255)); window pixels outside the slab are PIL crop's zeros.
__device__ __forceinline__ int slab_u8(const float *__restrict__ plane, int64_t sy, int H, int W, int y, int x) {
    if (y >= H || x >= W) return 0;
    return (int)rintf(plane[(int64_t)y * sy + x] * 255.0f);
}
__device__ __forceinline__ int blend_trunc(int a, int b, double w) {
    const double l = (double)a * w, r = (double)b * (1.0 - w);
    return (int)(unsigned char)(int)(l + r);
}
__device__ __forceinline__ int strip_value_u8(const float *__restrict__ plane, int64_t sy, int H, int W,
                                              const double *__restrict__ ramp, int n, int window, int stride, int row_i,
                                              int r, int X) {
    int j0 = max((X - window + stride) / stride, 0);
    while (X - stride * j0 >= window) ++j0;
    const int j1 = min(X / stride, n - 1);
    const int y = row_i * stride + r;
    int v = slab_u8(plane, sy, H, W, y, j0 * stride + (X - stride * j0));
    for (int j = j0 + 1; j <= j1; ++j) {
        const int c = X - stride * j;
        v = blend_trunc(v, slab_u8(plane, sy, H, W, y, j * stride + c), ramp[c]);
    }
    return v;
}

__global__ __launch_bounds__(256) void stitch_image_u8_kernel(const float *__restrict__ image, int64_t sc, int64_t sy,
                                                              int chans, int H, int W, uint8_t *__restrict__ out,
                                                              const double *__restrict__ ramp, int n, int window,
                                                              int stride, unsigned long long *hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const int S = window + (n - 1) * stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
        const int Y = i / S, X = i - Y * S;
        int i0 = max((Y - window + stride) / stride, 0);
        while (Y - stride * i0 >= window) ++i0;
        const int i1 = min(Y / stride, n - 1);
        unsigned int ch[3];
        for (int c = 0; c < chans; ++c) {
            const float *plane = image + c * sc;
            int v = strip_value_u8(plane, sy, H, W, ramp, n, window, stride, i0, Y - stride * i0, X);
            for (int ii = i0 + 1; ii <= i1; ++ii) {
                const int r = Y - stride * ii;
                v = blend_trunc(v, strip_value_u8(plane, sy, H, W, ramp, n, window, stride, ii, r, X), ramp[r]);
            }
            ch[c] = (unsigned int)v;
        }
        // .convert("RGB").convert("L"): PIL's (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (grey planes: identity)
        const unsigned int u = chans == 1 ? ch[0] : (19595u * ch[0] + 38470u * ch[1] + 7471u * ch[2] + 0x8000u) >> 16;
        out[i] = (uint8_t)u;
        if (hist) atomicAdd(&lh[u], 1u);
    }
    __syncthreads();
    if (hist && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_stitch_image_u8(const float *image, int64_t sc, int64_t sy, int chans, int H, int W, uint8_t *out,
                                  const double *ramp, int n, int window, int stride, unsigned long long *hist256,
                                  hipStream_t s) {
    if (hist256) {
        hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned long long), s);
        if (e != hipSuccess) return e;
    }
    const long S = window + (long)(n - 1) * stride;
    long blocks = (S * S + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    stitch_image_u8_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(image, sc, sy, chans, H, W, out, ramp, n, window,
                                                                         stride, hist256);
    return hipGetLastError();
}

// ---- sw_processing.py:43-46: result = (img * attention / np.max(attention)).astype(np.uint8) with attention =
// min_max_normalize(heat): uint8 * float32 -> float32 products, float32 division by the normalised map's maximum
// (1.0 unless the map is flat, then the map's own value), truncation. One pass also emits attention * 255 as uint8
// (:47-48) and both histograms.
__global__ __launch_bounds__(256) void weighted_u8_kernel(const float *__restrict__ heat, const uint8_t *__restrict__ img,
                                                          size_t count, const float *__restrict__ part, int nparts,
                                                          uint8_t *__restrict__ result, uint8_t *__restrict__ att_u8,
                                                          unsigned long long *hist_res, unsigned long long *hist_att) {
    __shared__ unsigned int lr[256], la[256];
    __shared__ float mm[2];
    lr[threadIdx.x] = 0;
    la[threadIdx.x] = 0;
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
    const float amax = mx != mn ? (mx - mn) / range : mx;  // np.max of the normalised map
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float a = heat[i];
        if (mx != mn) a = (a - mn) / range;
        const float prod = (float)img[i] * a;
        const uint8_t r = (uint8_t)(int)(prod / amax);
        const uint8_t u = (uint8_t)(int)(a * 255.0f);
        result[i] = r;
        att_u8[i] = u;
        atomicAdd(&lr[r], 1u);
        atomicAdd(&la[u], 1u);
    }
    __syncthreads();
    if (lr[threadIdx.x]) atomicAdd(&hist_res[threadIdx.x], (unsigned long long)lr[threadIdx.x]);
    if (la[threadIdx.x]) atomicAdd(&hist_att[threadIdx.x], (unsigned long long)la[threadIdx.x]);
}

hipError_t launch_weighted_u8(const float *heat, const uint8_t *img, size_t count, float *part /*[2*256]*/,
                              uint8_t *result, uint8_t *att_u8, unsigned long long *hist_res, unsigned long long *hist_att,
                              hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist_res, 0, 256 * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemsetAsync(hist_att, 0, 256 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    minmax_partial_kernel<<<dim3(256), dim3(256), 0, s>>>(heat, count, part);
    weighted_u8_kernel<<<dim3(1024), dim3(256), 0, s>>>(heat, img, count, part, 256, result, att_u8, hist_res, hist_att);
    return hipGetLastError();
}

// ---- eval.py:144,158: scipy.ndimage.median_filter(map, size=k) — k x k footprint, mode "reflect" (d c b a | a b c d |
// d c b a), origin 0 (window offsets -k/2 .. k-1-k/2), the element of rank (k*k)/2 of the sorted window. Pinned
// against scipy itself (tests/golden/median.npz, oracle/make_golden_median.py). Selection by counting: a window value
// whose number of smaller elements (ties broken by position) equals the rank is the answer; no per-thread arrays.
__device__ __forceinline__ int reflect_idx(int i, int n) {
    // scipy "reflect": ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...   (period 2n)
    const int p = 2 * n;
    i = ((i % p) + p) % p;
    return i < n ? i : p - 1 - i;
}

__global__ __launch_bounds__(256) void median_filter_kernel(const float *__restrict__ src, float *__restrict__ dst, int h,
                                                            int w, int k, size_t total) {
    const int n = k * k, rank = n / 2, lo = -(k / 2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const size_t t = i / w;
        const int y = (int)(t % h);
        const float *img = src + (t / h) * (size_t)h * w;
        float ans = 0.f;
        for (int a = 0; a < n; ++a) {
            const float va = img[(size_t)reflect_idx(y + lo + a / k, h) * w + reflect_idx(x + lo + a % k, w)];
            int less = 0;
            for (int b = 0; b < n; ++b) {
                const float vb = img[(size_t)reflect_idx(y + lo + b / k, h) * w + reflect_idx(x + lo + b % k, w)];
                less += (vb < va) || (vb == va && b < a);
            }
            if (less == rank) {
                ans = va;
                break;
            }
        }
        dst[i] = ans;
    }
}

hipError_t launch_median_filter(const float *src, float *dst, int tiles, int h, int w, int k, hipStream_t s) {
    if (tiles <= 0 || h <= 0 || w <= 0 || k < 1 || k > 15) return hipErrorInvalidValue;
    const size_t total = (size_t)tiles * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    median_filter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, dst, h, w, k, total);
    return hipGetLastError();
}

// ---- eval.py:169 / sw_processing.py:255: cv2.resize(map, (W / f, H / f)) (default INTER_LINEAR) for an even integer
// factor f: the sample point (x + 0.5) * f - 0.5 sits midway between source pixels f*x + f/2 - 1 and f*x + f/2, so
// the result is the horizontal pass 0.5 * a + 0.5 * b followed by the same vertically, in float32 (cv2's
// hresize / vresize order; cv2 is an un-vendored dependency: parity unpinned). Odd f: the centre pixel.
__global__ __launch_bounds__(256) void downscale_centre_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               int h, int w, int f, size_t total) {
    const int ho = h / f, wo = w / f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % wo);
        const size_t t = i / wo;
        const int y = (int)(t % ho);
        const float *img = src + (t / ho) * (size_t)h * w;
        if (f & 1) {
            dst[i] = img[(size_t)(y * f + f / 2) * w + x * f + f / 2];
        } else {
            const int y0 = y * f + f / 2 - 1, x0 = x * f + f / 2 - 1;
            const float r0 = img[(size_t)y0 * w + x0] * 0.5f + img[(size_t)y0 * w + x0 + 1] * 0.5f;
            const float r1 = img[(size_t)(y0 + 1) * w + x0] * 0.5f + img[(size_t)(y0 + 1) * w + x0 + 1] * 0.5f;
            dst[i] = r0 * 0.5f + r1 * 0.5f;
        }
    }
}

hipError_t launch_downscale_centre(const float *src, float *dst, int tiles, int h, int w, int f, hipStream_t s) {
    if (tiles <= 0 || h <= 0 || w <= 0 || f < 1 || h % f || w % f) return hipErrorInvalidValue;
    const size_t total = (size_t)tiles * (h / f) * (w / f);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    downscale_centre_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, dst, h, w, f, total);
    return hipGetLastError();
}

// 256-bin histogram of a uint8 image (the Otsu levels of threshold() are computed from it on the host)
__global__ __launch_bounds__(256) void histogram_u8_kernel(const uint8_t *__restrict__ img, size_t count,
                                                           unsigned long long *hist) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) atomicAdd(&lh[img[i]], 1u);
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_histogram_u8(const uint8_t *img, size_t count, unsigned long long *hist256, hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    histogram_u8_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(img, count, hist256);
    return hipGetLastError();
}
