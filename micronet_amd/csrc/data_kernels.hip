// On-device CIFAR-10 training input pipeline for gfx950 -- the transform chain of the reference's training scripts
// (wqaq/dorefa/main.py:203-210: RandomCrop(32, padding=4) -> RandomHorizontalFlip -> ToTensor -> Normalize) as ONE kernel over a batch that is
// gathered straight from the uint8 dataset resident in HBM (150 MB for the 50k training images).  At 80-100 k images/s a 2-worker CPU DataLoader
// (main.py:225-236) cannot feed the step; here a batch of 256 is 786 KB of reads and 3 MB of writes: a few microseconds.
//   images : uint8 [n_images][H][W][C]  (HWC, the layout of torchvision's CIFAR10.data)
//   index  : int32 [B] sample of each output image (the epoch's shuffle);  ox, oy : int32 [B] crop offset in [0, 2*pad];  flip : uint8 [B]
//   out    : fp32 [B][C][H][W] = ((pixel / 255) - mean[c]) / std[c], pixel = 0 outside the image (RandomCrop pads with zeros BEFORE ToTensor)
// Arithmetic exactly as torchvision: ToTensor = float32(pixel).div(255); Normalize = sub(mean).div(std) -- IEEE fp32 divisions, so the result is
// bit-identical to the CPU pipeline for the same random draws.  One thread = 4 consecutive output pixels of one channel row (float4 store).
#include "common.h"

struct AugParams { int B, H, W, C, pad; int64_t n_images; float mean[4], std[4]; };

__global__ __launch_bounds__(256) void k_cifar_augment(const unsigned char* __restrict__ images, const int* __restrict__ index, const int* __restrict__ ox,
                                                       const int* __restrict__ oy, const unsigned char* __restrict__ flip, float* __restrict__ out, AugParams p) {
    const int W4 = p.W >> 2;
    const int64_t total = (int64_t)p.B * p.C * p.H * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % W4);
        int64_t t = i / W4;
        const int h = (int)(t % p.H); t /= p.H;
        const int c = (int)(t % p.C);
        const int b = (int)(t / p.C);
        // caller-supplied draws are clamped: an out-of-range image index or offset must not become an out-of-bounds read
        int64_t idx = index[b];
        idx = idx < 0 ? 0 : (idx < p.n_images ? idx : p.n_images - 1);
        int oxb = ox[b], oyb = oy[b];
        oxb = oxb < 0 ? 0 : (oxb > 2 * p.pad ? 2 * p.pad : oxb);
        oyb = oyb < 0 ? 0 : (oyb > 2 * p.pad ? 2 * p.pad : oyb);
        const unsigned char* img = images + idx * p.H * p.W * p.C;
        const int sy = h + oyb - p.pad, dx0 = oxb - p.pad;
        const bool fl = flip[b] != 0;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = 4 * q + e;
            const int wc = fl ? (p.W - 1 - w) : w;          // flip AFTER the crop: output column w shows crop column W-1-w
            const int sx = wc + dx0;
            float v = 0.f;
            if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v = (float)img[((int64_t)sy * p.W + sx) * p.C + c];
            v = v / 255.0f;
            r[e] = (v - p.mean[c]) / p.std[c];
        }
        *reinterpret_cast<float4*>(out + (((int64_t)b * p.C + c) * p.H + h) * p.W + 4 * q) = make_float4(r[0], r[1], r[2], r[3]);
    }
}
extern "C" int mn_cifar_augment(const uint8_t* images, int64_t n_images, const int32_t* index, const int32_t* ox, const int32_t* oy, const uint8_t* flip, int64_t B,
                                int64_t H, int64_t W, int64_t Cc, int pad, const float* mean, const float* stdv, float* out, mn_stream_t stream) {
    if (!images || !index || !ox || !oy || !flip || !out || !mean || !stdv || B <= 0 || n_images <= 0 || H <= 0 || W <= 0 || W % 4 || Cc < 1 || Cc > 4 || pad < 0 || !aligned16(out))
        MN_FAIL(MN_EINVAL, "mn_cifar_augment: bad arguments (W must be a multiple of 4, 1 <= C <= 4, out 16-byte aligned)");
    AugParams p;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)Cc; p.pad = pad; p.n_images = n_images;
    for (int c = 0; c < 4; ++c) { p.mean[c] = c < Cc ? mean[c] : 0.f; p.std[c] = c < Cc ? stdv[c] : 1.f; }
    const int64_t total = B * Cc * H * (W / 4);
    hipLaunchKernelGGL(k_cifar_augment, dim3(mn_grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, images, index, ox, oy, flip, out, p);
    MN_CHECK_LAUNCH("mn_cifar_augment");
    return MN_OK;
}
