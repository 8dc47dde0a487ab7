// jpeg_dev.h -- launch plan of the GPU pixel half of JPEG decoding (jpeg.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace oar {
namespace pp {

struct JpegDevComp {
    const int16_t* coef;   // device: bw * bh blocks of 64 quantised coefficients (natural order)
    uint8_t* plane;        // device: (bh * 8) rows of (bw * 8) samples, written by the IDCT kernel
    int h, v, bw, bh, dw, dh;
};
struct JpegDevPlan {
    JpegDevComp comp[3];
    const uint16_t* q;     // device: ncomp x 64 quantisation values (natural order)
    int ncomp, hmax, vmax, color;
    uint32_t w, h;
    long total_blocks;
};
// IDCT of every block of every component, then upsampling + colour conversion into rgb (w * h * 3 bytes, device).  Enqueued on s.
void jpeg_render(hipStream_t s, const JpegDevPlan& plan, uint8_t* rgb);

}  // namespace pp
}  // namespace oar
