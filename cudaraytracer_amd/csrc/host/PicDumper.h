// PicDumper: the headless frame-buffer exit of the renderer.  Converts the
// float-RGBA frame to 8 bits per channel with the reference's rule
// clamp((int)(v * 256.0f), 0, 255) on all four channels and writes a PNG
// (cudaTracer/PicDumper.h:29-55).
#ifndef RT_HOST_PICDUMPER_H
#define RT_HOST_PICDUMPER_H

#include <ctime>
#include <string>
#include <vector>

#include "Png.h"

namespace rt {

class PicDumper {
    int picWidth;
    int picHeight;

   public:
    PicDumper(int picWidth, int picHeight) : picWidth(picWidth), picHeight(picHeight) {}

    static std::vector<unsigned char> copyFloatsToCharVector(const std::vector<float>& pixels) {
        std::vector<unsigned char> img(pixels.size());
        for (size_t i = 0; i < pixels.size(); ++i) {
            const int tmp = (int)(pixels[i] * 256.0f);
            img[i] = tmp > 255 ? 255 : (tmp < 0 ? 0 : (unsigned char)tmp);
        }
        return img;
    }

    // "<w>x<h>-<unix time>.png", as the reference names its dumps
    std::string defaultName() const {
        char buf[128];
        std::snprintf(buf, sizeof buf, "%ix%i-%li.png", picWidth, picHeight, (long)std::time(nullptr));
        return buf;
    }

    // `pixels`: tight width*height*4 floats (rt_surface_download output).
    bool toDisk(const std::vector<float>& pixels, const std::string& path = "") const {
        if (pixels.size() != (size_t)picWidth * picHeight * 4) return false;
        const std::vector<unsigned char> img = copyFloatsToCharVector(pixels);
        return png::encodeFile(path.empty() ? defaultName() : path, img.data(), picWidth, picHeight);
    }
};

}  // namespace rt
#endif
