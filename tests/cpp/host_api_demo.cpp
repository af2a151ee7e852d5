// Drives the C++ host mirror (csrc/host_api.hpp) the way the reference's main.cpp:74-133 drives its classes:
// scene_manager -> frame_buffer -> render_manager(init_renderer, init_device_params, render_cycle, update_fb).
// usage: host_api_demo <scene_id> <xres> <yres> <spp> <bounce> <chunk_w> <chunk_h> <out.bin>
// writes 3 row-major float planes (r, g, b; values 0..255) to out.bin.
#include <cstdio>
#include <cstdlib>

#include "../../cuda-spectral-ray-tracer_amd/csrc/host_api.hpp"

using namespace srt_host;

int main(int argc, char **argv) {
    if (argc < 9) { std::fprintf(stderr, "usage: %s scene xres yres spp bounce chunk_w chunk_h out.bin\n", argv[0]); return 2; }
    const int scene_id = std::atoi(argv[1]), xres = std::atoi(argv[2]), yres = std::atoi(argv[3]);
    const uint spp = (uint)std::atoi(argv[4]), bounce = (uint)std::atoi(argv[5]);
    const uint cw = (uint)std::atoi(argv[6]), ch = (uint)std::atoi(argv[7]);

    scene_manager sm(scene_id, xres, yres);
    if (!sm.isWorldInited()) { std::fprintf(stderr, "%s\n", sm.getResultMsg().c_str()); return 1; }
    camera cam = camera_builder().setVfov(40.0f).setLookfrom(point3(278, 278, -800)).setLookat(point3(278, 278, 0)).setVup(vec3(0, 1, 0))
                     .setDefocusAngle(0.0f).setFocusDist(10.0f).setBackground(color(0, 0, 0)).setImageSize(xres, yres).getCamera();
    frame_buffer fb((size_t)xres * yres);
    render_manager rm(sm.getScene(), &cam, &fb);
    rm.init_renderer(bounce, spp);
    if (cw && ch) rm.init_device_params(cw, ch); else rm.init_device_params();
    if (!rm.isReadyToRender()) { std::fprintf(stderr, "not ready to render\n"); return 1; }
    rm.render_cycle();
    while (rm.update_fb()) {}
    rm.end_render();
    image_channels chn(fb);
    std::FILE *f = std::fopen(argv[8], "wb");
    if (!f) return 1;
    std::fwrite(fb.r, sizeof(float), fb.channel_size, f);
    std::fwrite(fb.g, sizeof(float), fb.channel_size, f);
    std::fwrite(fb.b, sizeof(float), fb.channel_size, f);
    std::fclose(f);
    std::printf("done %zu pixels, first uchar %d\n", fb.channel_size, (int)chn.r[0]);
    return (rm.isDone() && rm.getError() == SRT_OK) ? 0 : 1;
}
