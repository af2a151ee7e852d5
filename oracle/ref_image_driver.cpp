// ref_image_driver.cpp -- TEST INFRASTRUCTURE (checker side), never part of the product.
//
// Harness of ours around the reference's image output, compiled UNMODIFIED from where it lies (oracle/Makefile target `ref`):
//   get_image   image/image.cpp:3-18      planar uchar r, g, b -> CImg<unsigned char>(w, h, 1, 3)
//   save_img    io/save_image.cpp:8-20    renders/<filename> through CImg::save (format by extension: .bmp)
// The vendored include/CImg.h is used as it is (cimg_display keeps its unix default, so the object files reference Xlib and the
// library is linked against the image's own libX11; no window is ever opened).  tests/test_ref_host.py compares the file
// srt_render's writer produces for the same planes (csrc/srt_main.cpp save_img) with this one.
#include "image.h"        // /root/reference/image
#include "save_image.h"   // /root/reference/io

extern "C" __attribute__((visibility("default"))) int ref_save_image(unsigned char *r, unsigned char *g, unsigned char *b, unsigned width,
                                                                     unsigned height, const char *filename) {
    uchar_img img = get_image(r, g, b, width, height);
    save_img(img, filename);
    return 0;
}
