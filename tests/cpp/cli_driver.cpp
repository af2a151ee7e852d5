// cli_driver.cpp -- test harness: the product's command-line parser and run log (csrc/srt_cli.hpp, what srt_render uses) behind the
// same C ABI as oracle/ref_host_driver.cpp exposes for the reference's io/params.h and _log_/log_context.cpp, so that
// tests/test_ref_host.py can hold one against the other.  No GPU, no library calls.
#include <string.h>

#include "../../cuda-spectral-ray-tracer_amd/csrc/srt_cli.hpp"

extern "C" {

struct ref_params {      // same layout as in oracle/ref_host_driver.cpp
    char title[256], log_subdir[256];
    unsigned scene, xres, yres;
    float ar;
    unsigned xcsize, ycsize, n_samples, bounce_limit;
    int do_log, show_render, do_save;
};

__attribute__((visibility("default"))) int cli_parse_args(int argc, char **argv, ref_params *out) {
    srt_cli::parameters p;
    srt_cli::parseArgs(argc, argv, p);
    memset(out, 0, sizeof(*out));
    // (scene ids >= 3 without a title: the reference reads past its three names; reported as "" on both sides)
    const std::string title = (p.scene < 3 || !p.image_title.empty()) ? p.getImgTitle() : std::string();
    strncpy(out->title, title.c_str(), sizeof(out->title) - 1);
    strncpy(out->log_subdir, p.log_subdir.c_str(), sizeof(out->log_subdir) - 1);
    out->scene = p.scene; out->xres = p.xres; out->yres = p.yres; out->ar = p.ar;
    out->xcsize = p.getXcsize(); out->ycsize = p.getYcsize(); out->n_samples = p.n_samples; out->bounce_limit = p.bounce_limit;
    out->do_log = p.do_log; out->show_render = p.show_render; out->do_save = p.do_save;
    return 0;
}

__attribute__((visibility("default"))) int cli_log_to_file(const char *title, const char *subdir, int n, const char **names, const int *kinds,
                                                           const char **svals, const double *dvals, char *content, size_t cap) {
    srt_cli::log_context lc;
    lc.title = title; lc.subdir = subdir;
    for (int k = 0; k < n; k++) {
        switch (kinds[k]) {
        case 0: lc.add_entry(names[k], std::string(svals[k])); break;
        case 1: lc.add_entry(names[k], (unsigned int)dvals[k]); break;
        case 2: lc.add_entry(names[k], (size_t)dvals[k]); break;
        case 3: lc.add_entry(names[k], (int)dvals[k]); break;
        case 4: lc.add_entry(names[k], (float)dvals[k]); break;
        case 5: lc.add_entry(names[k], (double)dvals[k]); break;
        case 6: lc.sum_value(names[k], (float)dvals[k]); break;
        default: return -1;
        }
    }
    const std::string c = lc.build_file_content();
    if (content && cap) { strncpy(content, c.c_str(), cap - 1); content[cap - 1] = 0; }
    return lc.to_file().empty() ? -2 : 0;
}

__attribute__((visibility("default"))) int cli_save_image(const unsigned char *r, const unsigned char *g, const unsigned char *b, unsigned width,
                                                          unsigned height, const char *filename) {
    return srt_cli::save_img(r, g, b, width, height, filename) ? 0 : -1;
}

}  // extern "C"
