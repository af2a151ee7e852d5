// ref_host_driver.cpp -- TEST INFRASTRUCTURE (checker side), never part of the product.
//
// A harness of ours around three pieces of the REFERENCE that compile in this image exactly as they lie under
// /root/reference (no stand-ins for CUDA headers needed; oracle/Makefile target `ref` compiles the reference's own
// primitives/transform.cu, io/params.cpp and _log_/log_context.cpp unmodified and links them with this file into
// oracle/_ref/libref_host.so, git-ignored):
//   transform::assign_rot_matrix   primitives/transform.cu:4-34   (__host__ __device__: the host half is called here)
//   param_manager::parseArgs       io/params.h:236-304 (+ `parameters`, :21-223)
//   log_context                    _log_/log_context.{h,cpp}
// This file only #includes the reference's headers from where they lie and hands their results through a C ABI to
// tests/test_ref_host.py, which compares them with the product (srt_rotation_matrix, csrc/srt_cli.hpp).  Both singletons
// have private constructors but public implicit copy constructors: every call works on a copy of the pristine instance,
// so the calls are independent of each other.
#include <string.h>

#include <string>

#include "log_context.h"   // /root/reference/_log_
#include "params.h"        // /root/reference/io
#include "transform.cuh"   // /root/reference/primitives

extern "C" {

struct ref_params {
    char title[256], log_subdir[256];
    unsigned scene, xres, yres;
    float ar;
    unsigned xcsize, ycsize, n_samples, bounce_limit;   // xcsize / ycsize as getXcsize() / getYcsize() resolve them (params.h:53-63)
    int do_log, show_render, do_save;
};

__attribute__((visibility("default"))) int ref_rot_matrix(float theta, int axis, float m[9]) {
    transform::assign_rot_matrix(theta, (transform::AXIS)axis, m);
    return 0;
}

// param_manager::parseArgs on a fresh parameter set (main.cpp:135-137).  Scene ids >= 3 with an empty title make the
// reference index sceneIdToStr out of bounds (params.h:19,28-31): the title is then reported as "" instead of read.
__attribute__((visibility("default"))) int ref_parse_args(int argc, char **argv, ref_params *out) {
    static const param_manager pristine = *param_manager::getInstance();
    param_manager pm = pristine;
    pm.parseArgs(argc, argv);
    const parameters p = pm.getParams();
    memset(out, 0, sizeof(*out));
    out->scene = p.getSceneId();
    std::string title;
    if (out->scene < 3) title = p.getImgTitle();
    else {      // an explicit title is returned whatever the scene id; the default differs between ids 0 and 1
        parameters q0 = p, q1 = p;
        q0.setScene("0"); q1.setScene("1");
        if (q0.getImgTitle() == q1.getImgTitle()) title = q0.getImgTitle();
    }
    strncpy(out->title, title.c_str(), sizeof(out->title) - 1);
    strncpy(out->log_subdir, p.getLogSubdir().c_str(), sizeof(out->log_subdir) - 1);
    out->xres = p.getXres(); out->yres = p.getYres(); out->ar = p.getAR();
    out->xcsize = p.getXcsize(); out->ycsize = p.getYcsize();
    out->n_samples = p.getNSamples(); out->bounce_limit = p.getBounceLimit();
    out->do_log = p.logActive(); out->show_render = p.showRender(); out->do_save = p.doSaveImage();
    return 0;
}

// The run log exactly as main.cpp:158-160 sets it up (append_dir(log subdir), add_title(image title), filename option
// TIMESTAMP), n entries of kind 0 string, 1 unsigned int, 2 size_t, 3 int, 4 float, 5 double, 6 sum_value(float), then
// to_file() relative to the current directory.  Returns the file content through `content` as well.
__attribute__((visibility("default"))) int ref_log_to_file(const char *title, const char *subdir, int n, const char **names, const int *kinds,
                                                           const char **svals, const double *dvals, char *content, size_t cap) {
    static const log_context pristine = *log_context::getInstance();
    log_context lc = pristine;
    lc.append_dir(subdir);
    lc.add_title(title);
    lc.add_filename_option(FilenameOption::TIMESTAMP);
    for (int k = 0; k < n; k++) {
        switch (kinds[k]) {
        case 0: lc.add_entry(std::string(names[k]), std::string(svals[k])); break;
        case 1: lc.add_entry(std::string(names[k]), (unsigned int)dvals[k]); break;
        case 2: lc.add_entry(std::string(names[k]), (size_t)dvals[k]); break;
        case 3: lc.add_entry(std::string(names[k]), (int)dvals[k]); break;
        case 4: lc.add_entry(std::string(names[k]), (float)dvals[k]); break;
        case 5: lc.add_entry(std::string(names[k]), (double)dvals[k]); break;
        case 6: lc.sum_value(std::string(names[k]), (float)dvals[k]); break;
        default: return -1;
        }
    }
    const std::string c = lc.build_file_content();
    if (content && cap) { strncpy(content, c.c_str(), cap - 1); content[cap - 1] = 0; }
    lc.to_file();
    return 0;
}

}  // extern "C"
