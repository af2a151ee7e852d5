// srt_cli.hpp -- command-line parameters and run log of srt_render with the reference's flag names, defaults, log file
// name and log line format: io/params.h:21-304 (`parameters`, `param_manager::parseArgs`),
// _log_/log_context.{h,cpp} (`log_context`), utils/utility.h:32-41 (`string_to_filename`).
//
// Header-only and free of GPU / library calls, so that the CPU suite can hold it against the reference's own params.cpp /
// log_context.cpp compiled from their sources (tests/test_ref_host.py).
#pragma once
#include <algorithm>
#include <chrono>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace srt_cli {

typedef unsigned int uint;

inline const char *const kSceneNames[] = {"Cornell Box", "Prism World", "Different Materials"};   // io/params.h:19

struct parameters {   // io/params.h:21-223
    std::string image_title, log_subdir;
    uint scene = 0, xres = 600, yres = 600;
    float ar = 1.0f;
    uint xcsize = 0, ycsize = 0, n_samples = 500, bounce_limit = 10;
    bool do_log = false, show_render = true, do_save = false;
    // this build's additions (no reference counterpart)
    int gpu = 0;
    int gpus = 1;      // --gpus N: devices 0 .. N-1 of this node render interleaved tiles of every chunk (one RCCL gather per chunk)
    bool sah = false;
    int quirks = -1;   // --physically-correct: 0, --reference-quirks: 1, neither: -1 (library default = reference quirks on)

    parameters() { resetYres(); }
    void resetYres() { yres = static_cast<uint>(xres / ar); yres = (yres < 1) ? 1 : yres; }   // params.h:176-180
    uint getXcsize() const { uint r = xcsize == 0 ? ycsize : xcsize; return r == 0 ? xres : r; }   // :53-57
    uint getYcsize() const { uint r = ycsize == 0 ? xcsize : ycsize; return r == 0 ? yres : r; }   // :59-63
    std::string getImgTitle() const {   // :28-31 (the reference indexes its three names with any scene id; ids >= 3 are this build's)
        if (!image_title.empty()) return image_title;
        return scene < 3 ? std::string(kSceneNames[scene]) : ("Scene " + std::to_string(scene));
    }
};

inline float parseAR(const std::string &s) {   // params.h:182-195
    std::stringstream ss(s);
    std::string num;
    std::getline(ss, num, '/');
    float ar = std::stof(num);
    if (std::getline(ss, num, '/')) {
        ar /= std::stof(num);
        if (std::getline(ss, num, '/'))
            std::cout << "Characters inserted after aspect ratio's denominator will be ignored, computed AR value is: " << ar << std::endl;
    }
    return ar;
}

// param_manager::parseArgs, params.h:236-304.  A value that does not parse keeps the previous one (each setter of the reference
// catches its own exception, params.h:99-165); a flag that needs a value and stands last is an unknown argument (`!is_last`).
inline void parseArgs(int argc, char **argv, parameters &p) {
    for (int i = 1; i < argc; i++) {
        const std::string arg(argv[i]);
        const bool is_last = i + 1 == argc;
        try {
            if (!is_last && (arg == "-t" || arg == "--title")) p.image_title = argv[++i];
            else if (!is_last && (arg == "-lsub" || arg == "--log-subdir")) p.log_subdir = argv[++i];
            else if (!is_last && (arg == "-s" || arg == "--scene")) p.scene = (uint)std::stoul(argv[++i]);
            else if (!is_last && (arg == "-xr" || arg == "--xres")) { p.xres = (uint)std::stoul(argv[++i]); p.resetYres(); }
            else if (!is_last && (arg == "-ar" || arg == "--aspect-ratio")) { p.ar = parseAR(argv[++i]); p.resetYres(); }
            else if (!is_last && (arg == "-xc" || arg == "--xcsize")) p.xcsize = (uint)std::stoul(argv[++i]);
            else if (!is_last && (arg == "-yc" || arg == "--ycsize")) p.ycsize = (uint)std::stoul(argv[++i]);
            else if (!is_last && (arg == "-ns" || arg == "--nsamples")) p.n_samples = (uint)std::stoul(argv[++i]);
            else if (!is_last && (arg == "-bl" || arg == "--bounce-limit")) p.bounce_limit = (uint)std::stoul(argv[++i]);
            else if (!is_last && arg == "--gpu") p.gpu = std::stoi(argv[++i]);
            else if (!is_last && arg == "--gpus") p.gpus = std::max(1, std::stoi(argv[++i]));
            else if (arg == "--sah") p.sah = true;
            else if (arg == "--physically-correct") p.quirks = 0;      // Q1 / Q2 off (not parity-checked)
            else if (arg == "--reference-quirks") p.quirks = 1;        // the default
            else if (arg == "--dump-params") {}                        // handled by srt_render's main
            else if (arg == "--do-log") p.do_log = true;
            else if (arg == "--no-show") p.show_render = false;
            else if (arg == "--save") p.do_save = true;
            else std::cout << "Unkown argument name: " << arg << std::endl;
        } catch (...) {
            std::cerr << "Error while parsing " << arg << " arg value, keeping previous (default most likely) value" << std::endl;
        }
    }
}

inline void string_to_filename(std::string &str) {   // utils/utility.h:32-41
    for (char &c : str) c = (c == ' ') ? '_' : (char)std::tolower(static_cast<unsigned char>(c));
}

// _log_/log_context.{h,cpp} as main.cpp:135-167 uses it (append_dir(log subdir), add_title(image title), filename option
// TIMESTAMP): ordered `key: value` lines -> logs[/<subdir>]/<epoch ms>_<title>_log.txt, the file name lower-cased with blanks
// turned into underscores (to_file -> string_to_filename, log_context.cpp:9-10).
struct log_context {
    std::vector<std::string> order;               // data_insertion_order: a name added twice is listed twice ...
    std::map<std::string, std::string> data;      // ... and both lines carry its last value (log_context.cpp:69-72)
    std::string title, subdir;

    void add_entry(const std::string &name, const std::string &value) { order.push_back(name); data[name] = value; }
    void add_entry(const std::string &name, const char *value) { add_entry(name, std::string(value)); }
    // integers: to_string; float / double: digits10 + 1 significant digits (log_context.cpp:74-111)
    template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    void add_entry(const std::string &name, T value) {
        std::ostringstream oss;
        if constexpr (std::is_floating_point<T>::value) oss << std::setprecision(std::numeric_limits<T>::digits10 + 1);
        oss << value;
        add_entry(name, oss.str());
    }
    // log_context::sum_value, log_context.cpp:113-125
    void sum_value(const std::string &name, float value) {
        auto it = data.find(name);
        if (it == data.end()) { add_entry(name, value); return; }
        try { add_entry(name, (float)(std::stof(it->second) + value)); }
        catch (...) { std::cerr << "Unexpected error while trying to sum a float value to entry with name \"" << name << "\"\n leaving entry unaltered" << std::endl; }
    }
    std::string build_file_content() const {      // log_context.cpp:28-38
        std::string content;
        for (const auto &k : order) content.append(k).append(": ").append(data.at(k)).append("\n");
        return content;
    }
    std::string dir() const {                     // rel_path "logs/log.txt" + append_dir(subdir), log_context.h:69-85
        std::string d = "logs";
        if (!subdir.empty()) d.append("/").append(subdir);
        return std::filesystem::path(d + "/log.txt").parent_path().string();
    }
    std::string build_filename(long long epoch_ms) const {   // log_context.cpp:40-65 with options {TIMESTAMP, TITLE} (set order: timestamp first)
        return std::to_string(epoch_ms) + "_" + title + "_" + "log.txt";
    }
    // returns the path written, empty on failure
    std::string to_file() const {                 // log_context.cpp:5-26
        const std::string d = dir();
        std::filesystem::create_directories(d);
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        std::string name = build_filename((long long)ms);
        string_to_filename(name);
        const std::string file = std::string(d).append("/").append(name);
        std::ofstream out(file);
        if (!out.is_open()) { std::cerr << "Failed to save log file at: " << file << std::endl; return std::string(); }
        out << build_file_content();
        out.close();
        std::clog << "Log file saved successfully at: " << file << std::endl;
        return file;
    }
};

// get_image + save_img (image/image.cpp:3-18, io/save_image.cpp:8-20): renders/<filename>, written as the 24-bit BMP CImg's
// save_bmp produces for a 3-channel uchar image (54-byte header, bottom-up rows padded to 4 bytes, B G R) -- CImg replaced by this
// writer; tests/test_ref_host.py compares the two files.
inline bool save_img(const unsigned char *r, const unsigned char *g, const unsigned char *b, uint width, uint height, const std::string &filename) {
    namespace fs = std::filesystem;
    const fs::path p("renders/" + filename);
    fs::create_directories(p.parent_path());
    const uint32_t row = (width * 3 + 3) & ~3u, size = 54 + row * height;
    std::vector<unsigned char> buf(size, 0);
    auto put32 = [&](size_t off, uint32_t v) { buf[off] = v & 255; buf[off + 1] = (v >> 8) & 255; buf[off + 2] = (v >> 16) & 255; buf[off + 3] = (v >> 24) & 255; };
    buf[0] = 'B'; buf[1] = 'M';
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, width); put32(22, height);
    buf[26] = 1; buf[28] = 24; put32(34, row * height);
    buf[0x27] = 1; buf[0x2b] = 1;      // biX/YPelsPerMeter = 256, as CImg's writer fills them (found by the byte comparison with its output)
    for (uint y = 0; y < height; y++)
        for (uint x = 0; x < width; x++) {
            unsigned char *px = &buf[54 + (size_t)(height - 1 - y) * row + 3 * (size_t)x];
            const size_t k = (size_t)y * width + x;
            px[0] = b[k]; px[1] = g[k]; px[2] = r[k];
        }
    std::ofstream out(p, std::ios::binary);
    if (!out.is_open()) return false;
    out.write(reinterpret_cast<const char *>(buf.data()), (std::streamsize)buf.size());
    std::clog << "Image saved as: " << p.filename() << std::endl;
    return true;
}

}  // namespace srt_cli
