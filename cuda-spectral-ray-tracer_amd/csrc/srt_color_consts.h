// srt_color_consts.h -- the XYZ -> linear sRGB matrix (D65), one definition for the render kernel's epilogue and for
// srt_color_tables (the accessor tests/test_ref_tables.py compares with the reference's own array, utils/color_const.cu:17-19,
// compiled unmodified from the reference's sources).
#pragma once
#define SRT_XYZ2RGB_00 3.2404542f
#define SRT_XYZ2RGB_01 -1.5371385f
#define SRT_XYZ2RGB_02 -0.4985314f
#define SRT_XYZ2RGB_10 -0.9692660f
#define SRT_XYZ2RGB_11 1.8760108f
#define SRT_XYZ2RGB_12 0.0415560f
#define SRT_XYZ2RGB_20 0.0556434f
#define SRT_XYZ2RGB_21 -0.2040259f
#define SRT_XYZ2RGB_22 1.0572252f
