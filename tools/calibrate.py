#!/usr/bin/env python3
"""Issue-rate calibration on the GPU box: runs every microkernel of csrc/srt_calib.hip at 1, 2 and 4 waves per SIMD
(one workgroup per CU) and prints / writes one JSON document.  The v_add_f32 row at 4 waves per SIMD is the peak
bench.py's roofline divides by (the render kernel runs 4 waves per SIMD)."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
NAMES = {0: "v_add_f32 x8 independent", 1: "v_pk_mul_f32 x8 independent", 2: "v_fma_f32 x8 independent", 3: "v_add_f32 dependent chain",
         4: "s_add_u32 x8 independent", 5: "v_add_f32 + s_add_u32 interleaved (32 instr = 16 VALU + 16 SALU)", 6: "v_cmp_lt_f32 + v_cndmask_b32 pairs",
         7: "ds_read_b64 lane-linear (+ lgkmcnt(0) per 8)", 8: "ds_read_b64 random 16 B records (+ lgkmcnt(0) per 8)", 9: "v_max3_f32 x8",
         10: "v_add_f32 x8, 26 of 64 lanes enabled",
         11: "buffer_load_dwordx4 x4 random 64 B records (16 MB table), 64 lanes", 12: "... 16 lanes, contiguous (4 full quads)",
         13: "... 16 lanes, one per quad", 14: "... 32 lanes, contiguous", 15: "... 32 lanes, two per quad",
         16: "buffer_load_dword x4 (same record offsets), 64 lanes", 17: "buffer_load_dwordx2 x4, 64 lanes", 18: "buffer_load_dwordx3 x4, 64 lanes",
         19: "buffer_load_dword x4, 16 lanes", 20: "buffer_load_dwordx2 x4, 16 lanes", 21: "buffer_load_dwordx3 x4, 16 lanes",
         22: "3 x dwordx4 + 1 x dword (52-byte record), 16 lanes",
         23: "v_add_f32_e64 (VOP3 encoding, two sources)", 24: "v_add_f32 + 32-bit literal (8 bytes)", 25: "v_fmac_f32 (VOP2, three reads)", 26: "v_mov_b32",
         27: "v_cndmask_b32_e64 (SGPR-pair mask)", 28: "v_cndmask_b32_e32 (VCC mask)", 29: "v_mul_f32 SGPR source (4 bytes)", 30: "v_xor_b32",
         31: "v_pk_add_f32", 32: "v_rcp_f32", 33: "v_cvt_f32_u32", 34: "v_add_f32 SDWA (8 bytes)", 35: "v_fmamk_f32 (VOP2 + literal, three reads)",
         36: "v_cmp_lt_f32 -> VCC", 37: "v_cmp_lt_f32_e64 -> SGPR pair", 38: "s_mov_b64 vcc + 8 x v_cndmask_b32_e32 (33 instr)", 39: "v_cndmask_b32_e64 with VCC as the mask", 40: "v_add_u32 SGPR source",
         41: "(v_cndmask_e32 stale VCC, v_add_f32) x4", 42: "(v_cmp, 2 x v_cndmask_e32, v_add) x2", 43: "(v_cmp, 3 x v_add, v_cndmask_e32, 3 x v_add)",
         44: "s_and vcc + 2 x v_cndmask_e32 + 6 x v_add (9 instr)", 45: "s_and vcc + 2 x v_cndmask_e64(vcc) + 6 x v_add (9 instr)", 46: "1 x v_cndmask_e32 stale VCC + 7 x v_add"}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40000)
ap.add_argument("--out", default=None)
a = ap.parse_args()
r = srt.Renderer(0)
rows = []
ap2 = a
for kind in (range(len(NAMES)) if not os.environ.get('CALIB_KINDS') else [int(x) for x in os.environ['CALIB_KINDS'].split(',')]):
    for w in (1, 2, 4):
        res = r.calibrate(kind, w, a.iters if (kind < 11 or kind >= 23) else max(1, a.iters // 40))
        res["name"] = NAMES[kind]
        rows.append(res)
        # rate = all instructions of a SIMD's waves / cycles of its LAST wave; check: rate x clock x SIMDs == chip rate
        chk = res["instr_per_cycle_per_simd"] * res["clock_ghz"] * res["n_simd"]
        print("%-70s w/SIMD %d: %.4f instr/cycle/SIMD (mean-wave-based %.4f), wave Mcycles min/mean/max %.2f/%.2f/%.2f, %.1f G wave-instr/s chip "
              "(= %.1f from rate x clock x SIMDs), clock %.3f GHz, wall %.2f ms" %
              (NAMES[kind], w, res["instr_per_cycle_per_simd"], res["instr_per_cycle_per_simd_mean_wave"], res["wave_cycles_min"] / 1e6,
               res["wave_cycles_mean"] / 1e6, res["wave_cycles_max"] / 1e6, res["instr_per_s"] / 1e9, chk, res["clock_ghz"], res["wall_ms"]), flush=True)
doc = {"device": "MI355X (gfx950)", "iters": a.iters, "rows": rows}
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(doc, open(a.out, "w"), indent=1)
